"""The request service: parse -> alias -> filter endpoints -> route -> stream-proxy -> stats.

Flow and status codes follow /root/reference/src/vllm_router/services/request_service/request.py
(general path :140-302, disaggregated prefill :339-431, sleep / wake :434-510,
transcriptions :513-689).  The routing log line
``Routing request <id> with session id <sid> to <url> at <t>, process time = <dt>`` is kept
byte-compatible because end-to-end tests parse it.  Differences by design: responses are
relayed chunk by chunk without being accumulated unless a post-request callback or the
semantic cache needs the body; connection failures can fail over to another endpoint;
disaggregated prefill balances over every prefill and decode endpoint.
"""

from __future__ import annotations

import asyncio
import inspect
import json
import os
import time
import uuid

import aiohttp
from aiohttp import web

from .. import tracing
from .log import init_logger
from .utils import dumps

logger = init_logger(__name__)

_HOP = {"connection", "keep-alive", "proxy-authenticate", "proxy-authorization", "te", "trailers",
        "transfer-encoding", "upgrade", "host", "content-length"}


def _json_error(status: int, message: str, request_id: str | None = None) -> web.Response:
    headers = {"X-Request-Id": request_id} if request_id else None
    return web.json_response({"error": message}, status=status, headers=headers)


def _fwd_headers(request: web.Request, body: bytes | None = None) -> dict:
    h = {k: v for k, v in request.headers.items() if k.lower() not in _HOP}
    if body is not None:
        h["Content-Length"] = str(len(body))
    return h


async def _maybe_await(x):
    return await x if inspect.isawaitable(x) else x


def _session_id(state, request) -> str | None:
    key = getattr(state.router, "session_key", None) or state.args.session_key
    return request.headers.get(key) if key else None


async def _relay(state, request: web.Request, method: str, url: str, headers: dict, body,
                 server_url: str, request_id: str, want_body: bool,
                 media_type: str | None = None, span=None) -> tuple[web.StreamResponse, bytes | None]:
    """Stream the backend response to the client; returns (response, accumulated body?)."""
    mon = state.request_stats_monitor
    mon.on_new_request(server_url, request_id, time.time())
    acc = bytearray() if want_body else None
    resp: web.StreamResponse | None = None
    client_gone = False
    try:
        async with state.client.request(method, url, headers=headers, data=body,
                                        timeout=aiohttp.ClientTimeout(total=None)) as backend:
            out_headers = {k: v for k, v in backend.headers.items() if k.lower() not in _HOP}
            out_headers["X-Request-Id"] = request_id
            if media_type and "Content-Type" not in out_headers:
                out_headers["Content-Type"] = media_type
            resp = web.StreamResponse(status=backend.status, headers=out_headers)
            await resp.prepare(request)
            first = True
            async for chunk in backend.content.iter_any():
                if not chunk:
                    continue
                now = time.time()
                if first:
                    mon.on_request_response(server_url, request_id, now)
                    first = False
                    if span is not None:
                        span.event("first_chunk")
                else:
                    mon.on_request_chunk(server_url, request_id, now)
                if acc is not None:
                    acc.extend(chunk)
                try:
                    await resp.write(chunk)
                except ConnectionResetError:  # includes aiohttp's ClientConnectionResetError
                    # the client went away mid-stream: stop relaying; leaving the ``async with``
                    # closes the backend connection, which makes the engine abort the request
                    client_gone = True
                    break
            if not client_gone:
                await resp.write_eof()
        if client_gone:
            mon.on_request_failed(server_url, request_id, time.time())
            logger.info("client of request %s disconnected before the end of the stream", request_id)
        else:
            mon.on_request_complete(server_url, request_id, time.time())
    except Exception:
        mon.on_request_failed(server_url, request_id, time.time())
        raise
    return resp, (bytes(acc) if acc is not None else None)


async def _traced(state, request: web.Request, endpoint: str, handler) -> web.StreamResponse:
    """Run ``handler(state, request, endpoint, span)`` inside the router's server span (a child
    of the caller's ``traceparent``; the engines' spans become children of it)."""
    tracer = getattr(state, "tracer", None) or tracing.DISABLED
    span = tracer.start_span("router " + endpoint, request.headers.get("traceparent"))
    span.set("http.route", endpoint).set("http.method", request.method)
    try:
        resp = await handler(state, request, endpoint, span)
        span.set("http.status_code", resp.status)
        if resp.status >= 500:
            span.error(f"HTTP {resp.status}")
        return resp
    except BaseException as e:
        span.error(repr(e))
        raise
    finally:
        span.end()


async def route_general_request(state, request: web.Request, endpoint: str) -> web.StreamResponse:
    from .routing import DisaggregatedPrefillRouter

    if isinstance(state.router, DisaggregatedPrefillRouter):
        return await _traced(state, request, endpoint, _route_disaggregated_prefill)
    return await _traced(state, request, endpoint, _route_general)


async def _route_general(state, request: web.Request, endpoint: str, span) -> web.StreamResponse:
    in_router_time = time.time()
    request_id = request.headers.get("X-Request-Id") or str(uuid.uuid4())
    body = await request.read()
    try:
        request_json = json.loads(body)
    except (json.JSONDecodeError, UnicodeDecodeError):
        return _json_error(400, "Invalid request: body is not valid JSON.", request_id)
    if not isinstance(request_json, dict):
        return _json_error(400, "Invalid request: body must be a JSON object.", request_id)
    pinned_id = request.query.get("id")
    if "kv_transfer_params" in request_json:
        # transfer instructions are the router's to give (disaggregated prefill): a client must not
        # be able to point an engine's KV push, and the deployment's bearer token, at a URL
        del request_json["kv_transfer_params"]
        body = dumps(request_json)

    if state.callbacks is not None:
        early = await _maybe_await(state.callbacks.pre_request(request, body, request_json))
        if early is not None:
            return early

    requested_model = request_json.get("model")
    if requested_model is None:
        return _json_error(400, "Invalid request: missing 'model' in request body.", request_id)

    rewriter = state.rewriter
    if rewriter is not None and not rewriter.is_noop:
        new_body = rewriter.rewrite_request(body.decode("utf-8"), requested_model, endpoint)
        body = new_body.encode("utf-8")
        request_json = json.loads(body)

    sd = state.service_discovery
    endpoints = sd.get_endpoint_info()
    aliases = getattr(sd, "aliases", None)
    if aliases and requested_model in aliases:
        requested_model = aliases[requested_model]
        request_json["model"] = requested_model
        body = dumps(request_json)

    if pinned_id:
        candidates = [e for e in endpoints
                      if requested_model in e.model_names and e.Id == pinned_id and not e.sleep]
    else:
        candidates = [e for e in endpoints if requested_model in e.model_names and not e.sleep]
    if not candidates:
        from . import metrics

        metrics.count_unroutable(str(requested_model))
        return _json_error(
            400, f"Model {requested_model} not found or vLLM engine is sleeping.", request_id)

    engine_stats = state.engine_stats_scraper.get_engine_stats()
    request_stats = state.request_stats_monitor.get_request_stats(time.time())

    want_body = (state.callbacks is not None and hasattr(state.callbacks, "post_request")) or \
        state.semantic_cache_store is not None
    attempts = 1 + max(0, state.args.max_failover_attempts)
    last_err: Exception | None = None
    for attempt in range(attempts):
        if not candidates:
            break
        if pinned_id:
            server_url = candidates[0].url
        else:
            server_url = await _maybe_await(state.router.route_request(
                candidates, engine_stats, request_stats, request, request_json))
        curr = time.time()
        logger.info("Routing request %s with session id %s to %s at %s, process time = %.4f",
                    request_id, _session_id(state, request), server_url, curr,
                    curr - in_router_time)
        span.set("vpsa.request_id", request_id).set("vpsa.model", requested_model)
        span.set("vpsa.routing_logic", type(state.router).__name__).set("vpsa.endpoint", server_url)
        span.set("vpsa.routing_seconds", curr - in_router_time)
        fwd = _fwd_headers(request, body)
        if span.traceparent:
            fwd["traceparent"] = span.traceparent
        try:
            resp, full = await _relay(state, request, request.method, server_url + endpoint,
                                      fwd, body, server_url, request_id,
                                      want_body, "text/event-stream", span)
        except aiohttp.ClientConnectorError as e:
            # nothing was sent to the client yet: another endpoint may take the request
            last_err = e
            logger.warning("endpoint %s refused request %s: %s", server_url, request_id, e)
            candidates = [c for c in candidates if c.url != server_url]
            continue
        if full is not None:
            if state.semantic_cache_store is not None and endpoint == "/v1/chat/completions" \
                    and resp.status == 200:
                await _maybe_await(state.semantic_cache_store(request_json, full))
            if state.callbacks is not None and hasattr(state.callbacks, "post_request"):
                asyncio.get_running_loop().create_task(
                    _maybe_await(state.callbacks.post_request(request, full)))
        return resp
    return _json_error(503, f"no endpoint could serve the request: {last_err}", request_id)


# ------------------------------------------------------------------ disaggregated prefill
async def route_disaggregated_prefill_request(state, request: web.Request,
                                              endpoint: str) -> web.StreamResponse:
    return await _traced(state, request, endpoint, _route_disaggregated_prefill)


async def _route_disaggregated_prefill(state, request: web.Request, endpoint: str,
                                       span) -> web.StreamResponse:
    in_router_time = time.time()
    request_id = request.headers.get("X-Request-Id") or str(uuid.uuid4())
    try:
        request_json = await request.json()
    except Exception:
        return _json_error(400, "Invalid request: body is not valid JSON.", request_id)
    model = request_json.get("model")
    sd = state.service_discovery
    endpoints = [e for e in sd.get_endpoint_info() if not e.sleep and
                 (model is None or model in e.model_names)]
    request_stats = state.request_stats_monitor.get_request_stats(time.time())
    engine_stats = state.engine_stats_scraper.get_engine_stats()
    router = state.router
    try:
        if hasattr(router, "route_pair"):  # both hops follow the conversation
            prefill_url, decode_url = await router.route_pair(endpoints, request_stats, request_json)
        else:
            prefill_url = router.route_request(endpoints, engine_stats, request_stats, request,
                                               {"max_tokens": 1})
            decode_url = router.route_request(endpoints, engine_stats, request_stats, request,
                                              {"max_tokens": 0})
    except ValueError as e:
        return _json_error(503, str(e), request_id)

    api_key = os.environ.get("OPENAI_API_KEY") or os.environ.get("VLLM_API_KEY")
    headers = {"Content-Type": "application/json", "X-Request-Id": request_id}
    if api_key:
        headers["Authorization"] = f"Bearer {api_key}"
    if span.traceparent:
        headers["traceparent"] = span.traceparent
    span.set("vpsa.request_id", request_id).set("vpsa.prefill_endpoint", prefill_url)
    span.set("vpsa.endpoint", decode_url).set("vpsa.routing_logic", "disaggregated_prefill")
    sid = _session_id(state, request)
    mon = state.request_stats_monitor

    # phase 1: prefill replica computes the prompt's KV and ships it to the chosen decoder
    pre = dict(request_json)
    pre["max_tokens"] = 1
    if "max_completion_tokens" in pre:
        pre["max_completion_tokens"] = 1
    pre["stream"] = False
    pre.pop("stream_options", None)
    pre["kv_transfer_params"] = {"do_remote_decode": True, "decode_target": decode_url,
                                 "request_id": request_id}
    st = time.time()
    mon.on_new_request(prefill_url, request_id, st)
    try:
        async with state.client.post(prefill_url + endpoint, json=pre, headers=headers,
                                     timeout=aiohttp.ClientTimeout(total=None)) as r:
            payload = await r.read()
            if r.status >= 400:
                mon.on_request_failed(prefill_url, request_id, time.time())
                try:
                    detail = json.loads(payload)
                except Exception:
                    detail = payload.decode(errors="replace")
                return web.json_response(
                    {"error": {"message": f"Prefiller error: {detail}", "type": "prefiller_error",
                               "code": r.status}}, status=r.status,
                    headers={"X-Request-Id": request_id})
            prefill_result = json.loads(payload) if payload else {}
    except aiohttp.ClientError as e:
        mon.on_request_failed(prefill_url, request_id, time.time())
        return web.json_response(
            {"error": {"message": f"Prefiller error: {e}", "type": "prefiller_error", "code": 500}},
            status=500, headers={"X-Request-Id": request_id})
    et = time.time()
    mon.on_request_response(prefill_url, request_id, et)
    mon.on_request_complete(prefill_url, request_id, et)
    logger.info("%s prefill time (TTFT): %.4f", request_id, et - st)
    span.event("prefill_done", seconds=et - st)
    logger.info("Routing request %s with session id %s to %s at %s, process time = %.4f",
                request_id, sid, prefill_url, et, et - in_router_time)

    # phase 2: decode replica finds the KV resident and streams the answer
    dec = dict(request_json)
    kvp = (prefill_result.get("kv_transfer_params") if isinstance(prefill_result, dict) else None) or {}
    kvp.update({"do_remote_prefill": True, "prefill_source": prefill_url, "request_id": request_id})
    dec["kv_transfer_params"] = kvp
    body = dumps(dec)
    curr = time.time()
    logger.info("Routing request %s with session id %s to %s at %s, process time = %.4f",
                request_id, sid, decode_url, curr, curr - in_router_time)
    try:
        resp, _ = await _relay(state, request, "POST", decode_url + endpoint,
                               {**headers, "Content-Length": str(len(body))}, body, decode_url,
                               request_id, False, "application/json", span)
        return resp
    except aiohttp.ClientError as e:
        return web.json_response(
            {"error": {"message": f"Decoder error: {e}", "type": "decoder_error", "code": 500}},
            status=500, headers={"X-Request-Id": request_id})


# ------------------------------------------------------------------ sleep / wake
async def route_sleep_wakeup_request(state, request: web.Request, endpoint: str) -> web.Response:
    in_router_time = time.time()
    request_id = request.headers.get("X-Request-Id") or str(uuid.uuid4())
    target_id = request.query.get("id")
    if target_id is None:
        return _json_error(400, "Invalid request: missing target Engine Id.", request_id)
    sd = state.service_discovery
    target = next((e for e in sd.get_endpoint_info() if e.Id == target_id), None)
    if target is None:
        return _json_error(400, f"Engine with Id {target_id} not found.", request_id)
    logger.info("Routing request %s to engine with Id: %s", request_id, target.Id)
    headers = {}
    api_key = os.environ.get("VLLM_API_KEY")
    if api_key:
        headers["Authorization"] = f"Bearer {api_key}"
    url = target.url + endpoint
    if request.query_string:
        extra = "&".join(p for p in request.query_string.split("&") if not p.startswith("id="))
        if extra:
            url += "?" + extra
    try:
        if endpoint == "/is_sleeping":
            async with state.client.get(url, headers=headers) as r:
                return web.json_response(await r.json(), status=r.status,
                                         headers={"X-Request-Id": request_id})
        async with state.client.post(url, headers=headers) as r:
            ok = r.status == 200
            status = r.status
    except aiohttp.ClientError as e:
        return _json_error(503, f"Failed to reach engine {target_id}: {e}", request_id)
    if ok:
        sleeping = endpoint == "/sleep"
        if hasattr(sd, "set_sleep_label") and (target.pod_name or target.service_name):
            await sd.set_sleep_label(target.pod_name or target.service_name, sleeping)
        elif hasattr(sd, "set_sleep"):
            sd.set_sleep(target.Id, sleeping)
        else:
            target.sleep = sleeping
    logger.info("%s request %s handled in %.4fs", endpoint, request_id, time.time() - in_router_time)
    return web.json_response({"status": "success" if ok else "failed"}, status=status,
                             headers={"X-Request-Id": request_id})


# ------------------------------------------------------------------ audio transcriptions
async def route_general_transcriptions(state, request: web.Request, endpoint: str) -> web.StreamResponse:
    in_router_time = time.time()
    request_id = request.headers.get("X-Request-Id") or str(uuid.uuid4())
    try:
        reader = await request.multipart()
    except Exception:
        return _json_error(400, "Invalid request: expected multipart/form-data.", request_id)
    fields: dict[str, str] = {}
    file_tuple = None
    async for part in reader:
        if part.name == "file":
            data = await part.read(decode=False)
            file_tuple = (part.filename or "audio.wav", data,
                          part.headers.get("Content-Type", "application/octet-stream"))
        else:
            fields[part.name] = (await part.read(decode=False)).decode(errors="replace")
    if file_tuple is None:
        return _json_error(400, "Invalid request: missing 'file' in form data.", request_id)
    model = fields.get("model")
    if not model:
        return _json_error(400, "Invalid request: missing 'model' in form data.", request_id)
    sd = state.service_discovery
    endpoints = [e for e in sd.get_endpoint_info() if model in e.model_names and not e.sleep]
    if not endpoints:
        return _json_error(404, f"Model {model} not found or vLLM engine is sleeping.", request_id)
    engine_stats = state.engine_stats_scraper.get_engine_stats()
    request_stats = state.request_stats_monitor.get_request_stats(time.time())
    server_url = await _maybe_await(state.router.route_request(
        endpoints, engine_stats, request_stats, request, {"model": model}))
    curr = time.time()
    logger.info("Routing request %s with session id %s to %s at %s, process time = %.4f",
                request_id, _session_id(state, request), server_url, curr, curr - in_router_time)
    form = aiohttp.FormData()
    form.add_field("file", file_tuple[1], filename=file_tuple[0], content_type=file_tuple[2])
    for k, v in fields.items():
        form.add_field(k, v)
    headers = {}
    if "Authorization" in request.headers:
        headers["Authorization"] = request.headers["Authorization"]
    mon = state.request_stats_monitor
    mon.on_new_request(server_url, request_id, time.time())
    try:
        async with state.client.post(server_url + endpoint, data=form, headers=headers,
                                     timeout=aiohttp.ClientTimeout(total=None)) as r:
            payload = await r.read()
            now = time.time()
            mon.on_request_response(server_url, request_id, now)
            mon.on_request_complete(server_url, request_id, now)
            return web.Response(body=payload, status=r.status,
                                content_type=r.headers.get("Content-Type", "application/json").split(";")[0],
                                headers={"X-Request-Id": request_id})
    except aiohttp.ClientError as e:
        mon.on_request_failed(server_url, request_id, time.time())
        return _json_error(503, f"Failed to connect to backend: {e}", request_id)
