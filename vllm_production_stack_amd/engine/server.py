"""OpenAI-compatible HTTP front of one engine replica (aiohttp).

Implements the endpoint, metric and auth contract the router, the LoRA operator and the
probes of the reference consume from an engine process
(/root/reference/src/vllm_router/routers/main_router.py:50-246,
/root/reference/src/vllm_router/service_discovery.py:414-563,
/root/reference/operator/internal/controller/loraadapter_controller.go:582-610).
"""

from __future__ import annotations

import asyncio
import hmac
import json
import logging
import time
import uuid

from aiohttp import web

from .. import __version__
from ..tracing import Tracer
from .async_engine import AsyncLLMEngine
from .llm_engine import EngineBusyError
from .request import SamplingParams
from .tokenizer import IncrementalDetokenizer
from .tools import parse_tool_calls

log = logging.getLogger("vpsa.server")

_OPEN_PATHS = {"/health", "/metrics", "/version", "/ping"}


def _err(status: int, message: str, etype: str = "BadRequestError") -> web.Response:
    return web.json_response(
        {"object": "error", "message": message, "type": etype, "param": None, "code": status},
        status=status)


def _sampling_from_body(body: dict, default_max: int) -> SamplingParams:
    mt = body.get("max_completion_tokens") or body.get("max_tokens") or default_max
    stop = body.get("stop") or []
    if isinstance(stop, str):
        stop = [stop]
    lp = body.get("logprobs")
    if lp is True:
        lp = body.get("top_logprobs") or 1
    elif lp is False:
        lp = None
    return SamplingParams(
        max_tokens=int(mt), temperature=float(body.get("temperature", 1.0) if body.get("temperature") is not None else 1.0),
        top_p=float(body.get("top_p") or 1.0), top_k=int(body.get("top_k") or -1),
        seed=body.get("seed"), stop=list(stop), stop_token_ids=list(body.get("stop_token_ids") or []),
        ignore_eos=bool(body.get("ignore_eos", False)), logprobs=lp, n=int(body.get("n") or 1),
        presence_penalty=float(body.get("presence_penalty") or 0.0),
        frequency_penalty=float(body.get("frequency_penalty") or 0.0),
        repetition_penalty=float(body.get("repetition_penalty") or 1.0),
        min_tokens=int(body.get("min_tokens") or 0))


class EngineServer:
    def __init__(self, aengine: AsyncLLMEngine):
        self.ae = aengine
        self.engine = aengine.engine
        self.cfg = self.engine.cfg
        self.created = int(time.time())
        self.app = web.Application(client_max_size=256 * 1024 * 1024, middlewares=[self._auth])
        r = self.app.router
        r.add_get("/health", self.health)
        r.add_get("/ping", self.health)
        r.add_get("/version", self.version)
        r.add_get("/metrics", self.metrics)
        r.add_get("/v1/models", self.models)
        r.add_post("/v1/chat/completions", self.chat)
        r.add_post("/v1/completions", self.completions)
        r.add_post("/v1/embeddings", self.embeddings)
        r.add_post("/tokenize", self.tokenize)
        r.add_post("/detokenize", self.detokenize)
        for p in ("/v1/rerank", "/rerank", "/v2/rerank"):
            r.add_post(p, self.rerank)
        for p in ("/v1/score", "/score"):
            r.add_post(p, self.score)
        r.add_post("/v1/audio/transcriptions", self.transcriptions)
        r.add_post("/sleep", self.sleep)
        r.add_post("/wake_up", self.wake_up)
        r.add_get("/is_sleeping", self.is_sleeping)
        r.add_post("/v1/load_lora_adapter", self.load_lora)
        r.add_post("/v1/unload_lora_adapter", self.unload_lora)
        r.add_post("/reset_prefix_cache", self.reset_prefix_cache)
        r.add_get("/kv/lookup", self.kv_lookup)
        r.add_post("/kv/lookup", self.kv_lookup)
        r.add_get("/debug/stats", self.debug_stats)
        r.add_post("/kv/reserve", self.kv_reserve)
        r.add_post("/kv/push", self.kv_push)
        r.add_post("/kv/commit", self.kv_commit)
        r.add_post("/kv/rccl_init", self.kv_rccl_init)
        r.add_post("/kv/rccl_recv", self.kv_rccl_recv)
        self._receiver = None
        # spans go to the collector named by the standard OTEL_* environment; off without it
        self.tracer = Tracer.from_env("vpsa-engine")
        self.app.on_cleanup.append(self._close_tracer)

    async def _close_tracer(self, app):
        await asyncio.get_running_loop().run_in_executor(None, self.tracer.shutdown)

    def _request_span(self, request, rid, ids, params, model):
        """Server span of one generation request, child of the caller's ``traceparent``."""
        span = self.tracer.start_span("llm_request", request.headers.get("traceparent"))
        if span.trace_id:
            span.set("gen_ai.request.id", rid).set("gen_ai.response.model", model)
            span.set("gen_ai.request.max_tokens", params.max_tokens)
            span.set("gen_ai.request.temperature", float(params.temperature))
            span.set("gen_ai.request.top_p", float(params.top_p))
            span.set("gen_ai.usage.prompt_tokens", len(ids)).set("http.route", request.path)
        return span

    @staticmethod
    def _end_span(span, t0, t_first, n_out, cached, queue_time, reason, error=None):
        if not span.trace_id:
            return
        now = time.time()
        span.set("gen_ai.usage.completion_tokens", n_out).set("gen_ai.usage.cached_tokens", cached)
        span.set("gen_ai.latency.time_in_queue", float(queue_time))
        if t_first is not None:
            span.set("gen_ai.latency.time_to_first_token", t_first - t0)
        span.set("gen_ai.latency.e2e", now - t0).set("gen_ai.response.finish_reason", reason)
        if error:
            span.error(error)
        span.end()

    # ------------------------------------------------------------------ middleware
    @web.middleware
    async def _auth(self, request: web.Request, handler):
        key = self.cfg.api_key
        if key and request.path not in _OPEN_PATHS:
            got = request.headers.get("Authorization", "").encode("utf-8", "replace")
            if not hmac.compare_digest(got, f"Bearer {key}".encode()):
                return web.json_response({"error": "Unauthorized"}, status=401)
        return await handler(request)

    # ------------------------------------------------------------------ small endpoints
    async def health(self, request):
        if not self.ae.healthy:
            return web.Response(status=503)
        return web.Response(status=200)

    async def version(self, request):
        return web.json_response({"version": __version__})

    async def metrics(self, request):
        return web.Response(body=self.engine.metrics.render(),
                            headers={"Content-Type": "text/plain; version=0.0.4; charset=utf-8"})

    def _model_cards(self) -> list[dict]:
        base = {"id": self.cfg.model_name, "object": "model", "created": self.created,
                "owned_by": "vllm", "root": self.cfg.model, "parent": None,
                "max_model_len": self.engine.max_model_len}
        cards = [base]
        lm = self.engine.lora_manager
        if lm is not None:
            for name, path in lm.list_adapters():
                cards.append({"id": name, "object": "model", "created": self.created,
                              "owned_by": "vllm", "root": path, "parent": self.cfg.model_name})
        return cards

    async def models(self, request):
        return web.json_response({"object": "list", "data": self._model_cards()})

    def _resolve_model(self, name: str | None):
        """-> (ok, lora_name)"""
        if not name or name == self.cfg.model_name or name == self.cfg.model:
            return True, None
        lm = self.engine.lora_manager
        if lm is not None and lm.has(name):
            return True, name
        return False, None

    async def tokenize(self, request):
        body = await request.json()
        tok = self.engine.tokenizer
        if "messages" in body:
            ids = tok.encode_chat(body["messages"], body.get("add_generation_prompt", True))
        else:
            ids = tok.encode(body.get("prompt", ""), body.get("add_special_tokens", True))
        return web.json_response({"count": len(ids), "max_model_len": self.engine.max_model_len,
                                  "tokens": ids})

    async def detokenize(self, request):
        body = await request.json()
        return web.json_response({"prompt": self.engine.tokenizer.decode(body.get("tokens", []))})

    # ------------------------------------------------------------------ sleep mode
    async def sleep(self, request):
        if not self.cfg.enable_sleep_mode:
            return _err(400, "sleep mode is not enabled (--enable-sleep-mode)")
        level = int(request.query.get("level", "1"))
        t0 = time.time()
        try:
            freed = await self.ae.call(lambda: self.engine.sleep(level))
        except EngineBusyError as e:
            return _err(409, str(e), "Conflict")
        log.info("Sleep mode freed %.2f GiB memory; it took %.6f seconds to fall asleep.",
                 freed / 2**30, time.time() - t0)
        return web.Response(status=200)

    async def wake_up(self, request):
        if not self.cfg.enable_sleep_mode:
            return _err(400, "sleep mode is not enabled (--enable-sleep-mode)")
        await self.ae.call(self.engine.wake_up)
        return web.Response(status=200)

    async def is_sleeping(self, request):
        return web.json_response({"is_sleeping": bool(self.engine.is_sleeping)})

    async def reset_prefix_cache(self, request):
        n = await self.ae.call(self.engine.bm.reset_prefix_cache)
        return web.json_response({"released_pages": n})

    async def kv_lookup(self, request):
        """How many leading tokens of a prompt are resident in this replica's KV tiers."""
        if request.method == "POST":
            body = await request.json()
        else:
            body = dict(request.query)
        tokens = body.get("tokens")
        if tokens is None:
            tokens = self.engine.tokenizer.encode(body.get("prompt", ""))
        from .block_manager import PAGE, chain_hashes

        hashes = chain_hashes(list(tokens))
        n = self.engine.bm.peek_prefix(hashes)
        if self.engine.connector is not None and n < len(hashes):
            n += self.engine.connector.lookup(hashes[n:])
        return web.json_response({"matched_tokens": n * PAGE, "total_tokens": len(tokens)})

    async def debug_stats(self, request):
        e = self.engine
        out = {"prof": e.prof, "preemptions": e.scheduler.num_preemptions,
               "prefix_hits": e.bm.hits, "prefix_queries": e.bm.queries,
               "free_pages": e.bm.num_free(), "graphs": sorted(e.runner._graphs, key=str),
               "page_bytes": e.runner.page_bytes}
        if e.connector is not None:
            out["pd_push"] = e.connector.pd_stats
            out["kv_tiers"] = e.connector.tier_stats()
        return web.json_response(out)

    # ------------------------------------------------------------------ P->D KV receive side
    def _kv_receiver(self):
        if self._receiver is None:
            from ..parallel.kv_transfer import KVReceiver

            self._receiver = KVReceiver(self.engine)
        return self._receiver

    async def kv_reserve(self, request):
        rcv = self._kv_receiver()
        try:
            d = await request.json()
            res = await self.ae.call(lambda: rcv.reserve(d["hashes"], int(d["page_bytes"]), d))
        except (ValueError, KeyError, TypeError, MemoryError) as e:
            return _err(409, str(e), "KVTransferError")
        return web.json_response(res)

    async def kv_push(self, request):
        rcv = self._kv_receiver()
        try:
            raw = await request.read()
            n = int.from_bytes(raw[:4], "little")
            meta = json.loads(raw[4:4 + n])
            pages = meta["pages"]
            if not isinstance(pages, list):
                raise ValueError("pages: a list of page indices")
            res = await self.ae.call(lambda: rcv.push(pages, raw[4 + n:]))
        except (ValueError, KeyError, TypeError) as e:  # malformed body or pages that are not reserved
            return _err(409, str(e), "KVTransferError")
        return web.json_response(res)

    async def kv_rccl_init(self, request):
        d = await request.json()
        rcv = self._kv_receiver()
        try:
            # joins the communicator on a worker thread: the engine loop keeps serving while the
            # prefill replica's side of the rendezvous arrives
            res = await asyncio.get_running_loop().run_in_executor(None, rcv.rccl_init, d["id"])
        except (ValueError, RuntimeError) as e:
            return _err(409, str(e), "KVTransferError")
        return web.json_response(res)

    async def kv_rccl_recv(self, request):
        rcv = self._kv_receiver()
        try:
            d = await request.json()
            res = await self.ae.call(lambda: rcv.rccl_recv(d["comm"], d["pages"], int(d["page_bytes"])))
        except (KeyError, ValueError, TypeError, RuntimeError) as e:
            return _err(409, str(e), "KVTransferError")
        return web.json_response(res)

    async def kv_commit(self, request):
        """Publish transferred pages.  The engine thread only enqueues the window -> pool copy and,
        once its event has fired, registers the pages; the wait for the event runs on a worker
        thread with a deadline (a dead RCCL sender cannot park the step loop)."""
        rcv = self._kv_receiver()
        try:
            d = await request.json()
            hashes, pages = d["hashes"], d["pages"]
            ev = await self.ae.call(lambda: rcv.commit_begin(hashes, pages, bool(d.get("landed")),
                                                             bool(d.get("rccl"))))
            if ev is not None:
                from ..parallel.kv_transfer import wait_event

                arrived = await asyncio.get_running_loop().run_in_executor(None, wait_event, ev, 10.0)
                if not arrived:
                    await self.ae.call(lambda: rcv.commit_abort(hashes))
                    return _err(409, "transferred KV pages did not arrive within 10 s", "KVTransferError")
            res = await self.ae.call(lambda: rcv.commit_finish(hashes, pages))
        except (KeyError, ValueError, TypeError, RuntimeError) as e:
            return _err(409, str(e), "KVTransferError")
        return web.json_response(res)

    # ------------------------------------------------------------------ LoRA
    async def load_lora(self, request):
        if not self.cfg.enable_lora:
            return _err(400, "LoRA is not enabled (--enable-lora)")
        body = await request.json()
        name, path = body.get("lora_name"), body.get("lora_path")
        if not name or not path:
            return _err(400, "lora_name and lora_path are required")
        lm = self.engine.lora_manager
        if lm.has(name):
            return _err(400, f"The lora adapter '{name}' has already been loaded.")
        try:
            await self.ae.call(lambda: lm.load(name, path))
        except FileNotFoundError as e:
            return _err(404, str(e), "NotFoundError")
        except Exception as e:
            return _err(400, f"failed to load adapter: {e}")
        return web.Response(text=f"Success: LoRA adapter '{name}' added successfully.")

    async def unload_lora(self, request):
        if not self.cfg.enable_lora:
            return _err(400, "LoRA is not enabled (--enable-lora)")
        body = await request.json()
        name = body.get("lora_name")
        lm = self.engine.lora_manager
        if not name or not lm.has(name):
            return _err(404, f"The lora adapter '{name}' cannot be found.", "NotFoundError")
        await self.ae.call(lambda: lm.unload(name))
        return web.Response(text=f"Success: LoRA adapter '{name}' removed successfully.")

    # ------------------------------------------------------------------ generation core
    async def _run_text(self, rid, ids, params, lora, kvp=None, span=None):
        """Collect a whole generation. -> (text, out_ids, finish_reason, usage dict)"""
        detok = IncrementalDetokenizer(self.engine.tokenizer)
        t0 = time.time()
        try:
            return await self._collect(rid, ids, params, lora, kvp, span, detok, t0)
        except BaseException as e:
            if span is not None:
                self._end_span(span, t0, None, 0, 0, 0.0, "error", error=repr(e))
            raise

    async def _await_kv_push(self, rid, kvp) -> None:
        """Prefill half of a disaggregated request: answer only when the pages are published on
        the decode replica (or the transfer has failed: the decode replica then recomputes)."""
        conn = self.engine.connector
        if conn is None or not (kvp or {}).get("do_remote_decode"):
            return
        fut = conn.push_future(rid)
        if fut is None:
            return
        try:
            await asyncio.wait_for(asyncio.wrap_future(fut), timeout=35.0)
        except Exception as e:  # logged by the connector's callback as well
            log.warning("KV push of %s did not complete: %s", rid, e)

    async def _collect(self, rid, ids, params, lora, kvp, span, detok, t0):
        text = ""
        out_ids: list[int] = []
        lp_infos: list = []
        reason = None
        last = None
        t_first = None
        async for so in self.ae.generate(rid, ids, params, lora, kvp):
            last = so
            if t_first is None and so.new_token_ids:
                t_first = time.time()
            if so.logprobs:
                lp_infos.extend(so.logprobs)
            new = [t for t in so.new_token_ids if not (so.finished and so.finish_reason == "stop"
                                                       and t in self.engine.eos_ids)]
            out_ids.extend(so.new_token_ids)
            text += detok.push(new)
            hit = self._stop_hit(text, params.stop)
            if hit is not None:
                text = text[:hit]
                reason = "stop"
                break
            if so.finished:
                reason = so.finish_reason
        await self._await_kv_push(rid, kvp)
        usage = {"prompt_tokens": len(ids), "completion_tokens": len(out_ids),
                 "total_tokens": len(ids) + len(out_ids)}
        if last is not None and last.num_cached_tokens:
            usage["prompt_tokens_details"] = {"cached_tokens": last.num_cached_tokens}
        if span is not None:
            self._end_span(span, t0, t_first, len(out_ids), last.num_cached_tokens if last else 0,
                           last.queue_time if last else 0.0, reason or "stop")
        if lp_infos:
            usage = dict(usage, _logprobs=lp_infos[:len(out_ids)])  # taken out again by the caller
        return text, out_ids, reason or "stop", usage

    # ------------------------------------------------------------------ logprobs
    def _tok_text(self, tid: int) -> str:
        try:
            return self.engine.tokenizer.decode([int(tid)])
        except Exception:
            return ""

    def _chat_logprob_items(self, token_ids: list[int], infos: list) -> list[dict]:
        """OpenAI chat shape: one entry per generated token with its alternatives."""
        items = []
        for tid, info in zip(token_ids, infos):
            if info is None:
                continue
            text = self._tok_text(tid)
            items.append({
                "token": text, "logprob": info["token_logprob"], "bytes": list(text.encode("utf-8")),
                "top_logprobs": [{"token": self._tok_text(a), "logprob": lp,
                                  "bytes": list(self._tok_text(a).encode("utf-8"))}
                                 for a, lp in zip(info["top_ids"], info["top_logprobs"])]})
        return items

    def _completion_logprobs(self, token_ids: list[int], infos: list, offset0: int = 0) -> dict:
        """OpenAI completions shape: parallel lists."""
        toks, lps, tops, offs = [], [], [], []
        off = offset0
        for tid, info in zip(token_ids, infos):
            if info is None:
                continue
            text = self._tok_text(tid)
            toks.append(text)
            lps.append(info["token_logprob"])
            tops.append({self._tok_text(a): lp for a, lp in zip(info["top_ids"], info["top_logprobs"])})
            offs.append(off)
            off += len(text)
        return {"tokens": toks, "token_logprobs": lps, "top_logprobs": tops, "text_offset": offs}

    @staticmethod
    def _stop_hit(text: str, stops: list[str]) -> int | None:
        best = None
        for s in stops:
            if s:
                i = text.find(s)
                if i >= 0 and (best is None or i < best):
                    best = i
        return best

    @staticmethod
    def _jobs(rid: str, ids_list: list[list[int]], params: SamplingParams):
        """One generation per (prompt, sample): [(choice index, request id, prompt ids, params)].
        Sample j of a seeded request uses seed + j, so the n choices differ and reproduce."""
        import copy

        n = max(1, int(params.n or 1))
        jobs = []
        for i, ids in enumerate(ids_list):
            for j in range(n):
                index = i * n + j
                p = params
                if n > 1:
                    p = copy.copy(params)
                    p.n = 1
                    if params.seed is not None:
                        p.seed = int(params.seed) + j
                jobs.append((index, rid if len(ids_list) * n == 1 else f"{rid}-{index}", ids, p))
        return jobs

    async def _stream(self, request, rid, jobs, lora, kind, model, include_usage, kvp=None, parse_tools=False):
        """SSE response; ``jobs`` as built by ``_jobs`` (several choices interleave their chunks,
        each tagged with its index)."""
        first_ids, first_params = jobs[0][2], jobs[0][3]
        span = self._request_span(request, rid, first_ids, first_params, model)
        t0, failure = time.time(), None
        tot = {"n_out": 0, "cached": 0, "queue_time": 0.0, "t_first": None, "reason": None}
        resp = web.StreamResponse(status=200, headers={
            "Content-Type": "text/event-stream", "Cache-Control": "no-cache", "X-Request-Id": rid})
        try:
            await resp.prepare(request)
        except (ConnectionError, OSError):  # ClientConnectionResetError is one: gone before the headers
            log.debug("client of %s hung up before the response began", rid)
            self._end_span(span, t0, None, 0, 0, 0.0, "abort", "client disconnected")
            return resp
        created = int(time.time())
        cid = ("chatcmpl-" if kind == "chat" else "cmpl-") + rid
        obj = "chat.completion.chunk" if kind == "chat" else "text_completion"
        wlock = asyncio.Lock()

        async def send(data: bytes):
            async with wlock:
                await resp.write(data)

        def chunk(index=0, delta_text=None, finish=None, role=False, usage=None, choices=True, logprobs=None,
                  tool_calls=None):
            d = {"id": cid, "object": obj, "created": created, "model": model}
            if choices:
                if kind == "chat":
                    delta = {}
                    if role:
                        delta["role"] = "assistant"
                        delta["content"] = ""
                    if delta_text is not None:
                        delta["content"] = delta_text
                    if tool_calls is not None:
                        delta["tool_calls"] = [dict(c, index=i) for i, c in enumerate(tool_calls)]
                    d["choices"] = [{"index": index, "delta": delta, "logprobs": logprobs,
                                     "finish_reason": finish}]
                else:
                    d["choices"] = [{"index": index, "text": delta_text or "", "logprobs": logprobs,
                                     "finish_reason": finish}]
            else:
                d["choices"] = []
            if usage is not None:
                d["usage"] = usage
            return b"data: " + json.dumps(d, separators=(",", ":")).encode() + b"\n\n"

        async def run_choice(index, crid, ids, params):
            detok = IncrementalDetokenizer(self.engine.tokenizer)
            pending_lp: list = []
            pending_lp_c: dict = {}
            lp_offset = 0
            text = ""
            sent = 0
            reason = None
            if kind == "chat":
                await send(chunk(index, role=True))
            async for so in self.ae.generate(crid, ids, params, lora, kvp):
                tot["n_out"] += len(so.new_token_ids)
                tot["cached"] = max(tot["cached"], so.num_cached_tokens)
                tot["queue_time"] = max(tot["queue_time"], so.queue_time)
                if tot["t_first"] is None and so.new_token_ids:
                    tot["t_first"] = time.time()
                if so.logprobs:  # sent with the next text delta (text may lag behind its tokens)
                    if kind == "chat":
                        pending_lp.extend(self._chat_logprob_items(so.new_token_ids, so.logprobs))
                    else:
                        part = self._completion_logprobs(so.new_token_ids, so.logprobs, lp_offset)
                        lp_offset += sum(len(t) for t in part["tokens"])
                        for k2, v2 in part.items():
                            pending_lp_c.setdefault(k2, []).extend(v2)
                new = [t for t in so.new_token_ids
                       if not (so.finished and so.finish_reason == "stop" and t in self.engine.eos_ids)]
                text += detok.push(new)
                hit = self._stop_hit(text, params.stop)
                if hit is not None:
                    text = text[:hit]
                    reason = "stop"
                elif so.finished:
                    reason = so.finish_reason
                # hold back a tail that could still turn into a stop string; with tool parsing the
                # whole answer is held: it is either content or a tool call, known only at the end
                hold = max((len(x) - 1 for x in params.stop), default=0) if reason is None else 0
                upto = max(sent, len(text) - hold)
                if parse_tools:
                    upto = sent
                if upto > sent:
                    lp_out = None
                    if pending_lp:
                        lp_out, pending_lp = {"content": pending_lp}, []
                    elif pending_lp_c:
                        lp_out, pending_lp_c = pending_lp_c, {}
                    await send(chunk(index, delta_text=text[sent:upto], logprobs=lp_out))
                    sent = upto
                if reason is not None:
                    break
            tail_lp = {"content": pending_lp} if pending_lp else (pending_lp_c or None)
            if parse_tools:
                calls, rest = parse_tool_calls(text, self.cfg.tool_call_parser)
                if calls:
                    if rest:
                        await send(chunk(index, delta_text=rest))
                    await send(chunk(index, tool_calls=calls, logprobs=tail_lp))
                    await send(chunk(index, delta_text=None, finish="tool_calls"))
                    tot["reason"] = "tool_calls"
                    return
                if text:
                    await send(chunk(index, delta_text=text, logprobs=tail_lp))
                    tail_lp = None
            await send(chunk(index, delta_text="", finish=reason or "stop", logprobs=tail_lp))
            tot["reason"] = reason or "stop"

        tasks = [asyncio.ensure_future(run_choice(*j)) for j in jobs]
        try:
            await asyncio.gather(*tasks)
            if include_usage:
                n_prompt = sum(len(j[2]) for j in jobs if j[0] % max(1, int(first_params.n or 1)) == 0) \
                    if len(jobs) > 1 else len(first_ids)
                usage = {"prompt_tokens": n_prompt, "completion_tokens": tot["n_out"],
                         "total_tokens": n_prompt + tot["n_out"]}
                if tot["cached"]:
                    usage["prompt_tokens_details"] = {"cached_tokens": tot["cached"]}
                await send(chunk(usage=usage, choices=False))
            await send(b"data: [DONE]\n\n")
        except (ConnectionResetError, asyncio.CancelledError) as gone:
            for t in tasks:
                t.cancel()  # generate()'s finally aborts the engine requests
            self._end_span(span, t0, tot["t_first"], tot["n_out"], tot["cached"], tot["queue_time"], "abort",
                           error="client disconnected")
            if isinstance(gone, asyncio.CancelledError):
                raise
            # the peer hung up mid-stream (a load generator at its time limit, a router failing
            # over): not an error of this server, and there is nobody left to answer
            log.debug("client of %s disconnected after %d tokens", rid, tot["n_out"])
            return resp
        except Exception as e:
            for t in tasks:
                t.cancel()
            log.exception("stream failed")
            failure = repr(e)
            await resp.write(b"data: " + json.dumps({"error": str(e)}).encode() + b"\n\n")
        self._end_span(span, t0, tot["t_first"], tot["n_out"], tot["cached"], tot["queue_time"],
                       tot["reason"] or "stop", error=failure)
        await resp.write_eof()
        return resp

    async def _run_jobs(self, request, rid, jobs, lora, model, kvp):
        """Non-streaming: run every job, -> [(index, text, out_ids, reason, usage, logprob infos)]
        in index order plus the summed usage."""
        async def one(index, crid, ids, params):
            span = self._request_span(request, crid, ids, params, model)
            text, out_ids, reason, usage = await self._run_text(crid, ids, params, lora, kvp, span)
            return index, text, out_ids, reason, usage, usage.pop("_logprobs", None)

        results = await asyncio.gather(*[one(*j) for j in jobs])
        n = max(1, int(jobs[0][3].n or 1)) if len(jobs) == 1 else None
        per_prompt = {}
        for index, crid, ids, params in jobs:
            per_prompt.setdefault(id(ids), len(ids))
        n_prompt = sum(per_prompt.values())
        n_out = sum(len(r[2]) for r in results)
        usage = {"prompt_tokens": n_prompt, "completion_tokens": n_out, "total_tokens": n_prompt + n_out}
        cached = max((r[4].get("prompt_tokens_details", {}).get("cached_tokens", 0) for r in results), default=0)
        if cached:
            usage["prompt_tokens_details"] = {"cached_tokens": cached}
        del n
        return sorted(results, key=lambda r: r[0]), usage

    async def _prepare(self, request, kind):
        """-> ((body, [prompt ids, ...], params, lora, request id), None) or (None, error response).
        Chat has one prompt; completions accept a string, a token list, or a list of either."""
        try:
            body = await request.json()
        except Exception:
            return None, _err(400, "request body is not valid JSON")
        ok, lora = self._resolve_model(body.get("model"))
        if not ok:
            return None, _err(404, f"The model `{body.get('model')}` does not exist.", "NotFoundError")
        if self.engine.is_sleeping:
            return None, _err(503, "engine is sleeping", "ServiceUnavailable")
        tok = self.engine.tokenizer
        try:
            if kind == "chat":
                msgs = body.get("messages")
                if not isinstance(msgs, list) or not msgs:
                    return None, _err(400, "'messages' must be a non-empty list")
                if body.get("tool_choice") == "none":
                    body.pop("tools", None)  # the model neither sees the tools nor has its text parsed
                ids_list = [tok.encode_chat(msgs, True, body.get("tools"))]
            else:
                prompt = body.get("prompt", "")
                if isinstance(prompt, list) and prompt and isinstance(prompt[0], int):
                    ids_list = [list(prompt)]
                elif isinstance(prompt, list) and prompt and isinstance(prompt[0], list):
                    ids_list = [list(p) for p in prompt]
                elif isinstance(prompt, list):
                    ids_list = [tok.encode(str(p)) for p in prompt] or [tok.encode("")]
                else:
                    ids_list = [tok.encode(str(prompt))]
        except Exception as e:
            return None, _err(400, f"failed to build prompt: {e}")
        for ids in ids_list:
            if len(ids) >= self.engine.max_model_len:
                return None, _err(400, f"This model's maximum context length is {self.engine.max_model_len} "
                                       f"tokens. However, you requested {len(ids)} prompt tokens.")
        params = _sampling_from_body(body, 16 if kind == "completion" else self.engine.max_model_len)
        if params.n < 1 or params.n * len(ids_list) > 128:
            return None, _err(400, "'n' times the number of prompts must be between 1 and 128")
        if params.n > 1 and params.greedy:
            return None, _err(400, "n > 1 needs temperature > 0 (greedy choices would be identical)")
        rid = request.headers.get("X-Request-Id") or uuid.uuid4().hex
        return (body, ids_list, params, lora, rid), None

    async def chat(self, request):
        prep, err = await self._prepare(request, "chat")
        if err is not None:
            return err
        body, ids_list, params, lora, rid = prep
        model = body.get("model") or self.cfg.model_name
        kvp = body.get("kv_transfer_params")
        jobs = self._jobs(rid, ids_list, params)
        if body.get("stream"):
            inc = bool((body.get("stream_options") or {}).get("include_usage"))
            return await self._stream(request, rid, jobs, lora, "chat", model, inc, kvp,
                                      parse_tools=bool(body.get("tools")) and self.cfg.enable_auto_tool_choice)
        try:
            results, usage = await self._run_jobs(request, rid, jobs, lora, model, kvp)
        except ValueError as e:
            return _err(400, str(e))
        choices = []
        for index, text, out_ids, reason, _u, lp_infos in results:
            logprobs = {"content": self._chat_logprob_items(out_ids, lp_infos)} if lp_infos else None
            message = {"role": "assistant", "content": text}
            if body.get("tools") and self.cfg.enable_auto_tool_choice:
                calls, rest = parse_tool_calls(text, self.cfg.tool_call_parser)
                if calls:
                    message = {"role": "assistant", "content": rest or None, "tool_calls": calls}
                    reason = "tool_calls"
            choices.append({"index": index, "message": message, "logprobs": logprobs, "finish_reason": reason})
        return web.json_response({
            "id": "chatcmpl-" + rid, "object": "chat.completion", "created": int(time.time()),
            "model": model, "choices": choices, "usage": usage}, headers={"X-Request-Id": rid})

    async def completions(self, request):
        prep, err = await self._prepare(request, "completion")
        if err is not None:
            return err
        body, ids_list, params, lora, rid = prep
        model = body.get("model") or self.cfg.model_name
        kvp = body.get("kv_transfer_params")
        jobs = self._jobs(rid, ids_list, params)
        if body.get("stream"):
            inc = bool((body.get("stream_options") or {}).get("include_usage"))
            return await self._stream(request, rid, jobs, lora, "completion", model, inc, kvp)
        try:
            results, usage = await self._run_jobs(request, rid, jobs, lora, model, kvp)
        except ValueError as e:
            return _err(400, str(e))
        choices = []
        for (index, text, out_ids, reason, _u, lp_infos), job in zip(results, jobs):
            logprobs = self._completion_logprobs(out_ids, lp_infos) if lp_infos else None
            if body.get("echo"):
                text = self.engine.tokenizer.decode(job[2]) + (" " if text else "") + text
            choices.append({"index": index, "text": text, "logprobs": logprobs, "finish_reason": reason})
        return web.json_response({
            "id": "cmpl-" + rid, "object": "text_completion", "created": int(time.time()),
            "model": model, "choices": choices, "usage": usage}, headers={"X-Request-Id": rid})

    # ------------------------------------------------------------------ pooling endpoints
    async def _embed_ids(self, ids: list[int]) -> list[float]:
        rid = uuid.uuid4().hex
        sp = SamplingParams(max_tokens=1, temperature=0.0, ignore_eos=True)
        ids = ids[: self.engine.max_model_len - 2]
        pooled = None
        async for so in self.ae.generate(rid, ids, sp, None, None, True):
            req = self.ae.live_request(rid)
            if req is not None and req.pooled is not None:
                pooled = req.pooled
        if pooled is None:
            raise RuntimeError("pooling produced no state")
        norm = float(pooled.norm()) or 1.0
        return (pooled / norm).tolist()

    async def embeddings(self, request):
        body = await request.json()
        ok, _ = self._resolve_model(body.get("model"))
        if not ok:
            return _err(404, f"The model `{body.get('model')}` does not exist.", "NotFoundError")
        inp = body.get("input", "")
        items = inp if isinstance(inp, list) else [inp]
        if items and isinstance(items[0], int):
            items = [items]
        tok = self.engine.tokenizer
        data, total = [], 0
        for i, it in enumerate(items):
            ids = list(it) if isinstance(it, list) else tok.encode(str(it))
            total += len(ids)
            data.append({"object": "embedding", "index": i, "embedding": await self._embed_ids(ids)})
        return web.json_response({
            "id": "embd-" + uuid.uuid4().hex, "object": "list", "created": int(time.time()),
            "model": body.get("model") or self.cfg.model_name, "data": data,
            "usage": {"prompt_tokens": total, "total_tokens": total}})

    async def _similarities(self, query: str, docs: list[str]) -> tuple[list[float], int]:
        tok = self.engine.tokenizer
        qi = tok.encode(query)
        qv = await self._embed_ids(qi)
        sims, total = [], len(qi)
        for d in docs:
            di = tok.encode(d)
            total += len(di)
            dv = await self._embed_ids(di)
            sims.append(float(sum(a * b for a, b in zip(qv, dv))))
        return sims, total

    async def rerank(self, request):
        body = await request.json()
        docs = [d if isinstance(d, str) else d.get("text", "") for d in body.get("documents", [])]
        sims, total = await self._similarities(body.get("query", ""), docs)
        order = sorted(range(len(docs)), key=lambda i: -sims[i])
        top_n = body.get("top_n") or len(docs)
        results = [{"index": i, "document": {"text": docs[i]}, "relevance_score": sims[i]}
                   for i in order[:top_n]]
        return web.json_response({"id": "rerank-" + uuid.uuid4().hex,
                                  "model": body.get("model") or self.cfg.model_name,
                                  "usage": {"total_tokens": total}, "results": results})

    async def score(self, request):
        body = await request.json()
        t1 = body.get("text_1", "")
        t2 = body.get("text_2", [])
        t2 = t2 if isinstance(t2, list) else [t2]
        q = t1 if isinstance(t1, str) else (t1[0] if t1 else "")
        sims, total = await self._similarities(q, [str(x) for x in t2])
        data = [{"index": i, "object": "score", "score": s} for i, s in enumerate(sims)]
        return web.json_response({"id": "score-" + uuid.uuid4().hex, "object": "list",
                                  "created": int(time.time()),
                                  "model": body.get("model") or self.cfg.model_name, "data": data,
                                  "usage": {"prompt_tokens": total, "total_tokens": total}})

    async def transcriptions(self, request):
        """Text decoders do not transcribe: a speech model is served by the speech replica
        (engine/speech.py), which `serve` starts for Whisper-class models."""
        return _err(501, "this replica does not serve a transcription model; start one with "
                         "`serve openai/whisper-small --task transcription`", "NotImplementedError")


async def serve(aengine: AsyncLLMEngine, host: str, port: int) -> web.AppRunner:
    server = EngineServer(aengine)
    runner = web.AppRunner(server.app, access_log=None)
    await runner.setup()
    site = web.TCPSite(runner, host, port, reuse_address=True, backlog=4096)
    await site.start()
    log.info("engine serving on http://%s:%d", host, port)
    return runner
