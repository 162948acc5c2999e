"""Endpoint-picker shim: an Envoy external-processing (ext_proc) gRPC server in front of
``vpsa-picker``.

A Gateway (kgateway, Istio, GKE) that routes to an InferencePool asks the pool's extension which
pod a request should go to.  The question arrives over Envoy's ext_proc stream: request headers,
then the (buffered) request body.  This shim answers it with the routing logic of this stack:

    Envoy --ext_proc stream--> this shim --POST /pick {pods, model, prompt, session}--> vpsa-picker
          <-- header mutation  x-gateway-destination-endpoint: <pod ip>:<port>  (and the same value
              as dynamic metadata under ``envoy.lb``, which is where newer gateways read it)

The reference gets there by patching its pickers into the upstream endpoint-picker binary
(/root/reference/src/gateway_inference_extension/scheduler.patch:13-77 replaces the scheduler's
picker and disables its filters and scorers; roundrobin_picker.go / prefix_aware_picker.go /
kv_aware_picker.go are the pickers).  Here the pickers live in ``vpsa-picker`` (same routing core as
the router) and this process is the protocol adapter.

Per stream (= per HTTP request):
  * ``request_headers``: remembered (path, session key).  With ``end_of_stream`` (no body) the
    pick happens at once on an empty prompt; otherwise the reply is CONTINUE and the pick waits
    for the body.
  * ``request_body``: chunks are collected until ``end_of_stream``; the JSON body's ``model`` and
    prompt (``prompt`` or the concatenated chat ``messages``) go to the picker; the reply carries
    the destination header and ``clear_route_cache``.
  * no ready pod -> ImmediateResponse 503; picker unreachable -> local round robin (the request
    must not fail because the picker restarted).
  * response-side messages are answered with CONTINUE.

Candidate pods come from ``--pods name=ip[,name=ip...]`` or from a watch on the pool's pods
(``--namespace`` + ``--label-selector``, ready containers only), the way the InferencePool selects
them.  ``grpc.health.v1.Health/Check`` is served for the Deployment's gRPC probes.
"""

from __future__ import annotations

import argparse
import asyncio
import itertools
import json
import logging
import os
import signal
import subprocess
import time

import aiohttp
import grpc

from . import extproc_pb as pb

log = logging.getLogger("vpsa.gateway.extproc")
DEST_HEADER = "x-gateway-destination-endpoint"
MAX_BODY = 64 << 20


def header_dict(hmap) -> dict[str, str]:
    out = {}
    for h in hmap.headers:
        v = h.raw_value.decode("utf-8", "replace") if h.raw_value else h.value
        out[h.key.lower()] = v
    return out


def prompt_of(body: dict) -> str:
    """What prefix-aware pickers hash: the completion prompt, or the chat turns in order."""
    p = body.get("prompt")
    if isinstance(p, str):
        return p
    if isinstance(p, list):
        return " ".join(map(str, p[:4096])) if p and not isinstance(p[0], str) else "\n".join(map(str, p))
    msgs = body.get("messages")
    if isinstance(msgs, list):
        parts = []
        for m in msgs:
            c = m.get("content") if isinstance(m, dict) else None
            if isinstance(c, str):
                parts.append(c)
            elif isinstance(c, list):
                parts.extend(str(x.get("text", "")) for x in c if isinstance(x, dict))
        return "\n".join(parts)
    inp = body.get("input")
    return inp if isinstance(inp, str) else ""


class PodSource:
    """The pool's candidate pods: a fixed list, or a watch on labelled pods."""

    def __init__(self, static: dict[str, str] | None = None, namespace: str = "default", selector: str = "",
                 client=None):
        self.pods: dict[str, str] = dict(static or {})
        self.static = static is not None
        self.namespace, self.selector = namespace, selector
        self.client = client
        self._task: asyncio.Task | None = None

    async def start(self) -> None:
        if self.static:
            return
        if self.client is None:
            from ..router.k8s_client import K8sClient

            self.client = K8sClient()
        self._task = asyncio.create_task(self._watch())

    async def close(self) -> None:
        if self._task:
            self._task.cancel()
        if self.client is not None and not self.static:
            await self.client.close()

    async def _watch(self) -> None:
        path = f"/api/v1/namespaces/{self.namespace}/pods"
        while True:
            try:
                async for ev in self.client.watch(path, self.selector, 0):
                    pod = ev.get("object") or {}
                    name = (pod.get("metadata") or {}).get("name")
                    if not name:
                        continue
                    st = pod.get("status") or {}
                    cs = st.get("containerStatuses") or []
                    ready = (ev.get("type") != "DELETED" and st.get("podIP") and cs and all(c.get("ready") for c in cs)
                             and not (pod.get("metadata") or {}).get("deletionTimestamp"))
                    if ready:
                        self.pods[name] = st["podIP"]
                    else:
                        self.pods.pop(name, None)
            except asyncio.CancelledError:
                raise
            except Exception as e:
                log.error("pod watch failed: %s", e)
            await asyncio.sleep(0.5)

    def snapshot(self) -> list[dict]:
        return [{"name": n, "address": a} for n, a in sorted(self.pods.items())]


class EndpointPicker:
    def __init__(self, pods: PodSource, picker_url: str, target_port: int, session_key: str = "x-user-id",
                 timeout_s: float = 2.0):
        self.pods, self.picker_url, self.port = pods, picker_url.rstrip("/"), target_port
        self.session_key, self.timeout = session_key.lower(), timeout_s
        self._http: aiohttp.ClientSession | None = None
        self._rr = itertools.count()
        self.stats = {"streams": 0, "picks": 0, "fallback_picks": 0, "no_pods": 0, "bad_bodies": 0}

    async def start(self) -> None:
        self._http = aiohttp.ClientSession(timeout=aiohttp.ClientTimeout(total=self.timeout))

    async def close(self) -> None:
        if self._http:
            await self._http.close()

    async def pick(self, headers: dict[str, str], body: bytes) -> str | None:
        """-> "ip:port" of the chosen pod (None: the pool is empty)."""
        pods = self.pods.snapshot()
        if not pods:
            self.stats["no_pods"] += 1
            return None
        model, prompt = "", ""
        if body:
            try:
                doc = json.loads(body)
                if isinstance(doc, dict):
                    model = doc.get("model") if isinstance(doc.get("model"), str) else ""
                    prompt = prompt_of(doc)
            except (ValueError, UnicodeDecodeError):
                self.stats["bad_bodies"] += 1  # the engine will refuse it; still needs a destination
        req = {"pods": pods, "model": model, "prompt": prompt, "session": headers.get(self.session_key, "")}
        address = None
        try:
            async with self._http.post(self.picker_url + "/pick", json=req) as r:
                if r.status == 200:
                    address = (await r.json(content_type=None)).get("address")
        except (aiohttp.ClientError, asyncio.TimeoutError, ValueError) as e:
            log.warning("picker unreachable (%s): falling back to round robin", e)
        if not address or address not in {p["address"] for p in pods}:
            self.stats["fallback_picks"] += 1
            address = pods[next(self._rr) % len(pods)]["address"]
        self.stats["picks"] += 1
        host = f"[{address}]" if ":" in address else address
        return f"{host}:{self.port}"

    # ------------------------------------------------------------------ the ext_proc stream
    def _destination(self, common, dest: str, resp) -> None:
        opt = common.header_mutation.set_headers.add()
        opt.header.key = DEST_HEADER
        opt.header.raw_value = dest.encode()
        opt.append_action = pb.OVERWRITE_IF_EXISTS_OR_ADD
        common.clear_route_cache = True
        resp.dynamic_metadata.update({"envoy.lb": {DEST_HEADER: dest}})

    @staticmethod
    def _unavailable(detail: str):
        resp = pb.ProcessingResponse()
        resp.immediate_response.status.code = 503
        resp.immediate_response.body = json.dumps({"error": {"message": detail, "type": "service_unavailable"}}).encode()
        resp.immediate_response.details = detail
        return resp

    async def process(self, request_iterator, context):
        self.stats["streams"] += 1
        headers: dict[str, str] = {}
        chunks: list[bytes] = []
        size = 0
        async for msg in request_iterator:
            kind = msg.WhichOneof("request")
            resp = pb.ProcessingResponse()
            if kind == "request_headers":
                headers = header_dict(msg.request_headers.headers)
                if msg.request_headers.end_of_stream:
                    dest = await self.pick(headers, b"")
                    if dest is None:
                        yield self._unavailable("no ready pod in the pool")
                        continue
                    self._destination(resp.request_headers.response, dest, resp)
                else:
                    resp.request_headers.response.status = 0
                    resp.request_headers.SetInParent()
            elif kind == "request_body":
                size += len(msg.request_body.body)
                if size <= MAX_BODY:
                    chunks.append(msg.request_body.body)
                if not msg.request_body.end_of_stream:
                    resp.request_body.SetInParent()
                else:
                    dest = await self.pick(headers, b"".join(chunks) if size <= MAX_BODY else b"")
                    chunks, size = [], 0
                    if dest is None:
                        yield self._unavailable("no ready pod in the pool")
                        continue
                    self._destination(resp.request_body.response, dest, resp)
            elif kind == "response_headers":
                resp.response_headers.SetInParent()
            elif kind == "response_body":
                resp.response_body.SetInParent()
            elif kind == "request_trailers":
                resp.request_trailers.SetInParent()
            elif kind == "response_trailers":
                resp.response_trailers.SetInParent()
            else:
                continue  # a message kind this shim does not know: nothing to answer
            yield resp


def build_server(picker: EndpointPicker, host: str, port: int) -> tuple[grpc.aio.Server, int]:
    async def check(request, context):
        ok = picker.pods.static or bool(picker.pods.pods) or picker.pods._task is not None
        return pb.HealthCheckResponse(status=1 if ok else 2)

    handlers = [
        grpc.method_handlers_generic_handler(pb.SERVICE, {
            "Process": grpc.stream_stream_rpc_method_handler(
                picker.process, request_deserializer=pb.ProcessingRequest.FromString,
                response_serializer=lambda m: m.SerializeToString())}),
        grpc.method_handlers_generic_handler(pb.HEALTH_SERVICE, {
            "Check": grpc.unary_unary_rpc_method_handler(
                check, request_deserializer=pb.HealthCheckRequest.FromString,
                response_serializer=lambda m: m.SerializeToString())}),
    ]
    server = grpc.aio.server(options=[("grpc.max_receive_message_length", MAX_BODY + (1 << 20))])
    server.add_generic_rpc_handlers(handlers)
    bound = server.add_insecure_port(f"{host}:{port}")
    return server, bound


def parse_pods(spec: str) -> dict[str, str]:
    out = {}
    for item in filter(None, (x.strip() for x in spec.split(","))):
        name, _, addr = item.partition("=")
        out[name] = addr or name
    return out


async def amain(a: argparse.Namespace) -> None:
    child = None
    picker_url = a.picker_url
    if a.spawn_picker:
        exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "_native", "vpsa-picker")
        cmd = [exe, "--picker", a.spawn_picker, "--host", "127.0.0.1", "--port", str(a.picker_port)]
        if a.controller:
            cmd += ["--controller", a.controller]
        child = subprocess.Popen(cmd)
        picker_url = f"http://127.0.0.1:{a.picker_port}"
    pods = PodSource(parse_pods(a.pods) if a.pods else None, a.namespace, a.label_selector)
    picker = EndpointPicker(pods, picker_url, a.target_port, a.session_key)
    await pods.start()
    await picker.start()
    server, bound = build_server(picker, a.host, a.port)
    await server.start()
    log.info("ext_proc endpoint picker listening on %s:%d (picker %s)", a.host, bound, picker_url)
    print(f"vpsa-extproc listening on {a.host}:{bound}", flush=True)
    stop = asyncio.Event()
    loop = asyncio.get_running_loop()
    for sig in (signal.SIGTERM, signal.SIGINT):
        loop.add_signal_handler(sig, stop.set)
    await stop.wait()
    await server.stop(2)
    await picker.close()
    await pods.close()
    if child is not None:
        child.terminate()
        try:
            child.wait(5)
        except subprocess.TimeoutExpired:
            child.kill()


def main() -> None:
    ap = argparse.ArgumentParser(description="Envoy ext_proc endpoint picker for InferencePool gateways")
    ap.add_argument("--host", default="0.0.0.0")
    ap.add_argument("--port", type=int, default=9002)
    ap.add_argument("--picker-url", default="http://127.0.0.1:9003", help="where vpsa-picker listens")
    ap.add_argument("--spawn-picker", default="", help="start vpsa-picker with this algorithm "
                                                       "(roundrobin|prefix|kvaware|session) as a child process")
    ap.add_argument("--picker-port", type=int, default=9003)
    ap.add_argument("--controller", default="", help="KV controller URL for the kvaware picker")
    ap.add_argument("--pods", default="", help="fixed candidates: name=ip,name=ip (default: watch the namespace)")
    ap.add_argument("--namespace", default=os.environ.get("POD_NAMESPACE", "default"))
    ap.add_argument("--label-selector", default="", help="the InferencePool's selector, e.g. app=vllm-llama3")
    ap.add_argument("--target-port", type=int, default=8000, help="the pool's targetPortNumber")
    ap.add_argument("--session-key", default="x-user-id")
    ap.add_argument("--log-level", default="info")
    a = ap.parse_args()
    logging.basicConfig(level=a.log_level.upper(), format="[%(asctime)s] %(levelname)s %(name)s: %(message)s")
    asyncio.run(amain(a))


if __name__ == "__main__":
    main()
