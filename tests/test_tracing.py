"""Request tracing: W3C context propagation, the hand-encoded OTLP export message (checked
with an independent wire-format decoder), both transports against in-process collectors, and
one trace across client -> router -> engine.

Parity note: the reference only sets OTEL_* variables on an external engine image
(/root/reference/tutorials/assets/values-12-otel-vllm.yaml:15-22); the opentelemetry packages
are not installed here, so the message layout is pinned against the field numbers of the
public opentelemetry-proto definitions written out in the decoder below."""

import asyncio
import struct
import threading
from concurrent import futures

import aiohttp
import pytest
from aiohttp import web

from vllm_production_stack_amd import tracing
from vllm_production_stack_amd.tracing import Tracer, format_traceparent, parse_traceparent


# ------------------------------------------------------------------ independent decoder
def _read_varint(buf, i):
    shift = n = 0
    while True:
        b = buf[i]
        i += 1
        n |= (b & 0x7F) << shift
        shift += 7
        if not b & 0x80:
            return n, i


def fields(buf):
    """[(field number, wire type, value)] of one protobuf message."""
    out, i = [], 0
    while i < len(buf):
        key, i = _read_varint(buf, i)
        no, wt = key >> 3, key & 7
        if wt == 0:
            v, i = _read_varint(buf, i)
        elif wt == 1:
            v, i = buf[i:i + 8], i + 8
        elif wt == 2:
            ln, i = _read_varint(buf, i)
            v, i = buf[i:i + ln], i + ln
        else:
            raise AssertionError(f"unexpected wire type {wt}")
        out.append((no, wt, v))
    return out


def one(msg, no):
    return [v for n, _, v in fields(msg) if n == no]


def any_value(buf):
    (no, wt, v), = fields(buf)
    if no == 1:
        return v.decode()
    if no == 2:
        return bool(v)
    if no == 3:
        return v if v < (1 << 63) else v - (1 << 64)
    if no == 4:
        return struct.unpack("<d", v)[0]
    if no == 5:
        return [any_value(x) for x in one(v, 1)]
    raise AssertionError(no)


def attrs(kvs):
    return {one(kv, 1)[0].decode(): any_value(one(kv, 2)[0]) for kv in kvs}


def decode_export(payload):
    """-> (resource attributes, [span dicts])"""
    spans, resource = [], {}
    for rs in one(payload, 1):
        resource = attrs(one(one(rs, 1)[0], 1))
        for ss in one(rs, 2):
            for sp in one(ss, 2):
                st = one(sp, 15)
                spans.append({
                    "trace_id": one(sp, 1)[0], "span_id": one(sp, 2)[0],
                    "parent": (one(sp, 4) or [b""])[0], "name": one(sp, 5)[0].decode(),
                    "kind": one(sp, 6)[0],
                    "start": struct.unpack("<Q", one(sp, 7)[0])[0],
                    "end": struct.unpack("<Q", one(sp, 8)[0])[0],
                    "attrs": attrs(one(sp, 9)),
                    "events": [one(e, 2)[0].decode() for e in one(sp, 11)],
                    "status": (one(st[0], 3) or [0])[0] if st else 0,
                })
    return resource, spans


# ------------------------------------------------------------------ unit level
def test_traceparent_roundtrip_and_rejects():
    tid, sid = bytes(range(16)), bytes(range(8, 16))
    tp = format_traceparent(tid, sid, 1)
    assert tp == "00-000102030405060708090a0b0c0d0e0f-08090a0b0c0d0e0f-01"
    assert parse_traceparent(tp) == (tid, sid, 1)
    for bad in (None, "", "garbage", "00-" + "0" * 32 + "-08090a0b0c0d0e0f-01", "ff-" + tp[3:],
                "00-xyz-08090a0b0c0d0e0f-01", tp[:-3]):
        assert parse_traceparent(bad) is None


def test_disabled_tracer_is_a_noop():
    t = Tracer.from_env("svc", env={})
    assert not t.enabled
    span = t.start_span("x", "00-" + "1" * 32 + "-" + "2" * 16 + "-01")
    span.set("a", 1).event("e")
    span.end()
    assert span.traceparent == "" and t.flush() == 0


def test_env_configuration():
    env = {"OTEL_EXPORTER_OTLP_ENDPOINT": "http://otel-collector:4317", "OTEL_SERVICE_NAME": "vllm-router",
           "OTEL_RESOURCE_ATTRIBUTES": "service.name=ignored,deployment.environment=test"}
    t = Tracer.from_env("default", env=env)
    assert t.enabled and t.protocol == "grpc" and t.endpoint == "http://otel-collector:4317"
    assert t.resource["service.name"] == "vllm-router" and t.resource["deployment.environment"] == "test"
    t = Tracer.from_env("default", env={"OTEL_EXPORTER_OTLP_ENDPOINT": "http://c:4318",
                                        "OTEL_EXPORTER_OTLP_PROTOCOL": "http/protobuf"})
    assert t.endpoint == "http://c:4318/v1/traces" and t.resource["service.name"] == "default"
    assert not Tracer.from_env("d", env={"OTEL_EXPORTER_OTLP_ENDPOINT": "http://c:4317",
                                         "OTEL_SDK_DISABLED": "true"}).enabled
    off = Tracer.from_env("d", env={"OTEL_EXPORTER_OTLP_ENDPOINT": "http://c:4317",
                                    "OTEL_TRACES_SAMPLER": "always_off"})
    s = off.start_span("x")
    s.end()
    assert s.traceparent.endswith("-00") and off._q.qsize() == 0  # propagated, not exported


def test_export_message_layout():
    t = Tracer("http://unused:4317", service_name="svc", resource={"k": "v"})
    parent = t.start_span("parent", "00-" + "ab" * 16 + "-" + "cd" * 8 + "-01")
    child = t.start_span("child", parent=parent, kind=tracing.KIND_CLIENT)
    child.set("n", 7).set("neg", -3).set("f", 0.25).set("b", True).set("s", "text").set("l", [1, "two"])
    child.set("skipped", None)
    child.event("first_chunk", seconds=0.5)
    child.error("boom")
    child.end_ns = child.start_ns + 5
    parent.end_ns = parent.start_ns + 10
    payload = tracing.encode_export_request([parent, child], t.resource, "scope", "1.0")
    resource, spans = decode_export(payload)
    assert resource["service.name"] == "svc" and resource["k"] == "v"
    p, c = spans
    assert p["trace_id"] == bytes.fromhex("ab" * 16) and p["parent"] == bytes.fromhex("cd" * 8)
    assert c["trace_id"] == p["trace_id"] and c["parent"] == p["span_id"] and len(c["span_id"]) == 8
    assert (p["kind"], c["kind"]) == (tracing.KIND_SERVER, tracing.KIND_CLIENT)
    assert c["attrs"] == {"n": 7, "neg": -3, "f": 0.25, "b": True, "s": "text", "l": [1, "two"]}
    assert c["events"] == ["first_chunk"] and c["status"] == tracing.STATUS_ERROR and p["status"] == 0
    assert c["end"] - c["start"] == 5 and p["name"] == "parent"


# ------------------------------------------------------------------ collectors
class GrpcCollector:
    """A gRPC server that accepts TraceService/Export and keeps the raw request bytes."""

    def __init__(self):
        import grpc

        self.payloads, self.metadata = [], []
        self.got = threading.Event()

        def export(request, context):
            self.payloads.append(request)
            self.metadata.append(dict(context.invocation_metadata()))
            self.got.set()
            return b""

        handler = grpc.method_handlers_generic_handler(
            "opentelemetry.proto.collector.trace.v1.TraceService",
            {"Export": grpc.unary_unary_rpc_method_handler(export, request_deserializer=lambda b: b,
                                                           response_serializer=lambda b: b)})
        self.server = grpc.server(futures.ThreadPoolExecutor(max_workers=2))
        self.server.add_generic_rpc_handlers((handler,))
        self.port = self.server.add_insecure_port("127.0.0.1:0")
        self.server.start()

    def spans(self):
        return [s for p in self.payloads for s in decode_export(p)[1]]

    def stop(self):
        self.server.stop(0)


def test_grpc_export_reaches_collector():
    col = GrpcCollector()
    try:
        t = Tracer.from_env("engine-a", env={"OTEL_EXPORTER_OTLP_ENDPOINT": f"http://127.0.0.1:{col.port}",
                                             "OTEL_EXPORTER_OTLP_HEADERS": "x-tenant=blue",
                                             "OTEL_BSP_SCHEDULE_DELAY": "50"})
        for i in range(3):
            t.start_span(f"s{i}").set("i", i).end()
        assert col.got.wait(10), "collector received nothing"
        t.shutdown()
        assert sorted(s["name"] for s in col.spans()) == ["s0", "s1", "s2"]
        assert col.metadata[0].get("x-tenant") == "blue"
        assert decode_export(col.payloads[0])[0]["service.name"] == "engine-a"
        assert t.exported == 3 and t.dropped == 0
    finally:
        col.stop()


def test_collector_outage_never_raises():
    t = Tracer("http://127.0.0.1:9", "grpc", timeout_s=0.5)
    t.start_span("lost").end()
    assert t.flush() == 0 and t.export_errors == 1 and t.dropped == 1
    t.shutdown()


def test_http_protobuf_export():
    async def go():
        got = []

        async def traces(request):
            assert request.headers["Content-Type"] == "application/x-protobuf"
            got.append(await request.read())
            return web.Response(body=b"", content_type="application/x-protobuf")

        app = web.Application()
        app.router.add_post("/v1/traces", traces)
        runner = web.AppRunner(app, access_log=None)
        await runner.setup()
        site = web.TCPSite(runner, "127.0.0.1", 0)
        await site.start()
        port = runner.addresses[0][1]
        t = Tracer.from_env("svc", env={"OTEL_EXPORTER_OTLP_ENDPOINT": f"http://127.0.0.1:{port}",
                                        "OTEL_EXPORTER_OTLP_PROTOCOL": "http/protobuf"})
        t.start_span("over-http").end()
        sent = await asyncio.get_running_loop().run_in_executor(None, t.flush)
        await runner.cleanup()
        assert sent == 1 and decode_export(got[0])[1][0]["name"] == "over-http"

    asyncio.run(go())


# ------------------------------------------------------------------ one trace across the stack
def test_trace_spans_router_and_engine(monkeypatch):
    from vllm_production_stack_amd.engine.async_engine import AsyncLLMEngine
    from vllm_production_stack_amd.engine.config import EngineConfig
    from vllm_production_stack_amd.engine.llm_engine import LLMEngine
    from vllm_production_stack_amd.engine.server import serve
    from vllm_production_stack_amd.router import app as router_app
    from vllm_production_stack_amd.router.config import parse_args

    col = GrpcCollector()
    monkeypatch.setenv("OTEL_EXPORTER_OTLP_ENDPOINT", f"http://127.0.0.1:{col.port}")
    monkeypatch.setenv("OTEL_BSP_SCHEDULE_DELAY", "50")
    monkeypatch.delenv("OTEL_SERVICE_NAME", raising=False)

    async def go():
        cfg = EngineConfig(model="tiny-llama", device="cpu", dtype="float32", max_model_len=512,
                           max_num_seqs=8, num_gpu_blocks_override=64)
        ae = AsyncLLMEngine(LLMEngine(cfg))
        ae.start(asyncio.get_running_loop())
        erunner = await serve(ae, "127.0.0.1", 0)
        eport = erunner.addresses[0][1]
        args = parse_args(["--service-discovery", "static", "--static-backends", f"http://127.0.0.1:{eport}",
                           "--static-models", "tiny-llama", "--routing-logic", "roundrobin"])
        rapp = router_app.create_app(args)
        rrunner = web.AppRunner(rapp, access_log=None)
        await rrunner.setup()
        site = web.TCPSite(rrunner, "127.0.0.1", 0)
        await site.start()
        rport = rrunner.addresses[0][1]
        client_tp = "00-" + "5a" * 16 + "-" + "c3" * 8 + "-01"
        try:
            async with aiohttp.ClientSession() as s:
                for stream in (True, False):
                    body = {"model": "tiny-llama", "messages": [{"role": "user", "content": "hello there"}],
                            "max_tokens": 4, "temperature": 0, "stream": stream}
                    async with s.post(f"http://127.0.0.1:{rport}/v1/chat/completions", json=body,
                                      headers={"traceparent": client_tp, "X-Request-Id": f"req-{stream}"}) as r:
                        assert r.status == 200
                        await r.read()
        finally:
            await rrunner.cleanup()
            await erunner.cleanup()
            ae.stop()

    try:
        asyncio.run(go())
        spans = col.spans()
    finally:
        col.stop()
    routers = [s for s in spans if s["name"].startswith("router ")]
    engines = [s for s in spans if s["name"] == "llm_request"]
    assert len(routers) == 2 and len(engines) == 2, [s["name"] for s in spans]
    for e in engines:
        assert e["trace_id"] == bytes.fromhex("5a" * 16)
        parent = [r for r in routers if r["span_id"] == e["parent"]]
        assert len(parent) == 1, "engine span must hang under the router span"
        assert parent[0]["parent"] == bytes.fromhex("c3" * 8)
        a = e["attrs"]
        assert a["gen_ai.request.id"] == parent[0]["attrs"]["vpsa.request_id"]
        assert a["gen_ai.usage.completion_tokens"] == 4 and a["gen_ai.usage.prompt_tokens"] > 0
        assert a["gen_ai.request.max_tokens"] == 4 and a["gen_ai.response.model"] == "tiny-llama"
        assert 0 < a["gen_ai.latency.time_to_first_token"] <= a["gen_ai.latency.e2e"]
        assert a["gen_ai.latency.time_in_queue"] >= 0
        assert parent[0]["attrs"]["http.status_code"] == 200
        assert parent[0]["attrs"]["vpsa.routing_logic"] == "RoundRobinRouter"
        assert "first_chunk" in parent[0]["events"]
