"""Router end to end against in-process fake engines (no GPU, no cluster).

The checks mirror what the reference verifies by log parsing and curl in its e2e scripts
(/root/reference/tests/e2e/test-routing.py:195-469, tests/e2e/test-sticky-routing.sh):
the routing log line format, round-robin alternation, session stickiness, prefix
affinity, two-hop disaggregated prefill."""

import asyncio
import json
import logging
import re

import aiohttp
import pytest
from aiohttp import web

from vllm_production_stack_amd.router import app as router_app
from vllm_production_stack_amd.router.config import parse_args
from vllm_production_stack_amd.testing.fake_engine import FakeEngine

ROUTE_RE = re.compile(r"Routing request (\S+) with session id (\S+) to (\S+) at ([\d.]+), process time = ([\d.]+)")


class Capture(logging.Handler):
    def __init__(self):
        super().__init__()
        self.lines = []

    def emit(self, record):
        self.lines.append(record.getMessage())


async def start_router(argv):
    args = parse_args(argv)
    app = router_app.create_app(args)
    runner = web.AppRunner(app, access_log=None)
    await runner.setup()
    site = web.TCPSite(runner, "127.0.0.1", 0)
    await site.start()
    port = runner.addresses[0][1]
    return runner, f"http://127.0.0.1:{port}", app[router_app.STATE_KEY]


class Stack:
    def __init__(self, n=2, routing="roundrobin", extra=None, labels=None, models=None, engine_kw=None):
        self.n, self.routing, self.extra, self.labels, self.models = n, routing, extra or [], labels, models
        self.engine_kw = engine_kw or {}

    async def __aenter__(self):
        self.engines = [await FakeEngine(model=(self.models[i] if self.models else "fake"),
                                         speed=5000, **self.engine_kw).start() for i in range(self.n)]
        argv = ["--service-discovery", "static", "--static-backends", ",".join(e.url for e in self.engines),
                "--static-models", ",".join(e.model for e in self.engines), "--routing-logic", self.routing,
                "--engine-stats-interval", "1"] + self.extra
        if self.labels:
            argv += ["--static-model-labels", ",".join(self.labels)]
        self.cap = Capture()
        logging.getLogger("vllm_production_stack_amd.router.proxy").addHandler(self.cap)
        self.runner, self.url, self.state = await start_router(argv)
        self.http = aiohttp.ClientSession()
        return self

    async def __aexit__(self, *a):
        logging.getLogger("vllm_production_stack_amd.router.proxy").removeHandler(self.cap)
        await self.http.close()
        await self.runner.cleanup()
        for e in self.engines:
            await e.stop()

    def routes(self):
        out = []
        for ln in self.cap.lines:
            m = ROUTE_RE.search(ln)
            if m:
                out.append((m.group(1), m.group(2), m.group(3)))
        return out

    async def chat(self, content="hi", headers=None, stream=True, model="fake", max_tokens=5, **kw):
        body = {"model": model, "messages": [{"role": "user", "content": content}],
                "max_tokens": max_tokens, "stream": stream, **kw}
        async with self.http.post(self.url + "/v1/chat/completions", json=body, headers=headers or {}) as r:
            data = await r.read()
            return r.status, data, r.headers


def run(coro):
    return asyncio.run(coro)


def test_roundrobin_stream_proxy_and_log_contract():
    async def go():
        async with Stack(2, "roundrobin") as s:
            for _ in range(6):
                st, data, hdr = await s.chat()
                assert st == 200 and data.count(b"data: ") == 6 and b"[DONE]" in data
                assert "X-Request-Id" in hdr
            urls = [u for _, _, u in s.routes()]
            assert len(urls) == 6
            assert all(urls[i] != urls[i + 1] for i in range(5))  # strict alternation
            assert sum(len(e.requests) for e in s.engines) == 6
            assert all(sid == "None" for _, sid, _ in s.routes())
    run(go())


def test_error_shapes():
    async def go():
        async with Stack(1) as s:
            async with s.http.post(s.url + "/v1/chat/completions", json={"messages": []}) as r:
                assert r.status == 400
                assert (await r.json()) == {"error": "Invalid request: missing 'model' in request body."}
                assert "X-Request-Id" in r.headers
            st, data, _ = await s.chat(model="nope")
            assert st == 400 and b"not found or vLLM engine is sleeping" in data
            # demand nobody could serve is counted per model (what a scale-from-zero trigger queries);
            # the model name is client input, so only a bounded number of names get their own series
            for i in range(70):
                await s.chat(model=f"made-up-{i}")
            async with s.http.get(s.url + "/metrics") as r:
                text = await r.text()
            assert 'vllm:router_unroutable_requests_total{model="nope"}' in text
            series = [ln for ln in text.splitlines() if ln.startswith("vllm:router_unroutable_requests_total{")]
            assert len(series) <= 65 and any('model="other"' in ln for ln in series)
            async with s.http.post(s.url + "/v1/chat/completions", data=b"{bad") as r:
                assert r.status == 400
    run(go())


def test_models_engines_health_metrics():
    async def go():
        async with Stack(2, extra=["--static-aliases", "gpt:fake"]) as s:
            async with s.http.get(s.url + "/v1/models") as r:
                d = await r.json()
                ids = [m["id"] for m in d["data"]]
                assert d["object"] == "list" and "fake" in ids and "gpt" in ids
                assert d["data"][0]["owned_by"] == "vllm" and d["data"][0]["parent"] is None
            async with s.http.get(s.url + "/engines") as r:
                e = await r.json()
                assert len(e) == 2 and set(e[0]) == {"engine_id", "serving_models", "created"}
            async with s.http.get(s.url + "/health") as r:
                assert (await r.json()) == {"status": "healthy"}
            async with s.http.get(s.url + "/version") as r:
                assert "version" in await r.json()
            st, _, _ = await s.chat(model="gpt")  # alias is rewritten for the backend
            assert st == 200
            assert any(q["model"] == "fake" for e in s.engines for q in e.requests)
            await asyncio.sleep(1.3)  # one scrape cycle
            async with s.http.get(s.url + "/metrics") as r:
                text = await r.text()
            for name in ("vllm:num_requests_running", "vllm:num_requests_waiting", "vllm:current_qps",
                         "vllm:gpu_prefix_cache_hit_rate", "vllm:healthy_pods_total", "vllm:avg_latency",
                         "vllm:avg_itl", "vllm:num_prefill_requests", "vllm:num_decoding_requests",
                         "vllm:num_requests_swapped", "vllm:avg_decoding_length",
                         "router_cpu_usage_percent", "router_memory_usage_percent",
                         "router_disk_usage_percent"):
                assert name in text, name
            assert 'vllm:healthy_pods_total{server="%s"} 1.0' % s.engines[0].url in text
    run(go())


def test_session_routing_sticky():
    async def go():
        async with Stack(3, "session", extra=["--session-key", "x-user-id"]) as s:
            for user in ("u1", "u2", "u3", "u4", "u5"):
                for _ in range(4):
                    await s.chat(headers={"x-user-id": user})
            by_user = {}
            for _, sid, url in s.routes():
                by_user.setdefault(sid, set()).add(url)
            assert set(by_user) == {"u1", "u2", "u3", "u4", "u5"}
            assert all(len(v) == 1 for v in by_user.values())
    run(go())


def test_prefixaware_routing_keeps_conversation_on_one_engine():
    async def go():
        async with Stack(3, "prefixaware") as s:
            convs = {c: f"conversation {c} " * 40 for c in "abcdef"}
            for turn in range(4):
                for c, base in convs.items():
                    await s.chat(content=base + " more" * turn * 30)
            per_conv = {}
            for e in s.engines:
                for q in e.requests:
                    text = q["messages"][0]["content"]
                    per_conv.setdefault(text.split(" ")[1], set()).add(e.url)
            assert all(len(v) == 1 for v in per_conv.values()), per_conv
    run(go())


def test_sleep_wakeup_and_pinning():
    async def go():
        async with Stack(2) as s:
            async with s.http.get(s.url + "/engines") as r:
                engines = await r.json()
            eid = engines[0]["engine_id"]
            async with s.http.post(s.url + "/sleep") as r:
                assert r.status == 400
                assert (await r.json()) == {"error": "Invalid request: missing target Engine Id."}
            async with s.http.post(s.url + f"/sleep?id={eid}") as r:
                assert (await r.json()) == {"status": "success"}
            async with s.http.get(s.url + f"/is_sleeping?id={eid}") as r:
                assert (await r.json()) == {"is_sleeping": True}
            for _ in range(4):
                st, _, _ = await s.chat()
                assert st == 200
            assert len({u for _, _, u in s.routes()}) == 1  # the sleeping engine gets nothing
            async with s.http.post(s.url + f"/wake_up?id={eid}") as r:
                assert (await r.json()) == {"status": "success"}
            eid2 = engines[1]["engine_id"]
            async with s.http.post(s.url + f"/v1/chat/completions?id={eid2}",
                                   json={"model": "fake", "messages": [], "max_tokens": 1}) as r:
                assert r.status == 200
    run(go())


def test_failover_to_live_endpoint():
    async def go():
        async with Stack(2, extra=["--max-failover-attempts", "1"]) as s:
            await s.engines[0].stop()
            for _ in range(4):
                st, _, _ = await s.chat()
                assert st == 200
            await s.engines[0].start()  # so that teardown can stop it
    run(go())


def test_disaggregated_prefill_two_hops():
    async def go():
        async with Stack(4, "disaggregated_prefill", labels=["prefill", "prefill", "decode", "decode"],
                         extra=["--prefill-model-labels", "prefill", "--decode-model-labels", "decode"]) as s:
            talks = [f"conversation {i}: " + " ".join(f"w{i}x{j}" for j in range(200)) for i in range(4)]
            for i in range(4):
                st, data, hdr = await s.chat(talks[i], headers={"X-Request-Id": f"req-{i}"}, max_tokens=7)
                assert st == 200 and data.count(b"Hello") == 7
            by_id = {}
            for rid, _, url in s.routes():
                by_id.setdefault(rid, []).append(url)
            prefill_urls = {e.url for e in s.engines[:2]}
            decode_urls = {e.url for e in s.engines[2:]}
            for i in range(4):
                hops = by_id[f"req-{i}"]
                assert len(hops) == 2 and hops[0] in prefill_urls and hops[1] in decode_urls
            for e in s.engines[:2]:
                for q in e.requests:
                    assert q["max_tokens"] == 1 and q["kv_transfer_params"]["do_remote_decode"]
                    assert q["kv_transfer_params"]["decode_target"] in decode_urls
            for e in s.engines[2:]:
                for q in e.requests:
                    assert q["max_tokens"] == 7 and q["kv_transfer_params"]["do_remote_prefill"]
            # both pools are used (the reference would pin one endpoint of each)
            assert all(len(e.requests) > 0 for e in s.engines)
            # a follow-up turn of a conversation takes the same pair of replicas
            st, _, _ = await s.chat(talks[2] + " and one more question", headers={"X-Request-Id": "req-2b"}, max_tokens=7)
            assert st == 200
            again = [url for rid, _, url in s.routes() if rid == "req-2b"]
            assert again == by_id["req-2"]
    run(go())


def test_files_and_batches_execute_for_real(tmp_path):
    async def go():
        async with Stack(2, extra=["--enable-batch-api", "--file-storage-path", str(tmp_path)]) as s:
            lines = [json.dumps({"custom_id": f"c{i}", "method": "POST", "url": "/v1/chat/completions",
                                 "body": {"model": "fake", "messages": [{"role": "user", "content": "x"}],
                                          "max_tokens": 3}}) for i in range(5)]
            lines.append(json.dumps({"custom_id": "bad", "method": "POST", "url": "/v1/chat/completions",
                                     "body": {"model": "missing"}}))
            form = aiohttp.FormData()
            form.add_field("file", "\n".join(lines).encode(), filename="in.jsonl")
            form.add_field("purpose", "batch")
            async with s.http.post(s.url + "/v1/files", data=form) as r:
                f = await r.json()
                assert r.status == 200 and f["id"].startswith("file-") and f["purpose"] == "batch"
                assert f["filename"] == "in.jsonl" and f["bytes"] > 0
            async with s.http.get(s.url + f"/v1/files/{f['id']}") as r:
                assert (await r.json())["id"] == f["id"]
            async with s.http.get(s.url + f"/v1/files/{f['id']}/content") as r:
                assert (await r.read()).count(b"\n") == 5
            async with s.http.get(s.url + "/v1/files/file-nope") as r:
                assert r.status == 404
            async with s.http.post(s.url + "/v1/batches", json={"endpoint": "/v1/chat/completions"}) as r:
                assert r.status == 400
            async with s.http.post(s.url + "/v1/batches", json={"input_file_id": f["id"],
                                                                 "endpoint": "/v1/chat/completions"}) as r:
                b = await r.json()
                assert b["id"].startswith("batch_") and b["completion_window"] == "5s"
            for _ in range(100):
                async with s.http.get(s.url + f"/v1/batches/{b['id']}") as r:
                    b = await r.json()
                if b["status"] == "completed":
                    break
                await asyncio.sleep(0.1)
            assert b["status"] == "completed"
            assert b["request_counts"] == {"total": 6, "completed": 5, "failed": 1}
            async with s.http.get(s.url + f"/v1/files/{b['output_file_id']}/content") as r:
                outs = [json.loads(x) for x in (await r.text()).splitlines()]
            assert {o["custom_id"] for o in outs} == {f"c{i}" for i in range(5)}
            assert outs[0]["response"]["body"]["choices"][0]["message"]["content"].startswith("Hello")
            assert b["error_file_id"]
            async with s.http.get(s.url + "/v1/batches") as r:
                assert len((await r.json())["data"]) == 1
            async with s.http.post(s.url + f"/v1/batches/{b['id']}/cancel") as r:
                assert (await r.json())["status"] == "completed"  # finished batches stay finished
            async with s.http.get(s.url + "/v1/batches/batch_nope") as r:
                assert r.status == 404
    run(go())


def test_semantic_cache_and_pii_gates(tmp_path):
    async def go():
        async with Stack(1, extra=["--feature-gates", "SemanticCache=true,PIIDetection=true",
                                   "--semantic-cache-dir", str(tmp_path / "sc"),
                                   "--semantic-cache-threshold", "0.9"]) as s:
            st, data, _ = await s.chat(content="what is the capital of france", stream=False)
            assert st == 200 and "cached" not in json.loads(data)
            st, data, _ = await s.chat(content="what is the capital of france", stream=False)
            d = json.loads(data)
            assert d.get("cached") is True and d["similarity_score"] >= 0.9
            assert len(s.engines[0].requests) == 1
            st, data, _ = await s.chat(content="what is the capital of france", stream=False, skip_cache=True)
            assert "cached" not in json.loads(data)
            st, data, _ = await s.chat(content="my email is bob@example.com")
            assert st == 400 and b"pii_detected" in data
            async with s.http.get(s.url + "/metrics") as r:
                text = await r.text()
            assert "vllm:semantic_cache_hits" in text and "vllm_pii_blocked_total" in text
    run(go())


def test_dynamic_config_reload(tmp_path, monkeypatch):
    async def go():
        e1 = await FakeEngine(model="m1", speed=5000).start()
        e2 = await FakeEngine(model="m2", speed=5000).start()
        cfg = tmp_path / "dyn.json"
        cfg.write_text(json.dumps({"service_discovery": "static", "routing_logic": "roundrobin",
                                   "static_backends": e1.url, "static_models": "m1"}))
        orig = router_app._dynamic_config_watcher
        monkeypatch.setattr(router_app, "_dynamic_config_watcher",
                            lambda state, path, interval=10.0: orig(state, path, 0.2))
        runner, url, state = await start_router(["--dynamic-config-json", str(cfg)])
        try:
            async with aiohttp.ClientSession() as http:
                async with http.get(url + "/health") as r:
                    d = await r.json()
                    assert d["status"] == "healthy" and d["dynamic_config"]["static_models"] == "m1"
                cfg.write_text(json.dumps({"service_discovery": "static", "routing_logic": "session",
                                           "session_key": "x-user-id",
                                           "static_backends": f"{e1.url},{e2.url}", "static_models": "m1,m2"}))
                for _ in range(40):
                    await asyncio.sleep(0.1)
                    async with http.get(url + "/v1/models") as r:
                        ids = {m["id"] for m in (await r.json())["data"]}
                    if ids == {"m1", "m2"}:
                        break
                assert ids == {"m1", "m2"}
                assert type(state.router).__name__ == "SessionRouter"
        finally:
            await runner.cleanup()
            await e1.stop()
            await e2.stop()
    run(go())


def test_static_health_checks_drop_dead_backend():
    async def go():
        e1 = await FakeEngine(model="m", speed=5000).start()
        e2 = await FakeEngine(model="m", speed=5000).start()
        runner, url, state = await start_router([
            "--service-discovery", "static", "--static-backends", f"{e1.url},{e2.url}",
            "--static-models", "m,m", "--static-model-types", "chat,chat",
            "--static-backend-health-checks", "--routing-logic", "roundrobin"])
        try:
            sd = state.service_discovery
            await sd.check_once()
            assert len(sd.get_endpoint_info()) == 2
            await e2.stop()
            await sd.check_once()
            assert [e.url for e in sd.get_endpoint_info()] == [e1.url]
        finally:
            await runner.cleanup()
            await e1.stop()
    run(go())


def test_transcription_proxy():
    async def go():
        async with Stack(1, models=["whisper"]) as s:
            from vllm_production_stack_amd.router.utils import silent_wav

            form = aiohttp.FormData()
            form.add_field("file", silent_wav(), filename="a.wav", content_type="audio/wav")
            form.add_field("model", "whisper")
            form.add_field("language", "en")
            async with s.http.post(s.url + "/v1/audio/transcriptions", data=form) as r:
                assert r.status == 200 and (await r.json()) == {"text": "fake transcript"}
            assert s.engines[0].requests[0]["language"] == "en"
            form = aiohttp.FormData()
            form.add_field("file", b"x", filename="a.wav")
            form.add_field("model", "nope")
            async with s.http.post(s.url + "/v1/audio/transcriptions", data=form) as r:
                assert r.status == 404
    run(go())


def test_client_disconnect_mid_stream_is_handled_quietly(caplog):
    """A client that hangs up while tokens are streaming: the router stops relaying, closes the
    engine connection (so the engine can abort), logs one info line and no traceback, and keeps
    serving."""
    async def go():
        async with Stack(1, engine_kw={}) as s:
            s.engines[0].speed = 100  # 400 tokens = 4 s of stream
            r = await s.http.post(s.url + "/v1/chat/completions", json={
                "model": "fake", "messages": [{"role": "user", "content": "x"}], "max_tokens": 400, "stream": True})
            await r.content.readany()
            r.close()  # hang up with most of the stream still to come
            await asyncio.sleep(0.6)
            st, data, _ = await s.chat(max_tokens=3)
            assert st == 200
            assert any("disconnected before the end" in ln for ln in s.cap.lines), s.cap.lines
            assert s.engines[0].aborted == 1  # the hang-up reached the engine

    with caplog.at_level(logging.INFO):
        run(go())
    errors = [r.getMessage() for r in caplog.records if r.levelno >= logging.ERROR]
    assert not errors, errors
