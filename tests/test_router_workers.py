"""Several router processes on one port (``--router-workers``): conversation affinity and the
round-robin position are properties of the router, not of the worker that happened to accept a
connection."""

import asyncio
import json
import os
import socket
import subprocess
import sys
import time
from collections import Counter

import aiohttp

from vllm_production_stack_amd.router import workers
from vllm_production_stack_amd.testing.fake_engine import FakeEngine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


async def _with_router(n_engines, routing, n_workers, body):
    engines = [await FakeEngine(model="fake", speed=5000).start() for _ in range(n_engines)]
    port = _free_port()
    proc = subprocess.Popen(
        [sys.executable, "-m", "vllm_production_stack_amd.router.app", "--host", "127.0.0.1", "--port", str(port),
         "--service-discovery", "static", "--static-backends", ",".join(e.url for e in engines),
         "--static-models", ",".join(["fake"] * n_engines), "--routing-logic", routing,
         "--router-workers", str(n_workers), "--log-level", "warning"], cwd=ROOT)
    url = f"http://127.0.0.1:{port}"
    try:
        async with aiohttp.ClientSession() as s:
            for _ in range(100):
                try:
                    async with s.get(url + "/health") as r:
                        if r.status == 200:
                            break
                except aiohttp.ClientError:
                    pass
                await asyncio.sleep(0.1)
            else:
                raise RuntimeError("router did not come up")
        await asyncio.sleep(0.5)  # every worker is listening
        await body(url, engines)
    finally:
        proc.terminate()
        proc.wait(10)
        for e in engines:
            await e.stop()


async def _chat(url, content):
    # a connection of its own per request: the kernel picks the worker anew every time
    async with aiohttp.ClientSession(connector=aiohttp.TCPConnector(force_close=True)) as s:
        async with s.post(url + "/v1/chat/completions",
                          json={"model": "fake", "messages": [{"role": "user", "content": content}],
                                "max_tokens": 3, "stream": True}) as r:
            await r.read()
            return r.status


def test_prefix_affinity_survives_the_choice_of_worker():
    async def body(url, engines):
        convs = {c: f"conversation {c} " * 40 for c in "abcdefgh"}
        for turn in range(5):
            assert set(await asyncio.gather(*[_chat(url, base + " more" * turn * 30) for base in convs.values()])) == {200}
        per_conv = {}
        for e in engines:
            for q in e.requests:
                per_conv.setdefault(q["messages"][0]["content"].split(" ")[1], set()).add(e.url)
        assert len(per_conv) == 8 and all(len(v) == 1 for v in per_conv.values()), per_conv

    asyncio.run(_with_router(3, "prefixaware", 3, body))


def test_round_robin_position_is_shared_between_workers():
    async def body(url, engines):
        for _ in range(10):
            await asyncio.gather(*[_chat(url, "hi") for _ in range(6)])
        counts = Counter({e.url: len(e.requests) for e in engines})
        assert sum(counts.values()) == 60 and max(counts.values()) - min(counts.values()) <= 1, counts

    asyncio.run(_with_router(3, "roundrobin", 3, body))


def test_shared_trie_service_roundtrip(tmp_path):
    srv = workers.SharedTrieServer(str(tmp_path / "t.sock"), 1000)
    srv.start()

    async def go():
        a, b = workers.SharedTrieClient(srv.path), workers.SharedTrieClient(srv.path)
        text = "x" * 1000
        assert await a.longest_prefix_match(text, {"e1", "e2"}) == (0, {"e1", "e2"})
        await a.insert(text, "e2")
        n, urls = await b.longest_prefix_match(text + "tail", {"e1", "e2"})  # seen through another connection
        assert urls == {"e2"} and n >= 896
        await b.remove_endpoint("e2")
        assert (await a.longest_prefix_match(text, {"e1", "e2"}))[1] == {"e1", "e2"}
        await a.close()
        await b.close()

    try:
        asyncio.run(go())
    finally:
        srv.stop()


def test_kvaware_stays_single_process():
    class A:
        router_workers, routing_logic = 4, "kvaware"

    assert workers.effective_workers(A) == 1
    A.routing_logic = "session"
    assert workers.effective_workers(A) == 4


def _metric(text: str, name: str) -> dict:
    """{server label: value} of one gauge in a Prometheus text page."""
    out = {}
    for line in text.splitlines():
        if line.startswith(name + "{"):
            labels, val = line[len(name):].rsplit(" ", 1)
            server = labels.split('server="')[1].split('"')[0] if 'server="' in labels else ""
            out[server] = float(val)
    return out


def test_a_scrape_of_any_worker_reports_the_whole_router():
    """Four workers, 200 requests in ten seconds: the kernel spreads the connections over the
    workers, so every worker saw about a quarter of the traffic, and every single ``/metrics``
    answer (whichever worker gives it) must report the whole: request rate within 5 % of what the
    client sent, and the same figure from scrape to scrape."""
    async def body(url, engines):
        n, spacing = 200, 0.02
        t0 = time.time()

        async def one(i):
            await asyncio.sleep(i * spacing)
            return await _chat(url, f"question {i}")

        assert set(await asyncio.gather(*[one(i) for i in range(n)])) == {200}
        sent_for = time.time() - t0
        await asyncio.sleep(1.2)  # every worker has published its snapshot since the last request
        pages = []
        for _ in range(12):  # a connection each: different workers answer
            async with aiohttp.ClientSession(connector=aiohttp.TCPConnector(force_close=True)) as s:
                async with s.get(url + "/metrics") as r:
                    assert r.status == 200
                    pages.append(await r.text())
        window = 60.0  # the router's default request-stats window: all 200 requests are inside it
        want_qps = n / window
        totals = [sum(_metric(p, "vllm:current_qps").values()) for p in pages]
        for t in totals:
            assert abs(t - want_qps) / want_qps < 0.05, (totals, want_qps, sent_for)
        assert all(sum(_metric(p, "vllm:num_prefill_requests").values()) == 0 for p in pages)
        lat = [v for p in pages for v in _metric(p, "vllm:avg_latency").values()]
        assert lat and all(0 < v < 5 for v in lat), lat
        # the engines together received exactly what the router reports having routed
        assert sum(len(e.requests) for e in engines) == n

    asyncio.run(_with_router(2, "roundrobin", 4, body))


def test_merge_rules_of_worker_snapshots():
    """Sums for shares of one quantity, request-rate-weighted means for averages, the answering
    worker's copy for what all workers see alike, -1 (no data) carries no weight."""
    import msgpack

    from vllm_production_stack_amd.router import metrics

    def snap(qps, lat, running, hits):
        fam = lambda name, val, typ="gauge": [name, typ, "doc", [[name, {"server": "http://e1"}, val]]]  # noqa: E731
        return msgpack.packb([fam("vllm:current_qps", qps), fam("vllm:avg_latency", lat),
                              fam("vllm:num_requests_running", running),
                              ["vllm_pii_requests", "counter", "doc", [["vllm_pii_requests_total", {}, hits]]]])

    text = metrics.merge_snapshots(1, [(0, snap(3.0, 2.0, 7, 5)), (1, snap(1.0, 6.0, 9, 2)), (2, snap(0.0, -1.0, 9, 0))]).decode()
    assert _metric(text, "vllm:current_qps") == {"http://e1": 4.0}
    assert _metric(text, "vllm:avg_latency") == {"http://e1": 3.0}  # (3 * 2 + 1 * 6) / 4
    assert _metric(text, "vllm:num_requests_running") == {"http://e1": 9.0}  # worker 1 answers
    assert "vllm_pii_requests_total 7.0" in text
