"""Router data-plane stress: N fake engines streaming tokens, C concurrent streamed chats
through the router; reports relay throughput (SSE events/s), request rate and latencies.

Counterpart of /root/reference/tests/e2e/stress-test.sh (ApacheBench against the router
with static backends), with live streaming backends instead of refused connections.

    python benchmarks/router_stress.py --engines 8 --concurrency 320 --tokens 100 --speed 50
"""
import argparse
import asyncio
import json
import os
import subprocess
import sys
import time

import aiohttp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


async def client(url, n_req, conc, tokens, routing_header):
    sem = asyncio.Semaphore(conc)
    lat, events = [], [0]
    async with aiohttp.ClientSession(connector=aiohttp.TCPConnector(limit=0)) as s:
        async def one(i):
            async with sem:
                t0 = time.time()
                body = {"model": "fake", "messages": [{"role": "user", "content": "x " * 50}],
                        "max_tokens": tokens, "stream": True}
                async with s.post(url + "/v1/chat/completions", json=body,
                                  headers={routing_header: f"u{i % conc}"}) as r:
                    assert r.status == 200, r.status
                    async for chunk in r.content.iter_any():
                        events[0] += chunk.count(b"data: ")
                lat.append(time.time() - t0)
        t0 = time.time()
        await asyncio.gather(*[one(i) for i in range(n_req)])
        dt = time.time() - t0
    lat.sort()
    return {"requests": n_req, "seconds": round(dt, 3), "req_per_s": round(n_req / dt, 1),
            "sse_events_per_s": round(events[0] / dt, 1),
            "p50_latency_s": round(lat[len(lat) // 2], 4), "p99_latency_s": round(lat[int(len(lat) * 0.99)], 4)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--engines", type=int, default=4)
    ap.add_argument("--concurrency", type=int, default=200)
    ap.add_argument("--requests", type=int, default=2000)
    ap.add_argument("--tokens", type=int, default=20)
    ap.add_argument("--speed", type=float, default=500.0)
    ap.add_argument("--routing-logic", default="session")
    ap.add_argument("--base-port", type=int, default=19100)
    ap.add_argument("--max-imbalance", type=float, default=None,
                    help="fail when the per-backend share of routing decisions (read from the router "
                         "log) deviates from the mean by more than this fraction")
    a = ap.parse_args()
    procs = []
    router_log = None
    if a.max_imbalance is not None:
        import tempfile

        router_log = tempfile.NamedTemporaryFile("w+", suffix="-router.log", delete=False)
    try:
        for i in range(a.engines):
            procs.append(subprocess.Popen([sys.executable, "-m", "vllm_production_stack_amd.testing.fake_engine",
                                           "--port", str(a.base_port + 1 + i), "--model", "fake",
                                           "--speed", str(a.speed)], cwd=ROOT))
        backends = ",".join(f"http://127.0.0.1:{a.base_port + 1 + i}" for i in range(a.engines))
        procs.append(subprocess.Popen([sys.executable, "-m", "vllm_production_stack_amd.router.app", "--port",
                                       str(a.base_port), "--host", "127.0.0.1", "--service-discovery", "static",
                                       "--static-backends", backends, "--static-models", ",".join(["fake"] * a.engines),
                                       "--routing-logic", a.routing_logic, "--session-key", "x-user-id",
                                       "--log-level", "info" if router_log else "warning"], cwd=ROOT,
                                      stdout=router_log, stderr=subprocess.STDOUT if router_log else None))
        time.sleep(2.0)
        res = asyncio.run(client(f"http://127.0.0.1:{a.base_port}", a.requests, a.concurrency, a.tokens, "x-user-id"))
        res.update(engines=a.engines, concurrency=a.concurrency, tokens_per_request=a.tokens, routing=a.routing_logic)
        if router_log is not None:
            import collections
            import re

            time.sleep(0.5)
            router_log.flush()
            with open(router_log.name, errors="replace") as f:
                counts = collections.Counter(re.findall(r"Routing request \S+ with session id \S+ to (\S+) at", f.read()))
            mean = sum(counts.values()) / max(1, a.engines)
            worst = max(abs(counts.get(f"http://127.0.0.1:{a.base_port + 1 + i}", 0) - mean) / mean
                        for i in range(a.engines)) if mean else 1.0
            res.update(routing_decisions=dict(counts), imbalance=round(worst, 4))
        print(json.dumps(res))
        if router_log is not None and (worst > a.max_imbalance or sum(counts.values()) < a.requests):
            sys.exit(f"routing imbalance {worst:.2%} (limit {a.max_imbalance:.0%}) over {sum(counts.values())} decisions")
    finally:
        for p in procs:
            p.terminate()
        for p in procs:
            p.wait(timeout=10)


if __name__ == "__main__":
    main()
