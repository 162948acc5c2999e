"""Router data-plane stress: N fake engines streaming tokens, C concurrent streamed chats
through the router; reports relay throughput (SSE events/s), request rate and latencies.

Counterpart of /root/reference/tests/e2e/stress-test.sh (ApacheBench against the router
with static backends), with live streaming backends instead of refused connections.

    python benchmarks/router_stress.py --engines 8 --concurrency 320 --tokens 100 --speed 50

``--speed`` is per stream: offered load = concurrency x speed events/s, so the ceiling of the
relay shows only when that is far above what comes out.  ``--client-procs`` spreads the client
over processes (one event loop counts ~40k events/s at most), ``--router-workers`` starts the
router with that many processes, ``--pin`` gives router, fake engines and clients disjoint CPU
sets, and the router's own CPU time per relayed event is reported from /proc.
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


async def client(url, n_req, conc, tokens, routing_header, offset=0):
    sem = asyncio.Semaphore(conc)
    lat, events = [], [0]
    async with aiohttp.ClientSession(connector=aiohttp.TCPConnector(limit=0)) as s:
        async def one(i):
            async with sem:
                t0 = time.time()
                body = {"model": "fake", "messages": [{"role": "user", "content": "x " * 50}],
                        "max_tokens": tokens, "stream": True}
                async with s.post(url + "/v1/chat/completions", json=body,
                                  headers={routing_header: f"u{offset + i % conc}"}) as r:
                    assert r.status == 200, r.status
                    async for chunk in r.content.iter_any():
                        events[0] += chunk.count(b"data: ")
                lat.append(time.time() - t0)
        t0 = time.time()
        await asyncio.gather(*[one(i) for i in range(n_req)])
        dt = time.time() - t0
    lat.sort()
    return {"requests": n_req, "seconds": round(dt, 3), "req_per_s": round(n_req / dt, 1),
            "sse_events_per_s": round(events[0] / dt, 1), "events": events[0], "lat": lat,
            "p50_latency_s": round(lat[len(lat) // 2], 4), "p99_latency_s": round(lat[int(len(lat) * 0.99)], 4)}


def _client_entry(cpus, url, n_req, conc, tokens, offset, q):
    if cpus:
        os.sched_setaffinity(0, cpus)
    q.put(asyncio.run(client(url, n_req, conc, tokens, "x-user-id", offset)))


def _cpu_seconds(pids):
    import psutil

    total = 0.0
    for pid in pids:
        try:
            procs = [psutil.Process(pid)] + psutil.Process(pid).children(recursive=True)
        except psutil.NoSuchProcess:
            continue
        for p in procs:
            try:
                t = p.cpu_times()
                total += t.user + t.system
            except psutil.NoSuchProcess:
                pass
    return total


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--router-workers", type=int, default=1)
    ap.add_argument("--client-procs", type=int, default=1)
    ap.add_argument("--pin", action="store_true", help="disjoint CPU sets for router / engines / clients")
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
        ncpu = os.cpu_count() or 8
        sets = None
        if a.pin and ncpu >= 4:
            k = max(1, min(a.router_workers, ncpu // 3))
            rest = list(range(k, ncpu))
            half = len(rest) // 2
            sets = {"router": list(range(k)), "engines": rest[:half], "clients": rest[half:]}

        def pinned(kind):
            if sets is None:
                return None
            return lambda: os.sched_setaffinity(0, sets[kind])

        for i in range(a.engines):
            procs.append(subprocess.Popen([sys.executable, "-m", "vllm_production_stack_amd.testing.fake_engine",
                                           "--port", str(a.base_port + 1 + i), "--model", "fake",
                                           "--speed", str(a.speed)], cwd=ROOT, preexec_fn=pinned("engines")))
        backends = ",".join(f"http://127.0.0.1:{a.base_port + 1 + i}" for i in range(a.engines))
        router = subprocess.Popen([sys.executable, "-m", "vllm_production_stack_amd.router.app", "--port",
                                   str(a.base_port), "--host", "127.0.0.1", "--service-discovery", "static",
                                   "--static-backends", backends, "--static-models", ",".join(["fake"] * a.engines),
                                   "--routing-logic", a.routing_logic, "--session-key", "x-user-id",
                                   "--router-workers", str(a.router_workers),
                                   "--log-level", "info" if router_log else "warning"], cwd=ROOT,
                                  stdout=router_log, stderr=subprocess.STDOUT if router_log else None,
                                  preexec_fn=pinned("router"))
        procs.append(router)
        time.sleep(3.0)
        url = f"http://127.0.0.1:{a.base_port}"
        cpu0 = _cpu_seconds([router.pid])
        t0 = time.time()
        if a.client_procs <= 1:
            parts = [asyncio.run(client(url, a.requests, a.concurrency, a.tokens, "x-user-id"))]
        else:
            import multiprocessing as mp

            ctx = mp.get_context("fork")
            q = ctx.Queue()
            n, c = a.requests // a.client_procs, max(1, a.concurrency // a.client_procs)
            cps = [ctx.Process(target=_client_entry, args=(sets["clients"] if sets else None, url, n, c,
                                                           a.tokens, j * c, q)) for j in range(a.client_procs)]
            for cp in cps:
                cp.start()
            parts = [q.get() for _ in cps]
            for cp in cps:
                cp.join()
        dt = time.time() - t0
        cpu = _cpu_seconds([router.pid]) - cpu0
        lat = sorted(x for p in parts for x in p["lat"])
        events = sum(p["events"] for p in parts)
        nreq = sum(p["requests"] for p in parts)
        res = {"requests": nreq, "seconds": round(dt, 3), "req_per_s": round(nreq / dt, 1),
               "sse_events_per_s": round(events / dt, 1),
               "offered_events_per_s": round(a.concurrency * a.speed, 1),
               "p50_latency_s": round(lat[len(lat) // 2], 4), "p99_latency_s": round(lat[int(len(lat) * 0.99)], 4),
               "router_workers": a.router_workers, "client_procs": a.client_procs,
               "router_cpu_s": round(cpu, 2), "router_cores_busy": round(cpu / dt, 2),
               "router_cpu_us_per_event": round(cpu / max(1, events) * 1e6, 1),
               "cpu_sets": sets}
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
