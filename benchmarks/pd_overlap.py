"""Does a KV push disturb the decode replica it lands on?

One prefill replica and one decode replica (two engine processes; on a one-GPU box they share the
card).  The decode replica runs a steady batch of streams; while it does, the prefill replica pushes
prompts whose KV it already holds (so the only work the push adds is the transfer itself: page
export, peer copy into the landing window, window -> pool copy on the receiver's side stream,
commit).  Reported: the decode replica's inter-token latency (= its step time, seen from the
clients) in the quiet windows and while pushes are in flight, the push sizes, durations and GB/s.

    python benchmarks/pd_overlap.py --streams 32 --push-tokens 8192 --pushes 4

The transports a push can take are ipc (hipIpc landing window), rccl and tcp; two engines on one
card cannot form an RCCL communicator, so a one-GPU run measures ipc (or tcp when forced with
VPSA_PD_TRANSPORT=tcp).
"""

import argparse
from http.client import HTTPConnection
import json
import os
import random
import statistics
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from benchmarks.pd_e2e import http, wait_http  # noqa: E402


def stream_tokens(port: int, body: dict, stamps: list, done: threading.Event) -> None:
    """One streaming completion; appends the arrival time of every chunk."""
    conn = HTTPConnection("127.0.0.1", port, timeout=600)
    try:
        conn.request("POST", "/v1/completions", json.dumps(dict(body, stream=True)),
                     {"Content-Type": "application/json"})
        resp = conn.getresponse()
        while True:
            line = resp.readline()
            if not line:
                break
            if line.startswith(b"data: ") and b"[DONE]" not in line:
                stamps.append(time.time())
    finally:
        conn.close()
        done.set()


def pct(xs, q):
    xs = sorted(xs)
    return xs[min(len(xs) - 1, int(q * len(xs)))] if xs else None


def summarize(gaps):
    return {"n": len(gaps), "mean_ms": round(statistics.fmean(gaps) * 1e3, 3) if gaps else None,
            "p50_ms": round(pct(gaps, 0.5) * 1e3, 3) if gaps else None,
            "p99_ms": round(pct(gaps, 0.99) * 1e3, 3) if gaps else None}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="llama-3-8b")
    ap.add_argument("--streams", type=int, default=32)
    ap.add_argument("--stream-prompt-tokens", type=int, default=2048)
    ap.add_argument("--stream-tokens", type=int, default=700)
    ap.add_argument("--push-tokens", type=int, default=8192, help="8192 tokens of Llama-3-8B KV = 1 GiB")
    ap.add_argument("--pushes", type=int, default=4)
    ap.add_argument("--quiet-s", type=float, default=3.0)
    ap.add_argument("--device", default="cuda")
    ap.add_argument("--gpu-memory-utilization", type=float, default=0.4)
    ap.add_argument("--base-port", type=int, default=18700)
    a = ap.parse_args()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    pp, dp = a.base_port + 1, a.base_port + 2
    common = [sys.executable, "-m", "vllm_production_stack_amd.engine.cli", "serve", a.model, "--host", "127.0.0.1",
              "--max-model-len", "16384", "--enable-prefix-caching", "--log-level", "info"]
    if a.device != "cuda":
        common += ["--device", "cpu", "--dtype", "float32"]
    log_dir = os.environ.get("PD_LOG_DIR") or (os.path.join(ROOT, "gpurun_out") if os.path.isdir(
        os.path.join(ROOT, "gpurun_out")) else "/tmp")
    procs, logs = [], {}

    def say(msg):
        print(f"[pd_overlap {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)

    try:
        for name, port, role in (("decode", dp, "kv_consumer"), ("prefill", pp, "kv_producer")):
            logs[name] = open(os.path.join(log_dir, f"pdov_{name}.log"), "w")
            util = a.gpu_memory_utilization if name == "decode" else 2 * a.gpu_memory_utilization
            say(f"starting {name} engine on port {port}")
            procs.append(subprocess.Popen(
                common + ["--gpu-memory-utilization", str(util), "--port", str(port), "--kv-transfer-config",
                          json.dumps({"kv_connector": "LMCacheConnectorV1", "kv_role": role})],
                env=env, cwd=ROOT, stdout=logs[name], stderr=subprocess.STDOUT))
            wait_http(f"http://127.0.0.1:{port}/health", 600, procs[-1])
        rng = random.Random(1)
        # the prompts to push: computed on the prefill replica now, so that the timed pushes find
        # them resident and add no prefill work to the card
        push_prompts = [[rng.randrange(16, 400) for _ in range(a.push_tokens)] for _ in range(a.pushes)]
        for i, ids in enumerate(push_prompts):
            http(f"http://127.0.0.1:{pp}/v1/completions", {"model": a.model, "prompt": ids, "max_tokens": 1,
                                                           "temperature": 0}, timeout=300)
            say(f"prefill replica holds prompt {i}")
        # the decode load
        stamps = [[] for _ in range(a.streams)]
        dones = [threading.Event() for _ in range(a.streams)]
        for s in range(a.streams):
            ids = [rng.randrange(16, 400) for _ in range(a.stream_prompt_tokens)]
            body = {"model": a.model, "prompt": ids, "max_tokens": a.stream_tokens, "temperature": 0,
                    "ignore_eos": True}
            threading.Thread(target=stream_tokens, args=(dp, body, stamps[s], dones[s]), daemon=True).start()
        t0 = time.time()
        while min(len(x) for x in stamps) < 5:
            if time.time() - t0 > 300:
                raise RuntimeError("decode streams did not start")
            time.sleep(0.1)
        say("all streams decoding")
        time.sleep(1.0)
        q0 = time.time()
        time.sleep(a.quiet_s)
        q1 = time.time()
        pushes = []
        for i, ids in enumerate(push_prompts):
            body = {"model": a.model, "prompt": ids, "max_tokens": 1, "temperature": 0,
                    "kv_transfer_params": {"do_remote_decode": True, "decode_target": f"http://127.0.0.1:{dp}",
                                           "request_id": f"pdov-{i}"}}
            t = time.time()
            out = http(f"http://127.0.0.1:{pp}/v1/completions", body, timeout=120)
            pushes.append({"start": t, "end": time.time(), "kv_transfer_params": out.get("kv_transfer_params")})
            say(f"push {i}: {pushes[-1]['end'] - t:.3f}s")
            time.sleep(0.25)
        q2 = time.time()
        time.sleep(a.quiet_s)
        q3 = time.time()
        if any(d.is_set() for d in dones):
            say("warning: a stream ended inside the measurement (raise --stream-tokens)")
        ps = http(f"http://127.0.0.1:{pp}/debug/stats")
        ds = http(f"http://127.0.0.1:{dp}/debug/stats")

        def gaps_in(lo, hi):
            out = []
            for st in stamps:
                for x, y in zip(st, st[1:]):
                    if lo <= x and y <= hi:
                        out.append(y - x)
            return out

        def gaps_touching(intervals):
            out = []
            for st in stamps:
                for x, y in zip(st, st[1:]):
                    if any(x < e and y > s for s, e in intervals):
                        out.append(y - x)
            return out

        quiet = gaps_in(q0, q1) + gaps_in(q2 + 0.3, q3)
        during = gaps_touching([(p["start"], p["end"]) for p in pushes])
        whole = gaps_in(q1, q2)
        push = ps.get("pd_push", {})
        gb = push.get("pages", 0) * ps.get("page_bytes", 0) / 1e9
        sq, sd, sw = summarize(quiet), summarize(during), summarize(whole)
        print(json.dumps({
            "case": "pd_overlap", "model": a.model, "streams": a.streams, "stream_prompt_tokens": a.stream_prompt_tokens,
            "push_tokens_each": a.push_tokens, "pushes": a.pushes, "push": push, "pushed_gb": round(gb, 3),
            "push_gb_per_s": round(gb / push["seconds"], 2) if push.get("seconds") else None,
            "push_wall_s": [round(p["end"] - p["start"], 4) for p in pushes],
            "decode_itl_quiet": sq, "decode_itl_steps_overlapping_a_push": sd, "decode_itl_whole_push_phase": sw,
            "inflation_steps_overlapping_a_push_pct": round((sd["mean_ms"] / sq["mean_ms"] - 1) * 100, 2) if during and quiet else None,
            "inflation_whole_push_phase_pct": round((sw["mean_ms"] / sq["mean_ms"] - 1) * 100, 2) if whole and quiet else None,
            "decode_prefix_hit_tokens": ds.get("prefix_hits"), "decode_kv_receiver": ds.get("kv_receiver"),
            "note": "both replicas share one GPU: the copy of a push and the decode steps compete for the same HBM"}), flush=True)
    finally:
        for p in procs:
            p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20)
            except Exception:
                p.kill()
        for f in logs.values():
            f.close()


if __name__ == "__main__":
    main()
