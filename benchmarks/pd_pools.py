"""Disaggregated prefill with POOLS of replicas: P prefill + D decode engine processes (default
2 + 6, the layout of deploy/examples/values-16-disaggregated-prefill.yaml; on a one-GPU box they
share the card, on a node give each its GPU with --gpus-per-node), the router in
``disaggregated_prefill`` mode balancing both pools, concurrent long prompts.

Reports how the requests spread over the two pools (router log: every request id is routed twice,
once to each pool), pages pushed per prefill replica and by which transport, prefix tokens the
decode replicas found resident, request latency.

    python benchmarks/pd_pools.py [--prefill 2 --decode 6 --requests 24 --prompt-tokens 6000] [--small]
"""

import argparse
import json
import os
import random
import re
import subprocess
import sys
import time
import urllib.request
import uuid
from collections import Counter
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ROUTE_RE = re.compile(r"Routing request (\S+) with session id (\S+) to (\S+) at ")


def http(url, body=None, headers=None, timeout=600):
    data = json.dumps(body).encode() if body is not None else None
    h = {"Content-Type": "application/json"} if data else {}
    h.update(headers or {})
    with urllib.request.urlopen(urllib.request.Request(url, data=data, headers=h), timeout=timeout) as r:  # noqa: S310
        return json.loads(r.read() or b"{}")


def wait_http(url, timeout, proc):
    t0 = time.time()
    while time.time() - t0 < timeout:
        if proc.poll() is not None:
            raise RuntimeError(f"process behind {url} exited with code {proc.returncode}")
        try:
            urllib.request.urlopen(url, timeout=2).read()  # noqa: S310
            return
        except Exception:
            time.sleep(0.5)
    raise RuntimeError(f"{url} did not come up in {timeout}s")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--prefill", type=int, default=2)
    ap.add_argument("--decode", type=int, default=6)
    ap.add_argument("--requests", type=int, default=24)
    ap.add_argument("--concurrency", type=int, default=8)
    ap.add_argument("--prompt-tokens", type=int, default=0)
    ap.add_argument("--max-tokens", type=int, default=16)
    ap.add_argument("--pages", type=int, default=0)
    ap.add_argument("--gpus-per-node", type=int, default=1, help="> 1: engine i takes GPU i modulo this")
    ap.add_argument("--model", default="", help="default: llama-3-8b (tiny-llama with --small)")
    ap.add_argument("--small", action="store_true", help="tiny CPU model (plumbing rehearsal, pages go by TCP)")
    ap.add_argument("--base-port", type=int, default=18900)
    ap.add_argument("--request-timeout", type=float, default=90.0)
    ap.add_argument("--prefill-first", action="store_true", help="start the prefill replicas before the decode ones")
    a = ap.parse_args()
    model = a.model or ("tiny-llama" if a.small else "llama-3-8b")
    tiny = model.startswith("tiny")
    n_tok = a.prompt_tokens or (500 if tiny else 6000)
    pages = a.pages or (64 if tiny else 900)
    log_dir = os.path.join(ROOT, "gpurun_out") if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else "/tmp"
    engines = [(f"prefill-{i}", "prefill", a.base_port + 1 + i) for i in range(a.prefill)] + \
              [(f"decode-{i}", "decode", a.base_port + 1 + a.prefill + i) for i in range(a.decode)]
    procs, logs = [], {}

    def say(msg):
        print(f"[pd_pools {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)

    def spawn(name, cmd, env):
        logs[name] = open(os.path.join(log_dir, f"pdpools_{name}.log"), "w")
        p = subprocess.Popen(cmd, env=env, stdout=logs[name], stderr=subprocess.STDOUT, cwd=ROOT)
        procs.append(p)
        return p

    try:
        order = engines if a.prefill_first else engines[a.prefill:] + engines[:a.prefill]
        for i, (name, role, port) in enumerate(order):
            env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", LMCACHE_ENABLE_NIXL="True", PYTHONFAULTHANDLER="1",
                       LMCACHE_NIXL_ROLE="sender" if role == "prefill" else "receiver",
                       LMCACHE_NIXL_RECEIVER_HOST="127.0.0.1", LMCACHE_LOCAL_CPU="False")
            if a.gpus_per_node > 1:
                env["HIP_VISIBLE_DEVICES"] = str(i % a.gpus_per_node)
            cmd = [sys.executable, "-m", "vllm_production_stack_amd.engine.cli", "serve", model, "--host", "127.0.0.1",
                   "--port", str(port), "--max-model-len", "4096" if tiny else "32000", "--enable-prefix-caching",
                   "--num-gpu-blocks-override", str(pages), "--max-num-seqs", "32", "--log-level", "info",
                   "--kv-transfer-config", json.dumps({"kv_connector": "LMCacheConnectorV1",
                                                       "kv_role": "kv_producer" if role == "prefill" else "kv_consumer"})]
            if a.small:
                cmd += ["--device", "cpu", "--dtype", "float32"]
            say(f"engine {name} on {port}")
            p = spawn(name, cmd, env)
            wait_http(f"http://127.0.0.1:{port}/health", 600, p)
        rp = a.base_port
        router = spawn("router", [
            sys.executable, "-m", "vllm_production_stack_amd.router.app", "--host", "127.0.0.1", "--port", str(rp),
            "--service-discovery", "static",
            "--static-backends", ",".join(f"http://127.0.0.1:{p}" for _, _, p in engines),
            "--static-models", ",".join([model] * len(engines)),
            "--static-model-labels", ",".join(r for _, r, _ in engines),
            "--routing-logic", "disaggregated_prefill", "--prefill-model-labels", "prefill",
            "--decode-model-labels", "decode", "--log-level", "info"], dict(os.environ, PYTHONFAULTHANDLER="1"))
        wait_http(f"http://127.0.0.1:{rp}/health", 60, router)
        say("router up")

        rng = random.Random(0)
        prompts = [[rng.randrange(16, 400) for _ in range(n_tok)] for _ in range(a.requests)]

        def one(i):
            rid = uuid.uuid4().hex
            t0 = time.time()
            out = http(f"http://127.0.0.1:{rp}/v1/completions",
                       {"model": model, "prompt": prompts[i], "max_tokens": a.max_tokens, "temperature": 0},
                       {"X-Request-Id": rid}, timeout=a.request_timeout)
            say(f"request {i} done in {time.time() - t0:.2f}s")
            return rid, time.time() - t0, out

        t0 = time.time()
        try:
            with ThreadPoolExecutor(a.concurrency) as ex:
                outs = list(ex.map(one, range(a.requests)))
        except Exception:
            # a request did not come back: have every process print its threads' stacks into its log
            import signal

            for p in procs:
                if p.poll() is None:
                    p.send_signal(signal.SIGABRT)
            time.sleep(2)
            for name in logs:
                logs[name].flush()
                with open(os.path.join(log_dir, f"pdpools_{name}.log")) as f:
                    tail = f.read().splitlines()[-60:]
                print(f"==== {name}\n" + "\n".join(tail), file=sys.stderr)
            raise
        wall = time.time() - t0
        # a prompt answered by a prefill replica directly (it still holds the KV) is the reference text
        ref = http(f"http://127.0.0.1:{engines[0][2]}/v1/completions",
                   {"model": model, "prompt": prompts[0], "max_tokens": a.max_tokens, "temperature": 0})
        logs["router"].flush()
        by_port = {f"http://127.0.0.1:{p}": n for n, _, p in engines}
        hops: dict[str, list[str]] = {}
        with open(os.path.join(log_dir, "pdpools_router.log")) as f:
            for ln in f:
                m = ROUTE_RE.search(ln)
                if m:
                    hops.setdefault(m.group(1), []).append(by_port.get(m.group(3), m.group(3)))
        two_pools = sum(1 for rid, _, _ in outs if len(hops.get(rid, [])) == 2 and
                        hops[rid][0].startswith("prefill") and hops[rid][1].startswith("decode"))
        pre_use = Counter(h[0] for h in hops.values() if h)
        dec_use = Counter(h[1] for h in hops.values() if len(h) > 1)
        push, resident = {}, {}
        for name, role, port in engines:
            st = http(f"http://127.0.0.1:{port}/debug/stats")
            if role == "prefill":
                push[name] = st.get("pd_push")
            else:
                resident[name] = {"prefix_hit_tokens": st.get("prefix_hits"), "prefill_rows": st["prof"]["rows"]}
        lat = sorted(t for _, t, _ in outs)
        cached = [(o.get("usage", {}).get("prompt_tokens_details") or {}).get("cached_tokens", 0) for _, _, o in outs]
        result = {
            "case": "pd_pools", "model": model, "prefill_replicas": a.prefill, "decode_replicas": a.decode,
            "requests": a.requests, "concurrency": a.concurrency, "prompt_tokens": n_tok,
            "requests_routed_prefill_then_decode": two_pools,
            "prefill_pool_use": dict(pre_use), "decode_pool_use": dict(dec_use),
            "push_by_prefill_replica": push, "decode_replicas_state": resident,
            "cached_tokens_min": min(cached), "cached_tokens_median": sorted(cached)[len(cached) // 2],
            "latency_s_p50": round(lat[len(lat) // 2], 3), "latency_s_max": round(lat[-1], 3),
            "requests_per_s": round(a.requests / wall, 2),
            "first_answer_equals_self_computed": outs[0][2]["choices"][0]["text"] == ref["choices"][0]["text"]}
        ok = (two_pools == a.requests and len(pre_use) == min(a.prefill, a.requests) and
              len(dec_use) == min(a.decode, a.requests) and result["cached_tokens_min"] >= n_tok - 128)
        result["pass"] = bool(ok)
        print(json.dumps(result), flush=True)
        sys.exit(0 if ok else 1)
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
