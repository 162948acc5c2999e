"""Disaggregated prefill end to end with real engine processes: one prefill replica and one decode
replica (separate processes; on a single-GPU box both share the card, on a node they take GPUs 0 and
1), the router in disaggregated_prefill mode, long prompts through the OpenAI API.

Reports: how the KV pages travelled (ipc = hipIpc peer copy, tcp), pages and GB/s pushed, how many
prompt tokens the decode replica found resident, TTFT split, and whether the decode replica's answer
equals the answer of a replica that computed the prompt itself.

    python benchmarks/pd_e2e.py --model llama-3-8b --prompt-tokens 8000 --requests 6
"""

import argparse
import json
import os
import subprocess
import sys
import time
import urllib.request

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def http(url, body=None, timeout=600):
    data = json.dumps(body).encode() if body is not None else None
    req = urllib.request.Request(url, data=data, headers={"Content-Type": "application/json"} if data else {})
    with urllib.request.urlopen(req, timeout=timeout) as r:  # noqa: S310
        return json.loads(r.read() or b"{}")


def wait_http(url, timeout, proc=None):
    t0 = time.time()
    while time.time() - t0 < timeout:
        if proc is not None and proc.poll() is not None:
            raise RuntimeError(f"process behind {url} exited with code {proc.returncode}")
        try:
            urllib.request.urlopen(url, timeout=2).read()  # noqa: S310
            return
        except Exception:
            time.sleep(0.5)
    raise RuntimeError(f"{url} did not come up")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="llama-3-8b")
    ap.add_argument("--prompt-tokens", type=int, default=8000)
    ap.add_argument("--requests", type=int, default=6)
    ap.add_argument("--max-tokens", type=int, default=16)
    ap.add_argument("--device", default="cuda")
    ap.add_argument("--gpu-memory-utilization", type=float, default=0.4)
    ap.add_argument("--base-port", type=int, default=18500)
    ap.add_argument("--two-gpus", action="store_true", help="prefill on GPU 0, decode on GPU 1")
    a = ap.parse_args()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    pp, dp, rp = a.base_port + 1, a.base_port + 2, a.base_port
    common = [sys.executable, "-m", "vllm_production_stack_amd.engine.cli", "serve", a.model, "--host", "127.0.0.1",
              "--max-model-len", "32000", "--enable-prefix-caching", "--log-level", "info"]
    if a.device != "cuda":
        common += ["--device", "cpu", "--dtype", "float32"]
    procs = []
    logs = {}
    log_dir = os.environ.get("PD_LOG_DIR") or (os.path.join(ROOT, "gpurun_out") if os.path.isdir(
        os.path.join(ROOT, "gpurun_out")) else "/tmp")

    def say(msg):
        print(f"[pd_e2e {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)

    try:
        for name, port, role, gpu in (("decode", dp, "receiver", 1), ("prefill", pp, "sender", 0)):
            e = dict(env, LMCACHE_ENABLE_NIXL="True", LMCACHE_NIXL_ROLE=role, LMCACHE_NIXL_RECEIVER_HOST="127.0.0.1",
                     LMCACHE_NIXL_RECEIVER_PORT="55555", LMCACHE_LOCAL_CPU="False")
            if a.two_gpus:
                e["HIP_VISIBLE_DEVICES"] = str(gpu)
            kv_role = "kv_producer" if role == "sender" else "kv_consumer"
            logs[name] = open(os.path.join(log_dir, f"pd_{name}.log"), "w")
            say(f"starting {name} engine on port {port}")
            # --gpu-memory-utilization caps the device's TOTAL use: on a shared card the second
            # replica's cap has to include what the first one already took
            util = a.gpu_memory_utilization if (a.two_gpus or name == "decode") else 2 * a.gpu_memory_utilization
            procs.append(subprocess.Popen(
                common + ["--gpu-memory-utilization", str(util), "--port", str(port), "--kv-transfer-config",
                          json.dumps({"kv_connector": "LMCacheConnectorV1", "kv_role": kv_role})],
                env=e, cwd=ROOT, stdout=logs[name], stderr=subprocess.STDOUT))
            wait_http(f"http://127.0.0.1:{port}/health", 600, procs[-1])  # one at a time: each sizes its pool from free HBM
        procs.append(subprocess.Popen(
            [sys.executable, "-m", "vllm_production_stack_amd.router.app", "--host", "127.0.0.1", "--port", str(rp),
             "--service-discovery", "static", "--static-backends", f"http://127.0.0.1:{pp},http://127.0.0.1:{dp}",
             "--static-models", f"{a.model},{a.model}", "--static-model-labels", "prefill,decode",
             "--routing-logic", "disaggregated_prefill", "--prefill-model-labels", "prefill",
             "--decode-model-labels", "decode", "--log-level", "warning"], env=env, cwd=ROOT))
        wait_http(f"http://127.0.0.1:{rp}/health", 60)
        say("router up; sending requests")

        # prompts as token-id lists (completions API accepts them), distinct per request
        import random

        rng = random.Random(0)
        results = []
        for i in range(a.requests):
            ids = [rng.randrange(16, 400) for _ in range(a.prompt_tokens)]
            body = {"model": a.model, "prompt": ids, "max_tokens": a.max_tokens, "temperature": 0}
            t0 = time.time()
            out = http(f"http://127.0.0.1:{rp}/v1/completions", body, timeout=120)
            dt = time.time() - t0
            say(f"request {i}: {dt:.3f}s")
            # the same prompt straight to the prefill replica (it holds the KV): reference answer
            ref = http(f"http://127.0.0.1:{pp}/v1/completions", body, timeout=120)
            usage = out.get("usage", {})
            results.append({"seconds": round(dt, 4), "same_text_as_self_computed": out["choices"][0]["text"] == ref["choices"][0]["text"],
                            "cached_tokens": (usage.get("prompt_tokens_details") or {}).get("cached_tokens")})
        # time to the first token of a NEW prompt: through the pair (prefill replica computes, pages
        # travel, decode replica answers) against one replica doing everything itself
        ttft_pd, ttft_single = [], []
        for i in range(a.requests):
            for path, sink in ((rp, ttft_pd), (pp, ttft_single)):
                ids = [rng.randrange(16, 400) for _ in range(a.prompt_tokens)]
                t0 = time.time()
                http(f"http://127.0.0.1:{path}/v1/completions", {"model": a.model, "prompt": ids, "max_tokens": 1,
                                                                 "temperature": 0}, timeout=120)
                sink.append(round(time.time() - t0, 4))
        ttft_pd.sort()
        ttft_single.sort()
        ps = http(f"http://127.0.0.1:{pp}/debug/stats")
        ds = http(f"http://127.0.0.1:{dp}/debug/stats")
        push = ps.get("pd_push", {})
        gb = push.get("pages", 0) * ps.get("page_bytes", 0) / 1e9
        print(json.dumps({
            "case": "pd_e2e", "model": a.model, "prompt_tokens": a.prompt_tokens, "requests": a.requests,
            "push": push, "pushed_gb": round(gb, 3),
            "push_gb_per_s": round(gb / push["seconds"], 2) if push.get("seconds") else None,
            "decode_prefix_hit_tokens": ds.get("prefix_hits"), "decode_prefix_query_tokens": ds.get("prefix_queries"),
            "decode_prefill_rows": ds["prof"]["rows"], "prefill_rows": ps["prof"]["rows"],
            "ttft_pair_s": ttft_pd, "ttft_single_replica_s": ttft_single,
            "ttft_pair_over_single_median": round(ttft_pd[len(ttft_pd) // 2] / ttft_single[len(ttft_single) // 2], 3),
            "push_tail_ms_per_request": round(1e3 * push.get("tail_seconds", 0.0) / max(push.get("pushes", 1), 1), 2),
            "requests_detail": results}), flush=True)
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
        for name in logs:
            with open(os.path.join(log_dir, f"pd_{name}.log")) as f:
                tail = [l for l in f.read().splitlines() if "pushed" in l or "ERROR" in l or "rror" in l][-5:]
            for l in tail:
                print(f"[{name}] {l}", file=sys.stderr)


if __name__ == "__main__":
    main()
