"""QPS sweep of the multi-round-QA benchmark against a locally launched stack (router + engine
replicas on this node's GPUs) - the procedure of the reference's run_single.sh / run.sh
(/root/reference/benchmarks/multi-round-qa/run_single.sh:12-39, run.sh:14-79): for each offered QPS
a fresh set of users runs for a fixed time; mean / p50 / p90 TTFT and token rates are recorded.

    python benchmarks/multi_round_qa/sweep.py --gpus 1 --num-users 15 --num-rounds 20 \\
        --qps 0.1,0.3,0.5,0.7,0.9,1.1 --time 100 --out sweep.jsonl
"""

import argparse
import asyncio
import io
import json
import os
import subprocess
import sys
import time
import urllib.request

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import multi_round_qa as mrq  # noqa: E402


def wait_http(url: str, timeout: float) -> None:
    t0 = time.time()
    while time.time() - t0 < timeout:
        try:
            with urllib.request.urlopen(url, timeout=2) as r:  # noqa: S310
                if r.status == 200:
                    return
        except Exception:
            time.sleep(0.5)
    raise RuntimeError(f"{url} did not come up in {timeout}s")


def engine_latency_sums(url: str) -> dict:
    """Sum and count of the engine's own latency histograms (arrival at the engine to first token,
    time in its waiting queue): what a replica saw, to set against what the client saw."""
    out = {}
    try:
        with urllib.request.urlopen(url + "/metrics", timeout=5) as r:  # noqa: S310
            for line in r.read().decode().splitlines():
                for key, name in (("ttft", "vllm:time_to_first_token_seconds"),
                                  ("queue", "vllm:request_queue_time_seconds")):
                    for part in ("sum", "count"):
                        if line.startswith(f"{name}_{part}"):
                            out[f"{key}_{part}"] = out.get(f"{key}_{part}", 0.0) + float(line.rsplit(" ", 1)[1])
    except Exception:  # noqa: BLE001
        pass
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--model", default="llama-3-8b")
    ap.add_argument("--num-users", type=int, default=15)
    ap.add_argument("--num-rounds", type=int, default=20)
    ap.add_argument("--shared-system-prompt", type=int, default=1000)
    ap.add_argument("--user-history-prompt", type=int, default=20000)
    ap.add_argument("--answer-len", type=int, default=100)
    ap.add_argument("--qps", default="0.1,0.3,0.5,0.7,0.9,1.1")
    ap.add_argument("--time", type=int, default=100)
    ap.add_argument("--routing-logic", default="session")
    ap.add_argument("--cpu-offload-gb", type=float, default=0.0)
    ap.add_argument("--no-prefix-caching", action="store_true",
                    help="the 'naive' arm of the A/B: every turn recomputes its whole history")
    ap.add_argument("--no-chunked-prefill", action="store_true",
                    help="whole prompts in one step (the reference's benchmark configuration)")
    ap.add_argument("--num-gpu-blocks", type=int, default=0,
                    help="KV pages per replica (default: what fits); a small pool forces evictions to the tiers")
    ap.add_argument("--base-port", type=int, default=18300)
    ap.add_argument("--out", default="sweep.jsonl")
    ap.add_argument("--device", default="cuda")
    a = ap.parse_args()

    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = []
    try:
        for g in range(a.gpus):
            e = dict(env)
            if a.device == "cuda":
                e["HIP_VISIBLE_DEVICES"] = str(g)
            if a.cpu_offload_gb > 0:
                e.update(LMCACHE_LOCAL_CPU="True", LMCACHE_MAX_LOCAL_CPU_SIZE=str(a.cpu_offload_gb))
            cmd = [sys.executable, "-m", "vllm_production_stack_amd.engine.cli", "serve", a.model,
                   "--host", "127.0.0.1", "--port", str(a.base_port + 1 + g), "--max-model-len", "32000",
                   "--max-num-batched-tokens", "32000" if a.no_chunked_prefill else "8192",
                   "--disable-log-requests", "--log-level", "warning",
                   "--no-enable-prefix-caching" if a.no_prefix_caching else "--enable-prefix-caching",
                   "--no-enable-chunked-prefill" if a.no_chunked_prefill else "--enable-chunked-prefill"]
            if a.num_gpu_blocks:
                cmd += ["--num-gpu-blocks-override", str(a.num_gpu_blocks)]
            if a.device != "cuda":
                cmd += ["--device", "cpu", "--dtype", "float32"]
            procs.append(subprocess.Popen(cmd, env=e, cwd=ROOT))
        backends = ",".join(f"http://127.0.0.1:{a.base_port + 1 + g}" for g in range(a.gpus))
        procs.append(subprocess.Popen(
            [sys.executable, "-m", "vllm_production_stack_amd.router.app", "--host", "127.0.0.1",
             "--port", str(a.base_port), "--service-discovery", "static", "--static-backends", backends,
             "--static-models", ",".join([a.model] * a.gpus), "--routing-logic", a.routing_logic,
             "--session-key", "x-user-id", "--log-level", "warning"], env=env, cwd=ROOT))
        for g in range(a.gpus):
            wait_http(f"http://127.0.0.1:{a.base_port + 1 + g}/health", 600)
        wait_http(f"http://127.0.0.1:{a.base_port}/health", 60)

        uid = 0
        with open(a.out, "a") as out:
            for q in [float(x) for x in a.qps.split(",")]:
                args = mrq.build_parser().parse_args([
                    "--num-users", str(a.num_users), "--shared-system-prompt", str(a.shared_system_prompt),
                    "--user-history-prompt", str(a.user_history_prompt), "--answer-len", str(a.answer_len),
                    "--num-rounds", str(a.num_rounds), "--qps", str(q), "--model", a.model,
                    "--base-url", f"http://127.0.0.1:{a.base_port}", "--time", str(a.time),
                    "--init-user-id", str(uid), "--request-with-user-id", "--log-interval", "1000000",
                    "--output", os.devnull])
                t0 = time.time()
                before = engine_latency_sums(f"http://127.0.0.1:{a.base_port + 1}")
                sessions = asyncio.run(mrq.run_open_loop(args))
                after = engine_latency_sums(f"http://127.0.0.1:{a.base_port + 1}")
                uid += len(sessions) + 1
                rows = mrq.rows_of(sessions)
                s = mrq.summarize(rows, t0, time.time(), 0, q, out=io.StringIO())
                rec = {"qps": q, "gpus": a.gpus, "num_users": a.num_users, "num_rounds": a.num_rounds,
                       "history_tokens": a.user_history_prompt, "system_tokens": a.shared_system_prompt,
                       "answer_len": a.answer_len, "seconds": a.time, "routing": a.routing_logic,
                       "prefix_caching": not a.no_prefix_caching, "chunked_prefill": not a.no_chunked_prefill,
                       "cpu_offload_gb": a.cpu_offload_gb,
                       **{k: round(v, 4) if isinstance(v, float) else v for k, v in s.items()}}
                n_eng = after.get("ttft_count", 0) - before.get("ttft_count", 0)
                if n_eng > 0:  # replica 0's own view (includes the client's ten warm-up prompts)
                    rec["engine0_mean_ttft_s"] = round((after["ttft_sum"] - before["ttft_sum"]) / n_eng, 4)
                    rec["engine0_mean_queue_s"] = round(
                        (after.get("queue_sum", 0) - before.get("queue_sum", 0)) / n_eng, 4)
                print(json.dumps(rec), flush=True)
                out.write(json.dumps(rec) + "\n")
                out.flush()
    finally:
        for p in procs:
            p.terminate()
        for p in procs:
            try:
                p.wait(timeout=15)
            except Exception:
                p.kill()


if __name__ == "__main__":
    main()
