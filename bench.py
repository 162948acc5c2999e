"""Headline benchmark: multi-round QA on Llama-3-8B through the full serving stack.

    python bench.py --gpus N --steps K --warmup W

One engine replica per GPU (one process per GPU; for N > 1 the driver launches this file
under ``torch.distributed.run`` and every rank hosts the replica of its GPU).  Rank 0 also
starts the router (prefix-aware routing over the N replicas, static discovery) and the
load-generator processes, which talk to the router over HTTP exactly like a client of a
deployed stack.

Workload (reference: /root/reference/benchmarks/multi-round-qa/run.sh:14-18,43-48 and
tutorials/08-benchmark-multi-round-qa-multi-gpu.md:36-72): users-per-GPU x N concurrent
users (default 40 per GPU = 320 on 8 GPUs), shared system prompt 1000 tokens, per-user
history 20000 tokens, 100-token answers, streamed chat completions with temperature 0.
A *step* is one conversation round: every user sends its next turn and the round ends when
every answer has been streamed back.  Warm-up rounds use separate throw-away users; the
timed region starts with fresh users, so round 1 of the timed region is the cold 21k-token
prefill of every user and later rounds hit the KV cache.  Weak scaling: per-GPU users fixed.

Prints one JSON line: value = aggregate output tokens/s over all GPUs (the throughput half
of the headline metric); the TTFT half (p50/mean/p90, overall and warm-round) rides along
in the same line.  Weights are random-init Llama-3-8B (bf16), prompts synthetic.
"""

from __future__ import annotations

import argparse
import asyncio
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _default_base_port() -> int:
    """Router on base, engine of rank r on base + 1 + r.  Derived from the rendezvous port when
    there is one, so that back-to-back launches with different rendezvous ports never meet a
    listener of the previous run that is still shutting down."""
    if os.environ.get("VPSA_BENCH_PORT"):
        return int(os.environ["VPSA_BENCH_PORT"])
    try:
        mp = int(os.environ.get("MASTER_PORT", "0"))
    except ValueError:
        mp = 0
    return 18100 + (mp % 500) * 20 if mp else 18100


def parse() -> argparse.Namespace:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10, help="timed conversation rounds")
    ap.add_argument("--warmup", type=int, default=2, help="untimed warm-up rounds (throw-away users)")
    ap.add_argument("--model", default="llama-3-8b")
    ap.add_argument("--users-per-gpu", type=int, default=40)
    ap.add_argument("--warmup-users-per-gpu", type=int, default=4)
    ap.add_argument("--sys-len", type=int, default=1000)
    ap.add_argument("--history-len", type=int, default=20000)
    ap.add_argument("--answer-len", type=int, default=100)
    ap.add_argument("--max-model-len", type=int, default=32000)
    ap.add_argument("--max-num-batched-tokens", type=int, default=8192)
    ap.add_argument("--routing-logic", default="prefixaware")
    ap.add_argument("--cpu-offload-gb", type=float, default=0.0)
    ap.add_argument("--num-gpu-blocks", type=int, default=0,
                    help="cap the HBM KV pool (pages of 64 tokens) to exercise the host tier; 0 = size from HBM")
    ap.add_argument("--prefill-step-ms", type=float, default=30.0,
                    help="engine's latency-aware prefill admission target (0 = token budget only)")
    ap.add_argument("--base-port", type=int, default=_default_base_port())
    ap.add_argument("--clients", type=int, default=0, help="load-generator processes (0 = auto)")
    ap.add_argument("--device", default="cuda", help="cpu runs a plumbing rehearsal (tiny models)")
    # internal: load-generator mode
    ap.add_argument("--client", action="store_true")
    ap.add_argument("--base-url", default="")
    ap.add_argument("--user-ids", default="")
    ap.add_argument("--warmup-user-ids", default="")
    return ap.parse_args()


# ------------------------------------------------------------------ load generator process
def client_main(a: argparse.Namespace) -> None:
    sys.path.insert(0, os.path.join(ROOT, "benchmarks", "multi_round_qa"))
    import multi_round_qa as mrq

    async def go():
        sys.stdin.readline()  # "WARMUP": every engine replica is listening
        wu = [int(x) for x in a.warmup_user_ids.split(",") if x]
        if wu and a.warmup > 0:
            await mrq.run_rounds(a.base_url, a.model, wu, a.warmup, a.sys_len, a.history_len,
                                 a.answer_len)
        print("READY", flush=True)
        sys.stdin.readline()  # "GO"
        users = [int(x) for x in a.user_ids.split(",") if x]
        round_times = []
        t0 = time.time()
        sessions = await mrq.run_rounds(a.base_url, a.model, users, a.steps, a.sys_len,
                                        a.history_len, a.answer_len,
                                        on_round=lambda r, dt, s: round_times.append(dt))
        t1 = time.time()
        rows = mrq.rows_of(sessions)
        print("RESULT " + json.dumps({"t0": t0, "t1": t1, "rows": rows, "round_times": round_times}),
              flush=True)

    asyncio.run(go())


# ------------------------------------------------------------------ engine rank
def spawn_children(a: argparse.Namespace, n: int):
    """Router + load generators.  Called before this process touches the GPU."""
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK"):
        env.pop(k, None)
    backends = ",".join(f"http://127.0.0.1:{a.base_port + 1 + r}" for r in range(n))
    router_cmd = [sys.executable, "-m", "vllm_production_stack_amd.router.app",
                  "--host", "127.0.0.1", "--port", str(a.base_port),
                  "--service-discovery", "static", "--static-backends", backends,
                  "--static-models", ",".join([a.model] * n),
                  "--routing-logic", a.routing_logic, "--session-key", "x-user-id",
                  "--engine-stats-interval", "5", "--log-level", "warning"]
    router = subprocess.Popen(router_cmd, env=env, cwd=ROOT)
    n_clients = a.clients or max(1, min(4, (n + 1) // 2))
    users = list(range(a.users_per_gpu * n))
    wusers = list(range(10_000_000, 10_000_000 + a.warmup_users_per_gpu * n))
    clients = []
    for c in range(n_clients):
        cmd = [sys.executable, os.path.abspath(__file__), "--client",
               "--base-url", f"http://127.0.0.1:{a.base_port}", "--model", a.model,
               "--steps", str(a.steps), "--warmup", str(a.warmup),
               "--sys-len", str(a.sys_len), "--history-len", str(a.history_len),
               "--answer-len", str(a.answer_len),
               "--user-ids", ",".join(map(str, users[c::n_clients])),
               "--warmup-user-ids", ",".join(map(str, wusers[c::n_clients]))]
        clients.append(subprocess.Popen(cmd, env=env, cwd=ROOT, stdin=subprocess.PIPE,
                                        stdout=subprocess.PIPE, text=True, bufsize=1))
    return router, clients


def wait_line(p: subprocess.Popen, prefix: str) -> str:
    while True:
        line = p.stdout.readline()
        if not line:
            raise RuntimeError(f"load generator exited early (code {p.poll()})")
        if line.startswith(prefix):
            return line[len(prefix):].strip()


async def rank_main(a: argparse.Namespace) -> None:
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    n = max(a.gpus, world)
    router, clients = spawn_children(a, n) if rank == 0 else (None, [])
    procs = ([router] if router else []) + clients

    import torch
    import torch.distributed as dist

    from vllm_production_stack_amd.engine.async_engine import AsyncLLMEngine
    from vllm_production_stack_amd.engine.config import EngineConfig, KVOffloadConfig
    from vllm_production_stack_amd.engine.llm_engine import LLMEngine
    from vllm_production_stack_amd.engine.server import serve

    try:
        on_gpu = a.device == "cuda"
        gloo = None
        if on_gpu:
            torch.cuda.set_device(local_rank)
        if world > 1:
            if on_gpu:
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
                gloo = dist.new_group(backend="gloo")
                dist.barrier()  # brings RCCL up over xGMI before serving starts
            else:
                dist.init_process_group("gloo")
                gloo = dist.group.WORLD

        def barrier():
            if on_gpu:
                torch.cuda.synchronize()
            if gloo is not None:
                dist.barrier(group=gloo)
            if on_gpu:
                torch.cuda.synchronize()

        loop = asyncio.get_running_loop()
        off = KVOffloadConfig(local_cpu=a.cpu_offload_gb > 0, max_local_cpu_gb=a.cpu_offload_gb)
        cfg = EngineConfig(model=a.model, served_model_name=a.model, host="127.0.0.1",
                           device="auto" if on_gpu else "cpu",
                           dtype="auto" if on_gpu else "float32",
                           port=a.base_port + 1 + rank, max_model_len=a.max_model_len,
                           max_num_seqs=256, max_num_batched_tokens=a.max_num_batched_tokens,
                           enable_prefix_caching=True, enable_chunked_prefill=True, kv_offload=off,
                           num_gpu_blocks_override=a.num_gpu_blocks or None,
                           prefill_step_ms=a.prefill_step_ms)
        engine = LLMEngine(cfg)
        ae = AsyncLLMEngine(engine)
        ae.start(loop)
        runner = await serve(ae, cfg.host, cfg.port)
        await loop.run_in_executor(None, barrier)  # every replica is listening

        if rank == 0:
            import aiohttp

            async with aiohttp.ClientSession() as s:
                for _ in range(300):
                    try:
                        async with s.get(f"http://127.0.0.1:{a.base_port}/health") as r:
                            if r.status == 200:
                                break
                    except aiohttp.ClientError:
                        pass
                    await asyncio.sleep(0.1)
                else:
                    raise RuntimeError("router did not come up")
            for p in clients:
                p.stdin.write("WARMUP\n")
                p.stdin.flush()
            for p in clients:
                await loop.run_in_executor(None, wait_line, p, "READY")  # warm-up rounds done

        await loop.run_in_executor(None, barrier)
        t_start = time.time()
        outs = []
        if rank == 0:
            for p in clients:
                p.stdin.write("GO\n")
                p.stdin.flush()
            outs = [json.loads(await loop.run_in_executor(None, wait_line, p, "RESULT "))
                    for p in clients]
        await loop.run_in_executor(None, barrier)
        elapsed = time.time() - t_start
        # what every replica did in the timed region (shows how evenly the router spread the users)
        mine = {"rank": rank, "steps": int(engine.prof["steps"]), "busy_s": round(float(engine.prof["t_execute"]), 2),
                "rows": int(engine.prof["rows"])}
        per_replica = [mine]
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=gloo)
            elapsed = float(t.item())
            per_replica = [None] * world
            dist.all_gather_object(per_replica, mine, group=gloo)

        if rank == 0:
            sys.path.insert(0, os.path.join(ROOT, "benchmarks", "multi_round_qa"))
            import multi_round_qa as mrq

            rows = [r for o in outs for r in o["rows"]]
            bad = [r for r in rows if r["ttft"] < 0]
            out_tokens = sum(r["generation_tokens"] for r in rows)
            in_tokens = sum(r["prompt_tokens"] for r in rows)
            ttfts = [r["ttft"] for r in rows if r["ttft"] >= 0]
            warm = [r["ttft"] for r in rows if r["ttft"] >= 0 and r["question_id"] > 1]
            cold = [r["ttft"] for r in rows if r["ttft"] >= 0 and r["question_id"] == 1]
            users_total = a.users_per_gpu * n
            model_label = "Llama-3-8B" if a.model == "llama-3-8b" else a.model
            result = {
                "metric": f"output tok/s (+ p50 TTFT), multi-round QA, {model_label}, router + N engine replicas",
                "value": round(out_tokens / elapsed, 2), "unit": "tokens/s", "n_gpus": n,
                "steps": a.steps, "warmup": a.warmup,
                "ms_per_step": round(elapsed / a.steps * 1e3, 2), "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if on_gpu else "fp32",
                "data": f"synthetic multi-turn prompts; random-init {model_label} weights",
                "p50_ttft_s": round(mrq.pct(ttfts, 0.5), 4),
                "mean_ttft_s": round(sum(ttfts) / max(1, len(ttfts)), 4),
                "p90_ttft_s": round(mrq.pct(ttfts, 0.9), 4),
                "p50_ttft_warm_rounds_s": round(mrq.pct(warm, 0.5), 4) if warm else None,
                "p50_ttft_cold_round_s": round(mrq.pct(cold, 0.5), 4) if cold else None,
                "input_tok_per_s": round(in_tokens / elapsed, 1), "requests": len(rows),
                "failed_requests": len(bad), "elapsed_s": round(elapsed, 3),
                "round_times_s": [round(x, 3) for x in outs[0]["round_times"]] if outs else [],
                "config": {"model": model_label, "global_batch": users_total,
                           "seq_len": a.sys_len + a.history_len, "answer_len": a.answer_len,
                           "rounds": a.steps, "parallelism": f"dp{n} (replicas behind router)",
                           "routing": a.routing_logic,
                           "kv_pages_per_replica": engine.runner.num_pages,
                           "cpu_offload_gb": a.cpu_offload_gb,
                           "prefill_step_ms": a.prefill_step_ms},
                "kv_tiers": ({"host_loads_pages": engine.connector.host.loads,
                              "host_stores_pages": engine.connector.host.stores,
                              "host_evictions_pages": engine.connector.host.evictions}
                             if engine.connector is not None and engine.connector.host is not None else None),
                "engine0_profile": {k: (round(v, 3) if isinstance(v, float) else v)
                                    for k, v in engine.prof.items()},
                "per_replica": per_replica,
            }
            print(json.dumps(result), flush=True)
        await runner.cleanup()
        ae.stop()
        if world > 1:
            dist.destroy_process_group()
    finally:
        for p in procs:
            try:
                p.terminate()
            except Exception:
                pass
        for p in procs:
            try:
                p.wait(timeout=10)
            except Exception:
                p.kill()


def main() -> None:
    a = parse()
    if a.client:
        client_main(a)
        return
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # direct invocation with N > 1: relaunch under torchrun, one rank per GPU
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
               f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1", "--master-port",
               str(a.base_port + 50), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    asyncio.run(rank_main(a))


if __name__ == "__main__":
    main()
