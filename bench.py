"""Headline benchmark: multi-round QA on Llama-3-8B through the full serving stack.

    python bench.py --gpus N --steps K --warmup W

One engine replica per GPU (one process per GPU; for N > 1 the driver launches this file
under ``torch.distributed.run`` and every rank hosts the replica of its GPU).  Rank 0 also
starts the router (prefix-aware routing over the N replicas, static discovery, several router
worker processes from 4 replicas on) and the load-generator processes, which talk to the router
over HTTP exactly like a client of a deployed stack.

Every rank pins its process to its GPU with ``HIP_VISIBLE_DEVICES`` before anything touches the
device, which is how the stack's own launcher (``vpsa-stack up``) and the chart place replicas:
inside a replica the GPU is always device 0, whatever thread uses it.  Data-parallel replicas
exchange nothing over the fabric, so the ranks synchronise (barriers, the max over ranks of the
timed region) over gloo; RCCL comes in with tensor-parallel replicas and KV transfers, not here.

Workload (reference: /root/reference/benchmarks/multi-round-qa/run.sh:14-18,43-48 and
tutorials/08-benchmark-multi-round-qa-multi-gpu.md:36-72): users-per-GPU x N concurrent
users (default 40 per GPU = 320 on 8 GPUs), shared system prompt 1000 tokens, per-user
history 20000 tokens, 100-token answers, streamed chat completions with temperature 0.
A *step* is one conversation round: every user sends its next turn and the round ends when
every answer has been streamed back.  Warm-up rounds use separate throw-away users; the
timed region starts with fresh users, so the first timed conversation round is the cold 21k-token
prefill of every user and later rounds hit the KV cache.  Weak scaling: per-GPU users fixed.

Prints one JSON line: value = aggregate output tokens/s over all GPUs (the throughput half
of the headline metric); the TTFT half (p50/mean/p90, overall and warm-round) rides along
in the same line.  Weights are random-init Llama-3-8B (bf16), prompts synthetic.

After the timed headline region a second, short phase (``--reference-rounds``, default 3; not part
of ``value``) runs the configuration the reference's own procedure prescribes
(/root/reference/tutorials/07-benchmark-multi-round-qa-single-gpu.md:31-67): engine prefix
caching off, chunked prefill off, KV offload to host DRAM on, ``session`` routing on
``x-user-id``, 15 users per GPU.  Every follow-up turn then takes its 21k-token history from the
pinned host pool over the side-stream copy engine; its tok/s, TTFT and tier counters are
reported under ``reference_config`` in the same line.
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
    # (below 32768, where the kernel starts handing out ports for outgoing connections)
    return 18100 + (mp % 350) * 40 if mp else 18100


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
    ap.add_argument("--router-workers", type=int, default=0, help="router processes (0 = 1 up to 3 replicas, "
                                                                  "then one per two replicas, at most 4)")
    ap.add_argument("--cpu-offload-gb", type=float, default=0.0)
    ap.add_argument("--num-gpu-blocks", type=int, default=0,
                    help="cap the HBM KV pool (pages of 64 tokens) to exercise the host tier; 0 = size from HBM")
    ap.add_argument("--prefill-step-ms", type=float, default=30.0,
                    help="engine's latency-aware prefill admission target (0 = token budget only)")
    ap.add_argument("--num-scheduler-steps", type=int, default=1,
                    help="decode steps per launch of the engines (see the engine's flag of the same name)")
    ap.add_argument("--kv-layer-groups", type=int, default=0,
                    help="KV pool of the headline phase's engines as this many pools of consecutive layers "
                         "(0 = the engine's choice; see the engine's flag of the same name)")
    ap.add_argument("--packed-decode-weights", default="", choices=["", "auto", "on", "off"],
                    help="second copy of the MLP weights in matrix-core tile order for decode steps "
                         "(empty = the engine's default)")
    ap.add_argument("--reference-rounds", type=int, default=3,
                    help="rounds of the second phase in the reference's own configuration (0 = skip)")
    ap.add_argument("--reference-users-per-gpu", type=int, default=15)
    ap.add_argument("--reference-offload-gb", type=float, default=45.0)
    ap.add_argument("--open-loop-seconds", type=int, default=100,
                    help="after the rounds of the reference-configuration phase: the reference's own "
                         "open-loop procedure (QPS-paced arrivals, users joining over time) for this long; 0 = skip")
    ap.add_argument("--open-loop-qps-per-gpu", type=float, default=1.1,
                    help="offered request rate per replica (1.1 is the top of the reference's single-GPU sweep)")
    ap.add_argument("--base-port", type=int, default=_default_base_port())
    ap.add_argument("--clients", type=int, default=0, help="load-generator processes (0 = auto)")
    ap.add_argument("--layout", default="",
                    help="whole-node layouts instead of N equal replicas (benchmarks/layouts.py): 'pd:2x6' = 2 prefill "
                         "+ 6 decode replicas behind the router in disaggregated_prefill mode; 'multimodel' = six "
                         "replicas of --model + a TP2 Mixtral-8x7B replica, kvaware routing, shared KV cache server. "
                         "VPSA_BENCH_SINGLE_DEVICE=1 (+ --num-gpu-blocks) rehearses either on one card")
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
def pin_gpu(a: argparse.Namespace) -> None:
    """One process, one GPU: restrict this rank to its device before any GPU call.  With
    ``VPSA_BENCH_SINGLE_DEVICE=1`` every rank takes the first device (a rehearsal of the N-rank
    launch path on a one-GPU machine; the KV pools must then be capped with --num-gpu-blocks)."""
    if a.device != "cuda" or os.environ.get("VPSA_BENCH_NO_PIN") == "1":
        return
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("VPSA_BENCH_SINGLE_DEVICE") == "1":
        local_rank = 0
    visible = [x for x in os.environ.get("HIP_VISIBLE_DEVICES", "").split(",") if x.strip()]
    dev = visible[local_rank] if local_rank < len(visible) else str(local_rank)
    os.environ["HIP_VISIBLE_DEVICES"] = dev
    os.environ["VPSA_BENCH_PINNED_GPU"] = dev


def spawn_children(a: argparse.Namespace, n: int, port0: int, routing: str, users_per_gpu: int,
                   warm_users_per_gpu: int, rounds: int, warm_rounds: int, user_base: int):
    """Router + load generators of one phase.  Never touches the GPU."""
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "HIP_VISIBLE_DEVICES"):
        env.pop(k, None)
    backends = ",".join(f"http://127.0.0.1:{port0 + 1 + r}" for r in range(n))
    workers = a.router_workers or (1 if n < 4 else min(4, n // 2))
    router_cmd = [sys.executable, "-m", "vllm_production_stack_amd.router.app",
                  "--host", "127.0.0.1", "--port", str(port0),
                  "--service-discovery", "static", "--static-backends", backends,
                  "--static-models", ",".join([a.model] * n),
                  "--routing-logic", routing, "--session-key", "x-user-id",
                  "--router-workers", str(workers),
                  "--engine-stats-interval", "5", "--log-level", "warning"]
    router = subprocess.Popen(router_cmd, env=env, cwd=ROOT)
    n_clients = a.clients or max(1, min(4, (n + 1) // 2))
    users = list(range(user_base, user_base + users_per_gpu * n))
    wusers = list(range(user_base + 10_000_000, user_base + 10_000_000 + warm_users_per_gpu * n))
    clients = []
    for c in range(n_clients):
        cmd = [sys.executable, os.path.abspath(__file__), "--client",
               "--base-url", f"http://127.0.0.1:{port0}", "--model", a.model,
               "--steps", str(rounds), "--warmup", str(warm_rounds),
               "--sys-len", str(a.sys_len), "--history-len", str(a.history_len),
               "--answer-len", str(a.answer_len),
               "--user-ids", ",".join(map(str, users[c::n_clients])),
               "--warmup-user-ids", ",".join(map(str, wusers[c::n_clients]))]
        clients.append(subprocess.Popen(cmd, env=env, cwd=ROOT, stdin=subprocess.PIPE,
                                        stdout=subprocess.PIPE, text=True, bufsize=1))
    return router, clients, workers


def run_open_loop_client(a: argparse.Namespace, port0: int, n: int, spec: dict) -> dict:
    """One point of the reference's QPS sweep (benchmarks/multi_round_qa/multi_round_qa.py is the
    same tool the sweep script drives), as a child process; returns its summary."""
    import csv
    import io
    import tempfile

    sys.path.insert(0, os.path.join(ROOT, "benchmarks", "multi_round_qa"))
    import multi_round_qa as mrq

    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "HIP_VISIBLE_DEVICES"):
        env.pop(k, None)
    out = os.path.join(tempfile.mkdtemp(prefix="vpsa-bench-"), "open_loop.csv")
    qps = spec["qps_per_gpu"] * n
    cmd = [sys.executable, os.path.join(ROOT, "benchmarks", "multi_round_qa", "multi_round_qa.py"),
           "--num-users", str(spec["users_per_gpu"] * n), "--num-rounds", str(spec["rounds"]),
           "--qps", str(qps), "--shared-system-prompt", str(a.sys_len),
           "--user-history-prompt", str(a.history_len), "--answer-len", str(a.answer_len),
           "--model", a.model, "--base-url", f"http://127.0.0.1:{port0}", "--time", str(spec["seconds"]),
           "--init-user-id", str(spec["user_base"]), "--request-with-user-id", "--output", out,
           "--log-interval", "1000"]
    t0 = time.time()
    proc = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True,
                          timeout=spec["seconds"] + 300)
    if proc.returncode != 0:
        return {"error": proc.stderr[-500:]}
    with open(out) as f:
        rows = [{k: float(v) for k, v in r.items()} for r in csv.DictReader(f)]
    s = mrq.summarize(rows, out=io.StringIO())
    if not s:
        return {"error": "no finished requests"}
    return {"what": "open loop, the reference's procedure: requests paced at a fixed rate, sessions pre-aged, "
                    "new users joining; mean TTFT at this rate is the reference's headline metric",
            "offered_qps": qps, "users": spec["users_per_gpu"] * n, "seconds": spec["seconds"],
            "wall_s": round(time.time() - t0, 1), "requests": s["requests"],
            "finished_qps": round(s["finished_qps"], 3), "mean_ttft_s": round(s["mean_ttft_s"], 4),
            "p50_ttft_s": round(s["p50_ttft_s"], 4), "p90_ttft_s": round(s["p90_ttft_s"], 4),
            "output_tok_per_s": round(s["output_tokens_per_s"], 1),
            "generation_tok_per_s_per_request": round(s["avg_generation_speed_per_request"], 1)}


def wait_line(p: subprocess.Popen, prefix: str) -> str:
    while True:
        line = p.stdout.readline()
        if not line:
            raise RuntimeError(f"load generator exited early (code {p.poll()})")
        if line.startswith(prefix):
            return line[len(prefix):].strip()


def stop_children(procs) -> None:
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


async def run_phase(a, ctx, *, port0: int, routing: str, users_per_gpu: int, warm_users_per_gpu: int,
                    rounds: int, warm_rounds: int, user_base: int, engine_kwargs: dict,
                    open_loop: dict | None = None) -> dict | None:
    """One measured phase: engine up, children up, warm-up, the timed rounds, everything down again
    (router and load generators first, so nothing scrapes an engine that is already gone).
    Returns the result dictionary on rank 0."""
    import torch
    import torch.distributed as dist

    from vllm_production_stack_amd.engine.async_engine import AsyncLLMEngine
    from vllm_production_stack_amd.engine.config import EngineConfig
    from vllm_production_stack_amd.engine.llm_engine import LLMEngine
    from vllm_production_stack_amd.engine.server import serve

    rank, world, n, on_gpu, gloo = ctx["rank"], ctx["world"], ctx["n"], ctx["on_gpu"], ctx["gloo"]
    router, clients, workers, procs = None, [], 0, []
    engine = ae = runner = None

    def barrier():
        if on_gpu:
            torch.cuda.synchronize()
        if gloo is not None:
            dist.barrier(group=gloo)
        if on_gpu:
            torch.cuda.synchronize()

    loop = asyncio.get_running_loop()
    try:
        # Bringing a replica up can fail on one rank only (host memory for the pinned tier, a port
        # in use): the outcome is agreed on by all ranks before anyone waits at a barrier, so a
        # failure ends the phase everywhere instead of hanging the ranks that succeeded.
        start_error = None
        try:
            if os.environ.get("VPSA_BENCH_FAIL_START_RANK") == str(rank):  # test hook
                raise MemoryError("injected start failure")
            cfg = EngineConfig(model=a.model, served_model_name=a.model, host="127.0.0.1",
                               device="auto" if on_gpu else "cpu",
                               dtype="auto" if on_gpu else "float32",
                               port=port0 + 1 + rank, max_model_len=a.max_model_len, max_num_seqs=256,
                               num_gpu_blocks_override=a.num_gpu_blocks or None, **engine_kwargs)
            engine = LLMEngine(cfg)
            ae = AsyncLLMEngine(engine)
            ae.start(loop)
            runner = await serve(ae, cfg.host, cfg.port)
        except Exception as e:  # noqa: BLE001
            start_error = e
        failed_ranks = [rank] if start_error is not None else []
        if world > 1:
            gathered = [None] * world
            await loop.run_in_executor(None, lambda: dist.all_gather_object(
                gathered, repr(start_error) if start_error is not None else None, group=gloo))
            failed_ranks = [r for r, g in enumerate(gathered) if g is not None]
            if failed_ranks and start_error is None:
                start_error = RuntimeError(f"rank {failed_ranks[0]}: {gathered[failed_ranks[0]]}")
        if failed_ranks:
            raise RuntimeError(f"engine start failed on rank(s) {failed_ranks}: {start_error!r}")
        await loop.run_in_executor(None, barrier)  # every replica is listening

        # Whatever goes wrong with the router or the load generators (rank 0 only) is kept until
        # every barrier of the phase has been passed: the other ranks wait at those barriers, and a
        # rank 0 that left early would leave them there until the process group's timeout.
        client_error = None
        if rank == 0:
            try:
                import aiohttp

                # only now: a router that starts before its backends logs a failed scrape for each
                router, clients, workers = spawn_children(a, n, port0, routing, users_per_gpu, warm_users_per_gpu,
                                                          rounds, warm_rounds, user_base)
                procs.extend([router] + clients)

                async with aiohttp.ClientSession() as s:
                    for _ in range(300):
                        try:
                            async with s.get(f"http://127.0.0.1:{port0}/health") as r:
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
            except Exception as e:  # noqa: BLE001
                client_error = e

        await loop.run_in_executor(None, barrier)
        prof0 = dict(engine.prof)  # after the barrier: the other ranks' replicas are still warming up before it
        t_start = time.time()
        outs = []
        if rank == 0 and client_error is None:
            try:
                if os.environ.get("VPSA_BENCH_FAIL_CLIENTS") == str(user_base):  # test hook
                    raise RuntimeError("injected load generator failure")
                for p in clients:
                    p.stdin.write("GO\n")
                    p.stdin.flush()
                outs = [json.loads(await loop.run_in_executor(None, wait_line, p, "RESULT "))
                        for p in clients]
            except Exception as e:  # noqa: BLE001
                client_error = e
        await loop.run_in_executor(None, barrier)
        elapsed = time.time() - t_start
        # what every replica did in the timed region (shows how evenly the router spread the users)
        mine = {"rank": rank, "steps": int(engine.prof["steps"] - prof0["steps"]),
                "busy_s": round(float(engine.prof["t_execute"] - prof0["t_execute"]), 2),
                "rows": int(engine.prof["rows"] - prof0["rows"]), "gpu": os.environ.get("VPSA_BENCH_PINNED_GPU")}
        conn = engine.connector
        if conn is not None and conn.host is not None:
            mine["host_loads_pages"] = conn.host.loads
            mine["host_stores_pages"] = conn.host.stores
        per_replica = [mine]
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=gloo)
            elapsed = float(t.item())
            per_replica = [None] * world
            dist.all_gather_object(per_replica, mine, group=gloo)
        kv_tiers = ({"host_loads_pages": conn.host.loads, "host_stores_pages": conn.host.stores,
                     "host_evictions_pages": conn.host.evictions, "host_slots": conn.host.n_slots}
                    if conn is not None and conn.host is not None else None)
        prof_delta = {k: (round(v - prof0.get(k, 0), 3) if isinstance(v, float) else v - prof0.get(k, 0))
                      for k, v in engine.prof.items()}
        open_loop_result = None
        if open_loop:
            # the reference's procedure against the same stack: requests arrive at a fixed rate,
            # sessions are pre-aged and new users keep joining (multi-round-qa.py's main loop)
            prof1 = dict(engine.prof)
            if rank == 0 and client_error is None:
                open_loop_result = await loop.run_in_executor(None, run_open_loop_client, a, port0, n, open_loop)
                if "error" not in open_loop_result:
                    # what the engine did during this phase alone: how its steps ran (graphs or eager
                    # launches) and how much host time went into issuing them
                    open_loop_result["engine0_profile"] = {
                        k: (round(v - prof1.get(k, 0), 3) if isinstance(v, float) else v - prof1.get(k, 0))
                        for k, v in engine.prof.items()}
            await loop.run_in_executor(None, barrier)
        if rank != 0:
            return None
        if client_error is not None:
            raise client_error

        sys.path.insert(0, os.path.join(ROOT, "benchmarks", "multi_round_qa"))
        import multi_round_qa as mrq

        rows = [r for o in outs for r in o["rows"]]
        bad = [r for r in rows if r["ttft"] < 0]
        ttfts = [r["ttft"] for r in rows if r["ttft"] >= 0]
        warm = [r["ttft"] for r in rows if r["ttft"] >= 0 and r["question_id"] > 1]
        cold = [r["ttft"] for r in rows if r["ttft"] >= 0 and r["question_id"] == 1]
        out_tokens = sum(r["generation_tokens"] for r in rows)
        in_tokens = sum(r["prompt_tokens"] for r in rows)
        return {
            "elapsed_s": round(elapsed, 3), "out_tokens": out_tokens, "in_tokens": in_tokens,
            "tok_per_s": round(out_tokens / elapsed, 2), "requests": len(rows), "failed_requests": len(bad),
            "p50_ttft_s": round(mrq.pct(ttfts, 0.5), 4) if ttfts else None,
            "mean_ttft_s": round(sum(ttfts) / max(1, len(ttfts)), 4),
            "p90_ttft_s": round(mrq.pct(ttfts, 0.9), 4) if ttfts else None,
            "p50_ttft_warm_rounds_s": round(mrq.pct(warm, 0.5), 4) if warm else None,
            "p50_ttft_cold_round_s": round(mrq.pct(cold, 0.5), 4) if cold else None,
            "round_times_s": [round(x, 3) for x in outs[0]["round_times"]] if outs else [],
            "per_replica": per_replica, "router_workers": workers, "open_loop": open_loop_result,
            "kv_pages_per_replica": engine.runner.num_pages,
            "kv_layer_groups": len(getattr(engine.runner.pool, "groups", [0])),
            "kv_tiers": kv_tiers, "engine0_profile": prof_delta,
        }
    finally:
        # order matters for a clean log: the router (rank 0) goes first, every rank waits for that,
        # and only then do the engines stop answering its metric scrapes
        stop_children(procs)
        if gloo is not None:
            try:
                import datetime

                dist.monitored_barrier(group=gloo, timeout=datetime.timedelta(seconds=30))
            except Exception:
                pass
        if runner is not None:
            await runner.cleanup()
        if ae is not None:
            ae.stop()


def reference_offload_gb(a: argparse.Namespace, world: int) -> float:
    """Pinned host memory per replica for the reference-configuration phase: what was asked for,
    but never more than this machine can pin for all replicas at once (a third of what is
    available, split evenly), and at least what one round of its users needs."""
    want = float(a.reference_offload_gb)
    try:
        import psutil

        avail_gb = psutil.virtual_memory().available / (1 << 30)
        want = min(want, max(1.0, avail_gb / 3 / max(1, world)))
    except Exception:  # noqa: BLE001
        pass
    return round(want, 2)


def release_gpu_memory() -> None:
    """Between phases: the previous engine's weights, KV pool and graphs are unreachable once
    ``run_phase`` has returned; collect them and hand the blocks back to the driver."""
    import gc

    import torch

    gc.collect()
    if torch.cuda.is_available():
        torch.cuda.synchronize()
        torch.cuda.empty_cache()


async def rank_main(a: argparse.Namespace) -> None:
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    n = max(a.gpus, world)
    pin_gpu(a)
    # stdout carries the one JSON line and nothing else: libraries that chat on file descriptor 1
    # (gloo announces its connections there) are sent to stderr
    sys.stdout.flush()
    result_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    from vllm_production_stack_amd.engine.config import KVOffloadConfig

    on_gpu = a.device == "cuda"
    gloo = None
    if on_gpu:
        torch.cuda.set_device(0 if os.environ.get("VPSA_BENCH_PINNED_GPU") is not None
                              else int(os.environ.get("LOCAL_RANK", "0")))
    if world > 1:
        dist.init_process_group("gloo")
        gloo = dist.group.WORLD
    ctx = {"rank": rank, "world": world, "n": n, "on_gpu": on_gpu, "gloo": gloo}

    async def heartbeat():
        # a run of several minutes that prints nothing looks hung to whatever is watching it
        t0 = time.time()
        while True:
            await asyncio.sleep(30)
            print(f"[bench] running, {time.time() - t0:.0f} s", file=sys.stderr, flush=True)

    beat = asyncio.create_task(heartbeat()) if rank == 0 else None
    try:
        off = KVOffloadConfig(local_cpu=a.cpu_offload_gb > 0, max_local_cpu_gb=a.cpu_offload_gb)
        head = await run_phase(
            a, ctx, port0=a.base_port, routing=a.routing_logic, users_per_gpu=a.users_per_gpu,
            warm_users_per_gpu=a.warmup_users_per_gpu, rounds=a.steps, warm_rounds=a.warmup, user_base=0,
            engine_kwargs=dict(max_num_batched_tokens=a.max_num_batched_tokens, enable_prefix_caching=True,
                               enable_chunked_prefill=True, kv_offload=off, prefill_step_ms=a.prefill_step_ms,
                               num_scheduler_steps=a.num_scheduler_steps,
                               kv_layer_groups=a.kv_layer_groups,
                               **({"packed_decode_weights": a.packed_decode_weights}
                                  if a.packed_decode_weights else {})))
        ref = None
        if a.reference_rounds > 0:
            release_gpu_memory()
            if gloo is not None:
                # every rank has let go of the first phase's memory before any rank allocates for the
                # second one (matters when a rehearsal puts several ranks on one card)
                await asyncio.get_running_loop().run_in_executor(None, lambda: dist.barrier(group=gloo))
            offload_gb = reference_offload_gb(a, world)
            try:
                ref = await run_phase(
                    a, ctx, port0=a.base_port + 20, routing="session", users_per_gpu=a.reference_users_per_gpu,
                    warm_users_per_gpu=1, rounds=a.reference_rounds, warm_rounds=1, user_base=20_000_000,
                    engine_kwargs=dict(max_num_batched_tokens=max(a.max_num_batched_tokens, a.max_model_len),
                                       enable_prefix_caching=False, enable_chunked_prefill=False,
                                       prefill_step_ms=0.0,
                                       kv_offload=KVOffloadConfig(local_cpu=True,
                                                                  max_local_cpu_gb=offload_gb)),
                    open_loop=(dict(qps_per_gpu=a.open_loop_qps_per_gpu, seconds=a.open_loop_seconds,
                                    users_per_gpu=a.reference_users_per_gpu, rounds=20, user_base=30_000_000)
                               if a.open_loop_seconds > 0 else None))
            except Exception as e:  # the headline result stands on its own
                ref = {"error": repr(e)} if rank == 0 else None
        if rank == 0:
            users_total = a.users_per_gpu * n
            model_label = "Llama-3-8B" if a.model == "llama-3-8b" else a.model
            result = {
                "metric": f"output tok/s (+ p50 TTFT), multi-round QA, {model_label}, router + N engine replicas",
                "value": head["tok_per_s"], "unit": "tokens/s", "n_gpus": n,
                "steps": a.steps, "warmup": a.warmup,
                "ms_per_step": round(head["elapsed_s"] / a.steps * 1e3, 2), "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if on_gpu else "fp32",
                "data": f"synthetic multi-turn prompts; random-init {model_label} weights",
                "p50_ttft_s": head["p50_ttft_s"], "mean_ttft_s": head["mean_ttft_s"],
                "p90_ttft_s": head["p90_ttft_s"],
                "p50_ttft_warm_rounds_s": head["p50_ttft_warm_rounds_s"],
                "p50_ttft_cold_round_s": head["p50_ttft_cold_round_s"],
                "input_tok_per_s": round(head["in_tokens"] / head["elapsed_s"], 1),
                "requests": head["requests"], "failed_requests": head["failed_requests"],
                "elapsed_s": head["elapsed_s"], "round_times_s": head["round_times_s"],
                "config": {"model": model_label, "global_batch": users_total,
                           "seq_len": a.sys_len + a.history_len, "answer_len": a.answer_len,
                           "rounds": a.steps, "parallelism": f"dp{n} (replicas behind router)",
                           "routing": a.routing_logic, "router_workers": head["router_workers"],
                           "kv_pages_per_replica": head["kv_pages_per_replica"],
                           "kv_layer_groups": head.get("kv_layer_groups", 1),
                           "cpu_offload_gb": a.cpu_offload_gb, "prefill_step_ms": a.prefill_step_ms},
                "kv_tiers": head["kv_tiers"], "engine0_profile": head["engine0_profile"],
                "per_replica": head["per_replica"],
            }
            if ref is not None:
                if "error" in ref:
                    result["reference_config"] = ref
                else:
                    result["reference_config"] = {
                        "what": "second phase, not part of value: engine prefix caching off, chunked prefill "
                                "off, KV offload to pinned host DRAM on, session routing on x-user-id "
                                "(the reference's own benchmark setup)",
                        "users": a.reference_users_per_gpu * n, "rounds": a.reference_rounds,
                        "cpu_offload_gb_per_replica": offload_gb,
                        "output_tok_per_s": ref["tok_per_s"], "p50_ttft_s": ref["p50_ttft_s"],
                        "p50_ttft_warm_rounds_s": ref["p50_ttft_warm_rounds_s"],
                        "p50_ttft_cold_round_s": ref["p50_ttft_cold_round_s"],
                        "round_times_s": ref["round_times_s"], "requests": ref["requests"],
                        "failed_requests": ref["failed_requests"], "open_loop": ref["open_loop"],
                        "kv_tiers": ref["kv_tiers"],
                        "per_replica": ref["per_replica"]}
            print(json.dumps(result), file=result_out, flush=True)
    finally:
        if beat is not None:
            beat.cancel()
        if world > 1:
            dist.destroy_process_group()


def main() -> None:
    a = parse()
    if a.client:
        client_main(a)
        return
    if a.layout:
        sys.path.insert(0, os.path.join(ROOT, "benchmarks"))
        import layouts

        layouts.main(a)
        return
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # direct invocation with N > 1: relaunch under torchrun, one rank per GPU
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
               f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1", "--master-port",
               str(a.base_port + 39), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    asyncio.run(rank_main(a))


if __name__ == "__main__":
    main()
