"""Decode step time of a tensor-parallel replica, eager against HIP-graph replay on every rank.

    python benchmarks/tp_graph_micro.py --model llama-3-8b --tp 2 --batch 8 --ctx 1024

On a box with fewer cards than ranks the ranks share cards (``VPSA_TP_BACKEND=gloo``: gloo for
control, one-shot all-reduce / all-gather kernels over hipIpc-mapped buffers inside the step), so
the numbers show what graph replay removes (host launch time of ~10 kernels per layer on every
rank, the pickled plan broadcast), not xGMI scaling.  One JSON line per mode."""

from __future__ import annotations

import argparse
import json
import os
import socket
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch
import torch.multiprocessing as mp


def _rank_main(rank, a, port, eager, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(a.tp), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    if torch.cuda.device_count() < a.tp:
        os.environ["VPSA_TP_BACKEND"] = "gloo"
    else:
        os.environ.setdefault("VPSA_ONESHOT_ALLREDUCE", "1")
    import torch.distributed as dist

    from vllm_production_stack_amd.engine.config import EngineConfig
    from vllm_production_stack_amd.engine.llm_engine import LLMEngine
    from vllm_production_stack_amd.engine.request import SamplingParams
    from vllm_production_stack_amd.parallel.launch import init_parallel_from_env
    from vllm_production_stack_amd.parallel.state import get_replica, get_tp

    init_parallel_from_env(a.tp, 1)
    cfg = EngineConfig(model=a.model, max_model_len=a.ctx + a.tokens + 64, max_num_seqs=a.batch,
                       max_num_batched_tokens=4096, num_gpu_blocks_override=a.batch * ((a.ctx + a.tokens) // 64 + 2),
                       port=0, tensor_parallel_size=a.tp, enforce_eager=eager, gpu_memory_utilization=0.4,
                       enable_expert_parallel=a.expert_parallel)
    eng = LLMEngine(cfg)
    if get_replica().is_driver:
        g = torch.Generator().manual_seed(1)
        prompts = [torch.randint(16, 30000, (a.ctx,), generator=g).tolist() for _ in range(a.batch)]
        sp = SamplingParams(max_tokens=a.tokens, temperature=0, ignore_eos=True)
        eng.generate(prompts[:1], SamplingParams(max_tokens=4, temperature=0, ignore_eos=True))
        p0 = dict(eng.prof)
        t0 = time.perf_counter()
        eng.generate(prompts, sp)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        d_steps = eng.prof["decode_only_steps"] - p0["decode_only_steps"]
        q.put({"mode": "eager" if eager else "graph", "model": a.model, "tp": a.tp, "batch": a.batch,
               "expert_parallel": bool(getattr(eng.runner.model, "ep", False)),
               "ctx": a.ctx, "new_tokens": a.tokens, "wall_s": round(wall, 3), "decode_only_steps": d_steps,
               "ranks_share_cards": torch.cuda.device_count() < a.tp,
               "oneshot_failed": bool(get_tp().oneshot.failed()) if get_tp().oneshot else None})
        eng.shutdown()
    else:
        eng.runner.worker_loop()
    dist.barrier()
    dist.destroy_process_group()


def _measure_decode(a, eager):
    """Decode step time = (wall of a long generation - wall of a short one) / extra steps."""
    ctx = mp.get_context("spawn")
    out = {}
    for tokens in (a.tokens, 8):
        b = argparse.Namespace(**vars(a))
        b.tokens = tokens
        q = ctx.Queue()
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        procs = [ctx.Process(target=_rank_main, args=(r, b, port, eager, q)) for r in range(a.tp)]
        for p in procs:
            p.start()
        try:
            out[tokens] = q.get(timeout=a.timeout)
            for p in procs:
                p.join(timeout=60)
        finally:
            for p in procs:
                if p.is_alive():
                    p.kill()
    long_, short = out[a.tokens], out[8]
    step_ms = (long_["wall_s"] - short["wall_s"]) / (a.tokens - 8) * 1e3
    long_["decode_step_ms"] = round(step_ms, 3)
    return long_


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="llama-3-8b")
    ap.add_argument("--tp", type=int, default=2)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--ctx", type=int, default=1024)
    ap.add_argument("--tokens", type=int, default=136)
    ap.add_argument("--timeout", type=int, default=240)
    ap.add_argument("--expert-parallel", action="store_true", help="whole experts per rank (mixture-of-experts models)")
    ap.add_argument("--modes", default="eager,graph")
    a = ap.parse_args()
    for mode in a.modes.split(","):
        print(json.dumps(_measure_decode(a, mode == "eager")), flush=True)


if __name__ == "__main__":
    main()
