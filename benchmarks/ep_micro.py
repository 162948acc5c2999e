"""One rank's expert-MLP work in a tensor-parallel Mixtral-8x7B layer (8 experts, top-2, hidden
4096, inter 14336), two ways of dividing it over N ranks:

  sliced   every rank holds a 1/N slice of every expert (all pairs, GEMMs of width inter / N)
  whole    ``--enable-expert-parallel``: every rank holds 8 / N whole experts and runs only the
           pairs routed to them (the others sort into a bucket no kernel visits)

Both end in the same all-reduce, which is not part of the timing.  For ``whole`` the time of the
slowest rank is reported as well (routing is uneven), since the all-reduce waits for it.

    python benchmarks/ep_micro.py [--ranks 2] [--tokens 8,40,1024,8192]
"""

import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_production_stack_amd import ops  # noqa: E402

E, K, H, I = 8, 2, 4096, 14336
DEV = torch.device("cuda", 0)


def timed(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=2)
    ap.add_argument("--tokens", default="8,40,1024,8192")
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    n = a.ranks
    torch.manual_seed(0)
    layers = 3  # visited round-robin so that nothing is served from cache
    per = E // n

    def rnd(*shape):
        return (torch.randn(*shape, device=DEV) * 0.02).to(torch.bfloat16)

    sliced = [(rnd(E, 2 * I // n, H), rnd(E, H, I // n)) for _ in range(layers)]
    whole = [(rnd(per, 2 * I, H), rnd(per, H, I)) for _ in range(layers)]
    for t in [int(v) for v in a.tokens.split(",")]:
        x = torch.randn(t, H, device=DEV).to(torch.bfloat16)
        w, ids = ops.moe_topk_softmax(torch.randn(t, E, device=DEV).to(torch.bfloat16), K)
        fn = ops.moe_experts_grouped if t <= ops.MOE_MFMA_MAX_TOKENS else ops.moe_experts_prefill
        state = {"i": 0}

        def nxt(ws):
            state["i"] = (state["i"] + 1) % layers
            return ws[state["i"]]

        def run_sliced():
            w13, w2 = nxt(sliced)
            fn(x, w13, w2, ids, w)

        row = {"tokens": t, "ranks": n, "sliced_ms": round(timed(run_sliced, a.iters), 3)}
        per_rank = []
        for r in range(n):
            lids, lw = ops.moe_localize(ids, w, r * per, per, bucket=True)

            def run_whole():
                w13, w2 = nxt(whole)
                fn(x, w13, w2, lids, lw, foreign=True)

            per_rank.append({"pairs": int((lids < per).sum()), "ms": round(timed(run_whole, a.iters), 3)})
        row["whole_per_rank"] = per_rank
        row["whole_slowest_ms"] = max(p["ms"] for p in per_rank)
        row["whole_over_sliced"] = round(row["whole_slowest_ms"] / row["sliced_ms"], 3)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
