"""Expert-MLP paths of one Mixtral-8x7B layer (8 experts, top-2, hidden 4096, inter 14336)
by batch size: per-pair GEMV kernels, expert-grouped matrix-core kernels, all-experts-on-all-
rows batched GEMM, and the per-expert library GEMM loop.  Prints time and the bandwidth at
which the routed experts' weights were streamed."""

import json
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_production_stack_amd import ops  # noqa: E402

E, K, H, I = 8, 2, 4096, 14336
DEV = torch.device("cuda", 0)


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def main():
    torch.manual_seed(0)
    # several layers' worth of weights, visited round-robin, so nothing is served from cache
    layers = 4
    w13 = [(torch.randn(E, 2 * I, H, device=DEV) * 0.02).to(torch.bfloat16) for _ in range(layers)]
    w2 = [(torch.randn(E, H, I, device=DEV) * 0.02).to(torch.bfloat16) for _ in range(layers)]
    expert_bytes = 3 * H * I * 2
    for t in [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1,4,8,16,32,64,128,256").split(",")]:
        x = torch.randn(t, H, device=DEV).to(torch.bfloat16)
        w, ids = ops.moe_topk_softmax(torch.randn(t, E, device=DEV).to(torch.bfloat16), K)
        n_routed = int(torch.unique(ids).numel())
        state = {"i": 0}

        def nxt():
            state["i"] = (state["i"] + 1) % layers
            return w13[state["i"]], w2[state["i"]]

        def grouped():
            a, b = nxt()
            ops.moe_experts_grouped(x, a, b, ids, w)

        def gemv():
            a, b = nxt()
            ops.moe_experts_small_batch(x, a, b, ids, w)

        def dense():
            a, b = nxt()
            d = torch.zeros((t, E), dtype=torch.float32, device=DEV)
            d.scatter_(1, ids.long(), w)
            gu = torch.matmul(x.unsqueeze(0), a.transpose(1, 2))
            act = ops.silu_and_mul(gu.view(E * t, -1)).view(E, t, -1)
            y = torch.matmul(act, b.transpose(1, 2))
            (y.float() * d.t().unsqueeze(-1)).sum(0).to(x.dtype)

        def loop():
            a, b = nxt()
            offsets, pos, sorted_tok = ops.moe_align(ids, E)
            xs = ops.moe_permute(x, sorted_tok)
            off = offsets.tolist()
            ys = torch.empty((t * K, H), dtype=x.dtype, device=DEV)
            for ex in range(E):
                lo, hi = off[ex], off[ex + 1]
                if hi > lo:
                    ys[lo:hi] = F.linear(ops.silu_and_mul(F.linear(xs[lo:hi], a[ex])), b[ex])
            ops.moe_combine(ys, w, pos, t)

        row = {"tokens": t, "routed_experts": n_routed}
        for name, fn in (("grouped_mfma", grouped), ("gemv", gemv), ("dense_all_experts", dense),
                         ("per_expert_loop", loop)):
            if name == "gemv" and t > 16:
                continue
            ms = timed(fn)
            row[name] = {"ms": round(ms, 3),
                         "routed_weight_tb_per_s": round(n_routed * expert_bytes / ms / 1e9, 2)}
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
