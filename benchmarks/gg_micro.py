"""Prefill-sized GEMMs: the library (hipBLASLt through torch) against the hand-written grouped
MFMA kernel (``ops.grouped_gemm``), on the four Llama-3-8B projections at M = 8192 and on the
expert GEMMs of one Mixtral-8x7B layer (8 experts, top-2 of 8192 tokens: 16384 routed rows).

    python benchmarks/gg_micro.py [--cases dense,moe] [--no-mine]
"""

import argparse
import json
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_production_stack_amd import ops  # noqa: E402

DEV = torch.device("cuda", 0)


ITERS = 10


def timed(fn, iters=None, warm=3):
    iters = iters or ITERS
    for _ in range(min(warm, iters)):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def rnd(*shape):
    return (torch.rand(*shape, device=DEV) * 2 - 1).to(torch.bfloat16)


def rel_err(a, b):
    return float((a.float() - b.float()).abs().max() / b.float().abs().max())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", default="dense,moe")
    ap.add_argument("--no-mine", action="store_true")
    ap.add_argument("--m", type=int, default=8192)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--debug", type=int, default=0, help="timing experiments of the grouped kernel (wrong results)")
    a = ap.parse_args()
    global ITERS
    ITERS = a.iters
    mine = not a.no_mine and hasattr(ops, "grouped_gemm")
    if a.debug:
        ops._kernels().vpsa_grouped_gemm_debug(a.debug)
    if "dense" in a.cases:
        for name, n, k in (("qkv", 6144, 4096), ("o", 4096, 4096), ("gate_up", 28672, 4096), ("down", 4096, 14336)):
            m = a.m
            # several weight copies visited in turn: a layer's weights are not in cache when it runs
            ws = [rnd(n, k) * 0.05 for _ in range(3)]
            x = rnd(m, k)
            st = {"i": 0}

            def lib():
                st["i"] = (st["i"] + 1) % len(ws)
                return F.linear(x, ws[st["i"]])

            flops = 2.0 * m * n * k
            row = {"case": name, "M": m, "N": n, "K": k, "library_ms": round(timed(lib) * 1e3, 3)}
            row["library_tflops"] = round(flops / row["library_ms"] / 1e9, 1)
            if mine:
                off = torch.tensor([0, m], dtype=torch.int32, device=DEV)

                def own():
                    st["i"] = (st["i"] + 1) % len(ws)
                    return ops.grouped_gemm(x, ws[st["i"]].unsqueeze(0), off)

                row["own_ms"] = round(timed(own) * 1e3, 3)
                row["own_tflops"] = round(flops / row["own_ms"] / 1e9, 1)
                st["i"] = 0
                row["own_rel_err"] = round(rel_err(own(), F.linear(x.float(), ws[1].float())), 5)
                if name == "gate_up":
                    def own_silu():
                        st["i"] = (st["i"] + 1) % len(ws)
                        return ops.grouped_gemm(x, ws[st["i"]].unsqueeze(0), off, silu_mul=True)

                    def lib_silu():
                        return ops.silu_and_mul(lib())

                    row["library_plus_silu_mul_ms"] = round(timed(lib_silu) * 1e3, 3)
                    row["own_fused_silu_mul_ms"] = round(timed(own_silu) * 1e3, 3)
            print(json.dumps(row), flush=True)
    if "moe" in a.cases:
        e, topk, h, inter, t = 8, 2, 4096, 14336, a.m
        w13 = [rnd(e, 2 * inter, h) * 0.05 for _ in range(2)]
        w2 = [rnd(e, h, inter) * 0.05 for _ in range(2)]
        x = rnd(t, h)
        w, ids = ops.moe_topk_softmax(torch.randn(t, e, device=DEV).to(torch.bfloat16), topk)
        st = {"i": 0}

        def loop():
            st["i"] ^= 1
            a13, a2 = w13[st["i"]], w2[st["i"]]
            offsets, pos, sorted_tok = ops.moe_align(ids, e)
            xs = ops.moe_permute(x, sorted_tok)
            off = offsets.tolist()
            ys = torch.empty((t * topk, h), dtype=x.dtype, device=DEV)
            for ex in range(e):
                lo, hi = off[ex], off[ex + 1]
                if hi > lo:
                    ys[lo:hi] = F.linear(ops.silu_and_mul(F.linear(xs[lo:hi], a13[ex])), a2[ex])
            return ops.moe_combine(ys, w, pos, t)

        flops = 2.0 * t * topk * 3 * h * inter
        row = {"case": "mixtral_layer_experts", "tokens": t, "routed_rows": t * topk,
               "per_expert_library_loop_ms": round(timed(loop) * 1e3, 3)}
        row["per_expert_library_loop_tflops"] = round(flops / row["per_expert_library_loop_ms"] / 1e9, 1)
        if mine:
            def own():
                st["i"] ^= 1
                return ops.moe_experts_prefill(x, w13[st["i"]], w2[st["i"]], ids, w)

            def own_permuted():
                st["i"] ^= 1
                return ops.moe_experts_prefill(x, w13[st["i"]], w2[st["i"]], ids, w, gather=False)

            def only_gate_up():
                st["i"] ^= 1
                return ops.grouped_gemm(x, w13[st["i"]], offsets0, rows=t * topk, arow_idx=sorted0, silu_mul=True)

            def only_down():
                st["i"] ^= 1
                return ops.grouped_gemm(act0, w2[st["i"]], offsets0, rows=t * topk)

            offsets0, pos0, sorted0 = ops.moe_align(ids, e)
            act0 = only_gate_up()
            row["grouped_permute_first_ms"] = round(timed(own_permuted) * 1e3, 3)
            row["grouped_gate_up_only_ms"] = round(timed(only_gate_up) * 1e3, 3)
            row["grouped_down_only_ms"] = round(timed(only_down) * 1e3, 3)
            row["grouped_ms"] = round(timed(own) * 1e3, 3)
            row["grouped_tflops"] = round(flops / row["grouped_ms"] / 1e9, 1)
            st["i"] = 0
            got = own()
            st["i"] = 0
            row["grouped_rel_err_vs_loop"] = round(rel_err(got, loop()), 5)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
