"""The prefill GEMM kernel (``ops.tile_gemm``) against the library (hipBLASLt through torch) and
against the earlier grouped kernel (``ops.grouped_gemm``), on the four Llama-3-8B projections.

Correctness first (full-tensor comparison with an fp32 product, asymmetric operands, ragged row
counts, gather, grouped problem lists, the fused epilogues), then timing: the variants are timed
in interleaved rounds inside ONE process on the same random operands, several weight copies are
visited in turn so that a layer's weights are not in cache when it runs, and the median and the
minimum over the rounds are reported.

    python benchmarks/tile_gemm_micro.py [--m 8192] [--rounds 7] [--skip-check]
"""

import argparse
import json
import os
import statistics
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_production_stack_amd import ops  # noqa: E402

DEV = torch.device("cuda", 0)


def rnd(*shape, scale=1.0):
    return ((torch.rand(*shape, device=DEV) * 2 - 1) * scale).to(torch.bfloat16)


def rel_err(a, b):
    return float((a.float() - b.float()).abs().max() / b.float().abs().max().clamp_min(1e-9))


def check():
    out = []
    torch.manual_seed(0)
    # ragged M, N not a multiple of the tile, small K, asymmetric operands
    for m, n, k in ((1, 8, 64), (255, 264, 128), (256, 256, 64), (257, 512, 192), (700, 1032, 320), (4616, 4096, 4096)):
        x, w = rnd(m, k), rnd(n, k, scale=0.1)
        ref = x.float() @ w.float().t()
        got = ops.tile_gemm(x, w)
        out.append({"check": "plain", "m": m, "n": n, "k": k, "rel_err": rel_err(got, ref)})
        r = rnd(m, n)
        got = ops.tile_gemm(x, w, residual=r)
        out.append({"check": "residual", "m": m, "n": n, "k": k, "rel_err": rel_err(got, ref + r.float())})
        r2 = r.clone()
        ops.tile_gemm(x, w, residual=r2, out=r2)
        out.append({"check": "residual_in_place", "m": m, "n": n, "k": k, "rel_err": rel_err(r2, ref + r.float())})
    for m, inter, k in ((300, 96, 128), (513, 14336, 256), (2048, 1024, 4096)):
        x, w = rnd(m, k), rnd(2 * inter, k, scale=0.1)
        y = x.float() @ w.float().t()
        ref = F.silu(y[:, :inter]) * y[:, inter:]
        got = ops.tile_gemm(x, w, silu_mul=True)
        out.append({"check": "swiglu", "m": m, "inter": inter, "k": k, "rel_err": rel_err(got, ref)})
    # grouped with gather, empty and ragged groups
    g, n, k, t = 5, 520, 256, 1500
    x, w = rnd(t, k), rnd(g, n, k, scale=0.1)
    counts = [0, 300, 1, 899, 0]
    rows = sum(counts)
    off = torch.tensor([0] + list(torch.tensor(counts).cumsum(0)), dtype=torch.int32, device=DEV)
    idx = torch.randint(0, t, (rows + 7,), dtype=torch.int32, device=DEV)
    got = ops.tile_gemm(x, w, offsets=off, rows=rows + 7, arow_idx=idx)
    ref = torch.zeros(rows, n, device=DEV)
    lo = 0
    for gi, cnt in enumerate(counts):
        ref[lo:lo + cnt] = x[idx[lo:lo + cnt].long()].float() @ w[gi].float().t()
        lo += cnt
    out.append({"check": "grouped_gather", "rel_err": rel_err(got[:rows], ref)})
    # exact integer data: A = I-like selection with an asymmetric W catches a transposed or permuted store
    m = n = 512
    k = 512
    x = torch.eye(m, k, device=DEV).to(torch.bfloat16)
    w = (torch.arange(n, device=DEV).view(n, 1) * 3 + torch.arange(k, device=DEV).view(1, k) % 7).float()
    w = (w % 251).to(torch.bfloat16)
    got = ops.tile_gemm(x, w)
    out.append({"check": "identity_exact", "max_abs_err": float((got.float() - w.float().t()).abs().max())})
    for row in out:
        print(json.dumps(row), flush=True)
    bad = [r for r in out if r.get("rel_err", 0) > 8e-3 or r.get("max_abs_err", 0) > 0]
    if bad:
        print(json.dumps({"FAILED": bad}))
        sys.exit(1)


def bench(m, rounds, inner, dbg=False, gauss=False):
    for name, n, k in (("qkv", 6144, 4096), ("o", 4096, 4096), ("gate_up", 28672, 4096), ("down", 4096, 14336)):
        if gauss:  # what a random-init model feeds its projections: unit-variance rows, weights of std 0.02
            ws = [(torch.randn(n, k, device=DEV) * 0.02).to(torch.bfloat16) for _ in range(3)]
            x = torch.randn(m, k, device=DEV).to(torch.bfloat16)
        else:
            ws = [rnd(n, k, scale=0.05) for _ in range(3)]
            x = rnd(m, k)
        off = torch.tensor([0, m], dtype=torch.int32, device=DEV)
        st = {"i": 0}

        def nxt():
            st["i"] = (st["i"] + 1) % len(ws)
            return ws[st["i"]]

        variants = {
            "library": lambda: F.linear(x, nxt()),
            "tile": lambda: ops.tile_gemm(x, nxt()),
            "grouped_r2": lambda: ops.grouped_gemm(x, nxt().unsqueeze(0), off),
        }
        if dbg:
            lib = ops._kernels()

            def with_dbg(d):
                def run():
                    lib.vpsa_tile_gemm_debug(d)
                    r = ops.tile_gemm(x, nxt())
                    lib.vpsa_tile_gemm_debug(0)
                    return r
                return run

            variants["tile_no_loads(wrong)"] = with_dbg(1)
            variants["tile_no_frag_reads(wrong)"] = with_dbg(2)
            variants["tile_neither(wrong)"] = with_dbg(3)
            variants["tile_loads_hit_cache(wrong)"] = with_dbg(4)
        if name == "gate_up":
            variants["library+silu_mul"] = lambda: ops.silu_and_mul(F.linear(x, nxt()))
            variants["tile_swiglu"] = lambda: ops.tile_gemm(x, nxt(), silu_mul=True)
        if name in ("o", "down"):
            res = rnd(m, n)
            variants["tile_residual"] = lambda: ops.tile_gemm(x, nxt(), residual=res)
        times = {v: [] for v in variants}
        for v, fn in variants.items():
            for _ in range(2):
                fn()
        torch.cuda.synchronize()
        for _ in range(rounds):
            for v, fn in variants.items():
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(inner):
                    fn()
                torch.cuda.synchronize()
                times[v].append((time.perf_counter() - t0) / inner)
        flops = 2.0 * m * n * k
        row = {"case": name, "M": m, "N": n, "K": k}
        for v, ts in times.items():
            row[v] = {"median_us": round(statistics.median(ts) * 1e6, 1), "min_us": round(min(ts) * 1e6, 1),
                      "tflops_median": round(flops / statistics.median(ts) / 1e12, 1)}
        row["tile_vs_library"] = round(times["library"] and statistics.median(times["library"]) / statistics.median(times["tile"]), 3)
        print(json.dumps(row), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=8192)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--inner", type=int, default=10)
    ap.add_argument("--skip-check", action="store_true")
    ap.add_argument("--gauss", action="store_true", help="normal operands (std 1 activations, std 0.02 weights) instead of uniform")
    ap.add_argument("--dbg", action="store_true", help="add the timing experiments (wrong results) to the rounds")
    a = ap.parse_args()
    if not a.skip_check:
        check()
    bench(a.m, a.rounds, a.inner, a.dbg, a.gauss)


if __name__ == "__main__":
    main()
