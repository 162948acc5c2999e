"""Decode-batch library GEMMs (tens of rows against every weight matrix): does another way of
asking the library for the same product stream the weights faster?  Timed inside HIP graphs (as the
engine runs them), weights rotated so they come from HBM.

Variants: ``linear`` (x @ W.T), ``swapped`` ((W @ x.T).T: the library sees the weight matrix as the
tall operand), ``pad64`` (rows padded to 64), ``split2`` (two column halves into one buffer).

    python benchmarks/decode_gemm_variants.py [--rows 40,48]
"""

import argparse
import json

import torch
import torch.nn.functional as F

SHAPES = [("qkv", 6144, 4096), ("o", 4096, 4096), ("gate_up", 28672, 4096), ("down", 4096, 14336)]


def graph_time(fn, n_weights, reps=8):
    for i in range(n_weights):
        fn(i)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            for i in range(n_weights):
                fn(i)
    g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5):
        g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / (5 * reps * n_weights)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", default="40")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    nw = 6
    for m in [int(r) for r in a.rows.split(",")]:
        for name, n, k in SHAPES:
            ws = [(torch.randn(n, k, device=dev) * 0.02).to(torch.bfloat16) for _ in range(nw)]
            x = torch.randn(m, k, device=dev).to(torch.bfloat16)
            xp = torch.zeros(64, k, device=dev, dtype=torch.bfloat16)
            xp[:m] = x
            out = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
            half = n // 2
            rec = {"case": name, "rows": m, "n": n, "k": k, "weight_MB": round(n * k * 2 / 1e6, 1)}
            variants = {
                "linear": lambda i: F.linear(x, ws[i]),
                "swapped": lambda i: torch.matmul(ws[i], x.t()),
                "pad64": lambda i: F.linear(xp, ws[i]),
                "split2": lambda i: (torch.matmul(x, ws[i][:half].t(), out=out[:, :half]),
                                     torch.matmul(x, ws[i][half:].t(), out=out[:, half:])),
            }
            for vname, fn in variants.items():
                us = graph_time(fn, nw)
                rec[vname + "_us"] = round(us, 1)
                rec[vname + "_TB_per_s"] = round(n * k * 2 / us / 1e6, 2)
            print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
