"""Is one wide library GEMM slower than the same product issued as two or three narrower ones?

Times ``x @ W.T`` for the fused q/k/v and gate/up matrices of Llama-3-8B at prefill size, whole and
split along the output columns (each part written into its column range of ONE output buffer, so
nothing downstream changes).  The library's heuristic picks a slow solution for some widths.

    python benchmarks/qkv_split_probe.py [--rows 8192]
"""

import argparse
import json
import time

import torch
import torch.nn.functional as F


def timed(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", default="8192")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    for m in [int(r) for r in a.rows.split(",")]:
        for name, n, k, cuts in (("qkv", 6144, 4096, [[4096], [2048, 4096], [3072]]),
                                 ("gate_up", 28672, 4096, [[14336], [7168, 14336, 21504]]),
                                 ("o", 4096, 4096, [[2048]]), ("down", 4096, 14336, [[2048]])):
            ws = [(torch.randn(n, k, device=dev) * 0.02).to(torch.bfloat16) for _ in range(4)]
            x = torch.randn(m, k, device=dev).to(torch.bfloat16)
            out = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
            i = [0]

            def whole():
                i[0] += 1
                return F.linear(x, ws[i[0] % 4])

            rec = {"case": name, "rows": m, "n": n, "k": k, "whole_us": round(timed(whole), 1)}
            for cut in cuts:
                edges = [0] + cut + [n]

                def split():
                    i[0] += 1
                    w = ws[i[0] % 4]
                    for lo, hi in zip(edges[:-1], edges[1:]):
                        torch.matmul(x, w[lo:hi].t(), out=out[:, lo:hi])

                rec["split_at_" + "_".join(map(str, cut)) + "_us"] = round(timed(split), 1)
            print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
