"""An expert's prefill GEMMs at the row counts one chunk gives it (about a quarter of the chunk's
tokens): the library's default against other formulations of the same product.

    python benchmarks/expert_gemm_variants.py [--rows 1024,2048,4096]
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
    ap.add_argument("--rows", default="1024,2048,4096")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    for m in [int(r) for r in a.rows.split(",")]:
        for name, n, k in (("w13", 28672, 4096), ("w2", 4096, 14336)):
            ws = [(torch.randn(n, k, device=dev) * 0.02).to(torch.bfloat16) for _ in range(4)]
            x = torch.randn(m, k, device=dev).to(torch.bfloat16)
            out = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
            i = [0]

            def nxt():
                i[0] += 1
                return ws[i[0] % 4]

            def linear():
                return F.linear(x, nxt())

            def swapped():
                return torch.matmul(nxt(), x.t())

            def split_k2():
                w = nxt()
                h = k // 2
                torch.matmul(x[:, :h], w[:, :h].t(), out=out)
                out.addmm_(x[:, h:], w[:, h:].t())

            def split_n2():
                w = nxt()
                h = n // 2
                torch.matmul(x, w[:h].t(), out=out[:, :h])
                torch.matmul(x, w[h:].t(), out=out[:, h:])

            rec = {"case": name, "rows": m, "n": n, "k": k}
            for vname, fn in (("linear", linear), ("swapped", swapped), ("split_k2", split_k2), ("split_n2", split_n2)):
                us = timed(fn)
                rec[vname + "_us"] = round(us, 1)
                rec[vname + "_tflops"] = round(2 * m * n * k / us / 1e6, 1)
            print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
