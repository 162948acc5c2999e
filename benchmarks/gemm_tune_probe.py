"""Does the library GEMM's default pick leave time on the table for the prefill shapes?

For every (rows, out, in) projection shape of a model this times ``F.linear`` with the library's
heuristic choice, then lets PyTorch's TunableOp time every hipBLASLt / rocBLAS solution for the shape
and times the winner (TunableOp writes its choices to ``--out`` when the process exits).

    python benchmarks/gemm_tune_probe.py [--rows 8192,4616] [--out gpurun_out/gemm_tuned.csv]
"""

import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch
import torch.nn.functional as F

SHAPES = {  # out features, in features
    "llama-3-8b": [("qkv", 6144, 4096), ("o", 4096, 4096), ("gate_up", 28672, 4096), ("down", 4096, 14336),
                   ("lm_head", 128256, 4096)],
}


def timed(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="llama-3-8b")
    ap.add_argument("--rows", default="8192")
    ap.add_argument("--out", default="gpurun_out/gemm_tuned.csv")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--skip", default="lm_head")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    tun = torch.cuda.tunable
    rows = [int(r) for r in a.rows.split(",")]
    skip = set(a.skip.split(","))
    cases = [(name, m, n, k) for name, n, k in SHAPES[a.model] if name not in skip for m in rows]
    # several weights per shape so that consecutive calls do not find their operands in L2 / MALL
    base = {}
    for name, m, n, k in cases:
        ws = [(torch.randn(n, k, device=dev) * 0.02).to(torch.bfloat16) for _ in range(4)]
        x = torch.randn(m, k, device=dev).to(torch.bfloat16)
        it = iter(range(10 ** 9))
        base[(name, m)] = (x, ws)
        dt = timed(lambda: F.linear(x, ws[next(it) % 4]), a.iters)
        print(json.dumps({"case": name, "rows": m, "n": n, "k": k, "mode": "default", "us": round(dt * 1e6, 1),
                          "tflops": round(2 * m * n * k / dt / 1e12, 1)}), flush=True)
    tun.enable(True)
    tun.tuning_enable(True)
    tun.set_max_tuning_duration(30)
    tun.set_max_tuning_iterations(20)
    tun.set_filename(a.out)
    for name, m, n, k in cases:
        x, ws = base[(name, m)]
        t0 = time.perf_counter()
        F.linear(x, ws[0])  # tunes
        torch.cuda.synchronize()
        tune_s = time.perf_counter() - t0
        it = iter(range(10 ** 9))
        dt = timed(lambda: F.linear(x, ws[next(it) % 4]), a.iters)
        print(json.dumps({"case": name, "rows": m, "n": n, "k": k, "mode": "tuned", "us": round(dt * 1e6, 1),
                          "tflops": round(2 * m * n * k / dt / 1e12, 1), "tuning_s": round(tune_s, 1)}), flush=True)
    print(json.dumps({"results": [list(map(str, r)) for r in tun.get_results()]}), flush=True)


if __name__ == "__main__":
    main()
