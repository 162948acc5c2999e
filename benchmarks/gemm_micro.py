"""Decode-batch projections: the weight-streaming MFMA kernel (csrc/kernels/gemm.hip) against
the library GEMM at the Llama-3-8B shapes, plus the HBM read roof of the card.

    python benchmarks/gemm_micro.py [--rows 1,8,16,40,64] [--variants 0,1,2,3]

Every shape rotates through enough weight copies to exceed the 256 MB last-level cache, so the
numbers are HBM numbers.  One JSON line per (shape, rows)."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vllm_production_stack_amd import ops  # noqa: E402

SHAPES = {  # name: (N, K, fused SwiGLU)
    "qkv": (6144, 4096, False), "o": (4096, 4096, False), "gate_up": (28672, 4096, True),
    "down": (4096, 14336, False), "lm_head": (128256, 4096, False),
}


def timed(fn, iters, graph=True):
    """Seconds per call.  The calls are captured into one HIP graph and replayed, as the
    engine's decode steps are, so that host launch gaps do not count."""
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if not graph:
        a.record()
        for i in range(iters):
            fn(i)
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / iters * 1e-3
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for i in range(3):
            fn(i)
        with torch.cuda.graph(g, stream=s):
            for i in range(iters):
                fn(i)
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    a.record()
    for _ in range(3):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / (3 * iters) * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", default="1,4,8,16,24,32,40,64")
    ap.add_argument("--variants", default="0")
    ap.add_argument("--shapes", default=",".join(SHAPES))
    ap.add_argument("--packed", action="store_true", help="also time the pre-packed-weight kernel")
    ap.add_argument("--packed-targets", default="512", help="workgroup targets to try for its split count")
    ap.add_argument("--no-probe", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    buf = torch.empty((64 << 20) if a.no_probe else (4 << 30), dtype=torch.uint8, device=dev).random_(0, 255)
    for blocks in (() if a.no_probe else (1024, 2048, 4096, 8192)):
        t = timed(lambda i=0: ops.hbm_read_probe(buf, blocks), 10, graph=False)
        print(json.dumps({"case": "hbm_read_probe", "blocks": blocks, "GB": 4.295,
                          "TB_per_s": round(buf.numel() / t / 1e12, 3)}), flush=True)
    del buf
    variants = [int(v) for v in a.variants.split(",")]
    for name in a.shapes.split(","):
        n, k, pair = SHAPES[name]
        copies = max(2, int(600e6 // (n * k * 2)) + 1)
        ws = [torch.randn(n, k, device=dev, dtype=torch.bfloat16) * 0.02 for _ in range(copies)]
        wbytes = n * k * 2
        for m in [int(r) for r in a.rows.split(",")]:
            x = torch.randn(m, k, device=dev, dtype=torch.bfloat16)
            iters = 60

            def lib(i=0):
                y = torch.nn.functional.linear(x, ws[i % copies])
                return ops.silu_and_mul(y) if pair else y

            def mine(i=0):
                return ops.gate_up_silu(x, ws[i % copies]) if pair else ops.linear(x, ws[i % copies])

            ref = torch.nn.functional.linear(x.float(), ws[0].float())
            if pair:
                h = n // 2
                ref = torch.nn.functional.silu(ref[:, :h]) * ref[:, h:]
            row = {"case": name, "rows": m, "N": n, "K": k, "weight_MB": round(wbytes / 1e6, 1)}
            t_lib = timed(lib, iters)
            row["library_us"] = round(t_lib * 1e6, 1)
            row["library_TB_per_s"] = round(wbytes / t_lib / 1e12, 2)
            for v in variants:
                ops.set_gemm_variant(v)
                got = mine(0).float()
                err = float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-6))
                t = timed(mine, iters)
                row[f"v{v}_us"] = round(t * 1e6, 1)
                row[f"v{v}_TB_per_s"] = round(wbytes / t / 1e12, 2)
                row[f"v{v}_rel_err"] = round(err, 5)
            for v in (0, 4):  # streaming kernel: chunks of 128 / 256 elements
                ops.set_gemm_variant(v)

                def stream(i=0):
                    return ops._stream_gemm(x, ws[i % copies], None, pair)

                got = stream(0)
                if got is None:
                    continue
                err = float((got.float() - ref).abs().max() / ref.abs().max().clamp_min(1e-6))
                t = timed(stream, iters)
                row[f"s{v}_us"] = round(t * 1e6, 1)
                row[f"s{v}_TB_per_s"] = round(wbytes / t / 1e12, 2)
                row[f"s{v}_rel_err"] = round(err, 5)
            ops.set_gemm_variant(0)
            if a.packed and ops.packed_supported(n, k, pair):  # pre-packed weights, register-direct
                wps = [ops.pack_weight(w) for w in ws]
                for target in [int(t) for t in a.packed_targets.split(",")]:
                    ops.set_packed_target(target)

                    def packed(i=0):
                        return ops.packed_linear(x, wps[i % copies], n, k, None, pair)

                    got = packed(0)
                    err = float((got.float() - ref).abs().max() / ref.abs().max().clamp_min(1e-6))
                    t = timed(packed, iters)
                    row[f"p{target}_us"] = round(t * 1e6, 1)
                    row[f"p{target}_TB_per_s"] = round(wbytes / t / 1e12, 2)
                    row[f"p{target}_rel_err"] = round(err, 5)
                del wps
            print(json.dumps(row), flush=True)
        del ws


if __name__ == "__main__":
    main()
