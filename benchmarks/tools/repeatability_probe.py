"""Which products of a Llama-3-8B step are bitwise repeatable?  Runs every projection 30 times on
the same operands, through the library and through this tree's dispatch (``ops.linear`` /
``ops.gate_up_silu``), at decode and prefill row counts; one JSON line per case.

    python benchmarks/tools/repeatability_probe.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vllm_production_stack_amd import ops  # noqa: E402

SHAPES = {"qkv": (6144, 4096), "o": (4096, 4096), "gate_up": (28672, 4096), "down": (4096, 14336),
          "lm_head": (128256, 4096)}


def main():
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    for name, (n, k) in SHAPES.items():
        w = torch.randn(n, k, device=dev, dtype=torch.bfloat16) * 0.02
        for m in (2, 40, 2000, 8192):
            if name == "lm_head" and m > 40:
                continue
            x = torch.randn(m, k, device=dev, dtype=torch.bfloat16)
            row = {"case": name, "rows": m}
            fns = {"library": lambda: torch.nn.functional.linear(x, w),
                   "dispatch": (lambda: ops.gate_up_silu(x, w)) if name == "gate_up" else (lambda: ops.linear(x, w))}
            for label, fn in fns.items():
                ref = fn().clone()
                row[label + "_runs_differing_of_30"] = sum(int(not torch.equal(fn(), ref)) for _ in range(30))
            print(json.dumps(row), flush=True)
        del w


if __name__ == "__main__":
    main()
