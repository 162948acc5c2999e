"""HBM read roof of the card by loads in flight, cache policy, access pattern and grid size
(`csrc/tools/roof_probe.hip`, built with the other native libraries).

    python benchmarks/roof_probe.py        (one JSON line per variant, best grid size of each)
"""
import ctypes
import json
import os
import time

import torch

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(root, "vllm_production_stack_amd", "_native", "libvpsa_roof_probe.so"))
dev = torch.device("cuda", 0)
buf = torch.empty(8 << 30, dtype=torch.uint8, device=dev).random_(0, 255)
sink = torch.zeros(1, dtype=torch.int32, device=dev)
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def run(blocks, unroll, nt, chunked):
    rc = lib.roof_probe(ctypes.c_void_p(buf.data_ptr()), ctypes.c_long(buf.numel()), ctypes.c_void_p(sink.data_ptr()),
                        blocks, unroll, nt, chunked, stream)
    assert rc == 0


for unroll in (2, 4, 8, 16):
    for nt in (0, 1):
        for chunked in (0, 1):
            best = None
            for blocks in (512, 1024, 2048, 4096, 8192, 16384):
                for _ in range(2):
                    run(blocks, unroll, nt, chunked)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(5):
                    run(blocks, unroll, nt, chunked)
                torch.cuda.synchronize()
                tbs = buf.numel() * 5 / (time.perf_counter() - t0) / 1e12
                if best is None or tbs > best[1]:
                    best = (blocks, tbs)
            print(json.dumps({"loads_in_flight": unroll, "nontemporal": bool(nt), "contiguous_pieces": bool(chunked),
                              "best_blocks": best[0], "TB_per_s": round(best[1], 3)}), flush=True)
