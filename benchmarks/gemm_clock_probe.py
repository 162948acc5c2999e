import sys, os, json, statistics, torch, torch.nn.functional as F
sys.path.insert(0, "/root/repo")
from vllm_production_stack_amd import ops
DEV = torch.device("cuda", 0)
def rnd(*s, scale=1.0): return ((torch.rand(*s, device=DEV) * 2 - 1) * scale).to(torch.bfloat16)
m = 8192
big_a = torch.empty(1 << 29, dtype=torch.uint8, device=DEV); big_b = torch.empty_like(big_a)
for name, n, k in (("o", 4096, 4096), ("gate_up", 28672, 4096), ("down", 4096, 14336)):
    ws = [rnd(n, k, scale=0.05) for _ in range(3)]; x = rnd(m, k)
    for mode in ("back_to_back", "after_1ms_copy", "after_idle_2ms"):
        res = {}
        for v, fn in (("library", lambda w: F.linear(x, w)), ("tile", lambda w: ops.tile_gemm(x, w))):
            ts = []
            for it in range(24):
                w = ws[it % 3]
                if mode == "after_1ms_copy": big_b.copy_(big_a)
                elif mode == "after_idle_2ms":
                    torch.cuda.synchronize(); import time; time.sleep(0.002)
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record(); fn(w); e1.record(); torch.cuda.synchronize()
                if it >= 4: ts.append(e0.elapsed_time(e1) * 1e3)
            res[v] = round(statistics.median(ts), 1)
        print(json.dumps({"case": name, "mode": mode, **res, "tile_vs_library": round(res["library"] / res["tile"], 3)}), flush=True)
