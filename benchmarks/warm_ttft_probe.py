"""Engine-only time to the first token of a 21k-token prompt whose KV is fully resident (what the
decode hop of a disaggregated pair, or a follow-up turn, costs): per-step times and how the steps ran."""
import copy, sys, time, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vllm_production_stack_amd.engine.config import EngineConfig
from vllm_production_stack_amd.engine.llm_engine import LLMEngine
from vllm_production_stack_amd.engine.request import SamplingParams
eng = LLMEngine(EngineConfig(model="llama-3-8b", max_model_len=32000, max_num_seqs=64, max_num_batched_tokens=8192,
                             enable_prefix_caching=True, gpu_memory_utilization=0.5))
g = torch.Generator().manual_seed(0)
sp = SamplingParams(max_tokens=1, temperature=0, ignore_eos=True)
p = torch.randint(16, 100000, (21000,), generator=g).tolist()
def run(p):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.add_request(p, copy.deepcopy(sp)); t1 = time.perf_counter()
    steps = []
    while eng.has_work():
        s0 = time.perf_counter(); eng.step(); steps.append(round((time.perf_counter() - s0) * 1e3, 2))
    return round((time.perf_counter() - t0) * 1e3, 2), round((t1 - t0) * 1e3, 2), steps
print("cold", run(p))
for _ in range(4):
    print("warm", run(p), dict(eng.runner.step_kinds))
