"""Sleep and wake a full-size replica (random-init Llama-3-8B, the pool sized from HBM); prints one
JSON line with the pool layout, the memory given back, the timings and whether the greedy tokens of
two 3000-token prompts are the same after waking, next to whether a plain repeat without a nap gives
the same tokens (on random-init weights the logits are nearly tied and full-size runs are not
bitwise repeatable on either pool layout, so only that pair of answers says anything about the nap).

    python benchmarks/tools/sleep_wake_full_size.py [--kv-layer-groups G]"""
import argparse
import copy
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vllm_production_stack_amd.engine.config import EngineConfig  # noqa: E402
from vllm_production_stack_amd.engine.llm_engine import LLMEngine  # noqa: E402
from vllm_production_stack_amd.engine.request import SamplingParams  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kv-layer-groups", type=int, default=0)
    ap.add_argument("--prompt-len", type=int, default=3000)
    ap.add_argument("--repeats", type=int, default=1, help="plain repeats before the nap")
    a = ap.parse_args()
    eng = LLMEngine(EngineConfig(model="llama-3-8b", device="cuda", dtype="bfloat16", max_model_len=8192,
                                 max_num_seqs=8, enable_sleep_mode=True, kv_layer_groups=a.kv_layer_groups, port=0))
    g = torch.Generator().manual_seed(0)
    ps = [torch.randint(16, 30000, (a.prompt_len,), generator=g).tolist() for _ in range(2)]
    sp = SamplingParams(max_tokens=16, temperature=0, ignore_eos=True)
    first = eng.generate(ps, copy.deepcopy(sp))
    same = []
    for _ in range(a.repeats):
        eng.bm.reset_prefix_cache()
        repeat = eng.generate(ps, copy.deepcopy(sp))  # the same work without a nap in between
        same.append(repeat == first)
    free0 = torch.cuda.mem_get_info()[0]
    t0 = time.time()
    freed = eng.sleep(1)
    t1 = time.time()
    free1 = torch.cuda.mem_get_info()[0]
    eng.wake_up()
    t2 = time.time()
    again = eng.generate(ps, copy.deepcopy(sp))
    print(json.dumps({"kv_layer_groups": len(getattr(eng.runner.pool, "groups", [0])),
                      "pages": eng.runner.num_pages, "freed_GiB": round(freed / 2**30, 1),
                      "hbm_free_GiB_before_asleep": [round(free0 / 2**30, 1), round(free1 / 2**30, 1)],
                      "sleep_s": round(t1 - t0, 2), "wake_s": round(t2 - t1, 2), "same_tokens_after_waking": again == first, "same_tokens_on_a_plain_repeat": all(same), "repeats_same": same, "prompt_len": a.prompt_len,
                      "first": first[0][:8], "repeat": repeat[0][:8], "again": again[0][:8]}))
    assert all(len(o) == 16 for o in again)


if __name__ == "__main__":
    main()
