"""Engine micro-benchmark on one GPU: cold prefill latency and decode step time for the
multi-round-QA shapes (Llama-3-8B, ~21k-token contexts).  Prints one JSON line per case.

    python benchmarks/engine_micro.py --model llama-3-8b --ctx 21000 --batches 1,8,40
"""

import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from vllm_production_stack_amd.engine.config import EngineConfig  # noqa: E402
from vllm_production_stack_amd.engine.llm_engine import LLMEngine  # noqa: E402
from vllm_production_stack_amd.engine.request import SamplingParams  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="llama-3-8b")
    ap.add_argument("--ctx", type=int, default=21000)
    ap.add_argument("--batches", default="1,8,40")
    ap.add_argument("--decode-steps", type=int, default=32)
    ap.add_argument("--chunk", type=int, default=8192)
    ap.add_argument("--no-chunked", action="store_true")
    ap.add_argument("--eager", action="store_true")
    ap.add_argument("--max-num-seqs", type=int, default=64)
    ap.add_argument("--temperature", type=float, default=0.0, help="> 0: stochastic sampling for every request")
    ap.add_argument("--top-p", type=float, default=1.0)
    ap.add_argument("--top-k", type=int, default=-1)
    args = ap.parse_args()

    cfg = EngineConfig(model=args.model, max_model_len=32000, max_num_seqs=args.max_num_seqs,
                       max_num_batched_tokens=args.chunk,
                       enable_chunked_prefill=not args.no_chunked, enforce_eager=args.eager)
    t0 = time.time()
    eng = LLMEngine(cfg)
    print(json.dumps({"case": "init", "seconds": round(time.time() - t0, 2),
                      "pages": eng.runner.num_pages}), flush=True)
    vocab = eng.model_cfg.vocab_size
    g = torch.Generator().manual_seed(0)

    def prompt(n):
        return torch.randint(16, vocab, (n,), generator=g).tolist()

    # warm-up: small prefill + a few decode steps
    eng.generate([prompt(300)], SamplingParams(max_tokens=4, temperature=0, ignore_eos=True))
    torch.cuda.synchronize()

    for b in [int(x) for x in args.batches.split(",")]:
        # early requests decode while later ones are still prefilled: give every request enough
        # tokens to be alive through the whole prefill phase and the timed steps
        prefill_steps = (b * args.ctx + args.chunk - 1) // args.chunk + b
        sp = SamplingParams(max_tokens=args.decode_steps + 4 + prefill_steps, temperature=args.temperature,
                            top_p=args.top_p, top_k=args.top_k, ignore_eos=True)
        reqs = [eng.add_request(prompt(args.ctx), SamplingParams(**sp.__dict__)) for _ in range(b)]
        torch.cuda.synchronize()
        t0 = time.time()
        while any(len(r.output_ids) == 0 for r in reqs):
            eng.step()
        torch.cuda.synchronize()
        t_prefill = time.time() - t0
        # first decode step may capture a graph: run 2 untimed
        eng.step()
        eng.step()
        torch.cuda.synchronize()
        t0 = time.time()
        n = 0
        while eng.has_work() and n < args.decode_steps - 4:
            eng.step()
            n += 1
        torch.cuda.synchronize()
        t_dec = (time.time() - t0) / max(1, n)
        for r in reqs:
            eng.abort(r.request_id)
        while eng.has_work():
            eng.step()
        print(json.dumps({
            "case": f"batch{b}", "ctx": args.ctx, "prefill_s_total": round(t_prefill, 4),
            "prefill_tok_per_s": round(b * args.ctx / t_prefill, 1),
            "decode_ms_per_step": round(t_dec * 1e3, 3),
            "decode_tok_per_s": round(b / t_dec, 1)}), flush=True)
        eng.bm.reset_prefix_cache()


if __name__ == "__main__":
    main()
