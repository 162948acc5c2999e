"""Staged GPU diagnostic: drives the engine through the async (threaded) front with the
benchmark's prompt shapes and prints every step's composition before it executes."""
import asyncio
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "multi_round_qa"))

from vllm_production_stack_amd.engine.async_engine import AsyncLLMEngine  # noqa: E402
from vllm_production_stack_amd.engine.config import EngineConfig  # noqa: E402
from vllm_production_stack_amd.engine.llm_engine import LLMEngine  # noqa: E402
from vllm_production_stack_amd.engine.request import SamplingParams  # noqa: E402
import multi_round_qa as mrq  # noqa: E402


def main():
    users = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    hist = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
    cfg = EngineConfig(model="llama-3-8b", max_model_len=32000, max_num_seqs=256,
                       max_num_batched_tokens=8192)
    eng = LLMEngine(cfg)
    orig_execute = eng.runner.execute

    def traced(out):
        print("STEP dec=%d pre=%s" % (len(out.decodes), [(s.start, s.n, len(s.req.pages)) for s in out.prefills]),
              flush=True)
        r = orig_execute(out)
        if os.environ.get('DIAG_SYNC'):
            torch.cuda.synchronize()
        return r

    eng.runner.execute = traced
    tok = eng.tokenizer

    async def go():
        ae = AsyncLLMEngine(eng)
        ae.start(asyncio.get_running_loop())

        async def one(u, rounds=2):
            msgs = [{"role": "user", "content": mrq.system_prompt(1000, u, hist) + "Here's question #1: tell me a story"}]
            for r in range(rounds):
                ids = tok.encode_chat(msgs)
                out = []
                async for so in ae.generate(f"u{u}r{r}", ids, SamplingParams(max_tokens=int(os.environ.get('DIAG_MAXTOK', '20')), temperature=0)):
                    out.extend(so.new_token_ids)
                msgs.append({"role": "assistant", "content": tok.decode(out)})
                msgs.append({"role": "user", "content": f"Here's question #{r + 2}: another one"})
            return u

        print("STAGE single user", flush=True)
        await one(0, 1)
        print("STAGE %d users" % users, flush=True)
        await asyncio.gather(*[one(u) for u in range(1, 1 + users)])
        print("DIAG OK", flush=True)
        ae.stop()

    asyncio.run(go())


if __name__ == "__main__":
    main()
