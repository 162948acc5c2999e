"""Where does cold-prompt TTFT go?  Engine-only timing of ~21k-token prompts under different idle
gaps and prompt-length patterns, with the step composition, then the same prompts through the HTTP
server of the engine (no router).  Diagnostic tool; prints JSON lines."""
import argparse
import asyncio
import copy
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
    ap.add_argument("--n", type=int, default=5)
    ap.add_argument("--gaps", default="0,3", help="idle seconds in front of every prompt, one series per value")
    ap.add_argument("--no-chunked-prefill", action="store_true", help="whole prompts in one step")
    ap.add_argument("--no-prefix-caching", action="store_true")
    ap.add_argument("--cpu-offload-gb", type=float, default=0.0)
    ap.add_argument("--engine-only", action="store_true", help="skip the HTTP half")
    ap.add_argument("--mrq-style", action="store_true",
                    help="HTTP half: the benchmark client's own prompts (new users, then a follow-up turn "
                         "each, 100 answer tokens), with the engine's time split per request")
    a = ap.parse_args()
    from vllm_production_stack_amd.engine.config import KVOffloadConfig

    eng = LLMEngine(EngineConfig(model=a.model, max_model_len=32000, max_num_seqs=256,
                                 max_num_batched_tokens=32000 if a.no_chunked_prefill else 8192,
                                 enable_chunked_prefill=not a.no_chunked_prefill,
                                 enable_prefix_caching=not a.no_prefix_caching,
                                 kv_offload=KVOffloadConfig(local_cpu=a.cpu_offload_gb > 0,
                                                            max_local_cpu_gb=a.cpu_offload_gb)))
    g = torch.Generator().manual_seed(0)
    sp = SamplingParams(max_tokens=1, temperature=0, ignore_eos=True)

    def prompt(n):
        return torch.randint(16, 100000, (n,), generator=g).tolist()

    def run(p):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.add_request(p, copy.deepcopy(sp))
        steps = []
        while eng.has_work():
            s0 = time.perf_counter()
            eng.step()
            steps.append(round((time.perf_counter() - s0) * 1e3, 1))
        torch.cuda.synchronize()
        return round(time.perf_counter() - t0, 4), steps

    eng.generate([prompt(300)], copy.deepcopy(sp))
    series = []
    for k, gap in enumerate(float(v) for v in a.gaps.split(",")):
        series += [(gap, [21000] * a.n), (gap, [21000 + 500 * k + 137 * i for i in range(a.n)])]
    for gap, lens in series:
        out = []
        for n in lens:
            if gap:
                time.sleep(gap)
            out.append(run(prompt(n)))
        print(json.dumps({"case": "engine_only", "idle_gap_s": gap, "lens": lens,
                          "seconds": [o[0] for o in out], "step_ms": [o[1] for o in out]}), flush=True)

    if a.engine_only:
        print(json.dumps({"case": "prof", **{k: round(v, 3) if isinstance(v, float) else v
                                             for k, v in eng.prof.items()}}), flush=True)
        eng.shutdown()
        return
    # through the engine's HTTP server, directly and behind the router
    import subprocess

    import aiohttp

    from vllm_production_stack_amd.engine.async_engine import AsyncLLMEngine
    from vllm_production_stack_amd.engine.server import serve

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    async def timed_chat(s, url, body):
        t0 = time.perf_counter()
        first = None
        buf = b""
        async with s.post(url + "/v1/chat/completions", json=body, headers={"x-user-id": "7"}) as r:
            async for chunk in r.content.iter_any():
                if first is None:
                    buf += chunk
                    for line in buf.split(b"\n"):
                        if line.startswith(b"data: {"):
                            try:
                                d = json.loads(line[6:])
                            except json.JSONDecodeError:
                                continue
                            ch = d.get("choices") or [{}]
                            if (ch[0].get("delta") or {}).get("content"):
                                first = time.perf_counter() - t0
                                break
        return {"ttft": round(first or -1, 4), "total": round(time.perf_counter() - t0, 4)}

    async def mrq_part():
        sys.path.insert(0, os.path.join(root, "benchmarks", "multi_round_qa"))
        import multi_round_qa as mrq

        ae = AsyncLLMEngine(eng)
        ae.start(asyncio.get_running_loop())
        runner = await serve(ae, "127.0.0.1", 18700)
        ex = mrq.RequestExecutor("http://127.0.0.1:18700", "EMPTY", a.model)
        await ex.start()
        cfg = mrq.WorkloadConfig(4, 1000, 20000, 100, 20, 1.0, a.model, True)
        users = [mrq.UserSession(900 + i, cfg) for i in range(a.n)]
        rows = []
        try:
            for turn in range(2):
                for u in users:
                    await asyncio.sleep(2.0)
                    before = dict(eng.prof)
                    r = await ex.chat(u.next_messages(), 100, u.headers())
                    u.on_answer(r)
                    d = {k: round(eng.prof[k] - before[k], 3) for k in ("t_schedule", "t_execute", "t_post")}
                    rows.append({"turn": turn + 1, "client_ttft": round(r.ttft, 4), "prompt_tokens": r.prompt_tokens,
                                 "steps": eng.prof["steps"] - before["steps"], **d})
        finally:
            await ex.close()
            await runner.cleanup()
            ae.stop()
        return rows

    if a.mrq_style:
        for row in asyncio.run(mrq_part()):
            print(json.dumps({"case": "http_mrq", **row}), flush=True)
        return

    async def http_part():
        ae = AsyncLLMEngine(eng)
        ae.start(asyncio.get_running_loop())
        runner = await serve(ae, "127.0.0.1", 18700)
        router = subprocess.Popen(
            [sys.executable, "-m", "vllm_production_stack_amd.router.app", "--host", "127.0.0.1", "--port", "18701",
             "--service-discovery", "static", "--static-backends", "http://127.0.0.1:18700", "--static-models",
             a.model, "--routing-logic", "session", "--session-key", "x-user-id", "--log-level", "warning"], cwd=root)
        await asyncio.sleep(3.0)
        res = {"direct": [], "via_router": []}
        try:
            async with aiohttp.ClientSession() as s:
                for i in range(2 * a.n):
                    await asyncio.sleep(1.0)
                    words = " ".join(f"w{(i * 7919 + j) % 50000}" for j in range(21000))
                    body = {"model": a.model, "stream": True, "max_tokens": 4, "temperature": 0,
                            "messages": [{"role": "user", "content": words}]}
                    key, url = (("direct", "http://127.0.0.1:18700") if i % 2 == 0
                                else ("via_router", "http://127.0.0.1:18701"))
                    res[key].append(await timed_chat(s, url, body))
        finally:
            router.terminate()
            router.wait(timeout=10)
            await runner.cleanup()
            ae.stop()
        return res

    print(json.dumps({"case": "http", **asyncio.run(http_part()),
                      "prof": {k: round(v, 3) if isinstance(v, float) else v for k, v in eng.prof.items()}}), flush=True)


if __name__ == "__main__":
    main()
