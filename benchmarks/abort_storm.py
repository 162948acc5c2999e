"""Clients that hang up mid-stream, at scale: waves of long-prompt streaming requests against an
engine's HTTP server, each cancelled after a random delay (during queueing, chunked prefill, a
host-tier reload or decoding).  Afterwards no page may hold a reference, the pool must be whole,
the engine healthy, and a normal request must still be answered.  Prints one JSON line.

    python benchmarks/abort_storm.py [--model llama-3-8b] [--waves 3] [--clients 48] [--cpu-offload-gb 8]
"""
import argparse
import asyncio
import json
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="llama-3-8b")
    ap.add_argument("--waves", type=int, default=3)
    ap.add_argument("--clients", type=int, default=48)
    ap.add_argument("--prompt-words", type=int, default=6000)
    ap.add_argument("--cpu-offload-gb", type=float, default=8.0)
    ap.add_argument("--num-gpu-blocks", type=int, default=4000)
    ap.add_argument("--device", default="cuda")
    ap.add_argument("--port", type=int, default=18760)
    a = ap.parse_args()

    import aiohttp

    from vllm_production_stack_amd.engine.async_engine import AsyncLLMEngine
    from vllm_production_stack_amd.engine.config import EngineConfig, KVOffloadConfig
    from vllm_production_stack_amd.engine.llm_engine import LLMEngine
    from vllm_production_stack_amd.engine.server import serve

    cpu = a.device == "cpu"
    eng = LLMEngine(EngineConfig(
        model=a.model, device=a.device, dtype="float32" if cpu else "auto", max_model_len=32000, max_num_seqs=64,
        max_num_batched_tokens=8192, num_gpu_blocks_override=a.num_gpu_blocks, port=0,
        kv_offload=KVOffloadConfig(local_cpu=a.cpu_offload_gb > 0, max_local_cpu_gb=a.cpu_offload_gb)))
    rng = random.Random(0)
    shared = " ".join(f"s{j}" for j in range(a.prompt_words // 3))
    stats = {"cancelled": 0, "finished": 0, "errors": 0}

    async def one(s, url, wave, i):
        # a third of the clients repeat a conversation of the wave before (prefix hits / host reloads)
        who = (wave - 1, i) if wave and i % 3 == 0 else (wave, i)
        words = shared + " " + " ".join(f"u{who[0]}x{who[1]}w{j}" for j in range(a.prompt_words - a.prompt_words // 3))
        body = {"model": a.model, "stream": True, "max_tokens": 64, "temperature": 0,
                "messages": [{"role": "user", "content": words}]}
        delay = rng.choice((0.0, 0.02, 0.1, 0.3, 0.8, 2.0, 99.0))
        try:
            async def read():
                async with s.post(url + "/v1/chat/completions", json=body) as r:
                    if r.status != 200:
                        stats["errors"] += 1
                        return
                    async for _ in r.content.iter_any():
                        pass
                    stats["finished"] += 1
            await asyncio.wait_for(read(), timeout=delay)
        except asyncio.TimeoutError:
            stats["cancelled"] += 1
        except aiohttp.ClientError:
            stats["errors"] += 1

    async def go():
        ae = AsyncLLMEngine(eng)
        ae.start(asyncio.get_running_loop())
        runner = await serve(ae, "127.0.0.1", a.port)
        url = f"http://127.0.0.1:{a.port}"
        t0 = time.time()
        try:
            for wave in range(a.waves):
                async with aiohttp.ClientSession(timeout=aiohttp.ClientTimeout(total=None)) as s:
                    await asyncio.gather(*[one(s, url, wave, i) for i in range(a.clients)])
            # the engine thread notices hang-ups at its next steps: let it drain
            for _ in range(600):
                if not eng.has_work():
                    break
                await asyncio.sleep(0.05)
            await asyncio.sleep(0.5)
            async with aiohttp.ClientSession() as s:
                async with s.post(url + "/v1/chat/completions", json={
                        "model": a.model, "max_tokens": 8, "temperature": 0,
                        "messages": [{"role": "user", "content": "still there?"}]}) as r:
                    ok = r.status == 200 and (await r.json())["usage"]["completion_tokens"] == 8
                async with s.get(url + "/health") as r:
                    healthy = r.status == 200
        finally:
            await runner.cleanup()
        return ok, healthy, time.time() - t0, ae

    ok, healthy, wall, ae = asyncio.run(go())
    if eng.connector is not None:
        for _ in range(200):
            eng.connector.poll()
            if not any(eng.bm.ref):
                break
            time.sleep(0.01)
    held = sum(1 for r in eng.bm.ref if r)
    free_ids = list(eng.bm.free_ids)
    res = {"case": "abort_storm", "model": a.model, "waves": a.waves, "clients": a.clients, **stats,
           "wall_s": round(wall, 1), "engine_has_work": eng.has_work(), "pages_still_referenced": held,
           "free_list_duplicates": len(free_ids) - len(set(free_ids)),
           "free_and_cached": len(set(free_ids) & set(eng.bm.evictable)),
           "pages_free_or_cached": eng.bm.num_free(), "pages": a.num_gpu_blocks,
           "request_after_the_storm_ok": ok, "healthy": healthy, "preemptions": eng.scheduler.num_preemptions}
    print(json.dumps(res), flush=True)
    ae.stop()
    bad = (held or res["free_list_duplicates"] or res["free_and_cached"] or not ok or not healthy
           or eng.bm.num_free() != a.num_gpu_blocks or stats["errors"])
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
