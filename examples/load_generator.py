#!/usr/bin/env python3
"""Constant-rate load generator for autoscaling demos: N requests/s of streamed chat completions;
prints the router's queue-depth gauge (the HPA / KEDA signal) every few seconds."""
import argparse
import asyncio
import re
import time

import aiohttp


async def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--base-url", default="http://localhost:30080")
    ap.add_argument("--model", default="llama-3-8b")
    ap.add_argument("--qps", type=float, default=5.0)
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--max-tokens", type=int, default=200)
    a = ap.parse_args()
    done = [0]

    async with aiohttp.ClientSession(connector=aiohttp.TCPConnector(limit=0)) as s:
        async def one(i):
            body = {"model": a.model, "stream": True, "max_tokens": a.max_tokens,
                    "messages": [{"role": "user", "content": f"Request {i}: write a short story about a GPU."}]}
            async with s.post(a.base_url + "/v1/chat/completions", json=body) as r:
                async for _ in r.content.iter_any():
                    pass
            done[0] += 1

        async def watch():
            while True:
                await asyncio.sleep(5)
                async with s.get(a.base_url + "/metrics") as r:
                    text = await r.text()
                waiting = sum(float(x) for x in re.findall(r"^vllm:num_requests_waiting\{[^}]*\} ([\d.e+-]+)$", text, re.M))
                print(f"t={time.time() - t0:5.0f}s finished={done[0]} waiting={waiting:.0f}", flush=True)

        t0 = time.time()
        w = asyncio.create_task(watch())
        tasks, i = [], 0
        while time.time() - t0 < a.seconds:
            tasks.append(asyncio.create_task(one(i)))
            i += 1
            await asyncio.sleep(1.0 / a.qps)
        await asyncio.gather(*tasks, return_exceptions=True)
        w.cancel()


if __name__ == "__main__":
    asyncio.run(main())
