"""A dependency-free fake engine replica for router tests and router load tests.

Streams ``Hello `` tokens at ``--speed`` tokens/s after ``--ttft`` seconds and exposes the
same control surface as a real replica (/v1/models, /metrics, /health, sleep mode, LoRA
list).  Role of /root/reference/src/tests/perftest/fake-openai-server.py:31-152, without
its dependency on the vLLM package."""

from __future__ import annotations

import argparse
import asyncio
import json
import time
import uuid

from aiohttp import web


class FakeEngine:
    def __init__(self, model: str = "fake_model_name", speed: float = 500.0, ttft: float = 0.0,
                 max_tokens: int = 100, lora: list[str] | None = None, api_key: str | None = None):
        self.model = model
        self.speed = speed
        self.ttft = ttft
        self.max_tokens = max_tokens
        self.running = 0
        self.sleeping = False
        self.api_key = api_key
        self.lora = list(lora or [])
        self.requests: list[dict] = []  # bodies seen, for assertions
        self.aborted = 0  # streams whose peer hung up before the end
        self.headers_seen: list[dict] = []
        self.app = web.Application(client_max_size=64 * 1024 * 1024)
        r = self.app.router
        r.add_get("/health", self.health)
        r.add_get("/v1/models", self.models)
        r.add_get("/metrics", self.metrics)
        r.add_post("/v1/chat/completions", self.chat)
        r.add_post("/v1/completions", self.completions)
        r.add_post("/v1/embeddings", self.embeddings)
        r.add_post("/tokenize", self.tokenize)
        r.add_post("/detokenize", self.detokenize)
        for p in ("/v1/rerank", "/rerank", "/v1/score", "/score"):
            r.add_post(p, self.generic)
        r.add_post("/v1/audio/transcriptions", self.transcribe)
        r.add_post("/sleep", self.sleep)
        r.add_post("/wake_up", self.wake)
        r.add_get("/is_sleeping", self.is_sleeping)
        r.add_post("/v1/load_lora_adapter", self.load_lora)
        r.add_post("/v1/unload_lora_adapter", self.unload_lora)

    def _auth_ok(self, request) -> bool:
        return not self.api_key or request.headers.get("Authorization") == f"Bearer {self.api_key}"

    async def health(self, request):
        return web.Response(status=200)

    async def models(self, request):
        if not self._auth_ok(request):
            return web.json_response({"error": "Unauthorized"}, status=401)
        data = [{"id": self.model, "object": "model", "created": 0, "owned_by": "vllm",
                 "root": self.model, "parent": None}]
        data += [{"id": n, "object": "model", "created": 0, "owned_by": "vllm", "root": "/lora/" + n,
                  "parent": self.model} for n in self.lora]
        return web.json_response({"object": "list", "data": data})

    async def metrics(self, request):
        text = (f'vllm:num_requests_running{{model_name="{self.model}"}} {self.running}\n'
                f'vllm:num_requests_waiting{{model_name="{self.model}"}} 0\n'
                f'vllm:num_requests_swapped{{model_name="{self.model}"}} 0\n'
                f'vllm:gpu_cache_usage_perc{{model_name="{self.model}"}} 0.1\n'
                f'vllm:gpu_prefix_cache_hit_rate{{model_name="{self.model}"}} 0.5\n'
                f'vllm:gpu_prefix_cache_hits_total{{model_name="{self.model}"}} 5\n'
                f'vllm:gpu_prefix_cache_queries_total{{model_name="{self.model}"}} 10\n')
        return web.Response(text=text)

    async def _generate(self, request, chat: bool):
        if not self._auth_ok(request):
            return web.json_response({"error": "Unauthorized"}, status=401)
        body = await request.json()
        self.requests.append(body)
        self.headers_seen.append(dict(request.headers))
        if self.sleeping:
            return web.json_response({"error": "sleeping"}, status=503)
        n = int(body.get("max_completion_tokens") or body.get("max_tokens") or self.max_tokens)
        rid = request.headers.get("X-Request-Id", uuid.uuid4().hex)
        model = body.get("model", self.model)
        self.running += 1
        try:
            if self.ttft:
                await asyncio.sleep(self.ttft)
            if not body.get("stream"):
                await asyncio.sleep(n / self.speed)
                text = "Hello " * n
                usage = {"prompt_tokens": 1, "completion_tokens": n, "total_tokens": n + 1}
                if chat:
                    return web.json_response({
                        "id": "chatcmpl-" + rid, "object": "chat.completion", "created": int(time.time()),
                        "model": model, "choices": [{"index": 0, "message": {"role": "assistant", "content": text},
                                                     "finish_reason": "length"}], "usage": usage})
                return web.json_response({
                    "id": "cmpl-" + rid, "object": "text_completion", "created": int(time.time()),
                    "model": model, "choices": [{"index": 0, "text": text, "finish_reason": "length"}],
                    "usage": usage})
            resp = web.StreamResponse(headers={"Content-Type": "text/event-stream"})
            await resp.prepare(request)
            t0 = time.time()
            for i in range(n):
                wait = t0 + (i + 1) / self.speed - time.time()
                if wait > 0:
                    await asyncio.sleep(wait)
                if chat:
                    d = {"id": "chatcmpl-" + rid, "object": "chat.completion.chunk", "model": model,
                         "choices": [{"index": 0, "delta": {"content": "Hello "}, "finish_reason": None}]}
                else:
                    d = {"id": "cmpl-" + rid, "object": "text_completion", "model": model,
                         "choices": [{"index": 0, "text": "Hello ", "finish_reason": None}]}
                try:
                    await resp.write(b"data: " + json.dumps(d).encode() + b"\n\n")
                except ConnectionResetError:
                    self.aborted += 1  # the peer hung up: a real engine aborts the request here
                    return resp
            if (body.get("stream_options") or {}).get("include_usage"):
                d = {"id": rid, "choices": [], "model": model,
                     "usage": {"prompt_tokens": 1, "completion_tokens": n, "total_tokens": n + 1}}
                await resp.write(b"data: " + json.dumps(d).encode() + b"\n\n")
            await resp.write(b"data: [DONE]\n\n")
            await resp.write_eof()
            return resp
        finally:
            self.running -= 1

    async def chat(self, request):
        return await self._generate(request, True)

    async def completions(self, request):
        return await self._generate(request, False)

    async def embeddings(self, request):
        body = await request.json()
        self.requests.append(body)
        return web.json_response({"object": "list", "model": body.get("model"),
                                  "data": [{"object": "embedding", "index": 0, "embedding": [0.0] * 8}]})

    async def tokenize(self, request):
        body = await request.json()
        text = body.get("prompt")
        if text is None:
            text = " ".join(str(m.get("content", "")) for m in body.get("messages", []))
        toks = [abs(hash(w)) % 50000 for w in text.split()]
        return web.json_response({"tokens": toks, "count": len(toks), "max_model_len": 4096})

    async def detokenize(self, request):
        body = await request.json()
        return web.json_response({"prompt": " ".join(f"t{t}" for t in body.get("tokens", []))})

    async def generic(self, request):
        body = await request.json()
        self.requests.append(body)
        return web.json_response({"id": uuid.uuid4().hex, "model": body.get("model"), "data": [], "results": []})

    async def transcribe(self, request):
        reader = await request.multipart()
        fields = {}
        async for part in reader:
            fields[part.name] = await part.read(decode=False)
        self.requests.append({k: (v if k == "file" else v.decode()) for k, v in fields.items()})
        return web.json_response({"text": "fake transcript"})

    async def sleep(self, request):
        self.sleeping = True
        return web.Response(status=200)

    async def wake(self, request):
        self.sleeping = False
        return web.Response(status=200)

    async def is_sleeping(self, request):
        return web.json_response({"is_sleeping": self.sleeping})

    async def load_lora(self, request):
        body = await request.json()
        if body["lora_name"] in self.lora:
            return web.json_response({"message": f"The lora adapter '{body['lora_name']}' has already been loaded."}, status=400)
        self.lora.append(body["lora_name"])
        return web.Response(text="Success")

    async def unload_lora(self, request):
        body = await request.json()
        if body["lora_name"] not in self.lora:
            return web.json_response({"message": "not found"}, status=404)
        self.lora.remove(body["lora_name"])
        return web.Response(text="Success")

    async def start(self, host: str = "127.0.0.1", port: int = 0) -> "FakeEngine":
        self.runner = web.AppRunner(self.app, access_log=None)
        await self.runner.setup()
        site = web.TCPSite(self.runner, host, port)
        await site.start()
        self.port = self.runner.addresses[0][1]
        self.url = f"http://{host}:{self.port}"
        return self

    async def stop(self) -> None:
        await self.runner.cleanup()


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--host", default="127.0.0.1")
    ap.add_argument("--port", type=int, default=9000)
    ap.add_argument("--model", default="fake_model_name")
    ap.add_argument("--speed", type=float, default=500)
    ap.add_argument("--ttft", type=float, default=0.0)
    ap.add_argument("--max-tokens", type=int, default=100)
    a = ap.parse_args()
    eng = FakeEngine(a.model, a.speed, a.ttft, a.max_tokens)
    web.run_app(eng.app, host=a.host, port=a.port, access_log=None)


if __name__ == "__main__":
    main()
