"""HTTP contract of an engine replica (CPU, tiny random-init model)."""

import asyncio
import json

import aiohttp
import pytest

from vllm_production_stack_amd.engine.async_engine import AsyncLLMEngine
from vllm_production_stack_amd.engine.config import EngineConfig
from vllm_production_stack_amd.engine.llm_engine import LLMEngine
from vllm_production_stack_amd.engine.server import serve


def run(coro):
    return asyncio.run(coro)


async def _with_server(fn, **cfg_kw):
    cfg = EngineConfig(model="tiny-llama", device="cpu", dtype="float32", max_model_len=512,
                       max_num_seqs=8, num_gpu_blocks_override=64, **cfg_kw)
    eng = LLMEngine(cfg)
    ae = AsyncLLMEngine(eng)
    ae.start(asyncio.get_running_loop())
    runner = await serve(ae, "127.0.0.1", 0)
    port = runner.addresses[0][1]
    try:
        async with aiohttp.ClientSession(base_url=f"http://127.0.0.1:{port}") as s:
            return await fn(s, eng)
    finally:
        await runner.cleanup()
        ae.stop()


def test_models_health_metrics_version():
    async def fn(s, eng):
        async with s.get("/health") as r:
            assert r.status == 200
        async with s.get("/v1/models") as r:
            d = await r.json()
            assert d["object"] == "list" and d["data"][0]["id"] == "tiny-llama"
            assert d["data"][0]["parent"] is None and d["data"][0]["owned_by"] == "vllm"
        async with s.get("/version") as r:
            assert "version" in await r.json()
        async with s.get("/metrics") as r:
            text = await r.text()
            for name in ("vllm:num_requests_running", "vllm:num_requests_waiting",
                         "vllm:gpu_cache_usage_perc", "vllm:gpu_prefix_cache_hit_rate",
                         "vllm:gpu_prefix_cache_hits_total", "vllm:gpu_prefix_cache_queries_total",
                         "vllm:time_to_first_token_seconds", "vllm:e2e_request_latency_seconds",
                         "vllm:time_per_output_token_seconds", "vllm:num_requests_swapped"):
                assert name in text, name
    run(_with_server(fn))


def test_chat_completion_non_stream_and_stream_agree():
    async def fn(s, eng):
        body = {"model": "tiny-llama", "messages": [{"role": "user", "content": "hello there"}],
                "max_tokens": 8, "temperature": 0, "ignore_eos": True}
        async with s.post("/v1/chat/completions", json=body) as r:
            assert r.status == 200
            d = await r.json()
        assert d["object"] == "chat.completion"
        assert d["usage"]["completion_tokens"] == 8
        text = d["choices"][0]["message"]["content"]
        body["stream"] = True
        body["stream_options"] = {"include_usage": True}
        pieces, usage, done = [], None, False
        async with s.post("/v1/chat/completions", json=body) as r:
            assert r.headers["Content-Type"].startswith("text/event-stream")
            async for line in r.content:
                line = line.decode().strip()
                if not line.startswith("data: "):
                    continue
                if line == "data: [DONE]":
                    done = True
                    continue
                ev = json.loads(line[6:])
                if ev.get("usage"):
                    usage = ev["usage"]
                for c in ev["choices"]:
                    pieces.append(c["delta"].get("content") or "")
        assert done and usage["completion_tokens"] == 8
        assert "".join(pieces) == text
    run(_with_server(fn))


def test_completions_tokenize_detokenize_roundtrip():
    async def fn(s, eng):
        async with s.post("/v1/completions", json={"model": "tiny-llama", "prompt": "a b c",
                                                    "max_tokens": 4, "temperature": 0,
                                                    "ignore_eos": True}) as r:
            d = await r.json()
            assert d["object"] == "text_completion" and d["usage"]["completion_tokens"] == 4
            gen = d["choices"][0]["text"]
        async with s.post("/tokenize", json={"prompt": gen, "add_special_tokens": False}) as r:
            toks = (await r.json())["tokens"]
            assert len(toks) == 4
        async with s.post("/detokenize", json={"tokens": toks}) as r:
            assert (await r.json())["prompt"] == gen
    run(_with_server(fn))


def test_errors_unknown_model_and_too_long():
    async def fn(s, eng):
        async with s.post("/v1/completions", json={"model": "nope", "prompt": "x"}) as r:
            assert r.status == 404
        async with s.post("/v1/completions", json={"model": "tiny-llama",
                                                    "prompt": "hi " * 600}) as r:
            assert r.status == 400
    run(_with_server(fn))


def test_api_key_required():
    async def fn(s, eng):
        async with s.get("/v1/models") as r:
            assert r.status == 401
        async with s.get("/v1/models", headers={"Authorization": "Bearer sk"}) as r:
            assert r.status == 200
        async with s.get("/health") as r:
            assert r.status == 200
    run(_with_server(fn, api_key="sk"))


def test_sleep_wake_cycle():
    async def fn(s, eng):
        async with s.get("/is_sleeping") as r:
            assert (await r.json()) == {"is_sleeping": False}
        async with s.post("/sleep?level=1") as r:
            assert r.status == 200
        async with s.get("/is_sleeping") as r:
            assert (await r.json())["is_sleeping"] is True
        async with s.post("/v1/completions", json={"model": "tiny-llama", "prompt": "x"}) as r:
            assert r.status == 503
        async with s.post("/wake_up") as r:
            assert r.status == 200
        async with s.post("/v1/completions", json={"model": "tiny-llama", "prompt": "x",
                                                    "max_tokens": 2}) as r:
            assert r.status == 200
    run(_with_server(fn, enable_sleep_mode=True))


def test_sleep_is_refused_while_requests_are_in_flight():
    async def fn(s, eng):
        body = {"model": "tiny-llama", "prompt": "a b c", "max_tokens": 200, "temperature": 0,
                "ignore_eos": True, "stream": True}
        async with s.post("/v1/completions", json=body) as r:
            await r.content.readline()           # generation has started
            async with s.post("/sleep?level=1") as z:
                assert z.status == 409
                assert "in flight" in (await z.json())["message"]
            async with s.get("/is_sleeping") as z:
                assert (await z.json())["is_sleeping"] is False
            async for _ in r.content:             # the stream still finishes
                pass
        async with s.post("/sleep?level=1") as z:
            assert z.status == 200
    run(_with_server(fn, enable_sleep_mode=True))


def test_embeddings_score_rerank():
    async def fn(s, eng):
        async with s.post("/v1/embeddings", json={"model": "tiny-llama", "input": ["a b", "c d e"]}) as r:
            d = await r.json()
            assert len(d["data"]) == 2 and len(d["data"][0]["embedding"]) == 256
        async with s.post("/v1/score", json={"model": "tiny-llama", "text_1": "q",
                                              "text_2": ["a", "b"]}) as r:
            assert len((await r.json())["data"]) == 2
        async with s.post("/v1/rerank", json={"model": "tiny-llama", "query": "q",
                                               "documents": ["a", "b", "c"], "top_n": 2}) as r:
            assert len((await r.json())["results"]) == 2
    run(_with_server(fn))


def test_stop_string_truncates():
    async def fn(s, eng):
        body = {"model": "tiny-llama", "prompt": "x y", "max_tokens": 12, "temperature": 0,
                "ignore_eos": True}
        async with s.post("/v1/completions", json=body) as r:
            full = (await r.json())["choices"][0]["text"]
        words = full.split(" ")
        i = next(k for k in range(1, len(words)) if words[k] not in " ".join(words[:k]))
        body["stop"] = [words[i]]
        async with s.post("/v1/completions", json=body) as r:
            d = await r.json()
        assert d["choices"][0]["text"] == " ".join(words[:i]) + " "
        assert d["choices"][0]["finish_reason"] == "stop"
    run(_with_server(fn))


def test_logprobs_in_chat_and_completions_stream_and_not():
    async def fn(s, eng):
        import math

        body = {"model": "tiny-llama", "messages": [{"role": "user", "content": "tell me"}],
                "max_tokens": 5, "temperature": 0, "logprobs": True, "top_logprobs": 3, "ignore_eos": True}
        async with s.post("/v1/chat/completions", json=body) as r:
            assert r.status == 200
            d = await r.json()
        items = d["choices"][0]["logprobs"]["content"]
        assert len(items) == d["usage"]["completion_tokens"] == 5
        for it in items:
            assert it["logprob"] <= 0 and len(it["top_logprobs"]) == 3
            # greedy: the chosen token is the most likely alternative
            assert math.isclose(it["logprob"], it["top_logprobs"][0]["logprob"], abs_tol=1e-5)
            assert it["bytes"] == list(it["token"].encode())
            assert it["top_logprobs"][0]["logprob"] >= it["top_logprobs"][1]["logprob"]
        # the same request streamed carries the same numbers
        got = []
        async with s.post("/v1/chat/completions", json=dict(body, stream=True)) as r:
            async for line in r.content:
                line = line.decode().strip()
                if line.startswith("data: ") and line != "data: [DONE]":
                    c = json.loads(line[6:])
                    for ch in c.get("choices", []):
                        if ch.get("logprobs"):
                            got.extend(ch["logprobs"]["content"])
        assert [round(x["logprob"], 4) for x in got] == [round(x["logprob"], 4) for x in items]
        # completions shape; sampled request (temperature > 0) reports the model's distribution too
        body = {"model": "tiny-llama", "prompt": "once upon", "max_tokens": 4, "temperature": 0.8, "seed": 7,
                "logprobs": 2, "ignore_eos": True}
        async with s.post("/v1/completions", json=body) as r:
            d = await r.json()
        lp = d["choices"][0]["logprobs"]
        assert len(lp["tokens"]) == len(lp["token_logprobs"]) == len(lp["top_logprobs"]) == len(lp["text_offset"]) == 4
        assert all(v <= 0 for v in lp["token_logprobs"]) and all(len(t) <= 2 for t in lp["top_logprobs"])
        # no logprobs asked: the field stays null
        async with s.post("/v1/completions", json={"model": "tiny-llama", "prompt": "x", "max_tokens": 2}) as r:
            assert (await r.json())["choices"][0]["logprobs"] is None

    run(_with_server(fn))


def test_n_choices_and_prompt_lists():
    async def fn(s, eng):
        # n sampled choices, seeded: distinct per choice, identical when repeated
        body = {"model": "tiny-llama", "messages": [{"role": "user", "content": "pick"}], "max_tokens": 6,
                "temperature": 1.0, "seed": 11, "n": 3, "ignore_eos": True}
        async with s.post("/v1/chat/completions", json=body) as r:
            assert r.status == 200
            d1 = await r.json()
        async with s.post("/v1/chat/completions", json=body) as r:
            d2 = await r.json()
        assert [c["index"] for c in d1["choices"]] == [0, 1, 2]
        texts = [c["message"]["content"] for c in d1["choices"]]
        assert texts == [c["message"]["content"] for c in d2["choices"]]
        assert len(set(texts)) > 1
        assert d1["usage"]["completion_tokens"] == 18 and d1["usage"]["prompt_tokens"] > 0
        # streamed: every chunk is tagged with its choice, each choice finishes once
        finished, seen = [], set()
        async with s.post("/v1/chat/completions", json=dict(body, stream=True,
                                                            stream_options={"include_usage": True})) as r:
            async for line in r.content:
                line = line.decode().strip()
                if line.startswith("data: ") and line != "data: [DONE]":
                    c = json.loads(line[6:])
                    for ch in c["choices"]:
                        seen.add(ch["index"])
                        if ch["finish_reason"]:
                            finished.append(ch["index"])
                    if c.get("usage"):
                        assert c["usage"]["completion_tokens"] == 18
        assert seen == {0, 1, 2} and sorted(finished) == [0, 1, 2]
        # greedy with n > 1 is refused; a list of prompts gives one choice per prompt
        async with s.post("/v1/chat/completions", json=dict(body, temperature=0)) as r:
            assert r.status == 400
        body = {"model": "tiny-llama", "prompt": ["first prompt", "second one"], "max_tokens": 3,
                "temperature": 0, "ignore_eos": True}
        async with s.post("/v1/completions", json=body) as r:
            d = await r.json()
        assert [c["index"] for c in d["choices"]] == [0, 1] and d["usage"]["completion_tokens"] == 6
        single = []
        for p in body["prompt"]:
            async with s.post("/v1/completions", json=dict(body, prompt=p)) as r:
                single.append((await r.json())["choices"][0]["text"])
        assert [c["text"] for c in d["choices"]] == single

    run(_with_server(fn))


def test_a_client_that_hangs_up_mid_stream_is_not_an_error(caplog):
    """A peer closing the connection while tokens are streaming (a load generator at its time limit,
    a router failing over) aborts the request in the engine and leaves no traceback in the log."""
    import logging

    async def fn(s, eng):
        body = {"model": "tiny-llama", "prompt": "a b c", "max_tokens": 400, "temperature": 0,
                "ignore_eos": True, "stream": True}
        r = await s.post("/v1/completions", json=body)
        await r.content.readline()               # generation has started
        r.close()                                 # hang up without reading the rest
        for _ in range(200):                      # the engine drops the request instead of finishing it
            await asyncio.sleep(0.05)
            if not eng.has_work():
                break
        assert not eng.has_work()
        assert eng.scheduler.running == [] and not eng.scheduler.waiting
        async with s.get("/health") as h:         # and keeps serving
            assert h.status == 200

    with caplog.at_level(logging.ERROR):
        run(_with_server(fn))
    assert not [rec for rec in caplog.records if "Error handling request" in rec.getMessage()
                or rec.exc_info], [rec.getMessage() for rec in caplog.records]


def test_waves_of_clients_hanging_up_leave_the_pool_whole():
    """``benchmarks/abort_storm.py`` on the CPU: streaming requests cancelled at random moments
    (before the response headers, while queued, during chunked prefill, while decoding), over a
    small pool with the host tier on.  The tool exits 0 only when no page holds a reference, the
    pool is whole, the engine is healthy and answers a request afterwards; a client that is gone
    before the headers is no error either."""
    import json
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    p = subprocess.run([sys.executable, str(root / "benchmarks" / "abort_storm.py"), "--device", "cpu", "--model",
                        "tiny-llama", "--waves", "3", "--clients", "24", "--prompt-words", "300", "--num-gpu-blocks",
                        "64", "--cpu-offload-gb", "0.01", "--port", "18763"],
                       capture_output=True, text=True, timeout=300, cwd=root)
    assert p.returncode == 0, (p.stdout[-500:], p.stderr[-2000:])
    res = json.loads(p.stdout.strip().splitlines()[-1])
    assert res["cancelled"] > 10 and res["finished"] > 0 and res["errors"] == 0
    assert res["pages_still_referenced"] == 0 and res["pages_free_or_cached"] == 64
    assert "Error handling request" not in p.stderr and "Traceback" not in p.stderr, p.stderr[-2000:]
