"""Tool calling: the three text parsers, tool_choice handling and the response shapes
(reference flow: /root/reference/src/examples/tool_calling_example.py:40-75 -- tools +
tool_choice="auto", reads message.tool_calls[i].function.{name,arguments})."""

import asyncio
import json

import aiohttp

from vllm_production_stack_amd.engine import server as server_mod
from vllm_production_stack_amd.engine.async_engine import AsyncLLMEngine
from vllm_production_stack_amd.engine.config import EngineConfig
from vllm_production_stack_amd.engine.llm_engine import LLMEngine
from vllm_production_stack_amd.engine.server import serve
from vllm_production_stack_amd.engine.tools import parse_tool_calls

TOOLS = [{"type": "function", "function": {"name": "get_weather", "description": "weather of a city",
                                           "parameters": {"type": "object", "properties": {"city": {"type": "string"}}}}}]


def test_llama3_json_parser():
    calls, rest = parse_tool_calls('<|python_tag|>{"name": "get_weather", "parameters": {"city": "Paris"}}', "llama3_json")
    assert rest == "" and len(calls) == 1
    assert calls[0]["type"] == "function" and calls[0]["function"]["name"] == "get_weather"
    assert json.loads(calls[0]["function"]["arguments"]) == {"city": "Paris"}
    assert calls[0]["id"].startswith("chatcmpl-tool-")
    # two calls separated by ';', surrounding prose and a non-call object stay content
    text = 'Sure. {"name": "a", "arguments": {"x": 1}}; {"name": "b", "parameters": {}} done {"k": 2}'
    calls, rest = parse_tool_calls(text, None)
    assert [c["function"]["name"] for c in calls] == ["a", "b"]
    assert "Sure." in rest and '{"k": 2}' in rest
    assert parse_tool_calls("just words { not json", "llama3_json") == ([], "just words { not json")


def test_hermes_and_mistral_parsers():
    text = 'thinking <tool_call>\n{"name": "f", "arguments": {"a": [1, 2]}}\n</tool_call> ok <tool_call>{bad}</tool_call>'
    calls, rest = parse_tool_calls(text, "hermes")
    assert len(calls) == 1 and json.loads(calls[0]["function"]["arguments"]) == {"a": [1, 2]}
    assert "thinking" in rest and "<tool_call>{\"name\"" not in rest
    calls, rest = parse_tool_calls('fine [TOOL_CALLS] [{"name": "g", "arguments": {"q": "x"}}, {"name": "h", "arguments": {}}]', "mistral")
    assert [c["function"]["name"] for c in calls] == ["g", "h"] and rest == "fine"
    assert parse_tool_calls("no calls here", "mistral") == ([], "no calls here")


def test_tool_calls_in_responses(monkeypatch):
    """The tiny random model never writes a call, so the parser is stubbed: everything the model
    says counts as one call.  Checks the non-streaming and streaming shapes and tool_choice=none."""
    def fake_parse(text, parser):
        return ([{"id": "chatcmpl-tool-1", "type": "function",
                  "function": {"name": "get_weather", "arguments": json.dumps({"said": len(text)})}}], "") if text else ([], text)

    monkeypatch.setattr(server_mod, "parse_tool_calls", fake_parse)

    async def go():
        cfg = EngineConfig(model="tiny-llama", device="cpu", dtype="float32", max_model_len=512, max_num_seqs=8,
                           num_gpu_blocks_override=64, enable_auto_tool_choice=True, tool_call_parser="llama3_json")
        ae = AsyncLLMEngine(LLMEngine(cfg))
        ae.start(asyncio.get_running_loop())
        runner = await serve(ae, "127.0.0.1", 0)
        port = runner.addresses[0][1]
        body = {"model": "tiny-llama", "messages": [{"role": "user", "content": "weather in Paris?"}],
                "tools": TOOLS, "tool_choice": "auto", "max_tokens": 6, "temperature": 0, "ignore_eos": True}
        try:
            async with aiohttp.ClientSession(base_url=f"http://127.0.0.1:{port}") as s:
                async with s.post("/v1/chat/completions", json=body) as r:
                    d = await r.json()
                ch = d["choices"][0]
                assert ch["finish_reason"] == "tool_calls" and ch["message"]["content"] is None
                assert ch["message"]["tool_calls"][0]["function"]["name"] == "get_weather"
                # streamed: no content deltas, one tool_calls delta with indices, then the finish chunk
                deltas, finishes = [], []
                async with s.post("/v1/chat/completions", json=dict(body, stream=True)) as r:
                    async for line in r.content:
                        line = line.decode().strip()
                        if line.startswith("data: ") and line != "data: [DONE]":
                            for c in json.loads(line[6:])["choices"]:
                                deltas.append(c["delta"])
                                if c["finish_reason"]:
                                    finishes.append(c["finish_reason"])
                assert finishes == ["tool_calls"]
                assert all(not dl.get("content") for dl in deltas)
                tc = [dl["tool_calls"] for dl in deltas if dl.get("tool_calls")]
                assert len(tc) == 1 and tc[0][0]["index"] == 0 and tc[0][0]["function"]["name"] == "get_weather"
                assert json.loads(tc[0][0]["function"]["arguments"]) == json.loads(
                    ch["message"]["tool_calls"][0]["function"]["arguments"])
                # tool_choice none: plain text answer, nothing parsed
                async with s.post("/v1/chat/completions", json=dict(body, tool_choice="none")) as r:
                    d = await r.json()
                assert d["choices"][0]["finish_reason"] == "length" and "tool_calls" not in d["choices"][0]["message"]
                assert d["choices"][0]["message"]["content"]
        finally:
            await runner.cleanup()
            ae.stop()

    asyncio.run(go())
