#!/usr/bin/env python3
"""Tool calling through the router (engine started with --enable-auto-tool-choice
--tool-call-parser llama3_json): the model answers with a tool call, the client runs the tool and
sends the result back for the final answer."""
import argparse
import json
import urllib.request

TOOLS = [{"type": "function", "function": {
    "name": "get_gpu_memory", "description": "HBM capacity of an accelerator in GB",
    "parameters": {"type": "object", "properties": {"model": {"type": "string"}}, "required": ["model"]}}}]


def get_gpu_memory(model: str) -> str:
    return json.dumps({"model": model, "hbm_gb": {"MI355X": 288, "MI300X": 192}.get(model, 0)})


def chat(base, body):
    req = urllib.request.Request(base + "/v1/chat/completions", data=json.dumps(body).encode(),
                                 headers={"Content-Type": "application/json"})
    with urllib.request.urlopen(req, timeout=300) as r:  # noqa: S310
        return json.loads(r.read())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--base-url", default="http://localhost:30080")
    ap.add_argument("--model", default="llama-3-8b")
    a = ap.parse_args()
    messages = [{"role": "user", "content": "How much HBM does an MI355X have?"}]
    out = chat(a.base_url, {"model": a.model, "messages": messages, "tools": TOOLS, "tool_choice": "auto"})
    msg = out["choices"][0]["message"]
    print("assistant:", json.dumps(msg))
    for call in msg.get("tool_calls") or []:
        args = json.loads(call["function"]["arguments"] or "{}")
        messages += [msg, {"role": "tool", "tool_call_id": call["id"], "content": get_gpu_memory(**args)}]
    if msg.get("tool_calls"):
        out = chat(a.base_url, {"model": a.model, "messages": messages, "tools": TOOLS})
        print("final:", out["choices"][0]["message"]["content"])


if __name__ == "__main__":
    main()
