"""Tool-call extraction from generated text (--enable-auto-tool-choice --tool-call-parser X).

Parsers: ``llama3_json`` (a bare JSON object {"name", "parameters"|"arguments"}, optionally
after <|python_tag|>), ``hermes`` (<tool_call>{...}</tool_call> blocks), ``mistral``
([TOOL_CALLS] [ {...}, ... ])."""

from __future__ import annotations

import json
import re
import uuid


def _mk(name: str, args) -> dict:
    if not isinstance(args, str):
        args = json.dumps(args)
    return {"id": "chatcmpl-tool-" + uuid.uuid4().hex[:24], "type": "function",
            "function": {"name": name, "arguments": args}}


def _from_obj(o) -> dict | None:
    if isinstance(o, dict) and "name" in o:
        return _mk(o["name"], o.get("parameters", o.get("arguments", {})))
    return None


def parse_tool_calls(text: str, parser: str | None) -> tuple[list[dict], str]:
    """-> (tool_calls, remaining content)."""
    parser = (parser or "llama3_json").lower()
    calls: list[dict] = []
    if parser == "hermes":
        rest = text
        for m in re.finditer(r"<tool_call>\s*(.*?)\s*</tool_call>", text, re.S):
            try:
                c = _from_obj(json.loads(m.group(1)))
            except json.JSONDecodeError:
                c = None
            if c:
                calls.append(c)
                rest = rest.replace(m.group(0), "")
        return calls, rest.strip()
    if parser == "mistral":
        i = text.find("[TOOL_CALLS]")
        if i >= 0:
            try:
                arr = json.loads(text[i + len("[TOOL_CALLS]"):].strip())
                for o in arr if isinstance(arr, list) else [arr]:
                    c = _from_obj(o)
                    if c:
                        calls.append(c)
            except json.JSONDecodeError:
                pass
            return calls, text[:i].strip()
        return [], text
    # llama3_json / pythonic-json default
    body = text.replace("<|python_tag|>", "").strip()
    dec = json.JSONDecoder()
    pos = 0
    rest_parts = []
    while pos < len(body):
        j = body.find("{", pos)
        if j < 0:
            rest_parts.append(body[pos:])
            break
        rest_parts.append(body[pos:j])
        try:
            obj, end = dec.raw_decode(body, j)
        except json.JSONDecodeError:
            rest_parts.append(body[j])
            pos = j + 1
            continue
        c = _from_obj(obj)
        if c:
            calls.append(c)
        else:
            rest_parts.append(body[j:end])
        pos = end
        while pos < len(body) and body[pos] in "; \n":
            pos += 1
    return calls, "".join(rest_parts).strip()
