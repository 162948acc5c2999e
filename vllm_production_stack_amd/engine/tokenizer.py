"""Tokenizers: a HF ``tokenizers`` wrapper when the checkpoint directory ships one, and a
deterministic synthetic word-level tokenizer for random-init serving (no files, no network).

The synthetic tokenizer maps one whitespace-delimited word to one id, so the benchmark's
"N repeated words" prompts are exactly N tokens, and ``decode`` followed by ``encode``
reproduces the same ids - generated answers re-sent as chat history therefore hit the
prefix cache, exactly like a real tokenizer's round trip.
"""

from __future__ import annotations

import json
import re
from collections import OrderedDict
from pathlib import Path

import xxhash

_SPECIALS = ["<|begin_of_text|>", "<|end_of_text|>", "<|start_header_id|>", "<|end_header_id|>",
             "<|eot_id|>", "<|pad|>", "<|python_tag|>", "<|eom_id|>"]


_TEMPLATES: dict[str, object] = {}


def _compile_template(source: str):
    """Chat templates are written against the environment Hugging Face renders them in: sandboxed,
    ``trim_blocks`` + ``lstrip_blocks``, loop controls, ``raise_exception``, ``strftime_now`` and a
    ``tojson`` that keeps non-ASCII text.  Whitespace handling is part of the prompt, so the
    environment has to match exactly."""
    tpl = _TEMPLATES.get(source)
    if tpl is None:
        import datetime

        import jinja2
        import jinja2.ext
        import jinja2.sandbox

        env = jinja2.sandbox.ImmutableSandboxedEnvironment(
            trim_blocks=True, lstrip_blocks=True, extensions=[jinja2.ext.loopcontrols])

        def raise_exception(message):
            raise jinja2.exceptions.TemplateError(message)

        def tojson(x, ensure_ascii=False, indent=None, separators=None, sort_keys=False):
            return json.dumps(x, ensure_ascii=ensure_ascii, indent=indent, separators=separators,
                              sort_keys=sort_keys)

        env.filters["tojson"] = tojson
        env.globals["raise_exception"] = raise_exception
        env.globals["strftime_now"] = lambda fmt: datetime.datetime.now().strftime(fmt)
        tpl = env.from_string(source)
        _TEMPLATES[source] = tpl
    return tpl


class BaseTokenizer:
    eos_token_id: int = 2
    bos_token_id: int | None = 1
    vocab_size: int = 0
    chat_template: str | None = None
    # special-token strings handed to the chat template (a checkpoint's tokenizer overrides them)
    bos_token: str = "<|begin_of_text|>"
    eos_token: str = "<|eot_id|>"

    def encode(self, text: str, add_special_tokens: bool = True) -> list[int]:
        raise NotImplementedError

    def decode(self, ids: list[int], skip_special_tokens: bool = True) -> str:
        raise NotImplementedError

    # -- chat -------------------------------------------------------------------
    @staticmethod
    def _content_text(content) -> str:
        if isinstance(content, str):
            return content
        if isinstance(content, list):
            return "\n".join(p.get("text", "") for p in content
                             if isinstance(p, dict) and p.get("type", "text") == "text")
        return "" if content is None else str(content)

    def render_chat(self, messages: list[dict], add_generation_prompt: bool = True,
                    tools: list | None = None) -> str:
        if self.chat_template:
            return _compile_template(self.chat_template).render(
                messages=messages, add_generation_prompt=add_generation_prompt, tools=tools,
                bos_token=self.bos_token, eos_token=self.eos_token)
        out = ["<|begin_of_text|>"]
        if tools:
            out.append("<|start_header_id|> system <|end_header_id|> "
                       "You may call tools by answering with a JSON object "
                       '{"name": ..., "parameters": ...}. Tools: ' + json.dumps(tools) + " <|eot_id|>")
        for m in messages:
            text = self._content_text(m.get("content"))
            if m.get("tool_calls"):
                text += " " + json.dumps(m["tool_calls"])
            out.append(f"<|start_header_id|> {m.get('role', 'user')} <|end_header_id|> {text} <|eot_id|>")
        if add_generation_prompt:
            out.append("<|start_header_id|> assistant <|end_header_id|>")
        return " ".join(out)

    def encode_chat(self, messages: list[dict], add_generation_prompt: bool = True,
                    tools: list | None = None) -> list[int]:
        return self.encode(self.render_chat(messages, add_generation_prompt, tools),
                           add_special_tokens=False)


class SyntheticTokenizer(BaseTokenizer):
    """Word-level hashing tokenizer.  ids [0, 16) are special tokens."""

    _WORD = re.compile(r"<\|[a-z_]+\|>|\S+")

    def __init__(self, vocab_size: int, eos_token_id: int | None = None):
        self.vocab_size = vocab_size
        self._special = {s: i for i, s in enumerate(_SPECIALS)}
        self._special_rev = {i: s for s, i in self._special.items()}
        self.bos_token_id = self._special["<|begin_of_text|>"]
        self.eos_token_id = self._special["<|eot_id|>"]
        self._base = 16
        self._space = max(1, vocab_size - self._base)
        self._words: dict[int, str] = {}
        self._memo: dict[str, int] = {}
        # per-message id cache for chat prompts: a conversation re-sends its whole history every
        # turn, and word-level tokenisation of space-joined segments is the concatenation of the
        # segments' ids, so only the new messages of a turn are tokenised
        self._seg_cache: "OrderedDict[tuple, list[int]]" = OrderedDict()
        self._seg_cached_ids = 0
        self._seg_cache_limit = 8_000_000  # ids

    def _word_id(self, w: str) -> int:
        if w in self._special:
            return self._special[w]
        if w[0] == "t" and w[1:].isdigit():
            v = int(w[1:])
            if self._base <= v < self.vocab_size:
                return v
        i = self._base + xxhash.xxh32_intdigest(w.encode()) % self._space
        self._words.setdefault(i, w)
        return i

    def encode(self, text: str, add_special_tokens: bool = True) -> list[int]:
        # str.split + a memo table handles the common case at C speed; the regex is only needed
        # when a special token is glued to other characters
        words = text.split()
        memo = self._memo
        try:
            ids = [memo[w] for w in words]
        except KeyError:
            ids = []
            for w in words:
                i = memo.get(w)
                if i is None:
                    if "<|" in w and w not in self._special:
                        sub = [self._word_id(x) for x in self._WORD.findall(w)]
                        ids.extend(sub)
                        continue
                    if len(memo) >= 2_000_000:  # arbitrary text never repeats its words: start over
                        memo.clear()
                    i = memo[w] = self._word_id(w)
                ids.append(i)
        if add_special_tokens and (not ids or ids[0] != self.bos_token_id):
            ids.insert(0, self.bos_token_id)
        return ids

    def _segment_ids(self, seg: str) -> list[int]:
        if len(seg) < 2048:
            return self.encode(seg, add_special_tokens=False)
        key = (len(seg), xxhash.xxh64_intdigest(seg))
        cache = self._seg_cache
        ids = cache.get(key)
        if ids is not None:
            cache.move_to_end(key)
            return ids
        ids = self.encode(seg, add_special_tokens=False)
        cache[key] = ids
        self._seg_cached_ids += len(ids)
        while self._seg_cached_ids > self._seg_cache_limit and len(cache) > 1:
            _, old = cache.popitem(last=False)
            self._seg_cached_ids -= len(old)
        return ids

    def encode_chat(self, messages: list[dict], add_generation_prompt: bool = True,
                    tools: list | None = None) -> list[int]:
        if self.chat_template or tools:
            return super().encode_chat(messages, add_generation_prompt, tools)
        # same text as render_chat's built-in template, tokenised segment by segment
        out = [self.bos_token_id]
        for m in messages:
            text = self._content_text(m.get("content"))
            if m.get("tool_calls"):
                text += " " + json.dumps(m["tool_calls"])
            out.extend(self._segment_ids(
                f"<|start_header_id|> {m.get('role', 'user')} <|end_header_id|> {text} <|eot_id|>"))
        if add_generation_prompt:
            out.extend(self.encode("<|start_header_id|> assistant <|end_header_id|>", add_special_tokens=False))
        return out

    def decode(self, ids: list[int], skip_special_tokens: bool = True) -> str:
        words = []
        for i in ids:
            if i in self._special_rev:
                if not skip_special_tokens:
                    words.append(self._special_rev[i])
                continue
            w = self._words.get(i)
            if w is None or self._word_id(w) != i:
                w = f"t{i}"
            words.append(w)
        return " ".join(words)


class HFTokenizer(BaseTokenizer):
    def __init__(self, path: Path):
        from tokenizers import Tokenizer

        self._tok = Tokenizer.from_file(str(path / "tokenizer.json"))
        self.vocab_size = self._tok.get_vocab_size()
        cfg = {}
        if (path / "tokenizer_config.json").exists():
            with open(path / "tokenizer_config.json") as f:
                cfg = json.load(f)
        tpl = cfg.get("chat_template")
        if isinstance(tpl, list):  # named templates: [{"name": ..., "template": ...}]
            named = {t.get("name"): t.get("template") for t in tpl if isinstance(t, dict)}
            tpl = named.get("default") or next(iter(named.values()), None)
        if (path / "chat_template.jinja").exists():  # newer checkpoints keep it in its own file
            tpl = (path / "chat_template.jinja").read_text()
        self.chat_template = tpl

        def tok_str(key):
            t = cfg.get(key)
            if isinstance(t, dict):
                t = t.get("content")
            return t if isinstance(t, str) else None

        def tok_id(key, default):
            t = tok_str(key)
            if t is not None:
                i = self._tok.token_to_id(t)
                if i is not None:
                    return i
            return default

        self.bos_token = tok_str("bos_token") or ""
        self.eos_token = tok_str("eos_token") or ""
        self.bos_token_id = tok_id("bos_token", None)
        self.eos_token_id = tok_id("eos_token", 2)
        # every id that ends generation (generation_config may list several, e.g. end-of-text
        # and end-of-turn); eos_token_id stays the tokenizer's own when it has one
        self.eos_token_ids: set[int] = {self.eos_token_id}
        gen = path / "generation_config.json"
        if gen.exists():
            with open(gen) as f:
                e = json.load(f).get("eos_token_id")
            ids = e if isinstance(e, list) else [e]
            ids = [i for i in ids if isinstance(i, int)]
            self.eos_token_ids.update(ids)
            if ids and tok_str("eos_token") is None:
                self.eos_token_id = ids[0]

    def encode(self, text: str, add_special_tokens: bool = True) -> list[int]:
        return self._tok.encode(text, add_special_tokens=add_special_tokens).ids

    def decode(self, ids: list[int], skip_special_tokens: bool = True) -> str:
        return self._tok.decode(ids, skip_special_tokens=skip_special_tokens)


def load_tokenizer(ckpt_dir: Path | None, vocab_size: int, chat_template_file: str | None = None):
    tok: BaseTokenizer
    if ckpt_dir is not None and (Path(ckpt_dir) / "tokenizer.json").exists():
        tok = HFTokenizer(Path(ckpt_dir))
    else:
        tok = SyntheticTokenizer(vocab_size)
    if chat_template_file:
        tok.chat_template = Path(chat_template_file).read_text()
    return tok


class IncrementalDetokenizer:
    """Streams text for a growing id list without re-decoding the whole sequence."""

    def __init__(self, tokenizer: BaseTokenizer):
        self.tok = tokenizer
        self.ids: list[int] = []
        self.prefix = 0
        self.read = 0
        self.text = ""

    def push(self, new_ids: list[int]) -> str:
        self.ids.extend(new_ids)
        if isinstance(self.tok, SyntheticTokenizer):
            piece = self.tok.decode(new_ids)
            if not piece:
                return ""
            delta = (" " if self.text else "") + piece
            self.text += delta
            return delta
        prefix_text = self.tok.decode(self.ids[self.prefix:self.read])
        full = self.tok.decode(self.ids[self.prefix:])
        if len(full) > len(prefix_text) and not full.endswith("�"):
            delta = full[len(prefix_text):]
            self.prefix = self.read
            self.read = len(self.ids)
            self.text += delta
            return delta
        return ""
