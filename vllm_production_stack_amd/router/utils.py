"""Router helpers: model-type probes, URL validation, fd limit, argument parsing helpers.

Behavioural reference: /root/reference/src/vllm_router/utils.py:18-223."""

from __future__ import annotations

import enum
import io
import json
import re
import resource
import struct
import wave

from .log import init_logger

logger = init_logger(__name__)


def silent_wav(seconds: float = 0.1, rate: int = 16000) -> bytes:
    """A short silent PCM16 WAV used to probe transcription backends."""
    buf = io.BytesIO()
    with wave.open(buf, "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(rate)
        n = int(seconds * rate)
        w.writeframes(struct.pack(f"<{n}h", *([0] * n)))
    return buf.getvalue()


class ModelType(enum.Enum):
    chat = "/v1/chat/completions"
    completion = "/v1/completions"
    embeddings = "/v1/embeddings"
    rerank = "/v1/rerank"
    score = "/v1/score"
    transcription = "/v1/audio/transcriptions"

    @staticmethod
    def get_url(model_type: str) -> str:
        return ModelType[model_type].value

    @staticmethod
    def get_test_payload(model_type: str) -> dict:
        mt = ModelType[model_type]
        if mt is ModelType.chat:
            return {"messages": [{"role": "user", "content": "Hello"}], "temperature": 0.0,
                    "max_tokens": 3, "max_completion_tokens": 3}
        if mt is ModelType.completion:
            return {"prompt": "Hello", "max_tokens": 3}
        if mt is ModelType.embeddings:
            return {"input": "Hello"}
        if mt is ModelType.rerank:
            return {"query": "Hello", "documents": ["Test"]}
        if mt is ModelType.score:
            return {"encoding_format": "float", "text_1": "Test", "text_2": "Test2"}
        return {"file": ("empty.wav", silent_wav(), "audio/wav")}

    @staticmethod
    def get_all_fields() -> list[str]:
        return [m.name for m in ModelType]


_URL = re.compile(r"^(http|https)://(([a-zA-Z0-9_-]+\.)*[a-zA-Z0-9_-]+|\[[0-9a-fA-F:]+\])(:\d{1,5})?(/.*)?$")


def validate_url(url: str) -> bool:
    return bool(_URL.match(url or ""))


def set_ulimit(target: int = 65535) -> None:
    try:
        soft, hard = resource.getrlimit(resource.RLIMIT_NOFILE)
        if soft < target:
            resource.setrlimit(resource.RLIMIT_NOFILE, (min(target, hard) if hard > 0 else target, hard))
    except (ValueError, OSError) as e:
        logger.warning("could not raise RLIMIT_NOFILE (%s); many concurrent requests may fail", e)


def parse_comma_separated_args(s: str | None) -> list[str] | None:
    if s is None:
        return None
    return [x.strip() for x in s.split(",") if x.strip() != ""] if s else []


def parse_static_urls(s: str) -> list[str]:
    out = []
    for u in parse_comma_separated_args(s) or []:
        if validate_url(u):
            out.append(u)
        else:
            logger.warning("Skipping invalid URL: %s", u)
    return out


def parse_static_aliases(s: str) -> dict[str, str]:
    out = {}
    for pair in parse_comma_separated_args(s) or []:
        alias, _, model = pair.partition(":")
        out[alias] = model
    return out


def replace_model_in_request_body(body: dict, model: str) -> dict:
    body = dict(body)
    body["model"] = model
    return body


def update_content_length(headers: dict, body: bytes) -> dict:
    headers = {k: v for k, v in headers.items() if k.lower() != "content-length"}
    headers["Content-Length"] = str(len(body))
    return headers


def dumps(obj) -> bytes:
    return json.dumps(obj, separators=(",", ":")).encode()
