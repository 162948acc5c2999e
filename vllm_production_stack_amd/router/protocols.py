"""Shapes of the router's own JSON answers, in one place: the model list of ``GET /v1/models``, the
OpenAI-style error body, and the check that names request fields the router does not know.

Plain dict builders (the router is an aiohttp application and answers with ``web.json_response``).
Same wire shapes as the pydantic models of /root/reference/src/vllm_router/protocols.py:39-56
({id, object, created, owned_by, root, parent}; {object: "error", message, type, param, code});
unknown request fields are logged, never rejected (protocols.py:15-36 there)."""

from __future__ import annotations

import time
from typing import Iterable

from .log import init_logger

logger = init_logger(__name__)


def model_card(model_id: str, created: int | float | None = None, owned_by: str = "vllm",
               root: str | None = None, parent: str | None = None) -> dict:
    return {"id": model_id, "object": "model", "created": int(time.time() if created is None else created),
            "owned_by": owned_by, "root": root, "parent": parent}


def model_list(cards: Iterable[dict]) -> dict:
    return {"object": "list", "data": list(cards)}


def error_body(message: str, err_type: str, code: int, param: str | None = None) -> dict:
    return {"object": "error", "message": message, "type": err_type, "param": param, "code": int(code)}


def unknown_fields(body: dict, known: Iterable[str]) -> set[str]:
    """Fields of a request body outside ``known``; they are passed on to the engine untouched and
    logged once per request, so that a typo in a client shows up in the router log."""
    extra = set(body) - set(known) if isinstance(body, dict) else set()
    if extra:
        logger.warning("request fields the router does not interpret (forwarded as they are): %s", sorted(extra))
    return extra
