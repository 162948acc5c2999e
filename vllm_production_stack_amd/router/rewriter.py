"""Pluggable request-body rewriting hook (only the no-op is built in).
Reference: /root/reference/src/vllm_router/services/request_service/rewriter.py:29-119."""

from __future__ import annotations

import abc


class RequestRewriter(abc.ABC):
    is_noop = False

    @abc.abstractmethod
    def rewrite_request(self, request_body: str, model: str, endpoint: str) -> str:
        """Return the (possibly modified) JSON body to forward."""


class NoopRequestRewriter(RequestRewriter):
    is_noop = True

    def rewrite_request(self, request_body: str, model: str, endpoint: str) -> str:
        return request_body


_REGISTRY = {"noop": NoopRequestRewriter}


def register_rewriter(name: str, cls) -> None:
    _REGISTRY[name] = cls


def get_request_rewriter(name: str = "noop", **kw) -> RequestRewriter:
    if name not in _REGISTRY:
        raise ValueError(f"Unknown request rewriter: {name}")
    return _REGISTRY[name](**kw)
