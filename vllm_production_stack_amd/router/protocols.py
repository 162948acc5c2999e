"""Wire models of the router's own responses (pydantic; unknown request fields are logged,
not rejected).  Reference: /root/reference/src/vllm_router/protocols.py:11-56."""

from __future__ import annotations

import time
from typing import List, Optional

from pydantic import BaseModel, ConfigDict, Field, model_validator

from .log import init_logger

logger = init_logger(__name__)


class OpenAIBaseModel(BaseModel):
    model_config = ConfigDict(extra="allow")

    @model_validator(mode="before")
    @classmethod
    def _warn_extra(cls, data):
        if isinstance(data, dict):
            known = set()
            for name, f in cls.model_fields.items():
                known.add(name)
                if f.alias:
                    known.add(f.alias)
            extra = set(data) - known
            if extra:
                logger.warning("The following fields were present in the request but ignored: %s", extra)
        return data


class ErrorResponse(OpenAIBaseModel):
    object: str = "error"
    message: str
    type: str
    param: Optional[str] = None
    code: int


class ModelCard(OpenAIBaseModel):
    id: str
    object: str = "model"
    created: int = Field(default_factory=lambda: int(time.time()))
    owned_by: str = "vllm"
    root: Optional[str] = None
    parent: Optional[str] = None


class ModelList(OpenAIBaseModel):
    object: str = "list"
    data: List[ModelCard] = Field(default_factory=list)
