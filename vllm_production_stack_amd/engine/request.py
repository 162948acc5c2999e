"""Request-side data types of the engine."""

from __future__ import annotations

import enum
import time
from dataclasses import dataclass, field


@dataclass
class SamplingParams:
    max_tokens: int = 16
    temperature: float = 1.0
    top_p: float = 1.0
    top_k: int = -1
    seed: int | None = None
    stop: list[str] = field(default_factory=list)
    stop_token_ids: list[int] = field(default_factory=list)
    ignore_eos: bool = False
    logprobs: int | None = None
    n: int = 1
    presence_penalty: float = 0.0
    frequency_penalty: float = 0.0
    repetition_penalty: float = 1.0
    min_tokens: int = 0

    @property
    def greedy(self) -> bool:
        return self.temperature <= 1e-5


class RequestState(enum.Enum):
    WAITING = "waiting"
    LOADING = "loading"  # KV being fetched from a lower tier / a prefill peer
    RUNNING = "running"
    FINISHED = "finished"


@dataclass(eq=False)  # identity semantics: membership tests must not compare token lists
class Request:
    request_id: str
    prompt_ids: list[int]
    params: SamplingParams
    arrival: float = field(default_factory=time.time)
    lora_name: str | None = None
    lora_slot: int = -1
    # progress
    state: RequestState = RequestState.WAITING
    output_ids: list[int] = field(default_factory=list)
    num_computed: int = 0  # tokens whose KV is resident in this request's pages
    pages: list[int] = field(default_factory=list)
    hashes: list[int] = field(default_factory=list)  # chain hash per full page (prompt+output)
    num_registered: int = 0  # pages already published to the prefix cache
    num_cached_tokens: int = 0  # prefix tokens served from HBM cache at admission
    num_external_tokens: int = 0  # prefix tokens loaded from a lower tier
    finish_reason: str | None = None
    preemptions: int = 0
    pending_stats: tuple | None = None  # (queried, hit) prefix-cache tokens, counted on admission
    # timing
    first_scheduled: float | None = None
    first_token: float | None = None
    last_token: float | None = None
    finished: float | None = None
    # P/D disaggregation
    kv_transfer_params: dict | None = None
    load_event: object | None = None  # device event the first compute step must wait on
    # pooling (embeddings / scoring) requests return the final hidden state instead of text
    pooling: bool = False
    pooled: object | None = None
    # streaming sink (set by the async engine)
    sink: object | None = None
    detok_offset: int = 0
    emitted_text: str = ""

    @property
    def all_ids(self) -> list[int]:
        return self.prompt_ids + self.output_ids

    @property
    def num_tokens(self) -> int:
        return len(self.prompt_ids) + len(self.output_ids)

    @property
    def prompt_len(self) -> int:
        return len(self.prompt_ids)

    @property
    def remaining(self) -> int:
        """Tokens whose KV still has to be computed (the newest token is always among them)."""
        return self.num_tokens - self.num_computed


@dataclass
class StepOutput:
    request_id: str
    new_token_ids: list[int]
    finished: bool
    finish_reason: str | None = None
    num_prompt_tokens: int = 0
    num_output_tokens: int = 0
    num_cached_tokens: int = 0
    queue_time: float = 0.0  # seconds between arrival and the first scheduled step
    logprobs: list | None = None  # per new token: {token_logprob, top_ids, top_logprobs}
