"""Continuous-batching scheduler with chunked prefill, prefix caching and preemption.

Every step is a mixed batch: all running sequences that need exactly one more token
(decode rows) plus as many prompt-chunk rows as fit in ``max_num_batched_tokens``.
Prompt chunks piggy-back on the decode batch so the layer weights are streamed from HBM
once per step for both.  With chunked prefill disabled a prompt is only admitted when it
fits the token budget whole (the budget is then at least ``max_model_len``).

Memory pressure: a sequence that cannot get a page for its next token preempts the most
recently admitted running sequence (its pages are released - full ones stay in the prefix
cache - and it re-enters the waiting queue at the front, to be recomputed).
"""

from __future__ import annotations

import time
from collections import deque
from dataclasses import dataclass, field

from .block_manager import PAGE, BlockManager, chain_hashes
from .request import Request, RequestState


@dataclass
class ScheduledSeq:
    req: Request
    start: int  # tokens already in cache
    n: int  # rows this step
    samples: bool  # chunk reaches the end of the known tokens -> sample a new token


@dataclass
class SchedulerOutput:
    decodes: list[ScheduledSeq] = field(default_factory=list)
    prefills: list[ScheduledSeq] = field(default_factory=list)
    preempted: list[Request] = field(default_factory=list)

    @property
    def num_tokens(self) -> int:
        return len(self.decodes) + sum(s.n for s in self.prefills)

    @property
    def empty(self) -> bool:
        return not self.decodes and not self.prefills


class Scheduler:
    def __init__(self, block_manager: BlockManager, max_num_seqs: int, max_num_batched_tokens: int,
                 max_model_len: int, enable_chunked_prefill: bool = True, connector=None):
        self.bm = block_manager
        self.max_num_seqs = max_num_seqs
        self.max_model_len = max_model_len
        self.enable_chunked_prefill = enable_chunked_prefill
        self.max_num_batched_tokens = max_num_batched_tokens if enable_chunked_prefill else max(
            max_num_batched_tokens, max_model_len)
        self.connector = connector
        # Latency-aware admission: ``prefill_cost_ms(n_new_tokens, n_cached_tokens)`` estimates
        # the GPU time of one sequence's prefill rows.  The first prefill of a step is always
        # taken (up to the token budget); further waiting requests join only while the summed
        # estimate stays under ``prefill_step_ms``.  A burst of short follow-up turns is then
        # served in several short steps (most users see their first token after one of them)
        # instead of one long step that everybody waits for.  A step whose first prefill alone
        # exceeds the target is filled to the token budget as usual.  None / 0 = token budget only.
        self.prefill_cost_ms = None
        self.prefill_step_ms = 0.0
        self.waiting: deque[Request] = deque()
        self.running: list[Request] = []
        self.num_preemptions = 0

    # ------------------------------------------------------------------ queue ops
    def add(self, req: Request) -> None:
        self.waiting.append(req)

    def abort(self, request_id: str) -> Request | None:
        for q in (self.waiting, self.running):
            for r in list(q):
                if r.request_id == request_id:
                    q.remove(r)
                    self._release(r)
                    r.state = RequestState.FINISHED
                    r.finish_reason = "abort"
                    return r
        return None

    def has_work(self) -> bool:
        return bool(self.waiting or self.running)

    # ------------------------------------------------------------------ helpers
    def _salt(self, req: Request) -> bytes:
        return req.lora_name.encode() if req.lora_name else b""

    def _extend_hashes(self, req: Request) -> None:
        """Keep ``req.hashes`` covering every full page of prompt+output."""
        n_full = req.num_tokens // PAGE
        if len(req.hashes) < n_full:
            prev = req.hashes[-1] if req.hashes else 0
            req.hashes.extend(chain_hashes(req.all_ids, self._salt(req), prev, len(req.hashes)))

    def _release(self, req: Request) -> None:
        if req.load_event is not None:
            # a reload from the lower tiers may still be writing these pages (an abort between
            # admission and the first step): the compute stream waits for it, so whoever is given
            # the pages next writes after the copy, not under it
            req.load_event.wait()
            req.load_event = None
        if req.pages:
            self.bm.free(req.pages)
        req.pages = []
        req.num_computed = 0
        req.num_registered = 0

    def _ensure_pages(self, req: Request, upto_tokens: int) -> bool:
        need = (upto_tokens + PAGE - 1) // PAGE - len(req.pages)
        if need <= 0:
            return True
        if not self.bm.can_allocate(need):
            return False
        req.pages.extend(self.bm.allocate(need))
        return True

    def _preempt_one(self, keep: Request, out: SchedulerOutput) -> bool:
        for victim in reversed(self.running):
            if victim is keep:
                continue
            self.running.remove(victim)
            self._release(victim)
            # a victim already placed in this step's batch is taken out again
            out.decodes = [s for s in out.decodes if s.req is not victim]
            out.prefills = [s for s in out.prefills if s.req is not victim]
            victim.state = RequestState.WAITING
            victim.preemptions += 1
            self.num_preemptions += 1
            self.waiting.appendleft(victim)
            out.preempted.append(victim)
            return True
        return False

    def _admit(self, req: Request) -> bool:
        """Give a waiting request its cached prefix pages (HBM first, then lower tiers)."""
        self._extend_hashes(req)
        # at least one token must be computed so that the step yields logits
        max_cached_pages = (req.num_tokens - 1) // PAGE
        hashes = req.hashes[:max_cached_pages]
        pages = self.bm.match_prefix(hashes, count_stats=False)
        req.pages = list(pages)
        req.pending_stats = (len(hashes) * PAGE, len(pages) * PAGE)
        req.num_computed = len(pages) * PAGE
        req.num_registered = len(pages)
        req.num_cached_tokens = req.num_computed
        req.num_external_tokens = 0
        if self.connector is not None and len(pages) < len(hashes):
            rest = hashes[len(pages):]
            n_ext = self.connector.lookup(rest)
            if n_ext > 0 and self.bm.can_allocate(n_ext):
                new_pages = self.bm.allocate(n_ext)
                # a page that left the lower tiers after lookup() saw it is a cache miss: keep the
                # leading pages that did arrive, give the rest back, recompute from there
                try:
                    req.load_event, n_ok = self.connector.load_pages(rest[:n_ext], new_pages)
                except Exception:
                    req.load_event, n_ok = None, 0
                if n_ok < n_ext:
                    self.bm.free(new_pages[n_ok:])
                    new_pages = new_pages[:n_ok]
                    n_ext = n_ok
                for h, p in zip(rest[:n_ext], new_pages):
                    self.bm.register(p, h)
                req.pages.extend(new_pages)
                req.num_computed += n_ext * PAGE
                req.num_registered += n_ext
                req.num_external_tokens = n_ext * PAGE
        return True

    # A prefill replica of a disaggregated pair pushes a chunk's pages while the next chunk
    # computes, so only the LAST chunk's pages travel after the prefill has ended and add to the
    # time to first token.  A prompt's last chunk is therefore kept this short for requests whose
    # pages are pushed (16 pages of Llama-3-8B: 128 MiB, 2-3 ms on a link).
    PUSH_TAIL_TOKENS = 1024

    def _chunk_rows(self, req: Request, rem: int, budget: int) -> int:
        """Rows of the prompt chunk to run now."""
        n = min(rem, budget)
        if (n == rem and rem > self.PUSH_TAIL_TOKENS + PAGE and self.enable_chunked_prefill
                and self.connector is not None and (req.kv_transfer_params or {}).get("do_remote_decode")
                and req.num_computed + rem <= req.prompt_len + 1):
            cut = (req.num_computed + rem - self.PUSH_TAIL_TOKENS) // PAGE * PAGE - req.num_computed
            if cut > 0:
                n = cut
        return n

    # ------------------------------------------------------------------ main entry
    def schedule(self) -> SchedulerOutput:
        out = SchedulerOutput()
        budget = self.max_num_batched_tokens
        now = time.time()

        # 1. sequences already running: decode rows first, then unfinished prompt chunks
        for req in sorted(self.running, key=lambda r: r.remaining != 1):
            if out.preempted and any(v is req for v in out.preempted):
                continue  # preempted by an earlier iteration
            rem = req.remaining
            n = 1 if rem == 1 else self._chunk_rows(req, rem, budget)
            if n <= 0:
                continue
            while not self._ensure_pages(req, req.num_computed + n):
                if not self._preempt_one(req, out):
                    break
            if (req.num_computed + n + PAGE - 1) // PAGE > len(req.pages):
                # could not make room even after preempting everyone else: requeue self
                self.running.remove(req)
                self._release(req)
                req.state = RequestState.WAITING
                self.waiting.appendleft(req)
                out.preempted.append(req)
                continue
            seq = ScheduledSeq(req, req.num_computed, n, req.num_computed + n == req.num_tokens)
            (out.decodes if n == 1 and req.num_computed > 0 else out.prefills).append(seq)
            budget -= n

        timed = self.prefill_cost_ms is not None and self.prefill_step_ms > 0
        spent_ms = sum(self.prefill_cost_ms(s.n, s.start) for s in out.prefills) if timed else 0.0

        # 2. admit waiting requests FCFS
        while self.waiting and budget > 0 and len(self.running) < self.max_num_seqs:
            req = self.waiting[0]
            if req in out.preempted:
                break  # do not re-admit in the step that preempted it
            if not req.pages:
                self._admit(req)
            rem = req.remaining
            n = self._chunk_rows(req, rem, budget)
            if timed:
                cost = self.prefill_cost_ms(n, req.num_computed)
                # A step that is already long (a chunk of a long prompt) is filled up to the
                # token budget as before: the rule only keeps short steps short.
                if out.prefills and spent_ms <= self.prefill_step_ms < spent_ms + cost:
                    break  # next step; its cached pages stay attached (the request is next in line)
                spent_ms += cost
            if not self.enable_chunked_prefill and n < rem:
                if not out.prefills and not out.decodes and rem > self.max_num_batched_tokens:
                    n = rem  # oversize prompt running alone
                else:
                    self._unadmit(req)
                    break
            if not self._ensure_pages(req, req.num_computed + n):
                # not enough memory now; give the cached pages back and wait
                self._unadmit(req)
                break
            self.waiting.popleft()
            req.state = RequestState.RUNNING
            if req.pending_stats is not None:
                self.bm.queries += req.pending_stats[0]
                self.bm.hits += req.pending_stats[1]
                req.pending_stats = None
            if req.first_scheduled is None:
                req.first_scheduled = now
            self.running.append(req)
            seq = ScheduledSeq(req, req.num_computed, n, req.num_computed + n == req.num_tokens)
            (out.decodes if n == 1 and req.num_computed > 0 else out.prefills).append(seq)
            budget -= n
        return out

    def _unadmit(self, req: Request) -> None:
        if req.load_event is None:
            self._release(req)
        # with a load in flight the pages stay attached (they are being written)

    # ------------------------------------------------------------------ post-step
    def commit(self, seq: ScheduledSeq) -> list[tuple[int, int]]:
        """Advance a sequence after its rows ran; returns newly full (hash, page) pairs."""
        req = seq.req
        req.num_computed = seq.start + seq.n
        if req.num_computed // PAGE <= req.num_registered:
            return []  # no page filled up in this step (63 of 64 decode steps)
        self._extend_hashes(req)
        n_full = min(req.num_computed // PAGE, len(req.hashes))
        fresh = []
        for i in range(req.num_registered, n_full):
            self.bm.register(req.pages[i], req.hashes[i])
            fresh.append((req.hashes[i], req.pages[i]))
        req.num_registered = max(req.num_registered, n_full)
        return fresh

    def finish(self, req: Request, reason: str) -> None:
        if req in self.running:
            self.running.remove(req)
        req.state = RequestState.FINISHED
        req.finish_reason = reason
        req.finished = time.time()
        if req.pages:
            self.bm.free(req.pages)
            req.pages = []
