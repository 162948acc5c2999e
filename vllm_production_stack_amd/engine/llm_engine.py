"""The engine core: request intake, the step loop body, stop handling, statistics.

``LLMEngine.step()`` = schedule -> execute on the GPU -> commit tokens / publish pages /
finish sequences.  ``AsyncLLMEngine`` (async_engine.py) drives it from a dedicated thread
and fans results out to asyncio consumers; tests and the benchmark can also call ``step``
directly.
"""

from __future__ import annotations

import logging
import os
import sys
import time
import uuid

import torch

from .block_manager import PAGE, BlockManager
from .config import EngineConfig, resolve_model
from .metrics import EngineMetrics
from .model_runner import ModelRunner
from .request import Request, RequestState, SamplingParams, StepOutput
from .scheduler import ScheduledSeq, Scheduler
from .tokenizer import load_tokenizer

log = logging.getLogger("vpsa.engine")
_TRACE = bool(os.environ.get("VPSA_TRACE_STEPS"))


class EngineBusyError(RuntimeError):
    """An operation that needs an idle engine was asked for while requests are in flight."""


class LLMEngine:
    def __init__(self, cfg: EngineConfig, device: torch.device | None = None):
        self.cfg = cfg
        self.model_cfg, self.ckpt_dir = resolve_model(cfg.model)
        self.device = device or cfg.resolve_device()
        self.runner = ModelRunner(cfg, self.model_cfg, self.ckpt_dir, self.device)
        self.max_model_len = self.runner.max_model_len
        self.tokenizer = load_tokenizer(self.ckpt_dir, self.model_cfg.vocab_size, cfg.chat_template)
        self.connector = None
        # transfers publish pages through the prefix cache, so a P/D replica always reuses in HBM
        self.bm = BlockManager(self.runner.num_pages, cfg.enable_prefix_caching or cfg.kv_offload.enabled,
                               reuse_in_hbm=cfg.enable_prefix_caching or cfg.kv_transfer is not None)
        self.scheduler = Scheduler(self.bm, cfg.max_num_seqs, cfg.max_num_batched_tokens,
                                   self.max_model_len, cfg.enable_chunked_prefill)
        if self.device.type == "cuda" and cfg.prefill_step_ms > 0:
            gemm_f, attn_f = self.model_cfg.prefill_flops()
            tp = max(1, cfg.tensor_parallel_size)
            g_ms = gemm_f / tp / (cfg.gemm_tflops_estimate * 1e9)
            a_ms = attn_f / tp / (cfg.attn_tflops_estimate * 1e9)
            self.scheduler.prefill_cost_ms = lambda n, start: n * g_ms + n * (start + 0.5 * n) * a_ms
            self.scheduler.prefill_step_ms = cfg.prefill_step_ms
        self.metrics = EngineMetrics(cfg.model_name)
        self.requests: dict[str, Request] = {}
        self.eos_ids = {self.tokenizer.eos_token_id, self.model_cfg.eos_token_id} if self.ckpt_dir else {
            self.tokenizer.eos_token_id}
        self.eos_ids |= set(getattr(self.tokenizer, "eos_token_ids", ()))
        if self.ckpt_dir:
            self.eos_ids |= set(self.model_cfg.extra_eos_token_ids)
        self.lora_manager = None
        self.step_count = 0
        # coarse step-loop profile (seconds), read by /debug/stats and the benchmark
        self.prof = {"steps": 0, "decode_only_steps": 0, "mixed_steps": 0, "t_schedule": 0.0,
                     "t_execute": 0.0, "t_post": 0.0, "rows": 0}
        self._init_kv_tiers()
        if cfg.capture_graphs_at_start:
            # decode graphs for every batch bucket now, not on the first step that needs one
            t0 = time.time()
            n = self.runner.capture_all_graphs()
            if n:
                log.info("captured %d decode graphs in %.1f s", n, time.time() - t0)
        log.info("engine ready: %d KV pages (%d tokens, %.1f GiB), max_model_len=%d",
                 self.runner.num_pages, self.runner.num_pages * PAGE,
                 self.runner.num_pages * self.runner.page_bytes / 2**30, self.max_model_len)

    def _init_kv_tiers(self) -> None:
        off = self.cfg.kv_offload
        kvt = self.cfg.kv_transfer
        if not (off.enabled or off.enable_pd or off.enable_controller or (kvt and kvt.kv_connector)):
            return
        if self.runner.distributed:
            # the pool of a tp/pp replica is sharded by KV head and by layer across processes;
            # the tiers move whole pages of one process's pool
            log.warning("KV offload / transfer tiers are disabled for tensor/pipeline-parallel replicas")
            return
        from ..kv.connector import KVConnector

        self.connector = KVConnector(self, off, kvt)
        self.scheduler.connector = self.connector
        self.runner.pool_listeners.append(self.connector.set_pool)
        self.bm.on_evict = self.connector.on_evict

    # ------------------------------------------------------------------ intake
    def add_request(self, prompt_ids: list[int], params: SamplingParams,
                    request_id: str | None = None, lora_name: str | None = None,
                    kv_transfer_params: dict | None = None, pooling: bool = False) -> Request:
        rid = request_id or uuid.uuid4().hex
        if len(prompt_ids) == 0:
            prompt_ids = [self.tokenizer.bos_token_id or 0]
        if len(prompt_ids) >= self.max_model_len:
            raise ValueError(
                f"prompt has {len(prompt_ids)} tokens; the context limit is {self.max_model_len}")
        params.max_tokens = max(1, min(params.max_tokens, self.max_model_len - len(prompt_ids)))
        req = Request(rid, list(prompt_ids), params, lora_name=lora_name,
                      kv_transfer_params=kv_transfer_params, pooling=pooling)
        if lora_name and self.lora_manager is not None:
            req.lora_slot = self.lora_manager.slot_of(lora_name)
        self.requests[rid] = req
        self.scheduler.add(req)
        return req

    def abort(self, request_id: str) -> None:
        live = self.requests.get(request_id)
        if live is not None and self.connector is not None and live.kv_transfer_params:
            self.connector.on_request_finished(live)  # closes a transfer that chunks already started
        req = self.scheduler.abort(request_id)
        self.requests.pop(request_id, None)
        if req is not None:
            self.runner.sampler.forget(request_id)

    def has_work(self) -> bool:
        return self.scheduler.has_work()

    # ------------------------------------------------------------------ step
    def idle_tick(self) -> None:
        """Called by the serving loop when there is nothing to step: bookkeeping that normally
        rides along with steps (references held for finished copies and pushes)."""
        if self.connector is not None:
            self.connector.poll()

    def step(self) -> list[StepOutput]:
        t_a = time.perf_counter()
        if self.connector is not None:
            self.connector.poll()
        out = self.scheduler.schedule()
        t_b = time.perf_counter()
        self.step_count += 1
        if out.empty:
            self.metrics.update_scheduler(self.scheduler, self.bm)
            return []
        now = time.time()
        if _TRACE:
            print("STEP dec=%d pre=%s free=%d" % (
                len(out.decodes), [(s.start, s.n, len(s.req.pages)) for s in out.prefills],
                self.bm.num_free()), file=sys.stderr, flush=True)
        k = self._decode_run_length(out)
        results = self.runner.execute_decode_run(out.decodes, k) if k > 1 else self.runner.execute(out)
        t_c = time.perf_counter()
        sampled = {id(s): (tok, info) for s, tok, info in results}
        outputs: list[StepOutput] = []
        fresh_pages: list[tuple[int, int]] = []
        done: list[tuple[Request, str, float]] = []
        n_generated = 0
        for s in out.decodes + out.prefills:
            req = s.req
            if s.start < req.prompt_len:
                self.metrics.prompt_tokens.inc(min(s.n, req.prompt_len - s.start))
            fresh_pages.extend(self.scheduler.commit(s))
            if self.connector is not None and req.kv_transfer_params and s.start < req.prompt_len:
                self.connector.on_chunk_computed(req)  # P->D: this chunk's pages leave now
            if id(s) not in sampled:
                continue
            toks, info = sampled[id(s)]
            if not isinstance(toks, list):
                toks = [toks]
            t = time.time()
            # several tokens: the launch ran that many decode steps of this row back to back
            gap = (t - req.last_token) / len(toks) if req.last_token is not None else None
            emitted: list[int] = []
            reason = None
            for j, tok in enumerate(toks):
                if j:  # the position of the previous token of this run has been computed as well
                    fresh_pages.extend(self.scheduler.commit(ScheduledSeq(req, s.start + j, 1, True)))
                req.output_ids.append(tok)
                emitted.append(tok)
                n_generated += 1
                if req.first_token is None:
                    req.first_token = t
                    self.metrics.ttft.observe(t - req.arrival)
                    if req.first_scheduled:
                        self.metrics.queue_time.observe(req.first_scheduled - req.arrival)
                elif gap is not None:
                    self.metrics.tpot.observe(gap)
                reason = self._stop_reason(req, tok)
                if reason:
                    break  # what the run computed past this token is dropped
            req.last_token = t
            if reason:
                done.append((req, reason, t))
            outputs.append(StepOutput(req.request_id, emitted, reason is not None, reason,
                                      req.prompt_len, len(req.output_ids), req.num_cached_tokens + req.num_external_tokens,
                                      (req.first_scheduled - req.arrival) if req.first_scheduled else 0.0,
                                      [info] if info is not None else None))
        # Pages filled in this step go to the lower tiers BEFORE their requests finish: the copy
        # takes a reference on each page, which only holds while somebody still owns it (a request
        # whose last token fills a page would otherwise have handed that page back already, and the
        # reference would land on a free page that the next request is given).
        if self.connector is not None and fresh_pages:
            self.connector.store_pages(fresh_pages)
        if n_generated:
            self.metrics.generation_tokens.inc(n_generated)
        for req, reason, t in done:
            if self.connector is not None:
                # while the request still owns its pages (P->D push reads them)
                self.connector.on_request_finished(req)
            self.scheduler.finish(req, reason)
            self.requests.pop(req.request_id, None)
            self.runner.sampler.forget(req.request_id)
            self.metrics.e2e.observe(t - req.arrival)
            self.metrics.success.inc()
        self.metrics.update_scheduler(self.scheduler, self.bm)
        del now
        pr = self.prof
        pr["steps"] += 1
        pr["decode_only_steps" if not out.prefills else "mixed_steps"] += 1
        pr["rows"] += out.num_tokens + (k - 1) * len(out.decodes)
        pr["t_schedule"] += t_b - t_a
        pr["t_execute"] += t_c - t_b
        pr["t_post"] += time.perf_counter() - t_c
        pr["t_plan"] = self.runner.t_plan
        pr["t_materialize"] = self.runner.t_materialize
        pr["t_launch"] = self.runner.t_launch
        pr["t_launch_graph_pairs"] = self.runner.t_launch_graph_pairs
        for kind, n in self.runner.step_kinds.items():  # how steps ran: captured graphs or eager launches
            pr["steps_" + kind] = n
        return outputs

    def _decode_run_length(self, out) -> int:
        """How many decode steps to launch back to back (``--num-scheduler-steps``): more than one
        only for a decode-only step of plain greedy rows with nothing waiting to be admitted, every
        row with that many tokens still to go and the pages for them in hand."""
        k = self.cfg.num_scheduler_steps
        if k <= 1 or out.prefills or not out.decodes or self.scheduler.waiting:
            return 1
        if not self.runner.decode_run_ok(out.decodes):
            return 1
        sampler = self.runner.sampler
        for s in out.decodes:
            req, p = s.req, s.req.params
            if not p.greedy or p.logprobs is not None or sampler._needs_penalty(p) or req.pooling:
                return 1
            k = min(k, p.max_tokens - len(req.output_ids), self.max_model_len - req.num_tokens - 1)
            if k <= 1:
                return 1
        for s in out.decodes:
            if not self.scheduler._ensure_pages(s.req, s.start + k):
                return 1  # (pages taken for rows before this one stay with their requests)
        return k

    def _stop_reason(self, req: Request, tok: int) -> str | None:
        p = req.params
        n_out = len(req.output_ids)
        if n_out >= p.min_tokens:
            if not p.ignore_eos and tok in self.eos_ids:
                return "stop"
            if tok in p.stop_token_ids:
                return "stop"
        if n_out >= p.max_tokens:
            return "length"
        if req.num_tokens >= self.max_model_len:
            return "length"
        return None

    # ------------------------------------------------------------------ convenience
    def generate(self, prompts: list[list[int]], params) -> list[list[int]]:
        """Blocking helper: run the prompts to completion, return generated ids per prompt.
        ``params``: one ``SamplingParams`` for all prompts or a list with one per prompt."""
        import copy

        per = params if isinstance(params, (list, tuple)) else [params] * len(prompts)
        reqs = [self.add_request(p, copy.deepcopy(sp)) for p, sp in zip(prompts, per)]
        while self.has_work():
            self.step()
        return [r.output_ids for r in reqs]

    # ------------------------------------------------------------------ sleep mode
    def sleep(self, level: int = 1) -> int:
        if self.scheduler.running or self.scheduler.waiting:
            raise EngineBusyError("cannot sleep with requests in flight")
        self.bm.reset_prefix_cache()
        return self.runner.sleep(level)

    def wake_up(self) -> None:
        self.runner.wake_up()

    @property
    def is_sleeping(self) -> bool:
        return self.runner.sleeping

    def shutdown(self, release_memory: bool = True) -> None:
        """Stop the side threads and (by default) free the device memory: weights, KV pool,
        captured graphs, host-tier slab."""
        if self.connector is not None:
            self.connector.close()
        self.runner.stop_workers()
        if release_memory:
            if self.connector is not None:
                self.connector.pool = None
            self.runner.release()
