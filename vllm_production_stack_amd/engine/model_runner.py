"""Executes scheduler output on one GPU: batch assembly, model forward, sampling.

Memory plan (per process = per GPU): weights, then one contiguous paged KV pool sized from
what is left of ``gpu_memory_utilization`` x HBM after an activation reserve.  The pool is
laid out [page][layer][K|V][kv_head][64*128] so a page's whole depth (all layers) is one
contiguous run: offload / peer transfer of a page is a single large copy.

Decode-only steps replay a HIP graph captured per padded batch size (the launch-bound
inner loop: ~10 launches per layer); mixed and prefill steps run eagerly.

Tensor / pipeline parallel replicas (one process per GPU): the driver rank turns the
scheduler output into a host-side *plan* (plain lists), broadcasts it to the replica's
ranks, and every rank materialises the same batch and runs its shard / stage of the model.
Worker ranks sit in ``worker_loop``.  Logits are produced on the first pipeline stage (it
holds the vocab-parallel LM head); the last stage returns the normed hidden rows to it.
"""

from __future__ import annotations

import logging
import time

import numpy as np
import torch

from .. import ops
from ..models.batch import BatchMeta
from ..models.loader import build_model
from ..parallel.state import (CTRL_INTS, check_collectives, enable_oneshot_all_reduce, get_pp, get_replica, get_tp,
                              pp_recv, pp_send, replica_broadcast, replica_broadcast_ints, replica_min_int,
                              replica_next)
from .block_manager import PAGE
from .config import EngineConfig, ModelConfig
from .sampler import Sampler
from .scheduler import ScheduledSeq, SchedulerOutput

log = logging.getLogger("vpsa.runner")

_GRAPH_BUCKETS = [1, 2, 4, 8, 12, 16, 24, 32, 40, 48, 64, 80, 96, 128, 160, 192, 256]
# Prefill parts small enough to be worth a graph: the new turn of a conversation whose history is
# cached, the tail of a chunked prompt.  A graph holds this many rows (padded) in up to this many
# sequences; larger prefills run eagerly (their GPU time dwarfs the launches).
_PREFILL_ROW_BUCKETS = [64, 128, 256, 512]
_PREFILL_GRAPH_SEQS = 8
_PREFILL_GRAPH_SPLIT_RECORDS = 72  # partial records a prefill graph has room for (1 MiB each for Llama-3-8B)


class ModelRunner:
    def __init__(self, cfg: EngineConfig, model_cfg: ModelConfig, ckpt_dir, device: torch.device):
        self.cfg = cfg
        self.model_cfg = model_cfg
        self.device = device
        self.dtype = cfg.torch_dtype() if device.type == "cuda" else (
            torch.float32 if cfg.dtype in ("auto", "float32", "float") else cfg.torch_dtype())
        self.max_model_len = cfg.max_model_len or min(model_cfg.max_position_embeddings, 32768)
        if device.type == "cuda":
            torch.cuda.set_device(device)
            ops.bind_device(device.index if device.index is not None else torch.cuda.current_device())
        t0 = time.time()
        self.model = build_model(model_cfg, self.dtype, device, self.max_model_len, ckpt_dir,
                                 cfg.seed, cfg.load_format, expert_parallel=cfg.enable_expert_parallel)
        log.info("model %s ready: %.2f GiB weights in %.1fs", model_cfg.name,
                 self.model.weight_bytes() / 2**30, time.time() - t0)
        self._pack_decode_weights()
        self.n_kv_local = self.model.n_kv
        self.replica = get_replica()
        self.distributed = self.replica.size > 1
        # layers whose KV this rank holds (its pipeline stage)
        self.n_layers_local = len(self.model.layers)
        # channels per head as stored in the pool (heads of 64 are padded to the kernels' 128 on a GPU)
        self.kv_head_dim = ops.kv_head_dim(model_cfg.head_dim, device)
        self.page_elems = self.n_layers_local * 2 * self.n_kv_local * PAGE * self.kv_head_dim
        self.page_bytes = self.page_elems * torch.empty((), dtype=self.dtype).element_size()
        self.max_pages_per_seq = (self.max_model_len + PAGE - 1) // PAGE
        self.num_pages = self._plan_pages()
        self.pool_listeners: list = []  # called with the new pool tensor after a wake-up
        self._pool_dev_ptr = 0       # an allocation of our own behind the pool (exported to peers), or 0
        self.pool_ipc_handle = None  # hex hipIpc handle of that allocation
        self.pool = self._alloc_pool()
        self.sampler = Sampler(device, cfg.seed)
        # graphs need the HIP attention kernels (the generic path reads lengths on the host)
        self.use_graphs = (device.type == "cuda" and not cfg.enforce_eager
                           and self.kv_head_dim == ops.HEAD_DIM_HIP
                           and (not self.distributed or self._tp_graphs_ready()))
        self._graphs: dict[int, dict] = {}
        self._prefill_graphs: dict[int, dict] = {}
        # steps by how they ran (decode graph, prefill graph, both, eager)
        self.step_kinds = {"decode_graph": 0, "decode_graph_follow_ups": 0, "prefill_graph": 0, "mixed_graphs": 0,
                           "eager": 0}
        self._row_track: dict[tuple, dict] = {}  # driver: what each graph row's block table holds
        self._graph_pool = None
        self._pin = device.type == "cuda"
        self.sleeping = False
        self._sleep_stash: list[tuple[torch.Tensor, torch.Tensor]] = []
        # host-side seconds of eager steps: plan lists, device tensors, kernel launches
        self.t_plan = self.t_materialize = self.t_launch = 0.0
        self.t_launch_graph_pairs = 0.0  # of t_launch: steps whose prefill part replayed a graph

    def _pack_decode_weights(self) -> None:
        mode = getattr(self.cfg, "packed_decode_weights", "off")
        if mode == "off" or self.device.type != "cuda" or not hasattr(self.model, "pack_decode_weights"):
            return
        total = torch.cuda.mem_get_info(self.device)[1]
        used = self.model.pack_decode_weights(total if mode == "on" else total // 10)
        if used:
            log.info("decode-step copy of the MLP weights in matrix-core tile order: %.2f GiB", used / 2**30)

    def _tp_graphs_ready(self) -> bool:
        """Tensor-parallel replicas (no pipeline stages) replay decode graphs on every rank when
        the collectives inside them are graph-replayable kernels of this package: the one-shot
        all-reduce and all-gather over mapped peer buffers.  The buffers are re-created here when
        the logits shard of the largest graph batch would not fit.  ``VPSA_TP_GRAPHS=rccl``
        captures the RCCL collectives instead (not measured on hardware); pipeline stages and
        everything else run eagerly."""
        import os

        tp, pp = get_tp(), get_pp()
        if pp.size > 1 or tp.size == 1 or self.dtype != torch.bfloat16:
            return False
        if tp.oneshot is None:
            return os.environ.get("VPSA_TP_GRAPHS", "") == "rccl"
        top = self._bucket(min(self.cfg.max_num_seqs, _GRAPH_BUCKETS[-1]))
        need = top * max(self.model.lm_head.shape[0], self.model_cfg.hidden_size) * 2
        if need > tp.oneshot.max_elems * 2:
            enable_oneshot_all_reduce(self.device, max_bytes=(need + (1 << 20) - 1) >> 20 << 20)
        return True

    # ------------------------------------------------------------------ memory plan
    def _exports_pool(self) -> bool:
        """A decode replica of a disaggregated pair on a GPU lets its prefill peers write KV pages
        straight into its pool (hipIpc): the pool is then one device allocation of its own, of a
        size hipIpc can map, instead of a block of the caching allocator."""
        import os

        kt = self.cfg.kv_transfer
        return (self.device.type == "cuda" and kt is not None and kt.kv_role in ("kv_consumer", "kv_both")
                and os.environ.get("VPSA_PD_EXPORT_POOL", "1") != "0" and get_replica().size == 1)

    # a pool below this size lies within the reach of address translation whatever its layout
    LAYER_GROUPS_FROM_BYTES = 48 << 30

    def _layer_groups(self) -> int:
        """In how many layer groups the KV pool is allocated (``ops.LayerGroupedPool``).  1 = whole
        pages, which the transfer of pages between replicas needs (a peer writes whole pages into
        this pool, or reads them from it, in one copy each); the lower tiers (host memory, disk,
        remote server) move a page piece by piece, which costs them: with every follow-up turn
        reloaded from pinned host memory (prefix caching off), four 2 MiB copies per page instead
        of one of 8 MiB took the reload-bound benchmark phase from 450 to 432 output tok/s.  So
        the default of 0 picks 4 only for a large GPU pool on a replica without lower tiers and
        outside such transfers; with tiers it is the operator's call (few reloads: take it)."""
        g = int(self.cfg.kv_layer_groups)
        off, kvt = self.cfg.kv_offload, self.cfg.kv_transfer
        whole_pages = (off.enable_pd or kvt is not None) and not self.distributed
        if g == 0:
            tiers = (off.enabled or off.enable_controller) and not self.distributed
            g = 4 if (self.device.type == "cuda" and not whole_pages and not tiers
                      and self.num_pages * self.page_bytes >= self.LAYER_GROUPS_FROM_BYTES) else 1
        if g > 1 and whole_pages:
            log.warning("--kv-layer-groups %d ignored: transfers between replicas move whole pages", g)
            g = 1
        while g > 1 and self.n_layers_local % g:
            g -= 1
        return max(1, g)

    def _alloc_pool(self):
        shape = (self.num_pages, self.n_layers_local, 2, self.n_kv_local, PAGE * self.kv_head_dim)
        g = self._layer_groups()
        if g > 1:
            return ops.LayerGroupedPool([
                torch.zeros((shape[0], shape[1] // g, *shape[2:]), dtype=self.dtype, device=self.device)
                for _ in range(g)])
        if self._exports_pool():
            try:
                return self._alloc_exported_pool()
            except Exception as e:
                log.warning("KV pool could not be exported to prefill peers (%s): pushes land in a window", e)
        return torch.zeros(shape, dtype=self.dtype, device=self.device)

    def _alloc_exported_pool(self) -> torch.Tensor:
        import ctypes

        from ..parallel import kv_transfer as kx

        pages = self.num_pages
        while kx.ipc_safe_size(pages * self.page_bytes) != pages * self.page_bytes:
            pages = kx.ipc_safe_size(pages * self.page_bytes) // self.page_bytes  # gives up < 2 GiB + one page
        if pages < self.max_pages_per_seq:
            raise MemoryError("no exportable pool size holds one sequence")
        lib = kx.comm()
        nbytes = pages * self.page_bytes
        ptr = lib.comm_dev_alloc(ctypes.c_size_t(nbytes))
        if not ptr:
            raise MemoryError(f"hipMalloc of {nbytes >> 20} MiB failed")
        buf = ctypes.create_string_buffer(lib.comm_ipc_handle_size())
        if lib.comm_ipc_export(ctypes.c_void_p(ptr), buf) != 0:
            lib.comm_dev_free(ctypes.c_void_p(ptr))
            raise RuntimeError("hipIpcGetMemHandle failed")

        class _DevMem:  # what torch.as_tensor needs to alias device memory it did not allocate
            __cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (int(ptr), False),
                                        "version": 2, "strides": None}

        raw = torch.as_tensor(_DevMem(), device=self.device)
        assert raw.data_ptr() == ptr
        raw.zero_()
        if pages != self.num_pages:
            log.info("KV pool: %d pages instead of %d so that peers can map it", pages, self.num_pages)
        self.num_pages = pages
        self._pool_dev_ptr, self.pool_ipc_handle = int(ptr), buf.raw.hex()
        return raw.view(self.dtype).view(pages, self.n_layers_local, 2, self.n_kv_local, PAGE * self.kv_head_dim)

    def _free_exported_pool(self) -> None:
        """After every tensor view of the pool is gone."""
        if self._pool_dev_ptr:
            import ctypes

            from ..parallel import kv_transfer as kx

            torch.cuda.synchronize(self.device)
            kx.comm().comm_dev_free(ctypes.c_void_p(self._pool_dev_ptr))
            self._pool_dev_ptr, self.pool_ipc_handle = 0, None

    def _plan_pages(self) -> int:
        if self.cfg.num_gpu_blocks_override:
            return int(self.cfg.num_gpu_blocks_override)
        return replica_min_int(self._plan_pages_local())

    def _plan_pages_local(self) -> int:
        mc = self.model_cfg
        if self.device.type != "cuda":
            # CPU replicas: a modest pool (enough for max_num_seqs short sequences)
            want = max(64, min(4096, self.cfg.max_num_seqs * 8))
            return want
        torch.cuda.synchronize()
        free, total = torch.cuda.mem_get_info(self.device)
        budget = total * self.cfg.gpu_memory_utilization - (total - free)
        rows = self.cfg.max_num_batched_tokens if self.cfg.enable_chunked_prefill else max(
            self.cfg.max_num_batched_tokens, self.max_model_len)
        inter = mc.intermediate_size // get_tp().size
        per_row = (mc.q_size + 2 * mc.kv_size + mc.q_size + 3 * inter + 4 * mc.hidden_size) * 2
        if mc.num_experts:
            per_row += 3 * inter * 2 * mc.num_experts_per_tok
        reserve = int(rows * per_row * 1.25) + self.cfg.max_num_seqs * mc.vocab_size * 6 + (2 << 30)
        pages = int((budget - reserve) // self.page_bytes)
        if pages < self.max_pages_per_seq:
            raise MemoryError(
                f"not enough HBM for one sequence of max_model_len={self.max_model_len}: "
                f"{pages} pages of {self.page_bytes >> 20} MiB available")
        return pages

    # ------------------------------------------------------------------ batch assembly
    def _to_dev(self, data: list[int], dtype: torch.dtype) -> torch.Tensor:
        t = torch.tensor(data, dtype=dtype)
        if self._pin:
            return t.pin_memory().to(self.device, non_blocking=True)
        return t

    def _plan(self, decodes: list[ScheduledSeq], prefills: list[ScheduledSeq]):
        """Host-side description of the step (plain lists: cheap to broadcast to worker ranks)
        and the sequences that sample, in logits-row order."""
        ids: list[int] = []
        pos: list[int] = []
        slots: list[int] = []
        logits_rows: list[int] = []
        sample_seqs: list[ScheduledSeq] = []
        row = 0
        for s in decodes:
            r = s.req
            tok_idx = s.start
            ids.append(r.prompt_ids[tok_idx] if tok_idx < r.prompt_len else r.output_ids[tok_idx - r.prompt_len])
            pos.append(tok_idx)
            slots.append(r.pages[tok_idx // PAGE] * PAGE + tok_idx % PAGE)
            if s.samples:
                logits_rows.append(row)
                sample_seqs.append(s)
            row += 1
        for s in prefills:
            r = s.req
            if s.start + s.n <= r.prompt_len:
                ids.extend(r.prompt_ids[s.start:s.start + s.n])
            else:
                ids.extend(r.all_ids[s.start:s.start + s.n])
            pos.extend(range(s.start, s.start + s.n))
            t = np.arange(s.start, s.start + s.n, dtype=np.int64)
            first = s.start // PAGE
            pg = np.asarray(r.pages[first:(s.start + s.n - 1) // PAGE + 1], dtype=np.int64)
            slots.extend((pg[t // PAGE - first] * PAGE + t % PAGE).tolist())
            row += s.n
            if s.samples:
                logits_rows.append(row - 1)
                sample_seqs.append(s)
        plan = {"ids": ids, "pos": pos, "slots": slots, "logits_rows": logits_rows,
                "num_decode": len(decodes), "num_prefill": len(prefills)}
        if decodes:
            plan["dec_pages"] = [s.req.pages for s in decodes]
            plan["dec_ctx"] = [s.start + 1 for s in decodes]
        if prefills:
            plan["pre_pages"] = [s.req.pages for s in prefills]
            plan["q_lens"] = [s.n for s in prefills]
            plan["pre_ctx"] = [s.start + s.n for s in prefills]
        if self.model.lora is not None:
            lslots = [s.req.lora_slot for s in decodes]
            for s in prefills:
                lslots.extend([s.req.lora_slot] * s.n)
            if any(x >= 0 for x in lslots):
                plan["lora_slots"] = lslots
        return plan, sample_seqs

    def _materialize(self, plan: dict):
        """Plan -> device tensors (input ids + BatchMeta)."""
        meta = BatchMeta(positions=self._to_dev(plan["pos"], torch.int32),
                         slots=self._to_dev(plan["slots"], torch.int64),
                         num_decode=plan["num_decode"], num_prefill_seqs=plan["num_prefill"])
        if plan["num_decode"]:
            pages = plan["dec_pages"]
            width = max(len(p) for p in pages)
            meta.dec_block_tables = self._to_dev([p + [0] * (width - len(p)) for p in pages], torch.int32)
            ctx = plan["dec_ctx"]
            meta.dec_ctx_lens = self._to_dev(ctx, torch.int32)
            # same split count as the captured graph of this batch bucket: eager and graph
            # steps then reduce in the same order and generate identical tokens
            meta.dec_num_splits = ops.decode_num_splits(self._bucket(len(ctx)), self.n_kv_local,
                                                        self.max_model_len)
        if plan["num_prefill"]:
            pages = plan["pre_pages"]
            width = max(len(p) for p in pages)
            meta.pre_block_tables = self._to_dev([p + [0] * (width - len(p)) for p in pages], torch.int32)
            q_lens = plan["q_lens"]
            cu = [0]
            for q in q_lens:
                cu.append(cu[-1] + q)
            meta.pre_cu_q = self._to_dev(cu, torch.int32)
            meta.pre_ctx_lens = self._to_dev(plan["pre_ctx"], torch.int32)
            if self.device.type == "cuda":
                rows = ops.prefill_tile_rows(q_lens, self.model.n_q, self.model.n_kv)
                ts, tq = ops.build_prefill_tiles(q_lens, self.model.n_q, self.model.n_kv, rows, plan["pre_ctx"])
                cut = ops.plan_prefill_splits(ts, tq, q_lens, plan["pre_ctx"], self.model.n_q, self.model.n_kv, rows)
                if cut is not None:  # few tiles, long contexts: the key ranges run as workgroups of their own
                    ts, tq, split, groups, n_rec = cut
                    meta.pre_split = self._to_dev(split, torch.int32)
                    meta.pre_split_groups = self._to_dev(groups, torch.int32)
                    meta.pre_split_ws = torch.empty(
                        n_rec * ops.prefill_split_record_bytes(self.model.n_q, self.model.n_kv) // 4,
                        dtype=torch.float32, device=self.device)
                meta.pre_tile_rows = rows
                meta.pre_tile_seq = self._to_dev(ts, torch.int32)
                meta.pre_tile_q0 = self._to_dev(tq, torch.int32)
        rows = plan["logits_rows"]
        meta.logits_rows = self._to_dev(rows, torch.int64) if rows else None
        if plan.get("lora_slots") is not None:
            meta.lora_slots = self._to_dev(plan["lora_slots"], torch.int32)
        return self._to_dev(plan["ids"], torch.int64), meta

    def _run_plan(self, plan: dict):
        """Every rank of the replica: run this rank's part of the step.  Returns the logits
        ([num_out, vocab]) on first-stage ranks when the step samples, else ``None``."""
        t0 = time.perf_counter()
        ids, meta = self._materialize(plan)
        t1 = time.perf_counter()
        hidden = self.model.forward(ids, self.pool, meta)
        self.t_materialize += t1 - t0
        self.t_launch += time.perf_counter() - t1
        n_out = len(plan["logits_rows"])
        if n_out == 0:
            return None, None
        pp = get_pp()
        if pp.size > 1:
            # the last stage hands the normed rows back to the first (same tensor-parallel shard)
            if pp.is_last:
                pp_send(hidden, pp.first)
                hidden = None
            elif pp.is_first:
                hidden = pp_recv((n_out, self.model_cfg.hidden_size), self.dtype, self.device, pp.last)
        if not pp.is_first:
            return None, None
        return hidden, self.model.compute_logits(hidden)

    # ------------------------------------------------------------------ execution
    @torch.inference_mode()
    def execute(self, out: SchedulerOutput) -> list[tuple[ScheduledSeq, int, dict | None]]:
        """Run one step; returns (sequence, sampled token, logprob info) per sampling sequence."""
        if out.empty:
            return []
        for s in out.decodes + out.prefills:
            ev = s.req.load_event
            if ev is not None:
                if self.device.type == "cuda" and hasattr(ev, "record"):
                    torch.cuda.current_stream(self.device).wait_event(ev)
                elif hasattr(ev, "wait"):
                    ev.wait()
                s.req.load_event = None
        if self.device.type == "cuda":
            ops.ensure_device()  # the calling thread may never have selected this process's GPU
        if self.use_graphs:
            dec_ok = (all(s.samples for s in out.decodes) and len(out.decodes) <= _GRAPH_BUCKETS[-1])
            if not out.prefills and dec_ok:
                self.step_kinds["decode_graph"] += 1
                return self._execute_decode_graph(out.decodes)
            if dec_ok and self._prefill_graph_ok(out.prefills):
                self.step_kinds["mixed_graphs" if out.decodes else "prefill_graph"] += 1
                return self._execute_graph_pair(out.decodes, out.prefills)
        self.step_kinds["eager"] += 1
        t0 = time.perf_counter()
        plan, sample_seqs = self._plan(out.decodes, out.prefills)
        self.t_plan += time.perf_counter() - t0
        if self.distributed:
            replica_broadcast(("step", plan))
        hidden, logits = self._run_plan(plan)
        if not sample_seqs:
            return []
        for i, s in enumerate(sample_seqs):
            if s.req.pooling:
                s.req.pooled = hidden[i].float().cpu()
        toks, infos = self.sampler.sample(logits, [s.req for s in sample_seqs])
        if self.distributed:
            check_collectives()  # the sampled tokens are on the host: every kernel of the step has run
        return list(zip(sample_seqs, toks, infos))

    # ------------------------------------------------------------------ worker ranks
    def worker_loop(self) -> None:
        """Non-driver ranks of a tensor/pipeline-parallel replica: execute what the driver
        broadcasts until it says stop."""
        assert self.distributed and not self.replica.is_driver
        with torch.inference_mode():
            while True:
                ctrl, msg = replica_next()
                if ctrl == CTRL_INTS:  # a decode step that replays this rank's captured graph
                    self._replay_decode(self._decode_plan_from_ints(msg))
                    self._check_worker_collectives()
                    continue
                kind = msg[0]
                if kind == "stop":
                    return
                if kind == "step":
                    self._run_plan(msg[1])
                    self._check_worker_collectives()
                elif kind == "dstep":
                    self._replay_decode(msg[1])
                    self._check_worker_collectives()
                elif kind == "sleep":
                    self._sleep_local(msg[1])
                elif kind == "wake":
                    self._wake_local()

    @staticmethod
    def _check_worker_collectives() -> None:
        """Lags a step on worker ranks (they never read back).  A failed collective ends the worker
        loop with the error: the rank's process exits non-zero and the replica is restarted."""
        try:
            check_collectives()
        except RuntimeError as e:
            log.error("%s", e)
            raise

    def stop_workers(self) -> None:
        if self.distributed and self.replica.is_driver:
            replica_broadcast(("stop",))

    def release(self) -> None:
        """Give the device memory back: graphs first (they pin their private pool and every
        tensor they captured), then the KV pool, the weights and the sampler state.  The runner
        is unusable afterwards.  Needed because a stopped engine can stay reachable for a long
        time (aiohttp keeps the handlers of a finished application in a module-level cache), and
        reachable must not mean resident."""
        import gc

        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        self._graphs.clear()
        self._prefill_graphs.clear()
        self._row_track.clear()
        self._graph_pool = None
        self._sleep_stash = []
        # shrink the storages themselves: the transfer receiver, the tier connector and the
        # attention layers hold views of the pool, and a view must not keep 200 GiB resident
        stores = list(getattr(self.pool, "groups", [self.pool])) if self.pool is not None else []
        if self.model is not None and hasattr(self.model, "weight_tensors"):
            stores += list(self.model.weight_tensors())
        if self.model is not None and hasattr(self.model, "packed_tensors"):
            stores += list(self.model.packed_tensors())
        for t in stores:
            try:
                t.untyped_storage().resize_(0)
            except Exception:  # storage that cannot be resized (shared with another process)
                pass
        self.pool = None
        self.model = None
        self.sampler = None
        gc.collect()
        self._free_exported_pool()
        if self.device.type == "cuda":
            torch.cuda.empty_cache()

    # ------------------------------------------------------------------ HIP-graph decode
    def _bucket(self, n: int) -> int:
        for b in _GRAPH_BUCKETS:
            if b >= n:
                return b
        return n

    def _capture(self, bsz: int, with_lora: bool = False) -> dict:
        dev = self.device
        mp = self.max_pages_per_seq
        # Graph inputs.  ids/slots and pos/ctx share one device tensor each, and every group has
        # a pinned host twin that is filled in place and sent with one copy per step (the step
        # ends with a read-back of the sampled tokens, so a twin is free again before the next
        # step touches it).  The block-table twin keeps what it holds per row between steps:
        # a row is rewritten only when another request moves into it or its page list grows.
        d64 = torch.zeros((2, bsz), dtype=torch.int64, device=dev)
        d32 = torch.zeros((2, bsz), dtype=torch.int32, device=dev)
        d64[1].fill_(-1)
        h64 = torch.zeros((2, bsz), dtype=torch.int64)
        h32 = torch.zeros((2, bsz), dtype=torch.int32)
        hbt = torch.zeros((bsz, mp), dtype=torch.int32)
        if self._pin:
            h64, h32, hbt = h64.pin_memory(), h32.pin_memory(), hbt.pin_memory()
        st = {
            "lora": torch.full((bsz,), -1, dtype=torch.int32, device=dev),
            "lora_version": self.model.lora.version if with_lora else 0,
            "d64": d64, "d32": d32, "ids": d64[0], "slots": d64[1], "pos": d32[0], "ctx": d32[1],
            "bt": torch.zeros((bsz, mp), dtype=torch.int32, device=dev),
            "h64": h64, "h32": h32, "hbt": hbt,
            "n64": h64.numpy(), "n32": h32.numpy(), "nbt": hbt.numpy(),
        }
        splits = ops.decode_num_splits(bsz, self.n_kv_local, self.max_model_len)
        mc = self.model_cfg
        ws_o = torch.empty((bsz, self.model.n_q, splits, self.kv_head_dim), dtype=torch.float32, device=dev)
        ws_ml = torch.empty((bsz, self.model.n_q, splits, 2), dtype=torch.float32, device=dev)
        meta = BatchMeta(positions=st["pos"], slots=st["slots"], num_decode=bsz,
                         dec_block_tables=st["bt"], dec_ctx_lens=st["ctx"], dec_num_splits=splits,
                         dec_ws_o=ws_o, dec_ws_ml=ws_ml, logits_rows=None,
                         lora_slots=st["lora"] if with_lora else None)

        def run():
            hidden = self.model.forward(st["ids"], self.pool, meta)
            logits = self.model.compute_logits(hidden)
            return logits, ops.argmax(logits)

        # warm up on a side stream (library workspaces, lazy inits), then capture
        s = torch.cuda.Stream(device=dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s):
            run()
        torch.cuda.current_stream(dev).wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, pool=self._graph_pool):
            logits, toks = run()
        if self._graph_pool is None:
            self._graph_pool = g.pool()
        # the graph's kernels hold raw pointers into the split-K workspaces: keep them alive
        st.update(graph=g, logits=logits, toks=toks, meta=meta, ws_o=ws_o, ws_ml=ws_ml)
        return st

    def _launch_decode_graph(self, decodes: list[ScheduledSeq]) -> dict:
        """Driver: describe the step as a compact plan (per-row scalars and the block-table rows
        that changed), send it to the replica's other ranks, replay the graph."""
        n = len(decodes)
        bsz = self._bucket(n)
        lora = self.model.lora
        with_lora = lora is not None and any(s.req.lora_slot >= 0 for s in decodes)
        ids, pos, slots, ctx = [0] * bsz, [0] * bsz, [-1] * bsz, [0] * bsz
        # which request's page list each graph row held last step, and how much of it: a row of
        # the block table is resent only when another request moves in or its list has grown
        key = (bsz, True) if with_lora else bsz
        st = self._graphs.get(key)
        if st is None or (with_lora and st["lora_version"] != lora.version):
            self._row_track.pop(key, None)  # the graph about to be captured starts from an empty table
        track = self._row_track.setdefault(key, {"pages": [None] * bsz, "len": [0] * bsz})
        row_pages, row_len = track["pages"], track["len"]
        bt: list[tuple[int, int, list[int]]] = []
        for i, s in enumerate(decodes):
            r = s.req
            t = s.start
            pages = r.pages
            ids[i] = r.output_ids[t - r.prompt_len] if t >= r.prompt_len else r.prompt_ids[t]
            pos[i] = t
            slots[i] = pages[t // PAGE] * PAGE + t % PAGE
            ctx[i] = t + 1
            k = len(pages)
            if row_pages[i] is not pages:  # another request (or a re-admitted one) moved in
                bt.append((i, 0, pages[:k]))
                row_pages[i], row_len[i] = pages, k
            elif row_len[i] != k:  # page lists only grow while a request keeps its list
                lo = min(row_len[i], k)
                bt.append((i, lo, pages[lo:k]))
                row_len[i] = k
        for i in range(n, bsz):  # rows past the batch hold no request (ctx 0: nothing is read)
            row_pages[i] = None
        plan = {"bsz": bsz, "ids": ids, "pos": pos, "slots": slots, "ctx": ctx, "bt": bt}
        if with_lora:
            plan["lora"] = [s.req.lora_slot for s in decodes] + [-1] * (bsz - n)
        if self.distributed:
            if not replica_broadcast_ints(self._decode_plan_to_ints(plan)):
                replica_broadcast(("dstep", plan))
        return self._replay_decode(plan)

    def _execute_decode_graph(self, decodes: list[ScheduledSeq]):
        st = self._launch_decode_graph(decodes)
        reqs = [s.req for s in decodes]
        toks, infos = self.sampler.sample(st["logits"], reqs, st["toks"])
        if self.distributed:
            check_collectives()
        return list(zip(decodes, toks, infos))

    # ------------------------------------------------------------------ several decode steps per launch
    def decode_run_ok(self, decodes: list[ScheduledSeq]) -> bool:
        """Can ``execute_decode_run`` take these rows?  Single-rank graph decode without adapters."""
        return bool(self.use_graphs and not self.distributed and self.device.type == "cuda" and decodes
                    and len(decodes) <= _GRAPH_BUCKETS[-1] and all(s.samples for s in decodes)
                    and (self.model.lora is None or all(s.req.lora_slot < 0 for s in decodes)))

    def execute_decode_run(self, decodes: list[ScheduledSeq], k: int):
        """``k`` greedy decode steps of the same rows in one go: the graph is replayed ``k`` times
        without the host in between.  After each replay the sampled ids become the next input ids
        on the device, and positions, slots and context lengths of the next step (known in advance:
        every row simply advances by one token, its pages are already there) are copied from a
        buffer sent up front.  The host reads all ``k`` x rows tokens back once, so the gap between
        steps in which the GPU waits for scheduling and bookkeeping is paid once per ``k`` steps.
        Returns (sequence, [k tokens], None) per row; the caller decides how many of them count
        (a row that stops early simply ignores the rest: they were written into its own pages)."""
        n = len(decodes)
        ops.ensure_device()  # the calling thread may never have selected this process's GPU
        self.step_kinds["decode_graph"] += 1
        self.step_kinds["decode_graph_follow_ups"] += k - 1
        st = self._launch_decode_graph(decodes)  # first replay is on the stream
        bsz = st["ids"].shape[0]
        nx = st.get("run")
        if nx is None or nx["k"] < k:
            dev = self.device
            h64 = torch.zeros((max(k - 1, 1), bsz), dtype=torch.int64)
            h32 = torch.zeros((max(k - 1, 1), 2, bsz), dtype=torch.int32)
            if self._pin:
                h64, h32 = h64.pin_memory(), h32.pin_memory()
            nx = st["run"] = {"k": k, "h64": h64, "h32": h32, "n64": h64.numpy(), "n32": h32.numpy(),
                              "d64": torch.zeros_like(h64, device=dev), "d32": torch.zeros_like(h32, device=dev),
                              "hist": torch.zeros((k, bsz), dtype=torch.int64, device=dev)}
        n64, n32 = nx["n64"], nx["n32"]
        for j in range(1, k):
            slots, pos, ctx = [-1] * bsz, [0] * bsz, [0] * bsz
            for i, s in enumerate(decodes):
                t = s.start + j
                slots[i] = s.req.pages[t // PAGE] * PAGE + t % PAGE
                pos[i] = t
                ctx[i] = t + 1
            n64[j - 1], n32[j - 1, 0], n32[j - 1, 1] = slots, pos, ctx
        nx["d64"].copy_(nx["h64"], non_blocking=True)
        nx["d32"].copy_(nx["h32"], non_blocking=True)
        hist = nx["hist"]
        for j in range(1, k):
            hist[j - 1].copy_(st["toks"])
            st["ids"].copy_(st["toks"])
            st["slots"].copy_(nx["d64"][j - 1])
            st["pos"].copy_(nx["d32"][j - 1, 0])
            st["ctx"].copy_(nx["d32"][j - 1, 1])
            st["graph"].replay()
        hist[k - 1].copy_(st["toks"])
        toks = hist[:k, :n].t().tolist()  # one read-back: [row][step]
        return [(s, toks[i], None) for i, s in enumerate(decodes)]

    # ------------------------------------------------------------------ HIP-graph prefill parts
    def _prefill_graph_ok(self, prefills: list[ScheduledSeq]) -> bool:
        """A prefill part that a captured graph can hold: one rank, no adapters, no pooling, at most
        ``_PREFILL_GRAPH_SEQS`` sequences and the largest row bucket."""
        if (not prefills or self.distributed or len(prefills) > _PREFILL_GRAPH_SEQS
                or sum(s.n for s in prefills) > _PREFILL_ROW_BUCKETS[-1]):
            return False
        lora = self.model.lora is not None
        return not any(s.req.pooling or (lora and s.req.lora_slot >= 0) for s in prefills)

    def _capture_prefill(self, rows: int) -> dict:
        """Graph of a prefill-only forward pass over ``rows`` rows in up to ``_PREFILL_GRAPH_SEQS``
        sequences.  Everything the kernels read that changes from step to step lives in two device
        arrays (one per integer width) with pinned host twins, sent with one copy each, and the
        block tables.  Unused rows carry slot -1 (nothing is cached), unused sequences are empty,
        unused work-list entries are -1 (their workgroups leave at once)."""
        dev = self.device
        S = _PREFILL_GRAPH_SEQS
        mp = self.max_pages_per_seq
        tile = ops.prefill_tile_rows([rows], self.model.n_q, self.model.n_kv)
        base_tiles = rows // tile + S  # every sequence may end in a partial tile
        # room for key-range splits of a short work list (ops.plan_prefill_splits)
        max_records = _PREFILL_GRAPH_SPLIT_RECORDS if ops.prefill_splits_supported(self.model.n_q, self.model.n_kv, tile) else 0
        n_tiles = base_tiles + max_records
        # int32: pos[rows] | cu_q[S + 1] | ctx[S] | tile_seq[n_tiles] | tile_q0[n_tiles] |
        # split[n_tiles][3] | groups[base_tiles][4], each part starting on a 16-byte boundary
        def up4(n):
            return (n + 3) // 4 * 4
        o_pos, o_cu = 0, up4(rows)
        o_ctx = o_cu + up4(S + 1)
        o_ts = o_ctx + up4(S)
        o_tq = o_ts + up4(n_tiles)
        o_sp = o_tq + up4(n_tiles)
        o_gr = o_sp + up4(3 * n_tiles)
        n32 = o_gr + up4(4 * base_tiles)
        # int64: ids[rows] | slots[rows] | logits_rows[S]
        n64 = 2 * rows + S
        d32 = torch.zeros(n32, dtype=torch.int32, device=dev)
        d64 = torch.zeros(n64, dtype=torch.int64, device=dev)
        h32 = torch.zeros(n32, dtype=torch.int32)
        h64 = torch.zeros(n64, dtype=torch.int64)
        hbt = torch.zeros((S, mp), dtype=torch.int32)
        if self._pin:
            h32, h64, hbt = h32.pin_memory(), h64.pin_memory(), hbt.pin_memory()
        h32[o_ts:o_ts + n_tiles] = -1
        h64[rows:2 * rows] = -1
        d32.copy_(h32)
        d64.copy_(h64)
        bt = torch.zeros((S, mp), dtype=torch.int32, device=dev)
        st = {"rows": rows, "tile": tile, "n_tiles": n_tiles, "d32": d32, "d64": d64, "h32": h32, "h64": h64,
              "hbt": hbt, "bt": bt, "n32": h32.numpy(), "n64": h64.numpy(), "nbt": hbt.numpy(),
              "off": (o_pos, o_cu, o_ctx, o_ts, o_tq), "o_sp": o_sp, "o_gr": o_gr, "base_tiles": base_tiles,
              "max_records": max_records, "copied": torch.cuda.Event()}
        split_ws = None
        if max_records:
            split_ws = torch.empty(max_records * ops.prefill_split_record_bytes(self.model.n_q, self.model.n_kv) // 4,
                                   dtype=torch.float32, device=dev)
            for i in range(n_tiles):  # (first key tile, end, record): whole range, no record
                h32[o_sp + 3 * i + 1] = 1 << 30
                h32[o_sp + 3 * i + 2] = -1
            d32.copy_(h32)
        ids = d64[:rows]
        meta = BatchMeta(positions=d32[o_pos:o_pos + rows], slots=d64[rows:2 * rows], num_decode=0,
                         num_prefill_seqs=S, pre_block_tables=bt, pre_cu_q=d32[o_cu:o_cu + S + 1],
                         pre_ctx_lens=d32[o_ctx:o_ctx + S], pre_tile_seq=d32[o_ts:o_ts + n_tiles],
                         pre_tile_q0=d32[o_tq:o_tq + n_tiles], pre_tile_rows=tile,
                         logits_rows=d64[2 * rows:2 * rows + S], pre_padded=True)
        if max_records:
            meta.pre_split = d32[o_sp:o_sp + 3 * n_tiles]
            meta.pre_split_groups = d32[o_gr:o_gr + 4 * base_tiles]
            meta.pre_split_ws = split_ws
            st["split_ws"] = split_ws

        def run():
            hidden = self.model.forward(ids, self.pool, meta)
            logits = self.model.compute_logits(hidden)
            return logits, ops.argmax(logits)

        s = torch.cuda.Stream(device=dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s):
            run()
        torch.cuda.current_stream(dev).wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, pool=self._graph_pool):
            logits, toks = run()
        if self._graph_pool is None:
            self._graph_pool = g.pool()
        st.update(graph=g, logits=logits, toks=toks, meta=meta)
        return st

    def _launch_prefill_graph(self, prefills: list[ScheduledSeq]):
        """Fill the graph inputs of the smallest row bucket that holds these sequences and replay;
        returns (state, sampling sequences in logits-row order)."""
        total = sum(s.n for s in prefills)
        rows = next(b for b in _PREFILL_ROW_BUCKETS if b >= total)
        st = self._prefill_graphs.get(rows)
        if st is None:
            t0 = time.time()
            st = self._prefill_graphs[rows] = self._capture_prefill(rows)
            log.info("captured prefill graph for %d rows in %.2fs", rows, time.time() - t0)
        S = _PREFILL_GRAPH_SEQS
        o_pos, o_cu, o_ctx, o_ts, o_tq = st["off"]
        n32, n64, nbt = st["n32"], st["n64"], st["nbt"]
        st["copied"].synchronize()  # the previous replay's copies have left the pinned twins
        n64[rows:2 * rows] = -1
        n64[2 * rows:] = 0
        n32[o_cu:o_cu + S + 1] = 0
        n32[o_ctx:o_ctx + S] = 0
        n32[o_ts:o_ts + st["n_tiles"]] = -1
        row = 0
        sample_seqs: list[ScheduledSeq] = []
        q_lens, ctx = [], []
        for i, s in enumerate(prefills):
            r = s.req
            a, n = s.start, s.n
            n64[row:row + n] = r.prompt_ids[a:a + n] if a + n <= r.prompt_len else r.all_ids[a:a + n]
            t = np.arange(a, a + n, dtype=np.int64)
            n32[o_pos + row:o_pos + row + n] = t
            first = a // PAGE
            pg = np.asarray(r.pages[first:(a + n - 1) // PAGE + 1], dtype=np.int64)
            n64[rows + row:rows + row + n] = pg[t // PAGE - first] * PAGE + t % PAGE
            k = len(r.pages)
            nbt[i, :k] = r.pages
            row += n
            n32[o_cu + i + 1] = row
            n32[o_ctx + i] = a + n
            q_lens.append(n)
            ctx.append(a + n)
            if s.samples:
                n64[2 * rows + len(sample_seqs)] = row - 1
                sample_seqs.append(s)
        n32[o_cu + len(prefills) + 1:o_cu + S + 1] = row  # the unused sequences are empty
        ts, tq = ops.build_prefill_tiles(q_lens, self.model.n_q, self.model.n_kv, st["tile"], ctx)
        if st["max_records"]:
            o_sp, o_gr, nt = st["o_sp"], st["o_gr"], st["n_tiles"]
            sp = n32[o_sp:o_sp + 3 * nt].reshape(nt, 3)
            sp[:, 0], sp[:, 1], sp[:, 2] = 0, 1 << 30, -1
            n32[o_gr:o_gr + 4 * st["base_tiles"]] = 0
            cut = ops.plan_prefill_splits(ts, tq, q_lens, ctx, self.model.n_q, self.model.n_kv, st["tile"],
                                          max_entries=nt, max_records=st["max_records"])
            if cut is not None:
                ts, tq, split, groups, _n_rec = cut
                n32[o_sp:o_sp + len(split)] = split
                n32[o_gr:o_gr + len(groups)] = groups
        n32[o_ts:o_ts + len(ts)] = ts
        n32[o_tq:o_tq + len(tq)] = tq
        st["d64"].copy_(st["h64"], non_blocking=True)
        st["d32"].copy_(st["h32"], non_blocking=True)
        st["bt"].copy_(st["hbt"], non_blocking=True)
        st["copied"].record()
        st["graph"].replay()
        return st, sample_seqs

    def _execute_graph_pair(self, decodes: list[ScheduledSeq], prefills: list[ScheduledSeq]):
        """A step whose decode rows and (small) prefill part each replay a captured graph: two
        launches instead of some four hundred, so the decode rows of a step that admits a new turn
        do not wait for the host to issue the prefill's kernels one by one.  Both graphs are
        launched before anything is read back."""
        t0 = time.perf_counter()
        dst = self._launch_decode_graph(decodes) if decodes else None
        pst, pre_samples = self._launch_prefill_graph(prefills)
        dt = time.perf_counter() - t0
        self.t_launch += dt
        self.t_launch_graph_pairs += dt
        out = []
        if dst is not None:
            toks, infos = self.sampler.sample(dst["logits"], [s.req for s in decodes], dst["toks"])
            out += list(zip(decodes, toks, infos))
        if pre_samples:
            k = len(pre_samples)
            toks, infos = self.sampler.sample(pst["logits"][:k], [s.req for s in pre_samples], pst["toks"][:k])
            out += list(zip(pre_samples, toks, infos))
        return out

    @staticmethod
    def _decode_plan_to_ints(plan: dict) -> list[int]:
        """[bsz, n_deltas, ids.., pos.., slots.., ctx.., (row, lo, count, pages..)..]"""
        out = [plan["bsz"], len(plan["bt"])]
        out += plan["ids"] + plan["pos"] + plan["slots"] + plan["ctx"]
        for row, lo, pages in plan["bt"]:
            out += [row, lo, len(pages)]
            out += pages
        return out

    @staticmethod
    def _decode_plan_from_ints(v: list[int]) -> dict:
        bsz, nd = v[0], v[1]
        cut = [2 + k * bsz for k in range(5)]
        plan = {"bsz": bsz, "ids": v[cut[0]:cut[1]], "pos": v[cut[1]:cut[2]],
                "slots": v[cut[2]:cut[3]], "ctx": v[cut[3]:cut[4]], "bt": []}
        p = cut[4]
        for _ in range(nd):
            row, lo, cnt = v[p:p + 3]
            plan["bt"].append((row, lo, v[p + 3:p + 3 + cnt]))
            p += 3 + cnt
        return plan

    def _replay_decode(self, plan: dict) -> dict:
        """Every rank of the replica: write the plan into this rank's graph inputs and replay."""
        bsz = plan["bsz"]
        with_lora = "lora" in plan
        lora = self.model.lora
        # graphs with adapter kernels are keyed apart and follow the adapter set: which
        # (projection, layer) pairs launch LoRA kernels is decided on the host at capture time
        key = (bsz, True) if with_lora else bsz
        st = self._graphs.get(key)
        if st is not None and with_lora and st["lora_version"] != lora.version:
            st = None
        if st is None:
            t0 = time.time()
            st = self._capture(bsz, with_lora)
            self._graphs[key] = st
            log.info("captured decode graph for batch %d%s in %.2fs", bsz,
                     " (LoRA)" if with_lora else "", time.time() - t0)
        nbt = st["nbt"]
        for row, lo, pages in plan["bt"]:
            nbt[row, lo:lo + len(pages)] = pages
        n64, n32 = st["n64"], st["n32"]
        n64[0], n64[1], n32[0], n32[1] = plan["ids"], plan["slots"], plan["pos"], plan["ctx"]
        st["d64"].copy_(st["h64"], non_blocking=True)
        st["d32"].copy_(st["h32"], non_blocking=True)
        if plan["bt"]:
            st["bt"].copy_(st["hbt"], non_blocking=True)
        if with_lora:
            st["lora"].copy_(self._to_dev(plan["lora"], torch.int32), non_blocking=True)
        st["graph"].replay()
        return st

    def capture_all_graphs(self) -> int:
        """Capture the decode graph of every batch bucket up to ``max_num_seqs`` (largest first:
        the shared graph memory pool is then sized once).  Returns the number captured."""
        if not self.use_graphs:
            return 0
        todo = [b for b in _GRAPH_BUCKETS if b <= self._bucket(min(self.cfg.max_num_seqs, _GRAPH_BUCKETS[-1]))
                and b not in self._graphs]
        for b in sorted(todo, reverse=True):
            self._graphs[b] = self._capture(b)
        n = len(todo)
        if not self.distributed:
            for rows in reversed(_PREFILL_ROW_BUCKETS):  # the small prefill parts of mixed steps
                if rows <= self.cfg.max_num_batched_tokens and rows not in self._prefill_graphs:
                    self._prefill_graphs[rows] = self._capture_prefill(rows)
                    n += 1
        return n

    # ------------------------------------------------------------------ sleep / wake
    def sleep(self, level: int = 1) -> int:
        """Release HBM: weights to pinned host memory (level 1) or dropped (level 2); the KV
        pool is freed.  Returns bytes released (this rank)."""
        if self.sleeping:
            return 0
        if self.distributed:
            replica_broadcast(("sleep", level))
        return self._sleep_local(level)

    def _sleep_local(self, level: int = 1) -> int:
        freed = 0
        self._graphs.clear()
        self._prefill_graphs.clear()
        self._row_track.clear()
        self._graph_pool = None
        if self.device.type == "cuda":
            if hasattr(self.model, "drop_packed"):
                packed_bytes = sum(t.numel() * t.element_size() for t in self.model.packed_tensors())
                self.model.drop_packed()  # derived copies: rebuilt from the weights on wake-up
                freed += packed_bytes
            for t in self.model.weight_tensors():
                host = None
                if level == 1:
                    host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
                    host.copy_(t, non_blocking=True)
                self._sleep_stash.append((t, host))
                freed += t.numel() * t.element_size()
            torch.cuda.synchronize()
            for t, _ in self._sleep_stash:
                t.set_(torch.empty(0, dtype=t.dtype, device=self.device).untyped_storage(), 0, (0,))
        freed += self.pool.numel() * self.pool.element_size()
        self._pool_shape = tuple(self.pool.shape)
        self.pool = torch.empty(0, dtype=self.dtype, device=self.device)
        for fn in self.pool_listeners:  # nobody may keep a view of memory that is about to go
            fn(self.pool)
        self._free_exported_pool()  # peers re-map the new allocation (new handle) after the wake-up
        if self.device.type == "cuda":
            torch.cuda.empty_cache()
        self.sleeping = True
        return freed

    def wake_up(self) -> None:
        if not self.sleeping:
            return
        if self.distributed:
            replica_broadcast(("wake",))
        self._wake_local()

    def _wake_local(self) -> None:
        if self.device.type == "cuda":
            for t, host in self._sleep_stash:
                shape = host.shape if host is not None else None
                if host is not None:
                    fresh = host.to(self.device, non_blocking=True)
                    t.set_(fresh.untyped_storage(), 0, shape, fresh.stride())
            torch.cuda.synchronize()
            if any(h is None for _, h in self._sleep_stash):
                self.model.init_random(self.cfg.seed)
            self._pack_decode_weights()
        self._sleep_stash = []
        self.pool = self._alloc_pool()
        for fn in self.pool_listeners:  # whoever keeps the pool tensor (KV tiers) takes the new one
            fn(self.pool)
        self.sleeping = False
        if self.cfg.capture_graphs_at_start:
            self.capture_all_graphs()
