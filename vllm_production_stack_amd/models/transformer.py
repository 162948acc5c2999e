"""Decoder-only transformer (Llama-3 family and Mixtral MoE) over the paged KV pool.

Functional style: weights are plain tensors created directly on the target device (bf16)
and every hot op goes through ``ops`` (hand-written gfx950 kernels on the GPU); the dense
projections are library GEMMs on the matrix cores (``F.linear`` -> hipBLASLt), except the
small-batch decode projections that ``ops.linear`` / ``ops.gate_up_silu`` route to the
weight-streaming MFMA kernels of ``csrc/kernels/gemm.hip``.
Tensor parallelism: QKV / gate-up are column-sharded by head / channel, O / down are
row-sharded and followed by one all-reduce each; the LM head is vocab-sharded.
Pipeline parallelism: a stage holds a contiguous run of layers (and the KV pool slices of
exactly those layers); the residual stream (h, residual) moves stage to stage point to point;
the first stage owns the embedding and the LM head, the last stage the final norm.
"""

from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from .. import ops
from ..engine.config import ModelConfig
from ..parallel.state import (get_pp, get_tp, pp_recv, pp_send, tp_all_gather_last_dim,
                              tp_all_reduce)
from .batch import BatchMeta


class LayerWeights:
    __slots__ = ("ln1", "wqkv", "bqkv", "wo", "ln2", "w_gate_up", "w_down", "router", "w13", "w2",
                 "w_gate_up_p", "w_down_p")  # *_p: the MLP matrices in matrix-core tile order (decode steps)

    def __init__(self):
        for s in self.__slots__:
            setattr(self, s, None)


class TransformerModel:
    def __init__(self, cfg: ModelConfig, dtype: torch.dtype, device: torch.device,
                 max_model_len: int, expert_parallel: bool = False):
        self.cfg = cfg
        self.dtype = dtype
        self.device = device
        tp = get_tp()
        self.tp_size = tp.size
        self.tp_rank = tp.rank
        assert cfg.num_heads % tp.size == 0 and cfg.num_kv_heads % tp.size == 0, \
            "tensor parallel size must divide the head counts"
        self.n_q = cfg.num_heads // tp.size
        self.n_kv = cfg.num_kv_heads // tp.size
        self.head_dim = cfg.head_dim
        self.inter = cfg.intermediate_size // tp.size
        # Expert parallelism inside the tensor-parallel group: a rank holds whole experts (a
        # contiguous range of them) instead of a slice of every expert.  Attention stays tensor
        # parallel, so every rank sees every token; it runs the pairs routed to its own experts and
        # the all-reduce that closes the layer sums the ranks' contributions, as it does for slices.
        self.ep = bool(expert_parallel) and tp.size > 1 and cfg.num_experts > 0
        if self.ep:
            assert cfg.num_experts % tp.size == 0, "expert parallelism needs the rank count to divide the experts"
        self.local_experts = cfg.num_experts // tp.size if self.ep else cfg.num_experts
        self.first_expert = tp.rank * self.local_experts if self.ep else 0
        self.moe_inter = cfg.intermediate_size if self.ep else self.inter
        self.vocab_shard = (cfg.vocab_size + tp.size - 1) // tp.size
        self.scale = 1.0 / math.sqrt(cfg.head_dim)
        self.layers: list[LayerWeights] = []
        self.embed = None
        self.norm = None
        self.lm_head = None
        self.cos_sin = ops.build_cos_sin(cfg.head_dim, max_model_len, cfg.rope_theta,
                                         cfg.rope_scaling, device=device)
        self.lora = None  # LoRAManager, attached by the runner when --enable-lora
        pp = get_pp()
        self.pp = pp
        self.first_layer, self.last_layer = pp.layer_range(cfg.num_layers)

    # ------------------------------------------------------------------ weights
    def _rand(self, seed: int, key: int, *shape, std: float = 0.02) -> torch.Tensor:
        """One full (unsharded) random tensor, a function of (seed, key) only."""
        g = torch.Generator(device=self.device).manual_seed((seed * 1000003 + key) & 0x7FFFFFFF)
        t = torch.empty(shape, dtype=self.dtype, device=self.device)
        t.normal_(0.0, std, generator=g)
        return t

    def init_random(self, seed: int = 0) -> None:
        """Random-init weights of the exact architecture (no checkpoint needed).

        Every tensor is drawn unsharded from its own generator and then cut to this rank's
        tensor-parallel shard / pipeline stage, so the model is the same function whatever the
        parallel layout (that is what the multi-process equivalence tests compare)."""
        cfg = self.cfg
        h, d = cfg.hidden_size, cfg.head_dim
        r, sz = self.tp_rank, self.tp_size
        nq, nkv, inter = self.n_q, self.n_kv, self.inter

        def rows(t, n):  # n rows per rank
            return t[r * n:(r + 1) * n]

        self.layers = []
        self.embed = self._rand(seed, 1, cfg.vocab_size, h) if self.pp.is_first else None
        for li in range(self.first_layer, self.last_layer):
            k = 64 * (li + 1)
            lw = LayerWeights()
            lw.ln1 = torch.ones(h, dtype=self.dtype, device=self.device)
            lw.ln2 = torch.ones(h, dtype=self.dtype, device=self.device)
            wq = self._rand(seed, k + 0, cfg.num_heads * d, h)
            wk = self._rand(seed, k + 1, cfg.num_kv_heads * d, h)
            wv = self._rand(seed, k + 2, cfg.num_kv_heads * d, h)
            lw.wqkv = torch.cat([rows(wq, nq * d), rows(wk, nkv * d), rows(wv, nkv * d)]).contiguous()
            del wq, wk, wv
            lw.bqkv = None
            if cfg.qkv_bias:
                bq = self._rand(seed, k + 40, cfg.num_heads * d)
                bk = self._rand(seed, k + 41, cfg.num_kv_heads * d)
                bv = self._rand(seed, k + 42, cfg.num_kv_heads * d)
                lw.bqkv = torch.cat([rows(bq, nq * d), rows(bk, nkv * d), rows(bv, nkv * d)]).contiguous()
            wo = self._rand(seed, k + 3, h, cfg.num_heads * d)
            lw.wo = wo[:, r * nq * d:(r + 1) * nq * d].contiguous()
            del wo
            if cfg.num_experts > 0:
                lw.router = self._rand(seed, k + 4, cfg.num_experts, h)
                w13, w2 = [], []
                for ex in range(self.first_expert, self.first_expert + self.local_experts):
                    g_ = self._rand(seed, k + 8 + 3 * ex, cfg.intermediate_size, h)
                    u_ = self._rand(seed, k + 9 + 3 * ex, cfg.intermediate_size, h)
                    d_ = self._rand(seed, k + 10 + 3 * ex, h, cfg.intermediate_size)
                    if self.ep:
                        w13.append(torch.cat([g_, u_]))
                        w2.append(d_)
                    else:
                        w13.append(torch.cat([rows(g_, inter), rows(u_, inter)]))
                        w2.append(d_[:, r * inter:(r + 1) * inter])
                    del g_, u_, d_
                lw.w13 = torch.stack(w13).contiguous()
                lw.w2 = torch.stack(w2).contiguous()
            else:
                g_ = self._rand(seed, k + 4, cfg.intermediate_size, h)
                u_ = self._rand(seed, k + 5, cfg.intermediate_size, h)
                lw.w_gate_up = torch.cat([rows(g_, inter), rows(u_, inter)]).contiguous()
                del g_, u_
                d_ = self._rand(seed, k + 6, h, cfg.intermediate_size)
                lw.w_down = d_[:, r * inter:(r + 1) * inter].contiguous()
                del d_
            self.layers.append(lw)
        self.norm = torch.ones(h, dtype=self.dtype, device=self.device) if self.pp.is_last else None
        if not self.pp.is_first:
            self.lm_head = None
        elif cfg.tie_word_embeddings and sz == 1:
            self.lm_head = self.embed
        else:
            full = self.embed if cfg.tie_word_embeddings else self._rand(seed, 2, cfg.vocab_size, h)
            shard = torch.zeros((self.vocab_shard, h), dtype=self.dtype, device=self.device)
            part = full[r * self.vocab_shard:(r + 1) * self.vocab_shard]
            shard[: part.shape[0]].copy_(part)
            self.lm_head = shard

    def pack_decode_weights(self, budget_bytes: int) -> int:
        """Second copy of the dense MLP matrices in matrix-core tile order (``ops.pack_weight``),
        which decode-only steps stream straight into MFMA operands (``csrc/kernels/packed_gemm.hip``:
        the down projection 27.9 against 39.4 us and the fused gate/up 49.6 against 56.8 us at 40
        rows of Llama-3-8B).  Only when both copies of every layer fit ``budget_bytes``.  Returns the
        bytes used (0 = not packed)."""
        cfg = self.cfg
        self.drop_packed()
        if (self.device.type != "cuda" or self.dtype != torch.bfloat16 or cfg.num_experts > 0
                or not self.layers or self.layers[0].w_gate_up is None):
            return 0
        gu, dn = self.layers[0].w_gate_up, self.layers[0].w_down
        if not (ops.packed_supported(gu.shape[0], gu.shape[1], True)
                and ops.packed_supported(dn.shape[0], dn.shape[1], False)):
            return 0
        need = len(self.layers) * (gu.numel() + dn.numel()) * gu.element_size()
        if need > budget_bytes:
            return 0
        for lw in self.layers:
            lw.w_gate_up_p = ops.pack_weight(lw.w_gate_up)
            lw.w_down_p = ops.pack_weight(lw.w_down)
        return need

    def weight_tensors(self):
        if self.embed is not None:
            yield self.embed
        if self.norm is not None:
            yield self.norm
        if self.lm_head is not None and self.lm_head is not self.embed:
            yield self.lm_head
        for lw in self.layers:
            for s in LayerWeights.__slots__:
                t = getattr(lw, s)
                if t is not None and not s.endswith("_p"):  # packed copies are derived, not weights
                    yield t

    def packed_tensors(self):
        for lw in self.layers:
            for t in (lw.w_gate_up_p, lw.w_down_p):
                if t is not None:
                    yield t

    def drop_packed(self) -> None:
        for lw in self.layers:
            lw.w_gate_up_p = lw.w_down_p = None

    def weight_bytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self.weight_tensors())

    # ------------------------------------------------------------------ forward
    def _attention(self, qkv: torch.Tensor, pool: torch.Tensor, layer: int, meta: BatchMeta):
        return ops.attention_block(qkv, pool, layer, meta, self.n_q, self.n_kv, self.head_dim, self.scale,
                                   self.cos_sin, rotary=True)

    def _moe(self, x: torch.Tensor, lw: LayerWeights, decode_only: bool) -> torch.Tensor:
        cfg = self.cfg
        t = x.shape[0]
        e, k = self.local_experts, cfg.num_experts_per_tok
        logits = F.linear(x, lw.router)
        w, ids = ops.moe_topk_softmax(logits, k)
        mfma = x.is_cuda and t <= ops.MOE_MFMA_MAX_TOKENS and ops.moe_mfma_supported(x.shape[1], lw.w2.shape[2])
        tiled = (x.is_cuda and not (decode_only or t <= 16)
                 and ops.grouped_gemm_supported(2 * lw.w2.shape[2], x.shape[1], True)
                 and ops.grouped_gemm_supported(x.shape[1], lw.w2.shape[2]))
        ep = self.ep
        if ep:
            # pairs routed to another rank's experts: the sorting kernels put them in a bucket past
            # the last local expert that no GEMM visits; everywhere else they run on expert 0 with
            # weight 0 (correct, and only taken on the CPU and on shapes the kernels do not tile)
            ids, w = ops.moe_localize(ids, w, self.first_expert, e, bucket=mfma or tiled)
        if decode_only and t <= 2 and x.is_cuda and not ep:
            # one or two tokens rarely share an expert: per-pair GEMV kernels, no sort
            return ops.moe_experts_small_batch(x, lw.w13, lw.w2, ids, w)
        if mfma:
            # Decode-sized batches: sort the (token, expert) pairs by expert and stream each
            # routed expert's weights once against all of its tokens (matrix-core kernels,
            # no host sync: HIP-graph capturable).
            return ops.moe_experts_grouped(x, lw.w13, lw.w2, ids, w, foreign=ep)
        if decode_only and t <= 16 and x.is_cuda:
            return ops.moe_experts_small_batch(x, lw.w13, lw.w2, ids, w)
        if decode_only or t <= 16:
            # Small batches on odd shapes / the CPU: run all experts on all rows (no host
            # sync) and mix with the routing matrix.
            dense = torch.zeros((t, e), dtype=torch.float32, device=x.device)
            dense.scatter_add_(1, ids.long(), w)
            gu = torch.matmul(x.unsqueeze(0), lw.w13.transpose(1, 2))  # [E, T, 2I]
            act = ops.silu_and_mul(gu.view(e * t, -1)).view(e, t, -1)
            y = torch.matmul(act, lw.w2.transpose(1, 2))  # [E, T, H]
            return (y.float() * dense.t().unsqueeze(-1)).sum(0).to(x.dtype)
        if tiled:
            # Prefill chunks of any size: two grouped matrix-core GEMMs over the device-sorted
            # (token, expert) pairs (gate_up gathers its rows and fuses SwiGLU).  The host never reads
            # the per-expert counts, so it keeps launching ahead of the GPU.
            return ops.moe_experts_prefill(x, lw.w13, lw.w2, ids, w, foreign=ep)
        # The CPU, and shapes the grouped kernel does not tile: per-expert products on the sorted rows
        offsets, pos, sorted_tok = ops.moe_align(ids, e)
        xs = ops.moe_permute(x, sorted_tok)
        off = offsets.tolist()
        ys = torch.empty((t * k, cfg.hidden_size), dtype=x.dtype, device=x.device)
        for ex in range(e):
            a, b = off[ex], off[ex + 1]
            if b > a:
                gu = F.linear(xs[a:b], lw.w13[ex])
                torch.matmul(ops.silu_and_mul(gu), lw.w2[ex].t(), out=ys[a:b])  # no staging copy
        return ops.moe_combine(ys, w, pos, t)

    def _out_proj(self, name: str, layer: int, x: torch.Tensor, w: torch.Tensor, packed, residual: torch.Tensor,
                  meta: BatchMeta):
        """An output projection (o, down) and its all-reduce.  Single-rank replicas let the prefill
        GEMM add the product straight into the residual stream (its epilogue) and return None;
        otherwise the product comes back for the fused add + norm kernel."""
        if self.tp_size == 1:
            h = ops.linear_add(x, w, residual, packed)
            if h is None:
                self._lora(name, layer, x, residual, meta)  # an adapter's delta lands in the stream as well
                return None
            return self._lora(name, layer, x, h, meta)
        return tp_all_reduce(self._lora(name, layer, x, ops.linear(x, w, None, packed), meta))

    def _lora(self, name: str, layer: int, x: torch.Tensor, y: torch.Tensor, meta: BatchMeta):
        if self.lora is not None and meta.lora_slots is not None:
            self.lora.apply(name, layer, x, y, meta.lora_slots)
        return y

    def forward(self, input_ids: torch.Tensor, pool: torch.Tensor, meta: BatchMeta) -> torch.Tensor:
        """Returns the final-normed hidden states of ``meta.logits_rows`` ([num_out, H]) on the
        last pipeline stage; earlier stages hand the residual stream on and return ``None``."""
        cfg = self.cfg
        pp = self.pp
        if pp.is_first:
            h = F.embedding(input_ids, self.embed)
            residual = None
        else:
            shape = (input_ids.shape[0], cfg.hidden_size)
            h = pp_recv(shape, self.dtype, self.device, pp.prev)
            residual = pp_recv(shape, self.dtype, self.device, pp.prev)
        decode_only = meta.num_prefill_seqs == 0
        for li, lw in enumerate(self.layers):
            gl = self.first_layer + li  # global layer index (adapters are keyed by it)
            if residual is None:
                residual = h
                x = ops.rms_norm(h, lw.ln1, cfg.norm_eps)
            elif h is None:  # the previous projection's epilogue already updated the residual stream
                x = ops.rms_norm(residual, lw.ln1, cfg.norm_eps)
            else:
                x = ops.rms_norm(h, lw.ln1, cfg.norm_eps, residual)
            qkv = self._lora("qkv", gl, x, ops.qkv_linear(x, lw.wqkv, lw.bqkv, self.n_q * self.head_dim), meta)
            attn = self._attention(qkv, pool, li, meta)
            h = self._out_proj("o", gl, attn, lw.wo, None, residual, meta)
            x = ops.rms_norm(residual, lw.ln2, cfg.norm_eps) if h is None else ops.rms_norm(h, lw.ln2, cfg.norm_eps, residual)
            if cfg.num_experts > 0:
                h = self._moe(x, lw, decode_only)
            else:
                if self.lora is not None and meta.lora_slots is not None:
                    gu = self._lora("gate_up", gl, x, F.linear(x, lw.w_gate_up), meta)
                    act = ops.silu_and_mul(gu)
                else:
                    act = ops.gate_up_silu(x, lw.w_gate_up, lw.w_gate_up_p if decode_only else None)
                h = self._out_proj("down", gl, act, lw.w_down, lw.w_down_p if decode_only else None, residual, meta)
            if cfg.num_experts > 0:
                h = tp_all_reduce(h)
        if not pp.is_last:
            if h is None:  # the last projection added itself into the residual stream
                h = torch.zeros_like(residual)
            pp_send(h, pp.next)
            pp_send(residual, pp.next)
            return None
        if meta.logits_rows is not None:
            h = None if h is None else h.index_select(0, meta.logits_rows)
            residual = residual.index_select(0, meta.logits_rows)
        if h is None:
            return ops.rms_norm(residual, self.norm, cfg.norm_eps)
        return ops.rms_norm(h, self.norm, cfg.norm_eps, residual)

    def compute_logits(self, hidden: torch.Tensor) -> torch.Tensor:
        """First-stage ranks only (they hold the LM head shards)."""
        logits = ops.linear(hidden, self.lm_head)
        if self.tp_size > 1:
            logits = tp_all_gather_last_dim(logits)
        return logits[:, : self.cfg.vocab_size]
