"""Decoder-only transformer (Llama-3 family and Mixtral MoE) over the paged KV pool.

Functional style: weights are plain tensors created directly on the target device (bf16)
and every hot op goes through ``ops`` (hand-written gfx950 kernels on the GPU); the dense
projections are library GEMMs on the matrix cores (``F.linear`` -> hipBLASLt).
Tensor parallelism: QKV / gate-up are column-sharded by head / channel, O / down are
row-sharded and followed by one all-reduce each; the LM head is vocab-sharded.
"""

from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from .. import ops
from ..engine.config import ModelConfig
from ..parallel.state import get_tp, tp_all_gather_last_dim, tp_all_reduce
from .batch import BatchMeta


class LayerWeights:
    __slots__ = ("ln1", "wqkv", "wo", "ln2", "w_gate_up", "w_down", "router", "w13", "w2")

    def __init__(self):
        for s in self.__slots__:
            setattr(self, s, None)


class TransformerModel:
    def __init__(self, cfg: ModelConfig, dtype: torch.dtype, device: torch.device,
                 max_model_len: int):
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
        self.vocab_shard = (cfg.vocab_size + tp.size - 1) // tp.size
        self.scale = 1.0 / math.sqrt(cfg.head_dim)
        self.layers: list[LayerWeights] = []
        self.embed = None
        self.norm = None
        self.lm_head = None
        self.cos_sin = ops.build_cos_sin(cfg.head_dim, max_model_len, cfg.rope_theta,
                                         cfg.rope_scaling, device=device)
        self.lora = None  # LoRAManager, attached by the runner when --enable-lora

    # ------------------------------------------------------------------ weights
    def init_random(self, seed: int = 0) -> None:
        """Random-init weights of the exact architecture (no checkpoint needed)."""
        cfg = self.cfg
        g = torch.Generator(device=self.device).manual_seed(seed + 1000 * self.tp_rank)
        gs = torch.Generator(device=self.device).manual_seed(seed)  # replicated tensors

        def rnd(*shape, std=0.02, gen=g):
            t = torch.empty(shape, dtype=self.dtype, device=self.device)
            t.normal_(0.0, std, generator=gen)
            return t

        h, d = cfg.hidden_size, cfg.head_dim
        self.embed = rnd(cfg.vocab_size, h, gen=gs)
        for _ in range(cfg.num_layers):
            lw = LayerWeights()
            lw.ln1 = torch.ones(h, dtype=self.dtype, device=self.device)
            lw.ln2 = torch.ones(h, dtype=self.dtype, device=self.device)
            lw.wqkv = rnd((self.n_q + 2 * self.n_kv) * d, h)
            lw.wo = rnd(h, self.n_q * d)
            if cfg.num_experts > 0:
                lw.router = rnd(cfg.num_experts, h, gen=gs)
                lw.w13 = rnd(cfg.num_experts, 2 * self.inter, h)
                lw.w2 = rnd(cfg.num_experts, h, self.inter)
            else:
                lw.w_gate_up = rnd(2 * self.inter, h)
                lw.w_down = rnd(h, self.inter)
            self.layers.append(lw)
        self.norm = torch.ones(h, dtype=self.dtype, device=self.device)
        if cfg.tie_word_embeddings and self.tp_size == 1:
            self.lm_head = self.embed
        else:
            self.lm_head = rnd(self.vocab_shard, h)

    def weight_tensors(self):
        yield self.embed
        yield self.norm
        if self.lm_head is not self.embed:
            yield self.lm_head
        for lw in self.layers:
            for s in LayerWeights.__slots__:
                t = getattr(lw, s)
                if t is not None:
                    yield t

    def weight_bytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self.weight_tensors())

    # ------------------------------------------------------------------ forward
    def _attention(self, qkv: torch.Tensor, pool: torch.Tensor, layer: int, meta: BatchMeta):
        cfg = self.cfg
        nq, nkv, d = self.n_q, self.n_kv, self.head_dim
        ops.rope_and_cache(qkv, meta.positions, meta.slots, self.cos_sin, pool, layer, nq, nkv, d)
        t = qkv.shape[0]
        out = torch.empty((t, nq * d), dtype=qkv.dtype, device=qkv.device)
        nd = meta.num_decode
        q = qkv[:, : nq * d]
        if nd > 0:
            ops.paged_decode(q[:nd], out[:nd], pool, layer, meta.dec_block_tables,
                             meta.dec_ctx_lens, nq, nkv, d, self.scale, meta.dec_num_splits,
                             meta.dec_ws_o, meta.dec_ws_ml)
        if meta.num_prefill_seqs > 0:
            ops.paged_prefill(q[nd:], out[nd:], pool, layer, meta.pre_block_tables, meta.pre_cu_q,
                              meta.pre_ctx_lens, meta.pre_tile_seq, meta.pre_tile_q0, nq, nkv, d,
                              self.scale)
        del cfg
        return out

    def _moe(self, x: torch.Tensor, lw: LayerWeights, decode_only: bool) -> torch.Tensor:
        cfg = self.cfg
        t = x.shape[0]
        e, k = cfg.num_experts, cfg.num_experts_per_tok
        logits = F.linear(x, lw.router)
        w, ids = ops.moe_topk_softmax(logits, k)
        if decode_only or t <= 16:
            # Small batches: every expert's weights are streamed anyway, so run all experts
            # on all rows (no host sync: HIP-graph capturable) and mix with the routing matrix.
            dense = torch.zeros((t, e), dtype=torch.float32, device=x.device)
            dense.scatter_(1, ids.long(), w)
            gu = torch.matmul(x.unsqueeze(0), lw.w13.transpose(1, 2))  # [E, T, 2I]
            act = ops.silu_and_mul(gu.view(e * t, -1)).view(e, t, -1)
            y = torch.matmul(act, lw.w2.transpose(1, 2))  # [E, T, H]
            return (y.float() * dense.t().unsqueeze(-1)).sum(0).to(x.dtype)
        offsets, pos, sorted_tok = ops.moe_align(ids, e)
        xs = ops.moe_permute(x, sorted_tok)
        off = offsets.tolist()
        ys = torch.empty((t * k, cfg.hidden_size), dtype=x.dtype, device=x.device)
        for ex in range(e):
            a, b = off[ex], off[ex + 1]
            if b > a:
                gu = F.linear(xs[a:b], lw.w13[ex])
                ys[a:b] = F.linear(ops.silu_and_mul(gu), lw.w2[ex])
        return ops.moe_combine(ys, w, pos, t)

    def _lora(self, name: str, layer: int, x: torch.Tensor, y: torch.Tensor, meta: BatchMeta):
        if self.lora is not None and meta.lora_slots is not None:
            self.lora.apply(name, layer, x, y, meta.lora_slots)
        return y

    def forward(self, input_ids: torch.Tensor, pool: torch.Tensor, meta: BatchMeta) -> torch.Tensor:
        """Returns the final-normed hidden states of ``meta.logits_rows`` ([num_out, H])."""
        cfg = self.cfg
        h = F.embedding(input_ids, self.embed)
        residual = None
        decode_only = meta.num_prefill_seqs == 0
        for li, lw in enumerate(self.layers):
            if residual is None:
                residual = h
                x = ops.rms_norm(h, lw.ln1, cfg.norm_eps)
            else:
                x = ops.rms_norm(h, lw.ln1, cfg.norm_eps, residual)
            qkv = self._lora("qkv", li, x, F.linear(x, lw.wqkv), meta)
            attn = self._attention(qkv, pool, li, meta)
            h = self._lora("o", li, attn, F.linear(attn, lw.wo), meta)
            h = tp_all_reduce(h)
            x = ops.rms_norm(h, lw.ln2, cfg.norm_eps, residual)
            if cfg.num_experts > 0:
                h = self._moe(x, lw, decode_only)
            else:
                gu = F.linear(x, lw.w_gate_up)
                h = F.linear(ops.silu_and_mul(gu), lw.w_down)
            h = tp_all_reduce(h)
        if meta.logits_rows is not None:
            h = h.index_select(0, meta.logits_rows)
            residual = residual.index_select(0, meta.logits_rows)
        return ops.rms_norm(h, self.norm, cfg.norm_eps, residual)

    def compute_logits(self, hidden: torch.Tensor) -> torch.Tensor:
        logits = F.linear(hidden, self.lm_head)
        if self.tp_size > 1:
            logits = tp_all_gather_last_dim(logits)
        return logits[:, : self.cfg.vocab_size]
