"""OPT / GPT-2 decoders (facebook/opt-125m, distilgpt2 class): learned positions, pre-LayerNorm,
biased projections, ReLU or tanh-GELU MLP.

Used for the CPU plumbing configuration (router in front of small CPU replicas); it shares
the paged KV pool and the scheduler with the Llama path and runs the PyTorch op
definitions when its tensors live on the CPU.
"""

from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from .. import ops
from ..engine.config import ModelConfig
from ..ops import reference
from .batch import BatchMeta


class OPTModel:
    def __init__(self, cfg: ModelConfig, dtype: torch.dtype, device: torch.device,
                 max_model_len: int):
        self.cfg = cfg
        self.dtype = dtype
        self.device = device
        self.n_q = cfg.num_heads
        self.n_kv = cfg.num_kv_heads
        self.head_dim = cfg.head_dim
        self.scale = 1.0 / math.sqrt(cfg.head_dim)
        self.tp_size = 1
        self.layers: list[dict[str, torch.Tensor]] = []
        self.embed = None
        self.pos_embed = None
        self.final_ln = None
        self.cos_sin = torch.zeros((1, cfg.head_dim), dtype=torch.float32, device=device)
        self.max_model_len = max_model_len
        self.lora = None

    def init_random(self, seed: int = 0) -> None:
        cfg = self.cfg
        g = torch.Generator(device=self.device).manual_seed(seed)

        def rnd(*shape, std=0.02):
            t = torch.empty(shape, dtype=self.dtype, device=self.device)
            t.normal_(0.0, std, generator=g)
            return t

        def zeros(*shape):
            return torch.zeros(shape, dtype=self.dtype, device=self.device)

        def ones(*shape):
            return torch.ones(shape, dtype=self.dtype, device=self.device)

        h, f = cfg.hidden_size, cfg.ffn_dim or cfg.intermediate_size
        self.embed = rnd(cfg.vocab_size, h)
        self.pos_embed = rnd(max(self.max_model_len, 1) + cfg.pos_offset, h)
        for _ in range(cfg.num_layers):
            self.layers.append({
                "ln1_w": ones(h), "ln1_b": zeros(h), "wqkv": rnd(3 * h, h), "bqkv": zeros(3 * h),
                "wo": rnd(h, h), "bo": zeros(h), "ln2_w": ones(h), "ln2_b": zeros(h),
                "fc1": rnd(f, h), "b1": zeros(f), "fc2": rnd(h, f), "b2": zeros(h),
            })
        self.final_ln = (ones(h), zeros(h))

    def weight_tensors(self):
        yield self.embed
        yield self.pos_embed
        yield from self.final_ln
        for lw in self.layers:
            yield from lw.values()

    def weight_bytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self.weight_tensors())

    def forward(self, input_ids: torch.Tensor, pool: torch.Tensor, meta: BatchMeta) -> torch.Tensor:
        cfg = self.cfg
        nq, nkv, d = self.n_q, self.n_kv, self.head_dim
        h = F.embedding(input_ids, self.embed) + F.embedding(
            meta.positions.long() + cfg.pos_offset, self.pos_embed)
        nd = meta.num_decode
        for li, lw in enumerate(self.layers):
            x = reference.layer_norm(h, lw["ln1_w"], lw["ln1_b"], 1e-5)
            qkv = F.linear(x, lw["wqkv"], lw["bqkv"])
            out = ops.attention_block(qkv, pool, li, meta, nq, nkv, d, self.scale, self.cos_sin, rotary=False)
            h = h + F.linear(out, lw["wo"], lw["bo"])
            x = reference.layer_norm(h, lw["ln2_w"], lw["ln2_b"], 1e-5)
            a = F.linear(x, lw["fc1"], lw["b1"])
            a = F.gelu(a, approximate="tanh") if cfg.activation == "gelu_new" else F.relu(a)
            h = h + F.linear(a, lw["fc2"], lw["b2"])
        if meta.logits_rows is not None:
            h = h.index_select(0, meta.logits_rows)
        return reference.layer_norm(h, self.final_ln[0], self.final_ln[1], 1e-5)

    def compute_logits(self, hidden: torch.Tensor) -> torch.Tensor:
        return F.linear(hidden, self.embed)
