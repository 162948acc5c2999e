"""Model construction: random init (no checkpoint on disk) or HF safetensors loading.

Checkpoint tensors are streamed shard by shard straight into the fused / TP-sharded
layout the engine uses (QKV and gate|up fused, column shards by head, row shards by
channel), so the host never holds more than one checkpoint tensor at a time.
"""

from __future__ import annotations

import json
import re
from pathlib import Path

import torch

from ..engine.config import ModelConfig
from ..parallel.state import get_tp
from .opt import OPTModel
from .transformer import LayerWeights, TransformerModel


def build_model(cfg: ModelConfig, dtype: torch.dtype, device: torch.device, max_model_len: int,
                ckpt_dir: Path | None = None, seed: int = 0, load_format: str = "auto",
                expert_parallel: bool = False):
    if cfg.architecture in ("opt", "gpt2"):
        model = OPTModel(cfg, dtype, device, max_model_len)
    else:
        model = TransformerModel(cfg, dtype, device, max_model_len, expert_parallel=expert_parallel)
    has_weights = ckpt_dir is not None and any(Path(ckpt_dir).glob("*.safetensors"))
    if has_weights and load_format != "dummy":
        model.init_random(seed)  # allocates every tensor; checkpoint values overwrite them
        load_hf_weights(model, Path(ckpt_dir))
    else:
        model.init_random(seed)
    return model


def _iter_safetensors(ckpt_dir: Path):
    from safetensors import safe_open

    index = ckpt_dir / "model.safetensors.index.json"
    if index.exists():
        with open(index) as f:
            files = sorted(set(json.load(f)["weight_map"].values()))
    else:
        files = sorted(p.name for p in ckpt_dir.glob("*.safetensors"))
    for fn in files:
        with safe_open(str(ckpt_dir / fn), framework="pt", device="cpu") as f:
            for k in f.keys():
                yield k, f.get_tensor(k)


def load_hf_weights(model, ckpt_dir: Path) -> None:
    if isinstance(model, OPTModel):
        (_load_gpt2 if model.cfg.architecture == "gpt2" else _load_opt)(model, ckpt_dir)
    else:
        _load_llama(model, ckpt_dir)


def _shard_rows(t: torch.Tensor, rank: int, size: int) -> torch.Tensor:
    n = t.shape[0] // size
    return t[rank * n:(rank + 1) * n]


def _shard_cols(t: torch.Tensor, rank: int, size: int) -> torch.Tensor:
    n = t.shape[1] // size
    return t[:, rank * n:(rank + 1) * n]


def _load_llama(model: TransformerModel, ckpt_dir: Path) -> None:
    cfg = model.cfg
    tp = get_tp()
    r, s = tp.rank, tp.size
    d = cfg.head_dim
    nq, nkv, inter = model.n_q, model.n_kv, model.inter
    pat = re.compile(r"model\.layers\.(\d+)\.(.+)")
    for name, t in _iter_safetensors(ckpt_dir):
        t = t.to(model.dtype)
        if name == "model.embed_tokens.weight":
            if model.embed is None:
                continue
            model.embed.copy_(t)
            if model.lm_head is not model.embed and cfg.tie_word_embeddings:
                model.lm_head[: min(model.vocab_shard, t.shape[0] - r * model.vocab_shard)].copy_(
                    t[r * model.vocab_shard:(r + 1) * model.vocab_shard])
            continue
        if name == "model.norm.weight":
            if model.norm is not None:
                model.norm.copy_(t)
            continue
        if name == "lm_head.weight":
            if model.lm_head is not None and model.lm_head is not model.embed:
                rows = t[r * model.vocab_shard:(r + 1) * model.vocab_shard]
                model.lm_head[: rows.shape[0]].copy_(rows)
            continue
        m = pat.match(name)
        if not m:
            continue
        li, sub = int(m.group(1)), m.group(2)
        if not (model.first_layer <= li < model.last_layer):
            continue  # another pipeline stage's layer
        lw: LayerWeights = model.layers[li - model.first_layer]
        if sub == "input_layernorm.weight":
            lw.ln1.copy_(t)
        elif sub == "post_attention_layernorm.weight":
            lw.ln2.copy_(t)
        elif sub == "self_attn.q_proj.weight":
            lw.wqkv[: nq * d].copy_(_shard_rows(t, r, s))
        elif sub == "self_attn.k_proj.weight":
            lw.wqkv[nq * d:(nq + nkv) * d].copy_(_shard_rows(t, r, s))
        elif sub == "self_attn.v_proj.weight":
            lw.wqkv[(nq + nkv) * d:].copy_(_shard_rows(t, r, s))
        elif sub == "self_attn.q_proj.bias" and lw.bqkv is not None:
            lw.bqkv[: nq * d].copy_(_shard_rows(t, r, s))
        elif sub == "self_attn.k_proj.bias" and lw.bqkv is not None:
            lw.bqkv[nq * d:(nq + nkv) * d].copy_(_shard_rows(t, r, s))
        elif sub == "self_attn.v_proj.bias" and lw.bqkv is not None:
            lw.bqkv[(nq + nkv) * d:].copy_(_shard_rows(t, r, s))
        elif sub == "self_attn.o_proj.weight":
            lw.wo.copy_(_shard_cols(t, r, s))
        elif sub == "mlp.gate_proj.weight":
            lw.w_gate_up[:inter].copy_(_shard_rows(t, r, s))
        elif sub == "mlp.up_proj.weight":
            lw.w_gate_up[inter:].copy_(_shard_rows(t, r, s))
        elif sub == "mlp.down_proj.weight":
            lw.w_down.copy_(_shard_cols(t, r, s))
        elif sub == "block_sparse_moe.gate.weight":
            lw.router.copy_(t)
        else:
            me = re.match(r"block_sparse_moe\.experts\.(\d+)\.(w[123])\.weight", sub)
            if me:
                ex, which = int(me.group(1)), me.group(2)
                if model.ep:  # whole experts, this rank's range of them
                    ex -= model.first_expert
                    if 0 <= ex < model.local_experts:
                        full = cfg.intermediate_size
                        if which == "w1":
                            lw.w13[ex, :full].copy_(t)
                        elif which == "w3":
                            lw.w13[ex, full:].copy_(t)
                        else:
                            lw.w2[ex].copy_(t)
                elif which == "w1":
                    lw.w13[ex, :inter].copy_(_shard_rows(t, r, s))
                elif which == "w3":
                    lw.w13[ex, inter:].copy_(_shard_rows(t, r, s))
                else:
                    lw.w2[ex].copy_(_shard_cols(t, r, s))


def _load_opt(model: OPTModel, ckpt_dir: Path) -> None:
    h = model.cfg.hidden_size
    pat = re.compile(r"(?:model\.)?decoder\.layers\.(\d+)\.(.+)")
    for name, t in _iter_safetensors(ckpt_dir):
        t = t.to(model.dtype)
        base = name.removeprefix("model.")
        if base == "decoder.embed_tokens.weight":
            model.embed.copy_(t)
        elif base == "decoder.embed_positions.weight":
            n = min(t.shape[0], model.pos_embed.shape[0])
            model.pos_embed[:n].copy_(t[:n])
        elif base == "decoder.final_layer_norm.weight":
            model.final_ln[0].copy_(t)
        elif base == "decoder.final_layer_norm.bias":
            model.final_ln[1].copy_(t)
        else:
            m = pat.match(name)
            if not m:
                continue
            lw = model.layers[int(m.group(1))]
            sub = m.group(2)
            table = {
                "self_attn_layer_norm.weight": ("ln1_w", None), "self_attn_layer_norm.bias": ("ln1_b", None),
                "self_attn.q_proj.weight": ("wqkv", 0), "self_attn.k_proj.weight": ("wqkv", 1),
                "self_attn.v_proj.weight": ("wqkv", 2), "self_attn.q_proj.bias": ("bqkv", 0),
                "self_attn.k_proj.bias": ("bqkv", 1), "self_attn.v_proj.bias": ("bqkv", 2),
                "self_attn.out_proj.weight": ("wo", None), "self_attn.out_proj.bias": ("bo", None),
                "final_layer_norm.weight": ("ln2_w", None), "final_layer_norm.bias": ("ln2_b", None),
                "fc1.weight": ("fc1", None), "fc1.bias": ("b1", None),
                "fc2.weight": ("fc2", None), "fc2.bias": ("b2", None),
            }
            if sub in table:
                key, part = table[sub]
                if part is None:
                    lw[key].copy_(t)
                else:
                    lw[key][part * h:(part + 1) * h].copy_(t)


def _load_gpt2(model: OPTModel, ckpt_dir: Path) -> None:
    """GPT-2 checkpoints: ``wte / wpe / h.N.{ln_1, attn.c_attn, attn.c_proj, ln_2, mlp.c_fc,
    mlp.c_proj} / ln_f``; the Conv1D projections store [in, out] and are transposed here."""
    pat = re.compile(r"(?:transformer\.)?h\.(\d+)\.(.+)")
    table = {
        "ln_1.weight": ("ln1_w", False), "ln_1.bias": ("ln1_b", False),
        "attn.c_attn.weight": ("wqkv", True), "attn.c_attn.bias": ("bqkv", False),
        "attn.c_proj.weight": ("wo", True), "attn.c_proj.bias": ("bo", False),
        "ln_2.weight": ("ln2_w", False), "ln_2.bias": ("ln2_b", False),
        "mlp.c_fc.weight": ("fc1", True), "mlp.c_fc.bias": ("b1", False),
        "mlp.c_proj.weight": ("fc2", True), "mlp.c_proj.bias": ("b2", False),
    }
    for name, t in _iter_safetensors(ckpt_dir):
        t = t.to(model.dtype)
        base = name.removeprefix("transformer.")
        if base == "wte.weight":
            model.embed.copy_(t)
        elif base == "wpe.weight":
            n = min(t.shape[0], model.pos_embed.shape[0])
            model.pos_embed[:n].copy_(t[:n])
        elif base == "ln_f.weight":
            model.final_ln[0].copy_(t)
        elif base == "ln_f.bias":
            model.final_ln[1].copy_(t)
        else:
            m = pat.match(name)
            if m and m.group(2) in table:
                key, transpose = table[m.group(2)]
                model.layers[int(m.group(1))][key].copy_(t.t() if transpose else t)
