"""Multi-adapter LoRA: adapters live in fixed slots of stacked device tensors; one batched
shrink launch and one expand launch per projection apply, for every row of the step,
``y[t] += B_s (A_s x[t])`` with ``s`` the adapter slot of that row (rows without an adapter
are untouched).  The operator drives the slots through ``/v1/load_lora_adapter`` and
``/v1/unload_lora_adapter`` (/root/reference/operator/internal/controller/loraadapter_controller.go:582-610).

Adapters are read from a PEFT-style directory (adapter_config.json +
adapter_model.safetensors).  The fused projections of the engine take the per-module PEFT
tensors side by side: target "qkv" stacks q_proj | k_proj | v_proj, "gate_up" stacks
gate_proj | up_proj; "o" and "down" are single modules.  Per target and layer:

    A [slots, n_sub * r_pad, in]    rank rows zero-padded to r_pad (32 or 64: one or two
                                    MFMA reduction steps in the expand kernel)
    B[sub] [slots, out_sub, r_pad]  lora_alpha / r folded in
"""

from __future__ import annotations

import json
import re
from pathlib import Path

import torch

from .. import ops

# target -> list of (PEFT module name, which model dimension its output has)
_TARGETS = {
    "qkv": ("self_attn", ("q_proj", "k_proj", "v_proj")),
    "o": ("self_attn", ("o_proj",)),
    "gate_up": ("mlp", ("gate_proj", "up_proj")),
    "down": ("mlp", ("down_proj",)),
}
_KEY = re.compile(r".*layers\.(\d+)\.(self_attn|mlp)\.(\w+)\.lora_A(?:\.\w+)?\.weight$")


class LoRAManager:
    def __init__(self, model, max_loras: int, max_rank: int = 64):
        if max_rank > 64:
            raise ValueError("--max-lora-rank above 64 is not supported")
        self.model = model
        self.max_loras = max_loras
        self.max_rank = max_rank
        self.r_pad = 32 if max_rank <= 32 else 64
        self.names: list[str | None] = [None] * max_loras
        self.paths: list[str] = [""] * max_loras
        self.by_name: dict[str, int] = {}
        self.version = 1  # bumped whenever the adapter set changes (captured graphs follow it)
        cfg = model.cfg
        d = model.head_dim
        hidden = cfg.hidden_size
        self.dims: dict[str, tuple[int, list[int]]] = {
            "qkv": (hidden, [model.n_q * d, model.n_kv * d, model.n_kv * d]),
            "o": (model.n_q * d, [hidden]),
        }
        if cfg.num_experts == 0:  # dense MLP only: expert MLPs take no adapters
            self.dims["gate_up"] = (hidden, [cfg.intermediate_size, cfg.intermediate_size])
            self.dims["down"] = (cfg.intermediate_size, [hidden])
        # (target, layer) -> (A, [B_sub...]); allocated when the first adapter touches it
        self.stacks: dict[tuple[str, int], tuple[torch.Tensor, list[torch.Tensor]]] = {}
        # (target, layer) -> set of slots with non-zero weights there
        self.users: dict[tuple[str, int], set[int]] = {}

    # ------------------------------------------------------------------ registry
    def has(self, name: str) -> bool:
        return name in self.by_name

    def slot_of(self, name: str) -> int:
        return self.by_name.get(name, -1)

    def list_adapters(self) -> list[tuple[str, str]]:
        return [(n, self.paths[i]) for i, n in enumerate(self.names) if n is not None]

    def unload(self, name: str) -> None:
        s = self.by_name.pop(name)
        self.names[s] = None
        self.version += 1
        for key, slots in self.users.items():
            if s in slots:
                a, bs = self.stacks[key]
                a[s].zero_()
                for b in bs:
                    b[s].zero_()
                slots.discard(s)

    def load(self, name: str, path: str) -> None:
        if name in self.by_name:
            raise ValueError(f"adapter {name} already loaded")
        free = [i for i, n in enumerate(self.names) if n is None]
        if not free:
            raise RuntimeError(f"all {self.max_loras} LoRA slots are in use")
        p = Path(path)
        cfg_file = p / "adapter_config.json"
        if not cfg_file.exists():
            raise FileNotFoundError(f"no adapter_config.json under {path}")
        cfg = json.loads(cfg_file.read_text())
        r = int(cfg.get("r", 8))
        if r > self.max_rank:
            raise ValueError(f"adapter rank {r} exceeds --max-lora-rank {self.max_rank}")
        alpha = float(cfg.get("lora_alpha", r))
        scale = alpha / (r ** 0.5) if cfg.get("use_rslora") else alpha / r  # rank-stabilised variant
        slot = free[0]
        st_file = p / "adapter_model.safetensors"
        if st_file.exists():
            from safetensors import safe_open

            tensors = {}
            with safe_open(str(st_file), framework="pt", device="cpu") as f:
                for k in f.keys():
                    tensors[k] = f.get_tensor(k)
            self._place(slot, tensors, scale)
        self.names[slot] = name
        self.paths[slot] = str(path)
        self.by_name[name] = slot
        self.version += 1

    def _stack(self, target: str, layer: int):
        key = (target, layer)
        if key not in self.stacks:
            in_dim, outs = self.dims[target]
            m = self.model
            a = torch.zeros((self.max_loras, len(outs) * self.r_pad, in_dim), dtype=m.dtype, device=m.device)
            bs = [torch.zeros((self.max_loras, o, self.r_pad), dtype=m.dtype, device=m.device) for o in outs]
            self.stacks[key] = (a, bs)
            self.users[key] = set()
        return self.stacks[key]

    def _place(self, slot: int, tensors: dict[str, torch.Tensor], scale: float) -> None:
        first, last = self.model.first_layer, self.model.last_layer
        for k, a in tensors.items():
            mt = _KEY.match(k)
            if not mt:
                continue
            layer, block, mod = int(mt.group(1)), mt.group(2), mt.group(3)
            if not first <= layer < last:
                continue
            b = tensors.get(k.replace("lora_A", "lora_B"))
            if b is None:
                continue
            for target, (blk, subs) in _TARGETS.items():
                if blk == block and mod in subs and target in self.dims:
                    sub = subs.index(mod)
                    in_dim, outs = self.dims[target]
                    r = a.shape[0]
                    if a.shape[1] != in_dim or tuple(b.shape) != (outs[sub], r):
                        raise ValueError(f"{k}: shapes {tuple(a.shape)} / {tuple(b.shape)} do not fit "
                                         f"{target}[{mod}] ({in_dim} -> {outs[sub]})")
                    sa, sbs = self._stack(target, layer)
                    sa[slot, sub * self.r_pad: sub * self.r_pad + r] = a.to(sa.dtype)
                    sbs[sub][slot, :, :r] = (b.float() * scale).to(sa.dtype)
                    self.users[(target, layer)].add(slot)

    # ------------------------------------------------------------------ forward hook
    def apply(self, target: str, layer: int, x: torch.Tensor, y: torch.Tensor,
              lora_slots: torch.Tensor) -> None:
        """y += B_s (A_s x) row-wise, in place.  ``lora_slots`` [T] int32, -1 = no adapter."""
        key = (target, layer)
        if not self.users.get(key):
            return
        a, bs = self.stacks[key]
        h = ops.lora_shrink(x, a, lora_slots, a.shape[1])
        ops.lora_expand(h, bs, lora_slots, y, self.r_pad)
