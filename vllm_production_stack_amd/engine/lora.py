"""Multi-adapter LoRA: adapters live in fixed slots; a step applies, per target projection,
``y[rows of slot s] += (x[rows] @ A_s^T) @ B_s^T * scale_s`` for every active slot.

Adapters are read from a PEFT-style directory (adapter_config.json +
adapter_model.safetensors).  The fused projections of the engine (qkv, gate|up) take the
per-module PEFT tensors at their row offsets."""

from __future__ import annotations

import json
import re
from pathlib import Path

import torch


class LoRAAdapter:
    def __init__(self, name: str, path: str, scale: float):
        self.name = name
        self.path = path
        self.scale = scale
        # (target, layer) -> list of (A [r, in], B [out_rows, r], row_offset)
        self.parts: dict[tuple[str, int], list[tuple[torch.Tensor, torch.Tensor, int]]] = {}


class LoRAManager:
    def __init__(self, model, max_loras: int, max_rank: int = 64):
        self.model = model
        self.max_loras = max_loras
        self.max_rank = max_rank
        self.slots: list[LoRAAdapter | None] = [None] * max_loras
        self.by_name: dict[str, int] = {}

    def has(self, name: str) -> bool:
        return name in self.by_name

    def slot_of(self, name: str) -> int:
        return self.by_name.get(name, -1)

    def list_adapters(self) -> list[tuple[str, str]]:
        return [(a.name, a.path) for a in self.slots if a is not None]

    def unload(self, name: str) -> None:
        s = self.by_name.pop(name)
        self.slots[s] = None

    def load(self, name: str, path: str) -> None:
        if name in self.by_name:
            raise ValueError(f"adapter {name} already loaded")
        free = [i for i, a in enumerate(self.slots) if a is None]
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
        adapter = LoRAAdapter(name, str(path), float(cfg.get("lora_alpha", r)) / r)
        st_file = p / "adapter_model.safetensors"
        if st_file.exists():
            from safetensors import safe_open

            tensors = {}
            with safe_open(str(st_file), framework="pt", device="cpu") as f:
                for k in f.keys():
                    tensors[k] = f.get_tensor(k)
            self._index(adapter, tensors)
        self.slots[free[0]] = adapter
        self.by_name[name] = free[0]

    def _index(self, adapter: LoRAAdapter, tensors: dict[str, torch.Tensor]) -> None:
        m = self.model
        d = m.head_dim
        offs = {"q_proj": ("qkv", 0), "k_proj": ("qkv", m.n_q * d), "v_proj": ("qkv", (m.n_q + m.n_kv) * d),
                "o_proj": ("o", 0)}
        pat = re.compile(r".*layers\.(\d+)\.self_attn\.(q_proj|k_proj|v_proj|o_proj)\.lora_A\.weight")
        for k, a in tensors.items():
            mt = pat.match(k)
            if not mt:
                continue
            layer, mod = int(mt.group(1)), mt.group(2)
            b = tensors.get(k.replace("lora_A", "lora_B"))
            if b is None:
                continue
            target, off = offs[mod]
            adapter.parts.setdefault((target, layer), []).append(
                (a.to(m.dtype).to(m.device), b.to(m.dtype).to(m.device), off))

    def apply(self, target: str, layer: int, x: torch.Tensor, y: torch.Tensor,
              lora_slots: torch.Tensor) -> None:
        for s, adapter in enumerate(self.slots):
            if adapter is None:
                continue
            parts = adapter.parts.get((target, layer))
            if not parts:
                continue
            rows = (lora_slots == s).nonzero(as_tuple=True)[0]
            if rows.numel() == 0:
                continue
            xs = x.index_select(0, rows)
            for a, b, off in parts:
                delta = (xs @ a.t()) @ b.t() * adapter.scale
                y[rows, off:off + b.shape[0]] += delta.to(y.dtype)
