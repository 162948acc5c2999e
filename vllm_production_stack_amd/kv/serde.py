"""KV page serialisation for the disk and remote tiers.

``naive``    header + raw page bytes (what every reference example configures:
             /root/reference/tutorials/assets/values-06-shared-storage.yaml:53).
``cachegen`` lossy: anchor-delta + layer-dependent quantisation on the GPU
             (``cachegen_encode`` kernel), then an adaptive range coder on the host
             (``libvpsa_kvio``).  Typically 3-4x smaller than bf16.

Wire format: 32-byte header ``VPKV | version u16 | serde u16 | n_layers u32 | n_kv u32 |
head_dim u32 | page_tokens u32 | payload_len u64`` followed by the payload.  For cachegen
the payload is ``scales fp32[n_groups] | rc_len u64 | range-coded symbols``.
"""

from __future__ import annotations

import ctypes
import struct

import numpy as np
import torch

from .. import _native, ops

MAGIC = b"VPKV"
SERDE_NAIVE, SERDE_CACHEGEN = 0, 1
_HDR = struct.Struct("<4sHHIIIIQ")

_kvio = None


def kvio():
    global _kvio
    if _kvio is None:
        lib = _native.load("vpsa_kvio")
        lib.kvio_rc_encode.restype = ctypes.c_int64
        lib.kvio_rc_encode.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p,
                                       ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int]
        lib.kvio_rc_decode.restype = ctypes.c_int
        lib.kvio_rc_decode.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p,
                                       ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int]
        _kvio = lib
    return _kvio


def rc_encode(symbols: np.ndarray, ctx_period: int, n_ctx: int) -> bytes:
    symbols = np.ascontiguousarray(symbols, dtype=np.int8)
    cap = symbols.size + symbols.size // 8 + 64
    out = np.empty(cap, dtype=np.uint8)
    n = kvio().kvio_rc_encode(symbols.ctypes.data, symbols.size, out.ctypes.data, cap, ctx_period, n_ctx)
    if n < 0:
        raise RuntimeError("range coder output buffer too small")
    return out[:n].tobytes()


def rc_decode(data: bytes, n: int, ctx_period: int, n_ctx: int) -> np.ndarray:
    buf = np.frombuffer(data, dtype=np.uint8)
    out = np.empty(n, dtype=np.int8)
    kvio().kvio_rc_decode(buf.ctypes.data, buf.size, out.ctypes.data, n, ctx_period, n_ctx)
    return out


def default_levels(n_layers: int) -> list[int]:
    """Quantisation levels per layer: early layers are the most sensitive to KV error and
    keep 8 bits; the last two thirds step down to ~6 and ~5 bits."""
    a, b = n_layers // 3, 2 * n_layers // 3
    return [127 if i < a else (31 if i < b else 15) for i in range(n_layers)]


class PageSerde:
    def __init__(self, n_layers: int, n_kv: int, head_dim: int, dtype: torch.dtype, kind: str = "naive",
                 levels: list[int] | None = None):
        self.n_layers, self.n_kv, self.head_dim, self.dtype = n_layers, n_kv, head_dim, dtype
        self.kind = SERDE_CACHEGEN if kind == "cachegen" else SERDE_NAIVE
        self.page_elems = n_layers * 2 * n_kv * ops.PAGE * head_dim
        self.unit_elems = n_kv * ops.PAGE * head_dim
        self.levels = levels or default_levels(n_layers)
        self._levels_dev: dict = {}

    def _header(self, payload_len: int) -> bytes:
        return _HDR.pack(MAGIC, 1, self.kind, self.n_layers, self.n_kv, self.head_dim, ops.PAGE,
                         payload_len)

    # ------------------------------------------------------------------ encode
    def encode(self, page: torch.Tensor) -> bytes:
        """``page``: one KV page (any device), ``page_elems`` elements."""
        flat = page.reshape(-1)
        if self.kind == SERDE_NAIVE or self.unit_elems % 8192 != 0 or flat.dtype != torch.bfloat16:
            raw = flat.contiguous().cpu().view(torch.uint8).numpy().tobytes()
            return _HDR.pack(MAGIC, 1, SERDE_NAIVE, self.n_layers, self.n_kv, self.head_dim, ops.PAGE,
                             len(raw)) + raw
        sym_np, sc_np = self.quantize(flat)
        return self.pack(sym_np, sc_np)

    def quantize(self, flat: torch.Tensor) -> tuple[np.ndarray, np.ndarray]:
        """GPU half of the cachegen encoder: anchor/delta quantisation kernel, then the int8
        symbols and fp32 scales come to the host."""
        lv = self._levels_dev.get(flat.device)
        if lv is None:
            lv = torch.tensor(self.levels, dtype=torch.int32, device=flat.device)
            self._levels_dev[flat.device] = lv
        sym, scales = ops.cachegen_encode(flat.contiguous(), lv, self.unit_elems, self.n_layers)
        return sym.cpu().numpy(), scales.float().cpu().numpy()

    def pack(self, sym_np: np.ndarray, sc_np: np.ndarray) -> bytes:
        """Host half: range-code the symbols (releases the GIL: safe to run on worker threads)."""
        coded = rc_encode(sym_np, 2 * self.unit_elems, self.n_layers)
        payload = sc_np.tobytes() + struct.pack("<Q", len(coded)) + coded
        return self._header(len(payload)) + payload

    def uses_codec(self, page: torch.Tensor) -> bool:
        return (self.kind == SERDE_CACHEGEN and self.unit_elems % 8192 == 0
                and page.dtype == torch.bfloat16)

    # ------------------------------------------------------------------ decode
    def decode(self, blob: bytes, device: torch.device) -> torch.Tensor:
        magic, ver, kind, nl, nkv, hd, pt, plen = _HDR.unpack_from(blob, 0)
        if magic != MAGIC or ver != 1:
            raise ValueError("not a KV page blob")
        if (nl, nkv, hd, pt) != (self.n_layers, self.n_kv, self.head_dim, ops.PAGE):
            raise ValueError("KV page geometry mismatch")
        payload = memoryview(blob)[_HDR.size:_HDR.size + plen]
        if kind == SERDE_NAIVE:
            t = torch.frombuffer(bytearray(payload), dtype=torch.uint8).view(self.dtype)
            return t.to(device)
        sym_np, sc_np = self._unpack_payload(payload)
        return self.dequantize(sym_np, sc_np, device)

    def _unpack_payload(self, payload) -> tuple[np.ndarray, np.ndarray]:
        n_groups = self.page_elems // 64
        scales = np.frombuffer(payload[: 4 * n_groups], dtype=np.float32).copy()
        (clen,) = struct.unpack_from("<Q", payload, 4 * n_groups)
        coded = bytes(payload[4 * n_groups + 8: 4 * n_groups + 8 + clen])
        return rc_decode(coded, self.page_elems, 2 * self.unit_elems, self.n_layers), scales

    def unpack(self, blob: bytes):
        """Host half of decode for worker threads: ("raw", uint8 array) for a naive blob,
        ("codec", symbols, scales) for a cachegen one."""
        magic, ver, kind, nl, nkv, hd, pt, plen = _HDR.unpack_from(blob, 0)
        if magic != MAGIC or ver != 1:
            raise ValueError("not a KV page blob")
        if (nl, nkv, hd, pt) != (self.n_layers, self.n_kv, self.head_dim, ops.PAGE):
            raise ValueError("KV page geometry mismatch")
        payload = memoryview(blob)[_HDR.size:_HDR.size + plen]
        if kind == SERDE_NAIVE:
            return ("raw", np.frombuffer(payload, dtype=np.uint8).copy())
        return ("codec",) + self._unpack_payload(payload)

    def dequantize(self, sym_np: np.ndarray, sc_np: np.ndarray, device: torch.device) -> torch.Tensor:
        sym = torch.from_numpy(sym_np)
        scales = torch.from_numpy(sc_np)
        return ops.cachegen_decode(sym.to(device), scales.to(device), self.unit_elems)
