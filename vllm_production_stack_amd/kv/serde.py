"""KV page serialisation for the disk and remote tiers.

``naive``    header + raw page bytes (what every reference example configures:
             /root/reference/tutorials/assets/values-06-shared-storage.yaml:53).
``cachegen`` lossy: anchor-delta + layer-dependent quantisation on the GPU
             (``cachegen_encode`` kernel), then an entropy stage.  On a GPU that stage is the
             interleaved rANS coder of ``csrc/kernels/rans.hip`` (serde kind 2: the page is
             quantised, coded and packed on the device, the compressed bytes are what crosses
             PCIe; decoding mirrors it).  On a CPU-only host it is the adaptive range coder of
             ``libvpsa_kvio`` (kind 1, ~30 M symbols/s).  Both are 2-4x smaller than bf16; either
             side reads both kinds.

Wire format: 32-byte header ``VPKV | version u16 | serde u16 | n_layers u32 | n_kv u32 |
head_dim u32 | page_tokens u32 | payload_len u64`` followed by the payload.  For cachegen
the payload is ``scales fp32[n_groups] | rc_len u64 | range-coded symbols`` (kind 1) or
``spc u32 | streams u32 | coded_len u64 | scales fp32[n_groups] | freq u16[n_layers * 256] |
states u32[streams] | sizes u16[streams] | coded bytes`` (kind 2).
"""

from __future__ import annotations

import ctypes
import os
import struct

import numpy as np
import torch

from .. import _native, ops

MAGIC = b"VPKV"
SERDE_NAIVE, SERDE_CACHEGEN = 0, 1
# quantiser + GPU rANS entropy stage (what a GPU engine writes for "cachegen"); kind 1, the
# host range coder, remains for CPU hosts and for blobs already stored
SERDE_CACHEGEN_RANS = 2
_RANS_HDR = struct.Struct("<IIQ")  # streams per context, streams, coded bytes
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
    if kvio().kvio_rc_decode(buf.ctypes.data, buf.size, out.ctypes.data, n, ctx_period, n_ctx) != 0:
        raise ValueError("range-coded KV payload is truncated or corrupt")
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

    def _rans_ok(self, device) -> bool:
        return (torch.device(device).type == "cuda" and (2 * self.unit_elems) % ops.RANS_SPC == 0
                and os.environ.get("VPSA_CACHEGEN_HOST_CODER", "0") != "1")

    def quantize(self, flat: torch.Tensor):
        """GPU half of the cachegen encoder.  On a GPU: anchor/delta quantisation kernel and the
        rANS entropy stage, one device-to-host copy of the finished payload -> (payload, None).
        Otherwise the int8 symbols and fp32 scales come back for the host range coder
        -> (symbols, scales)."""
        lv = self._levels_dev.get(flat.device)
        if lv is None:
            lv = torch.tensor(self.levels, dtype=torch.int32, device=flat.device)
            self._levels_dev[flat.device] = lv
        sym, scales = ops.cachegen_encode(flat.contiguous(), lv, self.unit_elems, self.n_layers)
        if self._rans_ok(flat.device):
            return self.rans_payload(sym, scales), None
        return sym.cpu().numpy(), scales.float().cpu().numpy()

    def rans_payload(self, sym: torch.Tensor, scales: torch.Tensor) -> np.ndarray:
        """Entropy-code quantised symbols where they are and return the kind-2 payload bytes."""
        freq, states, sizes, coded = ops.rans_encode(sym.view(-1), self.n_layers)
        body = torch.cat([scales.float().contiguous().view(-1).view(torch.uint8),
                          freq.view(torch.uint8), states.view(torch.uint8), sizes.view(torch.uint8), coded])
        head = _RANS_HDR.pack(ops.RANS_SPC, states.numel(), coded.numel())
        return np.concatenate([np.frombuffer(head, dtype=np.uint8), body.cpu().numpy()])

    def pack(self, sym_np: np.ndarray, sc_np: np.ndarray | None) -> bytes:
        """Host half: frame a finished rANS payload, or range-code the symbols (releases the
        GIL: safe to run on worker threads)."""
        if sc_np is None:
            return _HDR.pack(MAGIC, 1, SERDE_CACHEGEN_RANS, self.n_layers, self.n_kv, self.head_dim,
                             ops.PAGE, sym_np.size) + sym_np.tobytes()
        coded = rc_encode(sym_np, 2 * self.unit_elems, self.n_layers)
        payload = sc_np.tobytes() + struct.pack("<Q", len(coded)) + coded
        return self._header(len(payload)) + payload

    def uses_codec(self, page: torch.Tensor) -> bool:
        return (self.kind == SERDE_CACHEGEN and self.unit_elems % 8192 == 0
                and page.dtype == torch.bfloat16)

    # ------------------------------------------------------------------ decode
    def _open(self, blob: bytes):
        """Header checks shared by every decoder: (kind, payload).  Anything that is not a whole
        page blob of this model's geometry is a ``ValueError``: blobs come from disk and from
        other hosts, and a damaged one must read as a cache miss, never as a wrong page."""
        if len(blob) < _HDR.size:
            raise ValueError("not a KV page blob")
        magic, ver, kind, nl, nkv, hd, pt, plen = _HDR.unpack_from(blob, 0)
        if magic != MAGIC or ver != 1:
            raise ValueError("not a KV page blob")
        if (nl, nkv, hd, pt) != (self.n_layers, self.n_kv, self.head_dim, ops.PAGE):
            raise ValueError("KV page geometry mismatch")
        if plen != len(blob) - _HDR.size:
            raise ValueError("truncated KV page blob")
        payload = memoryview(blob)[_HDR.size:]
        if kind == SERDE_NAIVE:
            if plen != self.page_elems * torch.empty((), dtype=self.dtype).element_size():
                raise ValueError("KV page blob of the wrong size")
        elif kind == SERDE_CACHEGEN_RANS:
            if plen < _RANS_HDR.size:
                raise ValueError("damaged cachegen payload")
        elif kind == SERDE_CACHEGEN:
            if plen < 4 * (self.page_elems // 64) + 8:
                raise ValueError("damaged cachegen payload")
        else:
            raise ValueError(f"unknown KV page encoding {kind}")
        return kind, payload

    def decode(self, blob: bytes, device: torch.device) -> torch.Tensor:
        kind, payload = self._open(blob)
        if kind == SERDE_NAIVE:
            t = torch.frombuffer(bytearray(payload), dtype=torch.uint8).view(self.dtype)
            return t.to(device)
        if kind == SERDE_CACHEGEN_RANS:
            return self.restore_rans(np.frombuffer(payload, dtype=np.uint8), device)
        sym_np, sc_np = self._unpack_payload(payload)
        return self.dequantize(sym_np, sc_np, device)

    def restore_rans(self, payload: np.ndarray, device) -> torch.Tensor:
        """rANS payload (host bytes) -> KV page on ``device``: one host-to-device copy, entropy
        decode and dequantisation on the GPU (NumPy definitions on a CPU-only host)."""
        spc, n_streams, coded_len = _RANS_HDR.unpack_from(payload, 0)
        n_groups = self.page_elems // 64
        a = _RANS_HDR.size
        cuts = np.cumsum([a, 4 * n_groups, 2 * 256 * self.n_layers, 4 * n_streams, 2 * n_streams, coded_len])
        if n_streams != self.n_layers * spc or cuts[-1] > payload.size:
            raise ValueError("damaged cachegen payload")
        body = torch.from_numpy(payload[a:cuts[-1]].copy()).to(device)
        o = cuts - a
        scales = body[o[0]:o[1]].view(torch.float32)
        freq = body[o[1]:o[2]].view(torch.int16)
        states = body[o[2]:o[3]].view(torch.int32)
        sizes = body[o[3]:o[4]].view(torch.int16)
        coded = body[o[4]:o[5]]
        sym = ops.rans_decode(freq, states, sizes, coded, self.page_elems, self.n_layers, spc)
        return ops.cachegen_decode(sym, scales, self.unit_elems)

    def restore(self, got: tuple, device) -> torch.Tensor:
        """Finish what ``unpack`` started (its "codec" and "rans" results)."""
        if got[0] == "rans":
            return self.restore_rans(got[1], device)
        return self.dequantize(got[1], got[2], device)

    def _unpack_payload(self, payload) -> tuple[np.ndarray, np.ndarray]:
        n_groups = self.page_elems // 64
        scales = np.frombuffer(payload[: 4 * n_groups], dtype=np.float32).copy()
        (clen,) = struct.unpack_from("<Q", payload, 4 * n_groups)
        if clen != len(payload) - 4 * n_groups - 8:
            raise ValueError("damaged cachegen payload")
        coded = bytes(payload[4 * n_groups + 8:])
        return rc_decode(coded, self.page_elems, 2 * self.unit_elems, self.n_layers), scales

    def unpack(self, blob: bytes):
        """Host half of decode for worker threads: ("raw", uint8 array) for a naive blob,
        ("codec", symbols, scales) for a cachegen one."""
        kind, payload = self._open(blob)
        if kind == SERDE_NAIVE:
            return ("raw", np.frombuffer(payload, dtype=np.uint8).copy())
        if kind == SERDE_CACHEGEN_RANS:
            return ("rans", np.frombuffer(payload, dtype=np.uint8))  # entropy decode happens on the GPU
        return ("codec",) + self._unpack_payload(payload)

    def dequantize(self, sym_np: np.ndarray, sc_np: np.ndarray, device: torch.device) -> torch.Tensor:
        sym = torch.from_numpy(sym_np)
        scales = torch.from_numpy(sc_np)
        return ops.cachegen_decode(sym.to(device), scales.to(device), self.unit_elems)
