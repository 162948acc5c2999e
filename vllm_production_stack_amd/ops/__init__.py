"""Engine ops.  CUDA(HIP)-device tensors run the hand-written gfx950 kernels from
``libvpsa_kernels.so`` (loaded with ctypes and launched on the current torch stream, so
they are capturable into HIP graphs); CPU tensors run the PyTorch definitions in
``reference.py``.  There is no silent GPU fallback: a GPU tensor with the library
missing raises.
"""

from __future__ import annotations

import os

import ctypes
from ctypes import c_float, c_int, c_long, c_void_p

import torch

from .. import _native
from . import reference
from .reference import PAGE, build_cos_sin  # noqa: F401

HEAD_DIM_HIP = 128
_lib = None


def _kernels():
    global _lib
    if _lib is None:
        lib = _native.load("vpsa_kernels")
        for name in (
            "vpsa_rmsnorm", "vpsa_rope_cache", "vpsa_kv_cache_write", "vpsa_paged_prefill_rope", "vpsa_prefill_fuses_rope", "vpsa_silu_mul", "vpsa_argmax_bf16",
            "vpsa_paged_prefill", "vpsa_set_prefill_variant", "vpsa_prefill_tile_rows", "vpsa_set_decode_variant", "vpsa_paged_decode", "vpsa_kv_gather_pages",
            "vpsa_kv_scatter_pages", "vpsa_cachegen_encode", "vpsa_cachegen_decode",
            "vpsa_moe_topk_softmax", "vpsa_moe_align", "vpsa_moe_permute", "vpsa_moe_combine",
            "vpsa_moe_gemv", "vpsa_moe_mfma", "vpsa_lora_shrink", "vpsa_lora_expand",
            "vpsa_skinny_gemm", "vpsa_hbm_read_probe", "vpsa_set_gemm_variant", "vpsa_stream_gemm",
            "vpsa_stream_gemm_splits", "vpsa_set_stream_targets", "vpsa_rans_encode", "vpsa_rans_pack",
            "vpsa_rans_decode", "vpsa_sample", "vpsa_grouped_gemm", "vpsa_packed_gemm",
            "vpsa_packed_gemm_splits", "vpsa_set_packed_target", "vpsa_dense_decode_attention", "vpsa_dense_attention",
            "vpsa_tile_gemm", "vpsa_tile_gemm_debug", "vpsa_paged_prefill_split",
        ):
            getattr(lib, name).restype = c_int
        lib.vpsa_prefill_split_record_bytes.restype = c_long
        if os.environ.get("VPSA_PREFILL_VARIANT"):  # pin one prefill kernel for A/B runs
            lib.vpsa_set_prefill_variant(c_int(int(os.environ["VPSA_PREFILL_VARIANT"])))
        _lib = lib
    return _lib


def native_loaded() -> bool:
    return _lib is not None


def set_prefill_variant(v: int) -> int:
    """Select the prefill attention kernel (for in-process A/B timing and tests): -1 automatic,
    0 first generation, 1 second generation without wave priorities, 2 second generation,
    3 second generation with head-major tile order, 4 / 5 third generation (64-row tiles, one wave
    per SIMD) in plain / head-major tile order, 6 / 7 the same without the hand-interleaved
    instruction stream.  Returns the previous selection."""
    return int(_kernels().vpsa_set_prefill_variant(c_int(v)))


def set_decode_variant(v: int) -> int:
    """Select the decode attention kernel (-1: automatic by split count, 0: one register set per
    wave, 1: software-pipelined with two register sets, 2: a wave takes both halves of a page;
    3-5: the same three with non-temporal K/V loads, which the automatic choice takes).
    Returns the previous selection."""
    return int(_kernels().vpsa_set_decode_variant(c_int(v)))


_DEVICE: int | None = None


def bind_device(index: int | None) -> None:
    """One process drives one GPU.  The HIP current device is a per-thread setting that starts
    at 0 in every new thread, and both ``torch.cuda.current_stream()`` and the kernel launches
    of the native library follow it: record the process's device here and every launch from
    any thread (engine loop, executor threads) first moves its thread onto that device."""
    global _DEVICE
    _DEVICE = index


def ensure_device() -> None:
    if _DEVICE is not None and torch.cuda.current_device() != _DEVICE:
        torch.cuda.set_device(_DEVICE)


def _stream():
    ensure_device()
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return c_void_p(t.data_ptr()) if t is not None else c_void_p(0)


def _check(rc, name):
    if rc != 0:
        raise RuntimeError(f"{name}: unsupported arguments (rc={rc})")


def _bf16_cuda(*ts):
    for t in ts:
        if t is not None and (not t.is_cuda or t.dtype != torch.bfloat16):
            raise TypeError("HIP kernels take bf16 device tensors")


# --------------------------------------------------------------------------- norms
def rms_norm(x, weight, eps, residual=None, out=None):
    """y = rmsnorm(x [+ residual]) * weight.  ``residual`` is updated in place with the sum."""
    if not x.is_cuda:
        y = reference.rms_norm(x, weight, eps, residual)
        if out is not None:
            out.copy_(y)
            return out
        return y
    _bf16_cuda(x, weight, residual)
    hidden = x.shape[-1]
    x2 = x.view(-1, hidden)
    if x2.stride(-1) != 1:
        raise ValueError("rms_norm: last dim must be contiguous")
    if out is None:
        out = torch.empty((x2.shape[0], hidden), dtype=x.dtype, device=x.device)
    if residual is not None and not residual.is_contiguous():
        raise ValueError("rms_norm: residual must be contiguous")
    rc = _kernels().vpsa_rmsnorm(_p(x2), _p(residual), _p(weight), _p(out), c_int(x2.shape[0]),
                                 c_int(hidden), c_long(x2.stride(0)), c_long(out.stride(0)),
                                 c_float(eps), _stream())
    _check(rc, "rms_norm")
    return out.view(x.shape)


# --------------------------------------------------------------------------- rope + cache write
def _generic_attention(t, head_dim) -> bool:
    """The HIP attention kernels are built for head_dim 128 (Llama-3, Mixtral); heads of 64
    channels reach them padded (``kv_head_dim``).  Any other head size on a GPU takes the PyTorch
    definitions: correct, slow, announced once, never chosen for a model the kernels cover."""
    global _warned_generic
    if not t.is_cuda:
        return True
    if head_dim == HEAD_DIM_HIP:
        return False
    if not _warned_generic:
        _warned_generic = True
        import logging

        logging.getLogger("vpsa.ops").warning(
            "head_dim %d: attention runs on the generic PyTorch path (the HIP kernels cover head_dim %d)",
            head_dim, HEAD_DIM_HIP)
    return True


_warned_generic = False

# Heads of 64 channels (OPT, GPT-2, Llama-3.2-1B, Qwen2-0.5B) run on the head_dim-128 kernels with
# their channels zero-padded in the KV cache and in Q: zero channels add nothing to QK^T, and the
# padded half of the output is dropped.  Costs twice the KV bytes of those (small) models and buys
# the MFMA kernels and HIP-graph decode instead of per-sequence PyTorch attention.
PAD_HEAD_64 = os.environ.get("VPSA_PAD_HEAD64", "1") != "0"


def kv_head_dim(head_dim: int, device) -> int:
    """Channels per head as stored in the KV pool of a replica on ``device``."""
    if PAD_HEAD_64 and head_dim == 64 and torch.device(device).type == "cuda":
        return HEAD_DIM_HIP
    return head_dim


_pad_cos_sin: dict = {}


class LayerGroupedPool:
    """A paged KV pool cut by layer: ``groups[g]`` is a pool of its own,
    ``[pages][layers / G][K|V][kv_head][64 x d]``, for layers ``g * per`` to ``(g + 1) * per - 1``
    of every page; page ids are shared.  A decode step of one layer then walks an address range
    G times shorter than with all layers in one page: on an MI355X the K/V read rate of the decode
    kernel falls from 6.7 to 6.2-6.4 TB/s once the pages a step touches lie more than about 60 GB
    apart (address translation), which a 250 GB pool of 8 MiB pages always is.  The price: a page
    is G pieces; the lower tiers copy it piece by piece (``page_parts``), replicas that push pages
    to or take pages from other replicas keep whole pages."""

    def __init__(self, groups):
        self.groups = list(groups)
        self.per = self.groups[0].shape[1]
        assert all(g.shape == self.groups[0].shape and g.is_contiguous() for g in self.groups)

    def for_layer(self, layer: int):
        return self.groups[layer // self.per], layer % self.per

    @property
    def shape(self):  # the pool as a whole: [pages][layers][K|V][kv_head][64 x d]
        s = self.groups[0].shape
        return torch.Size((s[0], self.per * len(self.groups), *s[2:]))

    dtype = property(lambda self: self.groups[0].dtype)
    device = property(lambda self: self.groups[0].device)
    is_cuda = property(lambda self: self.groups[0].is_cuda)

    def numel(self) -> int:
        return sum(g.numel() for g in self.groups)

    def element_size(self) -> int:
        return self.groups[0].element_size()

    def page(self, p: int) -> torch.Tensor:
        """A copy of page ``p`` as one tensor [layers][K|V][kv_head][64 x d]."""
        return torch.cat([g[p] for g in self.groups], dim=0)

    def write_page(self, p: int, page: torch.Tensor) -> None:
        for i, g in enumerate(self.groups):
            g[p].copy_(page[i * self.per:(i + 1) * self.per])


# A page as the lower KV tiers see it is the flat [layers][K|V][kv_head][64 x d] array whatever
# the pool's layout: the pieces of a grouped pool, in group order, are exactly that array.
def page_parts(pool, p: int) -> list:
    """The contiguous device pieces of page ``p``, in layer order."""
    if isinstance(pool, LayerGroupedPool):
        return [g[p] for g in pool.groups]
    return [pool[p]]


def page_numel(pool) -> int:
    n = 1
    for d in pool.shape[1:]:
        n *= int(d)
    return n


def page_tensor(pool, p: int) -> torch.Tensor:
    """Page ``p`` as [layers][K|V][kv_head][64 x d]: a view of a whole-page pool, a copy of a grouped one."""
    return pool.page(p) if isinstance(pool, LayerGroupedPool) else pool[p]


def page_read_into(pool, p: int, flat: torch.Tensor, non_blocking: bool = False) -> None:
    """``flat`` (page_numel elements, any device) <- page ``p``."""
    off = 0
    for part in page_parts(pool, p):
        n = part.numel()
        flat[off:off + n].copy_(part.reshape(-1), non_blocking=non_blocking)
        off += n


def page_assign(pool, p: int, flat: torch.Tensor, non_blocking: bool = False) -> None:
    """Page ``p`` <- ``flat`` (page_numel elements, any device)."""
    off = 0
    for part in page_parts(pool, p):
        n = part.numel()
        part.view(-1).copy_(flat[off:off + n], non_blocking=non_blocking)
        off += n


def attention_block(qkv, pool, layer, meta, n_q, n_kv, head_dim, scale, cos_sin, rotary=True):
    """Rotary embedding, KV-cache write and paged attention of one layer for a mixed batch
    (``meta.num_decode`` decode rows first, then the prefill rows).  qkv [T, (n_q + 2 n_kv) * d];
    returns the attention output [T, n_q * d]."""
    if isinstance(pool, LayerGroupedPool):
        pool, layer = pool.for_layer(layer)
    t = qkv.shape[0]
    d = head_dim
    dk = kv_head_dim(d, qkv.device)
    if dk != d:
        # rotate at the true width, then widen every head with zero channels
        if rotary:
            reference.rope_and_cache(qkv, meta.positions, None, cos_sin, pool, layer, n_q, n_kv, d, True)
        wide = torch.nn.functional.pad(qkv.view(t, n_q + 2 * n_kv, d), (0, dk - d)).view(t, -1)
        cs = _pad_cos_sin.get(qkv.device)
        if cs is None:
            cs = _pad_cos_sin[qkv.device] = torch.zeros((1, dk), dtype=torch.float32, device=qkv.device)
        out = _attention_block(wide, pool, layer, meta, n_q, n_kv, dk, scale, cs, False)
        return out.view(t, n_q, dk)[:, :, :d].reshape(t, n_q * d)
    return _attention_block(qkv, pool, layer, meta, n_q, n_kv, d, scale, cos_sin, rotary)


def prefill_fuses_rope(n_q: int, n_kv: int, tile_rows) -> bool:
    """True when the prefill kernel that will run rotates its query rows itself."""
    return bool(tile_rows) and int(_kernels().vpsa_prefill_fuses_rope(c_int(n_q), c_int(n_kv), c_int(tile_rows))) == 1


def _attention_block(qkv, pool, layer, meta, n_q, n_kv, d, scale, cos_sin, rotary):
    nd = meta.num_decode
    # Prefill rows: their K / V go to the cache through the tile kernel (rotated K, V transposed
    # in LDS so that the page's [channel][32 keys] runs are written whole), and the query rows are
    # rotated by the attention kernel as it loads them when the kernel that will run can do that
    # (GQA groups of 4 and multiples), else in place by a pass over the query heads only.  Decode
    # rows keep the per-token pass with the in-place rotation: their kernel reads q as stored.
    tiled = meta.num_prefill_seqs > 0 and not _generic_attention(qkv, d)
    fuse = tiled and rotary and prefill_fuses_rope(n_q, n_kv, meta.pre_tile_rows)
    if tiled:
        if nd > 0:
            rope_and_cache(qkv[:nd], meta.positions[:nd], meta.slots[:nd], cos_sin, pool, layer, n_q, n_kv, d,
                           rotary)
        kv_cache_write(qkv[nd:], meta.positions[nd:], meta.slots[nd:], cos_sin, pool, layer, n_q, n_kv, d,
                       rotary)
        if rotary and not fuse:
            rope_and_cache(qkv[nd:], meta.positions[nd:], None, cos_sin, pool, layer, n_q, 0, d, True)
    else:
        rope_and_cache(qkv, meta.positions, meta.slots, cos_sin, pool, layer, n_q, n_kv, d, rotary)
    t = qkv.shape[0]
    alloc = torch.zeros if getattr(meta, "pre_padded", False) else torch.empty
    out = alloc((t, n_q * d), dtype=qkv.dtype, device=qkv.device)
    q = qkv[:, : n_q * d]
    if nd > 0:
        paged_decode(q[:nd], out[:nd], pool, layer, meta.dec_block_tables, meta.dec_ctx_lens, n_q, n_kv, d,
                     scale, getattr(meta, "dec_num_splits", 1) or 1, getattr(meta, "dec_ws_o", None),
                     getattr(meta, "dec_ws_ml", None))
    if meta.num_prefill_seqs > 0:
        paged_prefill(q[nd:], out[nd:], pool, layer, meta.pre_block_tables, meta.pre_cu_q, meta.pre_ctx_lens,
                      meta.pre_tile_seq, meta.pre_tile_q0, n_q, n_kv, d, scale, meta.pre_tile_rows,
                      rope=cos_sin if fuse else None, split=getattr(meta, "pre_split", None),
                      split_ws=getattr(meta, "pre_split_ws", None), groups=getattr(meta, "pre_split_groups", None))
    return out


def kv_cache_write(qkv, positions, slots, cos_sin, pool, layer, n_q, n_kv, head_dim, rotary=True):
    """K (rotated) and V heads of ``qkv`` rows -> paged cache; ``qkv`` itself is not modified and
    the query heads are not read (prefill rows: their queries are rotated by ``paged_prefill(rope=)``
    or by a query-only ``rope_and_cache`` pass).  ``rotary=False`` writes K as it is."""
    _bf16_cuda(qkv, pool)
    assert positions.dtype == torch.int32 and slots.dtype == torch.int64
    assert cos_sin.dtype == torch.float32 and cos_sin.shape[1] == head_dim == 128
    base = pool.data_ptr() + layer * pool.stride(1) * 2
    rc = _kernels().vpsa_kv_cache_write(_p(qkv), _p(positions), _p(slots), _p(cos_sin), c_void_p(base),
                                        c_int(qkv.shape[0]), c_long(qkv.stride(0)), c_int(n_q),
                                        c_int(n_kv), c_long(pool.stride(0)), c_long(pool.stride(2)),
                                        c_int(1 if rotary else 0), _stream())
    _check(rc, "kv_cache_write")


def rope_and_cache(qkv, positions, slots, cos_sin, pool, layer, n_q, n_kv, head_dim, rotary=True):
    if _generic_attention(qkv, head_dim):
        return reference.rope_and_cache(qkv, positions, slots, cos_sin, pool, layer, n_q, n_kv,
                                        head_dim, rotary)
    _bf16_cuda(qkv, pool)
    assert positions.dtype == torch.int32 and (slots is None or slots.dtype == torch.int64)
    assert cos_sin.dtype == torch.float32 and cos_sin.shape[1] == head_dim
    base = pool.data_ptr() + layer * pool.stride(1) * 2
    rc = _kernels().vpsa_rope_cache(_p(qkv), _p(positions), _p(slots), _p(cos_sin), c_void_p(base),
                                    c_int(qkv.shape[0]), c_long(qkv.stride(0)), c_int(n_q),
                                    c_int(n_kv), c_long(pool.stride(0)), c_long(pool.stride(2)),
                                    c_int(1 if rotary else 0), _stream())
    _check(rc, "rope_and_cache")


# --------------------------------------------------------------------------- attention
def prefill_tile_rows(q_lens, n_q, n_kv) -> int:
    """Query rows per work-list entry.  The third-generation kernel works on 64-row tiles (one
    GQA group of 4 heads x 64 rows per workgroup) and is taken whenever the GQA ratio is a
    multiple of 4.  The choice deliberately does not look at the batch: two kernels round
    differently, and a request's tokens must not depend on what shares its step."""
    del q_lens
    rows = int(_kernels().vpsa_prefill_tile_rows(c_int(n_q), c_int(n_kv), c_int(1)))
    if rows <= 0:
        raise ValueError(f"paged_prefill: unsupported head layout n_q={n_q} n_kv={n_kv}")
    return rows


def build_prefill_tiles(q_lens, n_q, n_kv, rows=None, ctx_lens=None):
    """Host-side work list for the prefill kernel: one (sequence, first row) entry per query
    tile of ``rows`` rows (default: ``prefill_tile_rows``).  Sequences with the longest context
    come first (the heaviest workgroups start earliest) and a sequence's tiles stay together,
    longest first: the workgroups that run side by side on one XCD then stream the SAME K/V
    pages through its L2 instead of one stream per sequence (8 equal sequences interleaved cost
    the kernel a fifth of its throughput in HBM waits)."""
    qt = rows or prefill_tile_rows(q_lens, n_q, n_kv)
    weight = ctx_lens if ctx_lens is not None else q_lens
    seqs, q0s = [], []
    for s in sorted(range(len(q_lens)), key=lambda i: -weight[i]):
        for q0 in range((q_lens[s] - 1) // qt * qt, -1, -qt):
            seqs.append(s)
            q0s.append(q0)
    return seqs, q0s


# Work lists that leave the chip part-filled are cut along the key range: one new turn behind a
# 21k-token history is 8 workgroups that each walk 330 key tiles alone (10 ms on a 256-CU chip), and
# 288 equal workgroups run as one full round and a nearly empty second.  The pieces run as
# workgroups of their own and leave partials for a merge pass.  One workgroup occupies a CU (96 KiB
# of LDS), so plans are priced in rounds of TILE_CUS workgroups.
SPLIT_MIN_KEY_TILES = 8     # a piece is at least this long (the kernel needs >= 3 in the last one)
SPLIT_MAX_ROUNDS = 3        # longer lists lose at most a quarter of a round and are left alone
SPLIT_CANDIDATES = 8        # round counts tried, starting at the list's own
SPLIT_WG_OVERHEAD = 10.0    # a workgroup's prologue + epilogue, in key-tile times (measured: 7 rounds
                            # of 55-tile pieces take 62 tile times each on a full chip)
SPLIT_MERGE_COST = 25.0     # the merge launch, in key-tile times
SPLIT_RECORD_COST = 0.15    # writing + reading one partial record, in key-tile times
PREFILL_SPLITS = os.environ.get("VPSA_PREFILL_SPLITS", "1") != "0"


def prefill_split_record_bytes(n_q: int, n_kv: int) -> int:
    return int(_kernels().vpsa_prefill_split_record_bytes(c_int(n_q), c_int(n_kv)))


def prefill_splits_supported(n_q: int, n_kv: int, rows: int) -> bool:
    """Only the 64-row kernel (GQA ratios of 3 and up: workgroups of 4 heads, the last one of a KV
    head padded with spare waves) takes split work lists."""
    return PREFILL_SPLITS and rows == 64 and n_kv > 0 and n_q % n_kv == 0 and n_q // n_kv >= 3


def _whole_cost(kts, head_groups):
    """Modelled time of the uncut list in key-tile times.  Workgroups start longest first, so the
    chip stays full until the last, partial round, which is made of the shortest entries."""
    total = sum(kts) + SPLIT_WG_OVERHEAD * len(kts)
    wgs = len(kts) * head_groups
    tail = (TILE_CUS - wgs % TILE_CUS) % TILE_CUS / TILE_CUS * (min(kts) + SPLIT_WG_OVERHEAD)
    return max(max(kts) + SPLIT_WG_OVERHEAD, total * head_groups / TILE_CUS + tail)


def _split_counts(lengths, budget):
    """``lengths``: {key tiles: entries of that length}.  Pieces per entry of each length so that
    the longest piece is as short as ``budget`` entries in all allow; None when the budget does not
    cover the list."""
    def entries(piece):
        return sum(m * max(1, min(-(-kt // piece), kt // SPLIT_MIN_KEY_TILES)) for kt, m in lengths.items())
    lo, hi = SPLIT_MIN_KEY_TILES, max(lengths)
    if entries(hi) > budget:
        return None
    while lo < hi:
        mid = (lo + hi) // 2
        if entries(mid) <= budget:
            hi = mid
        else:
            lo = mid + 1
    return {kt: max(1, min(-(-kt // lo), kt // SPLIT_MIN_KEY_TILES)) for kt in lengths}


def plan_prefill_splits(seqs, q0s, q_lens, ctx_lens, n_q, n_kv, rows, max_entries=None, max_records=None):
    """Cut the key ranges of a work list that would leave the chip part-filled.  Returns None when
    nothing is to be split, else (seqs, q0s, split, groups, n_records): the expanded work list,
    ``split`` = 3 ints per entry (first key tile, end key tile, partial record or -1), ``groups`` =
    4 ints per split tile (sequence, first row, first record, records).

    The uncut list is priced with the chip kept full (what longest-first dispatch achieves), a plan
    as whole rounds of its longest piece: a cut has to win against the better case of leaving the
    list alone."""
    if not prefill_splits_supported(n_q, n_kv, rows) or not seqs:
        return None
    head_groups = n_kv * ((n_q // n_kv + 3) // 4)
    if len(seqs) * head_groups > SPLIT_MAX_ROUNDS * TILE_CUS:
        return None
    kts = []
    for s, q0 in zip(seqs, q0s):
        last_row = min(q0 + rows - 1, q_lens[s] - 1)
        kts.append((ctx_lens[s] - q_lens[s] + last_row) // PAGE + 1)
    if max(kts) < 2 * SPLIT_MIN_KEY_TILES:
        return None
    whole = _whole_cost(kts, head_groups)
    ideal = (sum(kts) + SPLIT_WG_OVERHEAD * len(kts)) * head_groups / TILE_CUS
    if whole < 1.1 * ideal + SPLIT_MERGE_COST:  # balanced as it is
        return None
    lengths = {}
    for kt in kts:
        lengths[kt] = lengths.get(kt, 0) + 1
    per_round = max(1, TILE_CUS // head_groups)  # work-list entries that fill the chip once
    first = -(-len(seqs) // per_round)
    best, best_cost = None, 0.95 * whole  # a cut has to pay for itself
    for rounds in range(first, first + (SPLIT_CANDIDATES if len(lengths) <= 16 else 3)):
        budget = rounds * per_round
        if max_entries is not None:
            budget = min(budget, max_entries)
        if max_records is not None:  # every entry may end up with a record of its own
            budget = min(budget, max_records)
        counts = _split_counts(lengths, budget)
        if counts is None or max(counts.values()) == 1:
            continue
        n = sum(m * counts[kt] for kt, m in lengths.items())
        records = sum(m * counts[kt] for kt, m in lengths.items() if counts[kt] > 1)
        longest = max(-(-kt // c) for kt, c in counts.items())
        cost = (-(-n * head_groups // TILE_CUS) * (longest + SPLIT_WG_OVERHEAD)
                + SPLIT_MERGE_COST + SPLIT_RECORD_COST * records * head_groups / 8)
        if cost < best_cost:
            best, best_cost = counts, cost
    if best is None:
        return None
    best = [best[kt] for kt in kts]
    seqs2, q0s2, split, groups = [], [], [], []
    rec = 0
    for s, q0, kt, c in zip(seqs, q0s, kts, best):
        if c == 1:
            seqs2.append(s)
            q0s2.append(q0)
            split += [0, kt, -1]
            continue
        groups += [s, q0, rec, c]
        base, extra = divmod(kt, c)
        k0 = 0
        for j in range(c):
            k1 = k0 + base + (1 if j < extra else 0)
            seqs2.append(s)
            q0s2.append(q0)
            split += [k0, k1, rec]
            rec += 1
            k0 = k1
    return seqs2, q0s2, split, groups, rec


def paged_prefill(q, out, pool, layer, block_tables, cu_q, ctx_lens, tile_seq, tile_q0, n_q, n_kv,
                  head_dim, scale, tile_rows=None, rope=None, split=None, split_ws=None, groups=None):
    """``tile_rows``: the tile height ``tile_seq`` / ``tile_q0`` were built for (default: the
    32-row-wave layout of the first two kernel generations).  ``rope``: the [position, head_dim]
    fp32 cos / sin table when ``q`` holds unrotated rows (``prefill_fuses_rope`` says whether the
    kernel that will run takes it)."""
    if _generic_attention(q, head_dim):
        return reference.paged_prefill(q, out, pool, layer, block_tables, cu_q, ctx_lens, n_q,
                                       n_kv, head_dim, scale)
    _bf16_cuda(q, out, pool)
    for t in (block_tables, cu_q, ctx_lens, tile_seq, tile_q0):
        assert t.dtype == torch.int32 and t.is_cuda and t.is_contiguous()
    assert tile_seq.numel() == tile_q0.numel()
    assert cu_q.numel() == ctx_lens.numel() + 1 and block_tables.shape[0] >= ctx_lens.numel()
    if tile_rows is None:
        tile_rows = 32 * (4 // min(n_q // n_kv, 4))
    base = pool.data_ptr() + layer * pool.stride(1) * 2
    assert rope is None or (rope.dtype == torch.float32 and rope.is_contiguous() and rope.shape[1] == head_dim)
    if split is not None:
        # split [entries][3], groups [n][4] (int32 on the device), split_ws fp32 with room for every record
        assert split.dtype == torch.int32 and split.numel() == 3 * tile_seq.numel() and split.is_contiguous()
        assert groups.dtype == torch.int32 and groups.numel() % 4 == 0 and groups.is_contiguous()
        assert split_ws.dtype == torch.float32 and split_ws.is_contiguous()
    rc = _kernels().vpsa_paged_prefill_split(
        _p(q), c_long(q.stride(0)), _p(out), c_long(out.stride(0)), c_void_p(base),
        c_long(pool.stride(0)), c_long(pool.stride(2)), _p(block_tables),
        c_int(block_tables.stride(0)), _p(cu_q), _p(ctx_lens), _p(tile_seq), _p(tile_q0),
        c_int(tile_seq.numel()), c_int(ctx_lens.numel()), c_int(tile_rows), c_int(n_q), c_int(n_kv),
        c_float(scale), _p(rope), _p(split), _p(split_ws), _p(groups),
        c_int(0 if groups is None else groups.numel() // 4), _stream())
    _check(rc, "paged_prefill")


def dense_decode_attention_supported(q, k, v, heads: int) -> bool:
    """Shapes the dense single-query kernel covers: bf16 on the GPU, 64-channel heads, unit
    stride over channels, 16-byte aligned rows."""
    d = q.shape[-1]
    if not all(t.is_cuda and t.dtype == torch.bfloat16 for t in (q, k, v)):
        return False
    return (d == heads * 64 and q.dim() == 2 and k.dim() == 3
            and k.shape == v.shape and k.stride() == v.stride() and k.shape[0] == q.shape[0] and k.shape[2] == d
            and q.stride(1) == 1 and k.stride(2) == 1 and q.stride(0) % 8 == 0 and k.stride(0) % 8 == 0
            and k.stride(1) % 8 == 0 and (q.data_ptr() | k.data_ptr() | v.data_ptr()) % 16 == 0)


def dense_decode_attention(q, k, v, heads: int, pos=None, out=None):
    """softmax(q k^T / 8) v for ONE query per batch row over dense K/V: q [B, heads * 64],
    k / v [B, T, heads * 64] -> [B, heads * 64].  ``pos`` (one int64 on the device) limits the
    keys to positions 0..pos; None attends to all T.  Graph-capturable."""
    b, d = q.shape
    if out is None:
        out = torch.empty((b, d), dtype=q.dtype, device=q.device)
    rc = _kernels().vpsa_dense_decode_attention(
        _p(q), c_long(q.stride(0)), _p(k), _p(v), c_long(k.stride(0)), c_long(k.stride(1)), _p(pos),
        c_int(b), c_int(heads), c_int(k.shape[1]), _p(out), c_long(out.stride(0)), c_float(0.125), _stream())
    _check(rc, "dense_decode_attention")
    return out


def dense_attention_supported(q, k, v, heads: int) -> bool:
    """Shapes the dense bidirectional kernel covers: bf16 on the GPU, [B, T, heads * 64] with unit
    stride over channels and 16-byte aligned rows."""
    if not all(t.is_cuda and t.dtype == torch.bfloat16 and t.dim() == 3 for t in (q, k, v)):
        return False
    d = q.shape[-1]
    return (d == heads * 64 and k.shape == v.shape and k.stride() == v.stride() and k.shape[0] == q.shape[0]
            and k.shape[2] == d and k.shape[1] > 0 and q.stride(2) == 1 and k.stride(2) == 1
            and all(x % 8 == 0 for x in (q.stride(0), q.stride(1), k.stride(0), k.stride(1)))
            and (q.data_ptr() | k.data_ptr() | v.data_ptr()) % 16 == 0)


def dense_attention(q, k, v, heads: int, out=None):
    """softmax(q k^T / 8) v with every query attending to every key: q [B, Tq, heads * 64],
    k / v [B, Tk, heads * 64] -> [B, Tq, heads * 64] (MFMA flash kernel, scores never stored)."""
    b, tq, d = q.shape
    if out is None:
        out = torch.empty((b, tq, d), dtype=q.dtype, device=q.device)
    rc = _kernels().vpsa_dense_attention(
        _p(q), c_long(q.stride(0)), c_long(q.stride(1)), _p(k), _p(v), c_long(k.stride(0)), c_long(k.stride(1)),
        _p(out), c_long(out.stride(0)), c_long(out.stride(1)), c_int(b), c_int(heads), c_int(tq),
        c_int(k.shape[1]), c_float(0.125), _stream())
    _check(rc, "dense_attention")
    return out


def decode_num_splits(num_seqs: int, n_kv: int, max_ctx: int) -> int:
    """Splits of the key range per (sequence, KV head).  Measured on 21.5k-token contexts in the
    engine's page layout with streaming loads (`benchmarks/attn_micro.py --cases decode --layers 32
    --decode-seqs 4,8,16,24,40,64`): batches of a few sequences read fastest with about one
    workgroup per CU (4 sequences: 5.25 TB/s at 192 workgroups, 4.7 at 1024), larger ones with about
    three (16 sequences: 6.5-6.6 TB/s at 512-768 workgroups, 40 sequences: 6.45 at 960, 6.35 at
    640 or 1280); at least 256 keys per split."""
    per_split = max(1, num_seqs * n_kv)
    target = 256 if per_split < 96 else 768
    want = max(1, -(-target // per_split))
    by_len = max(1, max_ctx // 256)
    return int(max(1, min(want, by_len, 32)))


def paged_decode(q, out, pool, layer, block_tables, ctx_lens, n_q, n_kv, head_dim, scale,
                 num_splits=1, ws_o=None, ws_ml=None):
    if _generic_attention(q, head_dim):
        return reference.paged_decode(q, out, pool, layer, block_tables, ctx_lens, n_q, n_kv,
                                      head_dim, scale)
    _bf16_cuda(q, out, pool)
    assert block_tables.dtype == torch.int32 and ctx_lens.dtype == torch.int32
    n = q.shape[0]
    assert block_tables.shape[0] >= n and ctx_lens.numel() >= n
    if num_splits > 1:
        if ws_o is None:
            ws_o = torch.empty((n, n_q, num_splits, head_dim), dtype=torch.float32, device=q.device)
            ws_ml = torch.empty((n, n_q, num_splits, 2), dtype=torch.float32, device=q.device)
        assert ws_o.numel() >= n * n_q * num_splits * head_dim
    base = pool.data_ptr() + layer * pool.stride(1) * 2
    rc = _kernels().vpsa_paged_decode(
        _p(q), c_long(q.stride(0)), _p(out), c_long(out.stride(0)), c_void_p(base),
        c_long(pool.stride(0)), c_long(pool.stride(2)), _p(block_tables),
        c_int(block_tables.stride(0)), _p(ctx_lens), _p(ws_o), _p(ws_ml), c_int(n), c_int(n_q),
        c_int(n_kv), c_int(num_splits), c_float(scale), _stream())
    _check(rc, "paged_decode")


# --------------------------------------------------------------------------- projections
SKINNY_MAX_ROWS = 64
_skinny_enabled = os.environ.get("VPSA_SKINNY_GEMM", "1") != "0"


def set_skinny_gemm(on: bool) -> bool:
    """Switch the decode-batch projection kernels on or off (A/B timing); returns the old value."""
    global _skinny_enabled
    old, _skinny_enabled = _skinny_enabled, bool(on)
    return old


def set_gemm_variant(v: int) -> int:
    """Tiling knobs of the decode-batch projection kernel (bit 0: eight loads per operand in
    flight instead of four, bit 1: four waves per workgroup); returns the previous value."""
    return int(_kernels().vpsa_set_gemm_variant(c_int(v)))


def _skinny(x2, w, bias, pair):
    n_out = w.shape[0] // 2 if pair else w.shape[0]
    out = torch.empty((x2.shape[0], n_out), dtype=x2.dtype, device=x2.device)
    rc = _kernels().vpsa_skinny_gemm(_p(x2), c_long(x2.stride(0)), _p(w), _p(bias), _p(out),
                                     c_long(out.stride(0)), c_int(x2.shape[0]), c_int(w.shape[0]),
                                     c_int(w.shape[1]), c_int(1 if pair else 0), _stream())
    return out if rc == 0 else None


def _stream_gemm(x2, w, bias, pair):
    """Streaming projection kernel (LDS-staged, reduction split across workgroups)."""
    lib = _kernels()
    m, n, k = x2.shape[0], w.shape[0], w.shape[1]
    splits = int(lib.vpsa_stream_gemm_splits(c_int(m), c_int(n), c_int(k), c_int(1 if pair else 0)))
    if splits <= 0:
        return None
    out = torch.empty((m, n // 2 if pair else n), dtype=x2.dtype, device=x2.device)
    ws = torch.empty((splits * m * n,), dtype=torch.float32, device=x2.device) if splits > 1 else None
    rc = lib.vpsa_stream_gemm(_p(x2), c_long(x2.stride(0)), _p(w), _p(bias), _p(out), c_long(out.stride(0)),
                              _p(ws), c_int(m), c_int(n), c_int(k), c_int(1 if pair else 0),
                              c_int(splits), _stream())
    return out if rc == 0 else None


def pack_weight(w: torch.Tensor) -> torch.Tensor:
    """[N, K] -> the same matrix in matrix-core tile order, [N / 16][K / 32][16][32] flattened
    (N a multiple of 16, K of 32): what ``packed_linear`` streams.  A second copy of the weight."""
    n, k = w.shape
    assert n % 16 == 0 and k % 32 == 0
    return w.view(n // 16, 16, k // 32, 32).permute(0, 2, 1, 3).contiguous().view(-1)


def set_packed_target(workgroups: int) -> int:
    """Workgroups (row blocks x reduction slices) the packed kernel aims for; returns the previous value."""
    return int(_kernels().vpsa_set_packed_target(c_int(workgroups)))


def packed_supported(n: int, k: int, pair: bool = False) -> bool:
    return int(_kernels().vpsa_packed_gemm_splits(c_int(n), c_int(k), c_int(1 if pair else 0))) > 0


def packed_linear(x2, wp, n: int, k: int, bias=None, pair: bool = False):
    """y = x2 @ W.T (+ bias), or silu(x2 @ Wg.T) * (x2 @ Wu.T) for ``pair`` (W = [Wg; Wu]), with W
    given as ``pack_weight(W)``; up to 64 rows.  ``None`` when the kernel does not tile the shape."""
    lib = _kernels()
    m = x2.shape[0]
    splits = int(lib.vpsa_packed_gemm_splits(c_int(n), c_int(k), c_int(1 if pair else 0)))
    if splits <= 0 or m > 64:
        return None
    out = torch.empty((m, n // 2 if pair else n), dtype=x2.dtype, device=x2.device)
    ws = torch.empty((splits * m * n,), dtype=torch.float32, device=x2.device) if splits > 1 else None
    rc = lib.vpsa_packed_gemm(_p(x2), c_long(x2.stride(0)), _p(wp), _p(bias), _p(out), c_long(out.stride(0)),
                              _p(ws), c_int(m), c_int(n), c_int(k), c_int(1 if pair else 0), c_int(splits),
                              _stream())
    return out if rc == 0 else None


def _skinny_ok(x, w, bias):
    return (_skinny_enabled and x.is_cuda and x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16
            and w.is_contiguous() and x.dim() == 2 and 0 < x.shape[0] <= SKINNY_MAX_ROWS
            and x.stride(1) == 1 and (bias is None or (bias.dtype == torch.bfloat16 and bias.is_contiguous())))


# Dispatch thresholds measured on MI355X (profiles/gemm_micro_graph.jsonl, HIP-graph replay,
# Llama-3-8B shapes, microseconds: library / register-direct / streaming kernel):
#   o-proj  4096x4096   1-16 rows: 14.9-15.0 /  9.6-11.4 / 11.1-14.4   -> register-direct
#   qkv     6144x4096   1-16 rows: 14.2      / 14.0-19.2 / 12.8-14.0   -> streaming
#   gate_up 28672x4096  1-16 rows: 60.8-61.9 (+ activation) / 55.7-63 / 41.3-45.2; 24-32 rows:
#                                  55 / - / 48.5-49.3; 40 rows: 53.5-55.4 / - / 52.7; 48 rows (where
#                                  the library's pick is a poor one): 63-65 / - / 53.7; 64 rows:
#                                  54.7 / - / 58.5                       -> streaming up to 48 rows
#   lm_head 128256x4096 1-16 rows: 187-189   / 209-288   / 173-184     -> streaming
#   down    4096x14336: the library wins at every batch (24.8 vs 25.6)  -> library
# From ~24 rows on (40 rows: qkv 17 vs 21, o 13 vs 18, down 36 vs 37) the library GEMM is as good or better.
SMALL_BATCH_ROWS = 16
PACKED_MAX_ROWS = 64
PAIR_STREAM_MAX_ROWS = 48
DIRECT_MAX_WEIGHT_ELEMS = 4096 * 4096
STREAM_MAX_K = 8192


# The prefill GEMM (``tile_gemm``: 256 x 256 tiles, one workgroup per CU) is used when its tiles fill
# the chip's 256 CUs in whole rounds: measured on MI355X against the library at 8192 rows 0.95-0.98x
# plain, 1.08x with the SwiGLU epilogue fused (profiles/r3_tile_gemm_micro_v1.jsonl); with the last
# round of tiles half empty (e.g. 4096 x 4096 at 4616 rows: 304 tiles = 1.19 rounds) the library's
# stream-K decomposition wins, and below ~1k rows every 256-row tile is mostly padding.
TILE_MIN_ROWS = int(os.environ.get("VPSA_TILE_MIN_ROWS", "1024"))
TILE_MIN_FILL = float(os.environ.get("VPSA_TILE_MIN_FILL", "0.85"))
TILE_CUS = 256
_tile_enabled = os.environ.get("VPSA_TILE_GEMM", "1") != "0"


def set_tile_gemm(on: bool) -> bool:
    """Switch the hand-written prefill GEMM on or off (A/B timing); returns the old value."""
    global _tile_enabled
    old, _tile_enabled = _tile_enabled, bool(on)
    return old


_shape_log = {} if os.environ.get("VPSA_SHAPE_LOG") else None
if _shape_log is not None:
    import atexit
    import sys
    atexit.register(lambda: print("VPSA_SHAPE_LOG", sorted(_shape_log.items(), key=lambda kv: -kv[1])[:60], file=sys.stderr, flush=True))


def tile_rows(x, w, pair: bool = False) -> int:
    """How many leading rows of ``x`` the prefill GEMM should take for ``x @ w.T``: all of them
    (a ragged last row tile is handled inside the kernel; a separate small GEMM for the remainder
    costs more than the padding), or 0 when the library is the better choice for this shape."""
    if not (_tile_enabled and x.is_cuda and x.dim() == 2 and x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16
            and w.dim() == 2 and w.is_contiguous() and x.stride(1) == 1 and x.stride(0) % 8 == 0):
        return 0
    m, (n, k) = x.shape[0], w.shape
    if _shape_log is not None:
        _shape_log[(m, n, k)] = _shape_log.get((m, n, k), 0) + 1
    if m < TILE_MIN_ROWS or not tile_gemm_supported(n, k, pair):
        return 0
    m_tiles = (m + 255) // 256
    n_tiles = (n // 2 + 127) // 128 if pair else (n + 255) // 256
    tiles = m_tiles * n_tiles
    rounds = (tiles + TILE_CUS - 1) // TILE_CUS
    if tiles / (rounds * TILE_CUS) < TILE_MIN_FILL:
        return 0
    return m


def linear(x, w, bias=None, packed=None):
    """y = x @ w.T (+ bias).  Prefill-sized batches run the hand-written 256 x 256 MFMA kernel
    (``csrc/kernels/tile_gemm.hip``) when its tiles fill the chip; decode batches of up to 16 rows
    the weight-streaming MFMA kernels of ``csrc/kernels/gemm.hip`` (which one depends on the
    weight's shape); everything else goes to the library GEMM (hipBLASLt).  ``packed``:
    ``pack_weight(w)`` when the caller keeps one (decode-only steps): up to 64 rows then stream
    that copy."""
    if packed is not None and x.dim() == 2 and x.shape[0] <= PACKED_MAX_ROWS and _skinny_ok(x, w, bias):
        y = packed_linear(x, packed, w.shape[0], w.shape[1], bias, False)
        if y is not None:
            return y
    if x.dim() == 2 and x.shape[0] <= SMALL_BATCH_ROWS and _skinny_ok(x, w, bias):
        y = None
        if w.numel() <= DIRECT_MAX_WEIGHT_ELEMS:
            y = _skinny(x, w, bias, False)
        elif w.shape[1] <= STREAM_MAX_K:
            y = _stream_gemm(x, w, bias, False)
        if y is not None:
            return y
    rows = tile_rows(x, w) if (bias is None or bias.dtype == torch.bfloat16) else 0
    if rows:
        m = x.shape[0]
        out = torch.empty((m, w.shape[0]), dtype=x.dtype, device=x.device)
        # a bias is a "residual" with row stride 0
        tile_gemm(x[:rows], w, residual=None if bias is None else bias.view(1, -1).expand(rows, -1), out=out[:rows])
        if rows < m:
            out[rows:] = linear(x[rows:], w, bias)
        return out
    return torch.nn.functional.linear(x, w, bias)


def linear_add(x, w, residual, packed=None):
    """The residual-stream update of an output projection: ``residual += x @ w.T`` inside the GEMM's
    epilogue when the prefill kernel takes the shape (returns None: nothing is left to add), else
    returns ``x @ w.T`` for the caller's fused add + norm."""
    rows = tile_rows(x, w) if residual.is_contiguous() else 0
    if not rows:
        return linear(x, w, None, packed)
    tile_gemm(x[:rows], w, residual=residual[:rows], out=residual[:rows])
    if rows < x.shape[0]:
        residual[rows:] += linear(x[rows:], w)
    return None


# The library's pick for the fused q/k/v width of Llama-3-8B (6144 columns) is 12-14 % slower per
# column than its pick for 4096 and 2048 columns at 8192 rows, 7 % at 4096 rows, and faster below
# (profiles/r2_gemm_split_probe.jsonl); other projections gain nothing from a split.
SPLIT_QKV_MIN_ROWS = 4096


def qkv_linear(x, w, bias, q_cols: int):
    """The fused q/k/v projection: ``linear``, except that prefill-sized batches which the
    hand-written kernel does not take and that have no bias go to the library as two GEMMs,
    [q | k v], each writing its column range of ONE output buffer (same bits as the single call)."""
    if (bias is None and x.is_cuda and x.dim() == 2 and x.shape[0] >= SPLIT_QKV_MIN_ROWS
            and x.dtype == torch.bfloat16 and 0 < q_cols < w.shape[0] and not tile_rows(x, w)):
        out = torch.empty((x.shape[0], w.shape[0]), dtype=x.dtype, device=x.device)
        torch.matmul(x, w[:q_cols].t(), out=out[:, :q_cols])
        torch.matmul(x, w[q_cols:].t(), out=out[:, q_cols:])
        return out
    return linear(x, w, bias)


def gate_up_silu(x, w_gate_up, packed=None):
    """silu(x @ Wg.T) * (x @ Wu.T) for ``w_gate_up = [Wg; Wu]``: fused in one kernel for decode
    batches (weight-streaming kernels) and for prefill batches (SwiGLU epilogue of the 256 x 256
    MFMA kernel); library GEMM + activation kernel for what neither takes.  ``packed`` as in ``linear``."""
    if packed is not None and x.dim() == 2 and x.shape[0] <= PACKED_MAX_ROWS and _skinny_ok(x, w_gate_up, None):
        y = packed_linear(x, packed, w_gate_up.shape[0], w_gate_up.shape[1], None, True)
        if y is not None:
            return y
    if x.dim() == 2 and x.shape[0] <= PAIR_STREAM_MAX_ROWS and _skinny_ok(x, w_gate_up, None):
        y = _stream_gemm(x, w_gate_up, None, True)
        if y is None and x.shape[0] <= 8:
            y = _skinny(x, w_gate_up, None, True)
        if y is not None:
            return y
    rows = tile_rows(x, w_gate_up, True)
    if rows:
        m = x.shape[0]
        out = torch.empty((m, w_gate_up.shape[0] // 2), dtype=x.dtype, device=x.device)
        tile_gemm(x[:rows], w_gate_up, silu_mul=True, out=out[:rows])
        if rows < m:
            silu_and_mul(torch.nn.functional.linear(x[rows:], w_gate_up), out=out[rows:])
        return out
    return silu_and_mul(torch.nn.functional.linear(x, w_gate_up))


def hbm_read_probe(buf, blocks=2048):
    """Launch the read-bandwidth probe over ``buf`` (timing is the caller's)."""
    sink = torch.zeros(1, dtype=torch.int32, device=buf.device)
    rc = _kernels().vpsa_hbm_read_probe(_p(buf), c_long(buf.numel() * buf.element_size()), _p(sink),
                                        c_int(blocks), _stream())
    _check(rc, "hbm_read_probe")


# --------------------------------------------------------------------------- MLP / sampling
def silu_and_mul(x, out=None):
    if not x.is_cuda:
        y = reference.silu_and_mul(x)
        if out is not None:
            out.copy_(y)
            return out
        return y
    _bf16_cuda(x)
    inter = x.shape[-1] // 2
    x2 = x.view(-1, 2 * inter)
    assert x2.is_contiguous()
    if out is None:
        out = torch.empty((x2.shape[0], inter), dtype=x.dtype, device=x.device)
    rc = _kernels().vpsa_silu_mul(_p(x2), _p(out), c_long(x2.shape[0]), c_int(inter), _stream())
    _check(rc, "silu_and_mul")
    return out.view(*x.shape[:-1], inter)


def argmax(logits, out=None):
    if not logits.is_cuda:
        return reference.argmax(logits)
    if logits.dtype != torch.bfloat16 or logits.stride(-1) != 1 or logits.stride(0) % 8 != 0:
        return torch.argmax(logits, dim=-1)
    if out is None:
        out = torch.empty((logits.shape[0],), dtype=torch.int64, device=logits.device)
    rc = _kernels().vpsa_argmax_bf16(_p(logits), _p(out), c_int(logits.shape[0]),
                                     c_int(logits.shape[1]), c_long(logits.stride(0)), _stream())
    _check(rc, "argmax")
    return out


def sample_supported(logits) -> bool:
    return (logits.is_cuda and logits.dtype == torch.bfloat16 and logits.dim() == 2
            and logits.stride(1) == 1 and logits.stride(0) % 8 == 0 and logits.data_ptr() % 16 == 0)


def sample(logits, params, uniform, rows=None, want_stats=False):
    """Draw one token per sample from bf16 ``logits`` [*, vocab] in one launch.
    ``params`` float32 [n, 3] = (temperature, top_p, top_k; top_k <= 0 and top_p >= 1 mean off),
    ``uniform`` float32 [n] in [0, 1), ``rows`` int32 [n] (logits row of each sample; default
    sample i reads row i).  -> int64 token ids [n] (and, for tests, [n, 3] = threshold key,
    kept mass relative to the largest probability, kept count)."""
    if not sample_supported(logits):
        raise TypeError("sample: bf16 device logits with 16-byte aligned rows")
    n = params.shape[0]
    out = torch.empty((n,), dtype=torch.int64, device=logits.device)
    stats = torch.empty((n, 3), dtype=torch.float32, device=logits.device) if want_stats else None
    assert params.dtype == torch.float32 and params.is_contiguous() and uniform.dtype == torch.float32
    assert rows is None or (rows.dtype == torch.int32 and rows.numel() == n)
    rc = _kernels().vpsa_sample(_p(logits), c_long(logits.stride(0)), c_int(logits.shape[1]), _p(rows),
                                _p(params), _p(uniform), _p(out), _p(stats), c_int(n), _stream())
    _check(rc, "sample")
    return (out, stats) if want_stats else out


# --------------------------------------------------------------------------- KV page movement
def kv_gather_pages(pool, page_ids, staging):
    """Copy whole pages (all layers) ``pool[page_ids]`` into contiguous ``staging``."""
    if not pool.is_cuda:
        staging.view(-1)[: page_ids.numel() * pool[0].numel()].view(page_ids.numel(), -1).copy_(
            pool[page_ids.long()].reshape(page_ids.numel(), -1))
        return
    assert page_ids.dtype == torch.int64 and page_ids.is_cuda
    page_bytes = pool.stride(0) * pool.element_size()
    assert staging.numel() * staging.element_size() >= page_ids.numel() * page_bytes
    rc = _kernels().vpsa_kv_gather_pages(_p(pool), _p(page_ids), _p(staging),
                                         c_int(page_ids.numel()), c_long(page_bytes), _stream())
    _check(rc, "kv_gather_pages")


def kv_scatter_pages(staging, page_ids, pool):
    if not pool.is_cuda:
        pool[page_ids.long()] = staging.view(-1)[: page_ids.numel() * pool[0].numel()].view(
            page_ids.numel(), *pool.shape[1:]).to(pool.dtype)
        return
    assert page_ids.dtype == torch.int64 and page_ids.is_cuda
    page_bytes = pool.stride(0) * pool.element_size()
    rc = _kernels().vpsa_kv_scatter_pages(_p(staging), _p(page_ids), _p(pool),
                                          c_int(page_ids.numel()), c_long(page_bytes), _stream())
    _check(rc, "kv_scatter_pages")


def cachegen_encode(staging, levels, unit_elems, n_layers):
    """bf16 staging -> (int8 symbols, fp32 group scales).  ``levels``: int32 tensor [n_layers]."""
    if not staging.is_cuda:
        return reference.cachegen_encode(staging, levels.tolist(), unit_elems, n_layers)
    n = staging.numel()
    sym = torch.empty((n,), dtype=torch.int8, device=staging.device)
    scales = torch.empty((n // 64,), dtype=torch.float32, device=staging.device)
    assert levels.dtype == torch.int32 and levels.is_cuda
    rc = _kernels().vpsa_cachegen_encode(_p(staging), _p(sym), _p(scales), _p(levels), c_long(n),
                                         c_long(unit_elems), c_int(n_layers), _stream())
    _check(rc, "cachegen_encode")
    return sym, scales


def cachegen_decode(symbols, scales, unit_elems, out=None):
    if not symbols.is_cuda:
        y = reference.cachegen_decode(symbols, scales, unit_elems)
        if out is not None:
            out.view(-1).copy_(y)
            return out
        return y
    n = symbols.numel()
    if out is None:
        out = torch.empty((n,), dtype=torch.bfloat16, device=symbols.device)
    rc = _kernels().vpsa_cachegen_decode(_p(symbols), _p(scales), _p(out), c_long(n),
                                         c_long(unit_elems), _stream())
    _check(rc, "cachegen_decode")
    return out


# --------------------------------------------------------------------------- rANS entropy stage
RANS_SPC = 512  # streams per context: 256 symbols per stream for a Llama-3-8B layer (K + V)


def rans_encode(sym, n_ctx, spc=RANS_SPC):
    """Entropy-code int8/uint8 symbols (``n_ctx`` equal contexts) with the interleaved rANS
    coder.  -> (freq int16 [n_ctx * 256], states int32 [n_ctx * spc], sizes int16 [n_ctx * spc],
    coded uint8 [sum of sizes]); the 16/32-bit tensors hold unsigned values bit for bit.
    Synchronises once (the coded length sizes the output)."""
    if not sym.is_cuda:
        import numpy as np

        f, st, sz, co = reference.rans_encode(sym.contiguous().view(torch.uint8).numpy(), n_ctx, spc)
        return (torch.from_numpy(f.view(np.int16).reshape(-1).copy()), torch.from_numpy(st.view(np.int32).copy()),
                torch.from_numpy(sz.view(np.int16).copy()), torch.from_numpy(co.copy()))
    sym = sym.contiguous().view(torch.uint8).view(-1)
    n, dev = sym.numel(), sym.device
    n_streams = n_ctx * spc
    if n % n_ctx or (n // n_ctx) % spc:
        raise ValueError("rans_encode: symbols do not divide into contexts / streams")
    cap = 2 * (n // n_ctx // spc)
    hist = torch.zeros(n_ctx * 256, dtype=torch.int32, device=dev)
    freq = torch.empty(n_ctx * 256, dtype=torch.int16, device=dev)
    cum = torch.empty(n_ctx * 256, dtype=torch.int16, device=dev)
    scratch = torch.empty(n_streams * cap, dtype=torch.uint8, device=dev)
    states = torch.empty(n_streams, dtype=torch.int32, device=dev)
    sizes = torch.empty(n_streams, dtype=torch.int16, device=dev)
    err = torch.zeros(1, dtype=torch.int32, device=dev)
    lib = _kernels()
    _check(lib.vpsa_rans_encode(_p(sym), c_long(n), c_int(n_ctx), c_int(spc), _p(hist), _p(freq), _p(cum),
                                _p(scratch), _p(states), _p(sizes), _p(err), _stream()), "rans_encode")
    sz64 = sizes.to(torch.int64)
    ends = torch.cumsum(sz64, 0)
    offsets = ends - sz64
    total, bad = torch.stack([ends[-1], err[0].to(torch.int64)]).tolist()  # one synchronisation
    if bad:
        raise RuntimeError("rans_encode: frequency table could not be normalised")
    coded = torch.empty(int(total), dtype=torch.uint8, device=dev)
    _check(lib.vpsa_rans_pack(_p(scratch), _p(sizes), _p(offsets), _p(coded), c_long(n), c_int(n_ctx),
                              c_int(spc), _stream()), "rans_pack")
    return freq, states, sizes, coded


def rans_decode(freq, states, sizes, coded, n, n_ctx, spc=RANS_SPC):
    """Inverse of ``rans_encode`` -> int8 symbols [n] on the device of ``coded``."""
    if not coded.is_cuda:
        import numpy as np

        out = reference.rans_decode(freq.numpy().view(np.uint16), states.numpy().view(np.uint32),
                                    sizes.numpy().view(np.uint16), coded.numpy(), n, n_ctx, spc)
        return torch.from_numpy(out.view(np.int8).copy())
    dev = coded.device
    sz64 = sizes.to(torch.int64)
    offsets = torch.cumsum(sz64, 0) - sz64
    cum = torch.empty(n_ctx * 256, dtype=torch.int16, device=dev)
    sym = torch.empty(n, dtype=torch.int8, device=dev)
    _check(_kernels().vpsa_rans_decode(_p(coded), c_long(coded.numel()), _p(offsets), _p(states), _p(freq),
                                       _p(cum), _p(sym), c_long(n), c_int(n_ctx), c_int(spc), _stream()),
           "rans_decode")
    return sym


# --------------------------------------------------------------------------- MoE routing
def moe_topk_softmax(logits, topk, renorm=True):
    if not logits.is_cuda:
        return reference.moe_topk_softmax(logits, topk, renorm)
    _bf16_cuda(logits)
    t, e = logits.shape
    w = torch.empty((t, topk), dtype=torch.float32, device=logits.device)
    ids = torch.empty((t, topk), dtype=torch.int32, device=logits.device)
    rc = _kernels().vpsa_moe_topk_softmax(_p(logits.contiguous()), _p(w), _p(ids), c_int(t),
                                          c_int(e), c_int(topk), c_int(1 if renorm else 0),
                                          _stream())
    _check(rc, "moe_topk_softmax")
    return w, ids


def moe_localize(ids, weights, first, count, bucket):
    """Routing of a rank that holds experts ``first .. first + count`` only: ids become local, and a
    pair routed elsewhere gets weight 0 and the id ``count`` (``bucket``: a sorting bucket of its
    own that ``moe_experts_grouped`` / ``moe_experts_prefill`` skip with ``foreign=True``) or 0."""
    local = ids - first
    here = (local >= 0) & (local < count)
    elsewhere = count if bucket else 0
    return (torch.where(here, local, elsewhere).to(torch.int32),
            torch.where(here, weights, 0.0).to(weights.dtype))


def moe_align(ids, n_experts):
    """ids [T, K] int32 -> (offsets [E+1], pos [T*K], sorted_tok [T*K])."""
    t, k = ids.shape
    if not ids.is_cuda:
        flat = ids.reshape(-1).long()
        order = torch.sort(flat, stable=True).indices
        pos = torch.empty_like(order)
        pos[order] = torch.arange(order.numel())
        counts = torch.bincount(flat, minlength=n_experts)
        offsets = torch.zeros(n_experts + 1, dtype=torch.int32)
        offsets[1:] = counts.cumsum(0)
        return offsets, pos.int(), (order // k).int()
    offsets = torch.empty((n_experts + 1,), dtype=torch.int32, device=ids.device)
    pos = torch.empty((t * k,), dtype=torch.int32, device=ids.device)
    sorted_tok = torch.empty((t * k,), dtype=torch.int32, device=ids.device)
    rc = _kernels().vpsa_moe_align(_p(ids.contiguous()), _p(offsets), _p(pos), _p(sorted_tok),
                                   c_int(t * k), c_int(n_experts), c_int(k), _stream())
    _check(rc, "moe_align")
    return offsets, pos, sorted_tok


def moe_permute(x, sorted_tok):
    if not x.is_cuda:
        return x[sorted_tok.long()]
    _bf16_cuda(x)
    xs = torch.empty((sorted_tok.numel(), x.shape[1]), dtype=x.dtype, device=x.device)
    rc = _kernels().vpsa_moe_permute(_p(x.contiguous()), _p(sorted_tok), _p(xs),
                                     c_int(sorted_tok.numel()), c_int(x.shape[1]), _stream())
    _check(rc, "moe_permute")
    return xs


def moe_combine(y, weights, pos, tokens):
    k = weights.shape[1]
    if not y.is_cuda:
        g = y[pos.long()].view(tokens, k, -1).float()
        return (g * weights[:, :, None]).sum(1).to(y.dtype)
    _bf16_cuda(y)
    out = torch.empty((tokens, y.shape[1]), dtype=y.dtype, device=y.device)
    rc = _kernels().vpsa_moe_combine(_p(y.contiguous()), _p(weights), _p(pos), _p(out),
                                     c_int(tokens), c_int(y.shape[1]), c_int(k), _stream())
    _check(rc, "moe_combine")
    return out


# Decode batches up to this many tokens take the expert-grouped matrix-core kernels.  Measured
# on one Mixtral-8x7B layer (MI355X): 0.48-0.54 ms for 4-32 tokens (5.2 TB/s of expert weights)
# against 0.57-0.59 ms for all-experts batched GEMM; from ~64 tokens on the token tiles are
# re-read through L2 by every wave and the batched GEMM is faster.
MOE_MFMA_MAX_TOKENS = int(os.environ.get("VPSA_MOE_MFMA_MAX_TOKENS", "48"))


def moe_mfma_supported(hidden: int, inter: int) -> bool:
    return hidden % 128 == 0 and inter % 512 == 0


def moe_experts_grouped(x, w13, w2, ids, weights, foreign=False):
    """Expert MLP for a decode-sized batch: out[t] = sum_k weights[t,k] * MLP_{ids[t,k]}(x[t]).
    ``foreign``: ids may equal the expert count (a pair of an expert held by another rank, weight
    0): those pairs sort behind the last expert and no kernel touches them.

    Same contract as ``moe_experts_small_batch``; here the (token, k) pairs are sorted by expert
    first, and each expert's weights are streamed once against all of its tokens on the matrix
    cores.  Three launches after the sort (gate_up+SiLU, down, weighted combine), no host
    synchronisation: HIP-graph capturable."""
    t, k = ids.shape
    if not x.is_cuda:
        return reference.moe_experts(x, w13, w2, ids, weights)
    _bf16_cuda(x, w13, w2)
    n_experts, hidden, inter = w13.shape[0], x.shape[1], w2.shape[2]
    assert w13.shape[1] == 2 * inter and w13.shape[2] == hidden and w2.shape[1] == hidden
    assert w13.is_contiguous() and w2.is_contiguous() and ids.dtype == torch.int32
    assert moe_mfma_supported(hidden, inter), (hidden, inter)
    offsets, pos, sorted_tok = moe_align(ids, n_experts + (1 if foreign else 0))
    act = torch.empty((t * k, inter), dtype=x.dtype, device=x.device)
    y = torch.empty((t * k, hidden), dtype=x.dtype, device=x.device)
    rc = _kernels().vpsa_moe_mfma(_p(x.contiguous()), _p(w13), _p(w2), _p(offsets), _p(sorted_tok),
                                  _p(act), _p(y), c_int(t), c_int(k), c_int(n_experts),
                                  c_int(hidden), c_int(inter), _stream())
    _check(rc, "moe_mfma")
    return moe_combine(y, weights, pos, t)


def grouped_gemm(a, w, offsets, rows=None, arow_idx=None, silu_mul=False, out=None):
    """out[r, :] = a[arow_idx[r] if given else r, :] @ w[g]^T for offsets[g] <= r < offsets[g+1].

    a [*, K] bf16; w [G, N, K] bf16 contiguous; offsets [G + 1] int32 ON THE DEVICE (the host never
    reads them: one launch for all groups, no synchronisation, HIP-graph capturable); rows = upper
    bound on offsets[G] (default: a.shape[0], or len(arow_idx)).  silu_mul: w[g] is [gate rows | up
    rows] and the result is silu(gate) * up, [rows, N / 2].  Rows at or past offsets[G] are left
    unwritten.  CPU tensors take the PyTorch definition of the same product."""
    g, n, k = w.shape
    if rows is None:
        rows = a.shape[0] if arow_idx is None else arow_idx.numel()
    n_out = n // 2 if silu_mul else n
    if not a.is_cuda:
        off = offsets.tolist()
        res = torch.zeros((rows, n_out), dtype=a.dtype) if out is None else out
        src = a if arow_idx is None else a[arow_idx.long()]
        for i in range(g):
            lo, hi = off[i], off[i + 1]
            if hi > lo:
                y = src[lo:hi].float() @ w[i].float().t()
                if silu_mul:
                    y = torch.nn.functional.silu(y[:, :n_out]) * y[:, n_out:]
                res[lo:hi] = y.to(a.dtype)
        return res
    _bf16_cuda(a, w)
    assert w.is_contiguous() and a.stride(-1) == 1 and a.shape[1] == k and offsets.dtype == torch.int32
    assert offsets.numel() == g + 1 and offsets.is_cuda
    if arow_idx is not None:
        assert arow_idx.dtype == torch.int32 and arow_idx.is_cuda and arow_idx.numel() >= rows
    if out is None:
        out = torch.empty((rows, n_out), dtype=a.dtype, device=a.device)
    assert out.stride(-1) == 1 and out.shape[0] >= rows and out.shape[1] == n_out
    rc = _kernels().vpsa_grouped_gemm(_p(a), c_long(a.stride(0)), _p(arow_idx),
                                      _p(w), _p(offsets), c_int(g), _p(out), c_long(out.stride(0)), c_int(rows),
                                      c_int(n), c_int(k), c_int(1 if silu_mul else 0), _stream())
    _check(rc, "grouped_gemm")
    return out


def tile_gemm_supported(n: int, k: int, silu_mul: bool = False) -> bool:
    return k % 64 == 0 and n % 8 == 0 and (not silu_mul or (n // 2) % 32 == 0) and n * k * 2 < 2 ** 32


def tile_gemm(a, w, offsets=None, rows=None, arow_idx=None, silu_mul=False, residual=None, out=None):
    """The prefill GEMM (``csrc/kernels/tile_gemm.hip``): out[r] = epilogue(a[arow_idx[r] or r] @ w[g]^T).

    a [*, K] bf16; w [N, K] (dense) or [G, N, K] (grouped, with offsets [G + 1] int32 ON THE DEVICE:
    group g owns output rows offsets[g] .. offsets[g+1], no host synchronisation).  silu_mul: w is
    [gate rows | up rows], out = silu(gate) * up, [rows, N / 2].  residual [rows, N]: out = product +
    residual (``out`` may be ``residual`` itself).  Rows at or past offsets[G] are left unwritten."""
    if w.dim() == 2:
        w = w.unsqueeze(0)
    g, n, k = w.shape
    if rows is None:
        rows = a.shape[0] if arow_idx is None else arow_idx.numel()
    n_out = n // 2 if silu_mul else n
    _bf16_cuda(a, w, residual)
    assert w.is_contiguous() and a.stride(-1) == 1 and a.shape[1] == k
    assert not (silu_mul and residual is not None)
    if offsets is not None:
        assert offsets.dtype == torch.int32 and offsets.numel() == g + 1 and offsets.is_cuda
    else:
        assert g == 1
    if arow_idx is not None:
        assert arow_idx.dtype == torch.int32 and arow_idx.is_cuda and arow_idx.numel() >= rows
    if out is None:
        out = torch.empty((rows, n_out), dtype=a.dtype, device=a.device)
    assert out.stride(-1) == 1 and out.shape[0] >= rows and out.shape[1] == n_out
    if residual is not None:
        assert residual.stride(-1) == 1 and residual.shape[0] >= rows and residual.shape[1] == n_out
    epi = 1 if silu_mul else (2 if residual is not None else 0)
    rc = _kernels().vpsa_tile_gemm(_p(a), c_long(a.stride(0)), _p(arow_idx), _p(w), _p(offsets), c_int(g), _p(out),
                                   c_long(out.stride(0)), c_int(rows), c_int(n), c_int(k), c_int(epi), _p(residual),
                                   c_long(residual.stride(0) if residual is not None else 0), c_long(a.shape[0]),
                                   _stream())
    _check(rc, "tile_gemm")
    return out


def grouped_gemm_supported(n: int, k: int, silu_mul: bool = False) -> bool:
    return k % 64 == 0 and (not silu_mul or (n // 2) % 32 == 0)


def moe_experts_prefill(x, w13, w2, ids, weights, gather=True, foreign=False):
    """Expert MLP for prefill-sized batches: the (token, k) pairs are sorted by expert on the device
    and two grouped matrix-core GEMMs run over all experts at once: gate_up gathers its token
    rows by index and fuses silu(gate) * up, down writes the sorted rows that the weighted combine
    gathers back.  Four launches, no host synchronisation."""
    t, k = ids.shape
    if not x.is_cuda:
        return reference.moe_experts(x, w13, w2, ids, weights)
    n_experts = w13.shape[0]
    offsets, pos, sorted_tok = moe_align(ids, n_experts + (1 if foreign else 0))
    if foreign:  # the bucket of pairs held elsewhere ends the list: the GEMMs stop at offsets[n_experts]
        offsets = offsets[: n_experts + 1]
    gg = tile_gemm if (tile_gemm_supported(w13.shape[1], w13.shape[2], True)
                       and tile_gemm_supported(w2.shape[1], w2.shape[2])) else grouped_gemm
    if gather:
        act = gg(x, w13, offsets, rows=t * k, arow_idx=sorted_tok, silu_mul=True)
    else:  # copy the rows into sorted order first (timing comparison)
        act = gg(moe_permute(x, sorted_tok), w13, offsets, rows=t * k, silu_mul=True)
    y = gg(act, w2, offsets, rows=t * k)
    return moe_combine(y, weights, pos, t)


def moe_experts_small_batch(x, w13, w2, ids, weights):
    """Expert MLP for a handful of tokens (decode): out[t] = sum_k weights[t,k] * MLP_{ids[t,k]}(x[t]).

    x [T, H] bf16, w13 [E, 2I, H] (gate rows then up rows), w2 [E, H, I], ids [T, K] int32,
    weights [T, K] fp32.  On the GPU only the routed experts' weights are streamed (two GEMV
    kernels + the weighted combine); no host synchronisation, HIP-graph capturable."""
    t, k = ids.shape
    if not x.is_cuda:
        return reference.moe_experts(x, w13, w2, ids, weights)
    _bf16_cuda(x, w13, w2)
    hidden, inter = x.shape[1], w2.shape[2]
    assert w13.shape[1] == 2 * inter and w13.shape[2] == hidden and w2.shape[1] == hidden
    assert w13.is_contiguous() and w2.is_contiguous() and ids.dtype == torch.int32
    act = torch.empty((t * k, inter), dtype=x.dtype, device=x.device)
    y = torch.empty((t * k, hidden), dtype=x.dtype, device=x.device)
    rc = _kernels().vpsa_moe_gemv(_p(x.contiguous()), _p(w13), _p(w2), _p(ids.contiguous()), _p(act), _p(y),
                                  c_int(t), c_int(k), c_int(hidden), c_int(inter), _stream())
    _check(rc, "moe_gemv")
    pos = torch.arange(t * k, dtype=torch.int32, device=x.device)
    return moe_combine(y, weights, pos, t)


# ---------------------------------------------------------------------------- LoRA
def lora_shrink(x, a, slots, rank_total):
    """h[t] = A[slots[t]] @ x[t] for rows with slots[t] >= 0 (other rows are left unwritten).

    x [T, in] bf16, a [S, R, in] bf16 (R = stacked, padded ranks), slots [T] int32 -> h [T, R] bf16."""
    t = x.shape[0]
    if not x.is_cuda:
        return reference.lora_shrink(x, a, slots)
    _bf16_cuda(x, a)
    assert a.is_contiguous() and a.shape[1] == rank_total and a.shape[2] == x.shape[1]
    assert slots.dtype == torch.int32 and slots.numel() == t and x.stride(1) == 1
    h = torch.empty((t, rank_total), dtype=x.dtype, device=x.device)
    rc = _kernels().vpsa_lora_shrink(_p(x), c_long(x.stride(0)), _p(a), _p(slots), _p(h), c_int(t),
                                     c_int(x.shape[1]), c_int(rank_total), c_int(a.shape[0]), _stream())
    _check(rc, "lora_shrink")
    return h


def lora_expand(h, bs, slots, y, r_pad):
    """For each sub-module i (the parts of a fused projection, side by side in y and in h):
    y[t, off_i : off_i + out_i] += B_i[slots[t]] @ h[t, i * r_pad : (i + 1) * r_pad], slots[t] >= 0.

    h [T, n * r_pad] bf16, bs: list of [S, out_i, r_pad] bf16 (adapter scale folded in),
    y [T, >= sum(out_i)] bf16, updated in place by one launch."""
    if not h.is_cuda:
        off = 0
        for i, b in enumerate(bs):
            reference.lora_expand(h, i * r_pad, b, slots, y, off)
            off += b.shape[1]
        return y
    _bf16_cuda(h, y, *bs)
    assert h.is_contiguous() and y.stride(1) == 1 and 1 <= len(bs) <= 3
    n = len(bs)
    outs = [b.shape[1] for b in bs]
    assert all(b.is_contiguous() and b.shape[2] == r_pad for b in bs) and sum(outs) <= y.shape[1]
    y_offs = [sum(outs[:i]) for i in range(n)]
    ptrs = (c_void_p * n)(*[b.data_ptr() for b in bs])
    rc = _kernels().vpsa_lora_expand(_p(h), c_int(h.stride(0)), ptrs, (c_int * n)(*outs),
                                     (c_int * n)(*y_offs), (c_int * n)(*[i * r_pad for i in range(n)]),
                                     c_int(n), _p(slots), _p(y), c_long(y.stride(0)), c_int(h.shape[0]),
                                     c_int(r_pad), c_int(bs[0].shape[0]), _stream())
    _check(rc, "lora_expand")
    return y
