"""Numerics of every HIP kernel against the plain-PyTorch fp32 definition of the same op.

Runs only on a real MI355X (``-m gpu``).  The reference functions are executed on the
same device in fp32; inputs are random (never zero-filled) and sized so that edge paths
(ragged tails, partially filled pages, masked tiles, multi-split decode) are all hit.
"""

import math

import numpy as np

import pytest
import torch

from vllm_production_stack_amd import ops
from vllm_production_stack_amd.ops import reference as ref

pytestmark = pytest.mark.gpu

DEV = "cuda"
BF = torch.bfloat16


def _pool(pages, layers, n_kv, hd=128, seed=0):
    g = torch.Generator(device=DEV).manual_seed(seed)
    return torch.randn((pages, layers, 2, n_kv, ops.PAGE * hd), device=DEV, dtype=torch.float32,
                       generator=g).to(BF)


def test_native_library_loaded():
    x = torch.randn(4, 256, device=DEV).to(BF)
    ops.rms_norm(x, torch.ones(256, device=DEV, dtype=BF), 1e-5)
    assert ops.native_loaded()


@pytest.mark.parametrize("hidden", [128, 4096, 5120, 8192])
@pytest.mark.parametrize("with_res", [False, True])
def test_rmsnorm(hidden, with_res):
    torch.manual_seed(1)
    x = torch.randn(37, hidden, device=DEV).to(BF)
    w = (torch.rand(hidden, device=DEV) + 0.5).to(BF)
    res = torch.randn(37, hidden, device=DEV).to(BF) if with_res else None
    res_ref = res.clone() if with_res else None
    y = ops.rms_norm(x, w, 1e-5, res)
    y_ref = ref.rms_norm(x, w, 1e-5, res_ref)
    torch.testing.assert_close(y.float(), y_ref.float(), atol=2e-2, rtol=2e-2)
    if with_res:
        torch.testing.assert_close(res.float(), res_ref.float(), atol=1e-2, rtol=1e-2)


def test_rmsnorm_strided_input():
    torch.manual_seed(2)
    big = torch.randn(9, 3 * 4096, device=DEV).to(BF)
    x = big[:, 4096:8192]
    w = torch.ones(4096, device=DEV, dtype=BF)
    y = ops.rms_norm(x, w, 1e-6)
    torch.testing.assert_close(y.float(), ref.rms_norm(x, w, 1e-6).float(), atol=2e-2, rtol=2e-2)


@pytest.mark.parametrize("n_q,n_kv", [(32, 8), (8, 8), (16, 2), (28, 4)])
def test_rope_and_cache(n_q, n_kv):
    torch.manual_seed(3)
    t, hd, layers, pages = 150, 128, 3, 8
    qkv = torch.randn(t, (n_q + 2 * n_kv) * hd, device=DEV).to(BF)
    qkv_ref = qkv.clone()
    pos = torch.randint(0, 4000, (t,), device=DEV, dtype=torch.int32)
    slots = torch.randperm(pages * ops.PAGE, device=DEV)[:t].to(torch.int64)
    slots[5] = -1
    cs = ops.build_cos_sin(hd, 4096, 500000.0, device=DEV)
    pool = _pool(pages, layers, n_kv)
    pool_ref = pool.clone()
    ops.rope_and_cache(qkv, pos, slots, cs, pool, 1, n_q, n_kv, hd)
    ref.rope_and_cache(qkv_ref, pos, slots, cs, pool_ref, 1, n_q, n_kv, hd)
    torch.testing.assert_close(qkv.float(), qkv_ref.float(), atol=2e-2, rtol=2e-2)
    torch.testing.assert_close(pool.float(), pool_ref.float(), atol=2e-2, rtol=2e-2)
    # untouched layers stay bit-identical
    assert torch.equal(pool[:, 0], pool_ref[:, 0]) and torch.equal(pool[:, 2], pool_ref[:, 2])
    # the tile kernel for prefill rows: same cache contents from scattered slots, a skipped row and
    # a ragged last tile; qkv is read only
    torch.manual_seed(3)
    qkv2 = torch.randn(t, (n_q + 2 * n_kv) * hd, device=DEV).to(BF)
    keep = qkv2.clone()
    pool2 = _pool(pages, layers, n_kv)
    ops.kv_cache_write(qkv2, pos, slots, cs, pool2, 1, n_q, n_kv, hd)
    assert torch.equal(qkv2, keep)
    torch.testing.assert_close(pool2.float(), pool_ref.float(), atol=2e-2, rtol=2e-2)
    assert torch.equal(pool2[:, 0], pool_ref[:, 0]) and torch.equal(pool2[:, 2], pool_ref[:, 2])


def _block_tables(n_seqs, max_pages, total_pages, seed):
    g = torch.Generator().manual_seed(seed)
    perm = torch.randperm(total_pages, generator=g)[: n_seqs * max_pages]
    return perm.view(n_seqs, max_pages).to(torch.int32).to(DEV)


def _v_offset_pool(pages, layers, n_kv, hd=128, seed=0):
    """KV pool whose V channels carry a per-channel offset (|offset| up to 2): an attention output
    is then far from zero in every channel, so averaging over many keys cannot hide an error."""
    pool = _pool(pages, layers, n_kv, hd, seed)
    g = torch.Generator(device=DEV).manual_seed(seed + 1000)
    off = (torch.rand(hd, device=DEV, generator=g) * 4 - 2)
    # V part of a KV head: [key half][channel][32 keys]
    v = pool[:, :, 1].view(pages, layers, n_kv, ops.PAGE // 32, hd, 32).float()
    v += off[None, None, None, None, :, None]
    pool[:, :, 1] = v.view(pages, layers, n_kv, hd * ops.PAGE).to(BF)
    return pool


def _rel_row_err(out, out_ref, n_heads, hd=128):
    """max over (row, head) of max|delta| / max|reference| within that head's 128 channels."""
    o = out.float().view(-1, n_heads, hd)
    r = out_ref.float().view(-1, n_heads, hd)
    num = (o - r).abs().amax(-1)
    den = r.abs().amax(-1).clamp_min(1e-6)
    return (num / den).max().item()


REL_TOL = 2e-2


def _prefill_variant_ok(variant, n_q, n_kv):
    """Generations 1-2 (variants 0-3) take ratios that divide 4 or are multiples of it ... and 5, 6,
    7 with a part-filled last workgroup, but not 3; generation 3 (4-7) multiples of 4; generation 4
    (8-11, and the automatic choice) every ratio from 3 up (spare waves pad the last head group)."""
    g = n_q // n_kv
    if variant < 0:
        return True
    if variant < 4:
        return g != 3
    if variant < 8:
        return g % 4 == 0
    return g >= 3


def _run_prefill(q, pool, layer, bt, q_lens, ctx, n_q, n_kv, variant, hd=128):
    cu = torch.tensor([0] + list(torch.tensor(q_lens).cumsum(0)), dtype=torch.int32, device=DEV)
    ctx_t = torch.tensor(ctx, dtype=torch.int32, device=DEV)
    out = torch.zeros_like(q)
    scale = 1.0 / math.sqrt(hd)
    prev = ops.set_prefill_variant(variant)
    try:
        rows = ops.prefill_tile_rows(q_lens, n_q, n_kv)
        ts, tq = ops.build_prefill_tiles(q_lens, n_q, n_kv, rows)
        ts = torch.tensor(ts, dtype=torch.int32, device=DEV)
        tq = torch.tensor(tq, dtype=torch.int32, device=DEV)
        ops.paged_prefill(q, out, pool, layer, bt, cu, ctx_t, ts, tq, n_q, n_kv, hd, scale, rows)
    finally:
        ops.set_prefill_variant(prev)
    torch.cuda.synchronize()
    return out, cu, ctx_t, rows


@pytest.mark.parametrize("variant", [-1, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11])
@pytest.mark.parametrize("n_q,n_kv", [(32, 8), (8, 8), (8, 4), (16, 2), (28, 4), (10, 2), (12, 2), (12, 4), (24, 8)])
def test_paged_prefill_matches_reference(n_q, n_kv, variant):
    if not _prefill_variant_ok(variant, n_q, n_kv):
        pytest.skip("this kernel generation does not take this GQA ratio")
    torch.manual_seed(4)
    hd, layers = 128, 2
    # (new rows, cached prefix) per sequence: ragged tails, prefix hits, a 1-row chunk
    cases = [(200, 0), (33, 500), (1, 129), (64, 64), (97, 1000), (130, 3)]
    q_lens = [c[0] for c in cases]
    ctx = [c[0] + c[1] for c in cases]
    max_pages = (max(ctx) + ops.PAGE - 1) // ops.PAGE
    pool = _v_offset_pool(len(cases) * max_pages + 3, layers, n_kv, seed=5)
    bt = _block_tables(len(cases), max_pages, pool.shape[0], 6)
    q = torch.randn(sum(q_lens), n_q * hd, device=DEV).to(BF)
    out, cu, ctx_t, rows = _run_prefill(q, pool, 1, bt, q_lens, ctx, n_q, n_kv, variant)
    if variant >= 4 or (variant < 0 and n_q // n_kv >= 3):
        assert rows == 64
    out_ref = torch.zeros_like(q)
    ref.paged_prefill(q, out_ref, pool, 1, bt, cu, ctx_t, n_q, n_kv, hd, 1.0 / math.sqrt(hd))
    err = _rel_row_err(out, out_ref, n_q)
    assert err < REL_TOL, err


@pytest.mark.parametrize("variant", [-1, 8, 9, 10, 11])
@pytest.mark.parametrize("rope", [False, True])
@pytest.mark.parametrize("n_q,n_kv", [(32, 8), (28, 4), (12, 4), (10, 2)])
def test_paged_prefill_key_range_splits_match_reference(variant, rope, n_q, n_kv):
    """A short work list in front of long contexts is cut into key ranges that run as workgroups
    of their own and are merged by a second launch: a follow-up turn of 8 rows behind 21k cached
    tokens, a 100-row chunk (two tiles, the first one whole) behind 5k, and a short sequence that
    is not split, all in one launch.  Same result as the unsplit reference."""
    torch.manual_seed(50)
    hd = 128
    cases = [(8, 20992), (100, 5000), (40, 100)]
    q_lens = [c[0] for c in cases]
    ctx = [c[0] + c[1] for c in cases]
    max_pages = (max(ctx) + ops.PAGE - 1) // ops.PAGE
    pool = _v_offset_pool(sum((c + ops.PAGE - 1) // ops.PAGE for c in ctx) + 5, 1, n_kv, seed=51)
    perm = torch.randperm(pool.shape[0] - 1, device=DEV)[: len(cases) * max_pages].to(torch.int32) + 1
    bt = torch.zeros((len(cases), max_pages), dtype=torch.int32, device=DEV)
    used = 0
    for i, c in enumerate(ctx):
        n = (c + ops.PAGE - 1) // ops.PAGE
        bt[i, :n] = perm[used:used + n]
        used += n
    q = torch.randn(sum(q_lens), n_q * hd, device=DEV).to(BF)
    cu = torch.tensor([0] + list(torch.tensor(q_lens).cumsum(0)), dtype=torch.int32, device=DEV)
    ctx_t = torch.tensor(ctx, dtype=torch.int32, device=DEV)
    scale = 1.0 / math.sqrt(hd)
    cos_sin = ops.build_cos_sin(hd, max(ctx) + 1, 500000.0, device=DEV) if rope else None
    out = torch.zeros_like(q)
    prev = ops.set_prefill_variant(variant)
    try:
        rows = ops.prefill_tile_rows(q_lens, n_q, n_kv)
        assert rows == 64
        ts, tq = ops.build_prefill_tiles(q_lens, n_q, n_kv, rows, ctx)
        cut = ops.plan_prefill_splits(ts, tq, q_lens, ctx, n_q, n_kv, rows)
        assert cut is not None
        ts, tq, split, groups, n_rec = cut
        assert n_rec > 30 and len(groups) == 4 * 3  # both tiles of the 100-row chunk and the 8-row turn are cut
        assert -1 in split[2::3], "the short sequence keeps its single entry"
        ws = torch.full((n_rec * ops.prefill_split_record_bytes(n_q, n_kv) // 4,), float("nan"), device=DEV)
        i32 = lambda v: torch.tensor(v, dtype=torch.int32, device=DEV)  # noqa: E731
        ops.paged_prefill(q, out, pool, 0, bt, cu, ctx_t, i32(ts), i32(tq), n_q, n_kv, hd, scale, rows,
                          rope=cos_sin, split=i32(split), split_ws=ws, groups=i32(groups))
    finally:
        ops.set_prefill_variant(prev)
    torch.cuda.synchronize()
    q_ref = q
    if rope:  # the kernel rotated its query rows itself: rotate them for the reference
        q_ref = q.clone()
        pos = torch.cat([torch.arange(c - n, c, device=DEV, dtype=torch.int32) for n, c in zip(q_lens, ctx)])
        ops.rope_and_cache(q_ref, pos, None, cos_sin, pool, 0, n_q, 0, hd, True)  # queries only, in place
    out_ref = torch.zeros_like(q)
    ref.paged_prefill(q_ref, out_ref, pool, 0, bt, cu, ctx_t, n_q, n_kv, hd, scale)
    assert _rel_row_err(out, out_ref, n_q) < REL_TOL


@pytest.mark.parametrize("variant", [-1, 0, 2, 4, 8, 10])
def test_paged_prefill_skips_padding_of_a_fixed_size_work_list(variant):
    """A captured graph launches the kernel with a work list, a sequence table and a row count of
    fixed size: entries past the real ones are -1 (their workgroups leave at once), sequences past
    the real ones are empty, rows past the real ones belong to nobody and must stay untouched."""
    n_q, n_kv, hd = 32, 8, 128
    if not _prefill_variant_ok(variant, n_q, n_kv):
        pytest.skip("the 64-row kernel takes GQA ratios that are multiples of 4")
    torch.manual_seed(40)
    cases = [(70, 300), (5, 64), (33, 0)]
    q_lens = [c[0] for c in cases]
    ctx = [c[0] + c[1] for c in cases]
    S, rows_total = 8, 256
    max_pages = 16
    pool = _v_offset_pool(len(cases) * max_pages + 3, 1, n_kv, seed=41)
    bt = torch.zeros((S, max_pages), dtype=torch.int32, device=DEV)
    bt[:len(cases)] = _block_tables(len(cases), max_pages, pool.shape[0], 42)
    q = torch.randn(rows_total, n_q * hd, device=DEV).to(BF)
    real = sum(q_lens)
    cu = torch.tensor([0, 70, 75, 108] + [108] * (S - 3), dtype=torch.int32, device=DEV)
    ctx_t = torch.tensor(ctx + [0] * (S - 3), dtype=torch.int32, device=DEV)
    out = torch.full_like(q, 7.0)
    prev = ops.set_prefill_variant(variant)
    try:
        rows = ops.prefill_tile_rows(q_lens, n_q, n_kv)
        ts, tq = ops.build_prefill_tiles(q_lens, n_q, n_kv, rows, ctx)
        pad = rows_total // rows + S - len(ts)
        # padding in the middle as well as at the end: position in the list must not matter
        ts = torch.tensor(ts[:1] + [-1] * 2 + ts[1:] + [-1] * (pad - 2), dtype=torch.int32, device=DEV)
        tq = torch.tensor(tq[:1] + [0] * 2 + tq[1:] + [0] * (pad - 2), dtype=torch.int32, device=DEV)
        ops.paged_prefill(q, out, pool, 0, bt, cu, ctx_t, ts, tq, n_q, n_kv, hd, 1.0 / math.sqrt(hd), rows)
    finally:
        ops.set_prefill_variant(prev)
    torch.cuda.synchronize()
    out_ref = torch.zeros_like(q[:real])
    ref.paged_prefill(q[:real], out_ref, pool, 0, bt[:3], cu[:4], ctx_t[:3], n_q, n_kv, hd, 1.0 / math.sqrt(hd))
    assert _rel_row_err(out[:real], out_ref, n_q) < REL_TOL
    assert (out[real:] == 7.0).all(), "rows that belong to no sequence were written"


@pytest.mark.parametrize("variant", [-1, 8, 9, 10, 11])
@pytest.mark.parametrize("n_q,n_kv", [(32, 8), (8, 2)])
def test_paged_prefill_rotates_queries_itself(n_q, n_kv, variant):
    """Prefill rows: K / V go to the cache through the tile kernel (``kv_cache_write``: rotated K,
    LDS-transposed V, nothing written back to qkv) and the fourth-generation attention kernel
    rotates the query rows as it loads them.  Same output as rotating in place first, same cache
    contents; decode rows in front keep the per-token pass."""
    torch.manual_seed(9)
    hd, layers = 128, 2
    cases = [(200, 0), (33, 500), (1, 129), (64, 64), (97, 1000)]
    q_lens = [c[0] for c in cases]
    ctx = [c[0] + c[1] for c in cases]
    nd = 3  # decode rows in front of the prefill rows, as in a mixed step
    t = nd + sum(q_lens)
    max_pages = (max(ctx) + ops.PAGE - 1) // ops.PAGE
    pool = _v_offset_pool(len(cases) * max_pages + 3, layers, n_kv, seed=5)
    bt = _block_tables(len(cases), max_pages, pool.shape[0], 6)
    qkv = torch.randn(t, (n_q + 2 * n_kv) * hd, device=DEV).to(BF)
    cs = ops.build_cos_sin(hd, 2048, 500000.0, device=DEV)
    pos = [7, 70, 700]
    slots = [-1, -1, -1]
    for i, (n, pre) in enumerate(cases):
        for r in range(n):
            p_ = pre + r
            pos.append(p_)
            slots.append(int(bt[i, p_ // ops.PAGE]) * ops.PAGE + p_ % ops.PAGE)
    pos = torch.tensor(pos, dtype=torch.int32, device=DEV)
    slots = torch.tensor(slots, dtype=torch.int64, device=DEV)
    prev = ops.set_prefill_variant(variant)
    try:
        rows = ops.prefill_tile_rows(q_lens, n_q, n_kv)
        assert rows == 64 and ops.prefill_fuses_rope(n_q, n_kv, rows)
        ops.set_prefill_variant(5)
        assert not ops.prefill_fuses_rope(n_q, n_kv, rows)
        ops.set_prefill_variant(variant)
        cu = torch.tensor([0] + list(torch.tensor(q_lens).cumsum(0)), dtype=torch.int32, device=DEV)
        ctx_t = torch.tensor(ctx, dtype=torch.int32, device=DEV)
        ts, tq = ops.build_prefill_tiles(q_lens, n_q, n_kv, rows)
        ts = torch.tensor(ts, dtype=torch.int32, device=DEV)
        tq = torch.tensor(tq, dtype=torch.int32, device=DEV)
        scale = 1.0 / math.sqrt(hd)
        # unfused: rotate everything in place, then attend
        a, pool_a = qkv.clone(), pool.clone()
        ops.rope_and_cache(a, pos, slots, cs, pool_a, 1, n_q, n_kv, hd)
        out_a = torch.zeros(t - nd, n_q * hd, device=DEV, dtype=BF)
        ops.paged_prefill(a[nd:, : n_q * hd], out_a, pool_a, 1, bt, cu, ctx_t, ts, tq, n_q, n_kv, hd, scale, rows)
        # fused
        b, pool_b = qkv.clone(), pool.clone()
        ops.rope_and_cache(b[:nd], pos[:nd], slots[:nd], cs, pool_b, 1, n_q, n_kv, hd)
        ops.kv_cache_write(b[nd:], pos[nd:], slots[nd:], cs, pool_b, 1, n_q, n_kv, hd)
        out_b = torch.zeros_like(out_a)
        ops.paged_prefill(b[nd:, : n_q * hd], out_b, pool_b, 1, bt, cu, ctx_t, ts, tq, n_q, n_kv, hd, scale, rows,
                          rope=cs)
    finally:
        ops.set_prefill_variant(prev)
    torch.cuda.synchronize()
    assert torch.equal(pool_a[:, 0], pool_b[:, 0]), "other layers untouched"
    torch.testing.assert_close(pool_a.float(), pool_b.float(), atol=2e-2, rtol=2e-2)
    assert (pool_a != pool_b).float().mean().item() < 0.02, "at most rounding-level differences in K"
    assert torch.equal(a[:nd], b[:nd]), "decode rows are rotated in place either way"
    assert torch.equal(b[nd:], qkv[nd:]), "prefill rows of qkv stay as they were"
    err = _rel_row_err(out_b, out_a, n_q)
    assert err < 8e-3, err  # the two rotations may round a query element to neighbouring bf16 values


@pytest.mark.parametrize("variant", [2, 4, 8, 10])
def test_paged_prefill_forced_rescale(variant):
    """A key far larger than the rest late in the context forces the online-softmax
    rescale branch on a chosen tile (in the second half of the tile for the 64-row kernel)."""
    torch.manual_seed(7)
    n_q, n_kv, hd = 32, 8, 128
    ql, prefix = 64, 448
    ctx = ql + prefix
    max_pages = ctx // ops.PAGE
    pool = _v_offset_pool(max_pages + 1, 1, n_kv, seed=8)
    bt = _block_tables(1, max_pages, pool.shape[0], 9)
    q = torch.randn(ql, n_q * hd, device=DEV).to(BF)
    kview, _ = ref._kv_views(pool, 0, n_kv, hd)  # [pages, n_kv, PAGE, channel blocks, 32]
    # spikes: key 300 aligned with query row 10, key 360 (second half of its page) with row 40
    for key, row, gain in ((300, 10, 4.0), (360, 40, 6.0)):
        page, off = int(bt[0, key // ops.PAGE]), key % ops.PAGE
        kview[page, :, off] = (q[row].view(n_kv, 4, hd)[:, 0, :].float() * gain).to(BF).view(n_kv, hd // 32, 32)
    out, cu, ctx_t, _ = _run_prefill(q, pool, 0, bt, [ql], [ctx], n_q, n_kv, variant)
    out_ref = torch.zeros_like(q)
    ref.paged_prefill(q, out_ref, pool, 0, bt, cu, ctx_t, n_q, n_kv, hd, 1.0 / math.sqrt(hd))
    err = _rel_row_err(out, out_ref, n_q)
    assert err < REL_TOL, err


def _ref_prefill_long(q, pool, layer, bt_row, ql, ctx, n_q, n_kv, hd):
    """fp32 definition, one KV head at a time (a 4.6k x 21k score matrix per query head)."""
    g = n_q // n_kv
    k, v = ref._gather_kv(pool, layer, n_kv, hd, bt_row, ctx)
    out = torch.empty((ql, n_q * hd), dtype=torch.float32, device=DEV)
    qpos = torch.arange(ctx - ql, ctx, device=DEV)[:, None]
    kpos = torch.arange(ctx, device=DEV)[None, :]
    dead = kpos > qpos
    qs = q.view(ql, n_kv, g, hd).float()
    for h in range(n_kv):
        sc = torch.einsum("qgd,kd->gqk", qs[:, h], k[h]) / math.sqrt(hd)
        sc = sc.masked_fill(dead, float("-inf"))
        o = torch.einsum("gqk,kd->qgd", torch.softmax(sc, dim=-1), v[h])
        out.view(ql, n_kv, g, hd)[:, h] = o
    return out


@pytest.mark.parametrize("variant", [2, 3, 4, 5, 8, 9, 10, 11])
@pytest.mark.parametrize("ql,prefix", [(4616, 16384), (1024, 15360)])
def test_paged_prefill_long_context(ql, prefix, variant):
    """The last chunk of a 21k-token prompt (4616 rows behind 16384 cached keys, 329 pages deep in
    a shuffled pool) and a 1k chunk behind 15k: page-table indexing far into a long sequence."""
    torch.manual_seed(21)
    n_q, n_kv, hd = 32, 8, 128
    ctx = ql + prefix
    pages = (ctx + ops.PAGE - 1) // ops.PAGE
    pool = _v_offset_pool(pages + 40, 1, n_kv, seed=22)
    bt = _block_tables(1, pages, pool.shape[0], 23)
    q = torch.randn(ql, n_q * hd, device=DEV).to(BF)
    out, _, _, _ = _run_prefill(q, pool, 0, bt, [ql], [ctx], n_q, n_kv, variant)
    want = _ref_prefill_long(q, pool, 0, bt[0], ql, ctx, n_q, n_kv, hd)
    err = _rel_row_err(out, want, n_q)
    assert err < REL_TOL, err


@pytest.mark.parametrize("n_q,n_kv", [(32, 8), (8, 8), (64, 8), (28, 4), (10, 2)])
@pytest.mark.parametrize("splits", [1, 3, 8])
@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4, 5, -1])
def test_paged_decode_matches_reference(n_q, n_kv, splits, variant):
    torch.manual_seed(10)
    hd, layers = 128, 2
    ctx = [1, 31, 32, 33, 64, 65, 700, 2049]
    max_pages = (max(ctx) + ops.PAGE - 1) // ops.PAGE
    pool = _v_offset_pool(len(ctx) * max_pages + 2, layers, n_kv, seed=11)
    bt = _block_tables(len(ctx), max_pages, pool.shape[0], 12)
    q = torch.randn(len(ctx), n_q * hd, device=DEV).to(BF)
    ctx_t = torch.tensor(ctx, dtype=torch.int32, device=DEV)
    out, out_ref = torch.zeros_like(q), torch.zeros_like(q)
    scale = 1.0 / math.sqrt(hd)
    prev = ops.set_decode_variant(variant)
    try:
        ops.paged_decode(q, out, pool, 1, bt, ctx_t, n_q, n_kv, hd, scale, num_splits=splits)
    finally:
        ops.set_decode_variant(prev)
    ref.paged_decode(q, out_ref, pool, 1, bt, ctx_t, n_q, n_kv, hd, scale)
    err = _rel_row_err(out, out_ref, n_q)
    assert err < REL_TOL, err


@pytest.mark.parametrize("variant", [0, 2, 4, -1])
def test_paged_decode_bench_shape(variant):
    """The decode step of the headline bench: 40 sequences of ~22k keys each, their 345 pages
    scattered over a 14 000-page pool, the automatic split count."""
    torch.manual_seed(31)
    n_q, n_kv, hd, n_seqs = 32, 8, 128, 40
    ctx = [22000 - 37 * i for i in range(n_seqs)]
    max_pages = (max(ctx) + ops.PAGE - 1) // ops.PAGE
    pool = _v_offset_pool(14000, 1, n_kv, seed=32)
    g = torch.Generator().manual_seed(33)
    bt = torch.randperm(14000, generator=g)[: n_seqs * max_pages].view(n_seqs, max_pages).to(torch.int32).to(DEV)
    q = torch.randn(n_seqs, n_q * hd, device=DEV).to(BF)
    ctx_t = torch.tensor(ctx, dtype=torch.int32, device=DEV)
    out, out_ref = torch.zeros_like(q), torch.zeros_like(q)
    scale = 1.0 / math.sqrt(hd)
    splits = ops.decode_num_splits(n_seqs, n_kv, max(ctx))
    prev = ops.set_decode_variant(variant)
    try:
        ops.paged_decode(q, out, pool, 0, bt, ctx_t, n_q, n_kv, hd, scale, num_splits=splits)
    finally:
        ops.set_decode_variant(prev)
    ref.paged_decode(q, out_ref, pool, 0, bt, ctx_t, n_q, n_kv, hd, scale)
    err = _rel_row_err(out, out_ref, n_q)
    assert err < REL_TOL, err


def test_silu_and_mul():
    torch.manual_seed(13)
    x = torch.randn(77, 2 * 14336, device=DEV).to(BF)
    torch.testing.assert_close(ops.silu_and_mul(x).float(), ref.silu_and_mul(x).float(),
                               atol=2e-2, rtol=2e-2)


@pytest.mark.parametrize("m", [1, 7, 16, 17, 40, 64])
@pytest.mark.parametrize("n,k,bias", [(6144, 4096, False), (4096, 14336, False), (1536, 1024, True),
                                      (4096, 512, True)])
@pytest.mark.parametrize("variant", [0, 1, 2, 3])
def test_skinny_linear_matches_fp32(m, n, k, bias, variant):
    """Decode-batch projection kernel against the fp32 product of the same bf16 operands."""
    torch.manual_seed(m * 131 + n)
    x = torch.randn(m, k, device=DEV).to(BF)
    w = (torch.randn(n, k, device=DEV) * 0.03).to(BF)
    b = torch.randn(n, device=DEV).to(BF) if bias else None
    prev = ops.set_gemm_variant(variant)
    try:
        got = ops._skinny(x, w, b, False)
    finally:
        ops.set_gemm_variant(prev)
    assert got is not None, "shape was expected to be tiled by the kernel"
    want = torch.nn.functional.linear(x.float(), w.float(), None if b is None else b.float())
    err = (got.float() - want).abs().max().item() / want.abs().max().item()
    assert err < 1e-2, err


@pytest.mark.parametrize("m", [1, 7, 16, 17, 40, 48, 64])
@pytest.mark.parametrize("n,k,bias,pair", [(6144, 4096, False, False), (4096, 14336, False, False),
                                           (1536, 1024, True, False), (128, 256, True, False),
                                           (2 * 14336, 4096, False, True), (2 * 64, 512, False, True),
                                           (128256, 4096, False, False)])
@pytest.mark.parametrize("variant", [0, 4])
def test_stream_gemm_matches_fp32(m, n, k, bias, pair, variant):
    """LDS-staged streaming projection (split reduction + fixed-order combine) against fp32."""
    torch.manual_seed(m * 7 + n)
    x = torch.randn(m, k, device=DEV).to(BF)
    w = (torch.randn(n, k, device=DEV) * 0.03).to(BF)
    b = torch.randn(n, device=DEV).to(BF) if bias else None
    prev = ops.set_gemm_variant(variant)
    try:
        got = ops._stream_gemm(x, w, b, pair)
        again = ops._stream_gemm(x, w, b, pair)
    finally:
        ops.set_gemm_variant(prev)
    assert got is not None, "shape was expected to be tiled by the kernel"
    want = torch.nn.functional.linear(x.float(), w.float(), None if b is None else b.float())
    if pair:
        want = torch.nn.functional.silu(want[:, : n // 2]) * want[:, n // 2:]
    assert got.shape == want.shape
    err = (got.float() - want).abs().max().item() / want.abs().max().item()
    assert err < 1e-2, err
    assert torch.equal(got, again), "the split reduction must be reproducible"


@pytest.mark.parametrize("m", [1, 7, 16, 17, 40, 48, 64])
@pytest.mark.parametrize("n,k,bias,pair", [(6144, 4096, False, False), (4096, 14336, False, False),
                                           (1536, 1024, True, False), (128, 256, True, False),
                                           (2 * 14336, 4096, False, True), (2 * 64, 512, False, True),
                                           (64, 384, False, False)])
def test_packed_gemm_matches_fp32(m, n, k, bias, pair):
    """Projection over weights pre-packed in matrix-core tile order (register-direct weight
    stream, split reduction + fixed-order combine) against fp32; a row's bits do not depend on
    what else is in the batch."""
    torch.manual_seed(m * 11 + n)
    x = torch.randn(m, k, device=DEV).to(BF)
    w = (torch.randn(n, k, device=DEV) * 0.03).to(BF)
    b = torch.randn(n, device=DEV).to(BF) if bias else None
    wp = ops.pack_weight(w)
    assert ops.packed_supported(n, k, pair)
    got = ops.packed_linear(x, wp, n, k, b, pair)
    again = ops.packed_linear(x, wp, n, k, b, pair)
    assert got is not None
    want = torch.nn.functional.linear(x.float(), w.float(), None if b is None else b.float())
    if pair:
        want = torch.nn.functional.silu(want[:, : n // 2]) * want[:, n // 2:]
    assert got.shape == want.shape
    err = (got.float() - want).abs().max().item() / want.abs().max().item()
    assert err < 1e-2, err
    assert torch.equal(got, again), "the split reduction must be reproducible"
    alone = ops.packed_linear(x[:1].contiguous(), wp, n, k, b, pair)
    assert torch.equal(alone[0], got[0]), "row 0 alone and inside the batch"
    # the public entry points take the packed copy
    via = ops.gate_up_silu(x, w, wp) if pair else ops.linear(x, w, b, wp)
    assert torch.equal(via, got)


@pytest.mark.parametrize("b,heads,t,pos", [(1, 12, 1500, None), (3, 12, 448, 100), (5, 6, 448, 0),
                                           (32, 20, 1500, None), (2, 16, 448, 447), (4, 8, 300, 256)])
def test_dense_decode_attention_matches_fp32(b, heads, t, pos):
    """Single-query attention over dense K/V (speech decoder: self-attention up to a position
    read from device memory, cross-attention over all encoder positions) against fp32; the
    query is a strided slice of a fused q/k/v row, V has a per-channel offset."""
    torch.manual_seed(b * 7 + t)
    d = heads * 64
    qkv = torch.randn(b, 3 * d, device=DEV).to(BF)
    q = qkv[:, :d]
    k = torch.randn(b, t, d, device=DEV).to(BF)
    v = (torch.randn(b, t, d, device=DEV) + torch.linspace(-2, 2, d, device=DEV)).to(BF)
    pos_t = None if pos is None else torch.tensor([pos], dtype=torch.long, device=DEV)
    assert ops.dense_decode_attention_supported(q, k, v, heads)
    # poison what must not be read
    n = t if pos is None else pos + 1
    k[:, n:] = float("nan")
    v[:, n:] = float("nan")
    got = ops.dense_decode_attention(q, k, v, heads, pos_t).float()
    qh = q.float().view(b, heads, 1, 64)
    kh = k[:, :n].float().view(b, n, heads, 64).transpose(1, 2)
    vh = v[:, :n].float().view(b, n, heads, 64).transpose(1, 2)
    want = (torch.softmax(qh @ kh.transpose(-1, -2) * 0.125, dim=-1) @ vh).transpose(1, 2).reshape(b, d)
    err = ((got - want).abs().amax(dim=1) / want.abs().amax(dim=1)).max().item()
    assert err < 1e-2, err
    assert not ops.dense_decode_attention_supported(q.float(), k.float(), v.float(), heads)
    assert not ops.dense_decode_attention_supported(q, k, v, heads + 1)


@pytest.mark.parametrize("b,heads,tq,tk", [(2, 12, 1500, 1500), (1, 6, 100, 1500), (3, 20, 257, 333),
                                           (2, 8, 4, 1500), (1, 12, 128, 64), (2, 16, 129, 1)])
def test_dense_attention_matches_fp32(b, heads, tq, tk):
    """Bidirectional MFMA flash attention over dense Q/K/V (speech encoder shapes, ragged tails
    on both sides, strided views of a fused projection) against fp32; V has a per-channel offset
    and every key a different weight, so a misplaced key or channel shows."""
    torch.manual_seed(b * 13 + tq + tk)
    d = heads * 64
    qkv = torch.randn(b, tq, 3 * d, device=DEV).to(BF)
    q = qkv[:, :, d:2 * d]
    kv = torch.randn(b, tk, 2 * d, device=DEV)
    kv[:, :, d:] += torch.linspace(-2, 2, d, device=DEV)
    kv = kv.to(BF)
    k, v = kv[:, :, :d], kv[:, :, d:]
    assert ops.dense_attention_supported(q, k, v, heads)
    got = ops.dense_attention(q, k, v, heads).float()
    qh = q.float().view(b, tq, heads, 64).transpose(1, 2)
    kh = k.float().view(b, tk, heads, 64).transpose(1, 2)
    vh = v.float().view(b, tk, heads, 64).transpose(1, 2)
    want = (torch.softmax(qh @ kh.transpose(-1, -2) * 0.125, dim=-1) @ vh).transpose(1, 2).reshape(b, tq, d)
    err = ((got - want).abs().amax(dim=-1) / want.abs().amax(dim=-1)).max().item()
    assert err < 1.5e-2, err
    assert not ops.dense_attention_supported(q.float(), k.float(), v.float(), heads)


def test_qkv_projection_split_for_prefill_sized_batches():
    """From 4096 rows on the fused q/k/v projection is two library GEMMs writing into one buffer
    (the library is slow at the fused width): same result as the single call, one contiguous
    [rows, q + k + v] output either way."""
    torch.manual_seed(21)
    n_q_cols, n = 1024, 1536
    w = (torch.randn(n, 512, device=DEV) * 0.05).to(BF)
    for rows in (ops.SPLIT_QKV_MIN_ROWS, 64):
        x = torch.randn(rows, 512, device=DEV).to(BF)
        got = ops.qkv_linear(x, w, None, n_q_cols)
        want = torch.nn.functional.linear(x.float(), w.float())
        assert got.shape == (rows, n) and got.is_contiguous()
        err = (got.float() - want).abs().max().item() / want.abs().max().item()
        assert err < 1e-2, err
    b = torch.randn(n, device=DEV).to(BF)
    x = torch.randn(ops.SPLIT_QKV_MIN_ROWS, 512, device=DEV).to(BF)
    assert torch.equal(ops.qkv_linear(x, w, b, n_q_cols), ops.linear(x, w, b)), "a bias keeps the single call"


def test_skinny_linear_strided_rows_and_fallback():
    torch.manual_seed(3)
    base = torch.randn(9, 2 * 4096, device=DEV).to(BF)
    x = base[:, :4096]  # row stride 8192
    w = (torch.randn(2048, 4096, device=DEV) * 0.03).to(BF)
    want = torch.nn.functional.linear(x.float(), w.float())
    got = ops._skinny(x, w, None, False)
    assert (got.float() - want).abs().max().item() / want.abs().max().item() < 1e-2
    # a reduction length the kernel does not tile takes the library path, same answer
    w2 = (torch.randn(2048, 768, device=DEV) * 0.03).to(BF)
    x2 = torch.randn(9, 768, device=DEV).to(BF)
    assert ops._skinny(x2, w2, None, False) is None
    want2 = torch.nn.functional.linear(x2.float(), w2.float())
    assert (ops.linear(x2, w2).float() - want2).abs().max().item() / want2.abs().max().item() < 1e-2


@pytest.mark.parametrize("m", [1, 16, 33, 64])
@pytest.mark.parametrize("variant", [0, 1])
def test_skinny_gate_up_silu_matches_fp32(m, variant):
    torch.manual_seed(m)
    inter, k = 14336, 4096
    x = torch.randn(m, k, device=DEV).to(BF)
    w = (torch.randn(2 * inter, k, device=DEV) * 0.02).to(BF)
    prev = ops.set_gemm_variant(variant)
    try:
        got = ops._skinny(x, w, None, True)
    finally:
        ops.set_gemm_variant(prev)
    assert got is not None and got.shape == (m, inter)
    gu = torch.nn.functional.linear(x.float(), w.float())
    want = torch.nn.functional.silu(gu[:, :inter]) * gu[:, inter:]
    err = (got.float() - want).abs().max().item() / want.abs().max().item()
    assert err < 1e-2, err


def test_argmax():
    torch.manual_seed(14)
    logits = torch.randn(19, 128256, device=DEV).to(BF)
    logits[3, 77] = 100.0
    logits[4, 128255] = 100.0
    got = ops.argmax(logits)
    # ties in bf16 are possible: compare values at the returned index
    vals = logits.float().gather(1, got[:, None]).squeeze(1)
    assert torch.equal(vals, logits.float().max(dim=1).values)
    assert got[3].item() == 77 and got[4].item() == 128255


def test_kv_gather_scatter_roundtrip():
    pool = _pool(12, 4, 8, seed=15)
    ids = torch.tensor([7, 2, 9], dtype=torch.int64, device=DEV)
    staging = torch.empty((3, pool[0].numel()), dtype=BF, device=DEV)
    ops.kv_gather_pages(pool, ids, staging)
    assert torch.equal(staging.view(3, *pool.shape[1:]), pool[ids])
    pool2 = torch.zeros_like(pool)
    ops.kv_scatter_pages(staging, ids, pool2)
    assert torch.equal(pool2[ids], pool[ids])
    assert pool2[0].abs().sum().item() == 0


def test_cachegen_codec():
    n_layers, n_kv = 4, 8
    pool = _pool(2, n_layers, n_kv, seed=16)
    staging = pool.reshape(-1).contiguous()
    unit = n_kv * ops.PAGE * 128
    levels = torch.tensor([127, 127, 31, 15], dtype=torch.int32, device=DEV)
    sym, scales = ops.cachegen_encode(staging, levels, unit, n_layers)
    sym_ref, scales_ref = ref.cachegen_encode(staging, levels.tolist(), unit, n_layers)
    assert (sym.int() - sym_ref.int()).abs().max().item() <= 1
    torch.testing.assert_close(scales, scales_ref, rtol=1e-5, atol=1e-7)
    dec = ops.cachegen_decode(sym, scales, unit)
    dec_ref = ref.cachegen_decode(sym, scales, unit)
    torch.testing.assert_close(dec.float(), dec_ref.float(), atol=2e-2, rtol=2e-2)
    # 8-bit layers reconstruct closely
    x = staging.float().view(2, n_layers, -1)
    y = dec.float().view(2, n_layers, -1)
    rel = (x[:, 0] - y[:, 0]).abs().mean() / x[:, 0].abs().mean()
    assert rel < 0.03, rel


@pytest.mark.parametrize("n_ctx,spc,per_stream", [(4, 256, 64), (32, 512, 256), (2, 512, 1000)])
def test_rans_gpu_matches_numpy_definition_bit_for_bit(n_ctx, spc, per_stream):
    import numpy as np

    g = torch.Generator().manual_seed(n_ctx * 1000 + spc)
    ctx_len = spc * per_stream
    sym = torch.clamp(torch.round(torch.randn(n_ctx * ctx_len, generator=g) * 7), -128, 127).to(torch.int8)
    sym[ctx_len:2 * ctx_len] = 11                                              # single-symbol context
    sym[:ctx_len // 2] = torch.randint(-128, 128, (ctx_len // 2,), generator=g, dtype=torch.int8)  # flat half
    f_ref, st_ref, sz_ref, co_ref = ref.rans_encode(sym.numpy(), n_ctx, spc)
    freq, states, sizes, coded = ops.rans_encode(sym.to(DEV), n_ctx, spc)
    assert np.array_equal(freq.cpu().numpy().view(np.uint16).reshape(n_ctx, 256), f_ref)
    assert np.array_equal(sizes.cpu().numpy().view(np.uint16), sz_ref)
    assert np.array_equal(states.cpu().numpy().view(np.uint32), st_ref)
    assert np.array_equal(coded.cpu().numpy(), co_ref)
    back = ops.rans_decode(freq, states, sizes, coded, sym.numel(), n_ctx, spc)
    assert torch.equal(back.cpu(), sym)
    # damaged input: wrong bytes decode to garbage of the right shape, never out of bounds
    junk = ops.rans_decode(freq, states, sizes, coded[: coded.numel() // 2].contiguous(), sym.numel(), n_ctx, spc)
    assert junk.shape == sym.shape


def test_cachegen_page_on_gpu_uses_the_rans_stage_and_matches_the_host_coder(monkeypatch):
    from vllm_production_stack_amd.kv.serde import SERDE_CACHEGEN_RANS, PageSerde, _HDR

    n_layers, n_kv, hd = 32, 8, 128
    torch.manual_seed(9)
    page = (torch.randn(n_layers * 2 * n_kv * 64 * hd, device=DEV) * 0.7).to(BF)
    cg = PageSerde(n_layers, n_kv, hd, BF, "cachegen")
    blob = cg.encode(page)
    assert _HDR.unpack_from(blob, 0)[2] == SERDE_CACHEGEN_RANS
    back = cg.decode(blob, torch.device(DEV))
    monkeypatch.setenv("VPSA_CACHEGEN_HOST_CODER", "1")
    blob_host = cg.encode(page)
    assert _HDR.unpack_from(blob_host, 0)[2] == 1
    assert torch.equal(cg.decode(blob_host, torch.device(DEV)), back)      # same quantiser, lossless stages
    assert len(blob) < 1.08 * len(blob_host)                               # static tables cost a few per cent
    assert len(blob) < 0.5 * page.numel() * 2
    monkeypatch.delenv("VPSA_CACHEGEN_HOST_CODER")
    # a CPU host reads it too (its dequantiser is the PyTorch definition: equal up to rounding)
    on_cpu = cg.decode(blob, torch.device("cpu")).to(DEV).float()
    assert (on_cpu - back.float()).abs().max().item() <= 0.02 * back.float().abs().max().item()


def _sampling_reference(row_bf16, temperature, top_p, top_k):
    """Float64 definition of the kept set: (threshold value, kept mass / largest probability, kept count)."""
    import numpy as np

    v = row_bf16.float().cpu().numpy().astype(np.float64)
    p = np.exp((v - v.max()) / max(temperature, 1e-5))
    thr = -np.inf
    if 0 < top_k < v.size:
        thr = np.sort(v)[-top_k]
    keep = v >= thr
    z = p[keep].sum()
    if 0 < top_p < 1:
        vals = np.unique(v[keep])[::-1]                        # distinct values, descending
        mass = np.array([p[v == x].sum() for x in vals]) if vals.size < 4096 else None
        if mass is None:                                       # large rows: group by value with bincount
            order = np.argsort(-v[keep], kind="stable")
            sv, sp = v[keep][order], p[keep][order]
            cuts = np.flatnonzero(np.diff(sv)) + 1
            mass = np.add.reduceat(sp, np.r_[0, cuts])
            vals = sv[np.r_[0, cuts]]
        cum = np.cumsum(mass)
        j = int(np.searchsorted(cum, top_p * z * (1 - 1e-12)))
        thr = vals[min(j, vals.size - 1)]
        keep = v >= thr
        z = p[keep].sum()
    return thr, z, int(keep.sum()), p, v


@pytest.mark.parametrize("vocab,stride", [(128256, 128256), (1000, 1024), (50, 64)])
def test_sampling_kernel_keeps_the_right_set(vocab, stride):
    import numpy as np

    torch.manual_seed(vocab)
    cases = [(1.0, 1.0, -1), (0.7, 0.9, -1), (1.3, 1.0, 40), (0.8, 0.95, 50), (0.5, 0.3, 5), (1.0, 0.999, 1),
             (2.0, 0.5, 0)]
    n = len(cases)
    buf = torch.zeros(n, stride, device=DEV, dtype=BF)
    buf[:, :vocab] = (torch.randn(n, vocab, device=DEV) * 2.5).to(BF)
    logits = buf[:, :vocab]
    params = torch.tensor([[t, p, float(k)] for t, p, k in cases], dtype=torch.float32, device=DEV)
    u = torch.rand(n, device=DEV)
    toks, stats = ops.sample(logits, params, u, want_stats=True)
    toks2 = ops.sample(logits, params, u)
    assert torch.equal(toks, toks2), "same inputs, same tokens"
    stats = stats.cpu().numpy()
    for i, (t, p, k) in enumerate(cases):
        thr, z, kept, prob, v = _sampling_reference(logits[i], t, p, k)
        key = int(stats[i, 0])
        b = (key & 0x7FFF) if key & 0x8000 else (~key & 0xFFFF)
        got_thr = np.frombuffer(np.uint32(b << 16).tobytes(), dtype=np.float32)[0]
        got_thr = -np.inf if key == 0 else got_thr
        tok = int(toks[i])
        assert 0 <= tok < vocab and v[tok] >= got_thr, "the drawn token lies in the kept set"
        if got_thr != thr:
            # fp32 sums may settle a borderline top-p cut one distinct value away: both sides of
            # the cut then carry (almost) exactly the target mass
            lo, hi = min(got_thr, thr), max(got_thr, thr)
            between = prob[(v >= lo) & (v < hi)].sum()
            assert between <= 2e-3 * z, (i, got_thr, thr)
        else:
            assert int(stats[i, 2]) == kept, (i, stats[i], kept)
            assert abs(stats[i, 1] - z) <= 2e-3 * z


def test_sampling_kernel_draws_from_the_filtered_distribution():
    import numpy as np

    torch.manual_seed(1)
    vocab, n = 48, 40000
    row = (torch.randn(1, 64, device=DEV) * 2).to(BF)
    logits = row[:, :vocab]
    for t, p, k in [(1.0, 1.0, -1), (0.8, 0.9, -1), (1.0, 1.0, 5)]:
        params = torch.tensor([[t, p, float(k)]] * n, dtype=torch.float32, device=DEV)
        rows = torch.zeros(n, dtype=torch.int32, device=DEV)
        toks = ops.sample(logits, params, torch.rand(n, device=DEV), rows).cpu().numpy()
        thr, z, kept, prob, v = _sampling_reference(logits[0], t, p, k)
        want = np.where(v >= thr, prob, 0.0) / z
        got = np.bincount(toks, minlength=vocab) / n
        assert got[want == 0].sum() == 0
        assert np.abs(got - want).max() < 4 * np.sqrt(want.max() / n) + 2e-3, np.abs(got - want).max()


def test_moe_routing():
    torch.manual_seed(17)
    t, e, k, hidden = 300, 8, 2, 4096
    logits = torch.randn(t, e, device=DEV).to(BF)
    w, ids = ops.moe_topk_softmax(logits, k)
    w_ref, ids_ref = ref.moe_topk_softmax(logits, k)
    torch.testing.assert_close(w, w_ref, atol=1e-4, rtol=1e-4)
    # bf16 logits can tie; the chosen experts must carry the same probabilities
    p = torch.softmax(logits.float(), -1)
    same = torch.allclose(p.gather(1, ids.long()), p.gather(1, ids_ref.long()), atol=1e-6)
    assert same
    assert (ids[:, 0] != ids[:, 1]).all()
    offsets, pos, sorted_tok = ops.moe_align(ids, e)
    counts = torch.bincount(ids.view(-1).long(), minlength=e)
    assert torch.equal(offsets[1:].long() - offsets[:-1].long(), counts)
    assert sorted(pos.tolist()) == list(range(t * k))
    x = torch.randn(t, hidden, device=DEV).to(BF)
    xs = ops.moe_permute(x, sorted_tok)
    assert torch.equal(xs, x[sorted_tok.long()])
    out = ops.moe_combine(xs, w, pos, t)
    exp = (x.float() * w.sum(1, keepdim=True)).to(BF)
    torch.testing.assert_close(out.float(), exp.float(), atol=3e-2, rtol=3e-2)


@pytest.mark.parametrize("t,hidden,inter", [(1, 4096, 14336), (5, 256, 264), (16, 512, 1024)])
def test_moe_small_batch_experts(t, hidden, inter):
    torch.manual_seed(18)
    e, k = 8, 2
    x = torch.randn(t, hidden, device=DEV).to(BF)
    w13 = (torch.randn(e, 2 * inter, hidden, device=DEV) * 0.03).to(BF)
    w2 = (torch.randn(e, hidden, inter, device=DEV) * 0.03).to(BF)
    logits = torch.randn(t, e, device=DEV).to(BF)
    w, ids = ops.moe_topk_softmax(logits, k)
    out = ops.moe_experts_small_batch(x, w13, w2, ids, w)
    exp = ref.moe_experts(x, w13, w2, ids, w)
    torch.testing.assert_close(out.float(), exp.float(), atol=3e-2, rtol=3e-2)


@pytest.mark.parametrize("t,hidden,inter,skew", [
    (1, 4096, 14336, False), (8, 512, 1024, False), (40, 512, 1024, False),
    (100, 512, 1024, True), (256, 256, 512, False), (33, 4096, 14336, False)])
def test_moe_grouped_mfma_experts(t, hidden, inter, skew):
    """Expert-grouped matrix-core path against the per-pair fp32 definition; `skew` routes every
    token to the same two experts so that one expert's rows exceed a single pass of token tiles."""
    torch.manual_seed(19)
    e, k = 8, 2
    x = torch.randn(t, hidden, device=DEV).to(BF)
    w13 = (torch.randn(e, 2 * inter, hidden, device=DEV) * 0.03).to(BF)
    w2 = (torch.randn(e, hidden, inter, device=DEV) * 0.03).to(BF)
    logits = torch.randn(t, e, device=DEV).to(BF)
    if skew:
        logits[:, 3] += 20.0
        logits[:, 6] += 10.0
    w, ids = ops.moe_topk_softmax(logits, k)
    out = ops.moe_experts_grouped(x, w13, w2, ids, w).float()
    # fp32 definition without intermediate rounding (the kernel keeps gate/up in fp32 through
    # SiLU; only the activation and the per-pair output are rounded to bf16)
    exp = torch.zeros((t, hidden), dtype=torch.float32, device=DEV)
    xf = x.float()
    for ex in range(e):
        rows, slot = (ids == ex).nonzero(as_tuple=True)
        if rows.numel() == 0:
            continue
        gu = xf[rows] @ w13[ex].float().t()
        a = torch.nn.functional.silu(gu[:, :inter]) * gu[:, inter:]
        exp.index_add_(0, rows, (a @ w2[ex].float().t()) * w[rows, slot].unsqueeze(1))
    err = (out - exp).abs().max().item() / exp.abs().max().item()
    assert err < 8e-3, err


def _moe_fp32_definition(x, w13, w2, ids, w, inter):
    t, hidden = x.shape
    exp = torch.zeros((t, hidden), dtype=torch.float32, device=x.device)
    xf = x.float()
    for ex in range(w13.shape[0]):
        rows, slot = (ids == ex).nonzero(as_tuple=True)
        if rows.numel() == 0:
            continue
        gu = xf[rows] @ w13[ex].float().t()
        a = torch.nn.functional.silu(gu[:, :inter]) * gu[:, inter:]
        exp.index_add_(0, rows, (a @ w2[ex].float().t()) * w[rows, slot].unsqueeze(1))
    return exp


@pytest.mark.parametrize("m,n,k,groups", [
    (256, 256, 64, [256]), (1000, 768, 512, [1000]), (4096, 1024, 4096, [4096]),
    (3000, 512, 256, [0, 700, 1, 0, 1299, 255, 745, 0]), (777, 320, 192, [300, 477])])
def test_grouped_gemm_against_fp32(m, n, k, groups):
    """Grouped MFMA GEMM (device-side problem list): ragged group sizes including empty groups and
    single rows, N and M that are not multiples of the tile, rows past the last group untouched."""
    torch.manual_seed(31)
    g = len(groups)
    assert sum(groups) == m
    a = (torch.rand(m + 50, k, device=DEV) * 2 - 1).to(BF)
    w = ((torch.rand(g, n, k, device=DEV) * 2 - 1) * 0.1).to(BF)
    off = torch.tensor([0] + list(np.cumsum(groups)), dtype=torch.int32, device=DEV)
    out = torch.full((m + 50, n), 7.0, dtype=BF, device=DEV)
    ops.grouped_gemm(a, w, off, rows=m + 50, out=out)
    lo = 0
    for i, cnt in enumerate(groups):
        if cnt:
            exp = a[lo:lo + cnt].float() @ w[i].float().t()
            err = (out[lo:lo + cnt].float() - exp).abs().max().item() / exp.abs().max().item()
            assert err < 6e-3, (i, err)
        lo += cnt
    assert (out[m:] == 7.0).all()  # rows that belong to no group are not written


def test_grouped_gemm_gathered_rows_and_fused_swiglu():
    torch.manual_seed(32)
    t, k, inter, e = 600, 512, 384, 4
    x = (torch.rand(t, k, device=DEV) * 2 - 1).to(BF)
    w13 = ((torch.rand(e, 2 * inter, k, device=DEV) * 2 - 1) * 0.1).to(BF)
    ids = torch.randint(0, e, (t, 2), device=DEV, dtype=torch.int32)
    offsets, pos, sorted_tok = ops.moe_align(ids, e)
    act = ops.grouped_gemm(x, w13, offsets, rows=2 * t, arow_idx=sorted_tok, silu_mul=True)
    off = offsets.tolist()
    for ex in range(e):
        lo, hi = off[ex], off[ex + 1]
        gu = x[sorted_tok[lo:hi].long()].float() @ w13[ex].float().t()
        exp = torch.nn.functional.silu(gu[:, :inter]) * gu[:, inter:]
        err = (act[lo:hi].float() - exp).abs().max().item() / exp.abs().max().item()
        assert err < 8e-3, (ex, err)


def _rel(got, exp):
    return (got.float() - exp).abs().max().item() / exp.abs().max().clamp_min(1e-9).item()


@pytest.mark.parametrize("m,n,k", [
    (1, 8, 64), (255, 264, 128), (256, 256, 64), (257, 512, 192), (700, 1032, 320), (2048, 6144, 4096),
    (4616, 4096, 4096), (300, 4096, 14336)])
def test_tile_gemm_against_fp32(m, n, k):
    """The prefill GEMM kernel (256 x 256 tiles, 16x16x32 MFMA, ping-pong wave rows) against the fp32
    product: ragged M, N that is not a multiple of the tile, one and many K tiles (odd counts too),
    Llama-3-8B projection shapes; plain store, residual epilogue (separate output and in place),
    bias (a residual of row stride 0), rows beyond the problem untouched."""
    torch.manual_seed(41)
    x = (torch.rand(m, k, device=DEV) * 2 - 1).to(BF)
    w = ((torch.rand(n, k, device=DEV) * 2 - 1) * 0.1).to(BF)
    ref = x.float() @ w.float().t()
    out = torch.full((m + 3, n), 7.0, dtype=BF, device=DEV)
    ops.tile_gemm(x, w, out=out, rows=m)
    assert _rel(out[:m], ref) < 6e-3
    assert (out[m:] == 7.0).all()
    r = (torch.rand(m, n, device=DEV) * 2 - 1).to(BF)
    assert _rel(ops.tile_gemm(x, w, residual=r), ref + r.float()) < 6e-3
    r2 = r.clone()
    ops.tile_gemm(x, w, residual=r2, out=r2)
    assert _rel(r2, ref + r.float()) < 6e-3
    b = (torch.rand(n, device=DEV) * 2 - 1).to(BF)
    assert _rel(ops.tile_gemm(x, w, residual=b.view(1, -1).expand(m, -1)), ref + b.float()) < 6e-3


def test_tile_gemm_exact_integers_catch_a_permuted_store():
    """A = identity rows, W = asymmetric small integers (exact in bf16): the output must be W^T bit
    for bit.  A transposed accumulator map, a wrong column order inside the 8-column groups the
    epilogue writes, or a swizzle mismatch between the load and the fragment reads all fail this."""
    n = k = 768
    m = 512
    x = torch.zeros(m, k, device=DEV)
    x[torch.arange(m), (torch.arange(m) * 5 + 3) % k] = 1.0
    w = ((torch.arange(n, device=DEV).view(n, 1) * 3 + torch.arange(k, device=DEV).view(1, k) * 7) % 251).float()
    got = ops.tile_gemm(x.to(BF), w.to(BF))
    assert torch.equal(got.float(), x @ w.t())


@pytest.mark.parametrize("m,inter,k", [(300, 96, 128), (513, 14336, 256), (2048, 1024, 4096)])
def test_tile_gemm_swiglu_epilogue(m, inter, k):
    torch.manual_seed(42)
    x = (torch.rand(m, k, device=DEV) * 2 - 1).to(BF)
    w = ((torch.rand(2 * inter, k, device=DEV) * 2 - 1) * 0.1).to(BF)
    y = x.float() @ w.float().t()
    ref = torch.nn.functional.silu(y[:, :inter]) * y[:, inter:]
    assert _rel(ops.tile_gemm(x, w, silu_mul=True), ref) < 8e-3


def test_tile_gemm_grouped_with_gather():
    """Device-side problem list: empty and one-row groups, gathered input rows, rows past the last
    group untouched."""
    torch.manual_seed(43)
    g, n, k, t = 6, 520, 256, 1500
    counts = [0, 300, 1, 899, 0, 257]
    rows = sum(counts)
    x = (torch.rand(t, k, device=DEV) * 2 - 1).to(BF)
    w = ((torch.rand(g, n, k, device=DEV) * 2 - 1) * 0.1).to(BF)
    off = torch.tensor([0] + list(np.cumsum(counts)), dtype=torch.int32, device=DEV)
    idx = torch.randint(0, t, (rows + 9,), dtype=torch.int32, device=DEV)
    out = torch.full((rows + 9, n), 7.0, dtype=BF, device=DEV)
    ops.tile_gemm(x, w, offsets=off, rows=rows + 9, arow_idx=idx, out=out)
    lo = 0
    for gi, cnt in enumerate(counts):
        if cnt:
            exp = x[idx[lo:lo + cnt].long()].float() @ w[gi].float().t()
            assert _rel(out[lo:lo + cnt], exp) < 6e-3, gi
        lo += cnt
    assert (out[rows:] == 7.0).all()


def test_projection_ops_route_prefill_batches_to_the_tile_kernel():
    """``linear`` / ``linear_add`` / ``gate_up_silu`` on a prefill-sized batch whose tiles fill the
    chip go through the hand-written kernel (ragged last row tile included) and equal the fp32
    definition; small batches and half-empty tile rounds stay on the other paths."""
    torch.manual_seed(44)
    m, k, n = 4096 - 37, 512, 4096
    x = (torch.rand(m, k, device=DEV) * 2 - 1).to(BF)
    w = ((torch.rand(n, k, device=DEV) * 2 - 1) * 0.1).to(BF)
    assert ops.tile_rows(x, w) == m
    assert ops.tile_rows(x[:4096 - 256 * 3], w) == 0, "13 x 16 tiles leave the chip's only round a fifth empty"
    assert ops.tile_rows(x[:300], w) == 0, "small batches stay on the streaming / library path"
    ref = x.float() @ w.float().t()
    assert _rel(ops.linear(x, w), ref) < 6e-3
    b = (torch.rand(n, device=DEV) * 2 - 1).to(BF)
    assert _rel(ops.linear(x, w, b), ref + b.float()) < 6e-3
    res = (torch.rand(m, n, device=DEV) * 2 - 1).to(BF)
    want = ref + res.float()
    assert ops.linear_add(x, w, res) is None
    assert _rel(res, want) < 6e-3
    inter = n // 2
    want = torch.nn.functional.silu(ref[:, :inter]) * ref[:, inter:]
    assert _rel(ops.gate_up_silu(x, w), want) < 8e-3
    old = ops.set_tile_gemm(False)
    try:
        assert ops.tile_rows(x, w) == 0
        h = ops.linear_add(x, w, res)
        assert h is not None and _rel(h, ref) < 6e-3
    finally:
        ops.set_tile_gemm(old)


@pytest.mark.parametrize("t,hidden,inter,skew", [
    (300, 512, 1024, False), (1024, 512, 1024, True), (4096, 4096, 14336, False)])
def test_moe_prefill_grouped_gemm_experts(t, hidden, inter, skew):
    """Prefill-sized expert MLP (grouped GEMMs over the device-sorted pairs, no host sync) against
    the per-pair fp32 definition, up to a whole prefill chunk of Mixtral-8x7B shape."""
    torch.manual_seed(23)
    e, k = 8, 2
    x = torch.randn(t, hidden, device=DEV).to(BF)
    w13 = (torch.randn(e, 2 * inter, hidden, device=DEV) * 0.03).to(BF)
    w2 = (torch.randn(e, hidden, inter, device=DEV) * 0.03).to(BF)
    logits = torch.randn(t, e, device=DEV).to(BF)
    if skew:
        logits[:, 3] += 20.0
        logits[:, 6] += 10.0
    w, ids = ops.moe_topk_softmax(logits, k)
    out = ops.moe_experts_prefill(x, w13, w2, ids, w).float()
    exp = _moe_fp32_definition(x, w13, w2, ids, w, inter)
    err = (out - exp).abs().max().item() / exp.abs().max().item()
    assert err < 8e-3, err


@pytest.mark.parametrize("t,hidden,inter,ranks,skew", [
    (3, 512, 1024, 2, False), (40, 512, 1024, 4, False), (40, 512, 1024, 2, True),
    (300, 512, 1024, 2, False), (1024, 512, 1024, 4, True), (2048, 4096, 14336, 2, False)])
def test_moe_expert_parallel_ranks_sum_to_the_whole_layer(t, hidden, inter, ranks, skew):
    """Expert parallelism: every rank holds a range of whole experts and runs only the pairs routed
    there (the others sort into a bucket no kernel visits and carry weight 0).  The ranks' outputs,
    summed as the layer's all-reduce would, must be the whole layer's fp32 definition.  `skew`
    routes every token to experts 3 and 6, so some ranks have no pair at all."""
    torch.manual_seed(29)
    e, k = 8, 2
    x = torch.randn(t, hidden, device=DEV).to(BF)
    w13 = (torch.randn(e, 2 * inter, hidden, device=DEV) * 0.03).to(BF)
    w2 = (torch.randn(e, hidden, inter, device=DEV) * 0.03).to(BF)
    logits = torch.randn(t, e, device=DEV).to(BF)
    if skew:
        logits[:, 3] += 20.0
        logits[:, 6] += 10.0
    w, ids = ops.moe_topk_softmax(logits, k)
    exp = _moe_fp32_definition(x, w13, w2, ids, w, inter)
    per = e // ranks
    total = torch.zeros_like(exp)
    for r in range(ranks):
        lids, lw = ops.moe_localize(ids, w, r * per, per, bucket=True)
        assert int(lids.max()) <= per and bool(((lids == per) == (lw == 0)).all())
        mine = w13[r * per:(r + 1) * per].contiguous(), w2[r * per:(r + 1) * per].contiguous()
        # poison what the allocator hands out next: rows of foreign pairs are never written and must not be read
        junk = torch.full((t * k, max(hidden, inter)), float("nan"), device=DEV, dtype=BF)
        del junk
        fn = ops.moe_experts_grouped if t <= ops.MOE_MFMA_MAX_TOKENS else ops.moe_experts_prefill
        out = fn(x, mine[0], mine[1], lids, lw, foreign=True).float()
        assert torch.isfinite(out).all()
        total += out
    err = (total - exp).abs().max().item() / exp.abs().max().item()
    assert err < 8e-3, err


@pytest.mark.parametrize("t,in_dim,outs,r_pad", [
    (1, 4096, [4096, 1024, 1024], 32), (37, 512, [256], 32), (300, 1024, [512, 512], 64),
    (40, 14336, [4096], 32)])
def test_lora_shrink_expand_mixed_adapters(t, in_dim, outs, r_pad):
    """Batched multi-adapter LoRA kernels against the fp32 definition: rows carry different
    adapter slots (or none) in arbitrary order."""
    torch.manual_seed(21)
    n_slots = 4
    g = torch.Generator().manual_seed(5)
    slots = torch.randint(-1, n_slots, (t,), generator=g).to(torch.int32).to(DEV)
    if t > 32:
        slots[16:32] = 2  # a uniform tile next to mixed ones
    x = torch.randn(t, in_dim, device=DEV).to(BF)
    a = (torch.randn(n_slots, len(outs) * r_pad, in_dim, device=DEV) * 0.05).to(BF)
    bs = [(torch.randn(n_slots, o, r_pad, device=DEV) * 0.05).to(BF) for o in outs]
    y = torch.randn(t, sum(outs) + 64, device=DEV).to(BF)  # wider than the outputs: strided rows
    y0 = y.clone()
    h = ops.lora_shrink(x, a, slots, a.shape[1])
    h_ref = ref.lora_shrink(x, a, slots)
    has = slots >= 0
    torch.testing.assert_close(h[has].float(), h_ref[has].float(), atol=2e-2, rtol=2e-2)
    off = 0
    y_ref = y0.clone()
    ops.lora_expand(h, bs, slots, y, r_pad)
    for sub, b in enumerate(bs):
        ref.lora_expand(h, sub * r_pad, b, slots, y_ref, off)
        off += b.shape[1]
    torch.testing.assert_close(y.float(), y_ref.float(), atol=3e-2, rtol=2e-2)
    assert torch.equal(y[~has], y0[~has])  # rows without an adapter are untouched
    assert torch.equal(y[:, off:], y0[:, off:])
