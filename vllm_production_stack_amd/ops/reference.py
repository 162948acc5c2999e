"""Plain PyTorch (fp32 math) definitions of every engine op.

These are the numerical oracles for the HIP kernels in ``csrc/kernels`` and the execution
path for CPU tensors (unit tests, CPU-only plumbing replicas).  They share the paged
KV-cache layout with the kernels:

    pool[page, layer, 0, kv_head]  viewed as  [head_dim / 32, PAGE, 32]   (K: channel block, key, channel)
    pool[page, layer, 1, kv_head]  viewed as  [PAGE / 32, head_dim, 32]   (V: key half, channel, key)

(blocks of 32 = 64 bytes, so that a 16-row MFMA fragment load of the decode kernel is one
contiguous KiB; head sizes that are not a multiple of 32 use one block of the whole head).
``_kv_views`` presents both as [.., PAGE, head_dim] / [.., head_dim, PAGE] whatever the blocks.
"""

from __future__ import annotations

import torch

PAGE = 64


def rms_norm(x, weight, eps, residual=None):
    if residual is not None:
        s = x.float() + residual.float()
        residual.copy_(s.to(residual.dtype))
        xf = residual.float()
    else:
        xf = x.float()
    var = xf.pow(2).mean(-1, keepdim=True)
    return (xf * torch.rsqrt(var + eps) * weight.float()).to(x.dtype)


def layer_norm(x, weight, bias, eps):
    return torch.nn.functional.layer_norm(x.float(), (x.shape[-1],), weight.float(), bias.float(), eps).to(x.dtype)


def build_cos_sin(head_dim: int, max_pos: int, theta: float, scaling: dict | None = None, device="cpu"):
    """[max_pos, head_dim] fp32 table: cos for the head_dim/2 frequencies, then sin."""
    half = head_dim // 2
    inv = 1.0 / (theta ** (torch.arange(0, half, dtype=torch.float64) / half))
    if scaling and scaling.get("rope_type", scaling.get("type")) == "llama3":
        factor = scaling.get("factor", 8.0)
        lo = scaling.get("low_freq_factor", 1.0)
        hi = scaling.get("high_freq_factor", 4.0)
        orig = scaling.get("original_max_position_embeddings", 8192)
        wavelen = 2 * torch.pi / inv
        smooth = ((orig / wavelen) - lo) / (hi - lo)
        scaled = torch.where(wavelen > orig / lo, inv / factor, inv)
        mid = (1 - smooth) * inv / factor + smooth * inv
        is_mid = (wavelen <= orig / lo) & (wavelen >= orig / hi)
        inv = torch.where(is_mid, mid, scaled)
    ang = torch.arange(max_pos, dtype=torch.float64)[:, None] * inv[None, :]
    return torch.cat([ang.cos(), ang.sin()], dim=-1).float().to(device)


def _kv_views(pool, layer, n_kv, head_dim):
    """K as [pages, n_kv, PAGE, head_dim] and V as [pages, n_kv, head_dim, PAGE]: permuted views of
    the pool (reads and indexed writes go through to it)."""
    cb = 32 if head_dim % 32 == 0 else head_dim
    n = pool.shape[0]
    k = pool[:, layer, 0].view(n, n_kv, head_dim // cb, PAGE, cb).permute(0, 1, 3, 2, 4)
    v = pool[:, layer, 1].view(n, n_kv, PAGE // 32, head_dim, 32).permute(0, 1, 3, 2, 4)
    return k, v  # [n, n_kv, PAGE, blocks, cb], [n, n_kv, head_dim, 2, 32]


def rope_and_cache(qkv, positions, slots, cos_sin, pool, layer, n_q, n_kv, head_dim, rotary=True):
    """Rotate Q and K heads of ``qkv`` in place (NeoX pairing) and write K/V into the pool."""
    t = qkv.shape[0]
    if t == 0:
        return
    half = head_dim // 2
    x = qkv.view(t, n_q + 2 * n_kv, head_dim)
    if rotary:
        cs = cos_sin[positions.long()]
        cos, sin = cs[:, None, :half], cs[:, None, half:]
        qk = x[:, : n_q + n_kv].float()
        a, b = qk[..., :half], qk[..., half:]
        rot = torch.cat([a * cos - b * sin, b * cos + a * sin], dim=-1)
        x[:, : n_q + n_kv] = rot.to(qkv.dtype)
    if slots is None:
        return
    kview, vview = _kv_views(pool, layer, n_kv, head_dim)
    ok = slots >= 0
    s = slots[ok].long()
    page, off = s // PAGE, s % PAGE
    cb = kview.shape[-1]
    kview[page, :, off] = x[ok, n_q : n_q + n_kv].reshape(-1, n_kv, head_dim // cb, cb)
    vview[page, :, :, off // 32, off % 32] = x[ok, n_q + n_kv :]


def _gather_kv(pool, layer, n_kv, head_dim, block_row, length):
    kview, vview = _kv_views(pool, layer, n_kv, head_dim)
    npages = (length + PAGE - 1) // PAGE
    pages = block_row[:npages].long()
    k = kview[pages].reshape(npages, n_kv, PAGE, head_dim).permute(1, 0, 2, 3).reshape(
        n_kv, npages * PAGE, head_dim)[:, :length]
    v = vview[pages].reshape(npages, n_kv, head_dim, PAGE).permute(1, 2, 0, 3).reshape(
        n_kv, head_dim, npages * PAGE)[:, :, :length]
    return k.float(), v.float().transpose(1, 2)


def paged_prefill(q, out, pool, layer, block_tables, cu_q, ctx_lens, n_q, n_kv, head_dim, scale):
    """Causal attention of each sequence's new rows against its cached prefix + new keys."""
    g = n_q // n_kv
    cu = cu_q.tolist()
    for s in range(len(cu) - 1):
        r0, r1 = cu[s], cu[s + 1]
        ql = r1 - r0
        if ql == 0:
            continue
        ctx = int(ctx_lens[s])
        k, v = _gather_kv(pool, layer, n_kv, head_dim, block_tables[s], ctx)
        qs = q[r0:r1].view(ql, n_kv, g, head_dim).float()
        sc = torch.einsum("qhgd,hkd->hgqk", qs, k) * scale
        qpos = torch.arange(ctx - ql, ctx, device=q.device)[:, None]
        kpos = torch.arange(ctx, device=q.device)[None, :]
        sc = sc.masked_fill(kpos > qpos, float("-inf"))
        p = torch.softmax(sc, dim=-1)
        o = torch.einsum("hgqk,hkd->qhgd", p, v)
        out[r0:r1] = o.reshape(ql, n_q * head_dim).to(out.dtype)


def paged_decode(q, out, pool, layer, block_tables, ctx_lens, n_q, n_kv, head_dim, scale):
    g = n_q // n_kv
    for s in range(q.shape[0]):
        ctx = int(ctx_lens[s])
        if ctx == 0:
            out[s] = 0
            continue
        k, v = _gather_kv(pool, layer, n_kv, head_dim, block_tables[s], ctx)
        qs = q[s].view(n_kv, g, head_dim).float()
        sc = torch.einsum("hgd,hkd->hgk", qs, k) * scale
        p = torch.softmax(sc, dim=-1)
        o = torch.einsum("hgk,hkd->hgd", p, v)
        out[s] = o.reshape(n_q * head_dim).to(out.dtype)


def silu_and_mul(x):
    inter = x.shape[-1] // 2
    g, u = x[..., :inter].float(), x[..., inter:].float()
    return (torch.nn.functional.silu(g) * u).to(x.dtype)


def argmax(logits):
    return torch.argmax(logits.float(), dim=-1)


def moe_topk_softmax(logits, topk, renorm=True):
    p = torch.softmax(logits.float(), dim=-1)
    w, ids = torch.topk(p, topk, dim=-1)
    if renorm:
        w = w / w.sum(-1, keepdim=True)
    return w, ids.int()


def cachegen_encode(staging, levels, unit_elems, n_layers):
    """Reference of the codec front end: returns (int8 symbols, fp32 scales)."""
    flat = staging.reshape(-1).float()
    n_groups = flat.numel() // 64
    gpu = unit_elems // 64
    x = flat.view(n_groups, 64)
    gidx = torch.arange(n_groups, device=flat.device)
    unit = gidx // gpu
    gin = gidx % gpu
    is_k = (unit % 2) == 0
    lv = torch.as_tensor(levels, device=flat.device, dtype=torch.float32)[(unit // 2) % n_layers][:, None]
    tok = torch.where(is_k, gin % 32, torch.zeros_like(gin))  # key pair within the channel block
    anchor = x[gidx - tok]
    amax = anchor.abs().amax(-1, keepdim=True)
    sa = torch.where(amax > 0, amax / lv, torch.ones_like(amax))
    qa = torch.round(anchor / sa)
    d = x - qa * sa
    dmax = d.abs().amax(-1, keepdim=True)
    sd = torch.where(dmax > 0, dmax / lv, torch.ones_like(dmax))
    qd = torch.round(d / sd)
    first = (tok == 0)[:, None]
    sym = torch.where(first, qa, qd).to(torch.int8)
    scales = torch.where(first, sa, sd).squeeze(-1)
    return sym.view(-1), scales


def cachegen_decode(symbols, scales, unit_elems, dtype=torch.bfloat16):
    n_groups = scales.numel()
    gpu = unit_elems // 64
    q = symbols.view(n_groups, 64).float()
    gidx = torch.arange(n_groups, device=symbols.device)
    unit = gidx // gpu
    gin = gidx % gpu
    is_k = (unit % 2) == 0
    tok = torch.where(is_k, gin % 32, torch.zeros_like(gin))
    a = gidx - tok
    base = q[a] * scales[a][:, None]
    delta = torch.where((tok == 0)[:, None], torch.zeros_like(q), q * scales[:, None])
    return (base + delta).to(dtype).view(-1)


def moe_experts(x, w13, w2, ids, weights):
    """out[t] = sum_k weights[t,k] * W2[e] . (silu(Wg[e] x) * (Wu[e] x)),  e = ids[t,k]."""
    t, k = ids.shape
    inter = w2.shape[2]
    out = torch.zeros((t, x.shape[1]), dtype=torch.float32, device=x.device)
    for ti in range(t):
        for ki in range(k):
            e = int(ids[ti, ki])
            gu = (w13[e].float() @ x[ti].float()).to(x.dtype).float()
            a = (torch.nn.functional.silu(gu[:inter]) * gu[inter:]).to(x.dtype).float()
            out[ti] += float(weights[ti, ki]) * (w2[e].float() @ a).to(x.dtype).float()
    return out.to(x.dtype)


def lora_shrink(x, a, slots):
    """h[t] = A[slots[t]] @ x[t]; rows without an adapter (slot < 0) give zeros."""
    h = torch.zeros((x.shape[0], a.shape[1]), dtype=x.dtype, device=x.device)
    for s in torch.unique(slots).tolist():
        if s < 0:
            continue
        rows = (slots == s).nonzero(as_tuple=True)[0]
        h[rows] = (x[rows].float() @ a[s].float().t()).to(x.dtype)
    return h


def lora_expand(h, h_off, b, slots, y, y_off):
    out_dim, r_pad = b.shape[1], b.shape[2]
    for s in torch.unique(slots).tolist():
        if s < 0:
            continue
        rows = (slots == s).nonzero(as_tuple=True)[0]
        delta = h[rows, h_off:h_off + r_pad].float() @ b[s].float().t()
        y[rows, y_off:y_off + out_dim] = (y[rows, y_off:y_off + out_dim].float() + delta).to(y.dtype)
    return y


# --------------------------------------------------------------------------- rANS entropy stage
# NumPy definition of the interleaved rANS coder of csrc/kernels/rans.hip (same tables, same
# stream layout, same bytes): the oracle of the GPU kernels and the codec of CPU-only hosts.
RANS_BITS = 12
RANS_M = 1 << RANS_BITS
RANS_LOW = 1 << 23


def rans_tables(sym, n_ctx: int):
    """-> (freq [n_ctx, 256] uint16 summing to 4096 per row, cum [n_ctx, 256] uint32)."""
    import numpy as np

    sym = np.asarray(sym).view(np.uint8).reshape(n_ctx, -1)
    total = sym.shape[1]
    freq = np.zeros((n_ctx, 256), dtype=np.int64)
    for c in range(n_ctx):
        cnt = np.bincount(sym[c], minlength=256).astype(np.int64)
        f = np.where(cnt > 0, np.maximum(1, (cnt * RANS_M + total // 2) // total), 0)
        i = int(np.argmax(f))  # first index among equals
        f[i] += RANS_M - int(f.sum())
        if f[i] < 1:
            raise ValueError("frequency table cannot be normalised")
        freq[c] = f
    cum = np.cumsum(freq, axis=1) - freq
    return freq.astype(np.uint16), cum.astype(np.uint32)


def rans_encode(sym, n_ctx: int, spc: int):
    """-> (freq uint16 [n_ctx, 256], states uint32 [n_ctx * spc], sizes uint16, coded uint8)."""
    import numpy as np

    sym = np.asarray(sym).view(np.uint8).reshape(n_ctx, -1)
    ctx_len = sym.shape[1]
    length = ctx_len // spc
    cap = 2 * length
    freq, cum = rans_tables(sym, n_ctx)
    n_streams = n_ctx * spc
    buf = np.zeros((n_streams, cap), dtype=np.uint8)
    pos = np.full(n_streams, cap, dtype=np.int64)
    x = np.full(n_streams, RANS_LOW, dtype=np.uint64)
    ctx_of = np.repeat(np.arange(n_ctx), spc)
    rows = np.arange(n_streams)
    per = sym.reshape(n_ctx, length, spc)  # [ctx, k, j]
    for k in range(length - 1, -1, -1):
        s = per[:, k, :].reshape(-1)
        f = freq[ctx_of, s].astype(np.uint64)
        c = cum[ctx_of, s].astype(np.uint64)
        x_max = np.uint64((RANS_LOW >> RANS_BITS) << 8) * f
        while True:
            m = x >= x_max
            if not m.any():
                break
            pos[m] -= 1
            buf[rows[m], pos[m]] = (x[m] & np.uint64(0xFF)).astype(np.uint8)
            x[m] >>= np.uint64(8)
        x = ((x // f) << np.uint64(RANS_BITS)) + (x % f) + c
    sizes = (cap - pos).astype(np.uint16)
    coded = np.concatenate([buf[i, pos[i]:] for i in range(n_streams)]) if n_streams else np.zeros(0, np.uint8)
    return freq, x.astype(np.uint32), sizes, coded


def rans_decode(freq, states, sizes, coded, n: int, n_ctx: int, spc: int):
    """-> uint8 symbols [n]."""
    import numpy as np

    freq = np.asarray(freq, dtype=np.uint16).reshape(n_ctx, 256)
    cum = (np.cumsum(freq.astype(np.int64), axis=1) - freq).astype(np.int64)
    ctx_len = n // n_ctx
    length = ctx_len // spc
    n_streams = n_ctx * spc
    slot2sym = np.zeros((n_ctx, RANS_M), dtype=np.uint8)
    for c in range(n_ctx):
        slot2sym[c] = np.repeat(np.arange(256, dtype=np.uint8), freq[c].astype(np.int64))[:RANS_M]
    ctx_of = np.repeat(np.arange(n_ctx), spc)
    sizes64 = np.asarray(sizes).astype(np.int64)
    rp = np.cumsum(sizes64) - sizes64
    x = np.asarray(states).astype(np.uint64).copy()
    coded = np.concatenate([np.asarray(coded, dtype=np.uint8), np.zeros(8, np.uint8)])
    limit = coded.size - 8
    out = np.zeros((n_ctx, length, spc), dtype=np.uint8)
    f64 = freq.astype(np.uint64)
    for k in range(length):
        slot = (x & np.uint64(RANS_M - 1)).astype(np.int64)
        s = slot2sym[ctx_of, slot]
        out[:, k, :] = s.reshape(n_ctx, spc)
        x = f64[ctx_of, s] * (x >> np.uint64(RANS_BITS)) + slot.astype(np.uint64) - cum[ctx_of, s].astype(np.uint64)
        while True:
            m = x < np.uint64(RANS_LOW)
            if not m.any():
                break
            b = np.where(rp[m] < limit, coded[np.minimum(rp[m], limit)], 0).astype(np.uint64)
            x[m] = (x[m] << np.uint64(8)) | b
            rp[m] += 1
    return out.reshape(-1)
