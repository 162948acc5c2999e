"""Plain PyTorch (fp32 math) definitions of every engine op.

These are the numerical oracles for the HIP kernels in ``csrc/kernels`` and the execution
path for CPU tensors (unit tests, CPU-only plumbing replicas).  They share the paged
KV-cache layout with the kernels:

    pool[page, layer, 0, kv_head]  viewed as  [PAGE, head_dim]   (K, one row per key)
    pool[page, layer, 1, kv_head]  viewed as  [head_dim, PAGE]   (V, transposed)
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
    k = pool[:, layer, 0].view(pool.shape[0], n_kv, PAGE, head_dim)
    v = pool[:, layer, 1].view(pool.shape[0], n_kv, head_dim, PAGE)
    return k, v


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
    kview[page, :, off, :] = x[ok, n_q : n_q + n_kv]
    vview[page, :, :, off] = x[ok, n_q + n_kv :]


def _gather_kv(pool, layer, n_kv, head_dim, block_row, length):
    kview, vview = _kv_views(pool, layer, n_kv, head_dim)
    npages = (length + PAGE - 1) // PAGE
    pages = block_row[:npages].long()
    k = kview[pages].permute(1, 0, 2, 3).reshape(n_kv, npages * PAGE, head_dim)[:, :length]
    v = vview[pages].permute(1, 2, 0, 3).reshape(n_kv, head_dim, npages * PAGE)[:, :, :length]
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
    tok = torch.where(is_k, (gin // 2) % 64, torch.zeros_like(gin))
    anchor = x[gidx - tok * 2]
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
    tok = torch.where(is_k, (gin // 2) % 64, torch.zeros_like(gin))
    a = gidx - tok * 2
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
