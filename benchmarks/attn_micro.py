"""Kernel-level benchmark of the paged attention kernels at the multi-round-QA shapes
(Llama-3-8B geometry: 32 query heads, 8 KV heads, head_dim 128, 64-token pages).

Cases (one JSON line each):
  prefill_cold   one sequence, an 8192-token chunk behind an 8192-token cached prefix
  prefill_split  part-filled launches, whole vs. cut along the key range (off by default)
  prefill_warm   40 sequences, 128 new tokens each behind ~21k cached tokens (a follow-up turn)
  decode         40 sequences, one token each, ~21.5k-token contexts (distinct pages)

Every timed launch touches far more KV than the 256 MiB Infinity Cache holds, inputs are random
(never zero-filled), and variants are timed interleaved in one process.

    python benchmarks/attn_micro.py [--iters 10] [--cases prefill_cold,prefill_warm,decode]
"""

import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from vllm_production_stack_amd import ops  # noqa: E402

N_Q, N_KV, HD, PAGE = 32, 8, 128, 64
DEV = "cuda"


POOL_SPREAD = 1.0  # decode case: the pool is this many times larger than the pages that are read (--pool-spread)
PAGE_PAD = 0  # bytes between the end of a page and the start of the next (--page-pad-kb)
LAYERS = 1  # layers per page of the pool (--layers 32: the engine's page, 8 MiB for Llama-3-8B)


def make_pool(pages: int, seed: int = 0) -> torch.Tensor:
    g = torch.Generator(device=DEV).manual_seed(seed)
    if PAGE_PAD:  # pages PAGE_PAD bytes further apart than they are long
        elems = LAYERS * 2 * N_KV * PAGE * HD
        flat = torch.empty(pages * (elems + PAGE_PAD // 2), device=DEV, dtype=torch.bfloat16)
        pool = flat.as_strided((pages, LAYERS, 2, N_KV, PAGE * HD),
                               (elems + PAGE_PAD // 2, 2 * N_KV * PAGE * HD, N_KV * PAGE * HD, PAGE * HD, 1))
    else:
        pool = torch.empty((pages, LAYERS, 2, N_KV, PAGE * HD), device=DEV, dtype=torch.bfloat16)
    if LAYERS == 1:
        pool.normal_(0.0, 1.0, generator=g)
    else:  # only the layer that is read needs values
        pool[:, LAYERS // 2].normal_(0.0, 1.0, generator=g)
    return pool


def timed(fn, iters: int) -> float:
    """Median milliseconds per call."""
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]


def prefill_case(name: str, n_seqs: int, q_len: int, prefix: int, iters: int,
                 variants=(0, 1, 2)) -> dict:
    ctx = prefix + q_len
    pages_per_seq = (ctx + PAGE - 1) // PAGE
    pool = make_pool(n_seqs * pages_per_seq + 1)
    g = torch.Generator().manual_seed(1)
    perm = torch.randperm(n_seqs * pages_per_seq, generator=g).view(n_seqs, pages_per_seq)
    bt = perm.to(torch.int32).to(DEV)
    q = torch.randn((n_seqs * q_len, N_Q * HD), device=DEV).to(torch.bfloat16)
    out = torch.empty_like(q)
    cu = torch.arange(0, (n_seqs + 1) * q_len, q_len, dtype=torch.int32, device=DEV)
    ctxs = torch.full((n_seqs,), ctx, dtype=torch.int32, device=DEV)
    scale = HD ** -0.5
    tiles = {}
    for v in variants:  # the work list depends on the kernel's tile height
        ops.set_prefill_variant(v)
        rows = ops.prefill_tile_rows([q_len] * n_seqs, N_Q, N_KV)
        ts, tq = ops.build_prefill_tiles([q_len] * n_seqs, N_Q, N_KV, rows)
        tiles[v] = (torch.tensor(ts, dtype=torch.int32, device=DEV),
                    torch.tensor(tq, dtype=torch.int32, device=DEV), rows)
    cur = {"v": variants[0]}

    def run():
        ts, tq, rows = tiles[cur["v"]]
        ops.paged_prefill(q, out, pool, 0, bt, cu, ctxs, ts, tq, N_Q, N_KV, HD, scale, rows)

    keys = sum(prefix + i + 1 for i in range(q_len)) * n_seqs  # causal (query, key) pairs
    flops = 4.0 * keys * HD * N_Q
    res = {"case": name, "n_seqs": n_seqs, "q_len": q_len, "prefix": prefix}
    # variants interleaved in one process: round-robin over the kernel generations
    samples = {v: [] for v in variants}
    for _ in range(3):
        for v in variants:
            ops.set_prefill_variant(v)
            cur["v"] = v
            samples[v].append(timed(run, iters))
    for v in variants:
        ms = sorted(samples[v])[1]
        res[f"variant{v}"] = {"ms": round(ms, 4), "tflops": round(flops / ms / 1e9, 1)}
    ops.set_prefill_variant(-1)
    return res


def split_case(n_seqs: int, q_len: int, prefix: int, iters: int) -> dict:
    """The automatic kernel on a work list as built vs. the same list with the key ranges cut by
    ``ops.plan_prefill_splits`` (what the engine runs)."""
    ctx = prefix + q_len
    pages_per_seq = (ctx + PAGE - 1) // PAGE
    pool = make_pool(n_seqs * pages_per_seq + 1)
    g = torch.Generator().manual_seed(1)
    bt = torch.randperm(n_seqs * pages_per_seq, generator=g).view(n_seqs, pages_per_seq).to(torch.int32).to(DEV)
    q = torch.randn((n_seqs * q_len, N_Q * HD), device=DEV).to(torch.bfloat16)
    out = torch.empty_like(q)
    out2 = torch.empty_like(q)
    cu = torch.arange(0, (n_seqs + 1) * q_len, q_len, dtype=torch.int32, device=DEV)
    ctxs = torch.full((n_seqs,), ctx, dtype=torch.int32, device=DEV)
    scale = HD ** -0.5
    q_lens, ctx_lens = [q_len] * n_seqs, [ctx] * n_seqs
    rows = ops.prefill_tile_rows(q_lens, N_Q, N_KV)
    ts, tq = ops.build_prefill_tiles(q_lens, N_Q, N_KV, rows, ctx_lens)
    res = {"case": "prefill_split", "n_seqs": n_seqs, "q_len": q_len, "prefix": prefix,
           "entries": len(ts)}
    dev = lambda v: torch.tensor(v, dtype=torch.int32, device=DEV)
    d_ts, d_tq = dev(ts), dev(tq)
    keys = sum(prefix + i + 1 for i in range(q_len)) * n_seqs
    flops = 4.0 * keys * HD * N_Q

    def whole():
        ops.paged_prefill(q, out, pool, 0, bt, cu, ctxs, d_ts, d_tq, N_Q, N_KV, HD, scale, rows)
    cut = ops.plan_prefill_splits(ts, tq, q_lens, ctx_lens, N_Q, N_KV, rows)
    runs = {"whole": whole}
    if cut is not None:
        ts2, tq2, split, groups, n_rec = cut
        d_ts2, d_tq2, d_split, d_groups = dev(ts2), dev(tq2), dev(split), dev(groups)
        ws = torch.empty(n_rec * ops.prefill_split_record_bytes(N_Q, N_KV) // 4, dtype=torch.float32, device=DEV)
        pieces = [split[3 * i + 1] - split[3 * i] for i in range(len(ts2))]
        res.update(cut_entries=len(ts2), records=n_rec, piece_tiles=[min(pieces), max(pieces)])

        def planned():
            ops.paged_prefill(q, out2, pool, 0, bt, cu, ctxs, d_ts2, d_tq2, N_Q, N_KV, HD, scale, rows,
                              split=d_split, split_ws=ws, groups=d_groups)
        runs["planned"] = planned
    samples = {k: [] for k in runs}
    for _ in range(3):
        for k, fn in runs.items():
            samples[k].append(timed(fn, iters))
    for k in runs:
        ms = sorted(samples[k])[1]
        res[k] = {"ms": round(ms, 4), "tflops": round(flops / ms / 1e9, 1)}
    if cut is not None:
        res["max_abs_diff"] = float((out.float() - out2.float()).abs().max())
    return res


def decode_case(n_seqs: int, ctx: int, iters: int) -> dict:
    pages_per_seq = (ctx + PAGE - 1) // PAGE
    total = int((n_seqs * pages_per_seq) * POOL_SPREAD)
    pool = make_pool(total + 1)
    g = torch.Generator().manual_seed(2)
    perm = torch.randperm(total, generator=g)[: n_seqs * pages_per_seq].view(n_seqs, pages_per_seq)
    bt = perm.to(torch.int32).to(DEV)
    q = torch.randn((n_seqs, N_Q * HD), device=DEV).to(torch.bfloat16)
    out = torch.empty_like(q)
    ctxs = torch.full((n_seqs,), ctx, dtype=torch.int32, device=DEV)
    scale = HD ** -0.5
    res = {"case": "decode", "n_seqs": n_seqs, "ctx": ctx}
    kv_bytes = n_seqs * ctx * N_KV * HD * 2 * 2
    for splits in sorted({1, 2, 3, 4, 6, 8, ops.decode_num_splits(n_seqs, N_KV, ctx)}):
        ws_o = torch.empty((n_seqs, N_Q, splits, HD), dtype=torch.float32, device=DEV)
        ws_ml = torch.empty((n_seqs, N_Q, splits, 2), dtype=torch.float32, device=DEV)

        def run():
            ops.paged_decode(q, out, pool, LAYERS // 2, bt, ctxs, N_Q, N_KV, HD, scale, splits, ws_o, ws_ml)

        entry = {}
        for dv in (0, 1, 2, 3, 4, 5):  # 3-5: the same three with non-temporal K/V loads
            ops.set_decode_variant(dv)
            ms = timed(run, iters)
            entry[f"v{dv}"] = {"ms": round(ms, 4), "tb_per_s": round(kv_bytes / ms / 1e9, 3)}
        ops.set_decode_variant(-1)
        res[f"splits{splits}"] = entry
    res["default_splits"] = ops.decode_num_splits(n_seqs, N_KV, ctx)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--cases", default="prefill_cold,prefill_warm,decode")
    ap.add_argument("--variants", default="2,3,4,5,6,7,-1", help="prefill kernels to time (-1 = automatic choice)")
    ap.add_argument("--layers", type=int, default=1,
                    help="layers per page of the KV pool in the decode case (32: the engine's layout for Llama-3-8B)")
    ap.add_argument("--pool-spread", type=float, default=1.0,
                    help="decode case: size of the pool relative to the pages read (they are scattered over all of it)")
    ap.add_argument("--decode-seqs", default="40", help="decode case: batch sizes (comma separated)")
    ap.add_argument("--page-pad-kb", type=int, default=0, help="decode case: spacing added between pages of the pool")
    ap.add_argument("--heads", default="", help="query,kv head counts instead of Llama-3-8B's 32,8 (e.g. 28,4: Qwen2.5-7B)")
    a = ap.parse_args()
    global LAYERS, PAGE_PAD, POOL_SPREAD
    POOL_SPREAD = a.pool_spread
    LAYERS = a.layers
    PAGE_PAD = a.page_pad_kb * 1024
    if a.heads:
        global N_Q, N_KV
        N_Q, N_KV = (int(v) for v in a.heads.split(","))
    cases = a.cases.split(",")
    variants = tuple(int(v) for v in a.variants.split(","))
    if "prefill_cold" in cases:
        print(json.dumps(prefill_case("prefill_cold", 1, 8192, 8192, a.iters, variants)), flush=True)
    if "prefill_first" in cases:  # first chunk of a prompt: purely causal, nothing cached
        print(json.dumps(prefill_case("prefill_first", 1, 8192, 0, a.iters, variants)), flush=True)
    if "prefill_last" in cases:  # last chunk of a 21k-token prompt
        print(json.dumps(prefill_case("prefill_last", 1, 4616, 16384, a.iters, variants)), flush=True)
    if "prefill_bulk" in cases:  # enough workgroups that the tail of the launch does not matter
        print(json.dumps(prefill_case("prefill_bulk", 8, 4096, 8192, a.iters, variants)), flush=True)
    if "prefill_warm" in cases:
        print(json.dumps(prefill_case("prefill_warm", 40, 128, 21000, a.iters, variants)), flush=True)
    if "prefill_split" in cases:  # part-filled launches: whole work list vs. planned key-range cuts
        for n, ql, pre in ((1, 64, 21000), (1, 150, 21000), (4, 150, 21000), (12, 150, 21000),
                           (20, 150, 21000), (40, 128, 21000), (1, 1024, 8192), (1, 2048, 18000)):
            print(json.dumps(split_case(n, ql, pre, a.iters)), flush=True)
    if "decode" in cases:
        for n in (int(v) for v in a.decode_seqs.split(",")):
            print(json.dumps(decode_case(n, 21500, a.iters)), flush=True)


if __name__ == "__main__":
    main()
