"""Is a full-size step bitwise repeatable?  Prefills two prompts as pooling requests (the hidden state
of the last prompt token comes back: nothing but the prefill path is involved), several times over
with the prefix cache emptied in between, and compares the vectors bit by bit; then does the same with
greedy decoding.  One JSON line.

    python benchmarks/tools/engine_repeatability.py [--prompt-len N] [--eager] [--kv-layer-groups G]"""
import argparse
import copy
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vllm_production_stack_amd.engine.config import EngineConfig  # noqa: E402
from vllm_production_stack_amd.engine.llm_engine import LLMEngine  # noqa: E402
from vllm_production_stack_amd.engine.request import SamplingParams  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--prompt-len", type=int, default=1000)
    ap.add_argument("--repeats", type=int, default=6)
    ap.add_argument("--eager", action="store_true")
    ap.add_argument("--kv-layer-groups", type=int, default=0)
    ap.add_argument("--prefill-variant", type=int, default=-1, help="ops.set_prefill_variant")
    ap.add_argument("--trace-attention", action="store_true", help="look inside the second layer's attention")
    ap.add_argument("--trace", action="store_true", help="find the first op whose output differs between runs")
    a = ap.parse_args()
    eng = LLMEngine(EngineConfig(model="llama-3-8b", device="cuda", dtype="bfloat16", max_model_len=8192,
                                 max_num_seqs=8, enforce_eager=a.eager, kv_layer_groups=a.kv_layer_groups, port=0))
    if a.prefill_variant >= 0:
        from vllm_production_stack_amd import ops as _ops

        _ops.set_prefill_variant(a.prefill_variant)
    g = torch.Generator().manual_seed(0)
    ps = [torch.randint(16, 30000, (a.prompt_len,), generator=g).tolist() for _ in range(2)]
    sp = SamplingParams(max_tokens=1, temperature=0, ignore_eos=True)

    def pooled():
        eng.bm.reset_prefix_cache()
        reqs = [eng.add_request(p, copy.deepcopy(sp), pooling=True) for p in ps]
        pages = None
        while eng.has_work():
            eng.step()
            pages = pages or [list(r.pages) for r in reqs]
        return torch.stack([r.pooled for r in reqs]), pages

    if a.trace:
        # which op of the first two layers is the first whose output differs between two runs?
        from vllm_production_stack_amd import ops

        log: list = []
        names = ["rms_norm", "fused_add_rms_norm", "qkv_linear", "attention_block", "linear_add", "gate_up_silu", "linear"]
        orig = {n: getattr(ops, n) for n in names if hasattr(ops, n)}

        def wrap(n, fn):
            def inner(*args, **kw):
                out = fn(*args, **kw)
                if len(log) < 60:
                    t = out if isinstance(out, torch.Tensor) else (out[0] if isinstance(out, tuple) else None)
                    if t is None and n == "linear_add":
                        t = args[2]  # added into the residual stream in place
                    log.append((n, t.detach().clone() if t is not None else None))
                return out
            return inner

        for n, fn in orig.items():
            setattr(ops, n, wrap(n, fn))
        runs = []
        for _ in range(1 + a.repeats):
            log.clear()
            pooled()
            runs.append(list(log))
        for n, fn in orig.items():
            setattr(ops, n, fn)
        report = []
        for r in runs[1:]:  # per run: the first op whose output differs from the first run's, and where
            hit = None
            for i, (n, t) in enumerate(runs[0]):
                if t is None or torch.equal(r[i][1], t):
                    continue
                d = (r[i][1].float() - t.float()).abs()
                rows = torch.nonzero(d.reshape(d.shape[0], -1).amax(dim=1) > 0).flatten()
                cols = torch.nonzero(d.reshape(d.shape[0], -1).amax(dim=0) > 0).flatten()
                hit = {"op_index": i, "op": n, "shape": list(t.shape), "max_abs_diff": float(d.max()),
                       "rows": [int(rows[0]), int(rows[-1]), int(rows.numel())],
                       "cols": [int(cols[0]), int(cols[-1]), int(cols.numel())]}
                break
            report.append(hit)
        print(json.dumps({"prefill_variant": a.prefill_variant, "first_differing_op_by_run": report}))
        return
    if a.trace_attention:
        # inside the attention of the second layer: the pages after the cache write, the kernel's output
        from vllm_production_stack_amd import ops

        snaps: list = []
        o_write, o_pre = ops.kv_cache_write, ops.paged_prefill

        def write(qkv, positions, slots, cos_sin, pool, layer, *rest, **kw):
            o_write(qkv, positions, slots, cos_sin, pool, layer, *rest, **kw)
            if layer == 1:
                pages = torch.unique(slots[slots >= 0] // 64)
                snaps.append(("pages_after_write", pool[pages, layer].clone(), qkv.clone()))

        def pre(q, out, pool, layer, *rest, **kw):
            o_pre(q, out, pool, layer, *rest, **kw)
            if layer == 1:
                snaps.append(("prefill_out", out.clone(), q.clone()))

        ops.kv_cache_write, ops.paged_prefill = write, pre
        runs = []
        for _ in range(3):
            snaps.clear()
            pooled()
            runs.append(list(snaps))
        ops.kv_cache_write, ops.paged_prefill = o_write, o_pre
        rep = []
        for i, (n, t, src) in enumerate(runs[0]):
            for r in runs[1:]:
                d = (r[i][1].float() - t.float()).abs()
                rows = torch.nonzero(d.reshape(d.shape[0], -1).amax(dim=1) > 0).flatten().tolist()
                rep.append({"what": n, "shape": list(t.shape), "max_abs_diff": float(d.max()),
                            "input_identical": bool(torch.equal(r[i][2], src)),
                            "rows_or_pages_differing": len(rows), "first": rows[:12], "last": rows[-6:]})
        print(json.dumps({"attention_layer1": rep}))
        return
    first, pages0 = pooled()
    diffs, same_pages = [], []
    for _ in range(a.repeats):
        v, pages = pooled()
        diffs.append(float((v - first).abs().max()))
        same_pages.append(pages == pages0)
    sp2 = SamplingParams(max_tokens=24, temperature=0, ignore_eos=True)
    eng.bm.reset_prefix_cache()
    t0 = eng.generate(ps, copy.deepcopy(sp2))
    first_diff = []
    for _ in range(a.repeats):
        eng.bm.reset_prefix_cache()
        t = eng.generate(ps, copy.deepcopy(sp2))
        first_diff.append([next((i for i, (x, y) in enumerate(zip(u, w)) if x != y), -1) for u, w in zip(t, t0)])
    print(json.dumps({"prompt_len": a.prompt_len, "eager": a.eager,
                      "kv_layer_groups": len(getattr(eng.runner.pool, "groups", [0])),
                      "prefill_hidden_max_abs_diff_by_repeat": diffs, "prefill_same_pages_by_repeat": same_pages,
                      "hidden_scale": float(first.abs().mean()),
                      "decode_first_differing_token_by_repeat": first_diff}))


if __name__ == "__main__":
    main()
