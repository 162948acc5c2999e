"""KV tier benchmark on one MI355X (Llama-3-8B geometry: one 64-token page = 8 MiB over 32 layers).

  copy engine   HBM -> pinned host and back through the side-stream copy engine (GB/s), page batches
  cachegen      GPU encode / decode throughput of the quantising codec (GB/s of bf16 KV)
  ttft          time to first token of a 21k-token prompt: cold recompute vs. reload from the host
                tier after HBM eviction vs. HBM prefix-cache hit   (the A/B the offload tier exists for)

    python benchmarks/kv_tier_micro.py [--cases copy,cachegen,ttft]
"""

import argparse
import copy
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from vllm_production_stack_amd import ops  # noqa: E402
from vllm_production_stack_amd.kv.tiers import HostKVPool  # noqa: E402

DEV = "cuda"


def copy_engine(n_pages=384, batch=64):
    pool = torch.empty((n_pages, 32, 2, 8, 64 * 128), device=DEV, dtype=torch.bfloat16)
    pool.normal_()
    ref = pool.clone()
    host = HostKVPool(pool, n_pages * pool.stride(0) * 2)
    page_bytes = host.page_bytes
    res = {"case": "copy_engine", "page_mib": page_bytes / 2**20, "pages": n_pages, "batch_pages": batch}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev = None
    for b in range(0, n_pages, batch):
        ev = host.store([(1000 + p, p) for p in range(b, min(n_pages, b + batch))])
    ev.wait()
    dt = time.perf_counter() - t0
    res["offload_gb_per_s"] = round(n_pages * page_bytes / dt / 1e9, 2)
    pool.zero_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b in range(0, n_pages, batch):
        ev = host.load([(1000 + p, p) for p in range(b, min(n_pages, b + batch))])
    ev.wait()
    dt = time.perf_counter() - t0
    res["reload_gb_per_s"] = round(n_pages * page_bytes / dt / 1e9, 2)
    torch.cuda.synchronize()
    res["bit_exact"] = bool(torch.equal(pool, ref))
    # overlap: reload while the compute stream runs GEMMs.  Every timing is taken warm (the first
    # pass of a loop also pays clock ramp-up and lazy initialisation) and repeated; the GEMM count
    # is sized so that GEMMs alone take about as long as the reload alone.
    a = torch.randn(8192, 8192, device=DEV, dtype=torch.bfloat16)

    def gemms(n):
        for _ in range(n):
            a @ a

    def reload():
        ev = None
        for b in range(0, n_pages, batch):
            ev = host.load([(1000 + p, p) for p in range(b, min(n_pages, b + batch))])
        return ev

    def wall(fn, reps=3):
        best = float("inf")
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ev = fn()
            torch.cuda.synchronize()
            if ev is not None:
                ev.wait()
            best = min(best, time.perf_counter() - t0)
        return best

    gemms(20)
    torch.cuda.synchronize()
    one_gemm = wall(lambda: gemms(20)) / 20
    reload_alone = wall(reload)
    n_gemm = max(4, int(round(reload_alone / one_gemm)))
    gemm_alone = wall(lambda: gemms(n_gemm))

    def both():
        ev = reload()
        gemms(n_gemm)
        return ev

    concurrent = wall(both)
    serial = gemm_alone + reload_alone
    res["reload_under_compute"] = {
        "gemms": n_gemm, "gemm_alone_s": round(gemm_alone, 4), "reload_alone_s": round(reload_alone, 4),
        "serial_sum_s": round(serial, 4), "concurrent_wall_s": round(concurrent, 4),
        # 1.0: the shorter of the two is hidden completely behind the longer; 0.0: they ran back to back
        "hidden_fraction_of_the_shorter": round((serial - concurrent) / min(gemm_alone, reload_alone), 3),
        "slowdown_vs_the_longer_alone_pct": round((concurrent / max(gemm_alone, reload_alone) - 1) * 100, 1)}
    host.close()
    return res


def cachegen(n_pages=32):
    pool = torch.empty((n_pages, 32, 2, 8, 64 * 128), device=DEV, dtype=torch.bfloat16)
    pool.normal_()
    staging = pool.view(-1)
    unit = 8 * 64 * 128
    levels = torch.tensor([127] * 10 + [63] * 11 + [31] * 11, dtype=torch.int32, device=DEV)
    nbytes = staging.numel() * 2

    def timed(fn, iters=5):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / iters

    sym, scales = ops.cachegen_encode(staging, levels, unit, 32)
    t_enc = timed(lambda: ops.cachegen_encode(staging, levels, unit, 32))
    t_dec = timed(lambda: ops.cachegen_decode(sym, scales, unit))
    dec = ops.cachegen_decode(sym, scales, unit)
    rel = ((dec.float() - staging.float()).abs().mean() / staging.float().abs().mean()).item()
    return {"case": "cachegen", "bf16_mib": nbytes / 2**20, "encode_gb_per_s": round(nbytes / t_enc / 1e9, 1),
            "decode_gb_per_s": round(nbytes / t_dec / 1e9, 1), "mean_rel_error": round(rel, 4),
            "bytes_before_entropy_coding": sym.numel() + scales.numel() * 4}


def ttft(ctx=21000):
    from vllm_production_stack_amd.engine.config import EngineConfig, KVOffloadConfig
    from vllm_production_stack_amd.engine.llm_engine import LLMEngine
    from vllm_production_stack_amd.engine.request import SamplingParams

    pages_per_seq = ctx // 64 + 2
    cfg = EngineConfig(model="llama-3-8b", max_model_len=32000, max_num_seqs=8, max_num_batched_tokens=8192,
                       num_gpu_blocks_override=2 * pages_per_seq + 8,
                       kv_offload=KVOffloadConfig(local_cpu=True, max_local_cpu_gb=16.0))
    eng = LLMEngine(cfg)
    g = torch.Generator().manual_seed(0)
    sp = SamplingParams(max_tokens=1, temperature=0, ignore_eos=True)

    def prompt():
        return torch.randint(16, 100000, (ctx,), generator=g).tolist()

    def first_token(p):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = eng.add_request(p, copy.deepcopy(sp))
        while eng.has_work():
            eng.step()
        torch.cuda.synchronize()
        return time.perf_counter() - t0, r

    eng.generate([prompt()[:300]], copy.deepcopy(sp))  # warm-up
    a, b, c = prompt(), prompt(), prompt()
    t_cold, _ = first_token(a)
    t_hbm, r_hbm = first_token(a)
    first_token(b)
    first_token(c)  # a's pages are now evicted from the 2-sequence HBM pool into the host tier
    t_host, r_host = first_token(a)
    return {"case": "ttft", "ctx": ctx, "cold_recompute_s": round(t_cold, 4),
            "hbm_prefix_hit_s": round(t_hbm, 4), "hbm_hit_tokens": r_hbm.num_cached_tokens,
            "host_tier_reload_s": round(t_host, 4), "host_tier_tokens": r_host.num_external_tokens,
            "hbm_tokens_on_reload": r_host.num_cached_tokens,
            "host_tier_loads": eng.connector.host.loads, "host_tier_stores": eng.connector.host.stores}


def remote(ctx=21000, serde="naive"):
    """Shared remote cache (lm://): raw put/get throughput of 8 MiB pages over loopback, and the
    TTFT of a 21k-token prompt whose KV is only in the cache server (the replica's own HBM
    prefix cache is dropped first) against recomputing it."""
    import subprocess
    import sys as _sys

    from vllm_production_stack_amd.engine.config import EngineConfig, KVOffloadConfig
    from vllm_production_stack_amd.engine.llm_engine import LLMEngine
    from vllm_production_stack_amd.engine.request import SamplingParams
    from vllm_production_stack_amd.kv.remote import RemoteKVClient

    port = 18455
    srv = subprocess.Popen([_sys.executable, "-m", "vllm_production_stack_amd.cache_server", "127.0.0.1",
                            str(port), "--max-size-gb", "24"])
    url = f"lm://127.0.0.1:{port}"
    try:
        cli = RemoteKVClient(url)
        for _ in range(100):
            try:
                cli.stats()
                break
            except OSError:
                time.sleep(0.1)
        blob = bytes(bytearray(os.urandom(1 << 20)) * 8)
        n = 64
        t0 = time.perf_counter()
        for i in range(n):
            cli.put(b"raw-%d" % i, blob)
        t_put = time.perf_counter() - t0
        t0 = time.perf_counter()
        for i in range(n):
            assert len(cli.get(b"raw-%d" % i)) == len(blob)
        t_get = time.perf_counter() - t0
        out = {"case": "remote", "serde": serde, "raw_put_gb_per_s": round(n * len(blob) / t_put / 1e9, 2),
               "raw_get_gb_per_s": round(n * len(blob) / t_get / 1e9, 2)}

        pages_per_seq = ctx // 64 + 2
        cfg = EngineConfig(model="llama-3-8b", max_model_len=32000, max_num_seqs=8, max_num_batched_tokens=8192,
                           num_gpu_blocks_override=2 * pages_per_seq + 8,
                           kv_offload=KVOffloadConfig(remote_url=url, remote_serde=serde))
        eng = LLMEngine(cfg)
        g = torch.Generator().manual_seed(0)
        sp = SamplingParams(max_tokens=1, temperature=0, ignore_eos=True)
        eng.generate([torch.randint(16, 100000, (300,), generator=g).tolist()], copy.deepcopy(sp))
        a = torch.randint(16, 100000, (ctx,), generator=g).tolist()

        def first_token(p):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r = eng.add_request(p, copy.deepcopy(sp))
            while eng.has_work():
                eng.step()
            torch.cuda.synchronize()
            return time.perf_counter() - t0, r

        t_cold, _ = first_token(a)
        t0 = time.perf_counter()
        want = ctx // 64
        while cli.stats().get("keys", 0) < n + want - 1 and time.perf_counter() - t0 < 120:
            time.sleep(0.05)  # background uploads of the fresh pages
        out["upload_drain_s"] = round(time.perf_counter() - t0, 2)
        eng.bm.reset_prefix_cache()
        t_remote, r = first_token(a)
        out.update({"ctx": ctx, "cold_recompute_s": round(t_cold, 4), "remote_reload_s": round(t_remote, 4),
                    "remote_tokens": r.num_external_tokens, "server": cli.stats()})
        eng.shutdown()
        return out
    finally:
        srv.terminate()
        srv.wait(timeout=10)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", default="copy,cachegen,ttft")
    a = ap.parse_args()
    for c in a.cases.split(","):
        out = {"copy": copy_engine, "cachegen": cachegen, "ttft": ttft, "remote": remote,
               "remote_cachegen": lambda: remote(serde="cachegen")}[c]()
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
