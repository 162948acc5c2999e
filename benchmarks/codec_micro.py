"""KV page codec on one GPU: quantiser + rANS entropy stage (all on the device) against the
quantiser + host range coder, for one Llama-3-8B page (64 tokens x 32 layers = 8 MiB of bf16).

    python benchmarks/codec_micro.py [--pages 8]

Prints one JSON line per stage; times are per page."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vllm_production_stack_amd import ops  # noqa: E402
from vllm_production_stack_amd.kv.serde import PageSerde  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pages", type=int, default=8)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    n_layers, n_kv, hd = 32, 8, 128
    cg = PageSerde(n_layers, n_kv, hd, torch.bfloat16, "cachegen")
    torch.manual_seed(0)
    # keys of neighbouring tokens are correlated: a shared per-channel component plus noise
    base = torch.randn(n_layers * 2 * n_kv, 1, hd, device=dev)
    pages = [((base + 0.3 * torch.randn(n_layers * 2 * n_kv, 64, hd, device=dev)).reshape(-1)).to(torch.bfloat16)
             for _ in range(a.pages)]
    raw = pages[0].numel() * 2
    lv = torch.tensor(cg.levels, dtype=torch.int32, device=dev)

    def sync():
        torch.cuda.synchronize()

    # device stages in isolation
    sym, scales = ops.cachegen_encode(pages[0], lv, cg.unit_elems, n_layers)
    sync()
    t0 = time.perf_counter()
    for p in pages:
        sym, scales = ops.cachegen_encode(p, lv, cg.unit_elems, n_layers)
    sync()
    t_q = (time.perf_counter() - t0) / len(pages)
    t0 = time.perf_counter()
    for _ in pages:
        freq, states, sizes, coded = ops.rans_encode(sym.view(-1), n_layers)
    sync()
    t_e = (time.perf_counter() - t0) / len(pages)
    t0 = time.perf_counter()
    for _ in pages:
        back = ops.rans_decode(freq, states, sizes, coded, sym.numel(), n_layers)
    sync()
    t_d = (time.perf_counter() - t0) / len(pages)
    assert torch.equal(back.view(-1), sym.view(-1))
    print(json.dumps({"case": "device_stages", "page_MB": round(raw / 1e6, 2),
                      "quantise_us": round(t_q * 1e6, 1), "rans_encode_us": round(t_e * 1e6, 1),
                      "rans_decode_us": round(t_d * 1e6, 1), "coded_MB": round(coded.numel() / 1e6, 3),
                      "symbols_per_s_encode": round(sym.numel() / t_e / 1e9, 2) * 1e9}), flush=True)

    # whole pages: GPU tensor -> blob bytes on the host -> GPU tensor
    for label, env in (("gpu_rans", "0"), ("host_range_coder", "1")):
        os.environ["VPSA_CACHEGEN_HOST_CODER"] = env
        blob = cg.encode(pages[0])
        cg.decode(blob, dev)
        sync()
        n = len(pages) if env == "0" else min(2, len(pages))
        t0 = time.perf_counter()
        blobs = [cg.encode(p) for p in pages[:n]]
        sync()
        t_enc = (time.perf_counter() - t0) / n
        t0 = time.perf_counter()
        for b in blobs:
            out = cg.decode(b, dev)
        sync()
        t_dec = (time.perf_counter() - t0) / n
        err = float((out.float() - pages[n - 1].float()).abs().mean() / pages[n - 1].float().abs().mean())
        print(json.dumps({"case": label, "encode_ms_per_page": round(t_enc * 1e3, 3),
                          "decode_ms_per_page": round(t_dec * 1e3, 3), "blob_MB": round(len(blobs[0]) / 1e6, 3),
                          "ratio_vs_bf16": round(raw / len(blobs[0]), 2), "mean_rel_err": round(err, 4),
                          "encode_GB_per_s_of_kv": round(raw / t_enc / 1e9, 2),
                          "decode_GB_per_s_of_kv": round(raw / t_dec / 1e9, 2)}), flush=True)
    os.environ.pop("VPSA_CACHEGEN_HOST_CODER", None)


if __name__ == "__main__":
    main()
