"""Throughput of the speech-to-text path (random-init Whisper-small shape, bf16): windows of
30 s audio per second for encoder-only and for encoder + N greedy decoder steps, by batch.

    python benchmarks/speech_micro.py [--model openai/whisper-small] [--tokens 64]
"""

import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

from vllm_production_stack_amd.models import whisper as W  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="openai/whisper-small")
    ap.add_argument("--tokens", type=int, default=64)
    ap.add_argument("--batches", default="1,8,32")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    cfg, _ = W.resolve_whisper(a.model)
    m = W.WhisperModel(cfg, torch.bfloat16, dev)
    m.init_random(0)
    # random weights rarely emit end-of-text early; forbid it so that every run does --tokens steps
    cfg.eos_token_id = cfg.vocab_size + 1
    rng = np.random.default_rng(0)
    for b in [int(x) for x in a.batches.split(",")]:
        waves = torch.from_numpy((0.1 * rng.standard_normal((b, W.N_SAMPLES))).astype(np.float32)).to(dev)
        prompts = [m.prompt_ids(None)] * b
        mel = W.log_mel(waves, cfg.num_mel_bins, m.filters)
        m.encode(mel)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            m.encode(W.log_mel(waves, cfg.num_mel_bins, m.filters))
        torch.cuda.synchronize()
        enc_ms = (time.perf_counter() - t0) / 3 * 1e3
        m.transcribe_windows(waves, prompts, max_new_tokens=4)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m.transcribe_windows(waves, prompts, max_new_tokens=a.tokens)
        torch.cuda.synchronize()
        full_ms = (time.perf_counter() - t0) * 1e3
        print(json.dumps({"model": a.model, "batch": b, "encoder_ms": round(enc_ms, 2),
                          "encoder_windows_per_s": round(b / enc_ms * 1e3, 1),
                          "transcribe_ms": round(full_ms, 1), "decode_tokens": a.tokens,
                          "audio_seconds_per_second": round(b * 30 / full_ms * 1e3, 1)}), flush=True)


if __name__ == "__main__":
    main()
