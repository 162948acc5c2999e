"""Cost of multi-adapter LoRA on one GPU (Llama-3-8B shape, synthetic rank-16 adapters on all
seven projections): cold prefill throughput and decode step time for batches without adapters,
with one adapter on every row, and with two adapters and plain rows mixed in one batch.

    python benchmarks/lora_micro.py [--ctx 4000] [--batch 40]
"""

import argparse
import json
import os
import sys
import tempfile
import time
from pathlib import Path

import torch
from safetensors.torch import save_file

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from vllm_production_stack_amd.engine.config import EngineConfig  # noqa: E402
from vllm_production_stack_amd.engine.llm_engine import LLMEngine  # noqa: E402
from vllm_production_stack_amd.engine.lora import LoRAManager  # noqa: E402
from vllm_production_stack_amd.engine.request import SamplingParams  # noqa: E402


def write_adapter(path: Path, model, seed: int, r: int = 16) -> None:
    g = torch.Generator().manual_seed(seed)
    cfg, d = model.cfg, model.head_dim
    dims = {"self_attn.q_proj": (cfg.hidden_size, model.n_q * d),
            "self_attn.k_proj": (cfg.hidden_size, model.n_kv * d),
            "self_attn.v_proj": (cfg.hidden_size, model.n_kv * d),
            "self_attn.o_proj": (model.n_q * d, cfg.hidden_size),
            "mlp.gate_proj": (cfg.hidden_size, cfg.intermediate_size),
            "mlp.up_proj": (cfg.hidden_size, cfg.intermediate_size),
            "mlp.down_proj": (cfg.intermediate_size, cfg.hidden_size)}
    tensors = {}
    for layer in range(cfg.num_layers):
        for mod, (i, o) in dims.items():
            base = f"base_model.model.model.layers.{layer}.{mod}"
            tensors[f"{base}.lora_A.weight"] = (torch.randn(r, i, generator=g) * 0.01).to(torch.bfloat16)
            tensors[f"{base}.lora_B.weight"] = (torch.randn(o, r, generator=g) * 0.01).to(torch.bfloat16)
    path.mkdir(parents=True, exist_ok=True)
    save_file(tensors, str(path / "adapter_model.safetensors"))
    (path / "adapter_config.json").write_text(json.dumps({"r": r, "lora_alpha": 2 * r}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="llama-3-8b")
    ap.add_argument("--ctx", type=int, default=4000)
    ap.add_argument("--batch", type=int, default=40)
    ap.add_argument("--decode-steps", type=int, default=40)
    a = ap.parse_args()
    cfg = EngineConfig(model=a.model, max_model_len=16000, max_num_seqs=64, max_num_batched_tokens=8192,
                       enable_lora=True, max_loras=4, max_lora_rank=16, enable_prefix_caching=False)
    eng = LLMEngine(cfg)
    eng.lora_manager = LoRAManager(eng.runner.model, cfg.max_loras, cfg.max_lora_rank)
    eng.runner.model.lora = eng.lora_manager
    tmp = Path(tempfile.mkdtemp(prefix="vpsa_lora_"))
    for i in (1, 2):
        write_adapter(tmp / f"a{i}", eng.runner.model, i)
        t0 = time.time()
        eng.lora_manager.load(f"a{i}", str(tmp / f"a{i}"))
        print(json.dumps({"case": "load_adapter", "name": f"a{i}", "seconds": round(time.time() - t0, 2)}),
              flush=True)
    vocab = eng.model_cfg.vocab_size
    g = torch.Generator().manual_seed(0)
    eng.generate([torch.randint(16, vocab, (300,), generator=g).tolist()],
                 SamplingParams(max_tokens=4, temperature=0, ignore_eos=True))
    prefill_steps = (a.batch * a.ctx + 8191) // 8192 + a.batch
    for case, names in (("no_adapter", [None]), ("one_adapter_all_rows", ["a1"]),
                        ("two_adapters_and_plain_rows_mixed", ["a1", None, "a2"])):
        sp = SamplingParams(max_tokens=a.decode_steps + 8 + prefill_steps, temperature=0, ignore_eos=True)
        reqs = []
        for i in range(a.batch):
            ids = torch.randint(16, vocab, (a.ctx,), generator=g).tolist()
            reqs.append(eng.add_request(ids, SamplingParams(**sp.__dict__), lora_name=names[i % len(names)]))
        torch.cuda.synchronize()
        t0 = time.time()
        while any(len(r.output_ids) == 0 for r in reqs):
            eng.step()
        torch.cuda.synchronize()
        t_prefill = time.time() - t0
        for _ in range(3):
            eng.step()
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(a.decode_steps):
            eng.step()
        torch.cuda.synchronize()
        t_dec = (time.time() - t0) / a.decode_steps
        for r in reqs:
            eng.abort(r.request_id)
        while eng.has_work():
            eng.step()
        print(json.dumps({"case": case, "batch": a.batch, "ctx": a.ctx,
                          "prefill_tok_per_s": round(a.batch * a.ctx / t_prefill, 1),
                          "decode_ms_per_step": round(t_dec * 1e3, 3)}), flush=True)


if __name__ == "__main__":
    main()
