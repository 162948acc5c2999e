"""LoRA end to end on the CPU path: a PEFT-style adapter directory is loaded into a slot and
requests that name it must generate exactly what a model with the merged weights
W + (alpha / r) B A generates; requests without an adapter keep the base output; two adapters
in one batch do not leak into each other."""

import copy
import json

import torch
from safetensors.torch import save_file

from vllm_production_stack_amd.engine.config import EngineConfig
from vllm_production_stack_amd.engine.llm_engine import LLMEngine
from vllm_production_stack_amd.engine.lora import LoRAManager
from vllm_production_stack_amd.engine.request import SamplingParams

SP = SamplingParams(max_tokens=8, temperature=0, ignore_eos=True)


def _engine():
    cfg = EngineConfig(model="tiny-llama-gqa", device="cpu", dtype="float32", max_model_len=512,
                       max_num_seqs=8, max_num_batched_tokens=256, num_gpu_blocks_override=32,
                       enable_lora=True, max_loras=2, max_lora_rank=8)
    eng = LLMEngine(cfg)
    eng.lora_manager = LoRAManager(eng.runner.model, cfg.max_loras, cfg.max_lora_rank)
    eng.runner.model.lora = eng.lora_manager
    return eng


def _write_adapter(path, model, seed, r=4, alpha=8.0):
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
            tensors[f"{base}.lora_A.weight"] = torch.randn(r, i, generator=g) * 0.3
            tensors[f"{base}.lora_B.weight"] = torch.randn(o, r, generator=g) * 0.3
    path.mkdir(parents=True, exist_ok=True)
    save_file(tensors, str(path / "adapter_model.safetensors"))
    (path / "adapter_config.json").write_text(json.dumps({"r": r, "lora_alpha": alpha,
                                                          "target_modules": sorted(dims)}))
    return tensors, alpha / r


def _merge(model, tensors, scale):
    d = model.head_dim
    q, kv = model.n_q * d, model.n_kv * d
    inter = model.cfg.intermediate_size
    for layer, lw in enumerate(model.layers):
        def delta(mod):
            base = f"base_model.model.model.layers.{layer}.{mod}"
            return scale * tensors[f"{base}.lora_B.weight"] @ tensors[f"{base}.lora_A.weight"]

        lw.wqkv[:q] += delta("self_attn.q_proj")
        lw.wqkv[q:q + kv] += delta("self_attn.k_proj")
        lw.wqkv[q + kv:] += delta("self_attn.v_proj")
        lw.wo += delta("self_attn.o_proj")
        lw.w_gate_up[:inter] += delta("mlp.gate_proj")
        lw.w_gate_up[inter:] += delta("mlp.up_proj")
        lw.w_down += delta("mlp.down_proj")


def _gen(eng, prompts, loras):
    reqs = [eng.add_request(p, copy.deepcopy(SP), lora_name=name) for p, name in zip(prompts, loras)]
    while eng.has_work():
        eng.step()
    return [r.output_ids for r in reqs]


def test_adapter_equals_merged_weights_and_adapters_do_not_mix(tmp_path):
    g = torch.Generator().manual_seed(3)
    prompts = [torch.randint(16, 500, (n,), generator=g).tolist() for n in (70, 33, 90)]
    eng = _engine()
    base_out = _gen(eng, prompts, [None] * 3)
    t1, s1 = _write_adapter(tmp_path / "a1", eng.runner.model, seed=1)
    t2, s2 = _write_adapter(tmp_path / "a2", eng.runner.model, seed=2)
    eng.lora_manager.load("a1", str(tmp_path / "a1"))
    eng.lora_manager.load("a2", str(tmp_path / "a2"))
    assert eng.lora_manager.list_adapters() == [("a1", str(tmp_path / "a1")), ("a2", str(tmp_path / "a2"))]

    want = {}
    for name, (tensors, scale) in {"a1": (t1, s1), "a2": (t2, s2)}.items():
        merged = _engine()
        _merge(merged.runner.model, tensors, scale)
        want[name] = _gen(merged, prompts, [None] * 3)
    assert want["a1"] != base_out and want["a2"] != want["a1"]

    # one mixed batch: adapter 1, no adapter, adapter 2 (prefill rows and decode rows interleave)
    mixed = _gen(eng, prompts, ["a1", None, "a2"])
    assert mixed[0] == want["a1"][0]
    assert mixed[1] == base_out[1]
    assert mixed[2] == want["a2"][2]

    # unloading frees the slot and zeroes its weights; a new adapter can take it
    eng.lora_manager.unload("a1")
    assert not eng.lora_manager.has("a1")
    eng.lora_manager.load("a1-again", str(tmp_path / "a1"))
    assert _gen(eng, prompts[:1], ["a1-again"])[0] == want["a1"][0]
