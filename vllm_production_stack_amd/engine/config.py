"""Engine configuration: model architecture presets + the serving flags.

The CLI surface mirrors what the reference's Helm chart renders for an engine container
(/root/reference/helm/templates/deployment-vllm-multi.yaml:108-199) and the LMCACHE_* env
contract (same file :256-344), so an unmodified values.yaml drives this engine.
"""

from __future__ import annotations

import json
import os
from dataclasses import dataclass, field
from pathlib import Path
from typing import Any

import torch


@dataclass
class ModelConfig:
    name: str = "tiny-llama"
    architecture: str = "llama"  # llama | mixtral | opt | gpt2
    vocab_size: int = 512
    hidden_size: int = 256
    intermediate_size: int = 512
    num_layers: int = 2
    num_heads: int = 2
    num_kv_heads: int = 1
    head_dim: int = 128
    rope_theta: float = 10000.0
    rope_scaling: dict | None = None
    norm_eps: float = 1e-5
    max_position_embeddings: int = 8192
    tie_word_embeddings: bool = False
    num_experts: int = 0
    num_experts_per_tok: int = 0
    qkv_bias: bool = False  # Qwen2 family, Llama-style checkpoints with attention_bias
    bos_token_id: int = 1
    eos_token_id: int = 2
    extra_eos_token_ids: tuple = ()  # config.json may list several end-of-sequence ids
    # OPT / GPT-2 specifics (learned positions, LayerNorm, biased projections)
    ffn_dim: int = 0
    pos_offset: int = 0
    activation: str = "relu"  # relu (OPT) | gelu_new (GPT-2)

    @property
    def q_size(self) -> int:
        return self.num_heads * self.head_dim

    @property
    def kv_size(self) -> int:
        return self.num_kv_heads * self.head_dim

    def prefill_flops(self) -> tuple[float, float]:
        """(GEMM FLOPs per new token, attention FLOPs per (new token x context token) pair)."""
        h, i = self.hidden_size, self.intermediate_size
        attn_w = h * (self.q_size + 2 * self.kv_size) + self.q_size * h
        mlp_w = 3 * h * i * max(1, self.num_experts_per_tok if self.num_experts else 1)
        return 2.0 * self.num_layers * (attn_w + mlp_w), 4.0 * self.num_layers * self.q_size

    def num_params(self) -> int:
        h, i, l_ = self.hidden_size, self.intermediate_size, self.num_layers
        attn = h * (self.q_size + 2 * self.kv_size) + self.q_size * h
        mlp = 3 * h * i * max(1, self.num_experts)
        emb = self.vocab_size * h * (1 if self.tie_word_embeddings else 2)
        return l_ * (attn + mlp) + emb


PRESETS: dict[str, ModelConfig] = {
    "llama-3-8b": ModelConfig(
        name="llama-3-8b", architecture="llama", vocab_size=128256, hidden_size=4096,
        intermediate_size=14336, num_layers=32, num_heads=32, num_kv_heads=8, head_dim=128,
        rope_theta=500000.0, norm_eps=1e-5, max_position_embeddings=131072,
        rope_scaling={"rope_type": "llama3", "factor": 8.0, "low_freq_factor": 1.0,
                      "high_freq_factor": 4.0, "original_max_position_embeddings": 8192},
        bos_token_id=128000, eos_token_id=128009),
    "mixtral-8x7b": ModelConfig(
        name="mixtral-8x7b", architecture="mixtral", vocab_size=32000, hidden_size=4096,
        intermediate_size=14336, num_layers=32, num_heads=32, num_kv_heads=8, head_dim=128,
        rope_theta=1000000.0, norm_eps=1e-5, max_position_embeddings=32768, num_experts=8,
        num_experts_per_tok=2, bos_token_id=1, eos_token_id=2),
    "opt-125m": ModelConfig(
        name="opt-125m", architecture="opt", vocab_size=50272, hidden_size=768,
        intermediate_size=3072, num_layers=12, num_heads=12, num_kv_heads=12, head_dim=64,
        max_position_embeddings=2048, ffn_dim=3072, pos_offset=2, tie_word_embeddings=True,
        bos_token_id=2, eos_token_id=2),
    # the other models the reference's tutorials and CI values name
    "mistral-7b": ModelConfig(
        name="mistral-7b", architecture="llama", vocab_size=32000, hidden_size=4096,
        intermediate_size=14336, num_layers=32, num_heads=32, num_kv_heads=8, head_dim=128,
        rope_theta=1000000.0, norm_eps=1e-5, max_position_embeddings=32768, bos_token_id=1, eos_token_id=2),
    "llama-3-70b": ModelConfig(
        name="llama-3-70b", architecture="llama", vocab_size=128256, hidden_size=8192,
        intermediate_size=28672, num_layers=80, num_heads=64, num_kv_heads=8, head_dim=128,
        rope_theta=500000.0, norm_eps=1e-5, max_position_embeddings=131072,
        rope_scaling={"rope_type": "llama3", "factor": 8.0, "low_freq_factor": 1.0,
                      "high_freq_factor": 4.0, "original_max_position_embeddings": 8192},
        bos_token_id=128000, eos_token_id=128009),
    "llama-3.2-1b": ModelConfig(
        name="llama-3.2-1b", architecture="llama", vocab_size=128256, hidden_size=2048,
        intermediate_size=8192, num_layers=16, num_heads=32, num_kv_heads=8, head_dim=64,
        rope_theta=500000.0, norm_eps=1e-5, max_position_embeddings=131072, tie_word_embeddings=True,
        rope_scaling={"rope_type": "llama3", "factor": 32.0, "low_freq_factor": 1.0,
                      "high_freq_factor": 4.0, "original_max_position_embeddings": 8192},
        bos_token_id=128000, eos_token_id=128009),
    "smollm2-135m": ModelConfig(
        name="smollm2-135m", architecture="llama", vocab_size=49152, hidden_size=576,
        intermediate_size=1536, num_layers=30, num_heads=9, num_kv_heads=3, head_dim=64,
        rope_theta=100000.0, norm_eps=1e-5, max_position_embeddings=8192, tie_word_embeddings=True,
        bos_token_id=1, eos_token_id=2),
    "distilgpt2": ModelConfig(
        name="distilgpt2", architecture="gpt2", vocab_size=50257, hidden_size=768,
        intermediate_size=3072, num_layers=6, num_heads=12, num_kv_heads=12, head_dim=64,
        max_position_embeddings=1024, ffn_dim=3072, pos_offset=0, tie_word_embeddings=True,
        activation="gelu_new", bos_token_id=50256, eos_token_id=50256),
    "gpt2": ModelConfig(
        name="gpt2", architecture="gpt2", vocab_size=50257, hidden_size=768,
        intermediate_size=3072, num_layers=12, num_heads=12, num_kv_heads=12, head_dim=64,
        max_position_embeddings=1024, ffn_dim=3072, pos_offset=0, tie_word_embeddings=True,
        activation="gelu_new", bos_token_id=50256, eos_token_id=50256),
    "tiny-gpt2": ModelConfig(name="tiny-gpt2", architecture="gpt2", vocab_size=512, hidden_size=128,
                             intermediate_size=256, ffn_dim=256, num_layers=2, num_heads=4,
                             num_kv_heads=4, head_dim=32, pos_offset=0, tie_word_embeddings=True,
                             activation="gelu_new", max_position_embeddings=2048),
    "tiny-llama": ModelConfig(name="tiny-llama"),
    "tiny-llama-gqa": ModelConfig(name="tiny-llama-gqa", num_heads=8, num_kv_heads=2,
                                  hidden_size=512, intermediate_size=1024, num_layers=3),
    # Qwen2 family: biased q/k/v projections, GQA group of 7 on the 7B model
    "qwen2.5-7b": ModelConfig(
        name="qwen2.5-7b", architecture="llama", vocab_size=152064, hidden_size=3584,
        intermediate_size=18944, num_layers=28, num_heads=28, num_kv_heads=4, head_dim=128,
        rope_theta=1000000.0, norm_eps=1e-6, max_position_embeddings=32768, qkv_bias=True,
        bos_token_id=151643, eos_token_id=151645),
    "tiny-qwen2": ModelConfig(name="tiny-qwen2", num_heads=7, num_kv_heads=1, hidden_size=512,
                              intermediate_size=1024, num_layers=3, qkv_bias=True, norm_eps=1e-6),
    # three query heads per KV head (Llama-3.2-3B's ratio), 128-channel heads
    "tiny-llama-gqa3": ModelConfig(name="tiny-llama-gqa3", num_heads=6, num_kv_heads=2, hidden_size=768,
                                   intermediate_size=1024, num_layers=3),
    "tiny-mixtral": ModelConfig(name="tiny-mixtral", architecture="mixtral", num_heads=4,
                                num_kv_heads=2, hidden_size=256, intermediate_size=256,
                                num_experts=4, num_experts_per_tok=2),
    # expert widths the matrix-core decode kernels tile (whole and halved)
    "tiny-mixtral-wide": ModelConfig(name="tiny-mixtral-wide", architecture="mixtral", num_heads=4,
                                     num_kv_heads=2, hidden_size=256, intermediate_size=1024,
                                     num_experts=4, num_experts_per_tok=2),
    # heads of 64 channels (the geometry of OPT-125m, GPT-2, Llama-3.2-1B, Qwen2-0.5B)
    "tiny-opt-hd64": ModelConfig(name="tiny-opt-hd64", architecture="opt", vocab_size=512, hidden_size=256,
                                 intermediate_size=512, ffn_dim=512, num_layers=2, num_heads=4,
                                 num_kv_heads=4, head_dim=64, pos_offset=2, tie_word_embeddings=True,
                                 max_position_embeddings=2048),
    "tiny-llama-hd64": ModelConfig(name="tiny-llama-hd64", num_heads=8, num_kv_heads=2, head_dim=64,
                                   hidden_size=512, intermediate_size=1024, num_layers=3),
    "tiny-opt": ModelConfig(name="tiny-opt", architecture="opt", vocab_size=512, hidden_size=128,
                            intermediate_size=256, ffn_dim=256, num_layers=2, num_heads=4,
                            num_kv_heads=4, head_dim=32, pos_offset=2, tie_word_embeddings=True,
                            max_position_embeddings=2048),
}

_ALIASES = {
    "meta-llama/meta-llama-3-8b": "llama-3-8b",
    "meta-llama/meta-llama-3-8b-instruct": "llama-3-8b",
    "meta-llama/llama-3.1-8b": "llama-3-8b",
    "meta-llama/llama-3.1-8b-instruct": "llama-3-8b",
    "mistralai/mixtral-8x7b-v0.1": "mixtral-8x7b",
    "mistralai/mixtral-8x7b-instruct-v0.1": "mixtral-8x7b",
    "facebook/opt-125m": "opt-125m",
    "qwen/qwen2.5-7b-instruct": "qwen2.5-7b",
    "qwen/qwen2.5-7b": "qwen2.5-7b",
    "qwen/qwen2-7b-instruct": "qwen2.5-7b",
    "mistralai/mistral-7b-instruct-v0.2": "mistral-7b",
    "mistralai/mistral-7b-v0.1": "mistral-7b",
    "meta-llama/llama-3.3-70b-instruct": "llama-3-70b",
    "meta-llama/llama-3.1-70b-instruct": "llama-3-70b",
    "meta-llama/meta-llama-3-70b-instruct": "llama-3-70b",
    "meta-llama/llama-3.2-1b-instruct": "llama-3.2-1b",
    "meta-llama/llama-3.2-1b": "llama-3.2-1b",
    "huggingfacetb/smollm2-135m-instruct": "smollm2-135m",
    "distilbert/distilgpt2": "distilgpt2",
    "openai-community/gpt2": "gpt2",
}


def _from_hf_config(cfg: dict[str, Any], name: str) -> ModelConfig:
    arch = (cfg.get("architectures") or [""])[0].lower()
    mt = cfg.get("model_type", "").lower()
    if "opt" in (arch + mt):
        h = cfg["hidden_size"]
        return ModelConfig(
            name=name, architecture="opt", vocab_size=cfg["vocab_size"], hidden_size=h,
            intermediate_size=cfg["ffn_dim"], ffn_dim=cfg["ffn_dim"],
            num_layers=cfg["num_hidden_layers"], num_heads=cfg["num_attention_heads"],
            num_kv_heads=cfg["num_attention_heads"], head_dim=h // cfg["num_attention_heads"],
            max_position_embeddings=cfg.get("max_position_embeddings", 2048), pos_offset=2,
            tie_word_embeddings=True, bos_token_id=cfg.get("bos_token_id", 2),
            eos_token_id=cfg.get("eos_token_id", 2))
    if "gpt2" in (arch + mt):
        h = cfg["n_embd"]
        inner = cfg.get("n_inner") or 4 * h
        return ModelConfig(
            name=name, architecture="gpt2", vocab_size=cfg["vocab_size"], hidden_size=h,
            intermediate_size=inner, ffn_dim=inner, num_layers=cfg["n_layer"], num_heads=cfg["n_head"],
            num_kv_heads=cfg["n_head"], head_dim=h // cfg["n_head"],
            max_position_embeddings=cfg.get("n_positions", 1024), pos_offset=0, tie_word_embeddings=True,
            activation="gelu_new", norm_eps=cfg.get("layer_norm_epsilon", 1e-5),
            bos_token_id=cfg.get("bos_token_id", 50256), eos_token_id=cfg.get("eos_token_id", 50256))
    is_moe = "mixtral" in (arch + mt)
    if "qwen2" not in (arch + mt) and (cfg.get("attention_bias") or cfg.get("mlp_bias")):
        raise ValueError(f"{name}: projections with biases are only supported for the Qwen2 family")
    # rotary settings: top-level ``rope_theta`` / ``rope_scaling`` (checkpoints written by
    # transformers 4.x) or one ``rope_parameters`` object (5.x)
    rp = cfg.get("rope_parameters") or {}
    rope_theta = cfg.get("rope_theta", rp.get("rope_theta", 10000.0))
    rope_scaling = cfg.get("rope_scaling")
    if rope_scaling is None and rp.get("rope_type", "default") not in ("default", None):
        rope_scaling = {k: v for k, v in rp.items() if k != "rope_theta"}
    h, nh = cfg["hidden_size"], cfg["num_attention_heads"]
    eos = cfg.get("eos_token_id", 2)
    extra_eos: tuple = ()
    if isinstance(eos, list):
        extra_eos = tuple(int(x) for x in eos[1:])
        eos = eos[0]
    return ModelConfig(
        name=name, architecture="mixtral" if is_moe else "llama", vocab_size=cfg["vocab_size"],
        hidden_size=h, intermediate_size=cfg["intermediate_size"],
        num_layers=cfg["num_hidden_layers"], num_heads=nh,
        num_kv_heads=cfg.get("num_key_value_heads", nh), head_dim=cfg.get("head_dim") or h // nh,
        rope_theta=rope_theta, rope_scaling=rope_scaling,
        norm_eps=cfg.get("rms_norm_eps", 1e-5),
        max_position_embeddings=cfg.get("max_position_embeddings", 8192),
        tie_word_embeddings=cfg.get("tie_word_embeddings", False),
        num_experts=cfg.get("num_local_experts", 0),
        num_experts_per_tok=cfg.get("num_experts_per_tok", 0),
        qkv_bias="qwen2" in (arch + mt),
        bos_token_id=cfg.get("bos_token_id") or 1, eos_token_id=eos, extra_eos_token_ids=extra_eos)


def resolve_model(model: str) -> tuple[ModelConfig, Path | None]:
    """Model id or local directory -> (architecture, checkpoint dir or None for random init)."""
    p = Path(model)
    if p.is_dir() and (p / "config.json").exists():
        with open(p / "config.json") as f:
            return _from_hf_config(json.load(f), model), p
    hf_home = os.environ.get("HF_HOME")
    if hf_home:
        snap = Path(hf_home) / "hub" / ("models--" + model.replace("/", "--")) / "snapshots"
        if snap.is_dir():
            for d in sorted(snap.iterdir()):
                if (d / "config.json").exists():
                    with open(d / "config.json") as f:
                        return _from_hf_config(json.load(f), model), d
    key = model.lower()
    key = _ALIASES.get(key, key)
    base = key.split("/")[-1]
    for cand in (key, base):
        if cand in PRESETS:
            cfg = PRESETS[cand]
            return ModelConfig(**{**cfg.__dict__, "name": model}), None
    for preset in PRESETS:
        if preset in base:
            cfg = PRESETS[preset]
            return ModelConfig(**{**cfg.__dict__, "name": model}), None
    raise ValueError(f"unknown model {model!r}: not a checkpoint directory and not a known architecture")


_DTYPES = {"auto": torch.bfloat16, "bfloat16": torch.bfloat16, "bf16": torch.bfloat16,
           "float16": torch.float16, "half": torch.float16, "float32": torch.float32,
           "float": torch.float32}


@dataclass
class KVTransferConfig:
    """--kv-transfer-config JSON (kv_connector / kv_role / kv_connector_extra_config)."""
    kv_connector: str = ""
    kv_role: str = "kv_both"  # kv_both | kv_producer | kv_consumer
    extra: dict = field(default_factory=dict)

    @classmethod
    def from_json(cls, s: str | None) -> "KVTransferConfig | None":
        if not s:
            return None
        d = json.loads(s)
        return cls(d.get("kv_connector", ""), d.get("kv_role", "kv_both"),
                   d.get("kv_connector_extra_config", {}) or {})


@dataclass
class KVOffloadConfig:
    """Tiered KV settings; field names follow the LMCACHE_* env contract."""
    local_cpu: bool = False
    max_local_cpu_gb: float = 0.0
    max_local_disk_gb: float = 0.0
    local_disk_path: str = "/tmp/vpsa_kv_disk"
    remote_url: str | None = None  # lm://host:port
    remote_serde: str = "naive"  # naive | cachegen
    chunk_tokens: int = 256
    enable_controller: bool = False
    instance_id: str = ""
    controller_url: str | None = None
    worker_port: int = 0
    distributed_url: str | None = None
    enable_pd: bool = False  # LMCACHE_ENABLE_NIXL
    pd_role: str = ""  # sender | receiver
    pd_peer_host: str = ""
    pd_peer_port: int = 55555
    pd_buffer_bytes: int = 1 << 30
    pd_buffer_device: str = "hip"

    @classmethod
    def from_env(cls, env: dict[str, str] | None = None) -> "KVOffloadConfig":
        e = os.environ if env is None else env

        def flag(k: str) -> bool:
            return str(e.get(k, "")).strip().lower() in ("1", "true", "yes", "on")

        def num(k: str, d: float = 0.0) -> float:
            try:
                return float(e.get(k, d))
            except (TypeError, ValueError):
                return d

        return cls(
            local_cpu=flag("LMCACHE_LOCAL_CPU"),
            max_local_cpu_gb=num("LMCACHE_MAX_LOCAL_CPU_SIZE"),
            max_local_disk_gb=num("LMCACHE_MAX_LOCAL_DISK_SIZE"),
            local_disk_path=e.get("LMCACHE_LOCAL_DISK", "/tmp/vpsa_kv_disk").replace("file://", ""),
            remote_url=e.get("LMCACHE_REMOTE_URL") or None,
            remote_serde=(e.get("LMCACHE_REMOTE_SERDE") or "naive").lower(),
            chunk_tokens=int(num("LMCACHE_CHUNK_SIZE", 256)),
            enable_controller=flag("LMCACHE_ENABLE_CONTROLLER"),
            instance_id=e.get("LMCACHE_LMCACHE_INSTANCE_ID", ""),
            controller_url=e.get("LMCACHE_CONTROLLER_URL") or None,
            worker_port=int(num("LMCACHE_LMCACHE_WORKER_PORT", 0)),
            distributed_url=e.get("LMCACHE_DISTRIBUTED_URL") or None,
            enable_pd=flag("LMCACHE_ENABLE_NIXL"),
            pd_role=e.get("LMCACHE_NIXL_ROLE", ""),
            pd_peer_host=e.get("LMCACHE_NIXL_RECEIVER_HOST", e.get("LMCACHE_NIXL_PEER_HOST", "")),
            pd_peer_port=int(num("LMCACHE_NIXL_RECEIVER_PORT", num("LMCACHE_NIXL_PEER_PORT", 55555))),
            pd_buffer_bytes=int(num("LMCACHE_NIXL_BUFFER_SIZE", 1 << 30)),
            pd_buffer_device=e.get("LMCACHE_NIXL_BUFFER_DEVICE", "hip"),
        )

    @property
    def enabled(self) -> bool:
        return bool(self.local_cpu or self.max_local_disk_gb > 0 or self.remote_url)


@dataclass
class EngineConfig:
    model: str = "tiny-llama"
    served_model_name: str | None = None
    host: str = "0.0.0.0"
    port: int = 8000
    dtype: str = "auto"
    device: str = "auto"
    max_model_len: int | None = None
    max_num_seqs: int = 256
    max_num_batched_tokens: int = 8192
    gpu_memory_utilization: float = 0.9
    num_gpu_blocks_override: int | None = None
    enable_prefix_caching: bool = True
    enable_chunked_prefill: bool = True
    tensor_parallel_size: int = 1
    pipeline_parallel_size: int = 1
    enable_expert_parallel: bool = False  # MoE layers: whole experts per tensor-parallel rank
    # decode steps launched back to back before the host looks at the sampled tokens (greedy
    # decode-only batches of a single-rank replica with nothing waiting; 1 = every step)
    num_scheduler_steps: int = 1
    # KV pool allocated as this many pools of consecutive layers (a decode step of one layer then
    # walks a shorter address range); 0 = 4 for a large GPU pool on a replica without lower KV tiers
    # and without page transfers to or from other replicas, else 1.  Lower tiers work with any value
    # (their copies are then per piece); replicas that exchange pages always keep whole pages.
    kv_layer_groups: int = 0
    distributed_executor_backend: str | None = None
    enable_lora: bool = False
    max_loras: int = 1
    max_lora_rank: int = 16
    enable_sleep_mode: bool = False
    enable_auto_tool_choice: bool = False
    tool_call_parser: str | None = None
    chat_template: str | None = None
    kv_transfer: KVTransferConfig | None = None
    kv_offload: KVOffloadConfig = field(default_factory=KVOffloadConfig)
    swap_space: float = 0.0
    enforce_eager: bool = False
    capture_graphs_at_start: bool = True  # HIP graphs for all decode batch buckets at start-up
    # second copy of the dense MLP matrices in matrix-core tile order, streamed by decode-only steps
    # (Llama-3-8B, 40 sequences: decode step 23.0 -> 22.75 ms for 11 GiB of HBM, so off by default):
    # "off", "on", "auto" = on when it costs at most a tenth of the card's memory
    packed_decode_weights: str = "off"
    # Latency-aware prefill admission: after the first prefill of a step, more waiting requests
    # join only while the estimated prefill time of the step stays under this many ms
    # (0 = fill the token budget).  The estimate prices GEMM and attention FLOPs at the rates
    # below (what the library GEMM and the prefill attention kernel sustain on one MI355X).
    prefill_step_ms: float = 30.0
    gemm_tflops_estimate: float = 1500.0
    attn_tflops_estimate: float = 900.0
    seed: int = 0
    api_key: str | None = None
    disable_log_requests: bool = False
    trust_remote_code: bool = False
    task: str = "generate"
    load_format: str = "auto"  # auto | dummy

    def torch_dtype(self) -> torch.dtype:
        return _DTYPES[self.dtype.lower()]

    def resolve_device(self) -> torch.device:
        if self.device == "auto":
            if torch.cuda.is_available():
                return torch.device("cuda", torch.cuda.current_device())
            return torch.device("cpu")
        dev = torch.device(self.device)
        if dev.type == "cuda" and dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        return dev

    @property
    def model_name(self) -> str:
        return self.served_model_name or self.model
