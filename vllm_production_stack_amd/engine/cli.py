"""``vpsa-engine serve <model> [flags]`` - the engine replica process.

Accepts the flag set the reference's Helm chart and operator render for an engine
container (/root/reference/helm/templates/deployment-vllm-multi.yaml:108-199,
/root/reference/operator/internal/controller/vllmruntime_controller.go:228-286), including
the underscore spellings the chart uses, so the same values.yaml can launch this engine.
Flags this engine has no use for are accepted and ignored with a warning.
"""

from __future__ import annotations

import argparse
import asyncio
import logging
import os
import signal
import sys

from .config import EngineConfig, KVOffloadConfig, KVTransferConfig


def _bool_pair(ap: argparse.ArgumentParser, name: str, default: bool, help_: str) -> None:
    dest = name.replace("-", "_")
    ap.add_argument(f"--{name}", dest=dest, action="store_true", default=default, help=help_)
    ap.add_argument(f"--no-{name}", dest=dest, action="store_false")


def build_parser() -> argparse.ArgumentParser:
    ap = argparse.ArgumentParser(prog="vpsa-engine")
    sub = ap.add_subparsers(dest="cmd", required=True)
    s = sub.add_parser("serve", help="start an OpenAI-compatible engine replica")
    s.add_argument("model")
    s.add_argument("--host", default="0.0.0.0")
    s.add_argument("--port", type=int, default=8000)
    s.add_argument("--served-model-name", default=None)
    s.add_argument("--dtype", default="auto")
    s.add_argument("--device", default="auto")
    s.add_argument("--max-model-len", "--max_model_len", type=int, default=None)
    s.add_argument("--max-num-seqs", "--max_num_seqs", type=int, default=256)
    s.add_argument("--max-num-batched-tokens", "--max_num_batched_tokens", type=int, default=8192)
    s.add_argument("--gpu-memory-utilization", "--gpu_memory_utilization", type=float, default=0.9)
    s.add_argument("--num-gpu-blocks-override", type=int, default=None)
    s.add_argument("--tensor-parallel-size", "--tensor_parallel_size", "-tp", type=int, default=1)
    s.add_argument("--pipeline-parallel-size", "--pipeline_parallel_size", "-pp", type=int, default=1)
    s.add_argument("--enable-expert-parallel", "--enable_expert_parallel", action="store_true",
                   help="mixture-of-experts layers: every tensor-parallel rank holds whole experts "
                        "instead of a slice of each")
    s.add_argument("--num-scheduler-steps", "--num_scheduler_steps", type=int, default=1,
                   help="decode steps launched back to back before the host reads the sampled tokens "
                        "(greedy decode-only batches with nothing waiting; tokens then arrive in groups)")
    s.add_argument("--kv-layer-groups", "--kv_layer_groups", type=int, default=0,
                   help="allocate the KV pool as this many pools of consecutive layers (faster decode reads "
                        "on large pools); 0 = choose: 4 for a large GPU pool without KV tiers or transfers, else 1")
    s.add_argument("--distributed-executor-backend", default=None)
    _bool_pair(s, "enable-prefix-caching", True, "hash-chained prefix cache over KV pages")
    _bool_pair(s, "enable-chunked-prefill", True, "split long prompts across steps")
    s.add_argument("--enable-lora", action="store_true")
    s.add_argument("--max-loras", "--max_loras", type=int, default=1)
    s.add_argument("--max-lora-rank", type=int, default=64)
    s.add_argument("--enable-sleep-mode", action="store_true")
    s.add_argument("--enable-auto-tool-choice", action="store_true")
    s.add_argument("--tool-call-parser", default=None)
    s.add_argument("--chat-template", default=None)
    s.add_argument("--kv-transfer-config", default=None)
    s.add_argument("--swap-space", type=float, default=0.0)
    s.add_argument("--enforce-eager", action="store_true")
    s.add_argument("--packed-decode-weights", choices=("auto", "on", "off"), default="off",
                   help="second copy of the dense MLP matrices in matrix-core tile order for decode steps "
                        "(auto: when it costs at most a tenth of the card's memory)")
    s.add_argument("--prefill-step-ms", type=float, default=30.0,
                   help="latency-aware prefill admission: estimated prefill time per step (0 = token budget only)")
    s.add_argument("--seed", type=int, default=0)
    s.add_argument("--api-key", default=None)
    s.add_argument("--disable-log-requests", action="store_true")
    s.add_argument("--trust-remote-code", action="store_true")
    s.add_argument("--task", default="generate")
    s.add_argument("--load-format", default="auto")
    s.add_argument("--log-level", default="info")
    return ap


def config_from_args(args: argparse.Namespace) -> EngineConfig:
    return EngineConfig(
        model=args.model, served_model_name=args.served_model_name, host=args.host, port=args.port,
        dtype=args.dtype, device=args.device, max_model_len=args.max_model_len,
        max_num_seqs=args.max_num_seqs, max_num_batched_tokens=args.max_num_batched_tokens,
        gpu_memory_utilization=args.gpu_memory_utilization,
        num_gpu_blocks_override=args.num_gpu_blocks_override,
        enable_prefix_caching=args.enable_prefix_caching,
        enable_chunked_prefill=args.enable_chunked_prefill,
        tensor_parallel_size=args.tensor_parallel_size,
        pipeline_parallel_size=args.pipeline_parallel_size,
        enable_expert_parallel=args.enable_expert_parallel,
        num_scheduler_steps=max(1, args.num_scheduler_steps),
        kv_layer_groups=max(0, args.kv_layer_groups),
        distributed_executor_backend=args.distributed_executor_backend,
        enable_lora=args.enable_lora, max_loras=args.max_loras, max_lora_rank=args.max_lora_rank,
        enable_sleep_mode=args.enable_sleep_mode,
        enable_auto_tool_choice=args.enable_auto_tool_choice, tool_call_parser=args.tool_call_parser,
        chat_template=args.chat_template,
        kv_transfer=KVTransferConfig.from_json(args.kv_transfer_config),
        kv_offload=KVOffloadConfig.from_env(), swap_space=args.swap_space,
        enforce_eager=args.enforce_eager, seed=args.seed, prefill_step_ms=args.prefill_step_ms,
        packed_decode_weights=args.packed_decode_weights,
        api_key=args.api_key or os.environ.get("VLLM_API_KEY") or None,
        disable_log_requests=args.disable_log_requests, trust_remote_code=args.trust_remote_code,
        task=args.task, load_format=args.load_format)


async def run_server(cfg: EngineConfig, ready: "asyncio.Event | None" = None) -> None:
    from .async_engine import AsyncLLMEngine
    from .llm_engine import LLMEngine
    from .server import serve

    from ..models.whisper import is_whisper_name

    if is_whisper_name(cfg.model):
        await _run_speech_server(cfg, ready)
        return
    if cfg.task == "transcription":
        raise ValueError(f"--task transcription needs a speech model; {cfg.model!r} is a text model")
    world = cfg.tensor_parallel_size * max(1, cfg.pipeline_parallel_size)
    if world > 1:
        from ..parallel.launch import init_parallel_from_env

        init_parallel_from_env(cfg.tensor_parallel_size, max(1, cfg.pipeline_parallel_size))
        if cfg.enable_lora:
            raise ValueError("--enable-lora needs a single-process replica (tp = pp = 1)")
    engine = LLMEngine(cfg)
    if world > 1:
        from ..parallel.state import get_replica

        if not get_replica().is_driver:
            # worker rank: no HTTP front, no scheduler; execute the driver's step plans
            if ready is not None:
                ready.set()
            await asyncio.get_running_loop().run_in_executor(None, engine.runner.worker_loop)
            return
    if cfg.enable_lora:
        from .lora import LoRAManager

        engine.lora_manager = LoRAManager(engine.runner.model, cfg.max_loras, cfg.max_lora_rank)
        engine.runner.model.lora = engine.lora_manager
    ae = AsyncLLMEngine(engine)
    loop = asyncio.get_running_loop()
    ae.start(loop)
    runner = await serve(ae, cfg.host, cfg.port)
    stop = asyncio.Event()
    for sig in (signal.SIGINT, signal.SIGTERM):
        try:
            loop.add_signal_handler(sig, stop.set)
        except (NotImplementedError, RuntimeError):
            pass
    if ready is not None:
        ready.set()
    await stop.wait()
    await runner.cleanup()
    ae.stop()


async def _run_speech_server(cfg: EngineConfig, ready=None) -> None:
    from .speech import SpeechEngine, serve_speech

    engine = SpeechEngine(cfg)
    runner = await serve_speech(engine, cfg.host, cfg.port)
    loop = asyncio.get_running_loop()
    stop = asyncio.Event()
    for sig in (signal.SIGINT, signal.SIGTERM):
        try:
            loop.add_signal_handler(sig, stop.set)
        except (NotImplementedError, RuntimeError):
            pass
    if ready is not None:
        ready.set()
    await stop.wait()
    await runner.cleanup()
    engine.shutdown()


def main(argv: list[str] | None = None) -> None:
    ap = build_parser()
    args, unknown = ap.parse_known_args(argv)
    logging.basicConfig(level=getattr(logging, str(args.log_level).upper(), logging.INFO),
                        format="%(levelname)s %(asctime)s %(name)s] %(message)s")
    if unknown:
        logging.getLogger("vpsa.cli").warning("ignoring unsupported flags: %s", " ".join(unknown))
    cfg = config_from_args(args)
    asyncio.run(run_server(cfg))


if __name__ == "__main__":
    main(sys.argv[1:])
