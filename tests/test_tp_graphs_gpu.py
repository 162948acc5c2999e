"""Tensor-parallel replica with HIP-graph decode on every rank.

Two processes share the one card of the test box (``VPSA_TP_BACKEND=gloo``: RCCL refuses two
ranks on one device, so control messages and prefill-sized operands go over gloo through the
host, and the decode-step collectives are the one-shot all-reduce / all-gather kernels over
hipIpc-mapped buffers: the ones a captured graph contains).  The graph-replayed run must
generate exactly what the eager run of the same layout generates, the worker rank must have
captured and replayed its own graphs, and decode steps must travel as inline integer plans."""

import copy
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from vllm_production_stack_amd.engine.config import EngineConfig
from vllm_production_stack_amd.engine.request import SamplingParams

pytestmark = pytest.mark.gpu

SP = SamplingParams(max_tokens=12, temperature=0, ignore_eos=True)


def _prompts():
    g = torch.Generator().manual_seed(5)
    return [torch.randint(16, 500, (n,), generator=g).tolist() for n in (150, 70, 9, 300)]


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank_main(rank, world, port, eager, q, model="tiny-llama-gqa", ep=False):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), VPSA_TP_BACKEND="gloo",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist

    from vllm_production_stack_amd.engine.llm_engine import LLMEngine
    from vllm_production_stack_amd.parallel.launch import init_parallel_from_env
    from vllm_production_stack_amd.parallel.state import get_replica, get_tp

    init_parallel_from_env(world, 1)
    cfg = EngineConfig(model=model, max_model_len=1024, max_num_seqs=8,
                       max_num_batched_tokens=256, num_gpu_blocks_override=96, port=0,
                       tensor_parallel_size=world, enforce_eager=eager, enable_expert_parallel=ep)
    eng = LLMEngine(cfg)
    info = {"rank": rank, "experts": list(eng.runner.model.layers[0].w13.shape) if ep else None, "graphs": len(eng.runner._graphs), "use_graphs": eng.runner.use_graphs,
            "oneshot": get_tp().oneshot is not None}
    if get_replica().is_driver:
        outs = eng.generate(_prompts(), copy.deepcopy(SP))
        outs2 = eng.generate(_prompts()[:2], copy.deepcopy(SP))  # through the prefix cache
        info["failed"] = get_tp().oneshot.failed()
        q.put(("driver", info, outs, outs2))
        eng.shutdown()
    else:
        eng.runner.worker_loop()
        info["failed"] = get_tp().oneshot.failed()
        q.put(("worker", info, None, None))
    dist.barrier()
    get_tp().oneshot.close()
    dist.destroy_process_group()


def _run(eager: bool, model="tiny-llama-gqa", ep=False):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, eager, q, model, ep)) for r in range(world)]
    for p in procs:
        p.start()
    import queue as _queue
    import time as _time

    got, deadline = [], _time.time() + 100
    try:
        while len(got) < world:
            try:
                got.append(q.get(timeout=1))
            except _queue.Empty:
                dead = [p for p in procs if p.exitcode not in (None, 0)]
                if dead or _time.time() > deadline:
                    raise AssertionError(f"rank failed or timed out (exit codes {[p.exitcode for p in procs]})")
        for p in procs:
            p.join(timeout=30)
    finally:
        for p in procs:
            if p.is_alive():
                p.kill()
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    by = {g[0]: g for g in got}
    return by["driver"], by["worker"]


def test_tp2_graph_decode_matches_eager_and_runs_on_every_rank():
    d_eager, w_eager = _run(eager=True)
    d_graph, w_graph = _run(eager=False)
    assert not d_eager[1]["use_graphs"] and d_eager[1]["oneshot"]
    for info in (d_graph[1], w_graph[1]):
        assert info["use_graphs"] and info["oneshot"] and info["graphs"] > 0, info
        assert not info["failed"], info
    assert d_graph[2] == d_eager[2], (d_graph[2], d_eager[2])
    assert d_graph[3] == d_eager[3]
    assert all(len(o) == SP.max_tokens for o in d_graph[2])


def test_expert_parallel_replica_replays_graphs_and_agrees_with_sliced_experts():
    """``--enable-expert-parallel`` on two ranks: each holds two whole experts of four (full width),
    decode steps replay graphs whose MoE layers sort foreign pairs out of the way, and the tokens
    are those of the eager run; against the layout that slices every expert the two differ only by
    the order of bf16 sums, so the first tokens agree."""
    d_eager, _ = _run(eager=True, model="tiny-mixtral-wide", ep=True)
    d_graph, w_graph = _run(eager=False, model="tiny-mixtral-wide", ep=True)
    for info in (d_graph[1], w_graph[1]):
        assert info["experts"] == [2, 2048, 256], info
        assert info["use_graphs"] and info["graphs"] > 0 and not info["failed"], info
    assert d_graph[2] == d_eager[2], (d_graph[2], d_eager[2])
    assert d_graph[3] == d_eager[3]
    d_sliced, _ = _run(eager=True, model="tiny-mixtral-wide", ep=False)
    same_first = sum(a[0] == b[0] for a, b in zip(d_eager[2], d_sliced[2]))
    assert same_first >= len(d_eager[2]) - 1, (d_eager[2], d_sliced[2])


def test_pipeline_stages_on_the_hip_kernels_hand_over_through_the_host():
    """Two pipeline stages share the card (gloo default group): the hand-over between stages must
    wait for the producing kernels and land before the consuming ones, which gloo does not do by
    itself for a device pointer.  The rehearsal script generates through both stages and compares
    with one stage holding every layer."""
    import json
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    env = dict(os.environ, VPSA_TP_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    run = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                          str(root / "benchmarks" / "pp_rehearsal.py"), "--pp", "2", "--model", "tiny-llama-gqa",
                          "--prompt-len", "300", "--max-tokens", "8", "--num-gpu-blocks", "96"],
                         env=env, cwd=root, capture_output=True, text=True, timeout=300)
    line = next((ln for ln in run.stdout.splitlines() if ln.startswith("{")), None)
    assert run.returncode == 0 and line, run.stderr[-3000:]
    res = json.loads(line)
    assert res["device"] == "cuda" and res["agrees"], res
    assert [s["layers"] for s in res["stages"]] == [[0, 2], [2, 3]]
