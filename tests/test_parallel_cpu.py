"""Tensor- and pipeline-parallel replicas on CPU (gloo, several processes): every layout must
generate exactly what the single-process engine generates from the same seed, and worker ranks
must leave their loop when the driver shuts down."""

import copy
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from vllm_production_stack_amd.engine.config import EngineConfig
from vllm_production_stack_amd.engine.request import SamplingParams

SP = SamplingParams(max_tokens=6, temperature=0, ignore_eos=True)


def _prompts():
    g = torch.Generator().manual_seed(11)
    return [torch.randint(16, 500, (n,), generator=g).tolist() for n in (150, 70, 9)]


def _cfg(model, **kw):
    return EngineConfig(model=model, device="cpu", dtype="float32", max_model_len=1024,
                        max_num_seqs=8, max_num_batched_tokens=128, num_gpu_blocks_override=48,
                        port=0, **kw)


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank_main(rank, world, port, model, tp, pp, q, ep=False):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    from vllm_production_stack_amd.engine.llm_engine import LLMEngine
    from vllm_production_stack_amd.parallel.launch import init_parallel_from_env
    from vllm_production_stack_amd.parallel.state import get_replica

    init_parallel_from_env(tp, pp)
    eng = LLMEngine(_cfg(model, tensor_parallel_size=tp, pipeline_parallel_size=pp, enable_expert_parallel=ep))
    assert eng.runner.model.ep == ep
    if get_replica().is_driver:
        outs = eng.generate(_prompts(), copy.deepcopy(SP))
        # a second wave through the prefix cache and a chunked prefill (budget 128 < 150 rows)
        outs2 = eng.generate(_prompts()[:1], copy.deepcopy(SP))
        q.put((outs, outs2))
        eng.shutdown()
    else:
        eng.runner.worker_loop()
        q.put(("worker-exit", rank))
    import torch.distributed as dist

    dist.barrier()
    dist.destroy_process_group()


def _run_layout(model, tp, pp, ep=False):
    world = tp * pp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, model, tp, pp, q, ep)) for r in range(world)]
    for p in procs:
        p.start()
    import queue as _queue
    import time as _time

    got, deadline = [], _time.time() + 240
    while len(got) < world:
        try:
            got.append(q.get(timeout=1))
        except _queue.Empty:
            dead = [p for p in procs if p.exitcode not in (None, 0)]
            if dead or _time.time() > deadline:
                for p in procs:
                    p.kill()
                raise AssertionError(f"rank process failed or timed out (exit codes {[p.exitcode for p in procs]})")
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    result = [g for g in got if g[0] != "worker-exit"]
    assert len(result) == 1 and len(got) == world
    return result[0]


def _single(model):
    from vllm_production_stack_amd.engine.llm_engine import LLMEngine
    from vllm_production_stack_amd.parallel.state import reset_parallel_state

    reset_parallel_state()
    eng = LLMEngine(_cfg(model))
    outs = eng.generate(_prompts(), copy.deepcopy(SP))
    outs2 = eng.generate(_prompts()[:1], copy.deepcopy(SP))
    return outs, outs2


@pytest.mark.parametrize("model,tp,pp", [
    ("tiny-llama-gqa", 2, 1),
    ("tiny-llama-gqa", 1, 3),
    ("tiny-llama-gqa", 2, 2),
    ("tiny-mixtral", 2, 1),
])
def test_parallel_layout_matches_single_process(model, tp, pp):
    want = _single(model)
    got = _run_layout(model, tp, pp)
    assert got[0] == want[0]
    assert got[1] == want[1]


def test_expert_parallel_ranks_hold_whole_experts_and_match_single_process():
    """``--enable-expert-parallel``: the two ranks of a tensor-parallel Mixtral hold four whole
    experts each; pairs routed to the other rank's experts contribute nothing locally and the
    layer's all-reduce sums the halves."""
    want = _single("tiny-mixtral")
    got = _run_layout("tiny-mixtral", 2, 1, ep=True)
    assert got[0] == want[0]
    assert got[1] == want[1]


def test_layer_ranges_cover_model():
    from vllm_production_stack_amd.parallel.state import PPGroup

    for layers in (3, 32, 33):
        for size in (1, 2, 3, 4):
            spans = [PPGroup(rank=r, size=size).layer_range(layers) for r in range(size)]
            assert spans[0][0] == 0 and spans[-1][1] == layers
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))


def test_a_timed_out_one_shot_all_reduce_fails_the_step_and_falls_back():
    """The per-step check: a one-shot all-reduce that gave up waiting for a peer left its operand
    unreduced; the step must fail loudly, and the kernel must not be used again."""
    import pytest
    from vllm_production_stack_amd.parallel import state

    class Stub:
        def __init__(self):
            self.bad, self.closed = False, False

        def failed_nosync(self):
            return self.bad

        def close(self):
            self.closed = True

    tp = state.get_tp()
    old = tp.oneshot
    stub = Stub()
    tp.oneshot = stub
    try:
        state.check_collectives()  # healthy: nothing happens
        assert tp.oneshot is stub
        stub.bad = True
        with pytest.raises(state.CollectiveError, match="all-reduce timed out"):
            state.check_collectives()
        # nothing is unmapped: captured graphs and polling peers may still touch the buffers
        assert tp.oneshot is stub and not stub.closed and tp.failed
        stub.bad = False
        with pytest.raises(state.CollectiveError):  # the replica stays failed: one more step fails the same way
            state.check_collectives()
    finally:
        tp.oneshot = old
        tp.failed = False


def test_a_failed_collective_stops_the_engine_thread_for_good():
    """An injected all-reduce time-out during a step: in-flight requests get the error, the engine
    thread takes no further step (a step after the failure would replay graphs whose ranks have
    diverged), the replica reports unhealthy for good and refuses new work."""
    import asyncio

    from vllm_production_stack_amd.engine.async_engine import AsyncLLMEngine
    from vllm_production_stack_amd.parallel import state

    class Eng:
        device = None
        is_sleeping = False
        requests: dict = {}

        def __init__(self):
            self.steps = 0
            self.work = False

        def add_request(self, ids, params, rid, lora, kvp, pooling):
            self.work = True
            return object()

        def has_work(self):
            return self.work

        def step(self):
            self.steps += 1
            raise state.CollectiveError("tensor-parallel all-reduce timed out waiting for a peer rank")

        def abort(self, rid):
            pass

        def idle_tick(self):
            pass

        def shutdown(self):
            pass

    async def go():
        eng = Eng()
        ae = AsyncLLMEngine(eng)
        ae.start(asyncio.get_running_loop())
        with pytest.raises(RuntimeError, match="timed out"):
            async for _ in ae.generate("r1", [1, 2, 3], None):
                pass
        await asyncio.sleep(0.3)
        assert eng.steps == 1, "no step after the failed one"
        assert not ae.healthy and ae.fatal is not None
        with pytest.raises(RuntimeError, match="replica failed"):
            async for _ in ae.generate("r2", [1], None):
                pass
        ae.stop()

    asyncio.run(go())


def test_decode_plan_travels_as_integers_and_back():
    from vllm_production_stack_amd.engine.model_runner import ModelRunner as M

    plan = {"bsz": 4, "ids": [11, 12, 13, 0], "pos": [5, 6, 700, 0], "slots": [69, 134, 7000, -1],
            "ctx": [6, 7, 701, 0], "bt": [(0, 0, [1, 2, 3]), (2, 10, [109])]}
    assert M._decode_plan_from_ints(M._decode_plan_to_ints(plan)) == plan
    empty = {"bsz": 1, "ids": [3], "pos": [0], "slots": [64], "ctx": [1], "bt": []}
    assert M._decode_plan_from_ints(M._decode_plan_to_ints(empty)) == empty


def _ctrl_rank(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist

    from vllm_production_stack_amd.parallel import state
    from vllm_production_stack_amd.parallel.launch import init_parallel_from_env

    init_parallel_from_env(world, 1)
    if state.get_replica().is_driver:
        assert state.replica_broadcast_ints([7, -1, 1 << 40])
        assert not state.replica_broadcast_ints(list(range(state.CTRL_WORDS)))  # too long: nothing is sent
        state.replica_broadcast(("step", {"ids": [1, 2]}))
        assert state.replica_broadcast_ints([])
        state.replica_broadcast(("stop",))
    else:
        got = [state.replica_next() for _ in range(4)]
        q.put(got)
    dist.barrier()
    dist.destroy_process_group()


def test_control_channel_carries_integer_plans_inline_and_objects_pickled():
    from vllm_production_stack_amd.parallel.state import CTRL_INTS, CTRL_OBJECT

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ctrl_rank, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got == [(CTRL_INTS, [7, -1, 1 << 40]), (CTRL_OBJECT, ("step", {"ids": [1, 2]})),
                   (CTRL_INTS, []), (CTRL_OBJECT, ("stop",))]
