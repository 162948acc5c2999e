"""Process-group bring-up for a tensor-parallel replica (one process per GPU).

``vpsa-engine serve ... --tensor-parallel-size N`` is launched once per rank (torchrun or
the stack launcher sets RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT); rank 0
of the group serves HTTP and broadcasts each step's schedule, the other ranks execute it.
"""

from __future__ import annotations

import os

import torch
import torch.distributed as dist

from .state import TPGroup, init_tp, set_tp


def init_tp_from_env(tp_size: int) -> TPGroup:
    if tp_size <= 1:
        set_tp(TPGroup())
        return TPGroup()
    if "RANK" not in os.environ:
        raise RuntimeError(
            "tensor parallelism needs one process per GPU: launch with "
            "`python -m torch.distributed.run --nproc-per-node N -m vllm_production_stack_amd.engine.cli serve ...`")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29544")
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.is_available():
        torch.cuda.set_device(local)
        backend = "nccl"
    else:
        backend = "gloo"
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, init_method="env://")
    world = dist.get_world_size()
    if world % tp_size != 0:
        raise ValueError(f"world size {world} is not a multiple of tensor-parallel size {tp_size}")
    me = dist.get_rank()
    group_id = me // tp_size
    tp = None
    # every rank must take part in the creation of every group
    for g in range(world // tp_size):
        ranks = list(range(g * tp_size, (g + 1) * tp_size))
        grp = dist.new_group(ranks=ranks)
        cpu = dist.new_group(ranks=ranks, backend="gloo")
        if g == group_id:
            tp = TPGroup(rank=ranks.index(me), size=tp_size, group=grp, cpu_group=cpu)
    set_tp(tp)
    return tp
