"""Process-group state for one engine replica.

One process per GPU; a replica with tensor parallelism spans ``tp`` processes that
run the same step loop in lockstep.  ``torch.distributed`` backend "nccl" is RCCL on
ROCm (xGMI inside a node); CPU tests use gloo.

xGMI is a point-to-point mesh (7 links per GPU), so a ring all-reduce is bound by one
link.  For the small decode-time messages ([batch, hidden] bf16, <= 2 MiB) the all-reduce
here is issued as a single RCCL call per layer half (RCCL picks its one-shot/direct
algorithm under its tree/ring threshold); prefill-time messages (tens of MiB) are issued
as reduce-scatter + all-gather so that every link carries 1/tp of the payload.
"""

from __future__ import annotations

import os
from dataclasses import dataclass

import torch
import torch.distributed as dist


@dataclass
class TPGroup:
    rank: int = 0
    size: int = 1
    group: "dist.ProcessGroup | None" = None
    cpu_group: "dist.ProcessGroup | None" = None  # gloo side channel for control messages

    @property
    def is_driver(self) -> bool:
        return self.rank == 0


_TP = TPGroup()
# messages at or above this many bytes use reduce-scatter + all-gather
TWO_SHOT_BYTES = 8 << 20


def get_tp() -> TPGroup:
    return _TP


def init_tp(tp_size: int, ranks: list[int] | None = None, backend: str | None = None) -> TPGroup:
    """Create the tensor-parallel group from the default process group.

    ``ranks`` are global ranks of this replica's group (default: all ranks)."""
    global _TP
    if tp_size <= 1:
        _TP = TPGroup()
        return _TP
    if not dist.is_initialized():
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        dist.init_process_group(backend=backend, init_method="env://")
    world = dist.get_world_size()
    ranks = ranks or list(range(world))
    assert len(ranks) == tp_size, (ranks, tp_size)
    group = dist.new_group(ranks=ranks)
    cpu_group = dist.new_group(ranks=ranks, backend="gloo")
    _TP = TPGroup(rank=ranks.index(dist.get_rank()), size=tp_size, group=group, cpu_group=cpu_group)
    return _TP


def set_tp(tp: TPGroup) -> None:
    global _TP
    _TP = tp


def tp_all_reduce(x: torch.Tensor) -> torch.Tensor:
    tp = _TP
    if tp.size == 1:
        return x
    nbytes = x.numel() * x.element_size()
    if nbytes >= TWO_SHOT_BYTES and x.shape[0] % tp.size == 0 and x.is_contiguous():
        shard = torch.empty((x.shape[0] // tp.size, *x.shape[1:]), dtype=x.dtype, device=x.device)
        dist.reduce_scatter_tensor(shard, x, group=tp.group)
        dist.all_gather_into_tensor(x, shard, group=tp.group)
        return x
    dist.all_reduce(x, group=tp.group)
    return x


def tp_all_gather_last_dim(x: torch.Tensor) -> torch.Tensor:
    """[..., n] shards -> [..., n * tp] (used for the vocab-parallel LM head)."""
    tp = _TP
    if tp.size == 1:
        return x
    x = x.contiguous()
    out = torch.empty((tp.size, *x.shape), dtype=x.dtype, device=x.device)
    dist.all_gather_into_tensor(out, x, group=tp.group)
    return out.movedim(0, -2).reshape(*x.shape[:-1], x.shape[-1] * tp.size)


def broadcast_object(obj, src: int = 0):
    """Control-plane broadcast from the replica's driver rank over the gloo side channel."""
    tp = _TP
    if tp.size == 1:
        return obj
    box = [obj]
    src_global = dist.get_global_rank(tp.cpu_group, src)
    dist.broadcast_object_list(box, src=src_global, group=tp.cpu_group)
    return box[0]


def local_rank() -> int:
    return int(os.environ.get("LOCAL_RANK", "0"))
