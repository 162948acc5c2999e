"""Process-group state for one engine replica.

One process per GPU.  A replica spans ``pp x tp`` processes: rank = pp_rank * tp + tp_rank,
so a tensor-parallel group is a run of consecutive ranks (one node, xGMI) and pipeline
stages are ``tp`` ranks apart (node to node).  All ranks run the same step loop in lockstep:
the driver (global rank 0 of the replica) schedules and broadcasts each step's plan over a
gloo side channel; every rank executes its shard/stage of it.

``torch.distributed`` backend "nccl" is RCCL on ROCm; CPU tests use gloo.

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
    oneshot: object = None  # OneShotAllReduce when enabled (small bf16 all-reduces)
    failed: bool = False  # a one-shot collective timed out: this replica has to be restarted
    host_staged: "bool | None" = None  # GPU operands over a gloo group (one-card rehearsal)

    @property
    def is_driver(self) -> bool:
        return self.rank == 0


@dataclass
class PPGroup:
    """Pipeline position of this process.  ``prev``/``next`` are global ranks of the peers that
    hold the same tensor-parallel shard in the neighbouring stages; ``first``/``last`` are the
    global ranks of that shard's first and last stage."""
    rank: int = 0
    size: int = 1
    prev: int = -1
    next: int = -1
    first: int = 0
    last: int = 0

    @property
    def is_first(self) -> bool:
        return self.rank == 0

    @property
    def is_last(self) -> bool:
        return self.rank == self.size - 1

    def layer_range(self, num_layers: int) -> tuple[int, int]:
        """Contiguous layers [a, b) of this stage; earlier stages take the remainder."""
        base, extra = divmod(num_layers, self.size)
        a = self.rank * base + min(self.rank, extra)
        return a, a + base + (1 if self.rank < extra else 0)


@dataclass
class ReplicaGroup:
    """All processes of the replica: the plan broadcast runs here (gloo)."""
    rank: int = 0
    size: int = 1
    cpu_group: "dist.ProcessGroup | None" = None
    driver: int = 0  # global rank of the driver

    @property
    def is_driver(self) -> bool:
        return self.rank == 0


_TP = TPGroup()
_PP = PPGroup()
_REPLICA = ReplicaGroup()
# messages at or above this many bytes use reduce-scatter + all-gather
TWO_SHOT_BYTES = 8 << 20


def get_tp() -> TPGroup:
    return _TP


def get_pp() -> PPGroup:
    return _PP


def get_replica() -> ReplicaGroup:
    return _REPLICA


def set_tp(tp: TPGroup) -> None:
    global _TP
    _TP = tp


def set_pp(pp: PPGroup) -> None:
    global _PP
    _PP = pp


def set_replica(r: ReplicaGroup) -> None:
    global _REPLICA
    _REPLICA = r


def reset_parallel_state() -> None:
    global _PP_HOST_STAGED
    _PP_HOST_STAGED = None
    set_tp(TPGroup())
    set_pp(PPGroup())
    set_replica(ReplicaGroup())


def init_replica_parallel(tp_size: int, pp_size: int = 1) -> None:
    """Carve the default process group into replicas of ``pp_size x tp_size`` ranks and record
    this process's tensor-parallel group, pipeline neighbours and replica control group.
    Every rank takes part in the creation of every group (a ``new_group`` requirement)."""
    per = tp_size * pp_size
    if per <= 1:
        reset_parallel_state()
        return
    assert dist.is_initialized(), "init_process_group first"
    world, me = dist.get_world_size(), dist.get_rank()
    if world % per != 0:
        raise ValueError(f"world size {world} is not a multiple of tp*pp = {per}")
    for rep in range(world // per):
        base = rep * per
        rep_ranks = list(range(base, base + per))
        rep_cpu = dist.new_group(ranks=rep_ranks, backend="gloo")
        mine = me in rep_ranks
        if mine:
            set_replica(ReplicaGroup(rank=me - base, size=per, cpu_group=rep_cpu, driver=base))
        for stage in range(pp_size):
            ranks = [base + stage * tp_size + t for t in range(tp_size)]
            grp = dist.new_group(ranks=ranks) if tp_size > 1 else None
            cpu = dist.new_group(ranks=ranks, backend="gloo") if tp_size > 1 else None
            if me in ranks:
                t = ranks.index(me)
                set_tp(TPGroup(rank=t, size=tp_size, group=grp, cpu_group=cpu))
                set_pp(PPGroup(rank=stage, size=pp_size,
                               prev=me - tp_size if stage > 0 else -1,
                               next=me + tp_size if stage < pp_size - 1 else -1,
                               first=base + t, last=base + (pp_size - 1) * tp_size + t))


def enable_oneshot_all_reduce(device: torch.device, max_bytes: int = 4 << 20,
                              reduce_max_bytes: int = 4 << 20) -> bool:
    """Map the tensor-parallel peers' published buffers (hipIpc) so that small bf16 all-reduces
    and all-gathers take the one-shot kernels of ``parallel/oneshot.py``.  All ranks of the group
    call this; calling it again replaces the buffers (e.g. by larger ones)."""
    tp = _TP
    if tp.size == 1 or device.type != "cuda":
        return False
    from .oneshot import OneShotAllReduce

    if tp.oneshot is not None:
        torch.cuda.synchronize(device)
        dist.barrier(group=tp.cpu_group)  # nobody unmaps while a peer's kernel still reads
        tp.oneshot.close()
        tp.oneshot = None
        dist.barrier(group=tp.cpu_group)

    def exchange(handle: bytes) -> list[bytes]:
        got: list = [None] * tp.size
        dist.all_gather_object(got, handle, group=tp.cpu_group)
        return got

    tp.oneshot = OneShotAllReduce(tp.rank, tp.size, device, exchange, max_bytes, reduce_max_bytes)
    return True


class CollectiveError(RuntimeError):
    """A collective of this replica gave up waiting for a peer: the replica cannot go on."""


def check_collectives() -> None:
    """Once per engine step, after the step's read-back: did a one-shot all-reduce of the steps
    finished so far give up waiting for a peer?  Its operand was then left unreduced, so what
    was sampled from it is garbage, and the ranks no longer agree on where they are.  The replica
    is marked failed for good and the error is raised on this and on every later call: the engine
    stops stepping and reports unhealthy, worker ranks leave their loop, and whatever supervises
    the replica (torchrun, the pod's liveness probe) restarts it as fresh processes.  Nothing is
    unmapped or freed here: the captured decode graphs of this rank still hold launches of the
    one-shot kernels with these buffers, and a peer's kernel may still be polling them; the
    mappings go away with the process."""
    tp = _TP
    if tp.failed or (tp.oneshot is not None and tp.oneshot.failed_nosync()):
        tp.failed = True
        raise CollectiveError("tensor-parallel all-reduce timed out waiting for a peer rank; "
                              "the replica is marked failed and must be restarted")


def _host_staged(tp: TPGroup) -> bool:
    """GPU tensors over a gloo group (several ranks rehearsing on ONE card, where RCCL refuses to
    form a communicator): operands that the one-shot kernels do not take go through the host."""
    if tp.host_staged is None:
        tp.host_staged = dist.get_backend(tp.group) == "gloo"
    return tp.host_staged


def tp_all_reduce(x: torch.Tensor) -> torch.Tensor:
    tp = _TP
    if tp.size == 1:
        return x
    if tp.oneshot is not None and tp.oneshot.fits(x):
        return tp.oneshot.all_reduce(x)
    if x.is_cuda and _host_staged(tp):
        h = x.float().cpu()  # summed in fp32, like the one-shot kernel
        dist.all_reduce(h, group=tp.group)
        x.copy_(h)
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
    x = x.contiguous().view(-1, x.shape[-1])
    lead = x.shape[0]
    if tp.oneshot is not None and tp.oneshot.fits_gather(x):
        out = tp.oneshot.all_gather(x)  # [tp, lead, n]
        return out.movedim(0, 1).reshape(lead, tp.size * x.shape[1])
    if x.is_cuda and _host_staged(tp):
        parts = [torch.empty(x.shape, dtype=x.dtype) for _ in range(tp.size)]
        dist.all_gather(parts, x.cpu(), group=tp.group)
        return torch.cat(parts, dim=1).to(x.device)
    out = torch.empty((tp.size * lead, x.shape[1]), dtype=x.dtype, device=x.device)
    dist.all_gather_into_tensor(out, x, group=tp.group)  # rank-major concatenation along dim 0
    return out.view(tp.size, lead, x.shape[1]).movedim(0, 1).reshape(lead, tp.size * x.shape[1])


def _pp_host_staged() -> bool:
    """GPU stages over a gloo default group (ranks rehearsing on fewer cards than ranks): gloo takes
    a device pointer as it is and reads or writes it from the host without regard to the stream
    that produces or consumes it, so the hand-over goes through host memory explicitly."""
    global _PP_HOST_STAGED
    if _PP_HOST_STAGED is None:
        _PP_HOST_STAGED = dist.get_backend() == "gloo"
    return _PP_HOST_STAGED


_PP_HOST_STAGED: "bool | None" = None


def pp_send(x: torch.Tensor, dst: int) -> None:
    if x.is_cuda and _pp_host_staged():
        dist.send(x.contiguous().cpu(), dst=dst)  # .cpu() waits for the kernels that wrote x
        return
    dist.send(x.contiguous(), dst=dst)


def pp_recv(shape, dtype: torch.dtype, device: torch.device, src: int) -> torch.Tensor:
    if device.type == "cuda" and _pp_host_staged():
        h = torch.empty(shape, dtype=dtype)
        dist.recv(h, src=src)
        return h.to(device)
    t = torch.empty(shape, dtype=dtype, device=device)
    dist.recv(t, src=src)
    return t


# Control channel of a replica (driver -> every rank, gloo).  Every message starts with ONE
# fixed-size int64 tensor: word 0 is the kind, word 1 the payload length.  A decode step that
# replays a captured graph travels entirely inside that tensor (token ids, positions, slots,
# context lengths and the block-table rows that changed: a few hundred integers), so the common
# step costs one small broadcast and no pickling; everything else announces a pickled object that
# follows in a second broadcast.
CTRL_WORDS = 4096
CTRL_OBJECT, CTRL_INTS = 1, 2


def _ctrl_tensor(kind: int, payload=()) -> torch.Tensor:
    t = torch.zeros(CTRL_WORDS, dtype=torch.int64)
    t[0], t[1] = kind, len(payload)
    if len(payload):
        t[2:2 + len(payload)] = torch.as_tensor(payload, dtype=torch.int64)
    return t


def replica_broadcast(obj):
    """Driver: send a picklable control message to all ranks of the replica."""
    r = _REPLICA
    if r.size == 1:
        return obj
    dist.broadcast(_ctrl_tensor(CTRL_OBJECT), src=r.driver, group=r.cpu_group)
    box = [obj]
    dist.broadcast_object_list(box, src=r.driver, group=r.cpu_group)
    return box[0]


def replica_broadcast_ints(payload) -> bool:
    """Driver: send a list of integers inside the header tensor itself.  False when it does not
    fit (the caller then sends it as an object)."""
    r = _REPLICA
    if r.size == 1:
        return True
    if len(payload) > CTRL_WORDS - 2:
        return False
    dist.broadcast(_ctrl_tensor(CTRL_INTS, payload), src=r.driver, group=r.cpu_group)
    return True


def replica_next():
    """Worker ranks: the driver's next message, ``(CTRL_OBJECT, obj)`` or ``(CTRL_INTS, list)``."""
    r = _REPLICA
    t = torch.zeros(CTRL_WORDS, dtype=torch.int64)
    dist.broadcast(t, src=r.driver, group=r.cpu_group)
    kind, n = int(t[0]), int(t[1])
    if kind == CTRL_INTS:
        return kind, t[2:2 + n].tolist()
    box = [None]
    dist.broadcast_object_list(box, src=r.driver, group=r.cpu_group)
    return CTRL_OBJECT, box[0]


def replica_min_int(v: int) -> int:
    """Minimum of an integer over the replica (e.g. the KV page count every rank can hold)."""
    r = _REPLICA
    if r.size == 1:
        return v
    t = torch.tensor([v], dtype=torch.int64)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=r.cpu_group)
    return int(t.item())


def local_rank() -> int:
    return int(os.environ.get("LOCAL_RANK", "0"))
