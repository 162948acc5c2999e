"""One-shot all-reduce over hipIpc-mapped peer buffers (``csrc/comm/comm.cpp``).

For the small all-reduces of tensor-parallel decode steps ([batch, hidden] bf16, a few hundred
KB): every rank publishes its operand once and reads all peers' copies directly over its own
xGMI links, in one kernel that can be replayed from a HIP graph, instead of a multi-hop ring.
Opt-in (``VPSA_ONESHOT_ALLREDUCE=1``): the handshake and the arithmetic are tested with two
processes sharing one GPU; whoever enables it on a multi-GPU node should run
``tests/test_oneshot_allreduce.py`` there first.  Larger operands and other dtypes keep using
RCCL (``parallel/state.py::tp_all_reduce``)."""

from __future__ import annotations

import ctypes
from ctypes import c_int, c_long, c_size_t, c_void_p

import torch

from .. import _native

_lib = None


def _comm():
    global _lib
    if _lib is None:
        lib = _native.load("vpsa_comm")
        lib.comm_ipc_handle_size.restype = c_int
        lib.comm_bind_device.argtypes = [c_int]
        lib.comm_ar_create.restype = c_void_p
        lib.comm_ar_create.argtypes = [c_int, c_int, c_size_t]
        lib.comm_ar_export.restype = c_int
        lib.comm_ar_export.argtypes = [c_void_p, c_void_p]
        lib.comm_ar_open.restype = c_int
        lib.comm_ar_open.argtypes = [c_void_p, c_void_p]
        lib.comm_ar_max_elems.restype = c_long
        lib.comm_ar_max_elems.argtypes = [c_void_p]
        lib.comm_ar_allreduce_bf16.restype = c_int
        lib.comm_ar_allreduce_bf16.argtypes = [c_void_p, c_void_p, c_long, c_void_p]
        lib.comm_ar_allgather_bf16.restype = c_int
        lib.comm_ar_allgather_bf16.argtypes = [c_void_p, c_void_p, c_void_p, c_long, c_void_p]
        lib.comm_ar_failed.restype = c_int
        lib.comm_ar_failed.argtypes = [c_void_p]
        lib.comm_ar_failed_nosync.argtypes = [c_void_p]
        lib.comm_ar_fine_grained.argtypes = [c_void_p]
        lib.comm_ar_destroy.argtypes = [c_void_p]
        _lib = lib
    return _lib


class OneShotAllReduce:
    """``exchange(my_handle: bytes) -> list[bytes]`` must return every rank's handle in rank
    order (e.g. ``dist.all_gather_object`` over the group's gloo side channel)."""

    def __init__(self, rank: int, world: int, device: torch.device, exchange, max_bytes: int = 4 << 20,
                 reduce_max_bytes: int | None = None):
        lib = _comm()
        self.lib, self.rank, self.world, self.device = lib, rank, world, device
        lib.comm_bind_device(device.index if device.index is not None else torch.cuda.current_device())
        self.ctx = lib.comm_ar_create(rank, world, max_bytes)
        if not self.ctx:
            raise RuntimeError("one-shot all-reduce: could not allocate the published buffer")
        hs = lib.comm_ipc_handle_size()
        buf = ctypes.create_string_buffer(hs)
        rc = lib.comm_ar_export(self.ctx, buf)
        if rc != 0:
            raise RuntimeError(f"one-shot all-reduce: hipIpcGetMemHandle failed ({rc})")
        handles = exchange(buf.raw)
        if len(handles) != world or any(len(h) != hs for h in handles):
            raise RuntimeError("one-shot all-reduce: handle exchange returned the wrong shape")
        rc = lib.comm_ar_open(self.ctx, b"".join(handles))
        if rc != 0:
            raise RuntimeError(f"one-shot all-reduce: could not map the buffer of rank {-rc - 1}")
        self.max_elems = int(lib.comm_ar_max_elems(self.ctx))
        # sums above this size go to RCCL (32 workgroups are sized for decode-step operands); a
        # gather may use the whole buffer (every rank has to read every shard whatever the route)
        self.reduce_max_elems = min(self.max_elems, (reduce_max_bytes or max_bytes) // 2)

    def _ok(self, x: torch.Tensor, limit: int) -> bool:
        return (x.is_cuda and x.dtype == torch.bfloat16 and x.is_contiguous() and x.numel() % 8 == 0
                and 0 < x.numel() <= limit and x.data_ptr() % 16 == 0)

    def fits(self, x: torch.Tensor) -> bool:
        return self._ok(x, self.reduce_max_elems)

    def fits_gather(self, x: torch.Tensor) -> bool:
        return self._ok(x, self.max_elems)

    def all_reduce(self, x: torch.Tensor) -> torch.Tensor:
        """In-place sum over the ranks on the current stream."""
        rc = self.lib.comm_ar_allreduce_bf16(self.ctx, c_void_p(x.data_ptr()), c_long(x.numel()),
                                             c_void_p(torch.cuda.current_stream(self.device).cuda_stream))
        if rc != 0:
            raise RuntimeError("one-shot all-reduce: operand does not fit")
        return x

    def all_gather(self, x: torch.Tensor) -> torch.Tensor:
        """[world, *x.shape]: every rank's ``x`` in rank order, on the current stream (same
        publication and handshake as the sum; graph-replayable)."""
        out = torch.empty((self.world, *x.shape), dtype=x.dtype, device=x.device)
        rc = self.lib.comm_ar_allgather_bf16(self.ctx, c_void_p(x.data_ptr()), c_void_p(out.data_ptr()),
                                             c_long(x.numel()),
                                             c_void_p(torch.cuda.current_stream(self.device).cuda_stream))
        if rc != 0:
            raise RuntimeError("one-shot all-gather: operand does not fit")
        return out

    def failed(self) -> bool:
        """True if a call gave up waiting for a peer (synchronises the device)."""
        return bool(self.lib.comm_ar_failed(self.ctx))

    def failed_nosync(self) -> bool:
        """What the all-reduce kernels that have already finished reported, read from pinned host
        memory: no device synchronisation, meant to be called once per engine step after the
        step's own read-back."""
        v = int(self.lib.comm_ar_failed_nosync(self.ctx))
        return self.failed() if v < 0 else bool(v)

    @property
    def fine_grained(self) -> bool:
        return bool(self.lib.comm_ar_fine_grained(self.ctx))

    def close(self) -> None:
        if self.ctx:
            self.lib.comm_ar_destroy(self.ctx)
            self.ctx = None
