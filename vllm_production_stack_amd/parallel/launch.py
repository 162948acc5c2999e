"""Process-group bring-up for a tensor/pipeline-parallel replica (one process per GPU).

``vpsa-engine serve ... --tensor-parallel-size T --pipeline-parallel-size P`` is launched once
per rank (torchrun, the stack launcher or the StatefulSet of the chart sets RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_ADDR / MASTER_PORT).  The driver rank serves HTTP, schedules, and
broadcasts each step's plan; the other ranks sit in ``ModelRunner.worker_loop``.
"""

from __future__ import annotations

import os

import torch
import torch.distributed as dist

from .state import get_tp, init_replica_parallel, reset_parallel_state


def init_parallel_from_env(tp_size: int, pp_size: int = 1):
    if tp_size * pp_size <= 1:
        reset_parallel_state()
        return get_tp()
    if "RANK" not in os.environ:
        raise RuntimeError(
            "tensor/pipeline parallelism needs one process per GPU: launch with "
            "`python -m torch.distributed.run --nproc-per-node N -m "
            "vllm_production_stack_amd.engine.cli serve ...`")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29544")
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # VPSA_TP_BACKEND=gloo: several ranks rehearse on fewer cards than ranks (RCCL refuses two ranks
    # on one device): control and large operands over gloo through the host, small bf16 operands
    # through the one-shot kernels, every rank on card LOCAL_RANK mod the number of cards.
    rehearsal = os.environ.get("VPSA_TP_BACKEND", "") == "gloo"
    if torch.cuda.is_available():
        local = local % torch.cuda.device_count() if rehearsal else local
        torch.cuda.set_device(local)
        backend = "gloo" if rehearsal else "nccl"
    else:
        backend = "gloo"
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, init_method="env://")
    init_replica_parallel(tp_size, pp_size)
    if (tp_size > 1 and torch.cuda.is_available()
            and os.environ.get("VPSA_ONESHOT_ALLREDUCE", "1" if rehearsal else "0") == "1"):
        from .state import enable_oneshot_all_reduce

        # (the model runner re-creates it larger when the logits shard of its biggest graph batch
        # would not fit)
        enable_oneshot_all_reduce(torch.device("cuda", local))
    return get_tp()
