"""One-shot all-reduce (csrc/comm/comm.cpp): two processes map each other's published buffer
through hipIpc and reduce bf16 tensors in one kernel each.  On a one-GPU box both ranks share
the card (the handshake and the arithmetic are the same; the links are not exercised)."""

import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

SIZES = [8, 40 * 4096, 256 * 4096, 1 << 20]


def _rank(rank, world, q_in, q_out, result):
    os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    import torch

    from vllm_production_stack_amd.parallel.oneshot import OneShotAllReduce

    dev = torch.device("cuda", rank % torch.cuda.device_count())
    torch.cuda.set_device(dev)

    def exchange(handle):
        q_out.put((rank, handle))
        return q_in.get(timeout=60)

    ar = OneShotAllReduce(rank, world, dev, exchange, max_bytes=2 << 20)
    ok = True
    detail = ""
    for it, n in enumerate(SIZES * 3):
        xs = []
        for r in range(world):
            g = torch.Generator().manual_seed(1000 * it + r)
            xs.append(torch.randn(n, generator=g).to(torch.bfloat16))
        want = torch.stack([x.float() for x in xs]).sum(0).to(torch.bfloat16)
        mine = xs[rank].to(dev)
        assert ar.fits(mine)
        ar.all_reduce(mine)
        torch.cuda.synchronize()
        if not torch.equal(mine.cpu(), want):
            ok, detail = False, f"iteration {it}, n {n}: max diff {(mine.cpu().float() - want.float()).abs().max()}"
            break
    # all-gather through the same buffers: every rank's shard in rank order
    if ok:
        for n in (8, 7 * 1024, 1 << 19):
            shards = [torch.full((n,), float(r + 1), dtype=torch.bfloat16) + torch.arange(n).to(torch.bfloat16)
                      for r in range(world)]
            mine = shards[rank].to(dev)
            assert ar.fits_gather(mine)
            got = ar.all_gather(mine)
            torch.cuda.synchronize()
            if not torch.equal(got.cpu(), torch.stack(shards)):
                ok, detail = False, f"all-gather of {n} elements differs"
                break
    # replay from a HIP graph: the kernel advances its own epochs
    if ok:
        buf = torch.zeros(40 * 4096, dtype=torch.bfloat16, device=dev)
        s = torch.cuda.Stream(dev)
        with torch.cuda.stream(s):
            ar.all_reduce(buf)  # warm-up outside the graph keeps both ranks' call counts equal
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                ar.all_reduce(buf)
        torch.cuda.synchronize()
        for k in range(4):
            buf.fill_(float(rank + 1 + k))
            g.replay()
            torch.cuda.synchronize()
            expect = float(sum(r + 1 + k for r in range(world)))
            if not bool((buf.float() == expect).all()):
                ok, detail = False, f"graph replay {k}: got {buf[:4].tolist()}, want {expect}"
                break
    failed = ar.failed() or ar.failed_nosync()  # the second reads the pinned host word: no device sync
    big = torch.zeros(4 << 20, dtype=torch.bfloat16, device=dev)
    fits_big = ar.fits(big)
    result.put((rank, ok and not failed and not fits_big, detail or f"failed={failed} fits_big={fits_big} fine_grained={ar.fine_grained}"))
    q_in.get(timeout=60)  # hold the mapping until every rank is done
    ar.close()


def test_two_ranks_reduce_through_mapped_buffers():
    world = 2
    ctx = mp.get_context("spawn")
    q_ins = [ctx.Queue() for _ in range(world)]
    q_out, result = ctx.Queue(), ctx.Queue()
    procs = [ctx.Process(target=_rank, args=(r, world, q_ins[r], q_out, result)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        handles = dict(q_out.get(timeout=120) for _ in range(world))
        ordered = [handles[r] for r in range(world)]
        for q in q_ins:
            q.put(ordered)
        results = dict((r, (ok, d)) for r, ok, d in (result.get(timeout=120) for _ in range(world)))
        for q in q_ins:
            q.put("done")
        for p in procs:
            p.join(timeout=60)
    finally:
        for p in procs:
            if p.is_alive():
                p.kill()
    assert all(ok for ok, _ in results.values()), results
    assert all(p.exitcode == 0 for p in procs)
