"""Control flow of the disaggregated-prefill data plane on the CPU.

The device transports (hipIpc landing window, RCCL send/recv) are driven through a stub of the
``vpsa_comm`` library that moves bytes with ``ctypes.memmove`` between CPU tensors, so that the
protocol around them is covered without a GPU: reserve -> push -> commit in rounds through a
window smaller than the prompt, the RCCL unique-id exchange and its recv-before-send order, the
automatic choice of RCCL when no window is offered, the fallback after a failed communicator
setup, and the refusal of pages that are not reserved.  What the stub cannot show (the copies
themselves, RCCL over xGMI) is measured on hardware by benchmarks/pd_e2e.py and pd_pools.py.
"""

import asyncio
import copy
import ctypes
import json
import threading

import aiohttp
import pytest
import torch

from vllm_production_stack_amd.engine.async_engine import AsyncLLMEngine
from vllm_production_stack_amd.engine.block_manager import chain_hashes
from vllm_production_stack_amd.engine.config import EngineConfig, KVTransferConfig
from vllm_production_stack_amd.engine.llm_engine import LLMEngine
from vllm_production_stack_amd.engine.request import SamplingParams
from vllm_production_stack_amd.engine.server import serve
from vllm_production_stack_amd.parallel import kv_transfer as kvx

SP = SamplingParams(max_tokens=4, temperature=0, ignore_eos=True)


def _v(x):
    return x.value if hasattr(x, "value") else int(x)


class StubComm:
    """The subset of libvpsa_comm the P/D data plane calls, on host memory."""

    def __init__(self, export_ok=True, rccl_ok=True):
        self.export_ok, self.rccl_ok = export_ok, rccl_ok
        self.allocs: dict[int, ctypes.Array] = {}
        self.log: list[str] = []
        self._uid = 0
        self._comms: dict[bytes, dict] = {}
        self._by_handle: dict[int, dict] = {}
        self._lock = threading.Lock()
        self._cv = threading.Condition()

    # hipIpc: a "handle" is the address itself (both replicas live in this process)
    def comm_ipc_handle_size(self):
        return 8

    def comm_ipc_export(self, ptr, buf):
        if not self.export_ok:
            return 1
        ctypes.memmove(buf, _v(ptr).to_bytes(8, "little"), 8)
        return 0

    def comm_ipc_open(self, buf):
        self.log.append("ipc_open")
        return int.from_bytes(buf.raw[:8], "little")

    def comm_ipc_close(self, ptr):
        self.log.append("ipc_close")

    def comm_dev_alloc(self, nbytes):
        b = ctypes.create_string_buffer(_v(nbytes))
        self.allocs[ctypes.addressof(b)] = b
        return ctypes.addressof(b)

    def comm_dev_free(self, ptr):
        self.allocs.pop(_v(ptr), None)

    def comm_push_pages(self, src, dst, page_bytes, sp, dp, n, stream):
        pb = _v(page_bytes)
        for i in range(n):
            ctypes.memmove(_v(dst) + dp[i] * pb, _v(src) + sp[i] * pb, pb)
        self.log.append(f"push_pages:{n}")
        return 0

    # RCCL: two ranks meet on a unique id; a send must find its receive posted
    def comm_rccl_unique_id_size(self):
        return 16

    def comm_rccl_unique_id(self, out):
        with self._lock:
            self._uid += 1
            ctypes.memmove(out, self._uid.to_bytes(16, "little"), 16)
        return 0

    def comm_rccl_init(self, uid, nranks, rank, device):
        if not self.rccl_ok:
            return 0
        key = bytes(uid.raw[:16])
        with self._cv:
            c = self._comms.setdefault(key, {"n": 0, "recv": [], "id": len(self._comms) + 1})
            c["n"] += 1
            self._by_handle[c["id"]] = c
            self._cv.notify_all()
            # the rendezvous: returns when the other rank has arrived as well
            if not self._cv.wait_for(lambda: c["n"] >= _v(nranks), timeout=5):
                return 0
        self.log.append(f"rccl_init:rank{_v(rank)}")
        return c["id"]

    def comm_rccl_recv_pages(self, c, base, page_bytes, pages, n, peer, stream):
        self._by_handle[_v(c)]["recv"].append((_v(base), _v(page_bytes), [pages[i] for i in range(n)]))
        self.log.append(f"rccl_recv:{n}")
        return 0

    def comm_rccl_send_pages(self, c, base, page_bytes, pages, n, peer, stream):
        q = self._by_handle[_v(c)]["recv"]
        if not q:
            self.log.append("rccl_send_without_recv")
            return 5
        dbase, pb, dpages = q.pop(0)
        assert pb == _v(page_bytes) and len(dpages) == n
        for i in range(n):
            ctypes.memmove(dbase + dpages[i] * pb, _v(base) + pages[i] * pb, pb)
        self.log.append(f"rccl_send:{n}")
        return 0

    def comm_bind_device(self, d):
        pass


@pytest.fixture
def stub(monkeypatch):
    s = StubComm()
    monkeypatch.setattr(kvx, "_comm", s)
    monkeypatch.setattr(kvx, "_FAKE_DEVICE", True)
    monkeypatch.setattr(kvx, "_peer_maps", {})
    monkeypatch.setattr(kvx, "_rccl_peers", {})
    monkeypatch.setattr(kvx, "_rccl_broken", set())
    return s


def _engine(role, pages=32):
    return LLMEngine(EngineConfig(model="tiny-llama", device="cpu", dtype="float32", max_model_len=1024,
                                  num_gpu_blocks_override=pages, host="127.0.0.1", port=0,
                                  kv_transfer=KVTransferConfig("LMCacheConnectorV1", role)))


def _prompt(n=330, seed=4):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(16, 500, (n,), generator=g).tolist()


async def _pd_pair():
    dec = _engine("kv_consumer")
    dae = AsyncLLMEngine(dec)
    dae.start(asyncio.get_running_loop())
    drunner = await serve(dae, "127.0.0.1", 0)
    return dec, dae, drunner, f"http://127.0.0.1:{drunner.addresses[0][1]}"


def _prefill_and_push(pre, prompt, target):
    sp = SamplingParams(max_tokens=1, temperature=0, ignore_eos=True)
    req = pre.add_request(prompt, sp, kv_transfer_params={"do_remote_decode": True, "decode_target": target})
    steps = 0
    while pre.has_work():
        pre.step()
        steps += 1
    fut = pre.connector.push_future(req.request_id)
    return fut.result(30), steps


async def _decode_matches(dae, prompt):
    out, cached = [], 0
    async for so in dae.generate("d-" + str(len(prompt)), prompt, copy.deepcopy(SP)):
        out.extend(so.new_token_ids)
        cached = so.num_cached_tokens
    ref = _engine("kv_both").generate([prompt], copy.deepcopy(SP))[0]
    return out == ref, cached


def test_landing_window_in_rounds_and_publish_after_copy(stub, monkeypatch):
    """A window of two page slots carries a five-page prompt in three rounds; the decode replica
    publishes pages only through begin -> wait -> finish, and ends with every slot free."""
    async def go():
        dec, dae, drunner, url = await _pd_pair()
        page_bytes = dec.runner.pool.stride(0) * dec.runner.pool.element_size()
        monkeypatch.setattr(kvx, "ipc_safe_size", lambda n: 2 * page_bytes)  # a window of two slots
        pre = _engine("kv_producer")
        a = _prompt()
        res, _ = await asyncio.get_running_loop().run_in_executor(None, _prefill_and_push, pre, a, url)
        assert res["mode"] == "ipc" and res["pages"] == 5
        assert stub.log.count("ipc_open") == 1  # the window is mapped once per (target, handle)
        assert dec.bm.peek_prefix(chain_hashes(a)) == 5
        same, cached = await _decode_matches(dae, a)
        assert same and cached == 320
        await drunner.cleanup()
        dae.stop()
        pre.shutdown()

    asyncio.run(go())


def test_rccl_when_forced_recv_is_posted_before_send(stub, monkeypatch):
    monkeypatch.setenv("VPSA_PD_TRANSPORT", "rccl")

    async def go():
        dec, dae, drunner, url = await _pd_pair()
        pre = _engine("kv_producer")
        a, b = _prompt(seed=5), _prompt(seed=6)
        loop = asyncio.get_running_loop()
        res, _ = await loop.run_in_executor(None, _prefill_and_push, pre, a, url)
        assert res["mode"] == "rccl"
        order = [x for x in stub.log if x.startswith("rccl_")]
        assert sorted(order[:2]) == ["rccl_init:rank0", "rccl_init:rank1"]
        assert order[2:] == ["rccl_recv:5", "rccl_send:5"] and "rccl_send_without_recv" not in stub.log
        # the communicator of a (prefill, decode) pair is created once
        res, _ = await loop.run_in_executor(None, _prefill_and_push, pre, b, url)
        assert res["mode"] == "rccl" and sum(x.startswith("rccl_init") for x in stub.log) == 2
        same, cached = await _decode_matches(dae, a)
        assert same and cached == 320
        await drunner.cleanup()
        dae.stop()
        pre.shutdown()

    asyncio.run(go())


def test_rccl_is_chosen_when_no_window_is_offered(stub):
    """transport = auto, both replicas on a device, the decode replica cannot export a window
    (another host or namespace looks the same to the producer: no slots in the answer)."""
    stub.export_ok = False

    async def go():
        dec, dae, drunner, url = await _pd_pair()
        pre = _engine("kv_producer")
        res, _ = await asyncio.get_running_loop().run_in_executor(None, _prefill_and_push, pre, _prompt(seed=7), url)
        assert res["mode"] == "rccl"
        await drunner.cleanup()
        dae.stop()
        pre.shutdown()

    asyncio.run(go())


def test_failed_communicator_setup_falls_back_to_tcp_once(stub, monkeypatch):
    stub.export_ok, stub.rccl_ok = False, False

    async def go():
        dec, dae, drunner, url = await _pd_pair()
        pre = _engine("kv_producer")
        loop = asyncio.get_running_loop()
        a, b = _prompt(seed=8), _prompt(seed=9)
        res, _ = await loop.run_in_executor(None, _prefill_and_push, pre, a, url)
        assert res["mode"] == "tcp" and url in kvx._rccl_broken
        res, _ = await loop.run_in_executor(None, _prefill_and_push, pre, b, url)
        assert res["mode"] == "tcp"  # and no second attempt at the communicator
        same, cached = await _decode_matches(dae, b)
        assert same and cached == 320
        await drunner.cleanup()
        dae.stop()
        pre.shutdown()

    asyncio.run(go())


def test_unreserved_pages_and_malformed_bodies_are_refused():
    async def go():
        dec, dae, drunner, url = await _pd_pair()
        pool = dec.runner.pool
        page_bytes = pool.stride(0) * pool.element_size()
        before = pool.clone()
        async with aiohttp.ClientSession() as s:
            meta = json.dumps({"pages": [3], "page_bytes": page_bytes}).encode()
            body = len(meta).to_bytes(4, "little") + meta + b"\x01" * page_bytes
            async with s.post(url + "/kv/push", data=body) as r:
                assert r.status == 409  # page 3 was never reserved
            async with s.post(url + "/kv/push", data=b"\xff\xff") as r:
                assert r.status == 409  # not even a header
            async with s.post(url + "/kv/commit", json={"hashes": [1], "pages": [3]}) as r:
                assert r.status == 409
            async with s.post(url + "/kv/reserve", json={"hashes": "x", "page_bytes": page_bytes}) as r:
                assert r.status == 409
            # a real reservation accepts exactly its pages, with exactly their bytes
            async with s.post(url + "/kv/reserve", json={"hashes": [11, 12], "page_bytes": page_bytes}) as r:
                res = await r.json()
            meta = json.dumps({"pages": res["pages"], "page_bytes": page_bytes}).encode()
            async with s.post(url + "/kv/push", data=len(meta).to_bytes(4, "little") + meta + b"\x01" * page_bytes) as r:
                assert r.status == 409  # two pages announced, one page of bytes
            async with s.post(url + "/kv/push",
                              data=len(meta).to_bytes(4, "little") + meta + b"\x00" * (2 * page_bytes)) as r:
                assert r.status == 200
            async with s.post(url + "/kv/commit", json={"hashes": [11, 12], "pages": res["pages"]}) as r:
                assert r.status == 200 and (await r.json())["published"] == 2
        untouched = [p for p in range(pool.shape[0]) if p not in res["pages"]]
        assert torch.equal(pool[untouched], before[untouched])
        await drunner.cleanup()
        dae.stop()

    asyncio.run(go())


def test_push_target_allow_list(monkeypatch):
    assert kvx.target_allowed("http://10.0.0.5:8000")
    assert not kvx.target_allowed("file:///etc/passwd") and not kvx.target_allowed("gopher://x")
    monkeypatch.setenv("VPSA_PD_ALLOWED_TARGETS", "http://decode-0:8000,10.1.2.3")
    assert kvx.target_allowed("http://decode-0:8000") and kvx.target_allowed("http://10.1.2.3:9000")
    assert not kvx.target_allowed("http://evil.example:8000")


def test_step_loop_does_not_wait_for_a_slow_decode_replica(monkeypatch):
    """A decode replica that answers /kv/reserve after 1.5 s delays the push, not the steps of the
    prefill replica: another request is served to completion while the transfer is pending."""
    import time

    real_post = kvx._post

    def slow_post(url, payload, raw=None, timeout=10.0):
        if url.endswith("/kv/reserve"):
            time.sleep(1.5)
        return real_post(url, payload, raw, timeout)

    monkeypatch.setattr(kvx, "_post", slow_post)

    async def go():
        dec, dae, drunner, url = await _pd_pair()
        pre = _engine("kv_producer")

        def run():
            sp1 = SamplingParams(max_tokens=1, temperature=0, ignore_eos=True)
            r1 = pre.add_request(_prompt(seed=10), sp1,
                                 kv_transfer_params={"do_remote_decode": True, "decode_target": url})
            while pre.has_work():
                pre.step()
            t0 = time.time()
            other = pre.generate([_prompt(100, seed=11)], copy.deepcopy(SP))[0]
            dt = time.time() - t0
            fut = pre.connector.push_future(r1.request_id)
            pending_during_other = not fut.done()
            return other, dt, pending_during_other, fut.result(30)

        other, dt, pending, res = await asyncio.get_running_loop().run_in_executor(None, run)
        assert len(other) == 4 and dt < 1.0 and pending and res["mode"] == "tcp"
        await drunner.cleanup()
        dae.stop()
        pre.shutdown()

    asyncio.run(go())


def test_pool_exported_pages_land_where_they_live(stub):
    """A decode replica whose pool is one exportable allocation hands out the pool's own handle:
    the producer writes every page into its final place in ONE round (no window slots, no
    commit-time copy on the decode replica) and the mode is reported as ipc-direct."""
    async def go():
        dec, dae, drunner, url = await _pd_pair()
        dec.runner.pool_ipc_handle = dec.runner.pool.data_ptr().to_bytes(8, "little").hex()  # what the stub maps
        pre = _engine("kv_producer")
        a = _prompt(seed=21)
        res, _ = await asyncio.get_running_loop().run_in_executor(None, _prefill_and_push, pre, a, url)
        assert res["mode"] == "ipc-direct" and res["pages"] == 5
        # one copy call per push job on the producer, none on the decode replica
        assert sum(int(x.split(":")[1]) for x in stub.log if x.startswith("push_pages")) == 5
        assert dec.bm.peek_prefix(chain_hashes(a)) == 5
        same, cached = await _decode_matches(dae, a)
        assert same and cached == 320
        await drunner.cleanup()
        dae.stop()
        pre.shutdown()

    asyncio.run(go())


def test_pages_leave_chunk_by_chunk_while_the_prompt_is_still_being_prefilled():
    """Chunked prefill of a 330-token prompt in chunks of 128 tokens: the pages a chunk completes
    are handed to the push worker right after that chunk (pages are immutable once full), so the
    transfer has several jobs and only the last chunk's pages are pushed after the last step;
    the future the HTTP handler awaits completes after the LAST job, with the total."""
    async def go():
        dec, dae, drunner, url = await _pd_pair()
        pre = LLMEngine(EngineConfig(model="tiny-llama", device="cpu", dtype="float32", max_model_len=1024,
                                     num_gpu_blocks_override=32, host="127.0.0.1", port=0, max_num_batched_tokens=128,
                                     kv_transfer=KVTransferConfig("LMCacheConnectorV1", "kv_producer")))
        a = _prompt(seed=22)

        def run():
            sp = SamplingParams(max_tokens=1, temperature=0, ignore_eos=True)
            req = pre.add_request(a, sp, kv_transfer_params={"do_remote_decode": True, "decode_target": url})
            seen = []
            while pre.has_work():
                pre.step()
                seen.append(getattr(req, "_pushed_pages", 0))
            return req, seen, pre.connector.push_future(req.request_id).result(30)

        req, seen, res = await asyncio.get_running_loop().run_in_executor(None, run)
        assert seen == [2, 4, 5], seen  # after each of the three chunks: 128, 256, 330 tokens computed
        assert res["mode"] == "tcp" and res["pages"] == 5 and res["jobs"] == 4  # three chunks + the closing job
        assert dec.bm.peek_prefix(chain_hashes(a)) == 5
        same, cached = await _decode_matches(dae, a)
        assert same and cached == 320
        pre.idle_tick()
        assert pre.bm.num_free() == 32 and not any(pre.bm.ref)
        await drunner.cleanup()
        dae.stop()
        pre.shutdown()

    asyncio.run(go())


def test_a_dead_target_delays_only_its_own_pushes_and_is_skipped_for_a_while(monkeypatch):
    """Two decode targets, one of them unreachable: the transfer to the live one completes while
    the dead one's job is still waiting for its connection to fail (one worker per target), and
    the next push to the dead target fails at once instead of waiting again."""
    import time

    real_post = kvx._post

    def post(url, payload, raw=None, timeout=10.0):
        if "127.0.0.1:9/" in url:
            time.sleep(1.0)
            raise ConnectionError("connection refused")
        return real_post(url, payload, raw, timeout)

    monkeypatch.setattr(kvx, "_post", post)

    async def go():
        dec, dae, drunner, url = await _pd_pair()
        pre = _engine("kv_producer")
        dead = "http://127.0.0.1:9"

        def run():
            sp = SamplingParams(max_tokens=1, temperature=0, ignore_eos=True)
            r_dead = pre.add_request(_prompt(seed=23), sp, kv_transfer_params={"do_remote_decode": True, "decode_target": dead})
            r_live = pre.add_request(_prompt(seed=24), copy.deepcopy(sp), kv_transfer_params={"do_remote_decode": True, "decode_target": url})
            while pre.has_work():
                pre.step()
            f_dead, f_live = pre.connector.push_future(r_dead.request_id), pre.connector.push_future(r_live.request_id)
            t0 = time.time()
            live = f_live.result(30)
            t_live = time.time() - t0
            dead_pending = not f_dead.done()
            with pytest.raises(ConnectionError):
                f_dead.result(30)
            # within the cool-down the worker does not try again
            r2 = pre.add_request(_prompt(seed=25), copy.deepcopy(sp), kv_transfer_params={"do_remote_decode": True, "decode_target": dead})
            while pre.has_work():
                pre.step()
            t0 = time.time()
            with pytest.raises(ConnectionError, match="skipped"):
                pre.connector.push_future(r2.request_id).result(30)
            return live, t_live, dead_pending, time.time() - t0

        live, t_live, dead_pending, t_skip = await asyncio.get_running_loop().run_in_executor(None, run)
        assert live["mode"] == "tcp" and t_live < 0.9 and dead_pending and t_skip < 0.5
        pre.idle_tick()
        assert not any(pre.bm.ref)
        await drunner.cleanup()
        dae.stop()
        pre.shutdown()

    asyncio.run(go())


def test_stuck_rccl_receive_quarantines_its_pages():
    """A receive whose sender died: the pages stay out of the allocator until the receive's event
    fires (a late sender may still write them), the communicator is dropped, other transfers'
    events are their own."""
    dec = _engine("kv_consumer", pages=16)
    recv = kvx.KVReceiver(dec)
    page_bytes = dec.runner.pool.stride(0) * dec.runner.pool.element_size()

    class Ev:
        def __init__(self, done):
            self.done = done

        def query(self):
            return self.done

    free0 = dec.bm.num_free()
    a = recv.reserve([101, 102], page_bytes)
    b = recv.reserve([201], page_bytes)
    stuck, fine = Ev(False), Ev(True)
    recv._rccl["k1"], recv._rccl["k2"] = 1, 2
    recv._rccl_events[tuple(a["pages"])] = (stuck, "k1")
    recv._rccl_events[tuple(b["pages"])] = (fine, "k2")
    assert recv.commit_begin([201], b["pages"], rccl=True) is fine  # its own event, not "the last one posted"
    recv.commit_finish([201], b["pages"])
    recv.commit_abort([101, 102])
    assert dec.bm.num_free() == free0 - 2 and "k1" not in recv._rccl and "k2" in recv._rccl
    recv.reserve([301], page_bytes)  # sweeps: still stuck
    assert dec.bm.num_free() == free0 - 3
    stuck.done = True
    recv.reserve([302], page_bytes)
    assert dec.bm.num_free() == free0 - 2  # the two quarantined pages are back, 301 and 302 are reserved


def test_pushed_prompts_end_in_a_short_chunk(monkeypatch):
    """Only the last chunk's pages travel after the prefill has ended, so a prompt whose pages are
    pushed is cut so that its last chunk is short (here 64 tokens: 330 = 256 + 74, page-aligned),
    even when the whole prompt would fit one step; a request that is not pushed runs whole."""
    from vllm_production_stack_amd.engine.scheduler import Scheduler

    monkeypatch.setattr(Scheduler, "PUSH_TAIL_TOKENS", 64)

    async def go():
        dec, dae, drunner, url = await _pd_pair()
        pre = LLMEngine(EngineConfig(model="tiny-llama", device="cpu", dtype="float32", max_model_len=1024,
                                     num_gpu_blocks_override=32, host="127.0.0.1", port=0, max_num_batched_tokens=512,
                                     kv_transfer=KVTransferConfig("LMCacheConnectorV1", "kv_producer")))

        def run(prompt, kvp):
            sp = SamplingParams(max_tokens=1, temperature=0, ignore_eos=True)
            req = pre.add_request(prompt, sp, kv_transfer_params=kvp)
            rows, r0 = [], pre.prof["rows"]
            while pre.has_work():
                pre.step()
                rows.append(pre.prof["rows"] - r0)
                r0 = pre.prof["rows"]
            if kvp:
                pre.connector.push_future(req.request_id).result(30)
            return rows, req.output_ids

        loop = asyncio.get_running_loop()
        pushed, out_a = await loop.run_in_executor(None, run, _prompt(seed=31), {"do_remote_decode": True, "decode_target": url})
        plain, out_b = await loop.run_in_executor(None, run, _prompt(seed=32), None)
        assert pushed == [256, 74], pushed
        assert plain == [330], plain
        same, cached = await _decode_matches(dae, _prompt(seed=31))
        assert same and cached == 320
        await drunner.cleanup()
        dae.stop()
        pre.shutdown()

    asyncio.run(go())
