"""Thread-hosted engine loop with an asyncio front.

The GPU step loop owns one OS thread (it blocks in HIP syncs and must never share the
event loop); HTTP handlers submit work through a lock-free deque and receive step
outputs on per-request ``asyncio.Queue``s, woken once per step with a single
``call_soon_threadsafe`` for the whole batch.
"""

from __future__ import annotations

import asyncio
import logging
import threading
import time
from collections import deque
from typing import AsyncIterator

from .llm_engine import LLMEngine
from .request import SamplingParams, StepOutput

log = logging.getLogger("vpsa.async")


class AsyncLLMEngine:
    def __init__(self, engine: LLMEngine):
        self.engine = engine
        self._inbox: deque = deque()
        self._wake = threading.Event()
        self._stop = False
        self._thread: threading.Thread | None = None
        self._loop: asyncio.AbstractEventLoop | None = None
        self._queues: dict[str, asyncio.Queue] = {}
        self._errors: dict[str, Exception] = {}
        self._live: dict[str, object] = {}  # request id -> Request (for pooled results)
        self.healthy = True
        self.fatal: Exception | None = None  # set when the engine thread has given up (see _run)
        self._control_lock = threading.Lock()

    def start(self, loop: asyncio.AbstractEventLoop | None = None) -> None:
        self._loop = loop or asyncio.get_event_loop()
        self._thread = threading.Thread(target=self._run, name="engine-loop", daemon=True)
        self._thread.start()

    def stop(self) -> None:
        self._stop = True
        self._wake.set()
        if self._thread:
            self._thread.join(timeout=10)
        self.engine.shutdown()

    # ------------------------------------------------------------------ engine thread
    def _deliver(self, batch: list[tuple[str, object]]) -> None:
        for rid, item in batch:
            q = self._queues.get(rid)
            if q is not None:
                q.put_nowait(item)

    def _run(self) -> None:
        eng = self.engine
        dev = getattr(eng, "device", None)
        if dev is not None and dev.type == "cuda":
            # the HIP current device is per thread and starts at 0: without this, streams, graph
            # captures and kernel launches of this thread would land on GPU 0 on every rank
            import torch

            torch.cuda.set_device(dev)
        while not self._stop:
            try:
                while self._inbox:
                    kind, payload = self._inbox.popleft()
                    if kind == "add":
                        rid, ids, params, lora, kvp, pooling = payload
                        try:
                            req = eng.add_request(ids, params, rid, lora, kvp, pooling)
                            self._live[rid] = req
                        except Exception as e:  # reported to the waiting handler
                            self._loop.call_soon_threadsafe(self._deliver, [(rid, e)])
                    elif kind == "abort":
                        eng.abort(payload)
                    elif kind == "call":
                        fn, fut = payload
                        try:
                            res = fn()
                            self._loop.call_soon_threadsafe(fut.set_result, res)
                        except Exception as e:
                            self._loop.call_soon_threadsafe(fut.set_exception, e)
                if eng.has_work() and not eng.is_sleeping:
                    outs = eng.step()
                    if outs:
                        self._loop.call_soon_threadsafe(
                            self._deliver, [(o.request_id, o) for o in outs])
                else:
                    eng.idle_tick()
                    self._wake.wait(timeout=0.05)
                    self._wake.clear()
            except Exception as e:  # engine failure: fail every in-flight request
                log.exception("engine step failed")
                self.healthy = False
                batch = [(rid, e) for rid in list(self._queues)]
                self._loop.call_soon_threadsafe(self._deliver, batch)
                for rid in list(eng.requests):
                    try:
                        eng.abort(rid)
                    except Exception:
                        pass
                if type(e).__name__ == "CollectiveError":
                    # the ranks of this replica no longer agree on where they are: no more steps;
                    # /health answers 503 from now on and the supervisor restarts the replica
                    self.fatal = e
                    log.error("replica failed for good: %s", e)
                    return
                time.sleep(0.1)
                self.healthy = True

    # ------------------------------------------------------------------ asyncio side
    async def generate(self, request_id: str, prompt_ids: list[int], params: SamplingParams,
                       lora_name: str | None = None, kv_transfer_params: dict | None = None,
                       pooling: bool = False) -> AsyncIterator[StepOutput]:
        if self.fatal is not None:  # the engine thread has ended: nothing would ever answer
            raise RuntimeError(f"engine replica failed: {self.fatal}")
        q: asyncio.Queue = asyncio.Queue()
        self._queues[request_id] = q
        self._inbox.append(("add", (request_id, prompt_ids, params, lora_name, kv_transfer_params, pooling)))
        self._wake.set()
        try:
            while True:
                item = await q.get()
                if isinstance(item, Exception):
                    raise item
                yield item
                if item.finished:
                    return
        finally:
            self._queues.pop(request_id, None)
            self._live.pop(request_id, None)
            # always: the add may still be in the inbox (a client that went away at once), and an
            # abort of an unknown or finished id is a no-op on the engine thread
            self._inbox.append(("abort", request_id))
            self._wake.set()

    def live_request(self, request_id: str):
        return self._live.get(request_id)

    async def call(self, fn):
        """Run ``fn`` on the engine thread (between steps) and await its result."""
        fut = asyncio.get_running_loop().create_future()
        self._inbox.append(("call", (fn, fut)))
        self._wake.set()
        return await fut
