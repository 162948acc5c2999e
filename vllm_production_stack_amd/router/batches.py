"""OpenAI Batch API: create / retrieve / list / cancel, a SQLite-backed queue and a
background worker that really executes the batch.

The reference's processor only simulates work (sleep + a text file:
/root/reference/src/vllm_router/services/batch_service/local_processor.py:190-203) and its
package does not import (services/batch_service/__init__.py:9).  Here each JSONL line
``{"custom_id", "method", "url", "body"}`` of the input file is dispatched through the
router's own routing logic to an engine; results are written as an OpenAI-style output
JSONL file registered in the Files store.  Schema reference: batch.py:19-104."""

from __future__ import annotations

import asyncio
import enum
import json
import sqlite3
import time
import uuid
from dataclasses import asdict, dataclass, field

from .files import Storage
from .log import init_logger

logger = init_logger(__name__)


class BatchStatus(str, enum.Enum):
    VALIDATING = "validating"
    FAILED = "failed"
    IN_PROGRESS = "in_progress"
    FINALIZING = "finalizing"
    COMPLETED = "completed"
    EXPIRED = "expired"
    CANCELLING = "cancelling"
    CANCELLED = "cancelled"


class BatchEndpoint(str, enum.Enum):
    CHAT_COMPLETION = "/v1/chat/completions"
    EMBEDDING = "/v1/embeddings"
    COMPLETION = "/v1/completions"


@dataclass
class BatchInfo:
    id: str
    status: str
    input_file_id: str
    created_at: int
    endpoint: str
    completion_window: str = "5s"
    output_file_id: str | None = None
    error_file_id: str | None = None
    in_progress_at: int | None = None
    completed_at: int | None = None
    failed_at: int | None = None
    cancelled_at: int | None = None
    metadata: dict | None = None
    request_counts: dict = field(default_factory=lambda: {"total": 0, "completed": 0, "failed": 0})
    object: str = "batch"

    def to_dict(self) -> dict:
        return asdict(self)


class LocalBatchProcessor:
    """``dispatch(url_path, body) -> awaitable[(status, json)]`` is supplied by the app and
    sends one request through the routing logic."""

    def __init__(self, db_dir: str, storage: Storage, dispatch=None, poll_interval: float = 1.0):
        self.db_path = f"{db_dir.rstrip('/')}/batch_queue.db"
        self.storage = storage
        self.dispatch = dispatch
        self.poll_interval = poll_interval
        self._task: asyncio.Task | None = None
        self._lock = asyncio.Lock()
        self._db: sqlite3.Connection | None = None

    # -- db --------------------------------------------------------------------------
    def _conn(self) -> sqlite3.Connection:
        if self._db is None:
            import os

            os.makedirs(self.db_path.rsplit("/", 1)[0], exist_ok=True)
            self._db = sqlite3.connect(self.db_path, check_same_thread=False)
            self._db.execute("CREATE TABLE IF NOT EXISTS batch_queue ("
                             "batch_id TEXT PRIMARY KEY, status TEXT, created_at INTEGER, info TEXT)")
            self._db.commit()
        return self._db

    def _put(self, info: BatchInfo) -> None:
        c = self._conn()
        c.execute("INSERT OR REPLACE INTO batch_queue VALUES (?, ?, ?, ?)",
                  (info.id, info.status, info.created_at, json.dumps(info.to_dict())))
        c.commit()

    def _get(self, batch_id: str) -> BatchInfo | None:
        row = self._conn().execute("SELECT info FROM batch_queue WHERE batch_id = ?", (batch_id,)).fetchone()
        return BatchInfo(**json.loads(row[0])) if row else None

    # -- lifecycle ---------------------------------------------------------------------
    async def initialize(self, run_queue: bool = True) -> None:
        """``run_queue=False``: this process only serves the API (another router worker on the
        same queue file executes the batches)."""
        self._conn()
        if run_queue and self._task is None:
            self._task = asyncio.create_task(self._worker())

    async def close(self) -> None:
        if self._task:
            self._task.cancel()
            self._task = None
        if self._db is not None:
            self._db.close()
            self._db = None

    # -- API ---------------------------------------------------------------------------
    async def create_batch(self, input_file_id: str, endpoint: str, completion_window: str = "5s",
                           metadata: dict | None = None) -> BatchInfo:
        if endpoint not in [e.value for e in BatchEndpoint]:
            raise ValueError(f"Unsupported batch endpoint: {endpoint}")
        await self.storage.get_file(input_file_id)  # FileNotFoundError when absent
        info = BatchInfo(id=f"batch_{uuid.uuid4().hex[:6]}", status=BatchStatus.VALIDATING.value,
                         input_file_id=input_file_id, created_at=int(time.time()), endpoint=endpoint,
                         completion_window=completion_window, metadata=metadata)
        async with self._lock:
            self._put(info)
        return info

    async def retrieve_batch(self, batch_id: str) -> BatchInfo:
        info = self._get(batch_id)
        if info is None:
            raise FileNotFoundError(f"Batch {batch_id} not found")
        return info

    async def list_batches(self, limit: int = 20, after: str | None = None) -> list[BatchInfo]:
        rows = self._conn().execute(
            "SELECT info FROM batch_queue ORDER BY created_at DESC, batch_id DESC").fetchall()
        infos = [BatchInfo(**json.loads(r[0])) for r in rows]
        if after:
            ids = [i.id for i in infos]
            if after in ids:
                infos = infos[ids.index(after) + 1:]
        return infos[:limit]

    async def cancel_batch(self, batch_id: str) -> BatchInfo:
        async with self._lock:
            info = await self.retrieve_batch(batch_id)
            if info.status in (BatchStatus.VALIDATING.value, BatchStatus.IN_PROGRESS.value):
                info.status = BatchStatus.CANCELLED.value
                info.cancelled_at = int(time.time())
                self._put(info)
        return info

    # -- worker ------------------------------------------------------------------------
    async def _worker(self) -> None:
        while True:
            try:
                row = self._conn().execute(
                    "SELECT batch_id FROM batch_queue WHERE status = ? ORDER BY created_at LIMIT 1",
                    (BatchStatus.VALIDATING.value,)).fetchone()
                if row:
                    await self._process(row[0])
                else:
                    await asyncio.sleep(self.poll_interval)
            except asyncio.CancelledError:
                raise
            except Exception as e:
                logger.error("batch worker error: %s", e)
                await asyncio.sleep(self.poll_interval)

    async def _process(self, batch_id: str) -> None:
        info = self._get(batch_id)
        if info is None:
            return
        try:
            raw = await self.storage.get_file_content(info.input_file_id)
            lines = [json.loads(ln) for ln in raw.decode().splitlines() if ln.strip()]
        except Exception as e:
            info.status = BatchStatus.FAILED.value
            info.failed_at = int(time.time())
            info.metadata = {**(info.metadata or {}), "error": f"invalid input file: {e}"}
            self._put(info)
            return
        info.status = BatchStatus.IN_PROGRESS.value
        info.in_progress_at = int(time.time())
        info.request_counts = {"total": len(lines), "completed": 0, "failed": 0}
        self._put(info)
        outs, errs = [], []
        for i, item in enumerate(lines):
            cur = self._get(batch_id)
            if cur is not None and cur.status == BatchStatus.CANCELLED.value:
                return
            cid = item.get("custom_id", f"request-{i}")
            url = item.get("url", info.endpoint)
            body = item.get("body", {})
            try:
                if self.dispatch is None:
                    raise RuntimeError("no dispatcher configured")
                status, payload = await self.dispatch(url, body)
            except Exception as e:
                status, payload = 500, {"error": str(e)}
            rec = {"id": f"batch_req_{uuid.uuid4().hex[:12]}", "custom_id": cid,
                   "response": {"status_code": status, "request_id": uuid.uuid4().hex, "body": payload},
                   "error": None}
            if 200 <= status < 300:
                info.request_counts["completed"] += 1
                outs.append(rec)
            else:
                info.request_counts["failed"] += 1
                rec["error"] = {"code": str(status), "message": json.dumps(payload)[:500]}
                errs.append(rec)
        info.status = BatchStatus.FINALIZING.value
        self._put(info)
        out_file = await self.storage.save_file(
            file_name=f"{batch_id}_output.jsonl", purpose="batch_output",
            content=("\n".join(json.dumps(r) for r in outs) + ("\n" if outs else "")).encode())
        info.output_file_id = out_file.id
        if errs:
            err_file = await self.storage.save_file(
                file_name=f"{batch_id}_errors.jsonl", purpose="batch_output",
                content=("\n".join(json.dumps(r) for r in errs) + "\n").encode())
            info.error_file_id = err_file.id
        info.status = BatchStatus.COMPLETED.value
        info.completed_at = int(time.time())
        self._put(info)
        logger.info("batch %s completed: %s", batch_id, info.request_counts)


def initialize_batch_processor(kind: str, storage_path: str, storage: Storage, dispatch=None):
    if kind == "local":
        return LocalBatchProcessor(storage_path, storage, dispatch)
    raise ValueError(f"Unknown batch processor type: {kind}")
