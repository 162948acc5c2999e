"""Adapter download sidecar.

Runs next to an engine that has LoRA enabled and a model volume mounted, on port 30090.  The
operator POSTs ``/model/download {"model_id", "local_dir"?, "token"?, "revision"?, "allow_patterns"?}``
before it asks the engine to load an adapter; the sidecar materialises the repository below
``$LORA_DOWNLOAD_BASE_DIR`` (shared with the engine container through the volume) and answers
with the directory the engine should load from.

Behaviour provided: /root/reference/scripts/huggingface_downloader.py:17-44 (request shape,
base-directory confinement), /root/reference/helm/templates/deployment-vllm-multi.yaml:394-406
(injection next to the engine), /root/reference/operator/internal/controller/
loraadapter_controller.go:334-390 (the caller).  Additions: ``http(s)://`` archives and
``file://`` / plain directory sources are copied the same way, so clusters without egress to
the hub can still stage adapters; concurrent requests for one target are serialised.
"""

from __future__ import annotations

import asyncio
import logging
import os
import shutil
import tarfile
import tempfile
import urllib.request
import zipfile
from pathlib import Path

from aiohttp import web

log = logging.getLogger("vpsa.sidecar")

DEFAULT_BASE = "/data/lora-adapters"


class DownloadError(Exception):
    def __init__(self, status: int, message: str):
        super().__init__(message)
        self.status = status


def resolve_target(base: Path, model_id: str, local_dir: str | None) -> Path:
    """Directory for the download, guaranteed to stay below ``base``."""
    base = base.resolve()
    rel = Path(local_dir if local_dir else model_id.replace("/", "--"))
    target = (rel if rel.is_absolute() else base / rel).resolve()
    if target == base or base not in target.parents:
        raise DownloadError(400, "local_dir must name a directory below the download base directory")
    return target


def _fetch_hub(model_id: str, target: Path, token: str | None, revision: str | None,
               allow_patterns) -> None:
    try:
        from huggingface_hub import snapshot_download
    except ImportError as e:  # pragma: no cover - the image ships it
        raise DownloadError(500, f"huggingface_hub is not installed: {e}")
    snapshot_download(repo_id=model_id, local_dir=str(target), token=token or None,
                      revision=revision or None, allow_patterns=allow_patterns or None)


def _fetch_url(url: str, target: Path) -> None:
    target.mkdir(parents=True, exist_ok=True)
    with tempfile.TemporaryDirectory(dir=str(target.parent)) as tmp:
        name = url.rsplit("/", 1)[-1] or "adapter"
        dst = Path(tmp) / name
        with urllib.request.urlopen(url, timeout=600) as r, open(dst, "wb") as f:  # noqa: S310
            shutil.copyfileobj(r, f)
        if tarfile.is_tarfile(dst):
            with tarfile.open(dst) as tf:
                for m in tf.getmembers():
                    p = (target / m.name).resolve()
                    if not str(p).startswith(str(target.resolve())):
                        raise DownloadError(400, "archive member escapes the target directory")
                tf.extractall(target)  # noqa: S202 - members checked above
        elif zipfile.is_zipfile(dst):
            with zipfile.ZipFile(dst) as zf:
                for n in zf.namelist():
                    p = (target / n).resolve()
                    if not str(p).startswith(str(target.resolve())):
                        raise DownloadError(400, "archive member escapes the target directory")
                zf.extractall(target)
        else:
            shutil.move(str(dst), str(target / name))


def _fetch_dir(src: str, target: Path) -> None:
    p = Path(src.removeprefix("file://"))
    if not p.is_dir():
        raise DownloadError(404, f"source directory {p} does not exist")
    shutil.copytree(p, target, dirs_exist_ok=True)


def download(base: Path, body: dict) -> Path:
    model_id = str(body.get("model_id") or "").strip()
    if not model_id:
        raise DownloadError(400, "model_id is required")
    source_is_path = model_id.startswith(("file://", "/"))
    source_is_url = model_id.startswith(("http://", "https://"))
    default_dir = Path(model_id.removeprefix("file://")).name if (source_is_path or source_is_url) else None
    target = resolve_target(base, model_id, body.get("local_dir") or default_dir)
    target.parent.mkdir(parents=True, exist_ok=True)
    if source_is_url:
        _fetch_url(model_id, target)
    elif source_is_path:
        _fetch_dir(model_id, target)
    else:
        _fetch_hub(model_id, target, body.get("token"), body.get("revision"), body.get("allow_patterns"))
    return target


def build_app(base_dir: str | None = None) -> web.Application:
    base = Path(base_dir or os.environ.get("LORA_DOWNLOAD_BASE_DIR", DEFAULT_BASE))
    locks: dict[str, asyncio.Lock] = {}

    async def handle_download(request: web.Request) -> web.Response:
        try:
            body = await request.json()
        except Exception:
            return web.json_response({"error": "request body must be JSON"}, status=400)
        if not isinstance(body, dict):
            return web.json_response({"error": "request body must be a JSON object"}, status=400)
        try:
            key = str(resolve_target(base, str(body.get("model_id") or "x"), body.get("local_dir") or None))
        except DownloadError as e:
            return web.json_response({"error": str(e)}, status=e.status)
        lock = locks.setdefault(key, asyncio.Lock())
        async with lock:
            try:
                target = await asyncio.get_running_loop().run_in_executor(None, download, base, body)
            except DownloadError as e:
                return web.json_response({"error": str(e)}, status=e.status)
            except Exception as e:
                log.exception("download of %s failed", body.get("model_id"))
                return web.json_response({"error": f"download failed: {e}"}, status=502)
        return web.json_response({"status": "success", "model_id": body.get("model_id"),
                                  "local_dir": str(target)})

    async def handle_list(_request: web.Request) -> web.Response:
        items = sorted(p.name for p in base.iterdir() if p.is_dir()) if base.is_dir() else []
        return web.json_response({"base_dir": str(base), "adapters": items})

    async def handle_health(_request: web.Request) -> web.Response:
        return web.json_response({"status": "ok"})

    app = web.Application(client_max_size=1 << 20)
    app.router.add_post("/model/download", handle_download)
    app.router.add_get("/model/list", handle_list)
    app.router.add_get("/health", handle_health)
    return app


def main() -> None:
    logging.basicConfig(level=logging.INFO, format="%(levelname)s %(asctime)s %(name)s] %(message)s")
    port = int(os.environ.get("PORT", "30090"))
    web.run_app(build_app(), host="0.0.0.0", port=port, print=None)
