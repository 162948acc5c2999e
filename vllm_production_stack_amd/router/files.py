"""OpenAI Files API backing store (local filesystem: ``<base>/<user>/<file-id>`` + a JSON
metadata sidecar).  Reference: /root/reference/src/vllm_router/services/files_service/
(storage.py:20-152, file_storage.py:27-136, openai_files.py:18-61)."""

from __future__ import annotations

import abc
import asyncio
import json
import os
import time
import uuid
from dataclasses import asdict, dataclass

DEFAULT_USER = "uid_default"


@dataclass
class OpenAIFile:
    id: str
    bytes: int
    created_at: int
    filename: str
    purpose: str
    object: str = "file"

    def metadata(self) -> dict:
        return asdict(self)


class Storage(abc.ABC):
    @abc.abstractmethod
    async def save_file(self, file_id=None, user_id=DEFAULT_USER, file_name=None, content=b"",
                        purpose="unknown") -> OpenAIFile: ...

    @abc.abstractmethod
    async def get_file(self, file_id: str, user_id=DEFAULT_USER) -> OpenAIFile: ...

    @abc.abstractmethod
    async def get_file_content(self, file_id: str, user_id=DEFAULT_USER) -> bytes: ...

    @abc.abstractmethod
    async def list_files(self, user_id=DEFAULT_USER) -> list[str]: ...

    @abc.abstractmethod
    async def delete_file(self, file_id: str, user_id=DEFAULT_USER) -> None: ...


class FileStorage(Storage):
    def __init__(self, base_path: str = "/tmp/vllm_files"):
        self.base_path = base_path
        os.makedirs(base_path, exist_ok=True)

    def _dir(self, user_id: str) -> str:
        safe = os.path.basename(user_id) or DEFAULT_USER
        return os.path.join(self.base_path, safe)

    def _path(self, user_id: str, file_id: str) -> str:
        if os.path.basename(file_id) != file_id or not file_id:
            raise FileNotFoundError(file_id)
        return os.path.join(self._dir(user_id), file_id)

    async def save_file(self, file_id=None, user_id=DEFAULT_USER, file_name=None, content=b"",
                        purpose="unknown") -> OpenAIFile:
        file_id = file_id or f"file-{uuid.uuid4().hex[:6]}"
        path = self._path(user_id, file_id)

        def _write():
            os.makedirs(self._dir(user_id), exist_ok=True)
            with open(path, "wb") as f:
                f.write(content)
            meta = OpenAIFile(file_id, len(content), int(time.time()), file_name or file_id, purpose)
            with open(path + ".meta.json", "w") as f:
                json.dump(meta.metadata(), f)
            return meta

        return await asyncio.get_running_loop().run_in_executor(None, _write)

    async def get_file(self, file_id: str, user_id=DEFAULT_USER) -> OpenAIFile:
        path = self._path(user_id, file_id)

        def _read():
            if not os.path.exists(path):
                raise FileNotFoundError(f"File {file_id} not found")
            if os.path.exists(path + ".meta.json"):
                with open(path + ".meta.json") as f:
                    return OpenAIFile(**json.load(f))
            st = os.stat(path)
            return OpenAIFile(file_id, st.st_size, int(st.st_mtime), file_id, "unknown")

        return await asyncio.get_running_loop().run_in_executor(None, _read)

    async def get_file_content(self, file_id: str, user_id=DEFAULT_USER) -> bytes:
        path = self._path(user_id, file_id)

        def _read():
            if not os.path.exists(path):
                raise FileNotFoundError(f"File {file_id} not found")
            with open(path, "rb") as f:
                return f.read()

        return await asyncio.get_running_loop().run_in_executor(None, _read)

    async def list_files(self, user_id=DEFAULT_USER) -> list[str]:
        d = self._dir(user_id)
        if not os.path.isdir(d):
            return []
        return sorted(f for f in os.listdir(d) if not f.endswith(".meta.json"))

    async def delete_file(self, file_id: str, user_id=DEFAULT_USER) -> None:
        path = self._path(user_id, file_id)
        if not os.path.exists(path):
            raise FileNotFoundError(f"File {file_id} not found")
        os.remove(path)
        if os.path.exists(path + ".meta.json"):
            os.remove(path + ".meta.json")


def initialize_storage(storage_class: str = "local_file", base_path: str | None = None) -> Storage:
    if storage_class == "local_file":
        return FileStorage(base_path or "/tmp/vllm_files")
    raise ValueError(f"Unsupported storage type: {storage_class}")
