"""Adapter download sidecar: request validation, base-directory confinement, local sources,
and the hub path with the hub client stubbed (there is no network in the test environment)."""

import asyncio

import pytest
from aiohttp.test_utils import TestClient, TestServer

from vllm_production_stack_amd import sidecar


def _run(coro):
    return asyncio.new_event_loop().run_until_complete(coro)


async def _client(base):
    c = TestClient(TestServer(sidecar.build_app(str(base))))
    await c.start_server()
    return c


def test_traversal_is_rejected(tmp_path):
    with pytest.raises(sidecar.DownloadError):
        sidecar.resolve_target(tmp_path, "org/repo", "../outside")
    with pytest.raises(sidecar.DownloadError):
        sidecar.resolve_target(tmp_path, "org/repo", "/etc")
    assert sidecar.resolve_target(tmp_path, "org/repo", None) == (tmp_path / "org--repo").resolve()
    assert sidecar.resolve_target(tmp_path, "org/repo", "a/b") == (tmp_path / "a" / "b").resolve()


def test_download_from_directory_and_list(tmp_path):
    src = tmp_path / "src" / "sql-lora"
    src.mkdir(parents=True)
    (src / "adapter_config.json").write_text("{}")
    base = tmp_path / "base"

    async def go():
        c = await _client(base)
        r = await c.post("/model/download", json={"model_id": f"file://{src}"})
        assert r.status == 200
        body = await r.json()
        assert body["status"] == "success" and body["local_dir"] == str((base / "sql-lora").resolve())
        assert (base / "sql-lora" / "adapter_config.json").exists()
        r = await c.get("/model/list")
        assert (await r.json())["adapters"] == ["sql-lora"]
        r = await c.post("/model/download", json={"model_id": "x/y", "local_dir": "../../etc"})
        assert r.status == 400
        r = await c.post("/model/download", json={})
        assert r.status == 400
        r = await c.post("/model/download", data=b"not json")
        assert r.status == 400
        await c.close()

    _run(go())


def test_hub_download_is_delegated(tmp_path, monkeypatch):
    calls = []

    def fake_fetch(model_id, target, token, revision, allow_patterns):
        calls.append((model_id, str(target), token, revision))
        target.mkdir(parents=True)
        (target / "adapter_model.safetensors").write_bytes(b"\0")

    monkeypatch.setattr(sidecar, "_fetch_hub", fake_fetch)

    async def go():
        c = await _client(tmp_path)
        r = await c.post("/model/download", json={"model_id": "org/adapter", "token": "t", "revision": "main"})
        assert r.status == 200
        assert (await r.json())["local_dir"] == str((tmp_path / "org--adapter").resolve())
        await c.close()

    _run(go())
    assert calls == [("org/adapter", str((tmp_path / "org--adapter").resolve()), "t", "main")]
