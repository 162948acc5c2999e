"""Minimal asyncio Kubernetes REST client (list / watch / get / patch) for service discovery.

No kubernetes Python package is needed: in-cluster credentials are read from the service
account mount, and ``VPSA_K8S_API`` (+ optional ``VPSA_K8S_TOKEN``) points the client at any
API endpoint (``kubectl proxy``, or the fake API server used by the tests)."""

from __future__ import annotations

import json
import os
import ssl
from typing import AsyncIterator

import aiohttp

_SA = "/var/run/secrets/kubernetes.io/serviceaccount"


class K8sClient:
    def __init__(self, api: str | None = None, token: str | None = None, ca_file: str | None = None):
        self.api = api or os.environ.get("VPSA_K8S_API")
        self.token = token or os.environ.get("VPSA_K8S_TOKEN")
        self.ssl: ssl.SSLContext | bool | None = None
        if not self.api:
            host = os.environ.get("KUBERNETES_SERVICE_HOST")
            port = os.environ.get("KUBERNETES_SERVICE_PORT", "443")
            if not host:
                raise RuntimeError("not running in a cluster and VPSA_K8S_API is not set")
            if ":" in host and not host.startswith("["):
                host = f"[{host}]"
            self.api = f"https://{host}:{port}"
            if self.token is None and os.path.exists(f"{_SA}/token"):
                with open(f"{_SA}/token") as f:
                    self.token = f.read().strip()
            ca_file = ca_file or f"{_SA}/ca.crt"
        if self.api.startswith("https"):
            if ca_file and os.path.exists(ca_file):
                self.ssl = ssl.create_default_context(cafile=ca_file)
            else:
                self.ssl = False
        self._session: aiohttp.ClientSession | None = None

    def _headers(self, extra: dict | None = None) -> dict:
        h = {"Accept": "application/json"}
        if self.token:
            h["Authorization"] = f"Bearer {self.token}"
        if extra:
            h.update(extra)
        return h

    async def session(self) -> aiohttp.ClientSession:
        if self._session is None or self._session.closed:
            self._session = aiohttp.ClientSession(timeout=aiohttp.ClientTimeout(total=None))
        return self._session

    async def close(self) -> None:
        if self._session is not None and not self._session.closed:
            await self._session.close()

    async def get(self, path: str, params: dict | None = None) -> dict:
        s = await self.session()
        async with s.get(self.api + path, params=params, headers=self._headers(), ssl=self.ssl,
                         timeout=aiohttp.ClientTimeout(total=30)) as r:
            r.raise_for_status()
            return await r.json()

    async def patch(self, path: str, body: dict, strategic: bool = True) -> dict:
        s = await self.session()
        ct = "application/strategic-merge-patch+json" if strategic else "application/merge-patch+json"
        async with s.patch(self.api + path, data=json.dumps(body),
                           headers=self._headers({"Content-Type": ct}), ssl=self.ssl,
                           timeout=aiohttp.ClientTimeout(total=30)) as r:
            r.raise_for_status()
            return await r.json()

    async def watch(self, path: str, label_selector: str = "", timeout_seconds: int = 0,
                    resource_version: str | None = None) -> AsyncIterator[dict]:
        """Yields {"type": ADDED|MODIFIED|DELETED, "object": {...}} events."""
        params = {"watch": "true"}
        if label_selector:
            params["labelSelector"] = label_selector
        if timeout_seconds:
            params["timeoutSeconds"] = str(timeout_seconds)
        if resource_version:
            params["resourceVersion"] = resource_version
        s = await self.session()
        async with s.get(self.api + path, params=params, headers=self._headers(), ssl=self.ssl,
                         timeout=aiohttp.ClientTimeout(total=None, sock_read=None)) as r:
            r.raise_for_status()
            buf = b""
            async for chunk in r.content.iter_any():
                buf += chunk
                while b"\n" in buf:
                    line, buf = buf.split(b"\n", 1)
                    if line.strip():
                        yield json.loads(line)
