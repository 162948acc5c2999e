"""KV controller: the cluster-wide index of which engine instance holds which KV chunks.

Role in the reference: the LMCache controller the router embeds on port 9000 and queries
with ``LookupMsg(tokens)`` / ``QueryInstMsg(ip)``
(/root/reference/src/vllm_router/routers/routing_logic.py:238-262,288-344), and the HTTP
``/lookup`` + ``/query`` service the Go KV-aware picker calls
(/root/reference/src/gateway_inference_extension/kv_aware_picker.go:28-133).

Here engines report the chain hashes of KV pages they admit / evict (any tier); a lookup
walks a prompt's hash chain and returns, per instance, how many leading tokens are
resident.  The same object is used in-process by the router and served over HTTP (JSON)
for engines and for the native picker service.
"""

from __future__ import annotations

import asyncio
import time

from aiohttp import web

PAGE = 64


class KVController:
    def __init__(self, stale_after: float = 120.0):
        self.instances: dict[str, dict] = {}  # instance_id -> {url, ip, model, last_seen}
        self.holders: dict[int, set[str]] = {}  # page hash -> instance ids
        self.by_instance: dict[str, set[int]] = {}
        self.stale_after = stale_after

    # ------------------------------------------------------------------ worker side
    def register(self, instance_id: str, url: str = "", ip: str = "", model: str = "") -> None:
        self.instances[instance_id] = {"url": url, "ip": ip, "model": model, "last_seen": time.time()}
        self.by_instance.setdefault(instance_id, set())

    def deregister(self, instance_id: str) -> None:
        for h in self.by_instance.pop(instance_id, set()):
            s = self.holders.get(h)
            if s:
                s.discard(instance_id)
                if not s:
                    del self.holders[h]
        self.instances.pop(instance_id, None)

    def admit(self, instance_id: str, hashes: list[int]) -> None:
        mine = self.by_instance.setdefault(instance_id, set())
        for h in hashes:
            self.holders.setdefault(h, set()).add(instance_id)
            mine.add(h)
        if instance_id in self.instances:
            self.instances[instance_id]["last_seen"] = time.time()

    def evict(self, instance_id: str, hashes: list[int]) -> None:
        mine = self.by_instance.get(instance_id, set())
        for h in hashes:
            mine.discard(h)
            s = self.holders.get(h)
            if s:
                s.discard(instance_id)
                if not s:
                    del self.holders[h]

    # ------------------------------------------------------------------ query side
    def lookup(self, hashes: list[int], model: str | None = None) -> dict[str, int]:
        """-> {instance_id: matched leading tokens}."""
        alive: set[str] | None = None
        out: dict[str, int] = {}
        for i, h in enumerate(hashes):
            s = self.holders.get(h)
            if not s:
                break
            alive = set(s) if alive is None else alive & s
            if model:
                alive = {x for x in alive if self.instances.get(x, {}).get("model") in ("", model)}
            if not alive:
                break
            for inst in alive:
                out[inst] = (i + 1) * PAGE
        return out

    def query_instance(self, ip_or_url: str) -> str | None:
        for iid, info in self.instances.items():
            if ip_or_url and (info["ip"] == ip_or_url or info["url"] == ip_or_url
                              or ip_or_url in info["url"]):
                return iid
        return None

    def url_of(self, instance_id: str) -> str | None:
        info = self.instances.get(instance_id)
        return info["url"] if info else None

    def reap_stale(self) -> list[str]:
        now = time.time()
        dead = [i for i, v in self.instances.items() if now - v["last_seen"] > self.stale_after]
        for i in dead:
            self.deregister(i)
        return dead


def make_app(ctrl: KVController, tokenize=None) -> web.Application:
    """HTTP face of the controller.  ``tokenize(model, prompt) -> awaitable[list[int]]`` lets
    text lookups (picker service) be resolved to hash chains."""
    from ..engine.block_manager import chain_hashes

    app = web.Application(client_max_size=64 * 1024 * 1024)

    async def report(request: web.Request):
        d = await request.json()
        iid = d["instance_id"]
        op = d.get("op", "admit")
        if op == "register":
            ctrl.register(iid, d.get("url", ""), d.get("ip", ""), d.get("model", ""))
        elif op == "deregister":
            ctrl.deregister(iid)
        else:
            if iid not in ctrl.instances:
                ctrl.register(iid, d.get("url", ""), d.get("ip", ""), d.get("model", ""))
            if d.get("admit"):
                ctrl.admit(iid, d["admit"])
            if d.get("evict"):
                ctrl.evict(iid, d["evict"])
        return web.json_response({"ok": True})

    async def lookup(request: web.Request):
        d = await request.json()
        if "hashes" in d:
            hashes = d["hashes"]
        else:
            tokens = d.get("tokens")
            if tokens is None:
                if tokenize is None:
                    return web.json_response({"error": "text lookup needs a tokenizer"}, status=400)
                tokens = await tokenize(d.get("model", ""), d.get("prompt", ""))
            hashes = chain_hashes(list(tokens))
        matches = ctrl.lookup(hashes, d.get("model"))
        best = max(matches.items(), key=lambda kv: kv[1], default=(None, 0))
        return web.json_response({"matches": matches, "instance_id": best[0], "tokens": best[1],
                                  "url": ctrl.url_of(best[0]) if best[0] else None})

    async def query(request: web.Request):
        ip = request.query.get("ip", "")
        return web.json_response({"instance_id": ctrl.query_instance(ip)})

    async def instances(request: web.Request):
        return web.json_response({k: {**v, "pages": len(ctrl.by_instance.get(k, ()))}
                                  for k, v in ctrl.instances.items()})

    app.router.add_post("/report", report)
    app.router.add_post("/lookup", lookup)
    app.router.add_get("/query", query)
    app.router.add_get("/instances", instances)
    app.router.add_get("/health", lambda r: web.json_response({"status": "ok"}))
    return app


class ControllerHandle:
    def __init__(self, runner: web.AppRunner, reaper: asyncio.Task):
        self.runner, self.reaper = runner, reaper

    @property
    def addresses(self):
        return self.runner.addresses

    async def cleanup(self) -> None:
        self.reaper.cancel()
        await self.runner.cleanup()


async def start_controller(ctrl: KVController, host: str, port: int, tokenize=None) -> ControllerHandle:
    runner = web.AppRunner(make_app(ctrl, tokenize), access_log=None)
    await runner.setup()
    await web.TCPSite(runner, host, port, reuse_address=True).start()

    async def reaper():
        while True:
            await asyncio.sleep(30)
            ctrl.reap_stale()

    return ControllerHandle(runner, asyncio.create_task(reaper()))
