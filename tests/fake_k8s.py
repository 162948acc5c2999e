"""In-process stand-in for the Kubernetes API server, as far as the native operator uses it:
typed object store with resourceVersions, merge-patch, the status subresource, optimistic
concurrency on PUT (409 on a stale resourceVersion), list + watch streams (one JSON event per
line, chunked) for custom resources and pods, and Leases; optionally over TLS, with a bearer
token required on every call, and with the TokenReview / SubjectAccessReview endpoints a secured
metrics port asks (tokens and their users in ``users``, users allowed ``get /metrics`` in
``metrics_readers``).  Plays the role envtest plays for the
reference's controller tests (/root/reference/operator/internal/controller/suite_test.go:52-82)."""

from __future__ import annotations

import asyncio
import copy
import json
import threading

from aiohttp import web


def _merge(dst: dict, patch: dict) -> dict:
    for k, v in patch.items():
        if v is None:
            dst.pop(k, None)
        elif isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = copy.deepcopy(v)
    return dst


class FakeKube:
    def __init__(self, tls: tuple[str, str] | None = None, token: str | None = None):
        self.tls = tls  # (certificate chain file, key file): serve https
        self.token = token  # every call must carry it as a bearer token
        self.users: dict[str, str] = {}  # bearer token -> user name (TokenReview)
        self.metrics_readers: set[str] = set()  # users allowed `get /metrics` (SubjectAccessReview)
        self.objects: dict[str, dict] = {}  # full path -> object
        self.rv = 100
        self.watchers: dict[str, list[asyncio.Queue]] = {}  # collection path -> queues
        self.requests: list[tuple[str, str]] = []  # (method, path) log
        self.loop: asyncio.AbstractEventLoop | None = None
        self.port = 0
        self._lock = threading.Lock()

    # ------------------------------------------------------------------ store
    def _bump(self, obj: dict) -> dict:
        self.rv += 1
        if isinstance(obj.setdefault("metadata", {}), dict):  # hostile-object tests store other shapes
            obj["metadata"]["resourceVersion"] = str(self.rv)
        return obj

    @staticmethod
    def coll_of(path: str) -> str:
        return path.rsplit("/", 1)[0]

    def _emit(self, coll: str, typ: str, obj: dict) -> None:
        for q in self.watchers.get(coll, []):
            q.put_nowait({"type": typ, "object": copy.deepcopy(obj)})

    def put(self, path: str, obj: dict, typ: str | None = None) -> dict:
        """Test-side write (thread-safe): create or replace an object and notify watchers."""
        def do():
            existed = path in self.objects
            self.objects[path] = self._bump(copy.deepcopy(obj))
            self._emit(self.coll_of(path), typ or ("MODIFIED" if existed else "ADDED"), self.objects[path])
        self._call(do)
        return self.objects[path]

    def mutate(self, path: str, fn) -> None:
        def do():
            fn(self.objects[path])
            self._bump(self.objects[path])
            self._emit(self.coll_of(path), "MODIFIED", self.objects[path])
        self._call(do)

    def delete(self, path: str) -> None:
        def do():
            obj = self.objects.pop(path, None)
            if obj is not None:
                self._emit(self.coll_of(path), "DELETED", obj)
        self._call(do)

    def _call(self, fn) -> None:
        if self.loop is None or threading.current_thread() is self._thread:
            fn()
            return
        done = threading.Event()

        def run():
            try:
                fn()
            finally:
                done.set()
        self.loop.call_soon_threadsafe(run)
        done.wait(5)

    # ------------------------------------------------------------------ handlers
    def _match_selector(self, obj: dict, selector: str) -> bool:
        meta = obj.get("metadata", {})
        labels = (meta.get("labels", {}) if isinstance(meta, dict) else {}) or {}
        if not isinstance(labels, dict):
            return False
        for term in filter(None, selector.split(",")):
            if "=" in term:
                k, v = term.split("=", 1)
                if labels.get(k) != v:
                    return False
            elif term not in labels:
                return False
        return True

    async def handle(self, request: web.Request) -> web.StreamResponse:
        path, method = request.path, request.method
        self.requests.append((method, path))
        if self.token is not None and request.headers.get("Authorization") != f"Bearer {self.token}":
            return web.json_response({"kind": "Status", "reason": "Unauthorized", "code": 401}, status=401)
        if method == "POST" and path == "/apis/authentication.k8s.io/v1/tokenreviews":
            body = await request.json()
            user = self.users.get(body.get("spec", {}).get("token", ""))
            body["status"] = ({"authenticated": True, "user": {"username": user, "groups": ["system:authenticated"]}}
                              if user else {"authenticated": False})
            return web.json_response(body, status=201)
        if method == "POST" and path == "/apis/authorization.k8s.io/v1/subjectaccessreviews":
            body = await request.json()
            spec = body.get("spec", {})
            nra = spec.get("nonResourceAttributes", {})
            ok = spec.get("user") in self.metrics_readers and nra == {"path": "/metrics", "verb": "get"}
            body["status"] = {"allowed": bool(ok)}
            return web.json_response(body, status=201)
        if method == "GET" and request.query.get("watch") == "true":
            return await self._watch(request)
        if method == "GET":
            if path in self.objects:
                return web.json_response(self.objects[path])
            # a collection?
            items = [o for p, o in sorted(self.objects.items()) if self.coll_of(p) == path]
            if items or path.rsplit("/", 1)[-1] in ("pods", "leases", "vllmruntimes", "vllmrouters", "cacheservers",
                                                     "loraadapters", "deployments", "services"):
                sel = request.query.get("labelSelector", "")
                items = [o for o in items if self._match_selector(o, sel)]
                return web.json_response({"kind": "List", "metadata": {"resourceVersion": str(self.rv)}, "items": items})
            return web.json_response({"kind": "Status", "reason": "NotFound", "code": 404}, status=404)
        body = await request.json() if request.can_read_body else {}
        if method == "POST":
            name = body["metadata"]["name"]
            full = path + "/" + name
            if full in self.objects:
                return web.json_response({"reason": "AlreadyExists", "code": 409}, status=409)
            self.objects[full] = self._bump(body)
            self._emit(path, "ADDED", body)
            return web.json_response(body, status=201)
        if method == "PUT":
            cur = self.objects.get(path)
            if cur is None:
                return web.json_response({"reason": "NotFound", "code": 404}, status=404)
            sent = body.get("metadata", {}).get("resourceVersion")
            if sent and sent != cur["metadata"]["resourceVersion"]:
                return web.json_response({"reason": "Conflict", "code": 409}, status=409)
            self.objects[path] = self._bump(body)
            self._emit(self.coll_of(path), "MODIFIED", body)
            return web.json_response(body)
        if method == "PATCH":
            target = path[:-len("/status")] if path.endswith("/status") else path
            cur = self.objects.get(target)
            if cur is None:
                return web.json_response({"reason": "NotFound", "code": 404}, status=404)
            _merge(cur, body)
            self._bump(cur)
            # an object whose deletion was requested disappears when its last finalizer goes
            if cur["metadata"].get("deletionTimestamp") and not cur["metadata"].get("finalizers"):
                del self.objects[target]
                self._emit(self.coll_of(target), "DELETED", cur)
            else:
                self._emit(self.coll_of(target), "MODIFIED", cur)
            return web.json_response(cur)
        if method == "DELETE":
            obj = self.objects.pop(path, None)
            if obj is not None:
                self._emit(self.coll_of(path), "DELETED", obj)
            return web.json_response({})
        return web.json_response({}, status=405)

    async def _watch(self, request: web.Request) -> web.StreamResponse:
        coll = request.path
        sel = request.query.get("labelSelector", "")
        q: asyncio.Queue = asyncio.Queue()
        if request.query.get("resourceVersion", "") in ("", "0"):
            # a watch that names no version starts with the current state, as ADDED events
            for p, o in sorted(self.objects.items()):
                if self.coll_of(p) == coll:
                    q.put_nowait({"type": "ADDED", "object": copy.deepcopy(o)})
        self.watchers.setdefault(coll, []).append(q)
        resp = web.StreamResponse(headers={"Content-Type": "application/json"})
        resp.enable_chunked_encoding()
        await resp.prepare(request)
        try:
            while True:
                try:
                    ev = await asyncio.wait_for(q.get(), timeout=2.0)
                except asyncio.TimeoutError:
                    ev = {"type": "BOOKMARK", "object": {"metadata": {"resourceVersion": str(self.rv)}}}
                if ev["type"] == "STOP":
                    break
                if ev["type"] != "BOOKMARK" and sel and not self._match_selector(ev["object"], sel):
                    continue
                await resp.write(json.dumps(ev).encode() + b"\n")
        except (ConnectionResetError, asyncio.CancelledError, RuntimeError):
            pass
        finally:
            self.watchers[coll].remove(q)
        return resp

    # ------------------------------------------------------------------ lifecycle
    def start(self) -> "FakeKube":
        ready = threading.Event()

        def run():
            self.loop = asyncio.new_event_loop()
            asyncio.set_event_loop(self.loop)

            async def boot():
                app = web.Application()
                app.router.add_route("*", "/{tail:.*}", self.handle)
                self.runner = web.AppRunner(app, access_log=None)
                await self.runner.setup()
                ctx = None
                if self.tls:
                    import ssl

                    ctx = ssl.SSLContext(ssl.PROTOCOL_TLS_SERVER)
                    ctx.load_cert_chain(*self.tls)
                site = web.TCPSite(self.runner, "127.0.0.1", 0, ssl_context=ctx)
                await site.start()
                self.port = self.runner.addresses[0][1]
                ready.set()

            self.loop.run_until_complete(boot())
            self.loop.run_forever()

        self._thread = threading.Thread(target=run, daemon=True)
        self._thread.start()
        ready.wait(10)
        return self

    @property
    def url(self) -> str:
        return f"{'https' if self.tls else 'http'}://127.0.0.1:{self.port}"

    def stop(self) -> None:
        if self.loop is None:
            return

        async def down():
            for qs in self.watchers.values():
                for q in list(qs):
                    q.put_nowait({"type": "STOP"})
            await asyncio.sleep(0.05)
            await self.runner.cleanup()
            self.loop.stop()

        asyncio.run_coroutine_threadsafe(down(), self.loop)
        self._thread.join(5)
