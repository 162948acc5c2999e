"""Service discovery: the router's view of the engine fleet.

Three providers behind one interface (static list, Kubernetes pods by IP, Kubernetes
services by name), all asyncio tasks on the router's loop.  Behavioural reference:
/root/reference/src/vllm_router/service_discovery.py:42-1272 (EndpointInfo fields :80-175,
static health probing :234-319, pod readiness / model query / sleep label :395-709,
service-name variant :762-1081).  Differences by design: no threads or blocking HTTP on
the serving path; prefill/decode pools hold *all* matching endpoints so an N-prefill x
M-decode deployment is load balanced (the reference keeps one client of each).
"""

from __future__ import annotations

import abc
import asyncio
import hashlib
import time
import uuid
from dataclasses import dataclass, field

import aiohttp

from .k8s_client import K8sClient
from .log import init_logger
from .utils import ModelType

logger = init_logger(__name__)


@dataclass
class ModelInfo:
    id: str
    object: str = "model"
    created: int = 0
    owned_by: str = "vllm"
    root: str | None = None
    parent: str | None = None
    is_adapter: bool = False

    @classmethod
    def from_dict(cls, d: dict) -> "ModelInfo":
        return cls(id=d.get("id"), object=d.get("object", "model"),
                   created=d.get("created", int(time.time())), owned_by=d.get("owned_by", "vllm"),
                   root=d.get("root"), parent=d.get("parent"), is_adapter=d.get("parent") is not None)

    def to_dict(self) -> dict:
        return {"id": self.id, "object": self.object, "created": self.created,
                "owned_by": self.owned_by, "root": self.root, "parent": self.parent,
                "is_adapter": self.is_adapter}


@dataclass
class EndpointInfo:
    url: str
    model_names: list[str]
    Id: str
    added_timestamp: float
    model_label: str
    sleep: bool
    pod_name: str | None = None
    service_name: str | None = None
    namespace: str | None = None
    model_info: dict[str, ModelInfo] = field(default_factory=dict)

    def get_base_models(self) -> list[str]:
        return [m for m, i in self.model_info.items() if not i.parent]

    def get_adapters(self) -> list[str]:
        return [m for m, i in self.model_info.items() if i.parent]

    def get_adapters_for_model(self, base: str) -> list[str]:
        return [m for m, i in self.model_info.items() if i.parent == base]

    def has_model(self, model_id: str) -> bool:
        return model_id in self.model_names

    def get_model_info(self, model_id: str) -> ModelInfo | None:
        return self.model_info.get(model_id)


class ServiceDiscovery(abc.ABC):
    aliases: dict[str, str] | None = None
    prefill_model_labels: list[str] | None = None
    decode_model_labels: list[str] | None = None

    @abc.abstractmethod
    def get_endpoint_info(self) -> list[EndpointInfo]:
        ...

    def get_health(self) -> bool:
        return True

    async def start(self) -> None:
        return None

    async def close(self) -> None:
        return None

    # ---- P/D pools -----------------------------------------------------------------
    def prefill_endpoints(self) -> list[EndpointInfo]:
        labels = self.prefill_model_labels or []
        return [e for e in self.get_endpoint_info() if e.model_label in labels and not e.sleep]

    def decode_endpoints(self) -> list[EndpointInfo]:
        labels = self.decode_model_labels or []
        return [e for e in self.get_endpoint_info() if e.model_label in labels and not e.sleep]


# ------------------------------------------------------------------ health probe
async def is_model_healthy(session: aiohttp.ClientSession, url: str, model: str, model_type: str,
                           timeout: float = 10.0, api_key: str | None = None) -> bool:
    """Send a tiny real request of the model's type; healthy iff it answers 200."""
    target = url + ModelType.get_url(model_type)
    headers = {"Authorization": f"Bearer {api_key}"} if api_key else {}
    try:
        if model_type == "transcription":
            payload = ModelType.get_test_payload(model_type)
            form = aiohttp.FormData()
            name, data, ctype = payload["file"]
            form.add_field("file", data, filename=name, content_type=ctype)
            form.add_field("model", model)
            async with session.post(target, data=form, headers=headers,
                                    timeout=aiohttp.ClientTimeout(total=timeout)) as r:
                return r.status == 200
        body = {"model": model, **ModelType.get_test_payload(model_type)}
        async with session.post(target, json=body, headers=headers,
                                timeout=aiohttp.ClientTimeout(total=timeout)) as r:
            if r.status != 200:
                logger.warning("%s model %s at %s answered %d", model_type, model, url, r.status)
            return r.status == 200
    except Exception as e:
        logger.error("%s model %s at %s not healthy: %s", model_type, model, url, e)
        return False


def endpoint_hash(url: str, model: str) -> str:
    return hashlib.md5(f"{url}{model}".encode()).hexdigest()


# ------------------------------------------------------------------ static
class StaticServiceDiscovery(ServiceDiscovery):
    def __init__(self, urls: list[str], models: list[str], aliases: dict[str, str] | None = None,
                 model_labels: list[str] | None = None, model_types: list[str] | None = None,
                 static_backend_health_checks: bool = False,
                 prefill_model_labels: list[str] | None = None,
                 decode_model_labels: list[str] | None = None,
                 health_check_interval: float = 60.0, health_check_timeout: float = 10.0,
                 api_key: str | None = None):
        assert len(urls) == len(models), "URLs and models should have the same length"
        self.urls = urls
        self.models = models
        self.aliases = aliases
        self.model_labels = model_labels
        self.model_types = model_types
        self.engine_ids = [str(uuid.uuid4()) for _ in urls]
        self.added_timestamp = int(time.time())
        self.unhealthy: set[str] = set()
        self.health_checks = static_backend_health_checks
        self.interval = health_check_interval
        self.timeout = health_check_timeout
        self.prefill_model_labels = prefill_model_labels
        self.decode_model_labels = decode_model_labels
        self.api_key = api_key
        self.sleeping: set[str] = set()  # engine ids put to sleep through the router
        self._task: asyncio.Task | None = None
        self._cache: list[EndpointInfo] | None = None

    async def start(self) -> None:
        if self.health_checks and self._task is None:
            self._task = asyncio.create_task(self._health_loop())

    async def close(self) -> None:
        if self._task:
            self._task.cancel()
            self._task = None

    def get_health(self) -> bool:
        return self._task is None or not self._task.done()

    async def check_once(self, session: aiohttp.ClientSession | None = None) -> set[str]:
        own = session is None
        session = session or aiohttp.ClientSession()
        try:
            results = await asyncio.gather(*[
                is_model_healthy(session, u, m, t, self.timeout, self.api_key)
                for u, m, t in zip(self.urls, self.models, self.model_types or [])])
        finally:
            if own:
                await session.close()
        bad = {endpoint_hash(u, m) for (u, m), ok in zip(zip(self.urls, self.models), results) if not ok}
        self.unhealthy = bad
        self._cache = None
        return bad

    async def _health_loop(self) -> None:
        async with aiohttp.ClientSession() as session:
            while True:
                try:
                    await self.check_once(session)
                except asyncio.CancelledError:
                    raise
                except Exception as e:
                    logger.error("health check loop error: %s", e)
                await asyncio.sleep(self.interval)

    def set_sleep(self, engine_id: str, sleeping: bool) -> None:
        (self.sleeping.add if sleeping else self.sleeping.discard)(engine_id)
        self._cache = None

    def get_endpoint_info(self) -> list[EndpointInfo]:
        if self._cache is not None:
            return self._cache
        out = []
        for i, (url, model) in enumerate(zip(self.urls, self.models)):
            if endpoint_hash(url, model) in self.unhealthy:
                continue
            label = self.model_labels[i] if self.model_labels and i < len(self.model_labels) else "default"
            out.append(EndpointInfo(
                url=url, model_names=[model], Id=self.engine_ids[i],
                added_timestamp=self.added_timestamp, model_label=label,
                sleep=self.engine_ids[i] in self.sleeping,
                model_info={model: ModelInfo(id=model, created=self.added_timestamp)}))
        self._cache = out
        return out


# ------------------------------------------------------------------ kubernetes (shared)
class _K8sBase(ServiceDiscovery):
    def __init__(self, namespace: str, port: int | str, label_selector: str = "",
                 prefill_model_labels=None, decode_model_labels=None,
                 watcher_timeout_seconds: int = 0, health_check_timeout: float = 10.0,
                 client: K8sClient | None = None, api_key: str | None = None):
        self.namespace = namespace
        self.port = port
        self.label_selector = label_selector or ""
        self.prefill_model_labels = prefill_model_labels
        self.decode_model_labels = decode_model_labels
        self.watch_timeout = watcher_timeout_seconds
        self.timeout = health_check_timeout
        self.client = client or K8sClient()
        self.api_key = api_key
        self.available: dict[str, EndpointInfo] = {}
        self._task: asyncio.Task | None = None
        self._http: aiohttp.ClientSession | None = None
        self._snapshot: list[EndpointInfo] | None = None

    async def start(self) -> None:
        self._http = aiohttp.ClientSession()
        self._task = asyncio.create_task(self._watch_forever())

    async def close(self) -> None:
        if self._task:
            self._task.cancel()
        if self._http:
            await self._http.close()
        await self.client.close()

    def get_health(self) -> bool:
        return self._task is not None and not self._task.done()

    def get_endpoint_info(self) -> list[EndpointInfo]:
        if self._snapshot is None:
            self._snapshot = list(self.available.values())
        return self._snapshot

    def _auth(self) -> dict:
        return {"Authorization": f"Bearer {self.api_key}"} if self.api_key else {}

    async def _get_models(self, url: str) -> dict[str, ModelInfo]:
        try:
            async with self._http.get(f"{url}/v1/models", headers=self._auth(),
                                      timeout=aiohttp.ClientTimeout(total=self.timeout)) as r:
                r.raise_for_status()
                data = await r.json()
            return {m["id"]: ModelInfo.from_dict(m) for m in data.get("data", [])}
        except Exception as e:
            logger.error("Failed to get model info from %s: %s", url, e)
            return {}

    async def _is_sleeping(self, url: str) -> bool:
        try:
            async with self._http.get(f"{url}/is_sleeping", headers=self._auth(),
                                      timeout=aiohttp.ClientTimeout(total=self.timeout)) as r:
                r.raise_for_status()
                return bool((await r.json()).get("is_sleeping", False))
        except Exception as e:
            logger.warning("Failed to get the sleep status for %s: %s", url, e)
            return False

    def _delete(self, name: str) -> None:
        if self.available.pop(name, None) is not None:
            logger.info("Engine %s removed from discovery", name)
            self._snapshot = None

    async def _watch_forever(self) -> None:
        while True:
            try:
                async for ev in self.client.watch(self._watch_path(), self.label_selector,
                                                  self.watch_timeout):
                    await self._on_event(ev)
            except asyncio.CancelledError:
                raise
            except Exception as e:
                logger.error("K8s watcher error: %s", e)
            await asyncio.sleep(0.5)

    @abc.abstractmethod
    def _watch_path(self) -> str: ...

    @abc.abstractmethod
    async def _on_event(self, ev: dict) -> None: ...


class K8sPodIPServiceDiscovery(_K8sBase):
    """Watches pods; an engine is routable when every container is ready and the pod is not
    terminating.  Engine id = uuid5(DNS namespace, pod name)."""

    def _watch_path(self) -> str:
        return f"/api/v1/namespaces/{self.namespace}/pods"

    @staticmethod
    def _pod_ready(pod: dict) -> bool:
        st = (pod.get("status") or {}).get("containerStatuses")
        if not st:
            return False
        return all(c.get("ready") for c in st)

    @staticmethod
    def _terminating(pod: dict) -> bool:
        return (pod.get("metadata") or {}).get("deletionTimestamp") is not None

    @staticmethod
    def _sleep_mode_enabled(pod: dict) -> bool:
        for c in (pod.get("spec") or {}).get("containers", []):
            for field_ in ("command", "args"):
                if "--enable-sleep-mode" in (c.get(field_) or []):
                    return True
        return False

    async def _on_event(self, ev: dict) -> None:
        pod = ev.get("object") or {}
        meta = pod.get("metadata") or {}
        name = meta.get("name")
        if not name:
            return
        if ev.get("type") == "DELETED":
            self._delete(name)
            return
        ip = (pod.get("status") or {}).get("podIP")
        ready = self._pod_ready(pod) and not self._terminating(pod) and bool(ip)
        if not ready:
            self._delete(name)
            return
        host = f"[{ip}]" if ":" in ip else ip
        url = f"http://{host}:{self.port}"
        models = await self._get_models(url)
        if not models:
            self._delete(name)
            return
        label = (meta.get("labels") or {}).get("model", "default")
        sleeping = await self._is_sleeping(url) if self._sleep_mode_enabled(pod) else False
        prev = self.available.get(name)
        self.available[name] = EndpointInfo(
            url=url, model_names=list(models), Id=str(uuid.uuid5(uuid.NAMESPACE_DNS, name)),
            added_timestamp=prev.added_timestamp if prev else int(time.time()), model_label=label,
            sleep=sleeping, pod_name=name, namespace=self.namespace, model_info=models)
        self._snapshot = None
        if prev is None:
            logger.info("Discovered new serving engine %s at %s, running models: %s", name, url,
                        list(models))

    async def set_sleep_label(self, pod_name: str, sleeping: bool) -> None:
        """Mirror the sleep state on the pod as label ``sleeping=true`` (removed on wake)."""
        body = {"metadata": {"labels": {"sleeping": "true" if sleeping else None}}}
        try:
            await self.client.patch(f"/api/v1/namespaces/{self.namespace}/pods/{pod_name}", body)
        except Exception as e:
            logger.error("Failed to patch sleeping label on %s: %s", pod_name, e)
        ep = self.available.get(pod_name)
        if ep is not None:
            ep.sleep = sleeping


class K8sServiceNameServiceDiscovery(_K8sBase):
    """Watches Services; an engine is routable when its Endpoints object has addresses.
    URL = http://<service>:<port>."""

    def _watch_path(self) -> str:
        return f"/api/v1/namespaces/{self.namespace}/services"

    async def _service_ready(self, name: str) -> bool:
        try:
            ep = await self.client.get(f"/api/v1/namespaces/{self.namespace}/endpoints/{name}")
        except Exception as e:
            logger.error("Error checking service readiness: %s", e)
            return False
        return any(s.get("addresses") for s in ep.get("subsets") or [])

    async def _on_event(self, ev: dict) -> None:
        svc = ev.get("object") or {}
        meta = svc.get("metadata") or {}
        name = meta.get("name")
        if not name:
            return
        if ev.get("type") == "DELETED" or not await self._service_ready(name):
            self._delete(name)
            return
        url = f"http://{name}:{self.port}"
        models = await self._get_models(url)
        if not models:
            self._delete(name)
            return
        label = (meta.get("labels") or {}).get("model") or \
            ((svc.get("spec") or {}).get("selector") or {}).get("model", "default")
        prev = self.available.get(name)
        self.available[name] = EndpointInfo(
            url=url, model_names=list(models), Id=str(uuid.uuid5(uuid.NAMESPACE_DNS, name)),
            added_timestamp=prev.added_timestamp if prev else int(time.time()), model_label=label,
            sleep=await self._is_sleeping(url) if (meta.get("labels") or {}).get("sleeping") else False,
            service_name=name, namespace=self.namespace, model_info=models)
        self._snapshot = None
        if prev is None:
            logger.info("Discovered new serving engine service %s at %s", name, url)

    async def set_sleep_label(self, service_name: str, sleeping: bool) -> None:
        body = {"metadata": {"labels": {"sleeping": "true" if sleeping else None}}}
        try:
            await self.client.patch(
                f"/api/v1/namespaces/{self.namespace}/services/{service_name}", body)
        except Exception as e:
            logger.error("Failed to patch sleeping label on %s: %s", service_name, e)
        ep = self.available.get(service_name)
        if ep is not None:
            ep.sleep = sleeping


def create_service_discovery(kind: str, **kw) -> ServiceDiscovery:
    if kind == "static":
        return StaticServiceDiscovery(**kw)
    if kind in ("k8s", "k8s-pod-ip", "pod-ip"):
        return K8sPodIPServiceDiscovery(**kw)
    if kind in ("k8s-service-name", "service-name"):
        return K8sServiceNameServiceDiscovery(**kw)
    raise ValueError(f"Invalid service discovery type: {kind}")
