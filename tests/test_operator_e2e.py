"""The native operator driven end to end against a fake API server (tests/fake_k8s.py), fake engine
pods and fake download sidecars: the LoraAdapter life cycle, event-driven reconciliation, leader
election over a Lease, and the metrics endpoint.

Mirrors what the reference covers with envtest
(/root/reference/operator/internal/controller/loraadapter_controller_test.go:34-284: reconcile ->
finalizer -> missing pods) and what its controller does beyond that test
(loraadapter_controller.go:334-390 sidecar download, :527-542 "already loaded" is success,
:868-906 finalizer unloads everywhere, :234-275 pod-ready events trigger reconciles;
cmd/main.go:187 leader election, :252-259 probes)."""

import asyncio
import json
import socket
import subprocess
import threading
import time
import urllib.request
from pathlib import Path

import pytest
from aiohttp import web

from fake_k8s import FakeKube
from vllm_production_stack_amd.testing.fake_engine import FakeEngine

OPERATOR = Path(__file__).resolve().parents[1] / "vllm_production_stack_amd" / "_native" / "vpsa-operator"
CRS = "/apis/production-stack.vllm.ai/v1alpha1/namespaces/default"
PODS = "/api/v1/namespaces/default/pods"
LEASE = "/apis/coordination.k8s.io/v1/namespaces/default/leases/4549d26f.vllm.ai"
FINALIZER = "production-stack.vllm.ai/lora-finalizer"


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class StaleListingEngine(FakeEngine):
    """An engine whose /v1/models does not list adapters (a listing that lags behind): the
    operator finds out that the adapter is there only from the 400 of the load call."""

    async def models(self, request):
        return web.json_response({"object": "list", "data": [
            {"id": self.model, "object": "model", "created": 0, "owned_by": "vllm", "root": None, "parent": None}]})


class Backends:
    """Engine pods on 127.0.0.2-4 (one port for all, as in a cluster) and their sidecars."""

    def __init__(self):
        self.loop = asyncio.new_event_loop()
        self.engine_port, self.sidecar_port = _free_port(), _free_port()
        self.engines: dict[str, FakeEngine] = {}
        self.downloads: list[tuple[str, dict]] = []
        threading.Thread(target=self.loop.run_forever, daemon=True).start()

    def run(self, coro):
        return asyncio.run_coroutine_threadsafe(coro, self.loop).result(10)

    def add(self, ip: str, stale=False, lora=None) -> FakeEngine:
        eng = (StaleListingEngine if stale else FakeEngine)(model="llama", lora=lora)
        self.run(eng.start(ip, self.engine_port))
        self.engines[ip] = eng

        async def sidecar():
            app = web.Application()

            async def download(request):
                body = await request.json()
                self.downloads.append((ip, body))
                return web.json_response({"path": "/data/lora-adapters/" + body["local_dir"]})

            app.router.add_post("/model/download", download)
            runner = web.AppRunner(app, access_log=None)
            await runner.setup()
            await web.TCPSite(runner, ip, self.sidecar_port).start()

        self.run(sidecar())
        return eng

    def stop(self):
        for e in self.engines.values():
            self.run(e.stop())
        self.loop.call_soon_threadsafe(self.loop.stop)


def pod(name, ip, ready=True):
    return {"apiVersion": "v1", "kind": "Pod",
            "metadata": {"name": name, "namespace": "default", "labels": {"model": "llama"}},
            "status": {"podIP": ip, "containerStatuses": [{"name": "vllm", "ready": ready}]}}


def lora_cr(replicas=0):
    return {"apiVersion": "production-stack.vllm.ai/v1alpha1", "kind": "LoraAdapter",
            "metadata": {"name": "ad-sample", "namespace": "default"},
            "spec": {"baseModel": "llama", "vllmPort": None,
                     "adapterSource": {"type": "huggingface", "adapterName": "sql-lora",
                                       "repository": "acme/sql-lora"},
                     "loraAdapterDeploymentConfig": {"algorithm": "default", "replicas": replicas}}}


def wait_for(cond, timeout=10.0, what=""):
    t0 = time.time()
    while time.time() - t0 < timeout:
        if cond():
            return time.time() - t0
        time.sleep(0.05)
    raise AssertionError(f"timed out waiting for {what}")


class Operator:
    def __init__(self, kube, *extra, health_port=None):
        self.health_port = health_port or _free_port()
        self.proc = subprocess.Popen([str(OPERATOR), "run", "--api", kube.url, "--namespace", "default",
                                      "--interval", "30", "--health-port", str(self.health_port), *extra],
                                     stderr=subprocess.PIPE, text=True)

    def get(self, path):
        with urllib.request.urlopen(f"http://127.0.0.1:{self.health_port}{path}", timeout=5) as r:
            return r.status, r.read().decode()

    def metric(self, name):
        try:
            text = self.get("/metrics")[1]
        except OSError:  # not listening yet
            return None
        for ln in text.splitlines():
            if ln.startswith(name + " ") or ln.startswith(name + "{"):
                return float(ln.rsplit(" ", 1)[1])
        return None

    def stop(self):
        self.proc.terminate()
        try:
            self.proc.wait(5)
        except subprocess.TimeoutExpired:
            self.proc.kill()
        return self.proc.stderr.read()


@pytest.fixture
def kube():
    k = FakeKube().start()
    yield k
    k.stop()


def test_lora_adapter_life_cycle_is_event_driven(kube):
    be = Backends()
    e2, e3 = be.add("127.0.0.2"), be.add("127.0.0.3", stale=True, lora=["sql-lora"])
    e4 = be.add("127.0.0.4")
    kube.put(f"{PODS}/p2", pod("p2", "127.0.0.2"))
    kube.put(f"{PODS}/p3", pod("p3", "127.0.0.3"))
    kube.put(f"{PODS}/p4", pod("p4", "127.0.0.4", ready=False))
    cr = lora_cr()
    cr["spec"]["vllmPort"] = be.engine_port
    op = Operator(kube, "--sidecar-port", str(be.sidecar_port))
    path = f"{CRS}/loraadapters/ad-sample"
    try:
        wait_for(lambda: op.metric("vpsa_operator_passes_total") is not None, what="first pass")
        # the resync interval is 30 s: everything below happens because of watch events
        kube.put(path, cr)
        dt = wait_for(lambda: "sql-lora" in e2.lora, 5, "adapter loaded on the ready pod")
        assert dt < 3.0
        obj = kube.objects[path]
        assert FINALIZER in obj["metadata"]["finalizers"]
        # the sidecar of every target pod was asked to download the repository first
        assert {ip for ip, _ in be.downloads} >= {"127.0.0.2", "127.0.0.3"}
        assert be.downloads[0][1]["model_id"] == "acme/sql-lora"
        # p3 answered "has already been loaded" (400): that is success, not a failure
        wait_for(lambda: kube.objects[path].get("status", {}).get("phase") == "Ready", 5, "status Ready")
        st = kube.objects[path]["status"]
        assert sorted(a["podName"] for a in st["loadedAdapters"]) == ["p2", "p3"]
        assert "sql-lora" not in e4.lora  # not ready: not a target
        # a pod becoming ready is an event the LoraAdapter reconciler reacts to
        kube.mutate(f"{PODS}/p4", lambda p: p["status"]["containerStatuses"][0].update(ready=True))
        dt = wait_for(lambda: "sql-lora" in e4.lora, 5, "adapter loaded on the pod that became ready")
        assert dt < 3.0
        wait_for(lambda: len(kube.objects[path]["status"]["loadedAdapters"]) == 3, 5, "status lists p4")
        # deletion: unload everywhere, then let go of the finalizer (the object disappears)
        kube.mutate(path, lambda o: o["metadata"].update(deletionTimestamp="2026-01-01T00:00:00Z"))
        wait_for(lambda: path not in kube.objects, 5, "finalizer removed")
        assert "sql-lora" not in e2.lora and "sql-lora" not in e4.lora and "sql-lora" not in e3.lora
        assert op.metric('vpsa_operator_lora_actions_total{op="load",result="already_loaded"}') >= 1
        assert op.metric('vpsa_operator_passes_total{trigger="event"}') >= 3
        status, body = op.get("/readyz")
        assert status == 200 and op.get("/healthz")[0] == 200
    finally:
        err = op.stop()
        be.stop()
    assert "Traceback" not in err


def test_lora_adapter_without_ready_pods_is_pending(kube):
    op = Operator(kube)
    path = f"{CRS}/loraadapters/ad-sample"
    try:
        cr = lora_cr()
        cr["spec"]["adapterSource"] = {"type": "local", "adapterName": "x", "adapterPath": "/data/x"}
        kube.put(path, cr)
        wait_for(lambda: kube.objects[path].get("status", {}).get("phase") == "Pending", 5, "Pending status")
        assert "no ready pods" in kube.objects[path]["status"]["message"]
    finally:
        op.stop()


def test_leader_election_over_a_lease(kube):
    from test_native_control_plane import RUNTIME_CR

    kube.put(f"{CRS}/vllmruntimes/vllmruntime-sample", RUNTIME_CR)
    a = Operator(kube, "--leader-elect", "--leader-id", "op-a", "--lease-duration", "3")
    try:
        wait_for(lambda: LEASE in kube.objects and kube.objects[LEASE]["spec"]["holderIdentity"] == "op-a", 5, "a holds the lease")
        b = Operator(kube, "--leader-elect", "--leader-id", "op-b", "--lease-duration", "3")
        try:
            wait_for(lambda: "/apis/apps/v1/namespaces/default/deployments/vllmruntime-sample" in kube.objects, 5,
                     "the leader reconciled")
            time.sleep(1.5)
            assert a.metric("vpsa_operator_leader") == 1 and b.metric("vpsa_operator_leader") == 0
            assert b.get("/readyz")[0] == 200  # a standby is ready, just not leading
            assert (b.metric("vpsa_operator_passes_total") or 0) == 0  # and does not reconcile
            renew0 = kube.objects[LEASE]["spec"]["renewTime"]
            time.sleep(1.5)
            assert kube.objects[LEASE]["spec"]["renewTime"] > renew0  # the holder keeps renewing
            # the leader dies: the standby takes over once the lease has run out
            a.stop()
            wait_for(lambda: kube.objects[LEASE]["spec"]["holderIdentity"] == "op-b", 8, "b took over")
            assert kube.objects[LEASE]["spec"]["leaseTransitions"] == 1
            # and reconciles: drift introduced now is repaired by b
            key = "/apis/apps/v1/namespaces/default/deployments/vllmruntime-sample"
            kube.mutate(key, lambda d: d["spec"].update(replicas=9))
            kube.mutate(f"{CRS}/vllmruntimes/vllmruntime-sample", lambda o: o["metadata"].setdefault("annotations", {}).update(t="1"))
            wait_for(lambda: kube.objects[key]["spec"]["replicas"] == 2, 5, "b repaired the drift")
            assert b.metric("vpsa_operator_leader") == 1
        finally:
            b.stop()
    finally:
        a.stop()


def test_two_candidates_starting_together_elect_exactly_one(kube):
    ops = [Operator(kube, "--leader-elect", "--leader-id", f"op-{i}", "--lease-duration", "3") for i in range(3)]
    try:
        wait_for(lambda: LEASE in kube.objects, 5, "lease created")
        time.sleep(2.5)
        leaders = [o.metric("vpsa_operator_leader") for o in ops]
        assert sorted(leaders) == [0, 0, 1], leaders
    finally:
        for o in ops:
            o.stop()


def test_metrics_endpoint_is_prometheus_text(kube):
    op = Operator(kube)
    try:
        wait_for(lambda: op.metric("vpsa_operator_passes_total") is not None, what="first pass")
        status, body = op.get("/metrics")
        assert status == 200
        assert "# TYPE vpsa_operator_passes_total counter" in body and "# TYPE vpsa_operator_leader gauge" in body
        for ln in body.splitlines():
            if ln and not ln.startswith("#"):
                name, value = ln.rsplit(" ", 1)
                float(value)
                assert name[0].isalpha()
        with pytest.raises(urllib.error.HTTPError):
            op.get("/nope")
    finally:
        op.stop()
