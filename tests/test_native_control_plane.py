"""Native (C++) control-plane binaries: picker service and operator."""

import asyncio
import json
import subprocess
import threading
import time
import urllib.request

import pytest
from aiohttp import web

from vllm_production_stack_amd import _native
from vllm_production_stack_amd.testing.fake_engine import FakeEngine

PICKER = _native.binary_path("vpsa-picker")
OPERATOR = _native.binary_path("vpsa-operator")
pytestmark = pytest.mark.skipif(not PICKER.exists() or not OPERATOR.exists(),
                                reason="native binaries not built (python -m vllm_production_stack_amd._native.build)")


def post(url, payload):
    req = urllib.request.Request(url, data=json.dumps(payload).encode(),
                                 headers={"Content-Type": "application/json"})
    with urllib.request.urlopen(req, timeout=5) as r:
        return json.loads(r.read())


class PickerProc:
    def __init__(self, *args):
        self.p = subprocess.Popen([str(PICKER), "--port", "0", "--host", "127.0.0.1", *args],
                                  stdout=subprocess.PIPE, text=True)
        line = self.p.stdout.readline()
        self.port = int(line.strip().rsplit(":", 1)[1])
        self.url = f"http://127.0.0.1:{self.port}"

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.p.terminate()
        self.p.wait(timeout=5)


PODS = [{"name": f"default/pod-{c}", "address": f"10.0.0.{i}"} for i, c in enumerate("cab")]


def test_picker_roundrobin_sorted_rotation():
    with PickerProc("--picker", "roundrobin") as pk:
        got = [post(pk.url + "/pick", {"model": "m", "prompt": "x", "pods": PODS})["pod"] for _ in range(6)]
        assert got == ["default/pod-a", "default/pod-b", "default/pod-c"] * 2
        with urllib.request.urlopen(pk.url + "/health") as r:
            assert json.loads(r.read())["status"] == "ok"


def test_picker_prefix_affinity_matches_python_trie():
    from vllm_production_stack_amd.router.native import HashTrie

    with PickerProc("--picker", "prefix") as pk:
        base = "conversation-1 " * 50
        first = post(pk.url + "/pick", {"model": "m", "prompt": base, "pods": PODS})
        for turn in range(5):
            r = post(pk.url + "/pick", {"model": "m", "prompt": base + "more " * 40 * (turn + 1), "pods": PODS})
            assert r["pod"] == first["pod"] and r["matched"] >= 128
        # the service and the Python binding share the chunking: same matched length
        t = HashTrie(128)
        t.insert(base, "e")
        assert t.longest_prefix_match(base + "zzz", {"e"})[0] == \
            post(pk.url + "/pick", {"model": "m", "prompt": base + "zzz", "pods": PODS})["matched"]


def test_picker_session_consistent():
    with PickerProc("--picker", "session") as pk:
        a = {post(pk.url + "/pick", {"session": "alice", "prompt": "x", "pods": PODS})["pod"] for _ in range(5)}
        assert len(a) == 1


def _serve(app_factory):
    loop = asyncio.new_event_loop()
    box = {"ready": threading.Event()}

    def run():
        asyncio.set_event_loop(loop)

        async def start():
            runner = web.AppRunner(app_factory(), access_log=None, shutdown_timeout=0.1)
            await runner.setup()
            site = web.TCPSite(runner, "127.0.0.1", 0)
            await site.start()
            box["port"] = runner.addresses[0][1]
            box["runner"] = runner

        loop.run_until_complete(start())
        box["ready"].set()
        loop.run_forever()

    threading.Thread(target=run, daemon=True).start()
    box["ready"].wait(10)
    return loop, box


def test_picker_kvaware_uses_controller_then_falls_back():
    from vllm_production_stack_amd.controller import KVController, make_app
    from vllm_production_stack_amd.engine.block_manager import chain_hashes

    ctrl = KVController()
    ctrl.register("inst-b", "http://10.0.0.2:8000", "10.0.0.2")
    words = [f"w{i}" for i in range(640)]
    toks = list(range(100, 740))
    ctrl.admit("inst-b", chain_hashes(toks))

    async def tokenize(model, prompt):
        n = len(prompt.split())
        return toks[:n] if prompt.startswith("w0 ") else list(range(9000, 9000 + n))

    loop, box = _serve(lambda: make_app(ctrl, tokenize))
    try:
        with PickerProc("--picker", "kvaware", "--controller", f"http://127.0.0.1:{box['port']}",
                        "--threshold", "100") as pk:
            r = post(pk.url + "/pick", {"model": "m", "prompt": " ".join(words), "pods": PODS})
            assert r["address"] == "10.0.0.2" and r["matched"] == 640 and r["picker"] == "kvaware"
            r = post(pk.url + "/pick", {"model": "m", "prompt": "short unknown prompt " * 200, "pods": PODS})
            assert r["picker"].startswith("roundrobin")
    finally:
        loop.call_soon_threadsafe(loop.stop)


RUNTIME_CR = {
    "apiVersion": "production-stack.vllm.ai/v1alpha1", "kind": "VLLMRuntime",
    "metadata": {"name": "vllmruntime-sample", "namespace": "default", "uid": "u-1"},
    "spec": {
        "model": {"modelURL": "meta-llama/Llama-3.1-8B-Instruct", "enableLoRA": True, "enableTool": False,
                  "maxModelLen": 4096, "dtype": "bfloat16", "maxNumSeqs": 32,
                  "hfTokenSecret": {"name": "hf-token"}, "hfTokenName": "token"},
        "vllmConfig": {"enableChunkedPrefill": False, "enablePrefixCaching": False, "tensorParallelSize": 1,
                       "gpuMemoryUtilization": "0.8", "maxLoras": 4, "extraArgs": ["--disable-log-requests"],
                       "v1": True, "port": 8000, "env": [{"name": "HF_HOME", "value": "/data"}]},
        "lmCacheConfig": {"enabled": True, "cpuOffloadingBufferSize": "15", "diskOffloadingBufferSize": "0",
                          "remoteUrl": "lm://cacheserver-sample.default.svc.cluster.local:80",
                          "remoteSerde": "naive"},
        "storageConfig": {"enabled": True, "storageClassName": "standard", "size": "10Gi",
                          "accessMode": "ReadWriteMany", "mountPath": "/data"},
        "deploymentConfig": {"replicas": 2, "deploymentStrategy": "Recreate",
                             "resources": {"cpu": "10", "memory": "32Gi", "gpu": "1"},
                             "image": {"registry": "docker.io", "name": "lmcache/vllm-openai:latest"},
                             "sidecarConfig": {"enabled": True, "name": "sidecar",
                                               "image": {"registry": "docker.io", "name": "lmcache/lmstack-sidecar:latest"},
                                               "mountPath": "/data"}}}}


def render(kind, cr, *extra):
    out = subprocess.run([str(OPERATOR), "render", kind, *extra], input=json.dumps(cr), capture_output=True,
                         text=True, check=True).stdout
    return {(o["kind"], o["metadata"]["name"]): o for o in json.loads(out)["items"]}


def test_operator_renders_runtime_children():
    objs = render("VLLMRuntime", RUNTIME_CR)
    assert set(k for k, _ in objs) == {"Service", "PersistentVolumeClaim", "Deployment"}
    dep = objs[("Deployment", "vllmruntime-sample")]
    assert dep["spec"]["replicas"] == 2 and dep["spec"]["strategy"]["type"] == "Recreate"
    assert dep["metadata"]["ownerReferences"][0]["kind"] == "VLLMRuntime"
    c = dep["spec"]["template"]["spec"]["containers"][0]
    assert c["command"] == ["vllm", "serve"] and c["image"] == "docker.io/lmcache/vllm-openai:latest"
    a = c["args"]
    assert a[:5] == ["meta-llama/Llama-3.1-8B-Instruct", "--host", "0.0.0.0", "--port", "8000"]
    for flag in ("--enable-lora", "--no-enable-chunked-prefill", "--no-enable-prefix-caching",
                 "--disable-log-requests", "--kv-transfer-config"):
        assert flag in a
    assert a[a.index("--max-model-len") + 1] == "4096" and a[a.index("--max_loras") + 1] == "4"
    assert a[a.index("--gpu_memory_utilization") + 1] == "0.8"
    assert json.loads(a[a.index("--kv-transfer-config") + 1]) == {"kv_connector": "LMCacheConnectorV1", "kv_role": "kv_both"}
    env = {e["name"]: e.get("value") for e in c["env"]}
    assert env["LMCACHE_LOCAL_CPU"] == "True" and env["LMCACHE_MAX_LOCAL_CPU_SIZE"] == "15"
    assert env["LMCACHE_REMOTE_URL"].startswith("lm://") and env["LMCACHE_REMOTE_SERDE"] == "naive"
    assert env["VLLM_USE_V1"] == "1" and "HF_TOKEN" in env
    assert c["resources"]["limits"]["amd.com/gpu"] == "1"  # AMD resource name by default
    assert len(dep["spec"]["template"]["spec"]["containers"]) == 2  # LoRA download sidecar
    svc = objs[("Service", "vllmruntime-sample")]
    assert svc["spec"]["ports"][0]["port"] == 80 and svc["spec"]["ports"][0]["targetPort"] == 8000
    nv = render("VLLMRuntime", RUNTIME_CR, "nvidia.com/gpu")
    assert "nvidia.com/gpu" in nv[("Deployment", "vllmruntime-sample")]["spec"]["template"]["spec"]["containers"][0]["resources"]["limits"]


def test_operator_renders_router_and_cacheserver():
    router = {"metadata": {"name": "vllmrouter-sample", "namespace": "default"},
              "spec": {"serviceDiscovery": "k8s", "k8sLabelSelector": "app=vllmruntime-sample",
                       "routingLogic": "prefixaware", "engineScrapeInterval": 30, "requestStatsWindow": 60,
                       "port": 80, "serviceAccountName": "vllmrouter-sa",
                       "image": {"registry": "docker.io", "name": "lmcache/lmstack-router:latest"}}}
    objs = render("VLLMRouter", router)
    assert {k for k, _ in objs} == {"ServiceAccount", "Role", "RoleBinding", "Service", "Deployment"}
    role = objs[("Role", "pod-viewer-role")]
    assert set(role["rules"][0]["resources"]) >= {"pods", "services", "endpoints"}
    args = objs[("Deployment", "vllmrouter-sample")]["spec"]["template"]["spec"]["containers"][0]["args"]
    assert args[args.index("--routing-logic") + 1] == "prefixaware"
    assert args[args.index("--k8s-label-selector") + 1] == "app=vllmruntime-sample"
    cs = render("CacheServer", {"metadata": {"name": "cacheserver-sample"},
                                "spec": {"image": {"name": "lmcache/vllm-openai:latest"}, "port": 8000, "replicas": 1}})
    cmd = cs[("Deployment", "cacheserver-sample")]["spec"]["template"]["spec"]["containers"][0]["command"]
    assert cmd == ["lmcache_experimental_server", "0.0.0.0", "8000"]


def test_operator_lora_placement_algorithms():
    cr = {"metadata": {"name": "l"}, "spec": {"baseModel": "llama", "adapterSource": {"adapterName": "ad", "adapterPath": "/data/ad"},
                                              "loraAdapterDeploymentConfig": {"algorithm": "default", "replicas": 2}}}
    pods = [{"name": "p2", "ip": "2", "ready": True, "adapters": ["x", "y"]},
            {"name": "p0", "ip": "0", "ready": True, "adapters": ["x"]},
            {"name": "p1", "ip": "1", "ready": False, "adapters": []},
            {"name": "p3", "ip": "3", "ready": True, "adapters": ["ad"]}]

    def plan(cr, deleting=False):
        out = subprocess.run([str(OPERATOR), "plan-lora"], input=json.dumps({"cr": cr, "pods": pods, "deleting": deleting}),
                             capture_output=True, text=True, check=True).stdout
        return sorted((a["op"], a["pod"]) for a in json.loads(out))

    assert plan(cr) == [("load", "p0"), ("load", "p2"), ("unload", "p3")]  # first two ready pods by name
    cr["spec"]["loraAdapterDeploymentConfig"]["algorithm"] = "equalized"
    assert plan(cr) == [("load", "p0")]  # p3 (already holding it) and the least loaded p0
    assert plan(cr, deleting=True) == [("unload", "p3")]


def test_operator_control_loop_against_fake_api():
    store = {}
    status = {}

    def app_factory():
        app = web.Application()

        async def list_crs(request):
            kind = request.match_info["plural"]
            items = [RUNTIME_CR] if kind == "vllmruntimes" else []
            return web.json_response({"items": items})

        async def get_obj(request):
            key = request.path
            if key in store:
                return web.json_response(store[key])
            return web.json_response({"reason": "NotFound"}, status=404)

        async def create(request):
            obj = await request.json()
            obj["metadata"]["resourceVersion"] = "1"
            store[request.path + "/" + obj["metadata"]["name"]] = obj
            return web.json_response(obj, status=201)

        async def put(request):
            obj = await request.json()
            store[request.path] = obj
            return web.json_response(obj)

        async def patch_status(request):
            status[request.path] = await request.json()
            return web.json_response({})

        app.router.add_get("/apis/production-stack.vllm.ai/v1alpha1/namespaces/default/{plural}", list_crs)
        app.router.add_patch("/apis/production-stack.vllm.ai/v1alpha1/namespaces/default/{plural}/{name}/status", patch_status)
        for root in ("/api/v1", "/apis/apps/v1"):
            app.router.add_get(root + "/namespaces/default/{res}/{name}", get_obj)
            app.router.add_post(root + "/namespaces/default/{res}", create)
            app.router.add_put(root + "/namespaces/default/{res}/{name}", put)
        return app

    loop, box = _serve(app_factory)
    try:
        api = f"http://127.0.0.1:{box['port']}"
        run = lambda: subprocess.run([str(OPERATOR), "run", "--api", api, "--namespace", "default", "--once"],
                                     capture_output=True, text=True, timeout=30)
        r = run()
        assert r.returncode == 0, r.stderr
        assert "/apis/apps/v1/namespaces/default/deployments/vllmruntime-sample" in store
        assert "/api/v1/namespaces/default/services/vllmruntime-sample" in store
        assert "/api/v1/namespaces/default/persistentvolumeclaims/vllmruntime-sample-pvc" in store
        st = status["/apis/production-stack.vllm.ai/v1alpha1/namespaces/default/vllmruntimes/vllmruntime-sample/status"]
        assert st["status"]["status"] in ("NotReady", "Updating", "Ready")
        # drift: someone scales the deployment; the next pass restores the desired replicas
        key = "/apis/apps/v1/namespaces/default/deployments/vllmruntime-sample"
        store[key]["spec"]["replicas"] = 7
        r = run()
        assert store[key]["spec"]["replicas"] == 2 and "updated" in r.stderr
        r = run()
        assert "updated" not in r.stderr and "created" not in r.stderr  # steady state is quiet
    finally:
        loop.call_soon_threadsafe(loop.stop)
