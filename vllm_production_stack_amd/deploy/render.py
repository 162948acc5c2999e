"""values.yaml -> Kubernetes objects, and values.yaml -> local process plan.

One renderer feeds three consumers: ``vpsa-stack template`` (manifests for clusters without
Helm), ``vpsa-stack up`` (one engine process per GPU group + router + cache server on a single
MI355X node), and the tests that pin the engine CLI / environment contract.  The Helm chart in
``deploy/helm`` renders the same objects from the same values; tests compare the two where a
``helm`` binary exists.

Behaviour being provided (values schema and rendered contract):
  /root/reference/helm/templates/deployment-vllm-multi.yaml:1-485   engine Deployment
  /root/reference/helm/templates/ray-cluster.yaml:1-622             multi-node pipeline parallel
  /root/reference/helm/templates/deployment-router.yaml:74-151      router args and probes
  /root/reference/helm/templates/deployment-cache-server.yaml:34-43 cache server command
  /root/reference/helm/templates/service-*.yaml, role.yaml, rolebinding.yaml, hpa-router.yaml,
  poddisruptionbudget.yaml, pvc.yaml, shared-storage.yaml, secrets.yaml, configmap.yaml,
  ingress.yaml, route.yaml, deployment-lora-adapter.yaml, deployment-lora-controller.yaml
  /root/reference/helm/templates/_helpers.tpl:38-197                strategies, resources, labels

Differences by design: the GPU resource defaults to ``amd.com/gpu``; multi-node pipeline
parallelism (``raySpec``) is a StatefulSet of ``torch.distributed`` ranks behind a headless
Service instead of a Ray cluster; P/D transfer settings named ``nixl*`` configure this stack's
page-push receiver (hipIpc over xGMI inside a node, RCCL or TCP across nodes).
"""

from __future__ import annotations

import copy
import json
from pathlib import Path
from typing import Any

import yaml

CHART_DIR = Path(__file__).resolve().parents[2] / "deploy" / "helm"
DEFAULT_GPU_RESOURCE = "amd.com/gpu"
ENGINE_IMAGE = ("vpsa/engine", "latest")


class ValuesError(ValueError):
    """A required value is missing or inconsistent."""


# --------------------------------------------------------------------------- values
def deep_merge(base: dict, over: dict) -> dict:
    out = copy.deepcopy(base)
    for k, v in (over or {}).items():
        if isinstance(v, dict) and isinstance(out.get(k), dict):
            out[k] = deep_merge(out[k], v)
        else:
            out[k] = copy.deepcopy(v)
    return out


def chart_defaults() -> dict:
    with open(CHART_DIR / "values.yaml") as f:
        return yaml.safe_load(f) or {}


def load_values(*paths: str | Path, overrides: dict | None = None) -> dict:
    vals = chart_defaults()
    for p in paths:
        with open(p) as f:
            vals = deep_merge(vals, yaml.safe_load(f) or {})
    if overrides:
        vals = deep_merge(vals, overrides)
    return vals


def _require(d: dict, key: str, where: str) -> Any:
    if key not in d or d[key] in (None, ""):
        raise ValuesError(f"Required value '{where}.{key}' must be defined")
    return d[key]


def _truthy(v: Any) -> str:
    return "True" if v else "False"


# --------------------------------------------------------------------------- engine contract
def kv_role(lm: dict) -> str:
    return {"sender": "kv_producer", "receiver": "kv_consumer"}.get(str(lm.get("nixlRole", "")), "kv_both")


def kv_transfer_json(ms: dict) -> str | None:
    lm = ms.get("lmcacheConfig") or {}
    if not lm.get("enabled"):
        return None
    cfg: dict[str, Any] = {"kv_connector": "LMCacheConnectorV1", "kv_role": kv_role(lm)}
    if "enablePD" in lm:
        cfg["kv_connector_extra_config"] = {
            "discard_partial_chunks": False, "lmcache_rpc_port": str(lm.get("nixlRole", ""))}
    elif str((ms.get("vllmConfig") or {}).get("v0", "")) == "1":
        cfg["kv_connector"] = "LMCacheConnector"
    return json.dumps(cfg, separators=(",", ":"))


def engine_args(values: dict, ms: dict, *, model_path: str | None = None) -> list[str]:
    """``vpsa-engine`` argv for one modelSpec (without the program name)."""
    ses = values["servingEngineSpec"]
    port = str(ses.get("containerPort") or 8000)
    args = ["serve", str(model_path or _require(ms, "modelURL", "modelSpec")),
            "--host", "0.0.0.0", "--port", port]
    if ms.get("servedModelName"):
        args += ["--served-model-name", str(ms["servedModelName"])]
    if ms.get("enableLoRA"):
        args.append("--enable-lora")
    if ms.get("enableTool"):
        args.append("--enable-auto-tool-choice")
    if ms.get("toolCallParser"):
        args += ["--tool-call-parser", str(ms["toolCallParser"])]
    vc = ms.get("vllmConfig")
    if vc:
        if "enableChunkedPrefill" in vc:
            args.append("--enable-chunked-prefill" if vc["enableChunkedPrefill"] else "--no-enable-chunked-prefill")
        args.append("--enable-prefix-caching" if vc.get("enablePrefixCaching") else "--no-enable-prefix-caching")
        for key, flag in (("maxModelLen", "--max-model-len"), ("dtype", "--dtype"),
                          ("tensorParallelSize", "--tensor-parallel-size"),
                          ("pipelineParallelSize", "--pipeline-parallel-size"),
                          ("maxNumSeqs", "--max-num-seqs"),
                          ("gpuMemoryUtilization", "--gpu_memory_utilization"),
                          ("maxLoras", "--max_loras")):
            if key in vc:
                args += [flag, str(vc[key])]
        args += [str(a) for a in vc.get("extraArgs") or []]
    kvt = kv_transfer_json(ms)
    if kvt:
        args += ["--kv-transfer-config", kvt]
    if ms.get("chatTemplate"):
        args += ["--chat-template", f"/templates/{ms['chatTemplate']}"]
    return args


def _secret_ref(name: str, key: str) -> dict:
    return {"valueFrom": {"secretKeyRef": {"name": name, "key": key}}}


def engine_env(values: dict, ms: dict, release: str) -> list[dict]:
    """Container ``env`` list for one modelSpec."""
    ses = values["servingEngineSpec"]
    env: list[dict] = [
        {"name": "HF_HOME", "value": "/data" if "pvcStorage" in ms else "/tmp"},
        {"name": "POD_IP", "valueFrom": {"fieldRef": {"fieldPath": "status.podIP"}}},
        # dmabuf IPC is the only mode the MI355X host driver offers; RCCL and hipIpc need it
        {"name": "HSA_ENABLE_IPC_MODE_LEGACY", "value": "0"},
    ]
    vc = ms.get("vllmConfig")
    if vc:
        env.append({"name": "LMCACHE_LOG_LEVEL", "value": "DEBUG"})
        if str(vc.get("v0", "")) == "1":
            env.append({"name": "VLLM_USE_V1", "value": "0"})
        else:
            env.append({"name": "PROMETHEUS_MULTIPROC_DIR", "value": str(vc.get("prometheusMultiprocDir") or "/tmp")})
    tok = ms.get("hf_token")
    if tok:
        ref = (_secret_ref(f"{release}-secrets", f"hf_token_{ms['name']}") if isinstance(tok, str)
               else _secret_ref(tok["secretName"], tok["secretKey"]))
        env.append({"name": "HF_TOKEN", **ref})
    key = ses.get("vllmApiKey")
    if key:
        ref = (_secret_ref(f"{release}-secrets", "vllmApiKey") if isinstance(key, str)
               else _secret_ref(key["secretName"], key["secretKey"]))
        env.append({"name": "VLLM_API_KEY", **ref})
    env += copy.deepcopy(ms.get("env") or [])
    lm = ms.get("lmcacheConfig")
    if lm:
        if lm.get("enabled"):
            env += [{"name": "LMCACHE_USE_EXPERIMENTAL", "value": "True"},
                    {"name": "VLLM_RPC_TIMEOUT", "value": "1000000"}]
        vis = lm.get("hipVisibleDevices", lm.get("cudaVisibleDevices"))
        if vis is not None:
            env.append({"name": "HIP_VISIBLE_DEVICES", "value": str(vis)})
        if lm.get("enablePD"):
            env += [{"name": "LMCACHE_LOCAL_CPU", "value": "False"},
                    {"name": "LMCACHE_MAX_LOCAL_CPU_SIZE", "value": "0"},
                    {"name": "LMCACHE_REMOTE_SERDE", "value": "NULL"}]
            for k, name, conv in (
                    ("enableNixl", "LMCACHE_ENABLE_NIXL", _truthy),
                    ("nixlRole", "LMCACHE_NIXL_ROLE", str),
                    ("nixlPeerHost", "LMCACHE_NIXL_RECEIVER_HOST", str),
                    ("nixlPeerPort", "LMCACHE_NIXL_RECEIVER_PORT", str),
                    ("nixlBufferSize", "LMCACHE_NIXL_BUFFER_SIZE", str),
                    ("nixlBufferDevice", "LMCACHE_NIXL_BUFFER_DEVICE", str),
                    ("nixlEnableGc", "LMCACHE_NIXL_ENABLE_GC", _truthy)):
                if k in lm:
                    env.append({"name": name, "value": conv(lm[k])})
        if "cpuOffloadingBufferSize" in lm and int(float(lm["cpuOffloadingBufferSize"])) > 0:
            env += [{"name": "LMCACHE_LOCAL_CPU", "value": "True"},
                    {"name": "LMCACHE_MAX_LOCAL_CPU_SIZE", "value": str(lm["cpuOffloadingBufferSize"])}]
        if "diskOffloadingBufferSize" in lm:
            env.append({"name": "LMCACHE_MAX_LOCAL_DISK_SIZE", "value": str(lm["diskOffloadingBufferSize"])})
        cs = values.get("cacheserverSpec")
        if cs:
            env += [{"name": "LMCACHE_REMOTE_URL",
                     "value": f"lm://{release}-cache-server-service:{cs.get('servicePort', 81)}"},
                    {"name": "LMCACHE_REMOTE_SERDE", "value": str(cs.get("serde", "naive"))}]
        for k, name, conv in (
                ("enableController", "LMCACHE_ENABLE_CONTROLLER", _truthy),
                ("instanceId", "LMCACHE_LMCACHE_INSTANCE_ID", str),
                ("workerPort", "LMCACHE_LMCACHE_WORKER_PORT", str),
                ("distributedUrl", "LMCACHE_DISTRIBUTED_URL", str)):
            if k in lm:
                env.append({"name": name, "value": conv(lm[k])})
        if "controllerPort" in lm:
            env.append({"name": "LMCACHE_CONTROLLER_URL",
                        "value": f"{release}-router-service:{lm['controllerPort']}"})
    return env


def gpu_count(ms: dict) -> int:
    try:
        return int(ms.get("requestGPU", 0) or 0)
    except (TypeError, ValueError):
        return 0


def engine_resources(ms: dict) -> dict:
    gpu_type = ms.get("requestGPUType") or DEFAULT_GPU_RESOURCE
    req: dict[str, Any] = {"memory": str(_require(ms, "requestMemory", "modelSpec")),
                           "cpu": str(_require(ms, "requestCPU", "modelSpec"))}
    n = gpu_count(ms)
    if n > 0:
        req[gpu_type] = str(n)
    lim: dict[str, Any] = {}
    if "limitMemory" in ms:
        lim["memory"] = str(ms["limitMemory"])
    if "limitCPU" in ms:
        lim["cpu"] = str(ms["limitCPU"])
    if n > 0:
        lim[gpu_type] = str(n)
    # fractional-GPU extended resources (HAMi-style) pass through under the same vendor prefix
    vendor = gpu_type.rsplit("/", 1)[0]
    for k, suffix in (("GPUMem", "gpumem"), ("GPUMemPercentage", "gpumem-percentage"), ("GPUCores", "gpucores")):
        if f"request{k}" in ms:
            req[f"{vendor}/{suffix}"] = str(ms[f"request{k}"])
        if f"limit{k}" in ms:
            lim[f"{vendor}/{suffix}"] = str(ms[f"limit{k}"])
    out = {"requests": req}
    if lim:
        out["limits"] = lim
    return out


def _strategy(spec: dict) -> dict:
    return copy.deepcopy(spec.get("strategy")) or {"rollingUpdate": {"maxSurge": "100%", "maxUnavailable": 0}}


def label_selector(labels: dict) -> str:
    return ",".join(f"{k}={v}" for k, v in sorted((labels or {}).items()))


# --------------------------------------------------------------------------- router contract
def router_args(values: dict, release: str, namespace: str) -> list[str]:
    rs = values["routerSpec"]
    ses = values["servingEngineSpec"]
    disc = rs.get("serviceDiscovery") or "k8s"
    args = ["--host", "0.0.0.0", "--port", str(rs.get("containerPort", 8000)), "--service-discovery", disc]
    if disc == "k8s":
        args += ["--k8s-namespace", namespace]
        if ses.get("enableEngine"):
            args += ["--k8s-label-selector", label_selector(ses.get("labels") or {})]
    elif disc == "static":
        args += ["--static-backends", str(_require(rs, "staticBackends", "routerSpec")),
                 "--static-models", str(_require(rs, "staticModels", "routerSpec"))]
    args += ["--routing-logic", str(rs.get("routingLogic") or "roundrobin")]
    if rs.get("sessionKey"):
        args += ["--session-key", str(rs["sessionKey"])]
    if rs.get("engineScrapeInterval"):
        args += ["--engine-stats-interval", str(rs["engineScrapeInterval"])]
    if rs.get("requestStatsWindow"):
        args += ["--request-stats-window", str(rs["requestStatsWindow"])]
    args += [str(a) for a in rs.get("extraArgs") or []]
    if rs.get("lmcacheControllerPort"):
        args += ["--lmcache-controller-port", str(rs["lmcacheControllerPort"])]
    return args


# --------------------------------------------------------------------------- manifests
def _meta(name: str, namespace: str | None, labels: dict | None = None, annotations: dict | None = None) -> dict:
    m: dict[str, Any] = {"name": name}
    if namespace:
        m["namespace"] = namespace
    if labels:
        m["labels"] = dict(labels)
    if annotations:
        m["annotations"] = dict(annotations)
    return m


def _engine_pod(values: dict, ms: dict, release: str, chart: str, labels: dict, *, extra_args=(),
                extra_env=(), command_prefix=("vpsa-engine",)) -> dict:
    ses = values["servingEngineSpec"]
    port = int(ses.get("containerPort") or 8000)
    repo = ms.get("repository") or ENGINE_IMAGE[0]
    tag = ms.get("tag") or ENGINE_IMAGE[1]
    tp = "tensorParallelSize" in (ms.get("vllmConfig") or {})
    mounts, volumes = [], []
    if "pvcStorage" in ms:
        mounts.append({"name": f"{release}-storage", "mountPath": "/data"})
        volumes.append({"name": f"{release}-storage",
                        "persistentVolumeClaim": {"claimName": f"{release}-{ms['name']}-storage-claim"}})
    if "sharedPvcStorage" in values and values["sharedPvcStorage"]:
        mounts.append({"name": f"{release}-shared-pvc-storage", "mountPath": "/data/shared-pvc-storage"})
        volumes.append({"name": f"{release}-shared-pvc-storage",
                        "persistentVolumeClaim": {"claimName": f"{release}-shared-pvc-storage-claim"}})
    if tp:
        # RCCL bootstraps its intra-node rings through /dev/shm
        mounts.append({"name": "shm", "mountPath": "/dev/shm"})
        volumes.append({"name": "shm", "emptyDir": {"medium": "Memory", "sizeLimit": str(ms.get("shmSize") or "20Gi")}})
    if ms.get("chatTemplate"):
        if "chatTemplateConfigMap" in ms:
            vol = f"{release}-{ms['name']}-chat-templates"
            volumes.append({"name": vol, "configMap": {"name": vol}})
        else:
            vol = "vllm-templates"
            volumes.append({"name": vol, "persistentVolumeClaim": {"claimName": "vllm-templates-pvc"}})
        mounts.append({"name": vol, "mountPath": "/templates"})
    mounts += copy.deepcopy(ms.get("extraVolumeMounts") or [])
    volumes += copy.deepcopy(ms.get("extraVolumes") or [])

    container: dict[str, Any] = {
        "name": "vllm",
        "image": f"{repo}:{tag}",
        "imagePullPolicy": ses.get("imagePullPolicy") or "IfNotPresent",
        "securityContext": copy.deepcopy(ses.get("containerSecurityContext") or {}),
        "command": [*command_prefix, *engine_args(values, ms), *extra_args],
        "env": engine_env(values, ms, release) + list(extra_env),
        "ports": [{"name": "container-port", "containerPort": port},
                  {"name": "zmq-port", "containerPort": 55555},
                  {"name": "ucx-port", "containerPort": 9999},
                  *copy.deepcopy(ses.get("extraPorts") or [])],
        "resources": engine_resources(ms),
    }
    env_from = []
    if ses.get("configs"):
        env_from.append({"configMapRef": {"name": f"{release}-configs"}})
    if ms.get("envFromSecret"):
        env_from.append({"secretRef": {"name": ms["envFromSecret"]["name"]}})
    if env_from:
        container["envFrom"] = env_from
    for probe in ("startupProbe", "livenessProbe"):
        if ses.get(probe):
            container[probe] = copy.deepcopy(ses[probe])
    if mounts:
        container["volumeMounts"] = mounts
    containers = [container]
    if ms.get("enableLoRA") and "pvcStorage" in ms:
        sc = ses.get("sidecar") or {}
        containers.append({
            "name": "sidecar", "image": sc.get("image", "vpsa/sidecar:latest"),
            "imagePullPolicy": sc.get("imagePullPolicy", "IfNotPresent"),
            "command": ["python", "-m", "vllm_production_stack_amd.sidecar"],
            "env": [{"name": "PORT", "value": "30090"},
                    {"name": "LORA_DOWNLOAD_BASE_DIR", "value": "/data/lora-adapters"}],
            "volumeMounts": [{"name": f"{release}-storage", "mountPath": "/data"}]})
    spec: dict[str, Any] = {"containers": containers}
    if ms.get("initContainer"):
        ic = ms["initContainer"]
        init: dict[str, Any] = {"name": ic.get("name", "init"), "image": _require(ic, "image", "initContainer")}
        for k in ("command", "args", "env", "resources"):
            if ic.get(k):
                init[k] = copy.deepcopy(ic[k])
        if ic.get("mountPvcStorage") and "pvcStorage" in ms:
            init["volumeMounts"] = [{"name": f"{release}-storage", "mountPath": "/data"}]
        spec["initContainers"] = [init]
    if ses.get("securityContext"):
        spec["securityContext"] = copy.deepcopy(ses["securityContext"])
    if ms.get("priorityClassName"):
        spec["priorityClassName"] = ms["priorityClassName"]
    if ms.get("imagePullSecret"):
        spec["imagePullSecrets"] = [{"name": ms["imagePullSecret"]}]
    if volumes:
        spec["volumes"] = volumes
    if ses.get("tolerations"):
        spec["tolerations"] = copy.deepcopy(ses["tolerations"])
    if ses.get("runtimeClassName"):
        spec["runtimeClassName"] = ses["runtimeClassName"]
    if ses.get("schedulerName"):
        spec["schedulerName"] = ses["schedulerName"]
    if ms.get("nodeName"):
        spec["nodeName"] = ms["nodeName"]
    elif ms.get("nodeSelectorTerms"):
        spec["affinity"] = {"nodeAffinity": {"requiredDuringSchedulingIgnoredDuringExecution": {
            "nodeSelectorTerms": copy.deepcopy(ms["nodeSelectorTerms"])}}}
    return {"metadata": {"labels": labels, **({"annotations": ms["podAnnotations"]} if ms.get("podAnnotations") else {})},
            "spec": spec}


def render_engine(values: dict, ms: dict, release: str, namespace: str, chart: str) -> list[dict]:
    ses = values["servingEngineSpec"]
    _require(ms, "name", "modelSpec")
    labels = {"model": ms["name"], "helm-release-name": release, **(ses.get("labels") or {})}
    out: list[dict] = []
    ray = ms.get("raySpec")
    if ray:
        out += _render_multinode(values, ms, release, namespace, chart, labels)
    else:
        pod = _engine_pod(values, ms, release, chart, labels)
        pod["spec"]["hostname"] = f"{release}-{ms['name']}-{chart}"[:63]
        pod["spec"]["subdomain"] = f"{release}-{ms['name']}-engine-service"
        dep = {
            "apiVersion": "apps/v1", "kind": "Deployment",
            "metadata": _meta(f"{release}-{ms['name']}-deployment-vllm", namespace, labels),
            "spec": {"replicas": int(ms.get("replicaCount", 1)), "strategy": _strategy(ses),
                     "selector": {"matchLabels": labels}, "progressDeadlineSeconds": 1200, "template": pod}}
        out.append(dep)
    if ms.get("chatTemplate") and "chatTemplateConfigMap" in ms:
        out.append({"apiVersion": "v1", "kind": "ConfigMap",
                    "metadata": _meta(f"{release}-{ms['name']}-chat-templates", namespace),
                    "data": {ms["chatTemplate"]: ms["chatTemplateConfigMap"]}})
    # per-model Service: stable name for P/D peers and for debugging one model directly
    out.append({
        "apiVersion": "v1", "kind": "Service",
        "metadata": _meta(f"{release}-{ms['name']}-engine-service", namespace, labels),
        "spec": {"type": "ClusterIP", "selector": labels, "ports": [
            {"name": "service-port", "port": int(ses.get("servicePort") or ses.get("containerPort") or 8000),
             "targetPort": "container-port", "protocol": "TCP"},
            {"name": "zmq-port", "port": 55555, "targetPort": "zmq-port", "protocol": "TCP"},
            {"name": "ucx-port", "port": 9999, "targetPort": "ucx-port", "protocol": "TCP"}]}})
    if "pvcStorage" in ms:
        pvc: dict[str, Any] = {"accessModes": list(ms.get("pvcAccessMode") or ["ReadWriteOnce"]),
                               "resources": {"requests": {"storage": str(ms["pvcStorage"])}}}
        if "storageClass" in ms:
            pvc["storageClassName"] = ms["storageClass"]
        if ms.get("pvcMatchLabels"):
            pvc["selector"] = {"matchLabels": dict(ms["pvcMatchLabels"])}
        out.append({"apiVersion": "v1", "kind": "PersistentVolumeClaim",
                    "metadata": _meta(f"{release}-{ms['name']}-storage-claim", namespace),
                    "spec": pvc})
    return out


def _render_multinode(values, ms, release, namespace, chart, labels) -> list[dict]:
    """``raySpec`` -> StatefulSet of torch.distributed ranks (pipeline stages across nodes,
    tensor parallel inside a node over xGMI).  Rank 0 serves HTTP; the headless Service gives
    every rank a stable DNS name for the RCCL/TCP rendezvous."""
    ray = ms["raySpec"]
    head = ray.get("headNode") or {}
    vc = ms.get("vllmConfig") or {}
    pp = int(vc.get("pipelineParallelSize") or 1)
    tp = int(vc.get("tensorParallelSize") or 1)
    nodes = max(1, (pp * tp + max(gpu_count(ms), 1) - 1) // max(gpu_count(ms), 1))
    svc = f"{release}-{ms['name']}-ranks"
    sts = f"{release}-{ms['name']}-engine"
    extra_args = ["--distributed-executor-backend", "torchrun", "--nnodes", str(nodes),
                  "--master-addr", f"{sts}-0.{svc}", "--master-port", "29500"]
    extra_env = [{"name": "POD_NAME", "valueFrom": {"fieldRef": {"fieldPath": "metadata.name"}}},
                 {"name": "NCCL_SOCKET_IFNAME", "value": str(ray.get("socketIfname", "eth0"))}]
    pod = _engine_pod(values, ms, release, chart, {**labels, "role": "engine-rank"},
                      extra_args=extra_args, extra_env=extra_env)
    if head.get("requestCPU") or head.get("requestMemory"):
        res = pod["spec"]["containers"][0]["resources"]["requests"]
        res["cpu"] = str(head.get("requestCPU", res["cpu"]))
        res["memory"] = str(head.get("requestMemory", res["memory"]))
    out = [{
        "apiVersion": "v1", "kind": "Service",
        "metadata": _meta(svc, namespace, labels),
        "spec": {"clusterIP": "None", "publishNotReadyAddresses": True,
                 "selector": {**labels, "role": "engine-rank"},
                 "ports": [{"name": "rendezvous", "port": 29500}]}}, {
        "apiVersion": "apps/v1", "kind": "StatefulSet",
        "metadata": _meta(sts, namespace, labels),
        "spec": {"serviceName": svc, "replicas": nodes * int(ms.get("replicaCount", 1)),
                 "podManagementPolicy": "Parallel",
                 "selector": {"matchLabels": {**labels, "role": "engine-rank"}}, "template": pod}}]
    return out


def render_router(values: dict, release: str, namespace: str) -> list[dict]:
    rs = values["routerSpec"]
    if not rs.get("enableRouter"):
        return []
    labels = dict(rs.get("labels") or {})
    port = int(rs.get("containerPort", 8000))
    env: list[dict] = []
    if rs.get("hf_token"):
        tok = rs["hf_token"]
        ref = (_secret_ref(f"{release}-secrets", "hf_token_router") if isinstance(tok, str)
               else _secret_ref(tok["secretName"], tok["secretKey"]))
        env.append({"name": "HF_TOKEN", **ref})
    key = values["servingEngineSpec"].get("vllmApiKey") if values["servingEngineSpec"].get("enableEngine") \
        else rs.get("vllmApiKey")
    if key:
        ref = (_secret_ref(f"{release}-secrets", "vllmApiKey") if isinstance(key, str)
               else _secret_ref(key["secretName"], key["secretKey"]))
        env.append({"name": "VLLM_API_KEY", **ref})
    env.extend(copy.deepcopy(rs.get("env") or []))  # e.g. OTEL_* for router spans
    probe = {"httpGet": {"path": "/health", "port": port}, "periodSeconds": 5, "failureThreshold": 3}
    container: dict[str, Any] = {
        "name": "router-container", "image": f"{rs['repository']}:{rs['tag']}",
        "imagePullPolicy": rs.get("imagePullPolicy", "IfNotPresent"),
        "command": ["vpsa-router"], "args": router_args(values, release, namespace), "env": env,
        "ports": [{"name": "router-cport", "containerPort": port},
                  {"name": "lmcache-port", "containerPort": 9000}],
        "livenessProbe": {**probe, "initialDelaySeconds": 30},
        "startupProbe": {**probe, "initialDelaySeconds": 5}}
    if rs.get("resources"):
        container["resources"] = copy.deepcopy(rs["resources"])
    spec: dict[str, Any] = {"serviceAccountName": f"{release}-router-service-account", "containers": [container]}
    if rs.get("priorityClassName"):
        spec["priorityClassName"] = rs["priorityClassName"]
    if rs.get("nodeSelectorTerms"):
        spec["affinity"] = {"nodeAffinity": {"requiredDuringSchedulingIgnoredDuringExecution": {
            "nodeSelectorTerms": copy.deepcopy(rs["nodeSelectorTerms"])}}}
    dep_spec: dict[str, Any] = {"strategy": _strategy(rs), "selector": {"matchLabels": labels},
                                "template": {"metadata": {"labels": labels}, "spec": spec}}
    if not (rs.get("autoscaling") or {}).get("enabled"):
        dep_spec["replicas"] = int(rs.get("replicaCount", 1))
    out = [
        {"apiVersion": "v1", "kind": "ServiceAccount",
         "metadata": _meta(f"{release}-router-service-account", namespace)},
        {"apiVersion": "rbac.authorization.k8s.io/v1", "kind": "Role",
         "metadata": _meta(f"{release}-pod-reader", namespace),
         "rules": [{"apiGroups": [""], "resources": ["pods"], "verbs": ["get", "watch", "list", "patch"]},
                   {"apiGroups": [""], "resources": ["services", "endpoints"], "verbs": ["get", "watch", "list"]}]},
        {"apiVersion": "rbac.authorization.k8s.io/v1", "kind": "RoleBinding",
         "metadata": _meta(f"{release}-deployment-access-binding", namespace),
         "subjects": [{"kind": "ServiceAccount", "name": f"{release}-router-service-account", "namespace": namespace}],
         "roleRef": {"kind": "Role", "name": f"{release}-pod-reader", "apiGroup": "rbac.authorization.k8s.io"}},
        {"apiVersion": "apps/v1", "kind": "Deployment",
         "metadata": _meta(f"{release}-deployment-router", namespace, labels), "spec": dep_spec},
        {"apiVersion": "v1", "kind": "Service",
         "metadata": _meta(f"{release}-router-service", namespace, labels, rs.get("serviceAnnotations")),
         "spec": {"type": rs.get("serviceType", "ClusterIP"), "selector": labels, "ports": [
             {"name": "router-sport", "port": int(rs.get("servicePort", 80)), "targetPort": port, "protocol": "TCP"},
             {"name": "lmcache-port", "port": 9000, "targetPort": "lmcache-port", "protocol": "TCP"}]}},
    ]
    auto = rs.get("autoscaling") or {}
    if auto.get("enabled"):
        out.append({"apiVersion": "autoscaling/v2", "kind": "HorizontalPodAutoscaler",
                    "metadata": _meta(f"{release}-router-hpa", namespace),
                    "spec": {"scaleTargetRef": {"apiVersion": "apps/v1", "kind": "Deployment",
                                                "name": f"{release}-deployment-router"},
                             "minReplicas": int(auto.get("minReplicas", 1)),
                             "maxReplicas": int(auto.get("maxReplicas", 3)),
                             "metrics": [{"type": "Resource", "resource": {"name": "cpu", "target": {
                                 "type": "Utilization",
                                 "averageUtilization": int(auto.get("targetCPUUtilizationPercentage", 80))}}}]}})
    ing = rs.get("ingress") or {}
    if ing.get("enabled"):
        rules = []
        for h in ing.get("hosts") or []:
            rules.append({"host": h.get("host"), "http": {"paths": [
                {"path": p.get("path", "/"), "pathType": p.get("pathType", "Prefix"),
                 "backend": {"service": {"name": f"{release}-router-service",
                                         "port": {"number": int(rs.get("servicePort", 80))}}}}
                for p in h.get("paths") or []]}})
        spec_i: dict[str, Any] = {"rules": rules}
        if ing.get("className"):
            spec_i["ingressClassName"] = ing["className"]
        if ing.get("tls"):
            spec_i["tls"] = copy.deepcopy(ing["tls"])
        out.append({"apiVersion": "networking.k8s.io/v1", "kind": "Ingress",
                    "metadata": _meta(f"{release}-ingress-router", namespace, labels, ing.get("annotations")),
                    "spec": spec_i})
    for name, route in (rs.get("route") or {}).items():
        if not (route or {}).get("enabled"):
            continue
        rules = copy.deepcopy(route.get("additionalRules") or [])
        if route.get("httpsRedirect"):
            rules.append({"filters": [{"type": "RequestRedirect",
                                       "requestRedirect": {"scheme": "https", "statusCode": 301}}]})
        else:
            rule: dict[str, Any] = {"backendRefs": [{"name": f"{release}-router-service",
                                                     "port": int(rs.get("servicePort", 80))}]}
            if route.get("filters"):
                rule["filters"] = copy.deepcopy(route["filters"])
            if route.get("matches"):
                rule["matches"] = copy.deepcopy(route["matches"])
            rules.append(rule)
        rspec: dict[str, Any] = {"rules": rules}
        if route.get("parentRefs"):
            rspec["parentRefs"] = copy.deepcopy(route["parentRefs"])
        if route.get("hostnames"):
            rspec["hostnames"] = list(route["hostnames"])
        out.append({"apiVersion": route.get("apiVersion") or "gateway.networking.k8s.io/v1",
                    "kind": route.get("kind") or "HTTPRoute",
                    "metadata": _meta(f"{release}-router" + ("" if name == "main" else f"-{name}"), namespace,
                                      {**labels, **(route.get("labels") or {})}, route.get("annotations")),
                    "spec": rspec})
    return out


def render_cache_server(values: dict, release: str, namespace: str) -> list[dict]:
    cs = values.get("cacheserverSpec")
    if not cs:
        return []
    labels = dict(cs.get("labels") or {"app": f"{release}-cache-server"})
    port = int(cs.get("containerPort", 8080))
    cmd = ["vpsa-cache-server", "0.0.0.0", str(port)]
    if cs.get("maxBytes"):
        cmd += ["--max-bytes", str(cs["maxBytes"])]
    container: dict[str, Any] = {
        "name": "cache-server-container",
        "image": f"{cs.get('repository', 'vpsa/cache-server')}:{cs.get('tag', 'latest')}",
        "imagePullPolicy": cs.get("imagePullPolicy", "IfNotPresent"), "command": cmd,
        "ports": [{"name": "cache-port", "containerPort": port}]}
    if cs.get("resources"):
        container["resources"] = copy.deepcopy(cs["resources"])
    spec: dict[str, Any] = {"containers": [container]}
    if cs.get("nodeSelectorTerms"):
        spec["affinity"] = {"nodeAffinity": {"requiredDuringSchedulingIgnoredDuringExecution": {
            "nodeSelectorTerms": copy.deepcopy(cs["nodeSelectorTerms"])}}}
    return [
        {"apiVersion": "apps/v1", "kind": "Deployment",
         "metadata": _meta(f"{release}-deployment-cache-server", namespace, labels),
         "spec": {"replicas": int(cs.get("replicaCount", 1)), "strategy": _strategy(cs),
                  "selector": {"matchLabels": labels},
                  "template": {"metadata": {"labels": labels}, "spec": spec}}},
        {"apiVersion": "v1", "kind": "Service",
         "metadata": _meta(f"{release}-cache-server-service", namespace, labels),
         "spec": {"type": "ClusterIP", "selector": labels,
                  "ports": [{"name": "cache-port", "port": int(cs.get("servicePort", 81)),
                             "targetPort": port, "protocol": "TCP"}]}}]


def render_shared(values: dict, release: str, namespace: str) -> list[dict]:
    ses = values["servingEngineSpec"]
    out: list[dict] = []
    # Secret
    data: dict[str, str] = {}
    if isinstance(ses.get("vllmApiKey"), str) and ses["vllmApiKey"]:
        data["vllmApiKey"] = ses["vllmApiKey"]
    for ms in ses.get("modelSpec") or []:
        if isinstance(ms.get("hf_token"), str) and ms["hf_token"]:
            data[f"hf_token_{ms['name']}"] = ms["hf_token"]
    rtok = values["routerSpec"].get("hf_token")
    if isinstance(rtok, str) and rtok:
        data["hf_token_router"] = rtok
    for ad in values.get("loraAdapters") or []:
        cred = (ad.get("adapterSource") or {}).get("credentials")
        if isinstance(cred, str) and cred:
            data[f"lora_adapter_credentials_{ad['adapterSource']['adapterName']}"] = cred
    if data:
        out.append({"apiVersion": "v1", "kind": "Secret", "type": "Opaque",
                    "metadata": _meta(f"{release}-secrets", namespace), "stringData": data})
    if ses.get("configs"):
        out.append({"apiVersion": "v1", "kind": "ConfigMap",
                    "metadata": _meta(f"{release}-configs", namespace),
                    "data": {k: str(v) for k, v in ses["configs"].items()}})
    pdb = ses.get("maxUnavailablePodDisruptionBudget")
    if pdb not in (None, ""):
        out.append({"apiVersion": "policy/v1", "kind": "PodDisruptionBudget",
                    "metadata": _meta(f"{release}-pdb", namespace),
                    "spec": {"maxUnavailable": pdb, "selector": {"matchLabels": dict(ses.get("labels") or {})}}})
    shared = values.get("sharedPvcStorage")
    if shared:
        modes = list(shared.get("accessModes") or ["ReadWriteMany"])
        pv_spec: dict[str, Any] = {"storageClassName": shared.get("storageClass", "standard"),
                                   "capacity": {"storage": shared.get("size", "100Gi")}, "accessModes": modes}
        if shared.get("hostPath"):
            pv_spec["hostPath"] = {"path": shared["hostPath"]}
        else:
            nfs = shared.get("nfs") or {}
            pv_spec["nfs"] = {"server": _require(nfs, "server", "sharedPvcStorage.nfs"),
                              "path": _require(nfs, "path", "sharedPvcStorage.nfs")}
        lab = {"app": release, "component": "shared-pvc-storage"}
        out.append({"apiVersion": "v1", "kind": "PersistentVolume",
                    "metadata": _meta(f"{release}-shared-pvc-storage", None, {"type": "local", **lab}),
                    "spec": pv_spec})
        out.append({"apiVersion": "v1", "kind": "PersistentVolumeClaim",
                    "metadata": _meta(f"{release}-shared-pvc-storage-claim", namespace, lab),
                    "spec": {"storageClassName": shared.get("storageClass", "standard"), "accessModes": modes,
                             "resources": {"requests": {"storage": shared.get("size", "100Gi")}},
                             "selector": {"matchLabels": {"app": release, "component": "shared-pvc-storage"}}}})
    return out


def render_lora(values: dict, release: str, namespace: str) -> list[dict]:
    out: list[dict] = []
    ses = values["servingEngineSpec"]
    for ad in values.get("loraAdapters") or []:
        src = ad.get("adapterSource") or {}
        s: dict[str, Any] = {"type": str(_require(src, "type", "loraAdapters.adapterSource")),
                             "adapterName": str(_require(src, "adapterName", "loraAdapters.adapterSource"))}
        for k in ("adapterPath", "repository", "pattern", "maxAdapters"):
            if src.get(k):
                s[k] = src[k]
        cred = src.get("credentials")
        if cred:
            s["credentialsSecretRef"] = ({"name": f"{release}-secrets",
                                          "key": f"lora_adapter_credentials_{src['adapterName']}"}
                                         if isinstance(cred, str)
                                         else {"name": cred["secretName"], "key": cred["secretKey"]})
        spec: dict[str, Any] = {"baseModel": str(_require(ad, "baseModel", "loraAdapters"))}
        if ad.get("vllmApiKey"):
            spec["vllmApiKey"] = {"secretName": ad["vllmApiKey"]["secretName"],
                                  "secretKey": ad["vllmApiKey"]["secretKey"]}
        elif ses.get("vllmApiKey"):
            spec["vllmApiKey"] = {"secretName": f"{release}-secrets", "secretKey": "vllmApiKey"}
        spec["adapterSource"] = s
        dc = ad.get("loraAdapterDeploymentConfig") or {}
        spec["loraAdapterDeploymentConfig"] = {"algorithm": dc.get("algorithm") or "default"}
        if dc.get("replicas"):
            spec["loraAdapterDeploymentConfig"]["replicas"] = int(dc["replicas"])
        out.append({"apiVersion": "production-stack.vllm.ai/v1alpha1", "kind": "LoraAdapter",
                    "metadata": _meta(_require(ad, "name", "loraAdapters"), namespace,
                                      {"app.kubernetes.io/component": "lora-adapter",
                                       "helm-release-name": release}),
                    "spec": spec})
    lc = values.get("loraController") or {}
    if lc.get("enableLoraController"):
        name = f"{release}-lora-controller"
        labels = {"app.kubernetes.io/component": "lora-controller", "helm-release-name": release}
        img = lc.get("image") or {}
        metrics = lc.get("metrics") or {}
        cert_dir = "/tmp/k8s-metrics-server/metrics-certs"
        # no --api: inside the pod the operator finds the API server from the environment and
        # authenticates with its service account (CA bundle + token from the projected volume)
        args = ["run", "--namespace", namespace, "--gpu-resource", DEFAULT_GPU_RESOURCE, "--health-port", "8081"]
        if metrics.get("enabled"):
            args += ["--metrics-bind-address", ":8443"]
            if metrics.get("certSecret"):
                args += ["--metrics-cert-path", cert_dir]
        args += [str(a) for a in lc.get("extraArgs") or []]
        ctr: dict[str, Any] = {
            "name": "manager", "image": f"{img.get('repository', 'vpsa/operator')}:{img.get('tag', 'latest')}",
            "imagePullPolicy": img.get("pullPolicy", "IfNotPresent"),
            "command": ["vpsa-operator"], "args": args, "env": copy.deepcopy(lc.get("env") or []),
            "securityContext": copy.deepcopy(lc.get("containerSecurityContext") or {}),
            "livenessProbe": {"httpGet": {"path": "/healthz", "port": 8081},
                              "initialDelaySeconds": 15, "periodSeconds": 20},
            "readinessProbe": {"httpGet": {"path": "/readyz", "port": 8081},
                               "initialDelaySeconds": 5, "periodSeconds": 10}}
        if lc.get("resources"):
            ctr["resources"] = copy.deepcopy(lc["resources"])
        pod_spec: dict[str, Any] = {"serviceAccountName": name, "containers": [ctr],
                                    "securityContext": copy.deepcopy(lc.get("podSecurityContext") or {}),
                                    "terminationGracePeriodSeconds": 10}
        if metrics.get("enabled") and metrics.get("certSecret"):
            ctr["volumeMounts"] = [{"name": "metrics-certs", "mountPath": cert_dir, "readOnly": True}]
            pod_spec["volumes"] = [{"name": "metrics-certs", "secret": {"secretName": metrics["certSecret"]}}]
        for k in ("nodeSelector", "affinity", "tolerations", "imagePullSecrets"):
            if lc.get(k):
                pod_spec[k] = copy.deepcopy(lc[k])
        crd_rules = [
            {"apiGroups": ["production-stack.vllm.ai"],
             "resources": ["loraadapters", "vllmruntimes", "vllmrouters", "cacheservers"],
             "verbs": ["create", "delete", "get", "list", "patch", "update", "watch"]},
            {"apiGroups": ["production-stack.vllm.ai"],
             "resources": ["loraadapters/status", "vllmruntimes/status", "vllmrouters/status", "cacheservers/status",
                           "loraadapters/finalizers"],
             "verbs": ["get", "patch", "update"]},
            {"apiGroups": [""], "resources": ["pods", "services", "secrets", "configmaps",
                                              "persistentvolumeclaims", "serviceaccounts"],
             "verbs": ["get", "list", "watch", "create", "update", "patch", "delete"]},
            {"apiGroups": ["apps"], "resources": ["deployments"],
             "verbs": ["get", "list", "watch", "create", "update", "patch", "delete"]},
            {"apiGroups": ["rbac.authorization.k8s.io"], "resources": ["roles", "rolebindings"],
             "verbs": ["get", "list", "watch", "create", "update", "patch"]},
            {"apiGroups": ["coordination.k8s.io"], "resources": ["leases"],
             "verbs": ["get", "list", "watch", "create", "update", "patch", "delete"]},
            {"apiGroups": [""], "resources": ["events"], "verbs": ["create", "patch"]}]
        if metrics.get("enabled"):  # the secured metrics port reviews its callers' tokens
            crd_rules += [
                {"apiGroups": ["authentication.k8s.io"], "resources": ["tokenreviews"], "verbs": ["create"]},
                {"apiGroups": ["authorization.k8s.io"], "resources": ["subjectaccessreviews"], "verbs": ["create"]}]
        out += [
            {"apiVersion": "v1", "kind": "ServiceAccount", "metadata": _meta(name, namespace, labels)},
            {"apiVersion": "rbac.authorization.k8s.io/v1", "kind": "ClusterRole",
             "metadata": _meta(f"{name}-role", None, labels), "rules": crd_rules},
            {"apiVersion": "rbac.authorization.k8s.io/v1", "kind": "ClusterRoleBinding",
             "metadata": _meta(f"{name}-rolebinding", None, labels),
             "subjects": [{"kind": "ServiceAccount", "name": name, "namespace": namespace}],
             "roleRef": {"kind": "ClusterRole", "name": f"{name}-role", "apiGroup": "rbac.authorization.k8s.io"}},
            {"apiVersion": "apps/v1", "kind": "Deployment", "metadata": _meta(name, namespace, labels),
             "spec": {"replicas": int(lc.get("replicaCount", 1)), "selector": {"matchLabels": labels},
                      "template": {"metadata": {"labels": labels,
                                                **({"annotations": lc["podAnnotations"]}
                                                   if lc.get("podAnnotations") else {})},
                                   "spec": pod_spec}}}]
        if metrics.get("enabled"):
            out.append({"apiVersion": "rbac.authorization.k8s.io/v1", "kind": "ClusterRole",
                        "metadata": _meta(f"{name}-metrics-reader", None, labels),
                        "rules": [{"nonResourceURLs": ["/metrics"], "verbs": ["get"]}]})
            out.append({"apiVersion": "v1", "kind": "Service",
                        "metadata": _meta(f"{name}-metrics", namespace, labels),
                        "spec": {"selector": labels, "ports": [{"name": "https", "port": 8443, "targetPort": 8443}]}})
    return out


def render_all(values: dict, release: str = "vllm", namespace: str = "default",
               chart: str = "vllm-stack") -> list[dict]:
    ses = values["servingEngineSpec"]
    out: list[dict] = render_shared(values, release, namespace)
    if ses.get("enableEngine"):
        seen = set()
        for ms in ses.get("modelSpec") or []:
            if ms.get("name") in seen:
                raise ValuesError(f"duplicate modelSpec name {ms.get('name')!r}")
            seen.add(ms.get("name"))
            out += render_engine(values, ms, release, namespace, chart)
        if ses.get("modelSpec"):
            # one Service over every engine pod of the release (what the router-less path hits)
            out.append({"apiVersion": "v1", "kind": "Service",
                        "metadata": _meta(f"{release}-engine-service", namespace, dict(ses.get("labels") or {})),
                        "spec": {"type": "ClusterIP", "selector": dict(ses.get("labels") or {}), "ports": [
                            {"name": "service-port",
                             "port": int(ses.get("servicePort") or ses.get("containerPort") or 8000),
                             "targetPort": "container-port", "protocol": "TCP"}]}})
    out += render_router(values, release, namespace)
    out += render_cache_server(values, release, namespace)
    out += render_lora(values, release, namespace)
    return out


def to_yaml(objs: list[dict]) -> str:
    return "---\n".join(yaml.safe_dump(o, sort_keys=False, default_flow_style=False) for o in objs)
