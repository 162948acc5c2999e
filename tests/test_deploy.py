"""Deployment assets: the values renderer (engine CLI / env contract), the example values of every
feature, the chart templates' structure, CRDs, and dashboard queries vs. exported metric names."""

import json
import re
from pathlib import Path

import pytest
import yaml

from vllm_production_stack_amd.deploy import render
from vllm_production_stack_amd.deploy.stack import local_plan

ROOT = Path(__file__).resolve().parents[1]
EXAMPLES = sorted((ROOT / "deploy" / "examples").glob("values-*.yaml"))


def _objs(path, **kw):
    return render.render_all(render.load_values(path), **kw)


def _kinds(objs):
    return [(o["kind"], o["metadata"]["name"]) for o in objs]


def _engine_container(objs, name):
    for o in objs:
        if o["kind"] in ("Deployment", "StatefulSet") and o["metadata"]["name"].startswith(name):
            return o["spec"]["template"]["spec"]["containers"][0]
    raise AssertionError(f"no workload {name} in {_kinds(objs)}")


@pytest.mark.parametrize("path", EXAMPLES, ids=lambda p: p.stem)
def test_every_example_renders(path):
    objs = _objs(path)
    assert objs, path
    text = render.to_yaml(objs)
    for doc in yaml.safe_load_all(text):
        assert doc["apiVersion"] and doc["kind"] and doc["metadata"]["name"]
    # GPU requests always go through the AMD device-plugin resource
    for o in objs:
        if o["kind"] in ("Deployment", "StatefulSet"):
            for c in o["spec"]["template"]["spec"]["containers"]:
                for sect in (c.get("resources") or {}).values():
                    assert not any(k.startswith("nvidia.com") for k in sect)


def test_engine_command_and_env_contract():
    objs = _objs(ROOT / "deploy/examples/values-05-host-dram-offload.yaml")
    c = _engine_container(objs, "vllm-llama3-deployment")
    cmd = c["command"]
    assert cmd[:3] == ["vpsa-engine", "serve", "meta-llama/Llama-3.1-8B-Instruct"]
    for flag in ("--enable-prefix-caching", "--enable-chunked-prefill", "--max-model-len", "--dtype",
                 "--swap-space", "--kv-transfer-config"):
        assert flag in cmd
    kvt = json.loads(cmd[cmd.index("--kv-transfer-config") + 1])
    assert kvt == {"kv_connector": "LMCacheConnectorV1", "kv_role": "kv_both"}
    env = {e["name"]: e.get("value") for e in c["env"]}
    assert env["LMCACHE_LOCAL_CPU"] == "True" and env["LMCACHE_MAX_LOCAL_CPU_SIZE"] == "120"
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and "HF_TOKEN" in env
    assert c["resources"]["requests"]["amd.com/gpu"] == "1"
    ports = {p["name"]: p["containerPort"] for p in c["ports"]}
    assert ports == {"container-port": 8000, "zmq-port": 55555, "ucx-port": 9999}
    router = next(o for o in objs if o["metadata"]["name"] == "vllm-deployment-router")
    args = router["spec"]["template"]["spec"]["containers"][0]["args"]
    assert args[args.index("--routing-logic") + 1] == "session"
    assert args[args.index("--session-key") + 1] == "x-user-id"
    assert args[args.index("--k8s-label-selector") + 1] == "environment=test,release=test"


def test_disaggregated_prefill_roles_and_pools():
    objs = _objs(ROOT / "deploy/examples/values-16-disaggregated-prefill.yaml")
    p = _engine_container(objs, "vllm-llama-prefill-deployment")
    d = _engine_container(objs, "vllm-llama-decode-deployment")
    kp = json.loads(p["command"][p["command"].index("--kv-transfer-config") + 1])
    kd = json.loads(d["command"][d["command"].index("--kv-transfer-config") + 1])
    assert kp["kv_role"] == "kv_producer" and kd["kv_role"] == "kv_consumer"
    envp = {e["name"]: e.get("value") for e in p["env"]}
    assert envp["LMCACHE_NIXL_ROLE"] == "sender" and envp["LMCACHE_ENABLE_NIXL"] == "True"
    assert envp["LMCACHE_NIXL_RECEIVER_HOST"] == "vllm-llama-decode-engine-service"
    reps = {o["metadata"]["name"]: o["spec"]["replicas"] for o in objs if o["kind"] == "Deployment"}
    assert reps["vllm-llama-prefill-deployment-vllm"] == 2 and reps["vllm-llama-decode-deployment-vllm"] == 6
    plan = local_plan(render.load_values(ROOT / "deploy/examples/values-16-disaggregated-prefill.yaml"), gpus=8)
    engines = [p for p in plan if p.role == "engine"]
    assert len(engines) == 8
    router = next(p for p in plan if p.role == "router")
    assert "disaggregated_prefill" in router.argv


def test_cache_server_and_remote_url():
    objs = _objs(ROOT / "deploy/examples/values-06-shared-remote-cache.yaml", release="rel")
    c = _engine_container(objs, "rel-llama3-deployment")
    env = {e["name"]: e.get("value") for e in c["env"]}
    assert env["LMCACHE_REMOTE_URL"] == "lm://rel-cache-server-service:81"
    assert env["LMCACHE_REMOTE_SERDE"] == "cachegen"
    cs = next(o for o in objs if o["metadata"]["name"] == "rel-deployment-cache-server")
    cmd = cs["spec"]["template"]["spec"]["containers"][0]["command"]
    assert cmd[:3] == ["vpsa-cache-server", "0.0.0.0", "8080"] and "--max-bytes" in cmd


def test_multinode_layout():
    objs = _objs(ROOT / "deploy/examples/values-15-pipeline-parallel.yaml")
    sts = next(o for o in objs if o["kind"] == "StatefulSet")
    assert sts["spec"]["replicas"] == 2  # pp2 x tp8 over 8-GPU nodes
    cmd = sts["spec"]["template"]["spec"]["containers"][0]["command"]
    assert cmd[cmd.index("--pipeline-parallel-size") + 1] == "2"
    assert cmd[cmd.index("--tensor-parallel-size") + 1] == "8"
    assert any(o["kind"] == "Service" and o["spec"].get("clusterIP") == "None" for o in objs)


def test_secrets_are_collected_and_referenced():
    objs = _objs(ROOT / "deploy/examples/values-11-secure.yaml")
    sec = next(o for o in objs if o["kind"] == "Secret")
    assert sec["stringData"]["vllmApiKey"] == "change-me" and "hf_token_llama3" in sec["stringData"]
    c = _engine_container(objs, "vllm-llama3-deployment")
    key = next(e for e in c["env"] if e["name"] == "VLLM_API_KEY")
    assert key["valueFrom"]["secretKeyRef"] == {"name": "vllm-secrets", "key": "vllmApiKey"}


def test_missing_required_value_is_reported():
    vals = render.load_values(overrides={"servingEngineSpec": {"modelSpec": [{"name": "m", "modelURL": "x"}]}})
    with pytest.raises(render.ValuesError, match="requestMemory"):
        render.render_all(vals)


# --------------------------------------------------------------------------- chart structure
_OPEN = re.compile(r"{{-?\s*(if|range|with|define)\b")
_END = re.compile(r"{{-?\s*end\s*-?}}")


@pytest.mark.parametrize("path", sorted((ROOT / "deploy/helm/templates").glob("*")), ids=lambda p: p.name)
def test_chart_template_blocks_balance(path):
    text = path.read_text()
    # strip template comments (they may mention keywords)
    text = re.sub(r"{{-?\s*/\*.*?\*/\s*-?}}", "", text, flags=re.S)
    assert len(_OPEN.findall(text)) == len(_END.findall(text)), path.name
    # every action closes on its own line (text after it may contain literal braces, e.g. JSON)
    assert "{{" not in re.sub(r"{{.*?}}", "", text)


def test_chart_covers_what_the_renderer_emits():
    """Every object kind the renderer can emit has a chart template producing that kind."""
    chart = "\n".join(p.read_text() for p in (ROOT / "deploy/helm/templates").glob("*.yaml"))
    kinds = set()
    for path in EXAMPLES:
        kinds |= {o["kind"] for o in _objs(path)}
    vals = render.load_values(overrides={
        "servingEngineSpec": {"configs": {"A": "b"}, "maxUnavailablePodDisruptionBudget": "1",
                              "modelSpec": [{"name": "m", "modelURL": "x", "requestCPU": 1, "requestMemory": "1Gi"}]},
        "routerSpec": {"autoscaling": {"enabled": True}, "ingress": {"enabled": True}},
        "sharedPvcStorage": {"hostPath": "/x"}})
    kinds |= {o["kind"] for o in render.render_all(vals)}
    for k in kinds:
        assert re.search(rf"^kind: {k}\b", chart, flags=re.M) or k == "HTTPRoute", k
    assert "HTTPRoute" in chart


def test_crds_and_samples_agree():
    crd_dir = ROOT / "deploy/operator/crds"
    kinds = {}
    for p in crd_dir.glob("*.yaml"):
        crd = yaml.safe_load(p.read_text())
        v = crd["spec"]["versions"][0]
        kinds[crd["spec"]["names"]["kind"]] = v["schema"]["openAPIV3Schema"]["properties"]["spec"]["properties"]
    assert set(kinds) == {"VLLMRuntime", "VLLMRouter", "CacheServer", "LoraAdapter"}
    for p in (ROOT / "deploy/operator/samples").glob("*.yaml"):
        cr = yaml.safe_load(p.read_text())
        assert cr["apiVersion"] == "production-stack.vllm.ai/v1alpha1"
        unknown = set(cr["spec"]) - set(kinds[cr["kind"]])
        assert not unknown, (p.name, unknown)
    router_enum = kinds["VLLMRouter"]["routingLogic"]["enum"]
    assert set(router_enum) == {"roundrobin", "session", "prefixaware", "kvaware", "disaggregated_prefill"}


def test_dashboards_chart_exported_series():
    src = (ROOT / "vllm_production_stack_amd/engine/metrics.py").read_text()
    src += (ROOT / "vllm_production_stack_amd/router/metrics.py").read_text()
    dash = json.loads((ROOT / "deploy/observability/serving-dashboard.json").read_text())
    cm = yaml.safe_load((ROOT / "deploy/observability/kv-tier-dashboard-cm.yaml").read_text())
    dash2 = json.loads(cm["data"]["kv-tier-dashboard.json"])
    names = set()
    for d in (dash, dash2):
        for p in d["panels"]:
            for t in p.get("targets", []):
                names |= set(re.findall(r"(?:vllm|lmcache):[a-z_]+|router_[a-z_]+", t["expr"]))
    assert len(names) > 20
    for n in names:
        base = re.sub(r"_(bucket|total)$", "", n)
        assert base in src, f"dashboard charts {n}, which nothing exports"


def test_all_plain_yaml_parses():
    for p in list((ROOT / "deploy").rglob("*.yaml")):
        if "helm/templates" in str(p):
            continue
        list(yaml.safe_load_all(p.read_text()))


def _validate(schema, value, path="values"):
    """Tiny JSON-schema subset checker (type / enum / required / properties / items / oneOf)."""
    if "oneOf" in schema:
        errs = []
        for s in schema["oneOf"]:
            try:
                _validate(s, value, path)
                return
            except AssertionError as e:
                errs.append(str(e))
        raise AssertionError(f"{path}: no oneOf branch matches ({errs})")
    if "enum" in schema:
        assert value in schema["enum"], f"{path}: {value!r} not in {schema['enum']}"
    t = schema.get("type")
    if t:
        names = {"string": str, "integer": int, "number": (int, float), "boolean": bool, "object": dict, "array": list}
        ts = t if isinstance(t, list) else [t]
        ok = any(isinstance(value, names[x]) and not (x in ("integer", "number") and isinstance(value, bool)) for x in ts)
        assert ok, f"{path}: {type(value).__name__} is not {t}"
    if isinstance(value, dict):
        for r in schema.get("required", []):
            assert r in value, f"{path}: missing {r}"
        props = schema.get("properties")
        if props is not None and not schema.get("additionalProperties"):
            for k, v in value.items():
                assert k in props, f"{path}: unknown key {k}"
                _validate(props[k], v, f"{path}.{k}")
    if isinstance(value, list) and "items" in schema:
        for i, v in enumerate(value):
            _validate(schema["items"], v, f"{path}[{i}]")


@pytest.mark.parametrize("path", EXAMPLES + [ROOT / "deploy/helm/values.yaml"], ids=lambda p: p.stem)
def test_values_match_schema(path):
    schema = json.loads((ROOT / "deploy/helm/values.schema.json").read_text())
    _validate(schema, yaml.safe_load(path.read_text()) or {})


def test_operator_kustomize_tree_is_consistent():
    """deploy/operator/config: every kustomization resolves, the manager's flags exist in the
    operator binary's CLI, and its ClusterRole covers what the reconcilers create."""
    import re

    cfg = ROOT / "deploy" / "operator" / "config"
    kustomizations = list(cfg.rglob("kustomization.yaml"))
    assert len(kustomizations) >= 6
    for k in kustomizations:
        doc = yaml.safe_load(k.read_text())
        assert doc["kind"] == "Kustomization"
        for res in doc["resources"]:
            target = (k.parent / res).resolve()
            assert target.exists(), f"{k}: {res} does not exist"
            if target.is_dir():
                assert (target / "kustomization.yaml").exists()
    dep = yaml.safe_load((cfg / "manager" / "deployment.yaml").read_text())
    manager = dep["spec"]["template"]["spec"]["containers"][0]
    src = (ROOT / "csrc" / "operator" / "operator_main.cpp").read_text()
    for flag in [a for a in manager["args"] if a.startswith("--")]:
        assert f'"{flag}"' in src, f"operator has no flag {flag}"
    assert manager["args"][0] == "run" and manager["livenessProbe"]["httpGet"]["path"] == "/healthz"
    assert dep["spec"]["template"]["spec"]["serviceAccountName"] == "controller-manager"
    # the manager is alone in its pod and finds the API server itself (no proxy sidecar, no --api)
    assert len(dep["spec"]["template"]["spec"]["containers"]) == 1 and "--api" not in manager["args"]
    # the secured metrics port: port, Service, the roles for the reviews and for scrapers, the monitor
    assert ":8443" in manager["args"] and {"name": "https", "containerPort": 8443} in manager["ports"]
    svc = yaml.safe_load((cfg / "default" / "metrics_service.yaml").read_text())
    assert svc["spec"]["ports"][0]["targetPort"] == 8443
    auth = yaml.safe_load((cfg / "rbac" / "metrics_auth_role.yaml").read_text())
    assert {r["resources"][0] for r in auth["rules"]} == {"tokenreviews", "subjectaccessreviews"}
    reader = yaml.safe_load((cfg / "rbac" / "metrics_reader_role.yaml").read_text())
    assert reader["rules"] == [{"nonResourceURLs": ["/metrics"], "verbs": ["get"]}]
    mon = yaml.safe_load((cfg / "prometheus" / "monitor.yaml").read_text())
    assert mon["spec"]["endpoints"][0]["scheme"] == "https" and mon["spec"]["endpoints"][0]["port"] == "https"
    for patch in yaml.safe_load((cfg / "default" / "cert_metrics_manager_patch.yaml").read_text()):
        if patch["path"].endswith("/args/-"):
            assert f'"{patch["value"].split("=")[0]}"' in src
    role = yaml.safe_load((cfg / "rbac" / "role.yaml").read_text())
    granted = {(g, r) for rule in role["rules"] for g in rule["apiGroups"] for r in rule["resources"]}
    for need in [("apps", "deployments"), ("", "services"), ("", "persistentvolumeclaims"),
                 ("", "serviceaccounts"), ("rbac.authorization.k8s.io", "roles"),
                 ("rbac.authorization.k8s.io", "rolebindings"), ("", "pods"),
                 ("production-stack.vllm.ai", "vllmruntimes"), ("production-stack.vllm.ai", "loraadapters/status")]:
        assert need in granted, need
    # the generator reproduces the checked-in tree
    import subprocess, sys, filecmp, shutil, tempfile
    before = tempfile.mkdtemp()
    try:
        shutil.copytree(cfg, before, dirs_exist_ok=True)
        subprocess.run([sys.executable, str(ROOT / "tools" / "gen_operator_config.py")], check=True)
        cmp = filecmp.dircmp(before, cfg)
        assert not cmp.diff_files and not cmp.left_only and not cmp.right_only
    finally:
        shutil.rmtree(before)


def test_chart_readme_documents_every_schema_key():
    import json

    schema = json.loads((ROOT / "deploy" / "helm" / "values.schema.json").read_text())
    readme = (ROOT / "deploy" / "helm" / "README.md").read_text()
    missing = []
    for section in ("servingEngineSpec", "routerSpec", "cacheserverSpec"):
        for key in schema["properties"][section]["properties"]:
            if key not in readme:
                missing.append(f"{section}.{key}")
    for key, sub in schema["properties"]["servingEngineSpec"]["properties"]["modelSpec"]["items"]["properties"].items():
        if key not in readme:
            missing.append(f"modelSpec.{key}")
        for k2 in (sub.get("properties") or {}) if key in ("vllmConfig", "lmcacheConfig") else ():
            if k2 not in readme and k2[0].upper() + k2[1:] not in readme:
                missing.append(f"modelSpec.{key}.{k2}")
    assert not missing, missing


def test_tutorials_reference_existing_files():
    """Every repository path a tutorial names exists (values files may be given by bare name)."""
    import re

    tut = ROOT / "docs" / "tutorials"
    pages = sorted(tut.glob("*.md"))
    assert len(pages) >= 20
    pat = re.compile(r"`((?:deploy|utils|benchmarks|tests|examples|scripts|tools|csrc|profiles|docs)/[A-Za-z0-9_./-]+|"
                     r"values-[A-Za-z0-9_.-]+\.yaml)`")
    missing = []
    for page in pages:
        for m in pat.finditer(page.read_text()):
            ref = m.group(1)
            if not any((ROOT / c).exists() for c in (ref, f"deploy/examples/{ref}")):
                missing.append(f"{page.name}: {ref}")
    assert not missing, missing
    index = (tut / "README.md").read_text()
    for page in pages:
        assert page.name == "README.md" or f"({page.name})" in index, f"{page.name} is not linked from the index"


CLOUD = ROOT / "deploy" / "cloud"


@pytest.mark.parametrize("path", sorted(CLOUD.glob("**/production_stack_specification.yaml")),
                         ids=lambda p: "-".join(p.relative_to(CLOUD).parts[:-1]))
def test_cloud_values_render_with_the_chart(path):
    """The values the cloud bring-up scripts install: eight GPU replicas on the labelled pool, the
    storage class of that cloud, a multi-worker prefix-aware router."""
    objs = _objs(path)
    dep = next(o for o in objs if o["kind"] == "Deployment" and o["metadata"]["name"] == "vllm-llama3-deployment-vllm")
    assert dep["spec"]["replicas"] == 8
    c = dep["spec"]["template"]["spec"]["containers"][0]
    assert c["resources"]["limits"]["amd.com/gpu"] in (1, "1")
    pvc = next(o for o in objs if o["kind"] == "PersistentVolumeClaim")
    assert pvc["spec"]["storageClassName"] in ("efs-sc", "managed-csi-premium", "premium-rwo")
    router = next(o for o in objs if o["kind"] == "Deployment" and o["metadata"]["name"] == "vllm-deployment-router")
    rc = router["spec"]["template"]["spec"]["containers"][0]
    args = rc.get("args") or rc.get("command")
    assert "--router-workers" in args and "prefixaware" in args


def test_cloud_scripts_parse_and_share_the_common_tail():
    import subprocess

    scripts = sorted(CLOUD.glob("**/*.sh"))
    assert {p.parent.name for p in scripts} >= {"aws", "azure", "gcp", "common"}
    for p in scripts:
        r = subprocess.run(["bash", "-n", str(p)], capture_output=True, text=True)
        assert r.returncode == 0, (p, r.stderr)
    for cloud in ("aws", "azure", "gcp"):
        text = (CLOUD / cloud / "entry_point.sh").read_text()
        for step in ("need ", "wait_nodes_ready", "install_stack", "smoke_test", "clean_up.sh"):
            assert step in text, (cloud, step)
        # each cloud's clean-up walks its dependency chain: release and volumes before the cluster
        clean = (CLOUD / cloud / "clean_up.sh").read_text()
        assert clean.index("uninstall_stack") < clean.index({"aws": "eksctl delete cluster", "azure": "az aks delete",
                                                               "gcp": "gcloud container clusters delete"}[cloud]), cloud
    # the device plugin manifest the scripts apply: both DaemonSets tolerate the GPU pools' taint
    docs = [d for d in yaml.safe_load_all((ROOT / "deploy" / "manifests" / "amd-device-plugin-ds.yaml").read_text()) if d]
    daemonsets = [d for d in docs if d["kind"] == "DaemonSet"]
    assert len(daemonsets) == 2
    for ds in daemonsets:
        assert {"operator": "Exists"} in ds["spec"]["template"]["spec"]["tolerations"]
    assert "amd-device-plugin-ds.yaml" in (ROOT / "utils" / "init-amd-gpu-setup-k8s.sh").read_text()
    # every Terraform root module declares its providers and formats as HCL blocks that balance
    for tf in CLOUD.glob("terraform/**/*.tf"):
        body = tf.read_text()
        assert body.count("{") == body.count("}"), tf


def _workload_literals(objs):
    """Every command-line flag and every environment variable name that the renderer gives to any
    container of any workload."""
    flags, envs = set(), set()
    for o in objs:
        if o["kind"] not in ("Deployment", "StatefulSet"):
            continue
        pod = o["spec"]["template"]["spec"]
        for c in pod.get("containers", []) + pod.get("initContainers", []):
            for word in list(c.get("command") or []) + list(c.get("args") or []):
                flags |= set(re.findall(r"(?<![\w-])--[a-z][a-z0-9-]*", str(word)))
            for e in c.get("env") or []:
                envs.add(e["name"])
    return flags, envs


_ACTION_ONLY = re.compile(r"^{{-?\s*(if|else|end|range|with|define|include|toYaml|template|\$[\w]+\s*:?=)\b.*}}$")
_FLAG = re.compile(r"(?<![\w-])--[a-z][a-z0-9_-]*")


def _template_sites(text: str) -> set[tuple]:
    """Where each literal of a chart template sits: (kind, containers|initContainers, container
    name, command|args|env, literal) for every command-line flag and environment variable name.

    The Go templates cannot be rendered here, but their non-action lines are YAML whose
    indentation is the structure: actions that stand alone on a line are dropped, inline actions
    become a placeholder, and the path of a line is the chain of less-indented keys above it.  A
    container name that is itself templated reads as ``*``."""
    text = re.sub(r"{{-?\s*/\*.*?\*/\s*-?}}", "", text, flags=re.S)
    sites: set[tuple] = set()
    for doc in re.split(r"(?m)^---\s*$", text):
        kind = None
        stack: list[tuple[int, str]] = []  # (indent, key)
        container: dict[int, str] = {}  # indent of a "- " item under (init)containers -> its name
        for raw in doc.splitlines():
            line = raw.rstrip()
            body = line.strip()
            if not body or _ACTION_ONLY.match(body) or body.startswith("#"):
                continue
            indent = len(line) - len(line.lstrip())
            item = body.startswith("- ")
            content = body[2:].lstrip() if item else body
            key_indent = indent + 2 if item else indent
            while stack and stack[-1][0] >= (indent if item else key_indent):
                stack.pop()
            for ind in [i for i in container if i > indent or (i == indent and item)]:
                del container[ind]
            m = re.match(r"([A-Za-z_][\w.-]*):(?:\s+(.*))?$", content)
            keys = [k for _, k in stack]
            if indent == 0 and m and m.group(1) == "kind":
                kind = "*" if "{{" in (m.group(2) or "") else (m.group(2) or "").strip()
            section = next((k for k in reversed(keys) if k in ("containers", "initContainers")), None)
            if section and item and m and m.group(1) == "name" and keys and keys[-1] == section:
                container[indent] = "*" if "{{" in (m.group(2) or "") else (m.group(2) or "").strip().strip('"')
            cname = container[max(container)] if container else None
            field = None
            if section and cname is not None:
                after = keys[len(keys) - 1 - keys[::-1].index(section) + 1:]
                field = (m.group(1) if m and not after else after[0] if after else None)
            if field in ("command", "args"):
                for flag in _FLAG.findall(content):  # string literals inside an inline action count too
                    sites.add((kind, section, cname, "command/args", flag))
            if field == "env" and item and m and m.group(1) == "name" and "{{" not in (m.group(2) or ""):
                sites.add((kind, section, cname, "env", (m.group(2) or "").strip().strip('"')))
            if m:
                stack.append((key_indent, m.group(1)))
    return sites


def _rendered_sites(objs) -> set[tuple]:
    out = set()
    for o in objs:
        if o["kind"] not in ("Deployment", "StatefulSet"):
            continue
        pod = o["spec"]["template"]["spec"]
        for section in ("containers", "initContainers"):
            for c in pod.get(section) or []:
                for word in list(c.get("command") or []) + list(c.get("args") or []):
                    for flag in _FLAG.findall(str(word)):
                        out.add((o["kind"], section, c["name"], "command/args", flag))
                for e in c.get("env") or []:
                    out.add((o["kind"], section, c["name"], "env", e["name"]))
    return out


def test_template_site_parser_on_a_known_snippet():
    sites = _template_sites("""
apiVersion: apps/v1
kind: Deployment
spec:
  template:
    spec:
      initContainers:
        - name: {{ default "init" .name }}
          command: ["sh", "-c", "download --from x"]
      containers:
        - name: vllm
          {{- if .Values.x }}
          command:
            - vpsa-engine
            - "--port"
            {{- with .y }}
            - "--seed"
            - {{ . | quote }}
            {{- end }}
          {{- end }}
          env:
            - name: HF_HOME
              value: /data
            - name: {{ .custom }}
              value: "--not-a-flag-site"
          ports:
            - name: container-port
        - name: sidecar
          env:
            - name: PORT
              value: "1"
      volumes:
        - name: shm
---
kind: Service
spec:
  ports:
    - name: http
""")
    assert sites == {
        ("Deployment", "initContainers", "*", "command/args", "--from"),
        ("Deployment", "containers", "vllm", "command/args", "--port"),
        ("Deployment", "containers", "vllm", "command/args", "--seed"),
        ("Deployment", "containers", "vllm", "env", "HF_HOME"),
        ("Deployment", "containers", "sidecar", "env", "PORT"),
    }


@pytest.mark.parametrize("path", EXAMPLES, ids=lambda p: p.stem)
def test_chart_puts_every_flag_and_env_where_the_renderer_puts_it(path):
    """The chart and the Python renderer implement one contract twice, and no `helm` binary exists
    here to render the Go templates.  So the templates are read structurally: every flag and every
    environment variable the renderer emits for an example must be spelled out in a chart
    template under the same workload kind, the same container and the same field (a variable
    under the sidecar instead of the engine, or a flag in the init container, fails), unless it
    comes straight from the example's own values (extra args, user env)."""
    chart: set[tuple] = set()
    for tpl in (ROOT / "deploy/helm/templates").glob("*.yaml"):
        chart |= _template_sites(tpl.read_text())
    own = path.read_text()
    missing = []
    for kind, section, cname, field, literal in sorted(_rendered_sites(_objs(path))):
        if literal in own:
            continue
        if not any((kind, section, n, field, literal) in chart for n in (cname, "*")):
            elsewhere = sorted({(k, s_, n, f) for k, s_, n, f, lit in chart if lit == literal})
            missing.append(f"{literal} under {kind}/{section}[{cname}]/{field}" +
                           (f" (the chart has it under {elsewhere})" if elsewhere else " (nowhere in the chart)"))
    assert not missing, f"{path.name}: " + "; ".join(missing)


@pytest.mark.parametrize("path", EXAMPLES, ids=lambda p: p.stem)
def test_chart_names_every_flag_and_env_the_renderer_emits(path):
    """The chart and the Python renderer implement one contract twice.  No `helm` binary exists
    here to render the Go templates, so drift is caught literally: whatever flag or environment
    variable the renderer emits for an example must be spelled out in the chart's templates,
    unless it comes straight from the example's own values (extra args, user env)."""
    chart = "\n".join(p.read_text() for p in (ROOT / "deploy/helm/templates").glob("*"))
    own = path.read_text()
    flags, envs = _workload_literals(_objs(path))
    missing_flags = sorted(f for f in flags if f not in chart and f not in own)
    missing_envs = sorted(e for e in envs if e not in chart and e not in own)
    assert not missing_flags, f"{path.name}: flags the chart never emits: {missing_flags}"
    assert not missing_envs, f"{path.name}: env names the chart never emits: {missing_envs}"
