"""Write the kustomize tree that installs the native operator (deploy/operator/config/).

Counterpart of the reference's kubebuilder-generated /root/reference/operator/config/**
(manager Deployment, RBAC incl. per-CRD admin/editor/viewer roles, leader election, CRD and
sample kustomizations, network policy).  The manager here is the C++ ``vpsa-operator`` binary
talking to the API server through a loopback ``kubectl proxy`` sidecar.

    python tools/gen_operator_config.py          # rewrites deploy/operator/config
"""
import os

import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "deploy", "operator", "config")
GROUP = "production-stack.vllm.ai"
KINDS = {"vllmruntime": "vllmruntimes", "vllmrouter": "vllmrouters", "cacheserver": "cacheservers",
         "loraadapter": "loraadapters"}
LABELS = {"app.kubernetes.io/name": "vpsa-operator", "app.kubernetes.io/managed-by": "kustomize"}


class _Compact(yaml.SafeDumper):
    """Lists of plain scalars (verbs, apiGroups, resources, args) and small all-scalar maps
    (labels, selectors) on one line; everything else in block style."""


def _seq(dumper, data):
    flat = all(isinstance(x, (str, int, float, bool)) for x in data)
    return dumper.represent_sequence("tag:yaml.org,2002:seq", data, flow_style=flat)


def _map(dumper, data):
    flat = 0 < len(data) <= 3 and all(isinstance(v, (str, int, float, bool)) for v in data.values())
    return dumper.represent_mapping("tag:yaml.org,2002:map", data.items(), flow_style=flat)


_Compact.add_representer(list, _seq)
_Compact.add_representer(dict, _map)


def write(rel, *docs, header=None):
    path = os.path.join(OUT, rel)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        if header:
            f.write("".join(f"# {ln}\n" for ln in header.splitlines()))
        f.write(yaml.dump_all(docs, Dumper=_Compact, sort_keys=False, width=110))


def kustomization(rel, resources, **extra):
    write(rel, {"apiVersion": "kustomize.config.k8s.io/v1beta1", "kind": "Kustomization",
                "resources": resources, **extra})


def main():
    # ---- CRDs and samples
    kustomization("crd/kustomization.yaml", [f"../../crds/{GROUP}_{p}.yaml" for p in KINDS.values()])
    kustomization("samples/kustomization.yaml", [f"../../samples/{k}.yaml" for k in KINDS])

    # ---- manager
    write("manager/namespace.yaml", {"apiVersion": "v1", "kind": "Namespace",
                                     "metadata": {"name": "system", "labels": {**LABELS, "control-plane": "controller-manager"}}})
    probe = lambda path: {"httpGet": {"path": path, "port": 8081}, "initialDelaySeconds": 5, "periodSeconds": 10}
    write("manager/deployment.yaml", {
        "apiVersion": "apps/v1", "kind": "Deployment",
        "metadata": {"name": "controller-manager", "namespace": "system",
                     "labels": {**LABELS, "control-plane": "controller-manager"}},
        "spec": {
            "replicas": 1, "selector": {"matchLabels": {"control-plane": "controller-manager"}},
            "template": {
                "metadata": {"labels": {"control-plane": "controller-manager"},
                             "annotations": {"kubectl.kubernetes.io/default-container": "manager"}},
                "spec": {
                    "serviceAccountName": "controller-manager", "terminationGracePeriodSeconds": 10,
                    "securityContext": {"runAsNonRoot": True, "seccompProfile": {"type": "RuntimeDefault"}},
                    "containers": [
                        {"name": "manager", "image": "controller:latest",
                         "command": ["/usr/local/bin/vpsa-operator"],
                         "args": ["run", "--api", "http://127.0.0.1:8001", "--health-port", "8081",
                                  "--gpu-resource", "amd.com/gpu", "--interval", "10"],
                         "ports": [{"name": "health", "containerPort": 8081}],
                         "livenessProbe": probe("/healthz"), "readinessProbe": probe("/readyz"),
                         "securityContext": {"allowPrivilegeEscalation": False, "capabilities": {"drop": ["ALL"]}},
                         "resources": {"limits": {"cpu": "500m", "memory": "128Mi"},
                                       "requests": {"cpu": "10m", "memory": "64Mi"}}},
                        # the operator speaks plain HTTP to the API server through this loopback proxy,
                        # which adds the pod's service-account credentials
                        {"name": "kube-api-proxy", "image": "bitnami/kubectl:latest",
                         "args": ["proxy", "--port=8001", "--address=127.0.0.1"],
                         "securityContext": {"allowPrivilegeEscalation": False, "capabilities": {"drop": ["ALL"]}},
                         "resources": {"limits": {"cpu": "200m", "memory": "64Mi"},
                                       "requests": {"cpu": "10m", "memory": "32Mi"}}}]}}}})
    kustomization("manager/kustomization.yaml", ["namespace.yaml", "deployment.yaml"],
                  images=[{"name": "controller", "newName": "vpsa/operator", "newTag": "latest"}])

    # ---- RBAC
    crd_res = list(KINDS.values())
    rules = [
        {"apiGroups": [GROUP], "resources": crd_res,
         "verbs": ["create", "delete", "get", "list", "patch", "update", "watch"]},
        {"apiGroups": [GROUP], "resources": [f"{r}/status" for r in crd_res], "verbs": ["get", "patch", "update"]},
        {"apiGroups": [GROUP], "resources": [f"{r}/finalizers" for r in crd_res], "verbs": ["update"]},
        {"apiGroups": [""], "resources": ["pods", "services", "secrets", "configmaps", "persistentvolumeclaims",
                                          "serviceaccounts"],
         "verbs": ["create", "delete", "get", "list", "patch", "update", "watch"]},
        {"apiGroups": ["apps"], "resources": ["deployments"],
         "verbs": ["create", "delete", "get", "list", "patch", "update", "watch"]},
        {"apiGroups": ["rbac.authorization.k8s.io"], "resources": ["roles", "rolebindings"],
         "verbs": ["create", "get", "list", "patch", "update", "watch"]},
        {"apiGroups": [""], "resources": ["events"], "verbs": ["create", "patch"]},
    ]
    write("rbac/service_account.yaml", {"apiVersion": "v1", "kind": "ServiceAccount",
                                        "metadata": {"name": "controller-manager", "namespace": "system", "labels": LABELS}})
    write("rbac/role.yaml", {"apiVersion": "rbac.authorization.k8s.io/v1", "kind": "ClusterRole",
                             "metadata": {"name": "manager-role", "labels": LABELS}, "rules": rules})
    sa = [{"kind": "ServiceAccount", "name": "controller-manager", "namespace": "system"}]
    write("rbac/role_binding.yaml", {"apiVersion": "rbac.authorization.k8s.io/v1", "kind": "ClusterRoleBinding",
                                     "metadata": {"name": "manager-rolebinding", "labels": LABELS},
                                     "roleRef": {"apiGroup": "rbac.authorization.k8s.io", "kind": "ClusterRole",
                                                 "name": "manager-role"}, "subjects": sa})
    write("rbac/leader_election_role.yaml", {
        "apiVersion": "rbac.authorization.k8s.io/v1", "kind": "Role",
        "metadata": {"name": "leader-election-role", "labels": LABELS},
        "rules": [{"apiGroups": ["coordination.k8s.io"], "resources": ["leases"],
                   "verbs": ["get", "list", "watch", "create", "update", "patch", "delete"]},
                  {"apiGroups": [""], "resources": ["events"], "verbs": ["create", "patch"]}]})
    write("rbac/leader_election_role_binding.yaml", {
        "apiVersion": "rbac.authorization.k8s.io/v1", "kind": "RoleBinding",
        "metadata": {"name": "leader-election-rolebinding", "labels": LABELS},
        "roleRef": {"apiGroup": "rbac.authorization.k8s.io", "kind": "Role", "name": "leader-election-role"},
        "subjects": sa})
    files = ["service_account.yaml", "role.yaml", "role_binding.yaml", "leader_election_role.yaml",
             "leader_election_role_binding.yaml"]
    # helper roles for cluster users (not used by the operator itself)
    verbs = {"admin": ["*"], "editor": ["create", "delete", "get", "list", "patch", "update", "watch"],
             "viewer": ["get", "list", "watch"]}
    for kind, plural in KINDS.items():
        for role, vs in verbs.items():
            r = [{"apiGroups": [GROUP], "resources": [plural], "verbs": vs},
                 {"apiGroups": [GROUP], "resources": [f"{plural}/status"], "verbs": ["get"]}]
            name = f"{kind}_{role}_role.yaml"
            write("rbac/" + name, {"apiVersion": "rbac.authorization.k8s.io/v1", "kind": "ClusterRole",
                                   "metadata": {"name": f"{kind}-{role}-role", "labels": LABELS}, "rules": r})
            files.append(name)
    kustomization("rbac/kustomization.yaml", files)

    # ---- network policy: only the kubelet probes need to reach the manager
    write("network-policy/allow-health-traffic.yaml", {
        "apiVersion": "networking.k8s.io/v1", "kind": "NetworkPolicy",
        "metadata": {"name": "allow-health-traffic", "namespace": "system", "labels": LABELS},
        "spec": {"podSelector": {"matchLabels": {"control-plane": "controller-manager"}},
                 "policyTypes": ["Ingress"], "ingress": [{"ports": [{"port": 8081, "protocol": "TCP"}]}]}})
    kustomization("network-policy/kustomization.yaml", ["allow-health-traffic.yaml"])

    # ---- default overlay
    kustomization("default/kustomization.yaml", ["../crd", "../rbac", "../manager"],
                  namespace="production-stack-system", namePrefix="production-stack-",
                  labels=[{"pairs": {"app.kubernetes.io/part-of": "vllm-production-stack-amd"}}])


if __name__ == "__main__":
    main()
