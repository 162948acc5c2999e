"""Write the kustomize tree that installs the native operator (deploy/operator/config/).

Counterpart of the reference's kubebuilder-generated /root/reference/operator/config/**
(manager Deployment, RBAC incl. per-CRD admin/editor/viewer roles, leader election, CRD and
sample kustomizations, metrics Service + auth roles + ServiceMonitor, network policies).  The
manager is the C++ ``vpsa-operator`` binary alone in its pod: it finds the API server from the pod
environment, verifies it against the service account's CA bundle, authenticates with the
service account's token, and serves its metrics over TLS on :8443 to callers the API server
authenticates and authorizes.

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
                         "args": ["run", "--leader-elect", "--leader-elect-namespace", "$(POD_NAMESPACE)",
                                  "--health-port", "8081", "--metrics-bind-address", ":8443",
                                  "--gpu-resource", "amd.com/gpu", "--interval", "10"],
                         "env": [{"name": "POD_NAMESPACE", "valueFrom": {"fieldRef": {"fieldPath": "metadata.namespace"}}}],
                         "ports": [{"name": "health", "containerPort": 8081}, {"name": "https", "containerPort": 8443}],
                         "volumeMounts": [], 
                         "livenessProbe": probe("/healthz"), "readinessProbe": probe("/readyz"),
                         "securityContext": {"allowPrivilegeEscalation": False, "capabilities": {"drop": ["ALL"]}},
                         "resources": {"limits": {"cpu": "500m", "memory": "128Mi"},
                                       "requests": {"cpu": "10m", "memory": "64Mi"}}}],
                    "volumes": []}}}})
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
    # the secured metrics port asks the API server who a caller is and what it may do ...
    write("rbac/metrics_auth_role.yaml", {
        "apiVersion": "rbac.authorization.k8s.io/v1", "kind": "ClusterRole",
        "metadata": {"name": "metrics-auth-role", "labels": LABELS},
        "rules": [{"apiGroups": ["authentication.k8s.io"], "resources": ["tokenreviews"], "verbs": ["create"]},
                  {"apiGroups": ["authorization.k8s.io"], "resources": ["subjectaccessreviews"], "verbs": ["create"]}]})
    write("rbac/metrics_auth_role_binding.yaml", {
        "apiVersion": "rbac.authorization.k8s.io/v1", "kind": "ClusterRoleBinding",
        "metadata": {"name": "metrics-auth-rolebinding", "labels": LABELS},
        "roleRef": {"apiGroup": "rbac.authorization.k8s.io", "kind": "ClusterRole", "name": "metrics-auth-role"},
        "subjects": sa})
    # ... and this is the role to bind to whoever scrapes (e.g. the Prometheus service account)
    write("rbac/metrics_reader_role.yaml", {
        "apiVersion": "rbac.authorization.k8s.io/v1", "kind": "ClusterRole",
        "metadata": {"name": "metrics-reader", "labels": LABELS},
        "rules": [{"nonResourceURLs": ["/metrics"], "verbs": ["get"]}]})
    files = ["service_account.yaml", "role.yaml", "role_binding.yaml", "leader_election_role.yaml",
             "leader_election_role_binding.yaml", "metrics_auth_role.yaml", "metrics_auth_role_binding.yaml",
             "metrics_reader_role.yaml"]
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
    write("network-policy/allow-metrics-traffic.yaml", {
        "apiVersion": "networking.k8s.io/v1", "kind": "NetworkPolicy",
        "metadata": {"name": "allow-metrics-traffic", "namespace": "system", "labels": LABELS},
        "spec": {"podSelector": {"matchLabels": {"control-plane": "controller-manager"}},
                 "policyTypes": ["Ingress"],
                 # scrapers live in namespaces labelled metrics=enabled
                 "ingress": [{"from": [{"namespaceSelector": {"matchLabels": {"metrics": "enabled"}}}],
                              "ports": [{"port": 8443, "protocol": "TCP"}]}]}})
    kustomization("network-policy/kustomization.yaml", ["allow-health-traffic.yaml", "allow-metrics-traffic.yaml"])

    # ---- metrics: Service in front of the TLS port, ServiceMonitor for the Prometheus operator
    write("default/metrics_service.yaml", {
        "apiVersion": "v1", "kind": "Service",
        "metadata": {"name": "controller-manager-metrics-service", "namespace": "system",
                     "labels": {**LABELS, "control-plane": "controller-manager"}},
        "spec": {"ports": [{"name": "https", "port": 8443, "protocol": "TCP", "targetPort": 8443}],
                 "selector": {"control-plane": "controller-manager"}}})
    # with cert-manager: mount the issued secret and point the manager at it (it re-reads the
    # files when the secret is renewed); without this patch the manager signs its own certificate
    write("default/cert_metrics_manager_patch.yaml",
          [{"op": "add", "path": "/spec/template/spec/containers/0/volumeMounts/-",
            "value": {"mountPath": "/tmp/k8s-metrics-server/metrics-certs", "name": "metrics-certs", "readOnly": True}},
           {"op": "add", "path": "/spec/template/spec/containers/0/args/-",
            "value": "--metrics-cert-path=/tmp/k8s-metrics-server/metrics-certs"},
           {"op": "add", "path": "/spec/template/spec/volumes/-",
            "value": {"name": "metrics-certs",
                      "secret": {"secretName": "metrics-server-cert", "optional": False,
                                 "items": [{"key": k, "path": k} for k in ("ca.crt", "tls.crt", "tls.key")]}}}],
          header="JSON patch for the manager Deployment; enable it in kustomization.yaml (patches:) once\n"
                 "cert-manager issues the secret metrics-server-cert.")
    write("prometheus/monitor.yaml", {
        "apiVersion": "monitoring.coreos.com/v1", "kind": "ServiceMonitor",
        "metadata": {"name": "controller-manager-metrics-monitor", "namespace": "system",
                     "labels": {**LABELS, "control-plane": "controller-manager"}},
        "spec": {"endpoints": [{"path": "/metrics", "port": "https", "scheme": "https",
                                "bearerTokenFile": "/var/run/secrets/kubernetes.io/serviceaccount/token",
                                "tlsConfig": {"insecureSkipVerify": True}}],
                 "selector": {"matchLabels": {"control-plane": "controller-manager"}}}})
    write("prometheus/monitor_tls_patch.yaml",
          [{"op": "replace", "path": "/spec/endpoints/0/tlsConfig",
            "value": {"serverName": "SERVICE_NAME.SERVICE_NAMESPACE.svc", "insecureSkipVerify": False,
                      "ca": {"secret": {"name": "metrics-server-cert", "key": "ca.crt"}},
                      "cert": {"secret": {"name": "metrics-server-cert", "key": "tls.crt"}},
                      "keySecret": {"name": "metrics-server-cert", "key": "tls.key"}}}],
          header="JSON patch for the ServiceMonitor when the metrics certificate comes from cert-manager:\n"
                 "verify the server against the issuing CA instead of skipping verification.")
    kustomization("prometheus/kustomization.yaml", ["monitor.yaml"])

    # ---- default overlay
    kustomization("default/kustomization.yaml", ["../crd", "../rbac", "../manager", "metrics_service.yaml"],
                  namespace="production-stack-system", namePrefix="production-stack-",
                  labels=[{"pairs": {"app.kubernetes.io/part-of": "vllm-production-stack-amd"}}])


if __name__ == "__main__":
    main()
