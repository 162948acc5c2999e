#!/usr/bin/env bash
# Shared steps of the cloud bring-up scripts: everything after "kubectl points at a cluster that
# has a GPU node pool".  Sourced by aws/, azure/ and gcp/ entry points.

VPSA_ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/../../.." && pwd)"
RELEASE=${RELEASE:-vllm}
NAMESPACE=${NAMESPACE:-default}

need() {  # need <binary>...: fail early with the list of what is missing
  local missing=()
  for b in "$@"; do command -v "$b" >/dev/null 2>&1 || missing+=("$b"); done
  if [ "${#missing[@]}" -gt 0 ]; then
    echo "missing on PATH: ${missing[*]}" >&2
    exit 3
  fi
}

wait_nodes_ready() {  # wait_nodes_ready <label selector> <count>
  local sel=$1 want=$2 ready=0
  echo "waiting for $want node(s) with $sel to be Ready ..."
  for _ in $(seq 120); do
    ready=$(kubectl get nodes -l "$sel" --no-headers 2>/dev/null | awk '$2 == "Ready"' | wc -l)
    [ "$ready" -ge "$want" ] && { echo "  $ready ready"; return 0; }
    sleep 10
  done
  echo "only $ready of $want GPU node(s) became Ready in 20 minutes" >&2
  kubectl get nodes -o wide >&2 || true
  return 1
}

install_gpu_plugin() {  # device plugin + labeller, then wait until amd.com/gpu is allocatable
  "$VPSA_ROOT/utils/init-amd-gpu-setup-k8s.sh"
}

install_stack() {  # install_stack <values.yaml> [extra helm args...]
  local values=$1
  shift
  [ -f "$values" ] || { echo "values file $values not found" >&2; exit 2; }
  helm upgrade --install "$RELEASE" "$VPSA_ROOT/deploy/helm" --namespace "$NAMESPACE" --create-namespace \
    -f "$values" --wait --timeout 30m "$@"
  echo
  kubectl -n "$NAMESPACE" get pods -o wide
  echo
  echo "router: kubectl -n $NAMESPACE port-forward svc/$RELEASE-router-service 30080:80"
  echo "then:   curl -s localhost:30080/v1/models"
}

smoke_test() {  # one request through the router, from inside the cluster
  local model
  model=$(kubectl -n "$NAMESPACE" run vpsa-smoke --rm -i --restart=Never --image=curlimages/curl:8.8.0 --quiet -- \
    -s "http://$RELEASE-router-service/v1/models" | sed -n 's/.*"id": *"\([^"]*\)".*/\1/p' | head -1)
  [ -n "$model" ] || { echo "router lists no model" >&2; return 1; }
  echo "router serves: $model"
  kubectl -n "$NAMESPACE" run vpsa-smoke --rm -i --restart=Never --image=curlimages/curl:8.8.0 --quiet -- \
    -s "http://$RELEASE-router-service/v1/completions" -H 'Content-Type: application/json' \
    -d "{\"model\":\"$model\",\"prompt\":\"Hello\",\"max_tokens\":8}"
  echo
}

uninstall_stack() {  # release, its volumes, and anything that would keep cloud resources alive
  helm uninstall "$RELEASE" --namespace "$NAMESPACE" 2>/dev/null || true
  kubectl -n "$NAMESPACE" delete pvc --all --ignore-not-found --wait=false || true
  # LoadBalancer services hold cloud load balancers (and block VPC / resource-group deletion)
  kubectl get svc -A -o jsonpath='{range .items[?(@.spec.type=="LoadBalancer")]}{.metadata.namespace}{" "}{.metadata.name}{"\n"}{end}' \
    2>/dev/null | while read -r ns name; do
      [ -n "$name" ] && kubectl -n "$ns" delete svc "$name" --ignore-not-found || true
    done
}
