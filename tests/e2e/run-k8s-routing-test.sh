#!/usr/bin/env bash
# For each routing logic: helm install the matching example values, wait for the pods, port-forward the
# router, run test_routing.py with logs fetched from the router pod, uninstall.
set -euo pipefail
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)"
declare -A values=([roundrobin]=values-01-two-replicas [session]=values-05-host-dram-offload
                   [prefixaware]=values-18-prefix-aware [kvaware]=values-17-kv-aware
                   [disaggregated_prefill]=values-16-disaggregated-prefill)
for logic in "${!values[@]}"; do
  helm upgrade --install vllm "$root/deploy/helm" -f "$root/deploy/examples/${values[$logic]}.yaml" \
    --set routerSpec.routingLogic="$logic" --wait --timeout 20m
  kubectl port-forward svc/vllm-router-service 30080:80 >/dev/null 2>&1 &
  pf=$!
  sleep 3
  model=$(curl -fs localhost:30080/v1/models | python -c 'import json,sys; print(json.load(sys.stdin)["data"][0]["id"])')
  log=$(mktemp)
  ( kubectl logs -f -l environment=router --tail=-1 >"$log" ) &
  lf=$!
  python "$root/tests/e2e/test_routing.py" --router-url http://localhost:30080 --log-file "$log" \
    --routing-logic "$logic" --model "$model"
  kill $pf $lf 2>/dev/null || true
  helm uninstall vllm --wait
done
