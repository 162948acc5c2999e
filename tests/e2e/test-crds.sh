#!/usr/bin/env bash
# Applies the four CRDs and the sample resources, then reads each one back.
set -euo pipefail
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)"
kubectl apply -f "$root/deploy/operator/crds/"
kubectl wait --for=condition=Established crd --all --timeout=60s
kubectl apply -f "$root/deploy/operator/samples/"
for k in vllmruntime vllmrouter cacheserver loraadapter; do kubectl get "$k" -o name | grep -q sample; done
kubectl delete -f "$root/deploy/operator/samples/"
echo "CRDs and samples OK"
