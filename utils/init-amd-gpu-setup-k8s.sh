#!/usr/bin/env bash
# AMD GPU device plugin + node labeller: makes "amd.com/gpu" a schedulable resource.
# Applies the pinned manifest of this tree (deploy/manifests/amd-device-plugin-ds.yaml, tolerates
# tainted GPU pools); UPSTREAM=1 applies the ROCm project's current manifests instead.
set -euo pipefail
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
if [ "${UPSTREAM:-0}" = "1" ]; then
  kubectl apply -f https://raw.githubusercontent.com/ROCm/k8s-device-plugin/master/k8s-ds-amdgpu-dp.yaml
  kubectl apply -f https://raw.githubusercontent.com/ROCm/k8s-device-plugin/master/k8s-ds-amdgpu-labeller.yaml
else
  kubectl apply -f "$root/deploy/manifests/amd-device-plugin-ds.yaml"
fi
echo "waiting for amd.com/gpu to appear on a node ..."
for _ in $(seq 60); do
  n=$(kubectl get nodes -o jsonpath='{.items[*].status.allocatable.amd\.com/gpu}' | tr ' ' '\n' | awk '{s+=$1} END {print s+0}')
  [ "$n" -gt 0 ] && { echo "allocatable amd.com/gpu: $n"; exit 0; }
  sleep 5
done
echo "no amd.com/gpu resource after 5 minutes" >&2
kubectl -n kube-system get pods -l name=amdgpu-dp-ds -o wide >&2 || true
exit 1
