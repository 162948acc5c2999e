#!/usr/bin/env bash
# GKE with one GPU node pool and local SSDs for the disk KV tier.   entry_point.sh setup <values.yaml> | cleanup
set -euo pipefail
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/../../.." && pwd)"
CLUSTER=${CLUSTER_NAME:-vpsa-gke}; ZONE=${ZONE:-us-central1-a}
GPU_MACHINE_TYPE=${GPU_MACHINE_TYPE:?set GPU_MACHINE_TYPE to an AMD Instinct machine type}; GPU_NODES=${GPU_NODES:-1}
case "${1:-}" in
  setup)
    values=${2:?values file}
    gcloud container clusters create "$CLUSTER" --zone "$ZONE" --num-nodes 1 --machine-type e2-standard-8
    gcloud container node-pools create gpu --cluster "$CLUSTER" --zone "$ZONE" --machine-type "$GPU_MACHINE_TYPE" \
      --num-nodes "$GPU_NODES" --ephemeral-storage-local-ssd count=2 --node-labels accelerator=amd-instinct
    gcloud container clusters get-credentials "$CLUSTER" --zone "$ZONE"
    "$root/utils/init-amd-gpu-setup-k8s.sh"
    helm upgrade --install vllm "$root/deploy/helm" -f "$values"
    ;;
  cleanup)
    helm uninstall vllm || true
    gcloud container clusters delete "$CLUSTER" --zone "$ZONE" --quiet
    ;;
  *) echo "usage: $0 setup <values.yaml> | cleanup"; exit 2 ;;
esac
