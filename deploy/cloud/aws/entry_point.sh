#!/usr/bin/env bash
# EKS with one GPU node group (eksctl).   entry_point.sh setup <values.yaml> | cleanup
set -euo pipefail
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/../../.." && pwd)"
CLUSTER=${CLUSTER_NAME:-vpsa-eks}; REGION=${AWS_REGION:-us-west-2}
GPU_INSTANCE_TYPE=${GPU_INSTANCE_TYPE:?set GPU_INSTANCE_TYPE to an AMD Instinct instance type}; GPU_NODES=${GPU_NODES:-1}
case "${1:-}" in
  setup)
    values=${2:?values file}
    eksctl create cluster --name "$CLUSTER" --region "$REGION" --nodegroup-name cpu --node-type m6i.2xlarge --nodes 1
    eksctl create nodegroup --cluster "$CLUSTER" --region "$REGION" --name gpu --node-type "$GPU_INSTANCE_TYPE" \
      --nodes "$GPU_NODES" --node-volume-size 500 --node-labels accelerator=amd-instinct
    aws eks update-kubeconfig --name "$CLUSTER" --region "$REGION"
    "$root/utils/init-amd-gpu-setup-k8s.sh"
    helm upgrade --install vllm "$root/deploy/helm" -f "$values"
    ;;
  cleanup)
    helm uninstall vllm || true
    eksctl delete cluster --name "$CLUSTER" --region "$REGION"
    ;;
  *) echo "usage: $0 setup <values.yaml> | cleanup"; exit 2 ;;
esac
