#!/usr/bin/env bash
# AKS with one AMD Instinct GPU node pool.   entry_point.sh setup <values.yaml> | cleanup
set -euo pipefail
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/../../.." && pwd)"
RG=${RESOURCE_GROUP:-vpsa-rg}; CLUSTER=${CLUSTER_NAME:-vpsa-aks}; LOCATION=${LOCATION:-westus3}
GPU_VM_SIZE=${GPU_VM_SIZE:-Standard_ND96isr_MI300X_v5}; GPU_NODES=${GPU_NODES:-1}
case "${1:-}" in
  setup)
    values=${2:?values file}
    az group create --name "$RG" --location "$LOCATION"
    az aks create --resource-group "$RG" --name "$CLUSTER" --node-count 1 --node-vm-size Standard_D8s_v5 \
      --generate-ssh-keys --enable-managed-identity
    az aks nodepool add --resource-group "$RG" --cluster-name "$CLUSTER" --name gpupool \
      --node-count "$GPU_NODES" --node-vm-size "$GPU_VM_SIZE" --node-taints sku=gpu:NoSchedule \
      --labels accelerator=amd-instinct --skip-gpu-driver-install
    az aks get-credentials --resource-group "$RG" --name "$CLUSTER" --overwrite-existing
    "$root/utils/init-amd-gpu-setup-k8s.sh"
    helm upgrade --install vllm "$root/deploy/helm" -f "$values" \
      --set 'servingEngineSpec.tolerations[0].key=sku' --set 'servingEngineSpec.tolerations[0].value=gpu' \
      --set 'servingEngineSpec.tolerations[0].effect=NoSchedule'
    ;;
  cleanup)
    helm uninstall vllm || true
    az group delete --name "$RG" --yes --no-wait
    ;;
  *) echo "usage: $0 setup <values.yaml> | cleanup"; exit 2 ;;
esac
