#!/usr/bin/env bash
# AKS cluster with a system pool and an AMD Instinct GPU pool (ND MI300X v5 VMs carry 8 GPUs),
# then the serving chart.
#
#   deploy/cloud/azure/entry_point.sh setup deploy/cloud/azure/production_stack_specification.yaml
#   deploy/cloud/azure/entry_point.sh status | smoke | cleanup
#
# Environment: RESOURCE_GROUP (vpsa-rg), CLUSTER_NAME (vpsa-aks), LOCATION (westus3),
# GPU_VM_SIZE (Standard_ND96isr_MI300X_v5), GPU_NODES (1), GPU_OS_DISK_GB (1024).
# The GPU pool is tainted sku=gpu:NoSchedule; the chart gets the matching toleration.  AKS does not
# install AMD drivers (--skip-gpu-driver-install keeps it from trying the NVIDIA ones): the node
# image must carry the amdgpu driver, or install the AMD GPU operator instead of the plain device
# plugin (WITH_GPU_OPERATOR=1).
# Flow of /root/reference/deployment_on_cloud/azure/entry_point.sh:1-92, AMD device plugin in place
# of its nvidia-device-plugin-ds.yaml.
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
# shellcheck source=../common/lib.sh
. "$here/../common/lib.sh"

RG=${RESOURCE_GROUP:-vpsa-rg}
CLUSTER=${CLUSTER_NAME:-vpsa-aks}
LOCATION=${LOCATION:-westus3}
GPU_VM_SIZE=${GPU_VM_SIZE:-Standard_ND96isr_MI300X_v5}
GPU_NODES=${GPU_NODES:-1}
GPU_OS_DISK_GB=${GPU_OS_DISK_GB:-1024}

case "${1:-}" in
  setup)
    values=${2:?usage: $0 setup <values.yaml>}
    need az kubectl helm
    az account show >/dev/null || { echo "run 'az login' first" >&2; exit 3; }
    quota=$(az vm list-usage --location "$LOCATION" --query "[?contains(name.value, 'MI300X')].{l:limit,c:currentValue}" -o tsv 2>/dev/null || true)
    [ -n "$quota" ] && echo "MI300X vCPU quota in $LOCATION (limit, used): $quota"
    az group create --name "$RG" --location "$LOCATION" -o none
    if ! az aks show --resource-group "$RG" --name "$CLUSTER" -o none 2>/dev/null; then
      az aks create --resource-group "$RG" --name "$CLUSTER" --node-count 1 --node-vm-size Standard_D8s_v5 \
        --generate-ssh-keys --enable-managed-identity --network-plugin azure -o none
    fi
    if ! az aks nodepool show --resource-group "$RG" --cluster-name "$CLUSTER" --name gpupool -o none 2>/dev/null; then
      az aks nodepool add --resource-group "$RG" --cluster-name "$CLUSTER" --name gpupool \
        --node-count "$GPU_NODES" --node-vm-size "$GPU_VM_SIZE" --node-osdisk-size "$GPU_OS_DISK_GB" \
        --node-taints sku=gpu:NoSchedule --labels accelerator=amd-instinct --skip-gpu-driver-install -o none
    fi
    az aks get-credentials --resource-group "$RG" --name "$CLUSTER" --overwrite-existing
    wait_nodes_ready accelerator=amd-instinct "$GPU_NODES"
    if [ "${WITH_GPU_OPERATOR:-0}" = "1" ]; then
      helm repo add rocm https://rocm.github.io/gpu-operator >/dev/null && helm repo update >/dev/null
      helm upgrade --install amd-gpu-operator rocm/gpu-operator-charts --namespace kube-amd-gpu --create-namespace --wait
    else
      install_gpu_plugin
    fi
    install_stack "$values" \
      --set 'servingEngineSpec.tolerations[0].key=sku' --set 'servingEngineSpec.tolerations[0].operator=Equal' \
      --set 'servingEngineSpec.tolerations[0].value=gpu' --set 'servingEngineSpec.tolerations[0].effect=NoSchedule'
    ;;
  status)
    az aks nodepool list --resource-group "$RG" --cluster-name "$CLUSTER" -o table
    kubectl -n "$NAMESPACE" get pods,svc,pvc
    ;;
  smoke) smoke_test ;;
  cleanup)
    exec "$here/clean_up.sh" "$RG" "$CLUSTER"
    ;;
  *) echo "usage: $0 setup <values.yaml> | status | smoke | cleanup" >&2; exit 2 ;;
esac
