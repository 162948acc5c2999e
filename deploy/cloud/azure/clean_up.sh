#!/usr/bin/env bash
# Remove everything deploy/cloud/azure/entry_point.sh created, in the order Azure allows, and
# report what is left.      clean_up.sh [resource group] [cluster]
#
#   release + volumes   the chart, its PVCs (Azure Disk / Azure Files shares live in the node
#                       resource group and are deleted by the CSI driver only while the cluster
#                       is alive: delete the claims first, then wait for the volumes to go)
#   load balancers      Services of type LoadBalancer hold public IPs in the node resource group
#   GPU pool            scaled away first: ND MI300X v5 capacity is billed per minute, and a
#                       resource-group deletion can sit behind other resources for a long time
#   cluster, group      the AKS resource, then the resource group (which takes the MC_* node
#                       resource group with it)
# Environment: RESOURCE_GROUP (vpsa-rg), CLUSTER_NAME (vpsa-aks), KEEP_GROUP=1 keeps the group
# (for groups you did not create for this), WAIT=1 blocks until the group is gone.
# The reference has no Azure clean-up script; this follows the dependency walk of its AWS one
# (/root/reference/deployment_on_cloud/aws/clean_up.sh:1-117).
set -uo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
# shellcheck source=../common/lib.sh
. "$here/../common/lib.sh"
RG=${1:-${RESOURCE_GROUP:-vpsa-rg}}
CLUSTER=${2:-${CLUSTER_NAME:-vpsa-aks}}
need az

if ! az group show --name "$RG" -o none 2>/dev/null; then
  echo "resource group $RG does not exist: nothing to do"
  exit 0
fi

if az aks show --resource-group "$RG" --name "$CLUSTER" -o none 2>/dev/null; then
  node_rg=$(az aks show --resource-group "$RG" --name "$CLUSTER" --query nodeResourceGroup -o tsv)
  if command -v kubectl >/dev/null && az aks get-credentials --resource-group "$RG" --name "$CLUSTER" --overwrite-existing 2>/dev/null; then
    uninstall_stack
    kubectl delete -f "$VPSA_ROOT/deploy/manifests/amd-device-plugin-ds.yaml" --ignore-not-found >/dev/null 2>&1 || true
    helm uninstall amd-gpu-operator --namespace kube-amd-gpu 2>/dev/null || true
    echo "waiting for dynamically provisioned volumes to be released ..."
    for _ in $(seq 30); do
      left=$(kubectl get pv --no-headers 2>/dev/null | wc -l)
      [ "$left" = "0" ] && break
      sleep 5
    done
    [ "${left:-0}" != "0" ] && echo "  $left PersistentVolume(s) still there; their disks go with the node resource group"
  fi
  if az aks nodepool show --resource-group "$RG" --cluster-name "$CLUSTER" --name gpupool -o none 2>/dev/null; then
    echo "deleting the GPU pool ..."
    az aks nodepool delete --resource-group "$RG" --cluster-name "$CLUSTER" --name gpupool -o none
  fi
  echo "deleting cluster $CLUSTER ..."
  az aks delete --resource-group "$RG" --name "$CLUSTER" --yes -o none
  # what the cluster leaves in its node resource group when a finalizer was skipped
  if [ -n "${node_rg:-}" ] && az group show --name "$node_rg" -o none 2>/dev/null; then
    echo "node resource group $node_rg still exists; its contents:"
    az resource list --resource-group "$node_rg" --query '[].{type:type,name:name}' -o table || true
  fi
fi

if [ "${KEEP_GROUP:-0}" = "1" ]; then
  echo "kept resource group $RG; left in it:"
  az resource list --resource-group "$RG" --query '[].{type:type,name:name}' -o table
  exit 0
fi
if [ "${WAIT:-0}" = "1" ]; then
  az group delete --name "$RG" --yes && echo "resource group $RG deleted"
else
  az group delete --name "$RG" --yes --no-wait
  echo "resource group $RG is being deleted (az group show -n $RG to follow)"
fi
kubectl config delete-context "$CLUSTER" >/dev/null 2>&1 || true
