#!/usr/bin/env bash
# Remove everything deploy/cloud/gcp/entry_point.sh created, and report what GKE leaves behind.
#       GCP_PROJECT=my-project clean_up.sh [cluster] [zone]
#
#   release + volumes   the chart and its PVCs first: a persistent disk provisioned for a claim
#                       is deleted by the CSI driver only while the cluster is alive
#   load balancers      Services of type LoadBalancer / Ingresses hold forwarding rules, target
#                       pools, health checks and firewall rules that outlive the cluster
#   node pools, cluster the GPU pool first (it is what costs), then the cluster
#   leftovers           unattached gke-<cluster>-* disks, k8s-fw-* firewall rules and
#                       forwarding rules of the cluster's network tag are listed (DELETE_LEFTOVERS=1
#                       deletes them)
# Steps of /root/reference/deployment_on_cloud/gcp/clean_up_basic.sh:1-49 (uninstall, delete the
# claims and volumes, delete the cluster), plus the leftovers a deleted GKE cluster is known for.
set -uo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
# shellcheck source=../common/lib.sh
. "$here/../common/lib.sh"
CLUSTER=${1:-${CLUSTER_NAME:-vpsa-gke}}
ZONE=${2:-${GCP_ZONE:-us-central1-a}}
PROJECT=${GCP_PROJECT:?set GCP_PROJECT}
need gcloud
gcloud config set project "$PROJECT" >/dev/null

if ! gcloud container clusters describe "$CLUSTER" --zone "$ZONE" >/dev/null 2>&1; then
  echo "cluster $CLUSTER does not exist in $ZONE"
else
  if command -v kubectl >/dev/null && gcloud container clusters get-credentials "$CLUSTER" --zone "$ZONE" 2>/dev/null; then
    uninstall_stack
    kubectl delete ingress --all -A --ignore-not-found --wait=false >/dev/null 2>&1 || true
    kubectl delete -f "$VPSA_ROOT/deploy/manifests/amd-device-plugin-ds.yaml" --ignore-not-found >/dev/null 2>&1 || true
    echo "waiting for persistent volumes to be deleted ..."
    for _ in $(seq 36); do
      left=$(kubectl get pv --no-headers 2>/dev/null | wc -l)
      [ "$left" = "0" ] && break
      sleep 5
    done
    if [ "${left:-0}" != "0" ]; then
      echo "  $left volume(s) did not go on their own; deleting the objects (their disks are listed below)"
      kubectl get pv -o name | xargs -r kubectl delete --wait=false || true
    fi
  fi
  if gcloud container node-pools describe gpu --cluster "$CLUSTER" --zone "$ZONE" >/dev/null 2>&1; then
    echo "deleting the GPU node pool ..."
    gcloud container node-pools delete gpu --cluster "$CLUSTER" --zone "$ZONE" --quiet
  fi
  echo "deleting cluster $CLUSTER ..."
  gcloud container clusters delete "$CLUSTER" --zone "$ZONE" --quiet
fi

disks=$(gcloud compute disks list --filter="name~^(gke-$CLUSTER|pvc-) AND -users:*" --format="value(name,zone.basename())" 2>/dev/null || true)
rules=$(gcloud compute firewall-rules list --filter="name~^k8s- AND targetTags~gke-$CLUSTER" --format="value(name)" 2>/dev/null || true)
fwd=$(gcloud compute forwarding-rules list --filter="description~$CLUSTER" --format="value(name,region.basename())" 2>/dev/null || true)
if [ -z "$disks$rules$fwd" ]; then
  echo "nothing left behind"
elif [ "${DELETE_LEFTOVERS:-0}" = "1" ]; then
  echo "$disks" | while read -r name zone; do [ -n "$name" ] && gcloud compute disks delete "$name" --zone "$zone" --quiet; done
  for r in $rules; do gcloud compute firewall-rules delete "$r" --quiet; done
  echo "$fwd" | while read -r name region; do [ -n "$name" ] && gcloud compute forwarding-rules delete "$name" --region "$region" --quiet; done
else
  [ -n "$disks" ] && printf 'unattached disks:\n%s\n' "$disks"
  [ -n "$rules" ] && printf 'firewall rules:\n%s\n' "$rules"
  [ -n "$fwd" ] && printf 'forwarding rules:\n%s\n' "$fwd"
  echo "run again with DELETE_LEFTOVERS=1 to delete them"
fi
kubectl config delete-context "gke_${PROJECT}_${ZONE}_${CLUSTER}" >/dev/null 2>&1 || true
