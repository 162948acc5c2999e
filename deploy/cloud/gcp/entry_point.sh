#!/usr/bin/env bash
# GKE cluster with a default pool and a GPU node pool, then the serving chart.
#
#   GCP_PROJECT=my-project GPU_MACHINE_TYPE=<AMD Instinct machine type> \
#     deploy/cloud/gcp/entry_point.sh setup deploy/cloud/gcp/production_stack_specification.yaml
#   deploy/cloud/gcp/entry_point.sh status | smoke | cleanup
#   deploy/cloud/gcp/entry_point.sh setup-cpu          # two OPT-125m replicas on CPU nodes: a plumbing check
#
# Environment: GCP_PROJECT (required), CLUSTER_NAME (vpsa-gke), GCP_ZONE (us-central1-a),
# GPU_MACHINE_TYPE (required for setup), GPU_NODES (1), GPU_DISK_GB (500).
# Flow of /root/reference/deployment_on_cloud/gcp/entry_point_basic.sh:1-77 and
# OPT125_CPU/entrypoint_ql.sh:1-66 (the CPU quick look).
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
# shellcheck source=../common/lib.sh
. "$here/../common/lib.sh"

CLUSTER=${CLUSTER_NAME:-vpsa-gke}
ZONE=${GCP_ZONE:-us-central1-a}
GPU_NODES=${GPU_NODES:-1}
GPU_DISK_GB=${GPU_DISK_GB:-500}

create_cluster() {  # create_cluster <default pool machine type> <nodes>
  need gcloud kubectl helm
  PROJECT=${GCP_PROJECT:?set GCP_PROJECT}
  gcloud config set project "$PROJECT" >/dev/null
  gcloud services enable container.googleapis.com compute.googleapis.com
  if ! gcloud container clusters describe "$CLUSTER" --zone "$ZONE" >/dev/null 2>&1; then
    gcloud container clusters create "$CLUSTER" --zone "$ZONE" --num-nodes "$2" --machine-type "$1" \
      --release-channel regular --enable-ip-alias --workload-pool "$PROJECT.svc.id.goog"
  fi
  gcloud container clusters get-credentials "$CLUSTER" --zone "$ZONE"
}

case "${1:-}" in
  setup)
    values=${2:?usage: $0 setup <values.yaml>}
    GPU_MACHINE_TYPE=${GPU_MACHINE_TYPE:?set GPU_MACHINE_TYPE to an AMD Instinct machine type of your project}
    create_cluster e2-standard-8 1
    if ! gcloud container node-pools describe gpu --cluster "$CLUSTER" --zone "$ZONE" >/dev/null 2>&1; then
      gcloud container node-pools create gpu --cluster "$CLUSTER" --zone "$ZONE" --num-nodes "$GPU_NODES" \
        --machine-type "$GPU_MACHINE_TYPE" --disk-size "$GPU_DISK_GB" --disk-type pd-ssd \
        --node-labels accelerator=amd-instinct --node-taints sku=gpu:NoSchedule
    fi
    wait_nodes_ready accelerator=amd-instinct "$GPU_NODES"
    install_gpu_plugin
    install_stack "$values" \
      --set 'servingEngineSpec.tolerations[0].key=sku' --set 'servingEngineSpec.tolerations[0].operator=Equal' \
      --set 'servingEngineSpec.tolerations[0].value=gpu' --set 'servingEngineSpec.tolerations[0].effect=NoSchedule'
    ;;
  setup-cpu)
    create_cluster e2-standard-16 2
    install_stack "$VPSA_ROOT/deploy/examples/values-01-cpu-plumbing.yaml"
    smoke_test
    ;;
  status)
    gcloud container node-pools list --cluster "$CLUSTER" --zone "$ZONE"
    kubectl -n "$NAMESPACE" get pods,svc,pvc
    ;;
  smoke) smoke_test ;;
  cleanup)
    exec "$here/clean_up.sh" "$CLUSTER" "$ZONE"
    ;;
  *) echo "usage: $0 setup <values.yaml> | setup-cpu | status | smoke | cleanup" >&2; exit 2 ;;
esac
