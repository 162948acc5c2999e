#!/usr/bin/env bash
# CPU-only GKE cluster + the chart with two OPT-125m replicas + one request through the router.
# (/root/reference/deployment_on_cloud/gcp/OPT125_CPU/entrypoint_ql.sh:1-66.)
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
export CLUSTER_NAME=${CLUSTER_NAME:-vpsa-cpu-ql}
# shellcheck source=../../common/lib.sh
. "$here/../../common/lib.sh"
need gcloud kubectl helm
PROJECT=${GCP_PROJECT:?set GCP_PROJECT}
ZONE=${GCP_ZONE:-us-central1-a}
gcloud config set project "$PROJECT" >/dev/null
gcloud services enable container.googleapis.com compute.googleapis.com
if ! gcloud container clusters describe "$CLUSTER_NAME" --zone "$ZONE" >/dev/null 2>&1; then
  gcloud container clusters create "$CLUSTER_NAME" --zone "$ZONE" --num-nodes 2 --machine-type e2-standard-16 \
    --release-channel regular --enable-ip-alias
fi
gcloud container clusters get-credentials "$CLUSTER_NAME" --zone "$ZONE"
install_stack "$here/production_stack_specification_ql.yaml"
smoke_test
