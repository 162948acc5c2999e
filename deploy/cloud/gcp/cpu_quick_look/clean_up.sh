#!/usr/bin/env bash
# Remove the CPU quick-look cluster (/root/reference/deployment_on_cloud/gcp/OPT125_CPU/cleanup_ql.sh:1-40).
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
exec "$here/../clean_up.sh" "${CLUSTER_NAME:-vpsa-cpu-ql}" "${GCP_ZONE:-us-central1-a}"
