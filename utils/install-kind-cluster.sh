#!/usr/bin/env bash
# CPU-only kind cluster for router / operator / chart tests (no GPUs inside kind).
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
"$here/install-kubectl.sh"
if ! command -v kind >/dev/null; then
  curl -fsSLo /tmp/kind https://kind.sigs.k8s.io/dl/latest/kind-linux-amd64
  sudo install -m 0755 /tmp/kind /usr/local/bin/kind && rm /tmp/kind
fi
kind get clusters | grep -qx vpsa || kind create cluster --name vpsa
kubectl cluster-info --context kind-vpsa
