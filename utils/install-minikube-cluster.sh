#!/usr/bin/env bash
# Single-node minikube cluster on a machine with AMD Instinct GPUs: the "none" driver runs
# the kubelet on the host, so /dev/kfd and /dev/dri are visible to the device plugin.
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
"$here/install-kubectl.sh"
"$here/install-helm.sh"
if ! command -v minikube >/dev/null; then
  curl -fsSLo /tmp/minikube https://storage.googleapis.com/minikube/releases/latest/minikube-linux-amd64
  sudo install -m 0755 /tmp/minikube /usr/local/bin/minikube && rm /tmp/minikube
fi
[ -e /dev/kfd ] || echo "warning: /dev/kfd not found - is the amdgpu driver (ROCm) installed on this host?" >&2
sudo sysctl -w fs.protected_regular=0 >/dev/null
minikube start --driver=none --container-runtime=containerd
"$here/init-amd-gpu-setup-k8s.sh"
