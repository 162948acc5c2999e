#!/usr/bin/env bash
# The kind binary and a container runtime for it (Docker Engine from the distribution when none is
# present).  utils/install-kind-cluster.sh creates the cluster.
set -euo pipefail
KIND_VERSION=${KIND_VERSION:-v0.24.0}
if ! command -v docker >/dev/null && ! command -v podman >/dev/null; then
  sudo apt-get update && sudo apt-get install -y docker.io
  sudo systemctl enable --now docker
  sudo usermod -aG docker "$USER" || true
  echo "docker installed; log out and in again (or 'newgrp docker') before creating a cluster"
fi
if ! command -v kind >/dev/null; then
  arch=$(uname -m); case "$arch" in x86_64) arch=amd64 ;; aarch64) arch=arm64 ;; esac
  curl -fsSLo /tmp/kind "https://kind.sigs.k8s.io/dl/${KIND_VERSION}/kind-linux-${arch}"
  sudo install -m 0755 /tmp/kind /usr/local/bin/kind && rm /tmp/kind
fi
kind version
