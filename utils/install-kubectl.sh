#!/usr/bin/env bash
# kubectl (latest stable) into /usr/local/bin.
set -euo pipefail
command -v kubectl >/dev/null && { kubectl version --client; exit 0; }
arch=$(uname -m); case "$arch" in x86_64) arch=amd64 ;; aarch64) arch=arm64 ;; esac
ver=$(curl -fsSL https://dl.k8s.io/release/stable.txt)
curl -fsSLo /tmp/kubectl "https://dl.k8s.io/release/${ver}/bin/linux/${arch}/kubectl"
sudo install -m 0755 /tmp/kubectl /usr/local/bin/kubectl && rm /tmp/kubectl
kubectl version --client
