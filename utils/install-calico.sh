#!/usr/bin/env bash
set -euo pipefail
CALICO=${CALICO:-v3.28.0}
kubectl create -f "https://raw.githubusercontent.com/projectcalico/calico/${CALICO}/manifests/tigera-operator.yaml"
kubectl create -f "https://raw.githubusercontent.com/projectcalico/calico/${CALICO}/manifests/custom-resources.yaml"
kubectl taint nodes --all node-role.kubernetes.io/control-plane- || true
