#!/usr/bin/env bash
# usage: install.sh [kgateway|istio|gke]
set -euo pipefail
impl=${1:-kgateway}
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
IE_VERSION=${IE_VERSION:-v0.3.0}
kubectl apply -f "https://github.com/kubernetes-sigs/gateway-api/releases/download/v1.2.1/standard-install.yaml"
kubectl apply -f "https://github.com/kubernetes-sigs/gateway-api-inference-extension/releases/download/${IE_VERSION}/manifests.yaml"
kubectl apply -f "$here/configs/engines/llama3.yaml"
kubectl apply -f "$here/configs/inferencepool-resources.yaml"
kubectl apply -f "$here/configs/inferencemodel.yaml"
kubectl apply -f "$here/configs/gateway/$impl/gateway.yaml"
kubectl apply -f "$here/configs/httproute.yaml"
kubectl wait --for=condition=Programmed gateway/inference-gateway --timeout=300s || true
kubectl get gateway,httproute,inferencepool,inferencemodel
