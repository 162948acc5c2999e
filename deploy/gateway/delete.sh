#!/usr/bin/env bash
set -euo pipefail
impl=${1:-kgateway}
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
for f in configs/httproute.yaml "configs/gateway/$impl/gateway.yaml" configs/inferencemodel.yaml \
         configs/inferencepool-resources.yaml configs/engines/llama3.yaml; do
  kubectl delete -f "$here/$f" --ignore-not-found
done
