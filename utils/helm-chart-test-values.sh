#!/usr/bin/env bash
# Renders every example values file with the chart (helm) and with the built-in renderer
# (vpsa-stack template) and fails on the first error.  CI runs this on chart changes.
set -euo pipefail
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
for f in "$root"/deploy/examples/values-*.yaml; do
  echo "== $f"
  python -m vllm_production_stack_amd.deploy.stack template -f "$f" >/dev/null
  if command -v helm >/dev/null; then helm template vllm "$root/deploy/helm" -f "$f" >/dev/null; fi
done
echo "all example values render"
