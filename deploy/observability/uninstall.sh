#!/usr/bin/env bash
set -euo pipefail
helm -n monitoring uninstall prometheus-adapter || true
helm -n monitoring uninstall kube-prom-stack || true
kubectl -n monitoring delete configmap vpsa-serving-dashboard vpsa-kv-tier-dashboard --ignore-not-found
