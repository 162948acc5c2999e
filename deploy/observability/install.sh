#!/usr/bin/env bash
# Installs Prometheus + Grafana (kube-prometheus-stack), the custom-metrics adapter and the
# dashboards of this directory into the "monitoring" namespace.
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
helm repo add prometheus-community https://prometheus-community.github.io/helm-charts
helm repo update
helm upgrade --install kube-prom-stack prometheus-community/kube-prometheus-stack \
  --namespace monitoring --create-namespace -f "${here}/kube-prom-stack.yaml" --wait
helm upgrade --install prometheus-adapter prometheus-community/prometheus-adapter \
  --namespace monitoring -f "${here}/prom-adapter.yaml"
kubectl -n monitoring create configmap vpsa-serving-dashboard \
  --from-file="${here}/serving-dashboard.json" --dry-run=client -o yaml |
  kubectl label --local -f - grafana_dashboard=1 -o yaml | kubectl apply -f -
kubectl apply -f "${here}/kv-tier-dashboard-cm.yaml"
echo "Grafana: kubectl -n monitoring port-forward svc/kube-prom-stack-grafana 3000:80"
