#!/usr/bin/env bash
# Creates the PersistentVolume + claim "vllm-templates-pvc" that engines mount at /templates
# (modelSpec.chatTemplate without chatTemplateConfigMap), and copies *.jinja files into it.
#   scripts/setup_chat_templates.sh [-n namespace] [-d host_dir] [template.jinja ...]
set -euo pipefail
ns=default
dir=/mnt/vllm-templates
while getopts "n:d:" o; do
  case "$o" in n) ns=$OPTARG ;; d) dir=$OPTARG ;; *) echo "usage: $0 [-n ns] [-d dir] [files]"; exit 2 ;; esac
done
shift $((OPTIND - 1))
sudo mkdir -p "$dir"
for f in "$@"; do sudo cp "$f" "$dir/"; done
kubectl apply -n "$ns" -f - <<YAML
apiVersion: v1
kind: PersistentVolume
metadata:
  name: vllm-templates-pv
  labels: {app: vllm-templates}
spec:
  capacity: {storage: 1Gi}
  accessModes: [ReadOnlyMany]
  storageClassName: manual
  hostPath: {path: "$dir"}
---
apiVersion: v1
kind: PersistentVolumeClaim
metadata:
  name: vllm-templates-pvc
spec:
  accessModes: [ReadOnlyMany]
  storageClassName: manual
  resources: {requests: {storage: 1Gi}}
  selector: {matchLabels: {app: vllm-templates}}
YAML
echo "templates in $dir are served to engines at /templates"
