#!/usr/bin/env bash
# Waits for the release's pods, forwards the router Service and checks /v1/models + one completion.
set -euo pipefail
kubectl wait --for=condition=Ready pod -l environment=test --timeout=20m
kubectl port-forward svc/vllm-router-service 30080:80 >/dev/null 2>&1 &
pf=$!; trap 'kill $pf' EXIT; sleep 3
hdr=(); [ -n "${VLLM_API_KEY:-}" ] && hdr=(-H "Authorization: Bearer $VLLM_API_KEY")
models=$(curl -fs "${hdr[@]}" localhost:30080/v1/models)
echo "$models"
for m in $(python -c 'import json,sys; print(" ".join(d["id"] for d in json.load(sys.stdin)["data"]))' <<<"$models"); do
  curl -fs "${hdr[@]}" localhost:30080/v1/completions -H 'Content-Type: application/json' \
    -d "{\"model\": \"$m\", \"prompt\": \"Once upon a time,\", \"max_tokens\": 10}" | grep -q '"choices"'
done
echo "functionality OK"
