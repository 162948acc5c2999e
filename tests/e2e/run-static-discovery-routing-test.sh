#!/usr/bin/env bash
# Starts 4 fake engines (or uses $BACKENDS if real engines are already up), then for every routing
# logic starts a router with static discovery, runs test_routing.py against it and stops it.
set -euo pipefail
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)"
cd "$root"
model=${MODEL:-fake_model_name}
port=${ROUTER_PORT:-30080}
logs=$(mktemp -d)
pids=()
cleanup() { for p in "${pids[@]:-}"; do kill "$p" 2>/dev/null || true; done; }
trap cleanup EXIT
if [ -z "${BACKENDS:-}" ]; then
  for p in 9001 9002 9003 9004; do
    python -m vllm_production_stack_amd.testing.fake_engine --port $p --model "$model" --speed 2000 >"$logs/engine-$p.log" 2>&1 &
    pids+=($!)
  done
  BACKENDS=http://127.0.0.1:9001,http://127.0.0.1:9002,http://127.0.0.1:9003,http://127.0.0.1:9004
  sleep 2
fi
n=$(awk -F, '{print NF}' <<<"$BACKENDS")
models=$(printf "$model,%.0s" $(seq "$n")); models=${models%,}
for logic in roundrobin session prefixaware disaggregated_prefill; do
  extra=()
  labels=""
  if [ "$logic" = session ]; then extra=(--session-key x-user-id); fi
  if [ "$logic" = disaggregated_prefill ]; then
    half=$((n / 2))
    labels=$(for i in $(seq "$n"); do [ "$i" -le "$half" ] && printf "prefill," || printf "decode,"; done); labels=${labels%,}
    extra=(--static-model-labels "$labels" --prefill-model-labels prefill --decode-model-labels decode)
  fi
  python -m vllm_production_stack_amd.router.app --host 127.0.0.1 --port "$port" --service-discovery static \
    --static-backends "$BACKENDS" --static-models "$models" --routing-logic "$logic" "${extra[@]}" \
    >"$logs/router-$logic.log" 2>&1 &
  rp=$!
  for _ in $(seq 50); do curl -fs "http://127.0.0.1:$port/health" >/dev/null && break; sleep 0.2; done
  python tests/e2e/test_routing.py --router-url "http://127.0.0.1:$port" --log-file "$logs/router-$logic.log" \
    --routing-logic "$logic" --model "$model" || { tail -20 "$logs/router-$logic.log"; kill $rp; exit 1; }
  kill $rp; wait $rp 2>/dev/null || true
done
echo "all routing logics verified (logs in $logs)"
