#!/usr/bin/env bash
# The static-discovery routing test against real engine replicas: two `vpsa-engine serve`
# processes (facebook/opt-125m architecture, random weights) share the visible GPU, then
# run-static-discovery-routing-test.sh drives every routing logic through them.
# The reference runs the same check with two `vllm serve facebook/opt-125m` processes
# (/root/reference/tests/e2e/run-static-discovery-routing-test.sh:74-116).
set -euo pipefail
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)"
cd "$root"
model=${MODEL:-opt-125m}
logs=${LOG_DIR:-$(mktemp -d)}
mkdir -p "$logs"
pids=()
cleanup() { for p in "${pids[@]:-}"; do kill "$p" 2>/dev/null || true; done; }
trap cleanup EXIT
i=0
for port in 8001 8002; do
  i=$((i + 1))
  # --gpu_memory_utilization bounds the card's total use: the second replica starts on top of the first
  util=$(python3 -c "print(0.15 * $i)")
  python -m vllm_production_stack_amd.engine.cli serve "$model" --host 127.0.0.1 --port $port \
    --max-model-len 2048 --gpu_memory_utilization "$util" >"$logs/engine-$port.log" 2>&1 &
  pids+=($!)
  for _ in $(seq 600); do
    curl -fs "http://127.0.0.1:$port/health" >/dev/null 2>&1 && break
    kill -0 "${pids[-1]}" 2>/dev/null || { echo "engine on $port died"; tail -20 "$logs/engine-$port.log"; exit 1; }
    sleep 0.2
  done
  echo "engine on $port is up"
done
BACKENDS=http://127.0.0.1:8001,http://127.0.0.1:8002 MODEL="$model" bash tests/e2e/run-static-discovery-routing-test.sh
