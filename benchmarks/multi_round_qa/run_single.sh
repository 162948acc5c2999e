#!/usr/bin/env bash
# Single-GPU procedure: 15 users x 20 rounds, 1000 shared + 20000 per-user tokens, 100-token answers,
# offered QPS 0.1 .. 1.1, 100 s per point, against an already running stack.
#   run_single.sh <model> <base-url> [output-prefix]
set -euo pipefail
model=${1:?model}; url=${2:?base url}; key=${3:-single}
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
"$here/warmup_single.sh" "$model" "$url"
uid=1000
for qps in 0.1 0.3 0.5 0.7 0.9 1.1; do
  python "$here/multi_round_qa.py" --num-users 15 --num-rounds 20 --shared-system-prompt 1000 \
    --user-history-prompt 20000 --answer-len 100 --qps "$qps" --model "$model" --base-url "$url" \
    --time 100 --init-user-id "$uid" --request-with-user-id --output "${key}_output_${qps}.csv" \
    --log-interval 30
  uid=$((uid + 400))
done
