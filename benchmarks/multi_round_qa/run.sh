#!/usr/bin/env bash
# Multi-GPU procedure (8 replicas): 320 users x 10 rounds, same prompt lengths, QPS 0.1 .. 4.1 step 0.4.
#   run.sh <model> <base-url> [output-prefix]
set -euo pipefail
model=${1:?model}; url=${2:?base url}; key=${3:-stack}
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
python "$here/multi_round_qa.py" --num-users 1 --num-rounds 2 --shared-system-prompt 1000 \
  --user-history-prompt 20000 --answer-len 100 --qps 2 --model "$model" --base-url "$url" \
  --time 200 --output /dev/null --log-interval 30
uid=1000
for qps in 0.1 0.5 0.9 1.3 1.7 2.1 2.5 2.9 3.3 3.7 4.1; do
  python "$here/multi_round_qa.py" --num-users 320 --num-rounds 10 --shared-system-prompt 1000 \
    --user-history-prompt 20000 --answer-len 100 --qps "$qps" --model "$model" --base-url "$url" \
    --time 100 --init-user-id "$uid" --request-with-user-id --output "${key}_output_${qps}.csv" \
    --log-interval 30
  uid=$((uid + 4000))
done
