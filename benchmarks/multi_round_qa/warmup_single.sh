#!/usr/bin/env bash
# One user, two rounds, 10 s: loads kernels / captures decode graphs before the timed points.
set -euo pipefail
model=${1:?model}; url=${2:?base url}
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
python "$here/multi_round_qa.py" --num-users 1 --num-rounds 2 --shared-system-prompt 1000 \
  --user-history-prompt 20000 --answer-len 100 --qps 2 --model "$model" --base-url "$url" \
  --time 10 --output /dev/null --log-interval 30
