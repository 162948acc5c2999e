#!/usr/bin/env bash
# Router stress: 10000 chat requests at concurrency 2000 over 2 fake backends, then the split of
# routing decisions between the backends (from the router log) must be within 20 %.
set -euo pipefail
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)"
cd "$root"
python benchmarks/router_stress.py --engines 2 --requests "${REQUESTS:-10000}" --concurrency "${CONCURRENCY:-2000}" \
  --tokens 1 --routing-logic roundrobin --max-imbalance 0.2
