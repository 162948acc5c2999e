#!/usr/bin/env bash
# Multi-round chats with an x-user-id header against a running router: every session must stay on
# one endpoint (verified from the router log).   usage: test-sticky-routing.sh <router-url> <router-log>
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
python "$here/test_routing.py" --router-url "${1:-http://localhost:30080}" --log-file "${2:?router log file}" \
  --routing-logic session --model "${MODEL:-fake_model_name}"
