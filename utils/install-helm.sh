#!/usr/bin/env bash
set -euo pipefail
command -v helm >/dev/null && { helm version; exit 0; }
curl -fsSL https://raw.githubusercontent.com/helm/helm/main/scripts/get-helm-3 | bash
helm version
