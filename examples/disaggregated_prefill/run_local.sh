#!/usr/bin/env bash
# Disaggregated prefill on one node without Kubernetes: 1 prefill + 1 decode replica on GPUs 0 and 1
# and a router with the two-hop logic.  KV pages go prefill -> decode over xGMI (hipIpc peer copy).
#   examples/disaggregated_prefill/run_local.sh [model]      then: curl localhost:30080/v1/completions ...
set -euo pipefail
model=${1:-llama-3-8b}
export HSA_ENABLE_IPC_MODE_LEGACY=0
common=(serve "$model" --host 127.0.0.1 --max-model-len 32000 --enable-prefix-caching)
HIP_VISIBLE_DEVICES=1 LMCACHE_ENABLE_NIXL=True LMCACHE_NIXL_ROLE=receiver LMCACHE_NIXL_RECEIVER_PORT=55555 \
  python -m vllm_production_stack_amd.engine.cli "${common[@]}" --port 8200 \
  --kv-transfer-config '{"kv_connector":"LMCacheConnectorV1","kv_role":"kv_consumer"}' &
HIP_VISIBLE_DEVICES=0 LMCACHE_ENABLE_NIXL=True LMCACHE_NIXL_ROLE=sender LMCACHE_NIXL_RECEIVER_HOST=127.0.0.1 \
  LMCACHE_NIXL_RECEIVER_PORT=55555 \
  python -m vllm_production_stack_amd.engine.cli "${common[@]}" --port 8100 \
  --kv-transfer-config '{"kv_connector":"LMCacheConnectorV1","kv_role":"kv_producer"}' &
python -m vllm_production_stack_amd.router.app --host 0.0.0.0 --port 30080 --service-discovery static \
  --static-backends http://127.0.0.1:8100,http://127.0.0.1:8200 --static-models "$model,$model" \
  --static-model-labels prefill,decode --routing-logic disaggregated_prefill \
  --prefill-model-labels prefill --decode-model-labels decode &
trap 'kill $(jobs -p)' EXIT
wait
