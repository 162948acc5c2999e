set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 500 python -u -m pytest tests/test_kernels_gpu.py -x -q --timeout 120 --timeout-method thread -k "prefill or decode" > gpurun_out/t_attn.log 2>&1 && tail -3 gpurun_out/t_attn.log && \
timeout -k 10 300 python -u benchmarks/attn_micro.py --cases prefill_cold,prefill_last,prefill_warm --iters 5 > gpurun_out/attn_micro_v3.jsonl 2>gpurun_out/attn_micro_v3.err && cat gpurun_out/attn_micro_v3.jsonl
