# Grouped GEMM tests + micro benchmark, then the Mixtral-8x7B headline bench (short) with the
# per-expert library loop and with the grouped GEMM prefill.
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_kernels_gpu.py -x -q --timeout 120 --timeout-method thread -k "grouped_gemm or moe" > gpurun_out/t_gg.log 2>&1 || { tail -30 gpurun_out/t_gg.log; exit 1; }
tail -2 gpurun_out/t_gg.log
timeout -k 10 200 python3 benchmarks/gg_micro.py > gpurun_out/gg_micro.jsonl 2> gpurun_out/gg_micro.err || { tail -20 gpurun_out/gg_micro.err; exit 1; }
cat gpurun_out/gg_micro.jsonl
VPSA_MOE_PREFILL_GROUPED=0 timeout -k 10 500 python bench.py --model mixtral-8x7b --steps 3 --warmup 1 --reference-rounds 0 > gpurun_out/bench_mixtral_loop.json 2> gpurun_out/bench_mixtral_loop.err || { tail -30 gpurun_out/bench_mixtral_loop.err; exit 1; }
cut -c1-700 gpurun_out/bench_mixtral_loop.json
VPSA_MOE_PREFILL_GROUPED=1 timeout -k 10 500 python bench.py --model mixtral-8x7b --steps 3 --warmup 1 --reference-rounds 0 > gpurun_out/bench_mixtral_grouped.json 2> gpurun_out/bench_mixtral_grouped.err || { tail -30 gpurun_out/bench_mixtral_grouped.err; exit 1; }
cut -c1-700 gpurun_out/bench_mixtral_grouped.json
