# Full GPU test suite, the headline bench, and counters of the prefill attention kernels.
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/t_gpu.log 2>&1 || { tail -40 gpurun_out/t_gpu.log; exit 1; }
tail -3 gpurun_out/t_gpu.log
timeout -k 10 400 python bench.py --steps 10 --warmup 2 > gpurun_out/bench_1gpu.json 2> gpurun_out/bench_1gpu.err || { tail -30 gpurun_out/bench_1gpu.err; exit 1; }
cat gpurun_out/bench_1gpu.json
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_prefill -- python3 benchmarks/attn_micro.py --cases prefill_cold --variants 3,5 --iters 3 > gpurun_out/pmc_prefill.log 2>&1
python3 tools/pmc_summary.py gpurun_out/pmc_prefill paged_prefill > gpurun_out/pmc_prefill_summary.json
