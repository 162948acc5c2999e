# Counters of the prefill attention kernels (one pass, 8 SQ + 1 GRBM counters), then a bulk timing.
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_prefill -- python3 benchmarks/attn_micro.py --cases prefill_cold --variants 3,5 --iters 3 > gpurun_out/pmc_prefill.log 2>&1
python3 tools/pmc_summary.py gpurun_out/pmc_prefill paged_prefill > gpurun_out/pmc_prefill_summary.json
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_prefill -- python3 benchmarks/attn_micro.py --cases prefill_cold --variants 3,5 --iters 3 > gpurun_out/kt_prefill.log 2>&1
timeout -k 10 200 python3 -u benchmarks/attn_micro.py --cases prefill_bulk,prefill_first --variants 3,5,4 --iters 5 > gpurun_out/attn_micro_bulk.jsonl 2>&1
cat gpurun_out/pmc_prefill_summary.json gpurun_out/attn_micro_bulk.jsonl
