# Counters of the prefill attention kernels (one pass, 8 SQ + 1 GRBM counters), then of the decode
# attention kernel (4 TCC counters, one pass).  Usage: gpu_attn_pmc.sh [prefill variants, default 5,9]
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
VARIANTS=${1:-5,9}
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_prefill -- python3 benchmarks/attn_micro.py --cases prefill_cold --variants $VARIANTS --iters 3 > gpurun_out/pmc_prefill.log 2>&1
python3 tools/pmc_summary.py /tmp/pmc_prefill paged_prefill > gpurun_out/pmc_prefill_summary.json
timeout -s KILL 120 rocprofv3 --pmc TCC_EA0_RDREQ TCC_HIT TCC_MISS TCC_READ --output-format csv -d /tmp/pmc_decode -- python3 benchmarks/attn_micro.py --cases decode --iters 3 > gpurun_out/pmc_decode.log 2>&1
python3 tools/pmc_summary.py /tmp/pmc_decode paged_decode > gpurun_out/pmc_decode_summary.json
cat gpurun_out/pmc_prefill_summary.json gpurun_out/pmc_decode_summary.json
