# Counters of the grouped GEMM kernel and of the library GEMM on the same shapes (one pass each:
# 8 SQ + 1 GRBM counters, then 4 TCC counters), after a plain timing run.
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 200 python3 benchmarks/gg_micro.py > gpurun_out/gg_micro.jsonl 2> gpurun_out/gg_micro.err || { tail -20 gpurun_out/gg_micro.err; exit 1; }
cat gpurun_out/gg_micro.jsonl
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_gg -- python3 benchmarks/gg_micro.py --cases dense --iters 2 > gpurun_out/pmc_gg.log 2>&1
python3 tools/pmc_summary.py gpurun_out/pmc_gg grouped_gemm Cijk > gpurun_out/pmc_gg_summary.json
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum --output-format csv -d gpurun_out/pmc_gg_tcc -- python3 benchmarks/gg_micro.py --cases dense,moe --iters 2 > gpurun_out/pmc_gg_tcc.log 2>&1
python3 tools/pmc_summary.py gpurun_out/pmc_gg_tcc grouped_gemm Cijk > gpurun_out/pmc_gg_tcc_summary.json
python3 - <<'PY'
import json
for f in ("gpurun_out/pmc_gg_summary.json", "gpurun_out/pmc_gg_tcc_summary.json"):
    d = json.load(open(f))
    for k, v in d.items():
        print(k[:70], {c: (round(x["mean"] / 1e6, 2), x["dispatches"]) for c, x in v.items()})
PY
