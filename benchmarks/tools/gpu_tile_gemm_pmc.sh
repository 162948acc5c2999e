# Counters and launch geometry of the prefill GEMM kernel (tile_gemm.hip) and of the library GEMM on
# the same shapes: a kernel trace (registers, LDS, workgroup size per kernel), then one counter pass
# with 8 SQ + 1 GRBM counters and one with 4 TCC counters.
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/tg_trace -- python3 benchmarks/tile_gemm_micro.py --skip-check --rounds 1 --inner 2 > gpurun_out/tg_trace.log 2>&1
python3 - <<'PY'
import csv, glob
rows = {}
for f in glob.glob("gpurun_out/tg_trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"][:60]
        if "tile_gemm" in n or "Cijk" in n or "grouped_gemm" in n:
            key = (n, r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"), r.get("Workgroup_Size_X"), r.get("Grid_Size_X"))
            rows.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in rows.items():
    print(k, "n=%d" % len(v), "min_us=%.1f" % min(v))
PY
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_tg -- python3 benchmarks/tile_gemm_micro.py --skip-check --rounds 1 --inner 2 > gpurun_out/pmc_tg.log 2>&1
python3 tools/pmc_summary.py gpurun_out/pmc_tg tile_gemm Cijk > gpurun_out/pmc_tg_summary.json
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum --output-format csv -d gpurun_out/pmc_tg_tcc -- python3 benchmarks/tile_gemm_micro.py --skip-check --rounds 1 --inner 2 > gpurun_out/pmc_tg_tcc.log 2>&1
python3 tools/pmc_summary.py gpurun_out/pmc_tg_tcc tile_gemm Cijk > gpurun_out/pmc_tg_tcc_summary.json
python3 - <<'PY'
import json
for f in ("gpurun_out/pmc_tg_summary.json", "gpurun_out/pmc_tg_tcc_summary.json"):
    d = json.load(open(f))
    for k, v in d.items():
        print(k[:70], {c: (round(x["mean"] / 1e6, 2), x["dispatches"]) for c, x in v.items()})
PY
