# Per-kernel time of the headline bench phase (3 timed rounds, no second phase) on the final tree,
# then the counters of the prefill attention kernel, register-staged against direct-to-LDS.
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -s KILL 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bench_trace -- python3 bench.py --steps 3 --warmup 1 --reference-rounds 0 > gpurun_out/bench_trace.log 2>&1 || { tail -20 gpurun_out/bench_trace.log; exit 1; }
python3 - <<'PY'
import csv, glob
rows = []
for f in glob.glob("/tmp/bench_trace/**/*kernel_stats.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
agg = {}
for r in rows:
    a = agg.setdefault(r["Name"], [0, 0.0])
    a[0] += int(r["Calls"]); a[1] += float(r["TotalDurationNs"])
tot = sum(v[1] for v in agg.values())
with open("gpurun_out/r3_bench_kernel_stats.csv", "w") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage"])
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        w.writerow([n, c, int(t), round(t / c, 1), round(100 * t / tot, 3)])
print("total GPU ms", round(tot / 1e6, 1))
PY
head -25 gpurun_out/r3_bench_kernel_stats.csv | cut -c1-160
grep -o '"round_times_s": \[[^]]*\]' gpurun_out/bench_trace.log | head -1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_prefill -- python3 benchmarks/attn_micro.py --cases prefill_cold --variants 9,11 --iters 3 > gpurun_out/pmc_prefill.log 2>&1
python3 tools/pmc_summary.py /tmp/pmc_prefill paged_prefill > gpurun_out/r3_prefill_attention_direct_pmc.json
cat gpurun_out/r3_prefill_attention_direct_pmc.json
