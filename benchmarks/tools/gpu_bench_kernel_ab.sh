# Per-kernel time of the headline bench (short: 2 rounds) with the hand-written prefill GEMM off
# and on, same box: kernel trace + stats per arm, top kernels side by side, plus the bench's own
# round times under the profiler.
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
for v in 0 1; do
  VPSA_TILE_GEMM=$v timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/bench_kab_$v -- python3 bench.py --steps 2 --warmup 1 --reference-rounds 0 > gpurun_out/bench_kab_$v.log 2>&1 || { tail -20 gpurun_out/bench_kab_$v.log; exit 1; }
  echo tile=$v $(grep -o "\"round_times_s\": \[[^]]*\]" gpurun_out/bench_kab_$v.log | head -1)
done
python3 - <<'PY'
import csv, glob, re
for v in (0, 1):
    fs = glob.glob(f"gpurun_out/bench_kab_{v}/**/*kernel_stats.csv", recursive=True)
    rows = []
    for f in fs:
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print(f"--- VPSA_TILE_GEMM={v}: total {tot/1e6:.1f} ms")
    for r in rows[:16]:
        n = r["Name"]
        m = re.search(r"MT\d+x\d+x\d+", n)
        short = ("Cijk " + m.group(0)) if m else n[:48]
        print(f"{short:50s} calls={r['Calls']:>6} total_ms={float(r['TotalDurationNs'])/1e6:9.1f} avg_us={float(r['AverageNs'])/1e3:9.1f} pct={float(r['TotalDurationNs'])/tot*100:5.2f}")
PY
