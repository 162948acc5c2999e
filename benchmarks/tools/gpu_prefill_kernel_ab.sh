# Per-kernel time of a cold prefill (8 x 21k-token prompts through the engine) with the hand-written
# prefill GEMM on and off: kernel trace + stats of each arm, top kernels printed side by side.
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
for v in 0 1; do
  VPSA_TILE_GEMM=$v timeout -s KILL 280 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prefill_ab_$v -- python3 benchmarks/engine_micro.py --batches 8 --decode-steps 4 > gpurun_out/prefill_ab_$v.log 2>&1 || { tail -20 gpurun_out/prefill_ab_$v.log; exit 1; }
  grep -v amdgpu.ids gpurun_out/prefill_ab_$v.log | tail -3
done
python3 - <<'PY'
import csv, glob, re
for v in (0, 1):
    f = glob.glob(f"gpurun_out/prefill_ab_{v}/**/*kernel_stats.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print(f"--- VPSA_TILE_GEMM={v}: total {tot/1e6:.1f} ms")
    for r in rows[:14]:
        n = r["Name"]
        m = re.search(r"MT\d+x\d+x\d+", n)
        short = ("Cijk " + m.group(0)) if m else n[:48]
        print(f"{short:50s} calls={r['Calls']:>6} total_ms={float(r['TotalDurationNs'])/1e6:9.1f} avg_us={float(r['AverageNs'])/1e3:9.1f} pct={float(r['TotalDurationNs'])/tot*100:5.2f}")
PY
