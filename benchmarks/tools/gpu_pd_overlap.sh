# Decode step latency with and without a concurrent KV push (two engines on one card), then the
# P/D GPU tests.
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 560 python benchmarks/pd_overlap.py --streams 32 --pushes 4 > gpurun_out/pd_overlap.jsonl 2> gpurun_out/pd_overlap.err || { tail -30 gpurun_out/pd_overlap.err; tail -20 gpurun_out/pdov_prefill.log; tail -20 gpurun_out/pdov_decode.log; exit 1; }
cat gpurun_out/pd_overlap.jsonl
grep "pushed" gpurun_out/pdov_prefill.log | tail -4
timeout -k 10 500 python -u -m pytest tests/test_pd_gpu.py -x -q --timeout 300 --timeout-method thread > gpurun_out/t_pd.log 2>&1 || { tail -30 gpurun_out/t_pd.log; exit 1; }
tail -3 gpurun_out/t_pd.log
