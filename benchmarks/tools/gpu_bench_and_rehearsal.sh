# Headline bench on one GPU (with the reference-configuration phase), then the 8-rank launch
# path rehearsed on the same card (every rank pinned to device 0, pools capped).
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 500 python bench.py --steps 10 --warmup 2 > gpurun_out/bench_1gpu.json 2> gpurun_out/bench_1gpu.err || { tail -30 gpurun_out/bench_1gpu.err; exit 1; }
cat gpurun_out/bench_1gpu.json
tail -5 gpurun_out/bench_1gpu.err
VPSA_BENCH_SINGLE_DEVICE=1 timeout -k 10 600 python bench.py --gpus 8 --steps 3 --warmup 1 --users-per-gpu 6 --warmup-users-per-gpu 1 --history-len 4000 --num-gpu-blocks 900 --reference-rounds 2 --reference-users-per-gpu 3 --reference-offload-gb 4 --open-loop-seconds 10 --open-loop-qps-per-gpu 0.5 > gpurun_out/bench_8rank_rehearsal.json 2> gpurun_out/bench_8rank_rehearsal.err || { tail -40 gpurun_out/bench_8rank_rehearsal.err; exit 1; }
cat gpurun_out/bench_8rank_rehearsal.json
tail -5 gpurun_out/bench_8rank_rehearsal.err
