# Is the decode kernel's "14 % over-read" real?  Control: the same four L2 counters on the pure
# streaming-read probe (every byte of a 4 GiB buffer read exactly once), then on the decode kernel.
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/pmc_probe -- python3 benchmarks/gemm_micro.py --shapes qkv --rows 1 > gpurun_out/pmc_probe.log 2>&1
python3 tools/pmc_summary.py gpurun_out/pmc_probe hbm_read_probe > gpurun_out/pmc_probe_summary.json
timeout -s KILL 120 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/pmc_decode -- python3 benchmarks/attn_micro.py --cases decode --iters 2 > gpurun_out/pmc_decode.log 2>&1
python3 tools/pmc_summary.py gpurun_out/pmc_decode paged_decode > gpurun_out/pmc_decode_summary.json
timeout -s KILL 120 rocprofv3 --pmc TCC_REQ_sum TCC_READ_sum TCC_TAG_STALL_sum TCC_EA0_RDREQ_DRAM_sum --output-format csv -d gpurun_out/pmc_decode2 -- python3 benchmarks/attn_micro.py --cases decode --iters 2 > gpurun_out/pmc_decode2.log 2>&1 || true
python3 tools/pmc_summary.py gpurun_out/pmc_decode2 paged_decode > gpurun_out/pmc_decode2_summary.json || true
cat gpurun_out/pmc_probe_summary.json gpurun_out/pmc_decode_summary.json gpurun_out/pmc_decode2_summary.json
timeout -k 10 200 python3 benchmarks/kv_tier_micro.py > gpurun_out/kv_tier_micro.jsonl 2> gpurun_out/kv_tier_micro.err || { tail -20 gpurun_out/kv_tier_micro.err; exit 1; }
cut -c1-900 gpurun_out/kv_tier_micro.jsonl
