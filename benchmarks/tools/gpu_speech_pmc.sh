# Counters of the speech encoder's attention kernel (one pass, 8 SQ + 1 GRBM counters) over
# 32 windows of the Whisper-small shape.
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_speech -- python3 benchmarks/speech_micro.py --batches 32 --tokens 8 > gpurun_out/pmc_speech.log 2>&1
python3 tools/pmc_summary.py /tmp/pmc_speech dense_attn_fwd > gpurun_out/pmc_speech_summary.json
cat gpurun_out/pmc_speech_summary.json
