#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i -o "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*\|SQ_BUSY_CYCLES\|SQ_BUSY_CU_CYCLES\|GRBM_GUI_ACTIVE\|SQ_WAVE_CYCLES\|SQ_ACTIVE_INST_VALU\|SQ_INSTS_VALU\b" | sort -u | head -40
( cd /tmp && timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_enc" -o r1 -- \
    python "$GRAFT_REPO_ROOT/tools/pmc_encoder.py" > "$GRAFT_REPO_ROOT/gpurun_out/pmc_enc.out" 2> "$GRAFT_REPO_ROOT/gpurun_out/pmc_enc.err" )
echo "pmc rc=$?"; cat gpurun_out/pmc_enc.out; tail -3 gpurun_out/pmc_enc.err; find gpurun_out/pmc_enc -name "*.csv" | head
python tools/pmc_mfma_summary.py gpurun_out/pmc_enc gpurun_out/pmc_encoder_mfma.json
