#!/bin/bash
# Round 6, first GPU call: the encoder stack kernel (k_enc_stack) - parity against the launch-per-GEMM path, us per layer by rows, timeline.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6a; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
echo "== stack kernel tests (small presets)"
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -rf --tb=short -p no:cacheprovider \
    -k "encoder_stack or few_rows_paths" > $O/pytest_stack.log 2>&1; echo "rc=$?"; tail -15 $O/pytest_stack.log
echo "== us per encoder layer by rows (full preset)"
TAG=stack timeout 600 python tools/enc_rows_probe.py 25,1,8,16,32 750 30 2>&1 | tail -3 | tee $O/enc_rows.txt
TAG=launches VOX_HIP_DISABLE=enc_stack timeout 600 python tools/enc_rows_probe.py 25,1,8,16,32 750 30 2>&1 | tail -3 | tee -a $O/enc_rows.txt
echo "== timeline of the mid-stack layer (25 rows)"
VOX_HIP_ENC_TL=$O/enc_tl.txt timeout 600 python tools/enc_rows_probe.py 25 750 5 2>&1 | tail -2
python tools/enc_stack_timeline.py $O/enc_tl.txt.stack 2>&1 | tee $O/enc_stack_timeline_25rows.txt | head -40
