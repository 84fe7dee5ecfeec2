#!/bin/bash
# Round 6: k_enc_stack after a change - its parity tests (stack vs launches, stream goldens incl. config 3 at full size), timeline, rows probe
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6l; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
python -c "import sys; sys.path.insert(0,'tests'); from conftest import model_dir; print(model_dir('full'))" > /dev/null 2>&1
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -k "encoder_stack or few_rows or stream_small or smallrs or stream300 or full_stream or continuous" 2>&1 | tail -n 5 | tee $O/pytest.txt
VOX_HIP_ENC_TL=$O/enc_tl.txt timeout 300 python tools/enc_rows_probe.py 25 750 5 > /dev/null 2>&1
python tools/enc_stack_timeline.py $O/enc_tl.txt.stack > $O/enc_stack_timeline_25rows.txt 2>&1; cat $O/enc_stack_timeline_25rows.txt
TAG=tree python tools/enc_rows_probe.py 25,1,8,16,32 750 30 2>&1 | tail -n 1 | tee $O/enc_rows.txt
