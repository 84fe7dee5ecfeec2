#!/bin/bash
# Round 6: k_enc_stack, three builds on one box (ab_base, the tree, ab_b): us per layer by rows
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6n; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
python -c "import sys; sys.path.insert(0,'tests'); from conftest import model_dir; print(model_dir('full'))" > /dev/null 2>&1
for r in 1 2 3; do
  TAG="base" VOX_LIB_DIR=$(realpath ab_base) python tools/enc_rows_probe.py 25,1,8,16,32 750 30 2>&1 | tail -n 1 | tee -a $O/enc_rows_ab.txt
  TAG="tree" python tools/enc_rows_probe.py 25,1,8,16,32 750 30 2>&1 | tail -n 1 | tee -a $O/enc_rows_ab.txt
  [ -d ab_b ] && TAG="ab_b" VOX_LIB_DIR=$(realpath ab_b) python tools/enc_rows_probe.py 25,1,8,16,32 750 30 2>&1 | tail -n 1 | tee -a $O/enc_rows_ab.txt
done
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -k "encoder_stack or stream_small or smallrs" 2>&1 | tail -n 3 | tee -a $O/pytest.txt
