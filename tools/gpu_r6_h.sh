#!/bin/bash
# Round 6: fp8 agreement with the mixed variant, both checkpoint families
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6h; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
python -c "import sys; sys.path.insert(0,'tests'); from conftest import model_dir; print(model_dir('full')); print(model_dir('full-rs'))" > /dev/null 2>&1
timeout 900 python tools/fp8_agreement.py $O/r06_fp8_agreement.json full > $O/agree.log 2>&1; echo "rc=$?"
timeout 900 python tools/fp8_agreement.py $O/r06_fp8_agreement_rs.json full-rs stream_fullrs_batch.npz > $O/agree_rs.log 2>&1; echo "rc=$?"
tail -n 3 $O/agree.log; tail -n 3 $O/agree_rs.log
