#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6c; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -rf --tb=short -p no:cacheprovider -k "ring_wrap_at_the_real_window or switched_out" > $O/pytest.log 2>&1; echo "rc=$?"; tail -12 $O/pytest.log
for i in 1 2; do timeout 900 python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -k "switched_out" > $O/pytest_churn$i.log 2>&1; echo "churn$i rc=$?"; cat gpurun_out/diag/spin_holes_under_churn.json; echo; done
cat gpurun_out/diag/ring_wrap_real_window.json
