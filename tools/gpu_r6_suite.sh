#!/bin/bash
# Round 6: the whole GPU suite + smoke + headline bench (short), outputs under gpurun_out/r6suite
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6suite; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
python -c "import sys; sys.path.insert(0,'tests'); from conftest import model_dir; print(model_dir('full'))" > /dev/null 2>&1
echo "== pytest -m gpu"
timeout 2400 python -m pytest tests -m gpu -q -rf --tb=short -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest.log
grep -n "ERROR\|timed out\|holes" $O/pytest.log | head -20
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
echo "== headline bench"
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 1500 $O/bench.json
cp -r gpurun_out/diag $O/diag 2>/dev/null
