#!/bin/bash
# per-dispatch trace of the headline configuration (full model): launch patterns of the prefill and the flush pass
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-th}; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
python -c "import sys; sys.path.insert(0,'tests'); from conftest import model_dir; print(model_dir('full'))" > /dev/null 2>&1
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/$O/tr" -o h -- \
    python "$GRAFT_REPO_ROOT/bench.py" --steps 2 --warmup 0 --no-cpu-baseline --no-pmc > $GRAFT_REPO_ROOT/$O/bench.json 2> "$GRAFT_REPO_ROOT/$O/tr.err" )
python tools/trace_summary.py $O/tr --layer-of "${2:-k_qkv_finish}" --out $O/summary.txt
grep -B1 -A14 "pattern seen" $O/summary.txt | head -60
python tools/phase_breakdown.py $O/tr 2>/dev/null | tail -20
rm -rf $O/tr
