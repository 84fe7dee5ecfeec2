#!/bin/bash
# Round 6: kernel trace of the headline pass in front of the decode loop (mel .. prefill)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6f; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
python -c "import sys; sys.path.insert(0,'tests'); from conftest import model_dir; print(model_dir('full'))" > /dev/null 2>&1
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$O/prof" -o r6 -- \
    python "$GRAFT_REPO_ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --no-graph-floor --no-configs > $GRAFT_REPO_ROOT/$O/bench.json 2> "$GRAFT_REPO_ROOT/$O/prof.err" )
cp $(find $O/prof -name "r6_kernel_stats.csv" | head -1) $O/kernel_stats.csv 2>/dev/null
cp $(find $O/prof -name "r6_kernel_trace.csv" | head -1) $O/kernel_trace.csv; ls -la $O
rm -rf $O/prof
