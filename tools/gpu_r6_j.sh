#!/bin/bash
# Round 6: kernel trace of config 3's feed loop (60 s of audio) - where a feed's time goes with the stack kernel
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6j; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
python -c "import sys; sys.path.insert(0,'tests'); from conftest import model_dir; print(model_dir('full'))" > /dev/null 2>&1
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/$O/prof" -o s -- \
    python "$GRAFT_REPO_ROOT/bench.py" --mode stream --seconds 60 --steps 1 --warmup 0 --no-cpu-baseline --no-pmc --no-graph-floor --no-configs > $GRAFT_REPO_ROOT/$O/bench.json 2> "$GRAFT_REPO_ROOT/$O/prof.err" )
cp $(find $O/prof -name "s_kernel_trace.csv" | head -1) $O/kernel_trace.csv; rm -rf $O/prof; ls -la $O; tail -c 600 $O/bench.json
