#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_stream" -o r1 -- \
    python "$GRAFT_REPO_ROOT/bench.py" --mode stream --seconds 20 --steps 1 --warmup 1 > "$GRAFT_REPO_ROOT/gpurun_out/prof_stream.json" 2> "$GRAFT_REPO_ROOT/gpurun_out/prof_stream.err" )
echo rc=$?; cat gpurun_out/prof_stream.json; head -30 gpurun_out/prof_stream/r1_kernel_stats.csv | cut -c1-140
timeout 300 python -m pytest tests/test_gpu_multi.py -m gpu -q -x -k rccl -p no:cacheprovider 2>&1 | tail -3
