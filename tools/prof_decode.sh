#!/bin/bash
# rocprofv3 kernel stats of a decode-only run (tools/pmc_decode.py): per-kernel durations of the decode step.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import sys; sys.path.insert(0,'tests'); from conftest import model_dir; print(model_dir('full'))" > /dev/null 2>&1
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_dec" -o d -- \
    python "$GRAFT_REPO_ROOT/tools/pmc_decode.py" 40 > "$GRAFT_REPO_ROOT/gpurun_out/prof_dec.out" 2> "$GRAFT_REPO_ROOT/gpurun_out/prof_dec.err" )
cat gpurun_out/prof_dec.out
f=$(find gpurun_out/prof_dec -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f" | cut -c1-200
