#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/ktrace" -o r1 -- \
    python "$GRAFT_REPO_ROOT/tools/pmc_decode.py" 6 > "$GRAFT_REPO_ROOT/gpurun_out/ktrace.out" 2> "$GRAFT_REPO_ROOT/gpurun_out/ktrace.err" )
echo "rc=$?"; cat gpurun_out/ktrace.out; tail -2 gpurun_out/ktrace.err
python tools/pmc_decode.py 50
VOX_HIP_NO_PDL=1 python tools/pmc_decode.py 50
