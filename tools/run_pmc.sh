#!/bin/bash
# PMC pass only (own run: --pmc with --kernel-trace, nothing else)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
( cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/pmc" -o r1 -- \
    python "$GRAFT_REPO_ROOT/tools/pmc_decode.py" 4 > "$GRAFT_REPO_ROOT/gpurun_out/pmc_bench.json" 2> "$GRAFT_REPO_ROOT/gpurun_out/pmc.err" )
echo "pmc rc=$?"; cat gpurun_out/pmc_bench.json; tail -3 gpurun_out/pmc.err; find gpurun_out/pmc -type f | head
python tools/pmc_summary.py gpurun_out/pmc gpurun_out/pmc_summary.json 2>&1 | tail -20
