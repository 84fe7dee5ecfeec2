#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -s -k "long_context_and_ring" --durations=5 -p no:cacheprovider 2>&1 | tail -30
cat gpurun_out/diag/fast_vs_generic_*.json
