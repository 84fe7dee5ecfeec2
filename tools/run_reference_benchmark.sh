#!/bin/bash
# The reference's own benchmark.py, UNCHANGED (staged by oracle/Makefile into oracle/_ref/harness, not
# committed), against the unmodified reference CLI linked with the engine (oracle/_ref/voxtral_cli_hip)
# on the night1968 clips, with the full-size synthetic checkpoint.  SURVEY 8(f) row 4.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
mkdir -p gpurun_out
MODEL=$(python -c "import sys; sys.path.insert(0,'tests'); from conftest import model_dir; print(model_dir('full'))")
H=oracle/_ref/harness
timeout 900 python $H/benchmark.py --binary oracle/_ref/voxtral_cli_hip --model "$MODEL" \
    --samples-root $H/samples/benchmark/night1968 -n 2 --mode mi355x_hip --log gpurun_out/reference_benchmark.log \
    > gpurun_out/reference_benchmark_report.txt 2>&1
echo "benchmark.py rc=$?"; cat gpurun_out/reference_benchmark_report.txt
