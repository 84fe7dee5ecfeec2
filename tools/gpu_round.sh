#!/bin/bash
# One GPU-box session: parity tests, smoke, headline bench, rocprofv3 kernel stats, the reference CPU
# baseline through the unmodified CLI (in the background on the host cores) and the reference's benchmark.py.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== rocminfo"; /opt/rocm/bin/rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9" | head -4
# the full checkpoint once, up front (tests, bench and the CPU baseline share it)
python -c "import sys; sys.path.insert(0,'tests'); from conftest import model_dir; print(model_dir('full'))"
if [ "${CPU_BASELINE:-1}" = "1" ]; then
  ( timeout 1200 python tools/cpu_baseline_cli.py gpurun_out/cpu_baseline_cli.json > gpurun_out/cpu_baseline_cli.log 2>&1 ) &
  CPU_PID=$!
fi
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -q -rf --tb=short --durations=8 -p no:cacheprovider ${PYTEST_ARGS} > gpurun_out/pytest.log 2>&1
echo "pytest rc=$?"; tail -30 gpurun_out/pytest.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
echo "smoke rc=$?"; tail -3 gpurun_out/smoke.log
echo "== bench (full)"
timeout 900 python bench.py --steps 5 --warmup 2 > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err
echo "bench rc=$?"; tail -c 2500 gpurun_out/bench_full.json; tail -5 gpurun_out/bench_full.err
echo "== rocprofv3 kernel stats"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof" -o r2 -- \
    python "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-pmc > "$GRAFT_REPO_ROOT/gpurun_out/prof_bench.json" 2> "$GRAFT_REPO_ROOT/gpurun_out/prof.err" )
echo "rocprof rc=$?"
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -28 "$f"
echo "== reference benchmark.py against voxtral_cli_hip"
bash tools/run_reference_benchmark.sh 2>&1 | tail -20
if [ -n "$CPU_PID" ]; then echo "== waiting for the CPU baseline"; wait $CPU_PID; cat gpurun_out/cpu_baseline_cli.json; fi
