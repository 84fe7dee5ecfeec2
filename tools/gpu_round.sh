#!/bin/bash
# One GPU-box session: parity tests, smoke, headline bench, rocprofv3 kernel stats.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== rocminfo"; /opt/rocm/bin/rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9" | head -4
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -q -rf --tb=short --durations=8 -p no:cacheprovider > gpurun_out/pytest.log 2>&1
echo "pytest rc=$?"; tail -25 gpurun_out/pytest.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
echo "smoke rc=$?"; tail -3 gpurun_out/smoke.log
echo "== bench (full)"
timeout 900 python bench.py --steps 2 --warmup 1 > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err
echo "bench rc=$?"; tail -c 3000 gpurun_out/bench_full.json; tail -5 gpurun_out/bench_full.err
echo "== rocprofv3"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof" -o r1 -- \
    python "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/prof_bench.json" 2> "$GRAFT_REPO_ROOT/gpurun_out/prof.err" )
echo "rocprof rc=$?"; ls gpurun_out/prof 2>/dev/null | head; 
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -24 "$f"
echo "== rocprofv3 --pmc FETCH_SIZE (own pass, kernel-trace only)"
( cd /tmp && timeout 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/pmc" -o r1 -- \
    python "$GRAFT_REPO_ROOT/tools/pmc_decode.py" 4 > "$GRAFT_REPO_ROOT/gpurun_out/pmc_bench.json" 2> "$GRAFT_REPO_ROOT/gpurun_out/pmc.err" )
echo "pmc rc=$?"; ls gpurun_out/pmc 2>/dev/null | head
python tools/pmc_summary.py gpurun_out/pmc gpurun_out/pmc_summary.json 2>&1 | tail -20
