#!/bin/bash
# BASELINE config 3 (300 s in 0.5 s feeds, -I 0.5, continuous mode): bench line + rocprofv3 kernel stats of a 60 s run.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import sys; sys.path.insert(0,'tests'); from conftest import model_dir; print(model_dir('full'))" > /dev/null 2>&1
timeout 600 python bench.py --mode stream --seconds 300 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/bench_stream300.json 2> gpurun_out/bench_stream300.err
echo "stream rc=$?"; cat gpurun_out/bench_stream300.json
VOX_HIP_NO_SKINNY=1 timeout 600 python bench.py --mode stream --seconds 300 --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('NO_SKINNY', d['value'], d['chunk_latency_ms'])"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_stream" -o s -- \
    python "$GRAFT_REPO_ROOT/bench.py" --mode stream --seconds 60 --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> "$GRAFT_REPO_ROOT/gpurun_out/prof_stream.err" )
f=$(find gpurun_out/prof_stream -name "s_kernel_stats.csv" | head -1); [ -n "$f" ] && head -22 "$f" | cut -c1-170
timeout 600 python bench.py --seconds 300 --steps 2 --warmup 1 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('BATCH300', d['value'], d['decode_ms_per_token'], d['encode_ms'], d['prefill_ms'])"
timeout 600 python bench.py --seconds 600 --steps 1 --warmup 1 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('BATCH600', d['value'], d['decode_ms_per_token'], d['encode_ms'], d['prefill_ms'])"
