#!/bin/bash
# parity tests + headline bench (no profiling)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x -rf --tb=short -p no:cacheprovider > gpurun_out/pytest.log 2>&1
echo "pytest rc=$?"; tail -15 gpurun_out/pytest.log
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_quick.json 2> gpurun_out/bench_quick.err
echo "bench rc=$?"; python - <<'PY'
import json
d=json.load(open("gpurun_out/bench_quick.json"))
print({k:d[k] for k in ("value","ms_per_step","decode_tok_s","decode_ms_per_token","encode_ms","prefill_ms")})
print({k:v.get("avg_us") for k,v in d["roofline"]["kernels"].items()})
PY
tail -3 gpurun_out/bench_quick.err
