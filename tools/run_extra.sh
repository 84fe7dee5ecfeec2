#!/bin/bash
# extra measured configurations: fp8 decode weights (config 5), streaming 300 s (config 3)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --weights fp8 > gpurun_out/bench_fp8.json 2> gpurun_out/bench_fp8.err
echo "fp8 rc=$?"; python - <<'PY'
import json
d=json.load(open("gpurun_out/bench_fp8.json"))
print({k:d[k] for k in ("value","ms_per_step","decode_tok_s","decode_ms_per_token","encode_ms","prefill_ms")}, d["roofline"]["decode_step"])
PY
timeout 900 python bench.py --mode stream --seconds 300 --steps 1 --warmup 1 > gpurun_out/bench_stream300.json 2> gpurun_out/bench_stream300.err
echo "stream rc=$?"; cat gpurun_out/bench_stream300.json; tail -3 gpurun_out/bench_stream300.err
timeout 900 python bench.py --seconds 300 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/bench_batch300.json 2> gpurun_out/bench_batch300.err
echo "batch300 rc=$?"; python - <<'PY'
import json
d=json.load(open("gpurun_out/bench_batch300.json"))
print({k:d[k] for k in ("value","ms_per_step","decode_tok_s","decode_ms_per_token","encode_ms","prefill_ms","decoder_steps_per_pass")})
PY
