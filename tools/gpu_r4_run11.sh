#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4k; rm -rf $O; mkdir -p $O
python -c "import sys; sys.path.insert(0,'tests'); from conftest import model_dir; print(model_dir('full'))" > /dev/null 2>&1
echo "== decode tests"
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "fused or fp8 or production_kernels or residual_stream or two_decoders or stream_full_size or fast_decode" 2>&1 | tail -4
echo "== fp8 FFN fused A/B"
SWEEP_WEIGHTS=fp8 timeout 600 python tools/pf_sweep.py --reps 3 --iters 100 --kv 232 --profile fp8_ffn: "fp8_two_launches:VOX_HIP_NO_FFN_FUSED=1" 2>&1 | tail -4
echo "== headline bench"
timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo rc=$?
python - <<PY
import json
d=json.loads([l for l in open("$O/bench.json") if l.startswith("{")][-1]); print({k:d[k] for k in ("value","ms_per_step","decode_tok_s","decode_ms_per_token","encode_ms","prefill_ms","parity","active_paths")}); r=d["roofline"]; print(r["kernel"], r["achieved"], r["frac"], r["traffic"], r["avg_us_per_launch"], r["decode_step"]["frac_of_peak"], r["decode_step"]["ms"]); print(r["kernels"])
PY
