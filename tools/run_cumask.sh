#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
for m in full lo even q0; do
  if [ $m = full ]; then unset VOX_HIP_CUMASK; else export VOX_HIP_CUMASK=$m; fi
  timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/bench_cm_$m.json 2> gpurun_out/bench_cm_$m.err
  echo "== $m rc=$?"; python - <<PY
import json
d=json.load(open("gpurun_out/bench_cm_$m.json"))
print({k:d[k] for k in ("ms_per_step","decode_ms_per_token","encode_ms","prefill_ms")})
print({k:v.get("avg_us") for k,v in d["roofline"]["kernels"].items()})
PY
  tail -2 gpurun_out/bench_cm_$m.err
done
