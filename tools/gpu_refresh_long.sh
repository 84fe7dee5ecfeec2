#!/bin/bash
# refresh of the long-context lines of profiles/r03_* after a decode-side change (subset of tools/gpu_final_r3.sh)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/refresh; rm -rf $O; mkdir -p $O
python -c "import sys; sys.path.insert(0,'tests'); from conftest import model_dir; print(model_dir('full'))" > /dev/null 2>&1
python tools/dec_step_probe.py full 40 2>&1 | tail -1 | tee $O/decode_step_by_kv.txt
timeout 600 python bench.py --mode stream --seconds 300 --steps 1 --warmup 1 --no-cpu-baseline > $O/stream300_bench.json 2>/dev/null
timeout 600 python bench.py --seconds 300 --steps 2 --warmup 1 --no-cpu-baseline --no-pmc > $O/batch300_bench.json 2>/dev/null
timeout 600 python bench.py --seconds 600 --steps 1 --warmup 1 --no-cpu-baseline --no-pmc > $O/batch600_bench.json 2>/dev/null
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pmc > $O/headline_bench.json 2>/dev/null
for f in stream300 batch300 batch600 headline; do python - <<PY
import json
d=json.loads([l for l in open("$O/${f}_bench.json") if l.startswith("{")][-1]); print("$f", d["value"], d.get("ms_per_step"), d.get("decode_ms_per_token"), d.get("encode_ms"), d.get("prefill_ms"), d.get("chunk_latency_ms"), d.get("parity",{}).get("mismatches"))
PY
done
