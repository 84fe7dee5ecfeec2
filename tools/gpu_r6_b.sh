#!/bin/bash
# Round 6: the stack kernel end to end - streaming goldens (config 3 itself, the -rs restart case) and the config-3 bench line, with and without it
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6b; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
TAG=stack timeout 600 python tools/enc_rows_probe.py 25,1,8,16,32 750 30 2>&1 | tail -1 | tee $O/enc_rows.txt
TAG=launches VOX_HIP_DISABLE=enc_stack timeout 600 python tools/enc_rows_probe.py 25,1,8,16,32 750 30 2>&1 | tail -1 | tee -a $O/enc_rows.txt
echo "== streaming goldens"
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -rf --tb=short -p no:cacheprovider \
    -k "stream300 or config3 or smallrs or encoder_stack or (golden and stream)" > $O/pytest_stream.log 2>&1; echo "rc=$?"; tail -8 $O/pytest_stream.log
echo "== config 3 bench"
timeout 900 python bench.py --mode stream --steps 1 --warmup 1 --no-pmc --no-cpu-baseline > $O/stream300_bench.json 2> $O/stream300.err; echo "rc=$?"
VOX_HIP_DISABLE=enc_stack timeout 900 python bench.py --mode stream --steps 1 --warmup 1 --no-pmc --no-cpu-baseline > $O/stream300_bench_launches.json 2> $O/stream300_launches.err; echo "rc=$?"
for f in stream300_bench stream300_bench_launches; do python - <<PY
import json
try:
    d=json.loads([l for l in open("$O/${f}.json") if l.startswith("{")][-1]); print("$f", d["value"], d.get("ms_per_step"), d.get("chunk_latency_ms"), d.get("parity",{}).get("mismatches"), (d.get("roofline") or {}).get("frac"), d.get("encode_ms"))
except Exception as ex: print("$f", "FAILED", ex)
PY
done
cp -r gpurun_out/diag $O/diag 2>/dev/null
