#!/bin/bash
# Short evidence run at a new head (the full one is tools/gpu_final_r5.sh): GPU suite, smoke, headline bench line, kernel stats of the
# headline command, the configurations a decode-kernel change touches (300 s clip, step time by KV length, config 3's stream line).
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/head5; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
python -c "import sys; sys.path.insert(0,'tests'); from conftest import model_dir; print(model_dir('full'))" > /dev/null 2>&1
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -q -rf --tb=short -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
echo "== headline bench"
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
echo "== rocprofv3 kernel stats of the headline command"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$O/prof" -o r5 -- \
    python "$GRAFT_REPO_ROOT/bench.py" --steps 2 --warmup 0 --no-cpu-baseline --no-pmc --no-graph-floor > /dev/null 2> "$GRAFT_REPO_ROOT/$O/prof.err" )
cp $(find $O/prof -name "r5_kernel_stats.csv" | head -1) $O/kernel_stats.csv 2>/dev/null; head -6 $O/kernel_stats.csv | cut -c1-140
python tools/trace_summary.py $O/prof --layer-of "k_qkv_finish" --out $O/head_trace_summary.txt > /dev/null 2>&1
rm -rf $O/prof
python tools/dec_step_probe.py full 40 232,600,1000,1900,3800,8000 2>&1 | tail -1 | tee $O/decode_step_by_kv.txt
timeout 600 python bench.py --seconds 300 --steps 2 --warmup 1 --no-cpu-baseline --no-pmc > $O/batch300_bench.json 2>/dev/null
timeout 900 python bench.py --mode stream --steps 1 --warmup 1 --no-pmc > $O/stream300_bench.json 2>/dev/null
for f in bench batch300_bench stream300_bench; do python - <<PY
import json
try:
    d=json.loads([l for l in open("$O/${f}.json") if l.startswith("{")][-1]); print("$f", d["value"], d.get("ms_per_step"), d.get("decode_tok_s"), d.get("decode_ms_per_token"), d.get("encode_ms"), d.get("prefill_ms"), d.get("chunk_latency_ms"), d.get("parity",{}).get("mismatches"), (d.get("roofline") or {}).get("frac"), (d.get("cpu_baseline") or {}).get("value"))
except Exception as ex: print("$f", "FAILED", ex)
PY
done
