#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/final4c; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
python -c "import sys; sys.path.insert(0,'tests'); from conftest import model_dir; print(model_dir('full'))" > /dev/null 2>&1
echo "== pytest -m gpu"
timeout 2400 python -m pytest tests -m gpu -q -rf --tb=short -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
timeout 900 python bench.py --steps 10 --warmup 3 --cold-load > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$O/prof" -o r4 -- \
    python "$GRAFT_REPO_ROOT/bench.py" --steps 2 --warmup 0 --no-cpu-baseline --no-pmc --no-graph-floor > /dev/null 2> "$GRAFT_REPO_ROOT/$O/prof.err" )
cp $(find $O/prof -name "r4_kernel_stats.csv" | head -1) $O/kernel_stats.csv 2>/dev/null; head -14 $O/kernel_stats.csv | cut -c1-140
python tools/trace_summary.py $O/prof --layer-of "k_qkv_finish" --out $O/head_trace_summary.txt > /dev/null 2>&1; rm -rf $O/prof
( cd /tmp && timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace --output-format csv \
    -d "$GRAFT_REPO_ROOT/$O/pmc_enc" -o r4 -- python "$GRAFT_REPO_ROOT/tools/pmc_encoder.py" > /dev/null 2> "$GRAFT_REPO_ROOT/$O/pmc_enc.err" )
python tools/pmc_mfma_summary.py $O/pmc_enc $O/pmc_encoder_mfma.json 2>&1 | tail -6; rm -rf $O/pmc_enc
timeout 900 python bench.py --mode stream --steps 1 --warmup 1 > $O/stream300_bench.json 2>/dev/null
timeout 600 python bench.py --seconds 300 --steps 2 --warmup 1 --no-cpu-baseline --no-pmc > $O/batch300_bench.json 2>/dev/null
timeout 600 python bench.py --seconds 600 --steps 1 --warmup 1 --no-cpu-baseline --no-pmc > $O/batch600_bench.json 2>/dev/null
for f in stream300 batch300 batch600; do python - <<PY
import json
d=json.loads([l for l in open("$O/${f}_bench.json") if l.startswith("{")][-1]); print("$f", d["value"], d.get("ms_per_step"), d.get("decode_ms_per_token"), d.get("encode_ms"), d.get("chunk_latency_ms"), d.get("parity"))
PY
done
python - <<PY
import json
d=json.loads([l for l in open("$O/bench.json") if l.startswith("{")][-1]); print({k:d[k] for k in ("value","ms_per_step","decode_tok_s","decode_ms_per_token","encode_ms","prefill_ms","parity")}); print(d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"]["decode_step"]["frac_of_peak"]); print(d.get("cpu_baseline",{}).get("value"))
PY
