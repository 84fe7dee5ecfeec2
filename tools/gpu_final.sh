#!/bin/bash
# Round-end evidence run on one MI355X box: everything that goes under profiles/ (see DESIGN.md section 8).
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/final; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
python -c "import sys; sys.path.insert(0,'tests'); from conftest import model_dir; print(model_dir('full'))" > /dev/null 2>&1
if [ -z "$SKIP_TESTS" ]; then
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -q -rf --tb=short -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
fi
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
echo "== headline bench (with the live PMC sub-run and the live CPU sample)"
timeout 900 python bench.py --steps 10 --warmup 3 --cold-load > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
echo "== rocprofv3 kernel stats of the headline command"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$O/prof" -o r2 -- \
    python "$GRAFT_REPO_ROOT/bench.py" --steps 2 --warmup 0 --no-cpu-baseline --no-pmc > /dev/null 2> "$GRAFT_REPO_ROOT/$O/prof.err" )
cp $(find $O/prof -name "r2_kernel_stats.csv" | head -1) $O/kernel_stats.csv 2>/dev/null; head -12 $O/kernel_stats.csv | cut -c1-140
echo "== PMC FETCH_SIZE, decode only"
( cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/$O/pmc" -o r2 -- \
    python "$GRAFT_REPO_ROOT/tools/pmc_decode.py" 4 > /dev/null 2> "$GRAFT_REPO_ROOT/$O/pmc.err" )
python tools/pmc_summary.py $O/pmc $O/pmc_decode_summary.json 2>&1 | tail -8
echo "== PMC MFMA utilisation, encoder"
( cd /tmp && timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace --output-format csv \
    -d "$GRAFT_REPO_ROOT/$O/pmc_enc" -o r2 -- python "$GRAFT_REPO_ROOT/tools/pmc_encoder.py" > /dev/null 2> "$GRAFT_REPO_ROOT/$O/pmc_enc.err" )
python tools/pmc_mfma_summary.py $O/pmc_enc $O/pmc_encoder_mfma.json 2>&1 | tail -6
echo "== fused decode phase trace"
VOX_HIP_FUSE_TRACE=1 python tools/pmc_decode.py 20 2> $O/fused_decode_trace.txt | tail -1; grep -v synth $O/fused_decode_trace.txt
echo "== other configurations"
timeout 600 python bench.py --mode stream --seconds 300 --steps 1 --warmup 1 --no-cpu-baseline > $O/stream300_bench.json 2>/dev/null
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$O/prof_stream" -o s -- \
    python "$GRAFT_REPO_ROOT/bench.py" --mode stream --seconds 60 --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1 )
cp $(find $O/prof_stream -name "s_kernel_stats.csv" | head -1) $O/stream_kernel_stats.csv 2>/dev/null
timeout 600 python bench.py --seconds 300 --steps 2 --warmup 1 --no-cpu-baseline --no-pmc > $O/batch300_bench.json 2>/dev/null
timeout 600 python bench.py --seconds 600 --steps 1 --warmup 1 --no-cpu-baseline --no-pmc > $O/batch600_bench.json 2>/dev/null
timeout 600 python bench.py --weights fp8 --steps 3 --warmup 1 --no-cpu-baseline --no-pmc > $O/fp8_bench.json 2>/dev/null
VOX_HIP_NO_FUSED=1 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pmc > $O/chain_bench.json 2>/dev/null
VOX_DEVICES=0,0,0,0 timeout 600 python bench.py --seconds 300 --steps 1 --warmup 1 --no-cpu-baseline --no-pmc > $O/batch300_4engines_one_gpu_bench.json 2>/dev/null
for f in stream300 batch300 batch600 fp8 chain batch300_4engines_one_gpu; do python - <<PY
import json
try:
    d=json.load(open("$O/${f}_bench.json")); print("$f", d["value"], d.get("decode_ms_per_token"), d.get("encode_ms"), d.get("chunk_latency_ms"), d.get("parity",{}).get("mismatches"))
except Exception as ex: print("$f", "FAILED", ex)
PY
done
echo "== reference benchmark.py"
bash tools/run_reference_benchmark.sh > /dev/null 2>&1; cp gpurun_out/reference_benchmark_report.txt $O/ 2>/dev/null; tail -6 $O/reference_benchmark_report.txt
python - <<PY
import json
d=json.load(open("$O/bench.json")); print({k:d[k] for k in ("value","ms_per_step","decode_tok_s","decode_ms_per_token","encode_ms","prefill_ms","model_load_s","model_load_cold_s","parity")}); print(d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"]["decode_step"]["frac_of_peak"]); print(d.get("cpu_baseline",{}).get("value"))
PY
if [ -n "$CPU_BASELINE" ]; then
echo "== the unmodified reference CLI on this box's host cores, alone (nothing else running)"
timeout 1200 python tools/cpu_baseline_cli.py $O/cpu_baseline_cli.json > $O/cpu_baseline_cli.log 2>&1; head -30 $O/cpu_baseline_cli.json
fi
