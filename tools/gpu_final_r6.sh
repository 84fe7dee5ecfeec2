#!/bin/bash
# Round-6 evidence run on one MI355X box: everything that goes under profiles/r06_* (DESIGN.md section 8).
# usage: gpu_final_r5.sh   (env SKIP_TESTS=1 leaves the test-suite out, CPU_BASELINE=1 times the reference CLI on the host)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/final6; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
python -c "import sys; sys.path.insert(0,'tests'); from conftest import model_dir; print(model_dir('full'))" > /dev/null 2>&1
if [ -z "$SKIP_TESTS" ]; then
echo "== pytest -m gpu"
timeout 2400 python -m pytest tests -m gpu -q -rf --tb=short -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
fi
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
echo "== headline bench (live PMC sub-run, live CPU sample, cold load)"
timeout 1200 python bench.py --steps 10 --warmup 3 --cold-load > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
echo "== rocprofv3 kernel stats of the headline command"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$O/prof" -o r6 -- \
    python "$GRAFT_REPO_ROOT/bench.py" --no-configs --steps 2 --warmup 0 --no-cpu-baseline --no-pmc --no-graph-floor > /dev/null 2> "$GRAFT_REPO_ROOT/$O/prof.err" )
cp $(find $O/prof -name "r6_kernel_stats.csv" | head -1) $O/kernel_stats.csv 2>/dev/null; head -12 $O/kernel_stats.csv | cut -c1-140
python tools/trace_summary.py $O/prof --layer-of "k_qkv_finish" --out $O/head_trace_summary.txt > /dev/null 2>&1
echo "== PMC FETCH_SIZE, decode only"
( cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/$O/pmc" -o r6 -- \
    python "$GRAFT_REPO_ROOT/tools/pmc_decode.py" 4 > /dev/null 2> "$GRAFT_REPO_ROOT/$O/pmc.err" )
python tools/pmc_summary.py $O/pmc $O/pmc_decode_summary.json 2>&1 | tail -8
echo "== PMC MFMA utilisation, encoder"
( cd /tmp && timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace --output-format csv \
    -d "$GRAFT_REPO_ROOT/$O/pmc_enc" -o r6 -- python "$GRAFT_REPO_ROOT/tools/pmc_encoder.py" > /dev/null 2> "$GRAFT_REPO_ROOT/$O/pmc_enc.err" )
python tools/pmc_mfma_summary.py $O/pmc_enc $O/pmc_encoder_mfma.json 2>&1 | tail -6
echo "== fused decode timelines + step time by KV length, with and without the round-4 pieces"
VOX_HIP_FUSE_TL=$O/fuse_tl_232.txt python tools/fuse_tl_kv.py 232 > $O/tl232.log 2>&1; ( cat $O/tl232.log | tail -1; python tools/fuse_timeline.py $O/fuse_tl_232.txt ) > $O/fuse_timeline_kv232.txt 2>&1
VOX_HIP_FUSE_TL=$O/fuse_tl_1900.txt python tools/fuse_tl_kv.py 1900 > $O/tl1900.log 2>&1; ( cat $O/tl1900.log | tail -1; python tools/fuse_timeline.py $O/fuse_tl_1900.txt ) > $O/fuse_timeline_kv1900.txt 2>&1
head -12 $O/fuse_timeline_kv232.txt; rm -f $O/fuse_tl_232.txt $O/fuse_tl_1900.txt
python tools/dec_step_probe.py full 40 2>&1 | tail -1 | tee $O/decode_step_by_kv.txt
timeout 900 python tools/decode_ab.py --reps 3 --iters 100 --kv 232,600,1000,1900,3800,8000 round6: "one_launch_per_layer:VOX_HIP_DISABLE=stack" "round4:VOX_HIP_DISABLE=stack,merge12_long" \
    "two_launches_per_layer:VOX_HIP_DISABLE=merge12" "two_ffn_launches_too:VOX_HIP_DISABLE=merge12,ffn_fused" 2>&1 | tee $O/decode_ab.txt | tail -7
echo "== few-rows encoder layer: time by rows (stack kernel / launch-per-GEMM path), the stack kernel's per-phase timeline"
TAG=stack timeout 300 python tools/enc_rows_probe.py 25,1,8,16,32 750 30 2>&1 | tail -n 1 | tee $O/enc_rows.txt
TAG=launches VOX_HIP_DISABLE=enc_stack timeout 300 python tools/enc_rows_probe.py 25,1,8,16,32 750 30 2>&1 | tail -n 1 | tee -a $O/enc_rows.txt
VOX_HIP_ENC_TL=$O/enc_tl.txt timeout 300 python tools/enc_rows_probe.py 25 750 5 > /dev/null 2>&1
python tools/enc_stack_timeline.py $O/enc_tl.txt.stack > $O/enc_stack_timeline_25rows.txt 2>&1; rm -f $O/enc_tl.txt $O/enc_tl.txt.stack
echo "== other configurations"
timeout 900 python bench.py --no-configs --mode stream --steps 1 --warmup 1 > $O/stream300_bench.json 2>/dev/null
timeout 900 python bench.py --no-configs --mode stream --seconds 176 --steps 1 --warmup 0 --no-cpu-baseline > $O/stream176_bench.json 2>/dev/null
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$O/prof_stream" -o s -- \
    python "$GRAFT_REPO_ROOT/bench.py" --no-configs --mode stream --seconds 60 --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1 )
cp $(find $O/prof_stream -name "s_kernel_stats.csv" | head -1) $O/stream_kernel_stats.csv 2>/dev/null
python tools/trace_summary.py $O/prof_stream --layer-of "k_enc_stack" --out $O/stream_trace_summary.txt > /dev/null 2>&1
timeout 600 python bench.py --no-configs --seconds 300 --steps 2 --warmup 1 --no-cpu-baseline --no-pmc > $O/batch300_bench.json 2>/dev/null
timeout 600 python bench.py --no-configs --seconds 600 --steps 1 --warmup 1 --no-cpu-baseline --no-pmc > $O/batch600_bench.json 2>/dev/null
timeout 600 python bench.py --no-configs --weights fp8 --steps 3 --warmup 1 --no-cpu-baseline --no-pmc > $O/fp8_bench.json 2>/dev/null
VOX_HIP_DISABLE=fused timeout 600 python bench.py --no-configs --steps 3 --warmup 1 --no-cpu-baseline --no-pmc > $O/chain_bench.json 2>/dev/null
VOX_DEVICES=0,0,0,0,0,0,0,0 timeout 600 python bench.py --no-configs --seconds 600 --steps 1 --warmup 1 --no-cpu-baseline --no-pmc > $O/batch600_8engines_one_gpu_bench.json 2>/dev/null
echo "== distributed front end: N=1 over RCCL (self loop), N=2 sharing this GPU over gloo (bench.py launches its own ranks), config 4 on 2 ranks"
VOX_FORCE_DIST=1 VOX_DIST_SELF_LOOP=1 timeout 600 python -m torch.distributed.run --standalone --local-addr 127.0.0.1 --nnodes=1 --nproc-per-node=1 \
    bench.py --no-configs --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline 2> $O/dist1.err | grep "^{" > $O/dist1_rccl_bench.json
VOX_SHARE_GPU=1 timeout 900 python bench.py --no-configs --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline 2> $O/dist2.err | grep "^{" > $O/dist2_shared_gpu_gloo_bench.json
VOX_SHARE_GPU=1 VOX_DIST_MODE=single timeout 900 python bench.py --no-configs --gpus 2 --seconds 300 --steps 1 --warmup 0 --no-cpu-baseline 2> $O/dist2s.err | grep "^{" > $O/dist2_single_clip_600s_bench.json
rm -rf $O/prof $O/prof_stream $O/pmc $O/pmc_enc
for f in stream300 stream176 batch300 batch600 fp8 chain batch600_8engines_one_gpu dist1_rccl dist2_shared_gpu_gloo dist2_single_clip_600s; do python - <<PY
import json
try:
    d=json.loads([l for l in open("$O/${f}_bench.json") if l.startswith("{")][-1]); print("$f", d["value"], d.get("ms_per_step"), d.get("decode_ms_per_token"), d.get("encode_ms"), d.get("prefill_ms"), d.get("chunk_latency_ms"), d.get("parity",{}).get("mismatches"), (d.get("roofline") or {}).get("frac"), d.get("phases_ms"), (d.get("replica") or {}).get("value"))
except Exception as ex: print("$f", "FAILED", ex)
PY
done
echo "== reference benchmark.py"
bash tools/run_reference_benchmark.sh > /dev/null 2>&1; cp gpurun_out/reference_benchmark_report.txt $O/ 2>/dev/null; tail -6 $O/reference_benchmark_report.txt
python - <<PY
import json
d=json.loads([l for l in open("$O/bench.json") if l.startswith("{")][-1]); print({k:d[k] for k in ("value","ms_per_step","decode_tok_s","decode_ms_per_token","encode_ms","prefill_ms","model_load_s","model_load_cold_s","parity")}); print(d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"]["decode_step"]["frac_of_peak"]); print(d.get("cpu_baseline",{}).get("value"))
PY
if [ -n "$CPU_BASELINE" ]; then
echo "== the unmodified reference CLI on this box's host cores, alone (nothing else running)"
timeout 1200 python tools/cpu_baseline_cli.py $O/cpu_baseline_cli.json > $O/cpu_baseline_cli.log 2>&1; head -30 $O/cpu_baseline_cli.json
fi
