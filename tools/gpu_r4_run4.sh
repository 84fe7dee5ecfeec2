#!/bin/bash
# Round-4 call 4: w13x round-1 issue point A/B; fp8 fused kernel: correctness and speed.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4d; rm -rf $O; mkdir -p $O
python -c "import sys; sys.path.insert(0,'tests'); from conftest import model_dir; print(model_dir('full'))" > /dev/null 2>&1
echo "== correctness: EARLY variants + fp8 fused"
for E in 1 2; do VOX_HIP_W13_EARLY=$E timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "fused_decode_step_matches_the_launch or stream_full_size_matches_reference_golden" 2>&1 | tail -2; done
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "fp8" 2>&1 | tail -5
cat gpurun_out/diag/fp8_fused_vs_fp8_chain.json gpurun_out/diag/fp8_vs_bf16.json 2>/dev/null
echo "== sweep: w13x EARLY"
timeout 900 python tools/pf_sweep.py --reps 4 --iters 100 --kv 232,1900 --profile base: \
  "early1:VOX_HIP_W13_EARLY=1" "early2:VOX_HIP_W13_EARLY=2" "early1_pf:VOX_HIP_W13_EARLY=1;VOX_HIP_PF=24,0,3" "early2_pf:VOX_HIP_W13_EARLY=2;VOX_HIP_PF=24,0,3" 2>&1 | tee $O/sweep5.txt
echo "== fp8 speed"
for V in "" "VOX_HIP_FP8_ATTN_BF16=1"; do
  env $V timeout 600 python bench.py --weights fp8 --steps 3 --warmup 1 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$V', d['value'], d['decode_tok_s'], d['decode_ms_per_token'], d['roofline']['decode_step']['frac_of_peak'], d['dtype'][:30])"
done
for V in "base:" "early2:VOX_HIP_W13_EARLY=2"; do
    NAME=${V%%:*}; ENVS=${V#*:}
    env $(echo $ENVS | tr ';' ' ') VOX_HIP_FUSE_TL=$O/tl_$NAME.txt timeout 300 python tools/fuse_tl_kv.py 232 > $O/tl_$NAME.log 2>&1
    python tools/fuse_timeline.py $O/tl_$NAME.txt > $O/timeline_$NAME.txt 2>&1; rm -f $O/tl_$NAME.txt
    echo "=== $NAME"; grep -A8 "^k_gemv_w13x: 256" $O/timeline_$NAME.txt | head -9; grep -A6 "k_gemv_w13x phase" $O/timeline_$NAME.txt
done
