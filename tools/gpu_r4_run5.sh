#!/bin/bash
# Round-4 call 5: fp8 agreement table; fp8 decode-step anatomy (per-kernel profile + timeline).
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4e; rm -rf $O; mkdir -p $O
python -c "import sys; sys.path.insert(0,'tests'); from conftest import model_dir; print(model_dir('full'))" > /dev/null 2>&1
echo "== fp8 anatomy"
SWEEP_WEIGHTS=fp8 timeout 600 python tools/pf_sweep.py --reps 2 --iters 100 --kv 232,1900 --profile fp8: "fp8_attn_bf16:VOX_HIP_FP8_ATTN_BF16=1" "fp8_pf:VOX_HIP_PF=24,0,3" 2>&1 | tee $O/sweep_fp8.txt
for V in "fp8:" "fp8_attn_bf16:VOX_HIP_FP8_ATTN_BF16=1"; do
    NAME=${V%%:*}; ENVS=${V#*:}
    env $(echo $ENVS | tr ';' ' ') SWEEP_WEIGHTS=fp8 VOX_HIP_FUSE_TL=$O/tl_$NAME.txt timeout 300 python tools/fuse_tl_kv.py 232 > $O/tl_$NAME.log 2>&1
    python tools/fuse_timeline.py $O/tl_$NAME.txt > $O/timeline_$NAME.txt 2>&1; rm -f $O/tl_$NAME.txt
    echo "=== $NAME"; cat $O/timeline_$NAME.txt | head -60
done
echo "== fp8 agreement table"
timeout 900 python tools/fp8_agreement.py $O/fp8_agreement.json 2>&1 | tail -12
