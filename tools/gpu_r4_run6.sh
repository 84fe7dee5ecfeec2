#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4f; rm -rf $O; mkdir -p $O
python -c "import sys; sys.path.insert(0,'tests'); from conftest import model_dir; print(model_dir('full'))" > /dev/null 2>&1
echo "== dispatcher round robin"; ./tools/micro/xcd_rr | tee $O/xcd_rr.txt
echo "== fp8 correctness + speed"
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "fp8 or fused_decode_step_matches_the_launch" 2>&1 | tail -3
SWEEP_WEIGHTS=fp8 timeout 600 python tools/pf_sweep.py --reps 3 --iters 100 --kv 232 --profile fp8: "fp8_nopf:VOX_HIP_PF=0,0,0" "fp8_attn_bf16:VOX_HIP_FP8_ATTN_BF16=1" 2>&1 | tee $O/sweep_fp8.txt
SWEEP_WEIGHTS=fp8 VOX_HIP_FUSE_TL=$O/tl_fp8.txt timeout 300 python tools/fuse_tl_kv.py 232 > $O/tl_fp8.log 2>&1
python tools/fuse_timeline.py $O/tl_fp8.txt > $O/timeline_fp8.txt 2>&1; rm -f $O/tl_fp8.txt; grep -A14 "phase stamps over all" $O/timeline_fp8.txt | head -16
echo "== bf16 default (PF 24,0,3) vs off"
timeout 600 python tools/pf_sweep.py --reps 4 --iters 100 --kv 232,400 default: "off:VOX_HIP_PF=0,0,0" 2>&1 | tail -4
echo "== fp8 agreement table"
timeout 900 python tools/fp8_agreement.py $O/fp8_agreement.json 2>&1 | tail -4
