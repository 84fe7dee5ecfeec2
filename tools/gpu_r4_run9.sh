#!/bin/bash
# Round-4 call 9: k_ffn_fused (FFN block as one launch): correctness, step time A/B, timeline.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4i; rm -rf $O; mkdir -p $O
python -c "import sys; sys.path.insert(0,'tests'); from conftest import model_dir; print(model_dir('full'))" > /dev/null 2>&1
echo "== correctness"
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "fused_decode_step_matches_the_launch or stream_full_size_matches_reference_golden or fp8_fused or production_kernels or residual_stream or two_decoders or suspension" 2>&1 | tail -5
cat gpurun_out/diag/fused_vs_chain.json; echo
echo "== step time"
timeout 900 python tools/pf_sweep.py --reps 4 --iters 100 --kv 232,600,1900 --profile ffn: "two_launches:VOX_HIP_NO_FFN_FUSED=1" 2>&1 | tee $O/sweep_ffn.txt
SWEEP_WEIGHTS=fp8 timeout 600 python tools/pf_sweep.py --reps 3 --iters 100 --kv 232 --profile fp8_ffn: "fp8_two_launches:VOX_HIP_NO_FFN_FUSED=1" 2>&1 | tail -4
echo "== timeline"
VOX_HIP_FUSE_TL=$O/tl_ffn.txt timeout 300 python tools/fuse_tl_kv.py 232 > $O/tl_ffn.log 2>&1
python tools/fuse_timeline.py $O/tl_ffn.txt > $O/timeline_ffn.txt 2>&1; rm -f $O/tl_ffn.txt
grep -B1 -A8 "^k_gemv_w13x: 256" $O/timeline_ffn.txt | head -12; grep -A9 "k_ffn_fused phase" $O/timeline_ffn.txt
