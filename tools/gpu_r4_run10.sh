#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4j; rm -rf $O; mkdir -p $O
python -c "import sys; sys.path.insert(0,'tests'); from conftest import model_dir; print(model_dir('full'))" > /dev/null 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "fused_decode_step_matches_the_launch or stream_full_size_matches_reference_golden" 2>&1 | tail -2
timeout 900 python tools/pf_sweep.py --reps 3 --iters 100 --kv 232,1900 --profile "ffn_b:VOX_HIP_FFN_SWEEP=2" "ffn_b23:VOX_HIP_FFN_SWEEP=3" "two_launches:VOX_HIP_NO_FFN_FUSED=1" 2>&1 | tee $O/sweep_ffn.txt
VOX_HIP_FFN_SWEEP=3 VOX_HIP_FUSE_TL=$O/tl_ffn.txt timeout 300 python tools/fuse_tl_kv.py 232 > $O/tl_ffn.log 2>&1
python tools/fuse_timeline.py $O/tl_ffn.txt > $O/timeline_ffn.txt 2>&1; rm -f $O/tl_ffn.txt
grep -A4 "^k_gemv_w13x: 256" $O/timeline_ffn.txt | head -6; grep -A9 "k_ffn_fused phase" $O/timeline_ffn.txt
