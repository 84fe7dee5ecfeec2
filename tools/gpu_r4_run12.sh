#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4l; rm -rf $O; mkdir -p $O
python -c "import sys; sys.path.insert(0,'tests'); from conftest import model_dir; print(model_dir('full'))" > /dev/null 2>&1
VOX_HIP_FUSE_EARLY_SWEEP=1 VOX_HIP_PF=36,0,1 timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "fused_decode_step_matches_the_launch or stream_full_size_matches_reference_golden" 2>&1 | tail -2
timeout 900 python tools/pf_sweep.py --reps 4 --iters 100 --kv 100,232 --profile base: \
  "es_pf24_3:VOX_HIP_FUSE_EARLY_SWEEP=1" "es_pf0:VOX_HIP_FUSE_EARLY_SWEEP=1;VOX_HIP_PF=0,0,0" "es_pf24_1:VOX_HIP_FUSE_EARLY_SWEEP=1;VOX_HIP_PF=24,0,1" \
  "es_pf36_1:VOX_HIP_FUSE_EARLY_SWEEP=1;VOX_HIP_PF=36,0,1" "es_pf48_1:VOX_HIP_FUSE_EARLY_SWEEP=1;VOX_HIP_PF=48,0,1" "es_pf72_1:VOX_HIP_FUSE_EARLY_SWEEP=1;VOX_HIP_PF=72,0,1" 2>&1 | tee $O/sweep_es.txt
for V in "es_pf36_1:VOX_HIP_FUSE_EARLY_SWEEP=1;VOX_HIP_PF=36,0,1"; do
    NAME=${V%%:*}; ENVS=${V#*:}
    env $(echo $ENVS | tr ';' ' ') VOX_HIP_FUSE_TL=$O/tl_$NAME.txt timeout 300 python tools/fuse_tl_kv.py 232 > $O/tl_$NAME.log 2>&1
    python tools/fuse_timeline.py $O/tl_$NAME.txt > $O/timeline_$NAME.txt 2>&1; rm -f $O/tl_$NAME.txt
    echo "=== $NAME"; sed -n 2,6p $O/timeline_$NAME.txt; grep -A14 "phase stamps over all" $O/timeline_$NAME.txt | head -16
done
