#!/bin/bash
# Round-4 call 1: does the GPU box reach the model host?  L2-prefetch A/B of the decode step.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4a; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
echo "== network probe"
( timeout 8 curl -sS -o /dev/null -w "huggingface.co http %{http_code}\n" https://huggingface.co/mistralai/Voxtral-Mini-4B-Realtime-2602 2>&1 || echo "curl rc=$?" ) | tee $O/net_probe.txt
( timeout 5 getent hosts huggingface.co || echo "no DNS answer for huggingface.co" ) | tee -a $O/net_probe.txt
python -c "import sys; sys.path.insert(0,'tests'); from conftest import model_dir; print(model_dir('full'))" > /dev/null 2>&1
echo "== correctness with every prefetch on"
VOX_HIP_PF=72,1152,2 VOX_HIP_PF13=6,1650 VOX_HIP_PF2=2,850 timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider \
  -k "fused_decode_step_matches or stream_full_size_matches_reference_golden or fast_decode_kernels" 2>&1 | tail -3
VOX_HIP_PF=72,0,1 timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "fused_decode_step_matches_the_launch" 2>&1 | tail -2
echo "== sweep 1: fused-tail prefetch for k_gemv_w13x"
timeout 900 python tools/pf_sweep.py --profile base: \
  "a72_2:VOX_HIP_PF=72,0,2" "a36_2:VOX_HIP_PF=36,0,2" "a144_2:VOX_HIP_PF=144,0,2" \
  "a72_1:VOX_HIP_PF=72,0,1" "a36_1:VOX_HIP_PF=36,0,1" \
  "m72_all:VOX_HIP_PF=72,2304,2" "m72_half:VOX_HIP_PF=72,1152,2" "m36_all:VOX_HIP_PF=36,1152,2" "m108_half:VOX_HIP_PF=108,1728,2" "m144_half:VOX_HIP_PF=144,2304,2" \
  "m72_half1:VOX_HIP_PF=72,1152,1" 2>&1 | tee $O/sweep1.txt
echo "== sweep 2: early-finisher prefetch at the other two boundaries"
timeout 900 python tools/pf_sweep.py --profile base: \
  "p13_6:VOX_HIP_PF13=6,1650" "p13_3:VOX_HIP_PF13=3,1650" "p13_6all:VOX_HIP_PF13=6,999999" "p13_18all:VOX_HIP_PF13=18,999999" \
  "p2_2:VOX_HIP_PF2=2,850" "p2_2all:VOX_HIP_PF2=2,999999" "p2_6all:VOX_HIP_PF2=6,999999" \
  "all:VOX_HIP_PF=72,1152,2;VOX_HIP_PF13=6,1650;VOX_HIP_PF2=2,850" 2>&1 | tee $O/sweep2.txt
echo "== timelines"
for V in "base:" "m72_half:VOX_HIP_PF=72,1152,2" "a72_1:VOX_HIP_PF=72,0,1"; do
    NAME=${V%%:*}; ENVS=${V#*:}
    env $(echo $ENVS | tr ';' ' ') VOX_HIP_FUSE_TL=$O/tl_$NAME.txt timeout 300 python tools/fuse_tl_kv.py 232 > $O/tl_$NAME.log 2>&1
    python tools/fuse_timeline.py $O/tl_$NAME.txt > $O/timeline_$NAME.txt 2>&1; rm -f $O/tl_$NAME.txt
    echo "=== $NAME"; sed -n 2,6p $O/timeline_$NAME.txt; grep -A14 "phase stamps over all" $O/timeline_$NAME.txt | head -16; grep -A6 "k_gemv_w13x phase" $O/timeline_$NAME.txt
done
