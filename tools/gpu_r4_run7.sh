#!/bin/bash
# Round-4 call 7: L2 prefetch jobs in the few-rows encoder layer: correctness + per-layer time by variant.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4g; rm -rf $O; mkdir -p $O
python -c "import sys; sys.path.insert(0,'tests'); from conftest import model_dir; print(model_dir('full'))" > /dev/null 2>&1
echo "== correctness (default = prefetch on)"
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "streaming_feeds_match or stream_tiny or few_rows_paths or stream_small" 2>&1 | tail -3
echo "== per-layer time, 25-row chunk (and 1, 8, 16, 32 rows)"
run() { TAG="$1" VOX_HIP_ENC_PF="$2" timeout 300 python tools/enc_rows_probe.py 25,1,8,16,32 750 30 2>&1 | tail -1; }
for rep in 1 2; do
run off       "0,0,0,0,0,0,0"
run all       "224,512,4096,4096,4096,4096,4096"
run qkv_only  "224,0,4096,0,0,0,0"
run comb_only "0,512,0,4096,4096,0,0"
run fin1_only "224,0,0,0,0,4096,4096"
run all_256   "248,1024,4096,4096,4096,4096,4096"
run all_small "120,256,4096,4096,4096,4096,4096"
run lim       "224,512,2048,1024,1200,1200,1200"
done 2>&1 | tee $O/enc_pf.txt
echo "== kernel trace of the streaming path with and without"
for V in off on; do
  [ $V = off ] && export VOX_HIP_ENC_PF="0,0,0,0,0,0,0" || unset VOX_HIP_ENC_PF
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$O/prof_$V" -o s -- \
      python "$GRAFT_REPO_ROOT/tools/enc_rows_probe.py" 25 750 10 > /dev/null 2>&1 )
  f=$(find $O/prof_$V -name "s_kernel_stats.csv" | head -1); echo "--- $V"; [ -n "$f" ] && head -12 "$f" | cut -d, -f1-4,8 | cut -c1-150
  cp "$f" $O/stream_kernel_stats_$V.csv 2>/dev/null; rm -rf $O/prof_$V
done
