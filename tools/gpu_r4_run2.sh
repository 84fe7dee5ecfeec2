#!/bin/bash
# Round-4 call 2: fine sweep of the fused-tail prefetch (mode 1: behind the Wo rows), more repetitions.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4b; rm -rf $O; mkdir -p $O
python -c "import sys; sys.path.insert(0,'tests'); from conftest import model_dir; print(model_dir('full'))" > /dev/null 2>&1
timeout 900 python tools/pf_sweep.py --reps 5 --iters 100 --kv 232,600 base: \
  "u24_1:VOX_HIP_PF=24,0,1" "u36_1:VOX_HIP_PF=36,0,1" "u48_1:VOX_HIP_PF=48,0,1" "u60_1:VOX_HIP_PF=60,0,1" "u24_2:VOX_HIP_PF=24,0,2" \
  "u36_p13:VOX_HIP_PF=36,0,1;VOX_HIP_PF13=3,1650" 2>&1 | tee $O/sweep3.txt
