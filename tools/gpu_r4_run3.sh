#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4c; rm -rf $O; mkdir -p $O
python -c "import sys; sys.path.insert(0,'tests'); from conftest import model_dir; print(model_dir('full'))" > /dev/null 2>&1
timeout 900 python tools/pf_sweep.py --reps 5 --iters 100 --kv 232,400 base: \
  "u8_1:VOX_HIP_PF=8,0,1" "u12_1:VOX_HIP_PF=12,0,1" "u16_1:VOX_HIP_PF=16,0,1" "u24_1:VOX_HIP_PF=24,0,1" \
  "u12_3:VOX_HIP_PF=12,0,3" "u24_3:VOX_HIP_PF=24,0,3" "u36_3:VOX_HIP_PF=36,0,3" "u72_3:VOX_HIP_PF=72,0,3" 2>&1 | tee $O/sweep4.txt
