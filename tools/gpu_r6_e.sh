#!/bin/bash
# Round 6: A/B of the L2 prefetch in k_gemm_planes (VOX_HIP_GP_PF) on the 30 s chunk's encoder pass, same box, alternating
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6e; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
python -c "import sys; sys.path.insert(0,'tests'); from conftest import model_dir; print(model_dir('full'))" > /dev/null 2>&1
for r in 1 2 3; do
  TAG="base" python tools/enc_rows_probe.py 1664,1500 0 5 2>&1 | tail -1 | tee -a $O/ab.txt
  TAG="pf2 " VOX_HIP_GP_PF=1 python tools/enc_rows_probe.py 1664,1500 0 5 2>&1 | tail -1 | tee -a $O/ab.txt
done
VOX_HIP_GP_PF=1 timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider -k "gemm_planes or full_batch" 2>&1 | tail -3 | tee -a $O/ab.txt
