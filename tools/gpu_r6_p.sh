#!/bin/bash
# Round 6: large-M encoder layer with / without the two small fusions (VOX_HIP_DISABLE=enc_fuse), same box, alternating; parity tests
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6p; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
python -c "import sys; sys.path.insert(0,'tests'); from conftest import model_dir; print(model_dir('full'))" > /dev/null 2>&1
for r in 1 2 3; do
  TAG="unfused" VOX_HIP_DISABLE=enc_fuse python tools/enc_rows_probe.py 1664,1500,600 0 5 2>&1 | tail -n 1 | tee -a $O/ab.txt
  TAG="fused  " python tools/enc_rows_probe.py 1664,1500,600 0 5 2>&1 | tail -n 1 | tee -a $O/ab.txt
done
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -k "full_batch or twins or few_rows or encoder_forward or deep or multi_device or gemm_planes" 2>&1 | tail -n 4 | tee $O/pytest.txt
