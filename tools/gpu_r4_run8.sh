#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4h; rm -rf $O; mkdir -p $O
python -c "import sys; sys.path.insert(0,'tests'); from conftest import model_dir; print(model_dir('full'))" > /dev/null 2>&1
for N in 25 1; do
VOX_HIP_ENC_PF="0,0,0,0,0,0,0" VOX_HIP_ENC_TL=$O/enc_tl_$N.txt timeout 300 python tools/enc_rows_probe.py $N 750 10 2>&1 | tail -1
python tools/enc_timeline.py $O/enc_tl_$N.txt | tee $O/enc_timeline_$N.txt
done
