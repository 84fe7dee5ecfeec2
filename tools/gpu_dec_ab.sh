#!/bin/bash
# Decode-step A/B inside one GPU call: for every VARIANT ("name:ENV=1,ENV2=x" or "name:") the step time at five KV lengths
# (tools/dec_step_probe.py) and the per-workgroup timeline of the fused launch at 232 keys.
# usage: gpu_dec_ab.sh <outdir> variant...
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/$1; shift
rm -rf $O; mkdir -p $O
python -c "import sys; sys.path.insert(0,'tests'); from conftest import model_dir; print(model_dir('full'))" > /dev/null 2>&1
for V in "$@" "$@"; do
    NAME=${V%%:*}; ENVS=${V#*:}
    env $(echo $ENVS | tr ',' ' ') TAG=$NAME timeout 300 python tools/dec_step_probe.py full ${ITERS:-60} 2>&1 | tail -1
done
for V in "$@"; do
    NAME=${V%%:*}; ENVS=${V#*:}
    env $(echo $ENVS | tr ',' ' ') VOX_HIP_FUSE_TL=$O/tl_$NAME.txt timeout 300 python tools/fuse_tl_kv.py ${TL_KV:-232} > $O/tl_$NAME.log 2>&1
    python tools/fuse_timeline.py $O/tl_$NAME.txt > $O/timeline_$NAME.txt 2>&1; rm -f $O/tl_$NAME.txt
    echo "=== $NAME"; sed -n 2,6p $O/timeline_$NAME.txt; grep -A32 "phase stamps over all" $O/timeline_$NAME.txt | head -34
done
