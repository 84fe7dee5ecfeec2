#!/bin/bash
# Big-pass encoder kernels, per variant ("name:ENV=1,..."): rocprofv3 kernel stats of one 30 s batch transcription on the full preset
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/$1; shift; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
python -c "import sys; sys.path.insert(0,'tests'); from conftest import model_dir; print(model_dir('full'))" > /dev/null 2>&1
for V in "$@"; do
    NAME=${V%%:*}; ENVS=${V#*:}
    ( cd /tmp && env $(echo $ENVS | tr ',' ' ') timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$O/p_$NAME" -o t -- \
        python "$GRAFT_REPO_ROOT/bench.py" --steps 2 --warmup 0 --no-cpu-baseline --no-pmc > "$GRAFT_REPO_ROOT/$O/bench_$NAME.json" 2>/dev/null )
    echo "=== $NAME ($ENVS)"; python -c "
import json,sys
d=json.loads([l for l in open('$O/bench_$NAME.json') if l.startswith('{')][-1]); print({k:d.get(k) for k in ('value','encode_ms','prefill_ms','decode_ms_per_token')}, d['parity']['mismatches'])"
    grep -E "k_gemm_planes|k_attn_enc|k_splitk" $(find $O/p_$NAME -name "t_kernel_stats.csv" | head -1) | awk -F'","' '{printf "%-70s calls %s avg %.1f us\n", substr($1,2,70), $2, $4/1000}'
    rm -rf $O/p_$NAME
done
