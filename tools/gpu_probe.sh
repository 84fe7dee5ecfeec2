#!/bin/bash
# A/B harness: for every VARIANT ("name:ENV=1,ENV2=x" or "name:"), trace tools/enc_paths_probe.py under rocprofv3 and
# print the per-kernel summary.  usage: gpu_probe.sh <outdir> <seconds> <stream|batch|both> <layer-of substr> variant...
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/$1; SECS=$2; WHAT=$3; LAYER=$4; shift 4
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
python -c "import sys; sys.path.insert(0,'tests'); from conftest import model_dir; print(model_dir('${VOX_PROBE_PRESET:-small}'))" > /dev/null 2>&1
for V in "$@"; do
    NAME=${V%%:*}; ENVS=${V#*:}
    ( cd /tmp && env $(echo $ENVS | tr ',' ' ') VOX_PROBE_IDS="$GRAFT_REPO_ROOT/$O/ids_$NAME.npz" timeout 300 rocprofv3 --kernel-trace --output-format csv \
        -d "$GRAFT_REPO_ROOT/$O/tr_$NAME" -o t -- python "$GRAFT_REPO_ROOT/tools/enc_paths_probe.py" $SECS $WHAT > "$GRAFT_REPO_ROOT/$O/run_$NAME.log" 2>&1 )
    python tools/trace_summary.py $O/tr_$NAME --layer-of "$LAYER" --out $O/summary_$NAME.txt
    echo "=== $NAME ($ENVS)"; grep -E "^(stream|batch):" $O/run_$NAME.log | tail -2
    grep -A14 "pattern seen" $O/summary_$NAME.txt | head -${PROBE_LINES:-16}
    rm -rf $O/tr_$NAME
done
python - <<PY
import glob, numpy as np, os
fs = sorted(glob.glob("$O/ids_*.npz"))
if fs:
    base = np.load(fs[0])
    for f in fs[1:]:
        d = np.load(f)
        print(os.path.basename(f), {k: bool(np.array_equal(d[k], base[k])) for k in base.files})
PY
