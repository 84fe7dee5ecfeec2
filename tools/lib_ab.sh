#!/bin/bash
# Same-box A/B of a COMPILE-TIME change: two builds of libvoxhip.so / libvoxtral.so (the tree's own, and another one in $1, e.g. the
# previous commit's build copied to ab_base/), decode step time by KV length, alternating processes.
# usage: tools/lib_ab.sh <dir with the other build> [reps] [kv list]
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
BASE=$(realpath "$1"); REPS=${2:-3}; KV=${3:-232,600,1000,1900,3800,8000}
python -c "import sys; sys.path.insert(0,'tests'); from conftest import model_dir; print(model_dir('full'))" > /dev/null 2>&1
for r in $(seq 1 $REPS); do
    echo "rep $r other : $(VOX_LIB_DIR=$BASE python tools/dec_step_probe.py full 60 $KV 2>&1 | tail -1)"
    echo "rep $r tree  : $(python tools/dec_step_probe.py full 60 $KV 2>&1 | tail -1)"
done
