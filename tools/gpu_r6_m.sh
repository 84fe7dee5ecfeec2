#!/bin/bash
# Round 6: config 3's feed latency, tree build against ab_base (VOX_LIB_DIR), alternating; then the stream / decode tests
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6m; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
python -c "import sys; sys.path.insert(0,'tests'); from conftest import model_dir; print(model_dir('full'))" > /dev/null 2>&1
for r in 1 2; do
  for w in base tree; do
    if [ $w = base ]; then export VOX_LIB_DIR=$(realpath ab_base); else unset VOX_LIB_DIR; fi
    timeout 600 python bench.py --no-configs --mode stream --seconds 120 --steps 1 --warmup 0 --no-cpu-baseline --no-pmc > $O/s_$w$r.json 2>/dev/null
    python - <<PY | tee -a $O/ab.txt
import json
d=json.loads([l for l in open("$O/s_$w$r.json") if l.startswith("{")][-1]); print("$w", d["value"], d.get("chunk_latency_ms"), d.get("parity",{}).get("mismatches"))
PY
  done
done
unset VOX_LIB_DIR
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "stream or continuous or cli or api or restart or step" 2>&1 | tail -n 5 | tee $O/pytest.txt
