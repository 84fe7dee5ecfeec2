#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6d; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
python -c "import sys; sys.path.insert(0,'tests'); from conftest import model_dir; print(model_dir('full'))" > /dev/null 2>&1
if [ "$1" = "cli" ]; then
echo "== reference CLI on this host (cpu baseline, ~200 s)"
timeout 900 python tools/cpu_baseline_cli.py profiles/r06_cpu_baseline_cli.json > $O/cpu_cli.log 2>&1; echo "rc=$?"; tail -3 $O/cpu_cli.log
cp profiles/r06_cpu_baseline_cli.json $O/ 2>/dev/null
fi
echo "== default bench line (headline + configs)"
T0=$(date +%s.%N); timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$? wall $(echo "$(date +%s.%N) - $T0" | bc) s"
python - <<PY
import json
d=json.loads([l for l in open("$O/bench.json") if l.startswith("{")][-1])
print("headline", d["value"], d["ms_per_step"], d["decode_tok_s"], d["prefill_ms"], d["encode_ms"], d["parity"]["mismatches"], d["roofline"]["frac"], d["cpu_baseline"].get("value"))
for k, c in d.get("configs", {}).items():
    print(k, c.get("value"), c.get("parity"), (c.get("roofline") or {}).get("frac"), c.get("chunk_latency_ms"), c.get("error"))
PY
