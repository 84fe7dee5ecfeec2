#!/bin/bash
# tests + headline bench + stream bench (quick regression run)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3b; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
python -c "import sys; sys.path.insert(0,'tests'); from conftest import model_dir; print(model_dir('full'))" > /dev/null 2>&1
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -q -rf --tb=short -p no:cacheprovider -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest.log
echo "== headline bench"
timeout 600 python bench.py --steps 5 --warmup 2 --no-pmc > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
timeout 600 python bench.py --mode stream --seconds 120 --steps 1 --warmup 1 --no-cpu-baseline > $O/stream120_bench.json 2>/dev/null
python - <<PY
import json
for f in ("bench", "stream120_bench"):
    try:
        d = json.loads([l for l in open("$O/%s.json" % f) if l.startswith("{")][-1])
        print(f, d["value"], d["ms_per_step"], d.get("decode_tok_s"), d.get("decode_ms_per_token"), d.get("encode_ms"), d.get("prefill_ms"), d.get("chunk_latency_ms"), d.get("parity", {}).get("mismatches"))
    except Exception as ex:
        print(f, "FAILED", ex)
PY
