#!/bin/bash
# Round 3, first GPU session: everything new on the box (tests incl. the N>1 bench path and the new goldens), the headline
# line, the distributed path at N=1, and per-dispatch traces of the streaming and the headline configuration.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3a; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
python -c "import sys; sys.path.insert(0,'tests'); from conftest import model_dir; print(model_dir('full'))" > /dev/null 2>&1
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -q -rf --tb=short -p no:cacheprovider -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest.log
echo "== headline bench"
timeout 600 python bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
echo "== distributed path, N=1 (RCCL, self loop)"
VOX_FORCE_DIST=1 VOX_DIST_SELF_LOOP=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29611 \
    bench.py --gpus 1 --steps 5 --warmup 2 > $O/dist1_bench.json 2> $O/dist1.err; echo "dist1 rc=$?"
python - <<PY
import json
for f in ("bench", "dist1_bench"):
    try:
        d = json.loads([l for l in open("$O/%s.json" % f) if l.startswith("{")][-1])
        print(f, d["value"], d["ms_per_step"], d.get("decode_tok_s"), d.get("encode_ms"), d.get("phases_ms"), d.get("parity", {}).get("mismatches"), d.get("host_syncs_in_wavefront"), (d.get("replica") or {}).get("value"))
    except Exception as ex:
        print(f, "FAILED", ex)
PY
echo "== kernel trace: streaming 60 s"
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/$O/tr_stream" -o s -- \
    python "$GRAFT_REPO_ROOT/bench.py" --mode stream --seconds 60 --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> "$GRAFT_REPO_ROOT/$O/tr_stream.err" )
python tools/trace_summary.py $O/tr_stream --layer-of "k_skinny<1" --out $O/stream_trace_summary.txt; head -60 $O/stream_trace_summary.txt
echo "== kernel trace: headline"
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/$O/tr_head" -o h -- \
    python "$GRAFT_REPO_ROOT/bench.py" --steps 2 --warmup 0 --no-cpu-baseline --no-pmc > /dev/null 2> "$GRAFT_REPO_ROOT/$O/tr_head.err" )
python tools/trace_summary.py $O/tr_head --outlier k_dec_attn_fused --factor 4 --layer-of "k_dec_attn_fused" --out $O/head_trace_summary.txt; grep -A14 "outliers" $O/head_trace_summary.txt | head -70
rm -rf $O/tr_stream $O/tr_head
