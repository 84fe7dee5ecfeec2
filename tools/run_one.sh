#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step','decode_tok_s')}); print(d['roofline']['decode_step'])"
