#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step','decode_tok_s')}, {k:d['roofline'][k] for k in ('achieved','frac','avg_us_per_launch','traffic')})"
HSA_ENABLE_IPC_MODE_LEGACY=0 VOX_FORCE_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 1 --warmup 1 2>gpurun_out/dist.err | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step','decode_tok_s','n_gpus')}, d['roofline'].get('frac'), d['roofline'].get('error'))"
tail -3 gpurun_out/dist.err
