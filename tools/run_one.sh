#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
for i in 1 2; do
t0=$(date +%s.%N)
timeout 600 env HSA_ENABLE_IPC_MODE_LEGACY=0 VOX_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 2951$i bench.py --gpus 1 --steps 1 --warmup 0 --preset small --seconds 20 2>gpurun_out/dist$i.err | tail -1 | cut -c1-160
t1=$(date +%s.%N); echo "elapsed $(echo "$t1 - $t0" | bc) s"
done
