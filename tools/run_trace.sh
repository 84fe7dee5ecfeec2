cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; timeout 600 python tools/persist_trace.py 2>&1 | tail -80
