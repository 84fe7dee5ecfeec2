#!/bin/bash
# Round 6: timelines of the stack kernel, tree build against ab_base (VOX_LIB_DIR), kv 232
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6g; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
python -c "import sys; sys.path.insert(0,'tests'); from conftest import model_dir; print(model_dir('full'))" > /dev/null 2>&1
VOX_HIP_FUSE_TL=$O/tl_tree.txt python tools/fuse_tl_kv.py 232 > $O/tree_kv232.txt 2>&1; python tools/fuse_timeline.py $O/tl_tree.txt >> $O/tree_kv232.txt 2>&1
VOX_LIB_DIR=$(realpath ab_base) VOX_HIP_FUSE_TL=$O/tl_base.txt python tools/fuse_tl_kv.py 232 > $O/base_kv232.txt 2>&1; python tools/fuse_timeline.py $O/tl_base.txt >> $O/base_kv232.txt 2>&1
head -30 $O/tree_kv232.txt
