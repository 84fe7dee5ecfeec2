#!/usr/bin/env python3
"""Per-workgroup timeline of the fused decode step at a given KV length (VOX_HIP_FUSE_TL must name the dump file)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import voxtral_c_amd as v
from conftest import model_dir
with v.Model(model_dir("full"), weights=os.environ.get("SWEEP_WEIGHTS", "bf16")) as m:
    s = m.time_decoder_step(20, int(sys.argv[1]))
    print("decoder step at kv %d: %.3f ms" % (int(sys.argv[1]), s * 1e3))
