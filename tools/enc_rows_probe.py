#!/usr/bin/env python3
"""Time of the encoder stack on an n-row chunk behind a full K/V window (vox_hip_time_encoder_rows: HIP events, resident weights).
usage: enc_rows_probe.py [rows,rows,..] [ctx_rows] [iters]   (env TAG labels the line; the engine's switches come from the environment)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import voxtral_c_amd as v
from conftest import model_dir
rows = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "25").split(",")]
ctx = int(sys.argv[2]) if len(sys.argv) > 2 else 750
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 30
v.hip.vox_hip_time_encoder_rows.restype = C.c_double
v.hip.vox_hip_time_encoder_rows.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
with v.Model(model_dir(os.environ.get("PRESET", "full"))) as m:
    out = []
    for n in rows:
        best = min(v.hip.vox_hip_time_encoder_rows(m.engine, n, ctx, iters) for _ in range(3))
        out.append((n, round(best * 1e6 / m.dims.enc_layers, 2)))
    print(os.environ.get("TAG", ""), "us per encoder layer by rows:", out, flush=True)
