#!/usr/bin/env python3
"""Per-stage timeline of the overlapped decode chain (VOX_HIP_PDL=1): timestamps written by thread 0
of the first and last block of every k_gemv3 launch of one decode step.  GPU box only."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
out = os.path.join(ROOT, "gpurun_out", "pdl_trace.bin")
os.makedirs(os.path.dirname(out), exist_ok=True)
os.environ["VOX_HIP_PDL"] = "1"; os.environ["VOX_HIP_PDL_TRACE"] = out
import voxtral_c_amd as v
from conftest import model_dir
with v.Model(model_dir("full")) as m:
    print("ms/token:", m.time_decoder_step(10, 232) * 1e3)
t = np.fromfile(out, dtype=np.uint64).reshape(160, 2, 16).astype(np.int64)
names = ["qkv", "attn", "wo", "swiglu", "w2"]
stages = ["weights issued", "flag+acquire", "x landed", "prologue", "dots", "stores issued", "drained+signal"]
for blk in (0, 1):
    print("block", "first" if blk == 0 else "last")
    for kind in range(5):
        if kind == 1: continue
        d = []
        for l in range(2, 24):
            r = t[l * 5 + kind, blk]
            if r[7] > 0: d.append(np.diff(r[:8]) / 100.0)
        if d:
            d = np.mean(d, axis=0)
            print(f"  {names[kind]:7s} " + "  ".join(f"{s}={x:6.2f}" for s, x in zip(stages, d)) + f"  total={d.sum():6.2f}")
# cross-kernel: time from predecessor's signal (probe 7 of last block) to this kernel's flag seen (probe 2)
for kind in (0, 2, 3, 4):
    d = []
    for l in range(2, 24):
        k = l * 5 + kind
        if kind == 2: continue
        prev = t[k - 1]; cur = t[k]
        if prev[1, 7] > 0 and cur[0, 2] > 0: d.append((cur[0, 2] - max(prev[0, 7], prev[1, 7])) / 100.0)
    if d: print(f"  {names[kind]}: flag seen - predecessor signalled = {np.mean(d):.2f} us")
