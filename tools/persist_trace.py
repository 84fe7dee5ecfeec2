#!/usr/bin/env python3
"""Where does a phase of the persistent decode kernel spend its time?  Runs a short clip with
VOX_HIP_PERSIST=1 and VOX_HIP_PERSIST_TRACE set, then prints the mean gap between the control
wave's timestamps (100 MHz wall clock) for blocks 0 and 131.  GPU box only."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
out = os.path.join(ROOT, "gpurun_out", "persist_trace.bin")
os.makedirs(os.path.dirname(out), exist_ok=True)
os.environ["VOX_HIP_PERSIST"] = "1"
os.environ["VOX_HIP_PERSIST_TRACE"] = out

import voxtral_c_amd as v            # noqa: E402
from audio_util import synth_speech  # noqa: E402
from conftest import model_dir       # noqa: E402

with v.Model(model_dir("full")) as m:
    r = m.transcribe(synth_speech(6.0, 7))
    print("tokens:", len(r["tokens"]), "timing:", m.timing())

t = np.fromfile(out, dtype=np.uint64).reshape(2, 4096).astype(np.int64)
B = ["released", "arrived", "all-arrived", "acquired"]
names0 = (["P1 staged", "P1 streamed"] + B + ["P2 q staged", "P2 streamed"] + B + ["P3 merged", "P3 streamed"] + B
          + ["P4 staged", "P4 streamed"] + B + ["P5 staged", "P5 streamed"] + B)
names1 = names0[:6] + B + names0[12:]
for bi, names in ((0, names0), (1, names1)):
    row = t[bi]
    n = int((row > 0).sum())
    per_layer = len(names)
    per_step = per_layer * 26 + 4
    print(f"\nblock {'0' if bi == 0 else '131'}: {n} marks, {per_layer} per layer")
    if n < per_step * 2:
        print("  too few marks")
        continue
    # second step, layers 1..25
    base = per_step
    d = np.zeros(per_layer)
    cnt = 0
    for l in range(1, 26):
        seg = row[base + l * per_layer - 1: base + (l + 1) * per_layer]
        d += np.diff(seg)
        cnt += 1
    d = d / cnt / 100.0     # µs
    for nm, us in zip(names, d):
        print(f"  -> {nm:14s} {us:7.2f} us")
    print(f"  layer total {d.sum():.1f} us; step total {(row[base + per_step] - row[base]) / 100.0:.1f} us")
