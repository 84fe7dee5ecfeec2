#!/usr/bin/env python3
"""Summarise a VOX_HIP_ENC_TL dump (per-workgroup stamps of the four GEMM launches of one few-rows encoder layer).
usage: enc_timeline.py dump.txt"""
import sys
import numpy as np
rows = np.loadtxt(sys.argv[1], comments="#")
names = {0: "qkv (k_skinny<QKV>)", 1: "wo (k_skinny<PARTIAL, f32 X>)", 2: "w1;w3 (k_skinny<SWIGLU>)", 3: "w2 (k_skinny<PARTIAL>)"}
q = lambda a: "min %.2f p50 %.2f p90 %.2f max %.2f" % (a.min(), *np.percentile(a, [50, 90]), a.max())
for k in range(4):
    r = rows[rows[:, 0] == k]
    if not len(r):
        continue
    st, en = r[:, 2], r[:, 3]
    base = st.min()
    print(f"{names[k]}: {len(r)} workgroups, first entry at {base:.2f} us, kernel span {en.max() - base:.2f} us")
    print("   entry     ", q(st - base))
    for i, nm in enumerate(["loads issued", "chunk 0 done", "compute done", "tiles in LDS", "reduced"]):
        c = r[:, 6 + i]
        print(f"   {nm:13s}", q(c[c >= 0] - base))
    print("   exit      ", q(en - base))
