#!/usr/bin/env python3
"""Which workgroups of the three decode launches finish last?  usage: tl_slowest.py dump.txt (VOX_HIP_FUSE_TL dump)"""
import sys
import numpy as np
rows = np.loadtxt(sys.argv[1], comments="#")
for k, name in ((0, "k_dec_attn_fused"), (1, "k_gemv_w13x"), (2, "k_gemv_w2x")):
    r = rows[rows[:, 0] == k]
    if not len(r):
        continue
    order = np.argsort(-r[:, 3])[:8]
    print(name, "last exits (block, xcc, exit us):", [(int(r[i, 1]), int(r[i, 4]), round(float(r[i, 3]), 2)) for i in order],
          "| median exit %.2f" % np.median(r[:, 3]))
