#!/usr/bin/env python3
"""Summarise the stack kernel's VOX_HIP_ENC_TL dump (<file>.stack: per-workgroup stamps of the mid-stack layer's seven phases).
usage: enc_stack_timeline.py dump.txt.stack"""
import sys
import numpy as np
rows = np.loadtxt(sys.argv[1], comments="#")
if rows.ndim == 1:
    rows = rows[None]
names = ["P1 qkv", "P2 attention", "P3 wo", "F3 finish", "P4 w1;w3", "P5 w2", "F5 finish"]
q = lambda a: "min %7.2f  p50 %7.2f  p90 %7.2f  max %7.2f" % (a.min(), *np.percentile(a, [50, 90]), a.max())
st = rows[:, 5:]
print(f"{len(rows)} workgroups; XCDs seen: {sorted(set(rows[:, 3].astype(int)))}; times in us from the first workgroup's entry into P1")
prev_done = None
for p, nm in enumerate(names):
    w, d = st[:, 2 * p], st[:, 2 * p + 1]
    print(f"{nm:13s} hand-off seen   {q(w)}")
    print(f"{'':13s} body done       {q(d)}    body (p50) {np.median(d - w):6.2f}")
    if prev_done is not None:
        print(f"{'':13s} [slowest producer done -> first / median / last consumer released: {w.min() - prev_done.max():5.2f} / {np.median(w) - prev_done.max():5.2f} / {w.max() - prev_done.max():5.2f}]")
    prev_done = d
print(f"layer span (first P1 release -> last F5 body done): {st[:, -1].max() - st[:, 0].min():.2f} us")
