#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV (one row per dispatch) beyond what --stats gives:
  * per kernel: calls, mean / p50 / p99 / max duration;
  * the gaps between consecutive dispatches of the busiest queue (kernel boundary cost as the profiler sees it);
  * outliers: dispatches of kernels matching --outlier whose duration exceeds --factor x their median, each with the
    dispatches before and after it on ANY queue (what else was on the GPU);
  * with --layer-of KERNEL: the mean duration of every kernel between two consecutive dispatches of KERNEL (one
    "layer" of a repeating launch pattern) and the mean gap in front of each.
usage: trace_summary.py <dir-or-csv> [--outlier SUBSTR] [--factor 5] [--layer-of SUBSTR] [--out file]"""
import argparse
import csv
import glob
import os
import sys
from collections import defaultdict

import numpy as np


def load(path):
    if os.path.isdir(path):
        cands = glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True)
        if not cands:
            raise SystemExit(f"no *kernel_trace.csv under {path}")
        path = max(cands, key=os.path.getsize)
    rows = []
    with open(path, newline="") as fh:
        rd = csv.DictReader(fh)
        cols = rd.fieldnames
        name_c = next(c for c in cols if c.lower() in ("kernel_name", "name"))
        st_c = next(c for c in cols if "start" in c.lower())
        en_c = next(c for c in cols if "end" in c.lower())
        q_c = next((c for c in cols if c.lower().startswith("queue")), None)
        for r in rd:
            rows.append((int(r[st_c]), int(r[en_c]), r[name_c], r[q_c] if q_c else "0"))
    rows.sort()
    return rows, path


def short(n, w=70):
    n = n.replace("void ", "").replace("vox::", "")
    return n if len(n) <= w else n[:w - 3] + "..."


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("path")
    ap.add_argument("--outlier", default=None)
    ap.add_argument("--factor", type=float, default=5.0)
    ap.add_argument("--layer-of", default=None)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    rows, path = load(a.path)
    out = open(a.out, "w") if a.out else sys.stdout
    P = lambda *x: print(*x, file=out)
    P(f"# {path}: {len(rows)} dispatches")
    by = defaultdict(list)
    for st, en, n, q in rows:
        by[n].append((en - st) / 1e3)
    P("# per kernel (us): calls mean p50 p99 max  name")
    for n, d in sorted(by.items(), key=lambda kv: -sum(kv[1])):
        d = np.asarray(d)
        P(f"{len(d):7d} {d.mean():9.2f} {np.median(d):9.2f} {np.percentile(d, 99):9.2f} {d.max():10.2f}  {short(n)}")
    # gaps on the busiest queue
    qs = defaultdict(int)
    for r in rows:
        qs[r[3]] += 1
    mainq = max(qs, key=qs.get)
    mq = [r for r in rows if r[3] == mainq]
    gaps = np.asarray([(mq[i + 1][0] - mq[i][1]) / 1e3 for i in range(len(mq) - 1)])
    small = gaps[(gaps > -50) & (gaps < 50)]
    if len(small):
        P(f"# gaps between consecutive dispatches on queue {mainq} (|gap| < 50 us, n={len(small)}): mean {small.mean():.2f} p50 {np.median(small):.2f} "
          f"p90 {np.percentile(small, 90):.2f} us; gaps >= 50 us: {int((gaps >= 50).sum())}")
    if a.layer_of:
        idx = [i for i, r in enumerate(mq) if a.layer_of in r[2]]
        seqs = defaultdict(lambda: defaultdict(list))
        for i0, i1 in zip(idx, idx[1:]):
            key = tuple(short(mq[j][2], 40) for j in range(i0, i1))
            for k, j in enumerate(range(i0, i1)):
                seqs[key][k].append(((mq[j][1] - mq[j][0]) / 1e3, (mq[j][0] - mq[j - 1][1]) / 1e3 if j > 0 else 0.0,
                                     (mq[i1][0] - mq[i0][0]) / 1e3))
        P(f"# repeating launch patterns between consecutive '{a.layer_of}' dispatches (most frequent first)")
        for key, pos in sorted(seqs.items(), key=lambda kv: -len(kv[1][0]))[:3]:
            n = len(pos[0])
            per = np.mean([x[2] for x in pos[0]])
            P(f"## pattern seen {n} times, {len(key)} launches, start-to-start {per:.2f} us")
            for k, nm in enumerate(key):
                d = np.asarray([x[0] for x in pos[k]]); g = np.asarray([x[1] for x in pos[k]])
                P(f"   {k:2d} dur {d.mean():7.2f} (p50 {np.median(d):7.2f})  gap before {np.median(g):6.2f}  {nm}")
    if a.outlier:
        sel = [(i, r) for i, r in enumerate(rows) if a.outlier in r[2]]
        if sel:
            med = np.median([(r[1] - r[0]) / 1e3 for _, r in sel])
            bad = [(i, r) for i, r in sel if (r[1] - r[0]) / 1e3 > a.factor * med]
            P(f"# outliers of '{a.outlier}': median {med:.2f} us, {len(bad)} dispatches above {a.factor:g} x median")
            for i, r in bad[:12]:
                P(f"## dispatch #{i}: {(r[1] - r[0]) / 1e3:.2f} us on queue {r[3]}, start t={r[0]}")
                for j in range(max(0, i - 6), min(len(rows), i + 5)):
                    s = rows[j]
                    P(f"   {'>>' if j == i else '  '} q{s[3]} start {(s[0] - r[0]) / 1e3:10.2f} end {(s[1] - r[0]) / 1e3:10.2f} dur {(s[1] - s[0]) / 1e3:9.2f}  {short(s[2], 60)}")


if __name__ == "__main__":
    main()
