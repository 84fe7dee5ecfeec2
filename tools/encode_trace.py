#!/usr/bin/env python3
"""Where a batch pass's time in front of the decode loop goes: from a rocprofv3 --kernel-trace CSV, the LAST window that starts at a
k_mel_frames dispatch and ends at the first k_dec_stack dispatch behind it - per kernel: calls, total us, and the idle gaps between
consecutive dispatches (total, and the largest ones with their neighbours).
usage: encode_trace.py <dir-or-csv> [--out file]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from collections import defaultdict
from trace_summary import load, short

def main():
    rows, path = load(sys.argv[1])
    out = open(sys.argv[sys.argv.index("--out") + 1], "w") if "--out" in sys.argv else sys.stdout
    P = lambda *x: print(*x, file=out)
    first_stack = [i for i, r in enumerate(rows) if "k_dec_stack" in r[2] and (i == 0 or rows[i - 1][0] < r[0] - 0) and not any("k_dec_stack" in rows[j][2] for j in range(max(0, i - 3), i))]
    if not first_stack: raise SystemExit("no k_dec_stack dispatch")
    end = first_stack[-1]
    start = max(i for i in range(end) if "k_mel_frames" in rows[i][2] and not any("k_mel_frames" in rows[j][2] for j in range(max(0, i - 2), i)))
    # the mel kernel may be dispatched several times per pass: go back to the first one of the run of mel dispatches
    w = rows[start:end + 1]
    span = (w[-1][0] - w[0][0]) / 1e3
    P(f"# {path}\n# window: dispatch {start} .. {end} ({len(w)} dispatches), {span:.1f} us from the first mel kernel's start to the first k_dec_stack's start")
    by = defaultdict(lambda: [0, 0.0]); gaps = []
    for i, (st, en, n, q) in enumerate(w[:-1]):
        by[n][0] += 1; by[n][1] += (en - st) / 1e3
        gaps.append(((w[i + 1][0] - en) / 1e3, short(n, 50), short(w[i + 1][2], 50)))
    busy = sum(v[1] for v in by.values())
    P(f"# busy {busy:.1f} us, idle between dispatches {sum(g[0] for g in gaps if g[0] > 0):.1f} us")
    for n, (c, t) in sorted(by.items(), key=lambda kv: -kv[1][1]):
        P(f"{t:10.1f} us {c:5d} x {t / c:8.2f}  {short(n, 110)}")
    P("# largest gaps (us: after -> before)")
    for g in sorted(gaps, reverse=True)[:12]: P(f"{g[0]:9.1f}  {g[1]} -> {g[2]}")

if __name__ == "__main__":
    main()
