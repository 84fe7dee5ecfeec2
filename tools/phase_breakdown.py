#!/usr/bin/env python3
"""From a rocprofv3 kernel trace of tools/one_transcribe.py: what runs before the first decoder step of a transcription
(encoder passes + prefill) and what the passes cost, split at the k_mel_frames launches.  Usage: phase_breakdown.py <dir>"""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
# phases: a new phase starts at every k_mel_frames launch and at the first k_dec_attn_fused after a non-decode kernel
phases, cur = [], []
dec = lambda r: any(k in r["Kernel_Name"] for k in ("k_dec_attn_fused", "k_gemv_w13x", "k_gemv_w2x", "k_gemv<", "k_argmax"))
for r in rows:
    if "k_mel_frames" in r["Kernel_Name"] or (cur and dec(r) != dec(cur[-1])):
        if cur: phases.append(cur)
        cur = []
    cur.append(r)
if cur: phases.append(cur)
for ph in phases:
    if len(ph) < 20: continue
    span = (int(ph[-1]["End_Timestamp"]) - int(ph[0]["Start_Timestamp"])) / 1e3
    busy = sum(dur(r) for r in ph)
    kind = "decode" if dec(ph[0]) else "encode/prefill"
    print(f"{kind}: {len(ph)} launches, span {span/1e3:.2f} ms, kernel-busy {busy/1e3:.2f} ms")
    if kind == "decode": continue
    d = collections.defaultdict(lambda: [0, 0.0])
    for r in ph:
        k = (r["Kernel_Name"][:40], r.get("Grid_Size_X", ""), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""))
        d[k][0] += 1; d[k][1] += dur(r)
    for k, v in sorted(d.items(), key=lambda kv: -kv[1][1])[:12]:
        print("    ", k, v[0], "%.0f us" % v[1])
