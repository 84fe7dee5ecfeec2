#!/usr/bin/env python3
"""Summarise a VOX_HIP_FUSE_TL dump (per-workgroup start/end stamps of layer 13's blocks in a decoder step):
start skew, tails, per-XCD finishing times, the gaps between launches.  Usage: fuse_timeline.py dump.txt
Record 0 = the attention block (k_dec_attn_fused, or df_attn12_body inside k_ffn_attn12: then its "entry" is the moment a workgroup
leaves the previous layer's FFN block and there is no launch boundary in front of it), record 1 = the FFN block (k_ffn_fused / ffn_body
inside k_ffn_attn12, or k_gemv_w13x), record 2 = the separate W2 launch where there is one.  Times in us from the first attention entry."""
import sys
import numpy as np

rows = np.loadtxt(sys.argv[1], comments="#")
names = {0: "attention block (k_dec_attn_fused / df_attn12_body)", 1: "FFN block (k_ffn_fused / ffn_body / k_gemv_w13x)", 2: "W2 launch"}
prev_end = None
for k in (0, 1, 2):
    r = rows[rows[:, 0] == k]
    if not len(r):
        continue
    st, en, xcc = r[:, 2], r[:, 3], r[:, 4].astype(int)
    q = lambda a: "min %.2f p10 %.2f p50 %.2f p90 %.2f max %.2f" % (a.min(), *np.percentile(a, [10, 50, 90]), a.max())
    print(f"{names[k]}: {len(r)} workgroups")
    if prev_end is not None:
        print(f"   gap: previous launch's last exit -> first entry {st.min() - prev_end:.2f} us")
    print("   entry  ", q(st))
    print("   exit   ", q(en))
    print("   inside ", q(en - st))
    print("   per XCD (n, mean entry, mean exit, max exit):",
          " ".join(f"[{x}: {int((xcc == x).sum())} {st[xcc == x].mean():.1f} {en[xcc == x].mean():.1f} {en[xcc == x].max():.1f}]" for x in sorted(set(xcc))))
    same = np.mean((r[:, 1].astype(int) % 8) == xcc)
    print(f"   blockIdx %% 8 == XCC_ID for {100 * same:.0f}% of the workgroups")
    prev_end = en.max()

# phase stamps of k_dec_attn_fused per group (group = block % 8, member j = block // 8; members j < nsplit run attention)
r = rows[rows[:, 0] == 0]
if r.shape[1] >= 19:
    names = ["entry", "issued", "x landed", "norm", "dots done", "published", "sweep1", "attn", "sweep2", "wo landed", "wo done", "dma issued", "barrier"]
    st = r[:, 6:19]
    print("k_dec_attn_fused phase stamps over all workgroups (us): min / p50 / max")
    for i, n in enumerate(names):
        c = st[:, i]
        print(f"   {i:2d} {n:10s} {c.min():6.2f} {np.median(c):6.2f} {c.max():6.2f}")
    blk = r[:, 1].astype(int)
    print("   per group: last publish | sweep1 done (min..max) | att members' attention end (max) | sweep2 done (min..max) | exit max")
    for g in range(8):
        m = (blk % 8) == g
        att = m & (st[:, 7] - st[:, 6] > 0.3)
        print(f"   g{g}: {st[m, 5].max():6.2f} | {st[m, 6].min():6.2f}..{st[m, 6].max():6.2f} | {st[att, 7].max() if att.any() else -1:6.2f} ({int(att.sum())} members) | "
              f"{st[m, 8].min():6.2f}..{st[m, 8].max():6.2f} | {r[m, 3].max():6.2f}")
    print("   attention members of groups 0 and 1 (j: dots done, published, sweep1 done, [first tile in, scores done,] attention end | the group's last publish):")
    for g in (0, 1):
        m = (blk % 8) == g
        for i in np.nonzero(m & (st[:, 7] - st[:, 6] > 0.3))[0]:
            print(f"   g{g} j{blk[i] // 8}: {st[i, 4]:6.2f} {st[i, 5]:6.2f} {st[i, 6]:6.2f} [{st[i, 11]:6.2f} {st[i, 12]:6.2f}] {st[i, 7]:6.2f} | {st[m, 5].max():6.2f}")
r = rows[rows[:, 0] == 1]
if r.shape[1] >= 14 and (r[:, 13] > 0).any():        # k_ffn_fused (round 4): 8 stamps of wave 0
    st = r[:, 6:14]
    print("k_ffn_fused phase stamps, wave 0 of every workgroup (us): min / p50 / max")
    for i, n in enumerate(["entry", "round 0 issued", "prologue in", "normed", "h published (W2 row queued)", "all waves there", "h swept", "done"]):
        c = st[:, i]
        print(f"   {i:2d} {n:28s} {c.min():6.2f} {np.median(c):6.2f} {c.max():6.2f}")
elif r.shape[1] >= 11:
    st = r[:, 6:11]
    print("k_gemv_w13x phase stamps (us): min / p50 / max")
    for i, n in enumerate(["entry", "round 0 issued", "prologue in", "normed", "done (wave 0)"]):
        c = st[:, i]
        print(f"   {i:2d} {n:15s} {c.min():6.2f} {np.median(c):6.2f} {c.max():6.2f}")
