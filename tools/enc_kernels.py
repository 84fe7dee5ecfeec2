#!/usr/bin/env python3
"""Per-kernel (name x grid) averages from a rocprofv3 --kernel-trace csv directory.  Usage: enc_kernels.py <dir> [top]"""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    key = (r["Kernel_Name"][:44], r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""))
    d[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[: int(sys.argv[2]) if len(sys.argv) > 2 else 10]:
    print(k, len(v), "avg %.1f us total %.2f ms" % (sum(v) / len(v), sum(v) / 1e3))
