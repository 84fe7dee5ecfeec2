#!/usr/bin/env python3
"""MFMA utilisation per kernel from a `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace` run (tools/run_pmc_enc.sh).
Counters are summed over the chip: GRBM_GUI_ACTIVE over the 8 XCDs, SQ_VALU_MFMA_BUSY_CYCLES over
the 1024 SIMDs, so  utilisation = MFMA_BUSY / (1024 * GUI_ACTIVE / 8).  MOPS are 512-FLOP units."""
import collections
import csv
import glob
import json
import os
import sys


def main(src, dst):
    f = glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True)[0]
    disp = collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        d = disp[r["Dispatch_Id"]]
        d["name"] = r["Kernel_Name"].split("(")[0]
        d["grid"] = int(r["Grid_Size"])
        d["ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        d[r["Counter_Name"]] = float(r["Counter_Value"])
    agg = collections.defaultdict(list)
    for d in disp.values():
        if d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) > 1e6:
            agg[(d["name"], d["grid"])].append(d)
    out = []
    for (name, grid), v in sorted(agg.items(), key=lambda kv: -sum(x["ns"] for x in kv[1])):
        n = len(v)
        mean = lambda k: sum(x.get(k, 0.0) for x in v) / n
        busy, gui = mean("SQ_VALU_MFMA_BUSY_CYCLES"), mean("GRBM_GUI_ACTIVE")
        mops = mean("SQ_INSTS_VALU_MFMA_MOPS_BF16") + mean("SQ_INSTS_VALU_MFMA_MOPS_F32")
        us = mean("ns") / 1e3
        out.append({"kernel": name, "grid_threads": grid, "launches": n, "avg_us_under_pmc": round(us, 1),
                    "mfma_utilisation": round(busy / (1024 * gui / 8), 4),
                    "mfma_tflops": round(mops * 512 / (us * 1e-6) / 1e12, 1)})
        print(out[-1])
    json.dump({"source": "rocprofv3 --pmc (own pass, kernel-trace only), tools/pmc_encoder.py: 2 x 1664-row encoder chunks",
               "kernels": out}, open(dst, "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
