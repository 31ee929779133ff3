#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel trace / counter collection) per kernel.
usage: pmc_summary.py DIR [DIR...]   -- prints, per kernel name, the dispatches with the largest grid
(the full-size bench launches) averaged, one line per counter.  Development tool."""
import collections
import csv
import glob
import os
import sys


def main(dirs):
    for d in dirs:
        for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
            rows = list(csv.DictReader(open(f)))
            per = collections.defaultdict(lambda: collections.defaultdict(dict))  # kernel -> dispatch -> counter -> val
            meta = {}
            for r in rows:
                k = r["Kernel_Name"].split("(")[0]
                per[k][r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
                meta[(k, r["Dispatch_Id"])] = (int(r["Grid_Size"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]),
                                               r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"])
            print(f"== {f}")
            for k, disp in per.items():
                gmax = max(meta[(k, d_)][0] for d_ in disp)
                big = [d_ for d_ in disp if meta[(k, d_)][0] == gmax]
                dur = sum(meta[(k, d_)][1] for d_ in big) / len(big)
                m = meta[(k, big[0])]
                print(f"  {k}  [n={len(big)} grid={gmax} dur_ns={dur:.0f} vgpr={m[2]} agpr={m[3]} sgpr={m[4]} lds={m[5]} scratch={m[6]}]")
                names = sorted({c for d_ in big for c in disp[d_]})
                for c in names:
                    v = [disp[d_][c] for d_ in big if c in disp[d_]]
                    print(f"      {c:28s} {sum(v)/len(v):18.1f}")
        for f in sorted(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)):
            print(f"== {f}")
            print(open(f).read())


if __name__ == "__main__":
    main(sys.argv[1:] or ["."])
