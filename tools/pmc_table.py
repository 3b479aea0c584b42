#!/usr/bin/env python
"""Pivot rocprofv3 counter_collection.csv files into one row per kernel (mean over dispatches).
usage: pmc_table.py dir_with_passes [name_filter]"""
import csv, glob, os, sys, collections, re
root = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else "mcq::"
tab = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if flt not in n: continue
        n = re.sub(r"\(.*", "", n).replace("mcq::", "")
        tab[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[n].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for n in sorted(tab):
    print(f"== {n}  (avg dur {sum(dur[n])/len(dur[n])/1e3:.1f} us over {len(dur[n])} samples)")
    for c in sorted(tab[n]):
        v = tab[n][c]
        print(f"   {c:32s} {sum(v)/len(v):18.1f}")
