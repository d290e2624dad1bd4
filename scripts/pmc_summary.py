#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter rows per kernel: python scripts/pmc_summary.py <dir with *_counter_collection.csv> [out.json]"""
import csv, glob, json, os, sys
from collections import defaultdict

def main():
    d = sys.argv[1]
    acc = defaultdict(lambda: defaultdict(float))
    launches = defaultdict(set)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"][:60]
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            launches[k].add(row.get("Dispatch_Id"))
    out = {k: dict(v, launches=len(launches[k])) for k, v in acc.items()}
    for k, v in sorted(out.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
        print(k)
        wc = v.get("SQ_WAVE_CYCLES", 0)
        for c, x in sorted(v.items()):
            print(f"   {c:24s} {x:16.0f}" + (f"  {100 * x / wc:5.1f}% of wave cycles" if wc and c.startswith("SQ_") and c != "SQ_WAVE_CYCLES" and "INSTS" not in c else ""))
    if len(sys.argv) > 2:
        json.dump(out, open(sys.argv[2], "w"), indent=1)

if __name__ == "__main__":
    main()
