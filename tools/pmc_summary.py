#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel trace and/or --pmc counter collection) per kernel name.

    python tools/pmc_summary.py <dir> [--match "spmv|onepass"]   (any of the |-separated substrings)
"""
import argparse
import collections
import csv
import glob
import os


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--match", default="")
    a = ap.parse_args()
    for path in sorted(glob.glob(os.path.join(a.dir, "**", "*counter_collection.csv"), recursive=True)):
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        cnt = collections.defaultdict(lambda: collections.defaultdict(int))
        for row in csv.DictReader(open(path)):
            k = row.get("Kernel_Name", "")
            if a.match and not any(m in k for m in a.match.split("|")):
                continue
            k = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
            cnt[k][row["Counter_Name"]] += 1
        print("==", path)
        for k, d in agg.items():
            print(k)
            for c, v in sorted(d.items()):
                print(f"    {c:28s} {v / cnt[k][c]:16.1f}  (avg of {cnt[k][c]} dispatches)")
    for path in sorted(glob.glob(os.path.join(a.dir, "**", "*kernel_trace.csv"), recursive=True)):
        dur = collections.defaultdict(list)
        for row in csv.DictReader(open(path)):
            k = row.get("Kernel_Name", "")
            if a.match and not any(m in k for m in a.match.split("|")):
                continue
            dur[k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
        print("==", path)
        for k, v in dur.items():
            v.sort()
            print(f"{k:60s} n={len(v):4d} avg={sum(v) / len(v):9.2f}us med={v[len(v) // 2]:9.2f}us min={v[0]:9.2f}us")


if __name__ == "__main__":
    main()
