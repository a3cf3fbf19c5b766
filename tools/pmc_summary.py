#!/usr/bin/env python3
"""Sums rocprofv3 --pmc counter_collection.csv files per kernel dispatch (average over dispatches of
the kernels whose name contains `key`).  usage: pmc_summary.py key file.csv [file.csv ...]"""
import csv
import json
import sys
from collections import defaultdict


def main():
    key = sys.argv[1]
    out = {}
    for path in sys.argv[2:]:
        per = defaultdict(lambda: defaultdict(float))
        dur = {}
        for r in csv.DictReader(open(path)):
            if key not in r["Kernel_Name"]:
                continue
            per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
            dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        if not per:
            continue
        n = len(per)
        names = sorted({c for d in per.values() for c in d})
        for c in names:
            out[c] = sum(d.get(c, 0.0) for d in per.values()) / n
        out.setdefault("_dispatches", {})[path] = n
        out.setdefault("_avg_ns", {})[path] = sum(dur.values()) / n
    print(json.dumps(out, indent=1, sort_keys=True))


main()
