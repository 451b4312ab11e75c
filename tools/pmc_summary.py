#!/usr/bin/env python3
"""Average rocprofv3 --pmc counter values per kernel dispatch.

usage: pmc_summary.py <dir with *counter_collection.csv> [name-filter]
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    root = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                name = r["Kernel_Name"]
                if flt and flt not in name:
                    continue
                a = acc[name.split("(")[0][-44:]][r["Counter_Name"]]
                a[0] += float(r["Counter_Value"])
                a[1] += 1
    for k, ctrs in sorted(acc.items()):
        print(k)
        for c, (tot, cnt) in sorted(ctrs.items()):
            print(f"   {c:28s} {tot / cnt:16.1f}  (n={cnt})")


if __name__ == "__main__":
    main()
