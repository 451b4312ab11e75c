#!/usr/bin/env python3
"""Average rocprofv3 --pmc counter values per kernel dispatch.

usage: pmc_summary.py [--json OUT] [--note TEXT] DIR [DIR ...] [--filter NAME]

Each DIR is the output directory of one `rocprofv3 --pmc ... --output-format csv` pass (counters that
cannot share a pass are collected in separate passes and merged here).
"""
import argparse
import csv
import glob
import json
import os
from collections import defaultdict


def short(name):
    name = name.split("(")[0]
    return name[5:] if name.startswith("void ") else name


def collect(dirs, flt):
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for root in dirs:
        for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
            with open(path) as f:
                for r in csv.DictReader(f):
                    name = short(r["Kernel_Name"])
                    if flt and flt not in name:
                        continue
                    a = acc[name][r["Counter_Name"]]
                    a[0] += float(r["Counter_Value"])
                    a[1] += 1
    return {k: {c: tot / cnt for c, (tot, cnt) in sorted(v.items())} for k, v in sorted(acc.items())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dirs", nargs="+")
    ap.add_argument("--filter", default="")
    ap.add_argument("--json")
    ap.add_argument("--note", default="")
    ap.add_argument("--evals-per-launch", type=int, default=None,
                    help="evaluations per launch group of the profiled command (bench.py matches it before it "
                         "quotes these counters per launch)")
    args = ap.parse_args()
    res = collect(args.dirs, args.filter)
    if args.json:
        out = {"_note": args.note}
        if args.evals_per_launch is not None:
            out["_evals_per_launch"] = args.evals_per_launch
        out.update(res)
        with open(args.json, "w") as f:
            json.dump(out, f, indent=1)
    for k, ctrs in res.items():
        print(k)
        for c, v in ctrs.items():
            print("   %-28s %16.1f" % (c, v))


if __name__ == "__main__":
    main()
