#!/bin/bash
# One rocprofv3 --pmc pass of bench.py (3 steps) and the per-launch averages of the named kernel substring.
# usage (GPU box, repo root): bash tools/pmc_quick.sh <label> <kernel substring> "<counters>" [bench args...]
set -e
label=$1; kern=$2; ctrs=$3; shift 3
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmcq_$label
timeout -k 5 400 rocprofv3 --pmc $ctrs --output-format csv -d /tmp/pmcq_$label -o run -- python3 $root/bench.py --no-cpu-baseline --no-forward-rate --steps 3 --warmup 1 "$@" > /dev/null 2> /tmp/pmcq_$label.err
python3 - "$label" "$kern" $(find /tmp/pmcq_$label -name '*counter_collection.csv' | head -1) <<'PY'
import csv, sys, collections
label, kern, path = sys.argv[1:4]
acc = collections.defaultdict(list)
for row in csv.DictReader(open(path)):
    if kern in row["Kernel_Name"]:
        acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
print("==", label, kern)
for k, v in sorted(acc.items()):
    print("   %-24s %.4g (avg of %d launches)" % (k, sum(v) / len(v), len(v)))
PY
