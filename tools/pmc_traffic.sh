#!/bin/bash
# One rocprofv3 --pmc pass (FETCH_SIZE, WRITE_SIZE) of a short bench.py run; prints per-launch KiB of the K1 kernels.
# usage (GPU box, repo root): [VAR=... ] bash tools/pmc_traffic.sh [bench.py args]
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmct
rocprofv3 --pmc FETCH_SIZE WRITE_SIZE --output-format csv -d /tmp/pmct -o run -- \
  python3 $root/bench.py --no-cpu-baseline --no-forward-rate --no-check --no-extras --steps 2 --warmup 1 "$@" > /dev/null 2> /tmp/pmct.err
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/pmct/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'][:48]
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    n[k].add(r['Dispatch_Id'])
for k, v in acc.items():
    if 'prune' in k or 'schedule' in k:
        print(k, ' '.join('%s=%.0f KiB' % (c, v[c] / len(n[k])) for c in sorted(v)), 'per launch (%d launches)' % len(n[k]))
PY
