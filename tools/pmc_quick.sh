#!/bin/bash
# One rocprofv3 --pmc pass (instruction counts of every kernel) of a short bench.py run; prints K1's per-wave counts.
# usage (GPU box, repo root): bash tools/pmc_quick.sh [bench.py args]
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmcq
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d /tmp/pmcq -o run -- \
  python3 $root/bench.py --no-cpu-baseline --no-forward-rate --no-check --steps 2 --warmup 1 "$@" > /dev/null 2> /tmp/pmcq.err
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/pmcq/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'][:48]
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
for k, v in acc.items():
    if 'prune' in k or 'schedule' in k:
        w = v['SQ_WAVES']
        print(k, ' '.join('%s=%.0f' % (c.replace('SQ_INSTS_', ''), v[c] / w) for c in sorted(v) if c != 'SQ_WAVES'), 'per wave')
PY
