#!/bin/bash
# rocprofv3 --pmc passes (one per counter group, no tracing) of a short bench.py run; prints the prune kernels' per-wave
# averages.  usage (GPU box, repo root): bash tools/pmc_groups.sh "CTR1 CTR2 ..." "CTR3 ..." -- [bench.py args]
root=${GRAFT_REPO_ROOT:-$(pwd)}
groups=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do groups+=("$1"); shift; done
[ "${1:-}" = "--" ] && shift
cd /tmp && export TMPDIR=/tmp
i=0
for g in "${groups[@]}"; do
  rm -rf /tmp/pmcg_$i
  rocprofv3 --pmc $g --output-format csv -d /tmp/pmcg_$i -o run -- \
    python3 $root/bench.py --no-cpu-baseline --no-forward-rate --no-check --steps 2 --warmup 1 "$@" > /dev/null 2> /tmp/pmcg_$i.err
  python3 - /tmp/pmcg_$i <<'PY'
import csv, glob, collections, sys
fs = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)
if not fs:
    print('no counters collected in', sys.argv[1]); sys.exit(0)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for r in csv.DictReader(open(fs[0])):
    k = r['Kernel_Name'][:40]
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Counter_Name'] == list(acc[k])[0]: calls[k] += 1
for k, v in acc.items():
    if 'prune' in k:
        print(k, 'launches', calls[k], ' '.join('%s=%.4g' % (c, v[c] / calls[k] / 393216.0) for c in sorted(v)), '(per wave of 393216)')
PY
  i=$((i+1))
done
