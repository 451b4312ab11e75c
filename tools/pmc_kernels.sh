#!/bin/bash
# usage: pmc.sh <filter> <group1> [<group2> ...]   each group = space-separated counters in quotes
cd /tmp && export TMPDIR=/tmp
flt=$1; shift
i=0
dirs=""
for grp in "$@"; do
  rm -rf /tmp/pmc$i
  rocprofv3 --pmc $grp --output-format csv -d /tmp/pmc$i -o run -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-check --steps 3 --warmup 1 > /dev/null 2> /tmp/pmc_err$i.txt || tail -5 /tmp/pmc_err$i.txt
  dirs="$dirs /tmp/pmc$i"
  i=$((i+1))
done
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $dirs --filter "$flt"
