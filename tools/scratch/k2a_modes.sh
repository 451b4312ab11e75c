#!/bin/bash
cd /tmp && export TMPDIR=/tmp
for m in 0 1 2; do
  export LH_K2A_MODE=$m
  rm -rf /tmp/prof
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o run -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-check --steps 10 > /dev/null 2> /tmp/err.txt
  f=$(find /tmp/prof -name '*kernel_stats.csv' | head -1)
  echo "mode $m"; grep -E "emission|junction|prune" "$f" | awk -F, '{print substr($1,1,30), $(NF-4)}'
done
