#!/bin/bash
# kernel-trace stats of one bench run; output: gpurun_out/prof_stats.csv
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o run -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 10 > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof_bench.err
f=$(find /tmp/prof -name '*kernel_stats.csv' | head -1)
cp "$f" $GRAFT_REPO_ROOT/gpurun_out/prof_stats.csv
cut -c1-150 $GRAFT_REPO_ROOT/gpurun_out/prof_stats.csv
