#!/bin/bash
# Kernel times of bench.py under rocprofv3 --kernel-trace for a list of environment settings (no rebuild).
# usage: bash tools/exp_env.sh "label1:VAR=1 OTHER=2" "label2:" ...
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
for spec in "$@"; do
  label=${spec%%:*}
  envs=${spec#*:}
  rm -rf /tmp/prof_env_$label
  (cd /tmp && export TMPDIR=/tmp $envs && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_env_$label -o run -- \
    python3 $root/bench.py --no-cpu-baseline --no-forward-rate --steps 5 --warmup 2 $EXP_BENCH_ARGS > $root/gpurun_out/env_$label.json 2> /dev/null)
  echo "== $label ($envs)"
  python3 tools/kernel_stats.py "$(find /tmp/prof_env_$label -name '*kernel_stats.csv' | head -1)" | head -5
done
