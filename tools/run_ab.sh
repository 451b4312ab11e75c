#!/bin/bash
# K1 time of the round-2 tree (tmp_r2/, if present) and of variant libraries, back to back on one box
root=${GRAFT_REPO_ROOT:-$(pwd)}
one() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-10s %8d evals/s  K1 %.3f ms' % (sys.argv[1], d['value'], d['kernel_ms_per_step']['prune_K1']))" "$1"; }
for rep in 1 2; do
  if [ -d $root/tmp_r2 ]; then (cd $root/tmp_r2 && python bench.py --no-cpu-baseline --no-forward-rate --steps 10 --warmup 2 2>/dev/null | one r2); fi
  for v in "$@"; do
    d=$root/linearham_amd/lib_exp/$v; [ "$v" = product ] && d=$root/linearham_amd/lib
    LH_LIB_DIR=$d python $root/bench.py --no-cpu-baseline --no-forward-rate --no-extras --steps 10 --warmup 2 2>/dev/null | one $v
  done
done
