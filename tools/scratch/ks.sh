#!/bin/bash
for ks in 1 2 4; do
  export LH_K2A_KS=$ks
  echo "ks=$ks"
  bash $GRAFT_REPO_ROOT/tools/scratch/prof.sh > /dev/null
  python3 $GRAFT_REPO_ROOT/tools/scratch/kstat.py $GRAFT_REPO_ROOT/gpurun_out/prof_stats.csv emission
done
