#!/bin/bash
# rocprofv3 --kernel-trace --stats of `linearham --pipeline` on an n-row table of the configs[2] family
# (prepared by tools/e2e_cli.py).  usage (GPU box, repo root): bash tools/profile_pipeline.sh [n_rows=16384]
set -e
n=${1:-16384}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
python3 tools/e2e_cli.py $n > /dev/null 2>&1   # builds /tmp/lh_e2e_fam and the big table
fam=/tmp/lh_e2e_fam
mkdir -p gpurun_out
rm -rf /tmp/prof_pipe
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_pipe -o run -- $root/linearham_amd/lib/linearham --pipeline \
  --yaml-path $fam/cluster.yaml --cluster-ind 0 --hmm-param-dir $fam/hmm_params --input-path $fam/trees_big.tsv \
  --output-path $fam/lh_prof.tsv --num-rates 4 --seed 1 > /dev/null 2>&1
cp "$(find /tmp/prof_pipe -name '*kernel_stats.csv' | head -1)" $root/gpurun_out/pipeline_kernel_stats.csv
python3 $root/tools/kernel_stats.py $root/gpurun_out/pipeline_kernel_stats.csv | head -12
