#!/bin/bash
# Builds liblinearham_hip.so with each given set of extra compiler flags and prints the kernel times of
# `bench.py` under rocprofv3 --kernel-trace (for A/B experiments on the GPU box; run from the repo root).
# usage: bash tools/exp_variants.sh "label1=-DFOO" "label2=-DFOO -DBAR=2" ["bench args"]
#   env EXP_BENCH_ARGS: extra bench.py arguments (default: none)
set -e
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
for spec in "$@"; do
  label=${spec%%=*}
  flags=${spec#*=}
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-result $flags -I include -I linearham_amd/csrc \
    linearham_amd/csrc/lh_model.hip linearham_amd/csrc/lh_prune.hip linearham_amd/csrc/lh_forward.hip \
    linearham_amd/csrc/lh_asr.hip linearham_amd/csrc/lh_sample.hip linearham_amd/csrc/lh_capi.hip -o linearham_amd/lib/liblinearham_hip.so 2> /dev/null
  rm -rf /tmp/prof_exp_$label
  (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_exp_$label -o run -- \
    python3 $root/bench.py --no-cpu-baseline --no-forward-rate $EXP_BENCH_ARGS > $root/gpurun_out/exp_$label.json 2> /dev/null)
  echo "== $label ($flags)"
  python3 -c "import json,sys; d=json.load(open(sys.argv[1])); print('   value %.0f evals/s  kernels ms/step %s' % (d['value'], d['kernel_ms_per_step']))" gpurun_out/exp_$label.json
  python3 tools/kernel_stats.py "$(find /tmp/prof_exp_$label -name '*kernel_stats.csv' | head -1)" > /tmp/ks_$label.txt; head -5 /tmp/ks_$label.txt
done
