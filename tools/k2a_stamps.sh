#!/bin/bash
# K2a phase timing: builds the library with -DLH_EXP_K2A_STAMPS, runs a short bench, prints the stamp line, and
# rebuilds the normal library.  usage (GPU box, repo root): bash tools/k2a_stamps.sh [extra -D flags]
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
build() {
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-result "$@" -I include -I linearham_amd/csrc \
    linearham_amd/csrc/lh_model.hip linearham_amd/csrc/lh_prune.hip linearham_amd/csrc/lh_forward.hip \
    linearham_amd/csrc/lh_asr.hip linearham_amd/csrc/lh_sample.hip linearham_amd/csrc/lh_capi.hip -o linearham_amd/lib/liblinearham_hip.so 2> /dev/null
}
build -DLH_EXP_K2A_STAMPS "$@"
python3 bench.py --no-cpu-baseline --no-forward-rate --no-check --steps 3 --warmup 1 2>&1 | grep "stamps"
build
