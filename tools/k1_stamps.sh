#!/bin/bash
# K1 P-matrix load latency under full load: builds with -DLH_EXP_K1_STAMPS, runs a short bench, prints the line, rebuilds.
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
build() {
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-result "$@" -I include -I linearham_amd/csrc \
    linearham_amd/csrc/lh_model.hip linearham_amd/csrc/lh_prune.hip linearham_amd/csrc/lh_forward.hip \
    linearham_amd/csrc/lh_asr.hip linearham_amd/csrc/lh_sample.hip linearham_amd/csrc/lh_capi.hip -o linearham_amd/lib/liblinearham_hip.so 2>&1 | grep -i "error" 
}
build -DLH_EXP_K1_STAMPS "$@"
python3 bench.py --no-cpu-baseline --no-forward-rate --no-check --steps 3 --warmup 1 2>&1 | grep "K1 \|value" | cut -c1-400
build
