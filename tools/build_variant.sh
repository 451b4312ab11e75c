#!/bin/bash
# Builds a VARIANT of liblinearham_hip.so (extra compiler flags) into its own directory
# linearham_amd/lib_exp/<label>/ next to a copy of the host library (whose rpath is $ORIGIN), leaving the product
# library alone; LH_LIB_DIR=<that directory> makes linearham_amd/capi.py and host.py load it.
# usage (anywhere hipcc runs; the .so files travel to the GPU box): bash tools/build_variant.sh label "-DFOO -DBAR=2"
set -euo pipefail
root=$(cd "$(dirname "$0")/.." && pwd)
label=$1
flags=${2:-}
out=$root/linearham_amd/lib_exp/$label
mkdir -p "$out"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-result $flags -I "$root/include" -I "$root/linearham_amd/csrc" \
  "$root"/linearham_amd/csrc/lh_{model,prune,forward,asr,sample,capi}.hip -o "$out/liblinearham_hip.so"
cp "$root/linearham_amd/lib/liblinearham_host.so" "$out/"
echo "built $out ($flags)"
