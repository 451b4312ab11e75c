#!/bin/bash
# Builds a variant library whose assembly walk is generated with LH_ASM_OPTS=<opts> (tools/gen_walk_asm.py), into
# linearham_amd/lib_exp/<label>/ -- the product's lh_prune_walk_asm.inc is left alone.
# usage: bash tools/build_asm_variant.sh label "opt1,opt2" ["extra compiler flags"]
set -euo pipefail
root=$(cd "$(dirname "$0")/.." && pwd)
label=$1
tmp=$(mktemp -d)
LH_ASM_OPTS=$2 LH_ASM_OUT=$tmp/lh_prune_walk_asm.inc python3 "$root/tools/gen_walk_asm.py" > /dev/null
out=$root/linearham_amd/lib_exp/$label
mkdir -p "$out"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-result ${3:-} -I "$tmp" -I "$root/include" -I "$root/linearham_amd/csrc" \
  "$root"/linearham_amd/csrc/lh_{model,prune,forward,asr,sample,capi}.hip -o "$out/liblinearham_hip.so"
cp "$root/linearham_amd/lib/liblinearham_host.so" "$out/"
rm -rf "$tmp"
echo "built $out (asm opts: $2)"
