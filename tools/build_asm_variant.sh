#!/bin/bash
# Builds a variant library whose assembly walks are generated with LH_ASM_OPTS=<opts> (tools/gen_walk_asm.py), into
# linearham_amd/lib_exp/<label>/ -- from a COPY of csrc/ (the kernels include the walks by a quoted name, which resolves
# next to the including file first), so the product's generated files are left alone.
# usage: bash tools/build_asm_variant.sh label "opt1,opt2" ["extra compiler flags" ["sed script applied to the copy of lh_prune.hip"]]
set -euo pipefail
root=$(cd "$(dirname "$0")/.." && pwd)
label=$1
tmp=$(mktemp -d)
cp -r "$root/linearham_amd/csrc" "$tmp/csrc"
LH_ASM_OPTS=$2 LH_ASM_OUT=$tmp/csrc python3 "$root/tools/gen_walk_asm.py" > /dev/null
if [ -n "$2" ] && cmp -s "$tmp/csrc/lh_prune_walk_asm_s2.inc" "$root/linearham_amd/csrc/lh_prune_walk_asm_s2.inc"; then
  echo "warning: options '$2' did not change the generated two-site walk" >&2
fi
if [ -n "${4:-}" ]; then sed -i -E "$4" "$tmp/csrc/lh_prune.hip"; fi
out=$root/linearham_amd/lib_exp/$label
mkdir -p "$out"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-result ${3:-} -I "$root/include" -I "$tmp/csrc" \
  "$tmp"/csrc/lh_{model,prune,forward,asr,sample,capi}.hip -o "$out/liblinearham_hip.so"
cp "$root/linearham_amd/lib/liblinearham_host.so" "$out/"
rm -rf "$tmp"
echo "built $out (asm opts: $2)"
