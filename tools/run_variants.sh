#!/bin/bash
# Times bench.py (HIP-event kernel times, no parity check: variants may be wrong by construction) for each variant
# directory built by tools/build_variant.sh; "product" = the in-tree library.  Run on the GPU box from the repo root.
# usage: bash tools/run_variants.sh product label1 label2 ...   (env EXP_BENCH_ARGS: extra bench.py arguments)
set -uo pipefail
root=${GRAFT_REPO_ROOT:-$(pwd)}
for label in "$@"; do
  dir=$root/linearham_amd/lib_exp/$label
  [ "$label" = product ] && dir=$root/linearham_amd/lib
  if [ ! -f "$dir/liblinearham_hip.so" ]; then echo "== $label: not built"; continue; fi
  LH_LIB_DIR=$dir python3 "$root/bench.py" --no-cpu-baseline --no-forward-rate --no-check --steps 5 --warmup 2 ${EXP_BENCH_ARGS:-} \
    > "$root/gpurun_out/var_$label.json" 2> "$root/gpurun_out/var_$label.err" || { echo "== $label: bench failed"; tail -3 "$root/gpurun_out/var_$label.err"; continue; }
  python3 -c "import json,sys; d=json.load(open(sys.argv[1])); print('== %-14s %9.0f evals/s  K0 %.3f  K1 %.3f  K2 %.3f ms/step' % (sys.argv[2], d['value'], d['kernel_ms_per_step']['model_K0'], d['kernel_ms_per_step']['prune_K1'], d['kernel_ms_per_step']['forward_K2']))" "$root/gpurun_out/var_$label.json" "$label"
done
