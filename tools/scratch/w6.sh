#!/bin/bash
for w in 6 8; do
cp /tmp/prune_orig.hip linearham_amd/csrc/lh_prune.hip
sed -i "s/__global__ void __launch_bounds__(512)/__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu($w, $w)))/" linearham_amd/csrc/lh_prune.hip
python3 -m linearham_amd.build > /dev/null 2>&1
echo -n "w$w nopf=1: "; LH_K1_NOPF=1 timeout -k 10 200 python bench.py --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print(round(d['value']), d['kernel_ms_per_step'])"
done
