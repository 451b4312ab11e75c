#!/bin/bash
sed -i 's/__global__ void __launch_bounds__(512)/__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(5, 5)))/' linearham_amd/csrc/lh_prune.hip
python3 -m linearham_amd.build > /dev/null 2>&1
for nopf in 0 1; do for a in 4 8; do echo -n "w5 nopf=$nopf ahead=$a: "; LH_K1_AHEAD=$a LH_K1_NOPF=$nopf timeout -k 10 200 python bench.py --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print(round(d['value']), d['kernel_ms_per_step'])"; done; done
