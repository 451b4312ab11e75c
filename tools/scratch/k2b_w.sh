#!/bin/bash
cp linearham_amd/csrc/lh_forward.hip /tmp/fwd_orig.hip
for w in 3 4; do
cp /tmp/fwd_orig.hip linearham_amd/csrc/lh_forward.hip
sed -i "s/__global__ void __launch_bounds__(64 \* kJunctionWaves)/__global__ void __launch_bounds__(64 * kJunctionWaves) __attribute__((amdgpu_waves_per_eu($w, $w)))/" linearham_amd/csrc/lh_forward.hip
python3 -m linearham_amd.build > /dev/null 2>&1
echo "waves_per_eu $w"; bash tools/scratch/prof.sh > /dev/null; python3 tools/scratch/kstat.py gpurun_out/prof_stats.csv junction
done
cp /tmp/fwd_orig.hip linearham_amd/csrc/lh_forward.hip
