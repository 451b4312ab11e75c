#!/bin/bash
export LH_K2A_KS=1
echo base; bash tools/scratch/prof.sh > /dev/null; python3 tools/scratch/kstat.py gpurun_out/prof_stats.csv junction
for w in 4; do
sed -i "s/__launch_bounds__(64 \* kJunctionWaves)/__launch_bounds__(64 * kJunctionWaves) __attribute__((amdgpu_waves_per_eu($w, $w)))/" linearham_amd/csrc/lh_forward.hip
python3 -m linearham_amd.build > /dev/null 2>&1
echo "waves_per_eu $w"; bash tools/scratch/prof.sh > /dev/null; python3 tools/scratch/kstat.py gpurun_out/prof_stats.csv junction
done
