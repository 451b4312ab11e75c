#!/bin/bash
# Kernel times and counters of the device side of RunPipeline (K0-K2 with the forward arrays written, then K4) at the
# configs[2] size: `bench.py --no-mixed-n` runs it as its pipeline_rows_per_s extra.  Writes gpurun_out/k4_profile.txt.
# usage (GPU box, repo root): bash tools/profile_k4.sh
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/k4_profile.txt
mkdir -p $root/gpurun_out
cd /tmp && export TMPDIR=/tmp
args="--no-cpu-baseline --no-check --no-mixed-n --steps 3 --warmup 1"
rm -rf /tmp/k4p_trace
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k4p_trace -o run -- python3 $root/bench.py $args > /dev/null 2> /tmp/k4p_trace.err
echo "# kernel-trace averages (us): bench.py $args" > $out
python3 $root/tools/kernel_stats.py "$(find /tmp/k4p_trace -name '*kernel_stats.csv' | head -1)" | head -14 >> $out
i=0
for g in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_FLAT" \
         "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf /tmp/k4p_$i
  timeout -k 5 300 rocprofv3 --pmc $g --output-format csv -d /tmp/k4p_$i -o run -- python3 $root/bench.py $args > /dev/null 2> /tmp/k4p_$i.err
  echo "# pmc pass: $g (per-launch averages)" >> $out
  python3 $root/tools/pmc_summary.py /tmp/k4p_$i --filter sample_kernel >> $out 2>&1
  python3 $root/tools/pmc_summary.py /tmp/k4p_$i --filter junction >> $out 2>&1
  echo "[k4 profile] pass $i done"
  i=$((i+1))
done
