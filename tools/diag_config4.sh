#!/bin/bash
# Counter sets of K1 on the configs[4] shape for a list of environment settings (one rocprofv3 --pmc pass per counter
# group and setting, no tracing); per-launch averages of the prune kernels go to gpurun_out/diag_config4_<label>.txt.
# usage (GPU box, repo root): bash tools/diag_config4.sh "seg4:" ["label:VAR=1 ..."]
# (profiles/r04_config4_pmc_*.txt were made with `"seg4:" "ctseg:LH_K1_CT_SEGMENTS=1"` on the round-3 build 0b72f3b; the
# segment-wise assembly form that switch selected left the source in round 4.)
root=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $root/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $root/gpurun_out/rocprof_counters.txt 2>&1 || true
groups=(
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"
  "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INSTS_BRANCH"
  "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE SQC_TC_DATA_READ_REQ SQC_TC_REQ SQC_TC_STALL SQC_DCACHE_INPUT_VALID_READYB"
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum"
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"
)
for spec in "$@"; do
  label=${spec%%:*}
  envs=${spec#*:}
  out=$root/gpurun_out/diag_config4_$label.txt
  echo "# $label ($envs): bench.py --preset config4 --steps 2 --warmup 1; per-launch averages per kernel" > $out
  i=0
  for g in "${groups[@]}"; do
    rm -rf /tmp/diag_${label}_$i
    ([ -n "$envs" ] && export $envs; timeout -k 5 300 rocprofv3 --pmc $g --output-format csv -d /tmp/diag_${label}_$i -o run -- \
      python3 $root/bench.py --preset config4 --no-cpu-baseline --no-forward-rate --no-check --no-extras --steps 2 --warmup 1 \
      > /dev/null 2> /tmp/diag_${label}_$i.err) || { echo "# pass $i ($g) failed: $(tail -2 /tmp/diag_${label}_$i.err | tr '\n' ' ')" >> $out; }
    echo "[diag] $label pass $i done"
    python3 $root/tools/pmc_summary.py /tmp/diag_${label}_$i --filter prune >> $out 2>&1 || true
    i=$((i+1))
  done
done
