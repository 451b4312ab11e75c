#!/bin/bash
# Profile set of the ancestral-sequence sampling step under gpurun_out/profiles_<tag>/ (GPU box, repo root):
#   <tag>_asr.json  <tag>_asr_kernel_stats.csv  <tag>_asr_pmc_per_launch.json
# usage: bash tools/profile_asr.sh r01_asr_v1 [--batch 4096 ...]
set -e
tag=${1:-r01_asr}
shift || true
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/profiles_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
echo "[profile_asr] kernel trace"
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o run -- python3 $root/bench_asr.py "$@" > $out/${tag}_asr.json 2> $out/${tag}_asr.err
cp "$(find /tmp/prof_$tag -name '*kernel_stats.csv' | head -1)" $out/${tag}_asr_kernel_stats.csv
dirs=""
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS"; do
  rm -rf /tmp/pmc_${tag}_$i
  echo "[profile_asr] pmc pass $i: $grp"
  timeout -k 5 300 rocprofv3 --pmc $grp --output-format csv -d /tmp/pmc_${tag}_$i -o run -- python3 $root/bench_asr.py "$@" --steps 2 --warmup 1 > /dev/null 2> $out/pmc_$i.err || tail -3 $out/pmc_$i.err
  dirs="$dirs /tmp/pmc_${tag}_$i"
  i=$((i+1))
done
python3 $root/tools/pmc_summary.py $dirs --json $out/${tag}_asr_pmc_per_launch.json --note "per-launch averages over the launches of \`python3 bench_asr.py $* --steps 2 --warmup 1\`; separate rocprofv3 --pmc passes; FETCH_SIZE/WRITE_SIZE in KiB as reported (gfx950: FETCH_SIZE counts wide coalesced reads at half their bytes, MI355X_MICROARCH.md); SQ_WAVE_CYCLES/SQ_WAIT_*/SQ_ACTIVE_INST_* are quad-cycles" > $out/pmc_summary.txt
rm -f $out/pmc_*.err
cat $out/pmc_summary.txt
cut -c1-160 $out/${tag}_asr_kernel_stats.csv
