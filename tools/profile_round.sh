#!/bin/bash
# Produces the per-round profile set under gpurun_out/profiles_<tag>/ on the GPU box:
#   <tag>.json                 the bench line of the profiled command
#   <tag>_kernel_stats.csv     rocprofv3 --kernel-trace --stats summary of `python3 bench.py <args>`
#   <tag>_pmc_per_launch.json  per-launch PMC averages (separate --pmc passes, no tracing)
# usage (on the GPU box, from the repo root): bash tools/profile_round.sh r02_bench [evals-per-launch] [bench.py args...]
set -e
tag=${1:-r02_bench}
epl=${2:-49152}
shift || true
shift || true
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/profiles_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
echo "[profile_round] kernel trace: bench.py $@"
timeout -k 5 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o run -- python3 $root/bench.py --no-cpu-baseline --no-forward-rate "$@" > $out/${tag}.json 2> $out/${tag}.err
cp "$(find /tmp/prof_$tag -name '*kernel_stats.csv' | head -1)" $out/${tag}_kernel_stats.csv
dirs=""
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES"; do
  rm -rf /tmp/pmc_${tag}_$i
  echo "[profile_round] pmc pass $i: $grp"
  timeout -k 5 400 rocprofv3 --pmc $grp --output-format csv -d /tmp/pmc_${tag}_$i -o run -- python3 $root/bench.py --no-cpu-baseline --no-forward-rate --steps 3 --warmup 1 "$@" > /dev/null 2> $out/pmc_$i.err
  dirs="$dirs /tmp/pmc_${tag}_$i"
  i=$((i+1))
done
python3 $root/tools/pmc_summary.py $dirs --json $out/${tag}_pmc_per_launch.json --evals-per-launch $epl --note "per-launch averages over the launches of \`python3 bench.py --no-forward-rate --steps 3 --warmup 1 $*\`; separate rocprofv3 --pmc passes; FETCH_SIZE/WRITE_SIZE in KiB as reported (gfx950: FETCH_SIZE counts wide coalesced reads at half their bytes, MI355X_MICROARCH.md); SQ_WAVE_CYCLES/SQ_WAIT_*/SQ_ACTIVE_INST_* are quad-cycles" > $out/pmc_summary.txt
rm -f $out/pmc_*.err
ls -la $out
