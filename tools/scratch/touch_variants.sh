#!/bin/bash
cp linearham_amd/csrc/lh_prune.hip /tmp/prune_orig.hip
run() { python3 -m linearham_amd.build > /dev/null 2>&1; echo -n "$1: "; timeout -k 10 200 python bench.py --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print(round(d['value']), d['kernel_ms_per_step']['prune_K1'])"; }
# V1: barriers only
cp /tmp/prune_orig.hip linearham_amd/csrc/lh_prune.hip
sed -i 's/if ((k \& 3) == 0) {/if (false) {/' linearham_amd/csrc/lh_prune.hip; run "V1 barriers only"
# V2: every 2nd op, ops k+2..k+3
cp /tmp/prune_orig.hip linearham_amd/csrc/lh_prune.hip
sed -i 's/if ((k \& 3) == 0) {/if ((k \& 1) == 0) {/; s/const int hi = k + 8 < n_ops ? k + 8 : n_ops;/const int hi = k + 4 < n_ops ? k + 4 : n_ops;/; s/for (int o = k + 4 + wave; o < hi; o += n_waves) {/for (int o = k + 2 + wave; o < hi; o += n_waves) {/' linearham_amd/csrc/lh_prune.hip; run "V2 every 2nd, +2..+3"
# V3: no sched barriers
cp /tmp/prune_orig.hip linearham_amd/csrc/lh_prune.hip
sed -i 's/__builtin_amdgcn_sched_barrier(0);//' linearham_amd/csrc/lh_prune.hip; run "V3 touches, no barriers"
# V4: only wave 0 touches, one line per op (q[0]) for +4..+7
cp /tmp/prune_orig.hip linearham_amd/csrc/lh_prune.hip
sed -i 's/sink ^= q\[0\] ^ q\[16\] ^ q\[32\] ^ q\[48\];/sink ^= q[0];/' linearham_amd/csrc/lh_prune.hip; run "V4 one line per op"
cp /tmp/prune_orig.hip linearham_amd/csrc/lh_prune.hip
