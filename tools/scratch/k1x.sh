#!/bin/bash
for x in 0 1 2 4 8 15 16; do
  echo -n "x=$x: "
  LH_K1_X=$x timeout -k 10 200 python bench.py --no-cpu-baseline --no-check --steps 10 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['kernel_ms_per_step']['prune_K1'])"
done
