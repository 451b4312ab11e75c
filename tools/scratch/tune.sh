for a in 2 4 6 8 12 16 24; do
  echo -n "ahead $a: "
  LH_K1_AHEAD=$a timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['kernel_ms_per_step']['prune_K1'])"
done
