for w in 4 5 6; do
  sed -i "s/__global__ void __launch_bounds__(kFwdThreads[, 0-9]*)/__global__ void __launch_bounds__(kFwdThreads, $w)/" linearham_amd/csrc/lh_forward.hip
  python -m linearham_amd.build > /dev/null 2>&1
  echo -n "waves/eu $w: "
  timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['kernel_ms_per_step'])"
done
