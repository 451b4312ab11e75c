set -e
python -m pytest tests/test_gpu_forms.py tests/test_gpu_parity.py tests/test_gpu_asr.py -q -x > gpurun_out/s1_tests.log 2>&1 || { tail -30 gpurun_out/s1_tests.log; exit 1; }
tail -3 gpurun_out/s1_tests.log
for args in "--preset config2_ragged --brlen-mean 0.0015" "--preset config2_ragged --brlen-mean 0.002" "--preset config2_ragged --brlen-mean 0.003" "--preset config2_ragged"; do
  for hook in "" "LH_K1_STACK=1"; do
  env $hook python bench.py --no-cpu-baseline --no-live-pmc --no-mixed-n --no-extras --steps 20 $args 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$args $hook', 'patterns', d['config']['site_patterns'], 'value %.0f k1 %.3f ms' % (d['value'], d['roofline']['avg_launch_ms']))"
  done
done
