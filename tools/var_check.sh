# K1 time of variant builds (linearham_amd/lib_exp/<label>, tools/build_asm_variant.sh) against the product, bench.py configs[2]
set -e
root=${GRAFT_REPO_ROOT:-$(pwd)}
for v in "" $(ls $root/linearham_amd/lib_exp 2>/dev/null); do
  if [ -n "$v" ]; then export LH_LIB_DIR=$root/linearham_amd/lib_exp/$v; else unset LH_LIB_DIR; fi
  python bench.py --no-cpu-baseline --no-mixed-n --no-extras --no-check --no-live-pmc --steps 20 "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print('variant [$v]', d['config'].get('k1_form'), 'value %.0f k1 %.3f ms' % (d['value'], r['avg_launch_ms']))"
done
