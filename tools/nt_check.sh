# timing + HBM-side traffic of cache-policy variants of the cherry-table kernels (lib_exp/<label>, tools/build_asm_variant.sh)
set -e
root=${GRAFT_REPO_ROOT:-$(pwd)}
for v in "" nt_st nt_both nt_ld; do
  if [ -n "$v" ]; then export LH_LIB_DIR=$root/linearham_amd/lib_exp/$v; else unset LH_LIB_DIR; fi
  python bench.py --no-cpu-baseline --no-mixed-n --no-extras --steps 20 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print('variant [$v]', 'value %.0f k1 %.3f ms traffic %.2f GB  %s' % (d['value'], r['avg_launch_ms'], (r['traffic'] or 0)/1e9, (r.get('traffic_source') or '')[60:200]))"
done
