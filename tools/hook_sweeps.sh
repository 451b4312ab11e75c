for hook in LH_K1_SEGMENTS=1 LH_K1_STACK=1 LH_K1_CXX_WALK=1 LH_K1_NO_FUSE=1 LH_K1_TILE_CAP=64 LH_K1_NO_TABLES=1 "LH_K1_SEGMENTS=1 LH_K1_SEG_WAVES=5"; do
  echo "== $hook"
  env $hook python tests/dev_tools/random_sweep_forms.py 5000 300 2>&1 | grep "^seed\|^sweep\|^largest\|^forms" | cut -c1-330
done
