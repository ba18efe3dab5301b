cd $GRAFT_REPO_ROOT
for L in "" g1024 g512 g1024c2048 g512c2048; do
  bash profiles/draw_stats.sh abdraw_${L:-default} $L 2>&1 | grep "=="
done
for rep in 1 2; do for L in default g1024 g512 g1024c2048 g512c2048; do
  if [ "$L" = default ]; then unset TRX_LIB; else export TRX_LIB=$GRAFT_REPO_ROOT/profiles/ab_libs/libtrx_$L.so; fi
  echo "== $L"
  python profiles/batch_timing.py 2>&1 | grep -E "streams (4)" | cut -c1-110
  python profiles/e2e_streams.py 2>&1 | grep -E "threads 1 streams (4)"
done; done
