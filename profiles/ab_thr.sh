cd $GRAFT_REPO_ROOT
for L in thr10000 default thr10000 default; do
  if [ "$L" = default ]; then unset TRX_LIB; else export TRX_LIB=$GRAFT_REPO_ROOT/profiles/ab_libs/libtrx_$L.so; fi
  echo "== $L"
  python profiles/bounded_short.py 2>&1 | grep -E "TTP|STP" | cut -c1-190
  python profiles/batch_timing.py 2>&1 | grep -E "streams (3|6)"
  python profiles/e2e_streams.py 2>&1 | grep -E "threads 1 streams (3|6)"
done
