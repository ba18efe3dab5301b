# same-job A/B of libraries: bash profiles/ab_three.sh name1 name2 ... ("default" = the tree's library; others from profiles/ab_libs)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for L in "$@"; do
  if [ "$L" = default ]; then unset TRX_LIB; else export TRX_LIB=$GRAFT_REPO_ROOT/profiles/ab_libs/libtrx_$L.so; fi
  echo "== $L"
  python profiles/bounded_short.py 2>&1 | grep -E "TTP|STP" | sed 's/bounded 0: \([0-9.]*\) ms.*bounded 2: \([0-9.]*\) ms, \([0-9]*\) rows, \([0-9]*\) abandoned.*/b0 \1  b2 \2  (\3 rows, \4 abandoned)/'
  python profiles/batch_timing.py 2>&1 | grep -E "streams (3|6)" | cut -c1-110
  python profiles/e2e_streams.py 2>&1 | grep -E "threads 1 streams (3|6)"
done; done
