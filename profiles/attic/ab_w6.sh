cd $GRAFT_REPO_ROOT
for rep in 1 2; do for L in default w6; do
  if [ "$L" = default ]; then unset TRX_LIB; else export TRX_LIB=$GRAFT_REPO_ROOT/profiles/ab_libs/libtrx_$L.so; fi
  echo "== $L"
  python profiles/cells_batch_sweep.py 100000 100 200 2>&1 | grep n_time | cut -c1-100
  python profiles/bounded_short.py 2>&1 | grep -E "TTP" | sed 's/bounded 0: \([0-9.]*\) ms.*bounded 2: \([0-9.]*\) ms, \([0-9]*\) rows, \([0-9]*\) abandoned.*/b0 \1  b2 \2  (\3 rows, \4 abandoned)/'
  python profiles/batch_timing.py 2>&1 | grep -E "streams (4)" | cut -c1-110
done; done
