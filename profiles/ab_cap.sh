cd $GRAFT_REPO_ROOT
for rep in 1 2; do for C in 0 1280 2560 3840 5120 7680; do
  export TRX_GRID_CAP=$C
  echo "== cap $C"
  python profiles/cells_batch_sweep.py 100000 100 200 2>&1 | grep n_time | cut -c1-75
  python profiles/cells_batch_sweep.py 300000 100 2>&1 | grep n_time | cut -c1-75
  python profiles/bounded_short.py 2>&1 | grep -E "TTP|STP" | sed 's/bounded 0: \([0-9.]*\) ms.*bounded 2: \([0-9.]*\) ms, \([0-9]*\) rows, \([0-9]*\) abandoned.*/b0 \1  b2 \2  (\3 rows, \4 abandoned)/'
  python profiles/batch_timing.py 2>&1 | grep -E "streams (3|6)" | cut -c1-110
  python profiles/e2e_streams.py 2>&1 | grep -E "threads 1 streams (3|6)"
done; done
