cd $GRAFT_REPO_ROOT
for rep in 1 2; do for C in 0 2560 5120; do
  export TRX_GRID_CAP_P3=$C
  echo "== third-pass cap $C (0 = as the probe pass: 1280)"
  python profiles/bounded_short.py 2>&1 | grep -E "TTP|STP" | sed 's/bounded 0: \([0-9.]*\) ms.*bounded 2: \([0-9.]*\) ms, \([0-9]*\) rows, \([0-9]*\) abandoned.*/b0 \1  b2 \2  (\3 rows, \4 abandoned)/'
  python profiles/batch_timing.py 2>&1 | grep -E "streams (3|4|6)" | cut -c1-110
  python profiles/e2e_streams.py 2>&1 | grep -E "threads 1 streams (3|4|6)"
done; done
