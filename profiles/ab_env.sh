# same-job A/B of an environment switch: bash profiles/ab_env.sh VAR value1 value2 ...
cd $GRAFT_REPO_ROOT
V=$1; shift
for rep in 1 2; do for X in "$@"; do
  export $V=$X
  echo "== $V=$X"
  python profiles/bounded_short.py 2>&1 | grep -E "TTP|TEB" | sed 's/bounded 0: \([0-9.]*\) ms.*bounded 2: \([0-9.]*\) ms, \([0-9]*\) rows, \([0-9]*\) abandoned.*/b0 \1  b2 \2  (\3 rows, \4 abandoned)/'
  python profiles/batch_timing.py 2>&1 | grep -E "streams (3|4|6)" | cut -c1-110
  python profiles/e2e_streams.py 2>&1 | grep -E "threads 1 streams (4)"
done; done
