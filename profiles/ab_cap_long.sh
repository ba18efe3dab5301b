cd $GRAFT_REPO_ROOT
for rep in 1 2; do for C in 0 4096 8192 16384 32768 65536; do
  export TRX_GRID_CAP_LONG=$C
  echo "== long cap $C"
  python profiles/bounded_e2e.py 2>&1 | grep kep10 | head -2
done; done
unset TRX_GRID_CAP_LONG
for rep in 1 2; do for C in 1280 2560 3840; do
  export TRX_GRID_CAP=$C
  echo "== probe cap $C (plain 5120)"
  python profiles/batch_timing.py 2>&1 | grep -E "streams (3|4|6)" | cut -c1-110
  python profiles/e2e_streams.py 2>&1 | grep -E "threads 1 streams (3|6)"
done; done
