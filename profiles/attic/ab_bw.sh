cd $GRAFT_REPO_ROOT
for L in default bw2 bw1 bw8 slots160 default bw2 bw1 bw8 slots160; do
  if [ "$L" = default ]; then unset TRX_LIB; else export TRX_LIB=$GRAFT_REPO_ROOT/profiles/ab_libs/libtrx_$L.so; fi
  echo "== $L"
  python profiles/cells_batch_sweep.py 100000 50 100 200 2>&1 | grep n_time | cut -c1-100
  python profiles/cells_batch_sweep.py 30000 100 2>&1 | grep n_time | cut -c1-100
  python profiles/cells_batch_sweep.py 300000 100 2>&1 | grep n_time| cut -c1-100
done
