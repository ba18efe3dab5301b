cd $GRAFT_REPO_ROOT
for L in notaper slots320 default slots1280 slots2560 notaper slots320 default slots1280 slots2560; do
  if [ "$L" = default ]; then unset TRX_LIB; else export TRX_LIB=$GRAFT_REPO_ROOT/profiles/ab_libs/libtrx_$L.so; fi
  echo "== $L"
  python profiles/cells_batch_sweep.py 100000 50 100 200 2>&1 | grep n_time | cut -c1-40
  python profiles/cells_batch_sweep.py 30000 100 2>&1 | grep n_time | cut -c1-40
  python profiles/cells_batch_sweep.py 300000 100 2>&1 | grep n_time| cut -c1-40
  python profiles/bounded_short.py 2>&1 | grep -E "TTP" | sed 's/bounded 0.*bounded 2/b2/' | cut -c1-50
done
