# A/B of the tapered batch plan (batch_plan, trx_kernels.hip): base = the commit before, notaper = this source with
# -DTRX_NO_TAPER, default = this source.   bash profiles/ab_taper.sh   (on the GPU box; libs from profiles/build_variants.sh)
cd $GRAFT_REPO_ROOT
for L in base notaper default base notaper default; do
  if [ "$L" = default ]; then unset TRX_LIB; else export TRX_LIB=$GRAFT_REPO_ROOT/profiles/ab_libs/libtrx_$L.so; fi
  echo "== $L"
  python profiles/cells_batch_sweep.py 100000 50 100 200 2>&1 | cut -c1-60
  python profiles/cells_batch_sweep.py 30000 100 2>&1 | cut -c1-60
  python profiles/cells_batch_sweep.py 300000 100 2>&1 | cut -c1-60
  python bench.py --no-cpu-baseline --no-e2e --pmc off --no-batch-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench value %.4g  shapes'%d['value'], {k:'%.4g'%v['evals_per_s'] for k,v in (d.get('shapes') or {}).items()})"
  python profiles/bounded_short.py 2>&1 | grep -E "TTP" | sed 's/bounded 0.*bounded 2/b2/' | cut -c1-90
done
