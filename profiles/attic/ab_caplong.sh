cd $GRAFT_REPO_ROOT
for rep in 1 2; do for C in 0 8192 16384 32768 65536; do
  export TRX_GRID_CAP_LONG_HOST=$C
  echo "== host-known rows, one row per wave: cap $C"
  python bench.py --no-cpu-baseline --no-e2e --pmc off --no-batch-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench value %.4g launch %.3f ms'%(d['value'], d['roofline']['mean_launch_ms']), {k:'%.4g'%v['evals_per_s'] for k,v in (d.get('shapes') or {}).items()})"
done; done
