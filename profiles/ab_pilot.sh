cd $GRAFT_REPO_ROOT
for L in default pilot2048 pilot1024 default pilot2048 pilot1024; do
  if [ "$L" = default ]; then unset TRX_LIB; else export TRX_LIB=$GRAFT_REPO_ROOT/profiles/ab_libs/libtrx_$L.so; fi
  echo "== $L"
  python profiles/bounded_short.py 2>&1 | grep -E "TTP|STP" | sed 's/bounded 0.*bounded 2/b2/' | cut -c1-90
  python - <<'PY'
import os, sys, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np, torch, triceratops_amd
from triceratops_amd import sharding, synth
G = os.path.join(os.environ["GRAFT_REPO_ROOT"], "tests", "golden")
triceratops_amd.set_sampling("device")
jobs = synth.toi_jobs(64, n_time=200, N=1_000_000, seed=synth.SEED, trilegal_fname=os.path.join(G, "trilegal_synth.csv"), contrast_curve_file=os.path.join(G, "contrast_curve_synth.csv"))
small = synth.toi_jobs(2, n_time=200, N=20000, seed=synth.SEED, trilegal_fname=os.path.join(G, "trilegal_synth.csv"), contrast_curve_file=os.path.join(G, "contrast_curve_synth.csv"))
triceratops_amd.calc_probs_many(small)
ts = []
for rep in range(6):
    np.random.seed(5 + rep); torch.manual_seed(5 + rep); torch.cuda.synchronize(); t0 = time.perf_counter()
    out = triceratops_amd.calc_probs_many(jobs); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print("batch step best %.4f median %.4f  fpp checksum %.9f" % (min(ts), sorted(ts)[3], sum(float(t.FPP) for t in out)))
PY
done
