R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
export TRX_TESTING=1
for rep in 1 2; do
for pc in 16 8 12 20; do
  echo "== probe cells $pc rep $rep"
  TRX_PROBE_CELLS=$pc python profiles/r05/batch_step.py 5 2>/dev/null | grep "step [2345]" | cut -c1-18 | tr '\n' ' '; echo
done
done
