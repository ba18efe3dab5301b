# smoke(), the GPU suite and the default bench line on the tree's library: bash profiles/final_check.sh
cd ${GRAFT_REPO_ROOT:-/root/repo}
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python -m pytest tests -m gpu -x -q 2>&1 | tail -2
python bench.py 2>/dev/null > gpurun_out/final_check_bench.json
python - <<'PY'
import json
d = json.loads(open("gpurun_out/final_check_bench.json").read().strip().splitlines()[-1])
print("bench value %.4g frac %.4f batch %.1f ms e2e %s" % (d["value"], d["roofline"]["frac"], d["batch"]["ms_per_step"],
      {k: round(v, 4) for k, v in d["e2e"]["seconds"].items() if "device" in k and "numpy" not in k}))
PY
