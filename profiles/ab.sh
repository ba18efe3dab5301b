#!/bin/bash
# A/B of libtrx builds on the GPU box: profiles/ab.sh lib1.so lib2.so ...  (paths relative to repo root)
R=${GRAFT_REPO_ROOT:-/root/repo}
for L in "$@"; do
  export TRX_LIB=$R/$L
  echo "=== $L"
  python -m pytest $R/tests/test_gpu_kernels.py -x -q 2>&1 | tail -2
  for rep in 1 2; do
    python $R/bench.py --n-samples ${AB_SAMPLES:-20000} --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('evals/s %.4g  ms/launch %.3f' % (d['value'], d['kernels']['rows_kernel<lnl>']['mean_launch_ms']))"
  done
done
