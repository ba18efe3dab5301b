#!/bin/bash
# instruction-mix PMC passes for several libtrx builds: profiles/pmc_ab.sh lib1.so lib2.so ...
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for L in "$@"; do
  export TRX_LIB=$R/$L
  TAG=$(basename $L .so)
  i=0
  for PMC in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
             "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_INT32" \
             "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_IFETCH"; do
    i=$((i+1))
    rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}/pass$i -- python3 $R/bench.py --n-samples 20000 --steps 1 --warmup 0 --no-cpu-baseline --no-extras --pmc off > $R/gpurun_out/pmc_${TAG}_pass$i.log 2>&1
  done
  python3 $R/profiles/pmc_summary.py $R/gpurun_out/pmc_${TAG} > $R/gpurun_out/pmc_${TAG}_summary.txt 2>&1
  echo "=== $L"; grep -A30 "void  <0" $R/gpurun_out/pmc_${TAG}_summary.txt | head -32
done
