#!/bin/bash
# rocprofv3 PMC passes for the bench workload (run on the GPU box via gpurun).
# usage: profiles/pmc_run.sh <tag> [bench args...]
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
ARGS=${@:-"--n-samples 20000 --steps 1 --warmup 0 --no-cpu-baseline"}
cd /tmp && export TMPDIR=/tmp
i=0
for PMC in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_INT32" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}/pass$i -- python3 $R/bench.py $ARGS --no-extras --pmc off > $R/gpurun_out/pmc_${TAG}_pass$i.log 2>&1
done
python3 $R/profiles/pmc_summary.py $R/gpurun_out/pmc_${TAG} > $R/gpurun_out/pmc_${TAG}_summary.txt 2>&1
cat $R/gpurun_out/pmc_${TAG}_summary.txt
