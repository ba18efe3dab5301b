#!/bin/bash
# Round 6's measured artefacts in one GPU job:  bash profiles/r06/final_artifacts.sh <tag>   (writes gpurun_out/r06/<tag>_*)
R=${GRAFT_REPO_ROOT:-/root/repo}
T=$1
O=$R/gpurun_out/r06
mkdir -p $O
cd $R
python bench.py > $O/${T}_bench.out 2> $O/${T}_bench.err; tail -1 $O/${T}_bench.out > $O/${T}_bench.json
python bench.py --fp32-model --no-cpu-baseline --no-e2e 2>> $O/${T}_bench.err | tail -1 > $O/${T}_bench_fp32.json
python bench.py --mode batch 2>> $O/${T}_bench.err | tail -1 > $O/${T}_bench_batch.json
# the same line under an external launcher with ONE rank (the RCCL group of torchrun's environment)
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --pmc off 2>> $O/${T}_bench.err | grep '^{' | tail -1 > $O/${T}_bench_torchrun1.json
export TRX_TESTING=1      # the scripts below drive switches of include/trx_debug.h: the testing library
python profiles/r05/batch_timing.py > $O/${T}_batch_timing.txt 2>&1
python profiles/r05/e2e_timing.py > $O/${T}_e2e_timing.txt 2>&1
python profiles/fuzz_kernels.py 150 77 > $O/${T}_fuzz.txt 2>&1
python profiles/fuzz_bounded.py 150 5 > $O/${T}_fuzz_bounded.txt 2>&1
bash profiles/pmc_cells.sh ${T}2000 2000 100000 rows > /dev/null 2>&1
bash profiles/pmc_cells.sh ${T}100 100 100000 cells > /dev/null 2>&1
cp $R/gpurun_out/pmc_${T}2000_summary.txt $O/${T}_pmc_cells_2000.txt
cp $R/gpurun_out/pmc_${T}100_summary.txt $O/${T}_pmc_cells_100.txt
unset TRX_TESTING
python profiles/r06/graph_stress.py 6000 > $O/${T}_graph_stress.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_benchprof -- python3 $R/bench.py --no-cpu-baseline --no-extras --no-batch-leg --pmc off > $O/${T}_benchprof.log 2>&1
cp $(ls $O/${T}_benchprof/*/*kernel_stats.csv | head -1) $O/${T}_bench_kernel_stats.csv
cd $R
bash profiles/r06/chain_trace.sh ${T}_chain > $O/${T}_chain_one_stream.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_e2eprof -o e2e -- python3 $R/profiles/r05/e2e_step.py 3 > $O/${T}_e2eprof.log 2>&1
python3 $R/profiles/r05/trace_summary.py $O/${T}_e2eprof/e2e_kernel_trace.csv > $O/${T}_e2e_kernel_stats.txt 2>&1
grep "run " $O/${T}_e2eprof.log >> $O/${T}_e2e_kernel_stats.txt
rm -rf $O/${T}_benchprof $O/${T}_e2eprof $O/${T}_chain_prof $R/gpurun_out/pmc_${T}2000 $R/gpurun_out/pmc_${T}100
echo done
