#!/bin/bash
# Round 5's measured artefacts in one GPU job:  bash profiles/r05/final_artifacts.sh <tag>   (writes gpurun_out/<tag>_*)
R=${GRAFT_REPO_ROOT:-/root/repo}
T=$1
O=$R/gpurun_out
mkdir -p $O
cd $R
python bench.py > $O/${T}_bench.json 2> $O/${T}_bench.err
python bench.py --fp32-model --no-cpu-baseline --no-e2e > $O/${T}_bench_fp32.json 2>> $O/${T}_bench.err
python bench.py --mode batch > $O/${T}_bench_batch.json 2>> $O/${T}_bench.err
python profiles/r05/batch_timing.py > $O/${T}_batch_timing.txt 2>&1
python profiles/r05/e2e_timing.py > $O/${T}_e2e_timing.txt 2>&1
python profiles/fuzz_kernels.py 60 77 > $O/${T}_fuzz.txt 2>&1
python profiles/fuzz_bounded.py 150 5 > $O/${T}_fuzz_bounded.txt 2>&1
bash profiles/pmc_cells.sh ${T}2000 2000 100000 rows > /dev/null 2>&1
bash profiles/pmc_cells.sh ${T}100 100 100000 cells > /dev/null 2>&1
cp $O/pmc_${T}2000_summary.txt $O/${T}_pmc_cells_2000.txt
cp $O/pmc_${T}100_summary.txt $O/${T}_pmc_cells_100.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_benchprof -- python3 $R/bench.py --no-cpu-baseline --no-extras --pmc off > $O/${T}_benchprof.log 2>&1
cp $(ls $O/${T}_benchprof/*/*kernel_stats.csv | head -1) $O/${T}_bench_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_batchprof -o batch -- python3 $R/profiles/r05/batch_step.py 3 > $O/${T}_batchprof.log 2>&1
python3 $R/profiles/r05/trace_summary.py $O/${T}_batchprof/batch_kernel_trace.csv > $O/${T}_batch_kernel_stats.txt 2>&1
grep step $O/${T}_batchprof.log >> $O/${T}_batch_kernel_stats.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_batchprof1 -o batch -- python3 $R/profiles/r05/batch_step.py 2 64 1000000 1 > $O/${T}_batchprof1.log 2>&1
python3 $R/profiles/r05/trace_summary.py $O/${T}_batchprof1/batch_kernel_trace.csv > $O/${T}_batch_kernel_stats_one_stream.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_e2eprof -o e2e -- python3 $R/profiles/r05/e2e_step.py 3 > $O/${T}_e2eprof.log 2>&1
python3 $R/profiles/r05/trace_summary.py $O/${T}_e2eprof/e2e_kernel_trace.csv > $O/${T}_e2e_kernel_stats.txt 2>&1
grep "run " $O/${T}_e2eprof.log >> $O/${T}_e2e_kernel_stats.txt
echo done
