#!/bin/bash
# The measured artefacts of a round in one GPU job: profiles/final_artifacts.sh <tag>   (writes gpurun_out/<tag>_*)
R=${GRAFT_REPO_ROOT:-/root/repo}
T=$1
O=$R/gpurun_out
mkdir -p $O
cd $R
python bench.py > $O/${T}_bench.json 2> $O/${T}_bench.err
python bench.py --fp32-model --no-cpu-baseline > $O/${T}_bench_fp32.json 2>> $O/${T}_bench.err
python bench.py --mode batch > $O/${T}_bench_batch.json 2>> $O/${T}_bench.err
python bench.py --mode batch --fp32-model > $O/${T}_bench_batch_fp32.json 2>> $O/${T}_bench.err
python profiles/short_curves.py 100000 > $O/${T}_short_curves.txt 2>&1
python profiles/cells_batch_sweep.py 100000 50 100 200 > $O/${T}_cells_batch_sweep.txt 2>&1
python profiles/cells_batch_sweep.py 30000 100 >> $O/${T}_cells_batch_sweep.txt 2>&1
python profiles/cells_batch_sweep.py 300000 100 >> $O/${T}_cells_batch_sweep.txt 2>&1
python profiles/e2e_streams.py > $O/${T}_e2e_streams.txt 2>&1
python profiles/batch_timing.py > $O/${T}_batch_timing.txt 2>&1
python profiles/batch_host_profile.py 2>&1 | head -60 > $O/${T}_batch_host_profile.txt
python profiles/bounded_short.py > $O/${T}_bounded_short.txt 2>&1
python profiles/fuzz_kernels.py 300 77 > $O/${T}_fuzz.txt 2>&1
bash profiles/draw_stats.sh ${T} > $O/${T}_draw_kernel.txt 2>&1
bash profiles/stats_batch.sh ${T}batch > $O/${T}_batch_kernel_stats.txt 2>&1
python profiles/step_union.py $(ls $O/stats_${T}batch/*/*kernel_trace.csv | head -1) >> $O/${T}_batch_kernel_stats.txt 2>&1
python profiles/bounded_e2e.py > $O/${T}_bounded_e2e.txt 2>&1
python profiles/prune_potential.py > $O/${T}_prune_potential.txt 2>&1
python profiles/stencil_check.py 2000 3000 > $O/${T}_stencil.txt 2>&1
bash profiles/stats_e2e.sh ${T}e2e blend > $O/${T}_e2e_kernel_stats.txt 2>&1
bash profiles/pmc_cells.sh ${T}2000 2000 100000 rows > /dev/null 2>&1
bash profiles/pmc_cells.sh ${T}100 100 100000 cells > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_benchprof -- python3 $R/bench.py --no-cpu-baseline --no-extras --pmc off > $O/${T}_benchprof.log 2>&1
cp $(ls $O/${T}_benchprof/*/*kernel_stats.csv | head -1) $O/${T}_bench_kernel_stats.csv
echo done
