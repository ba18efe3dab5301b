# kernel-by-kernel trace of one lnZ_TTP call, bounded evaluation off / on: bash profiles/trace_call.sh [case]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/trace_call
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_call -- python3 $R/profiles/bounded_trace.py ${1:-toi465_nocc} > $R/gpurun_out/trace_call.log 2>&1
python3 $R/profiles/bounded_trace.py --summary $(ls $R/gpurun_out/trace_call/*/*kernel_trace.csv | head -1)
