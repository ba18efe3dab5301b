#!/bin/bash
# A/B of the scratch of captured calls:  bash profiles/r06/graph_stress.sh <cycles>
#   graph memory nodes (rounds 2-5; profiles/ab_libs/libtrx_graphmem.so, -DTRX_CAPTURE_GRAPH_MEM)  vs  the tree's library
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r06; mkdir -p $O
C=${1:-8000}
echo "== graph memory nodes, no host synchronisation before the replay"
TRX_LIB=$R/profiles/ab_libs/libtrx_graphmem.so timeout 1200 python profiles/r06/graph_stress.py $C nosync busy-nosync 2>&1 | grep -v amdgpu.ids | tee $O/graph_stress_graphmem.txt | tail -25
echo "== buffers owned by the captured graph (the tree's library), same variants"
timeout 1200 python profiles/r06/graph_stress.py $C nosync busy-nosync 2>&1 | grep -v amdgpu.ids | tee $O/graph_stress_owned.txt | tail -25
echo "== the tree's library, every variant"
timeout 1200 python profiles/r06/graph_stress.py 3000 2>&1 | grep -v amdgpu.ids | tee $O/graph_stress_owned_all.txt | tail -12
