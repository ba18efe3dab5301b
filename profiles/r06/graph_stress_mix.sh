#!/bin/bash
# the mix of variants under which the first run saw a wrong replay (graph memory nodes): A/B with the tree's library
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r06; mkdir -p $O
C=${1:-30000}
echo "== graph memory nodes, variants test nosync side busy torchcopy"
TRX_LIB=$R/profiles/ab_libs/libtrx_graphmem.so timeout 1500 python profiles/r06/graph_stress.py $C test nosync side busy torchcopy 2>&1 | grep -v amdgpu.ids > $O/graph_stress_mix_graphmem.txt; grep -v "^cycle" $O/graph_stress_mix_graphmem.txt | tail -30
echo "== buffers owned by the captured graph, same variants"
timeout 1500 python profiles/r06/graph_stress.py $C test nosync side busy torchcopy 2>&1 | grep -v amdgpu.ids > $O/graph_stress_mix_owned.txt; grep -v "^cycle" $O/graph_stress_mix_owned.txt | tail -30
