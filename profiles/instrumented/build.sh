#!/bin/bash
# Instrumented / experimental builds of libtrx.so for A/B runs (TRX_LIB=profiles/ab_libs/libtrx_<name>.so):
#   profiles/instrumented/build.sh <name> <patch.py> [extra hipcc flags]
# copies csrc to a scratch directory, applies the python patch (it receives the directory) and builds there;
# the tree is not touched.  agm_count_patch.py counts the AGM steps of cel_pair per lane and per wave
# (profiles/agm_steps.py reads them); an empty patch gives the reference point of an A/B ("base").
R=$(cd "$(dirname "$0")/../.." && pwd)
NAME=$1; PATCH=$2; shift 2
W=$(mktemp -d)
mkdir -p $W/triceratops_amd $W/include $R/profiles/ab_libs
cp -r $R/triceratops_amd/csrc $W/triceratops_amd/; cp $R/include/trx.h $W/include/
if [ -n "$PATCH" ] && [ "$PATCH" != none ]; then python3 $PATCH $W/triceratops_amd/csrc || exit 1; fi
cd $W/triceratops_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -mllvm -disable-machine-licm "$@" \
    -o $R/profiles/ab_libs/libtrx_$NAME.so trx_kernels.hip trx_draw.hip trx_scenario.hip 2>&1 | grep -E "error"
rm -rf $W
echo built $NAME
