#!/bin/bash
# experiment library exp/libx_<name>.so built with SNET_CODEGEN_OPTS=<opts>: the object cache is seeded from the working tree's
# build directory, so only the translation units the options change are recompiled
#   tools/build_variant.sh noxt xtile=0 [extra SNET_BUILD_DEFS]
set -e
NAME=${1:?name}; OPTS=${2:-}; DEFS=${3:-}
R=$(cd $(dirname $0)/.. && pwd)
mkdir -p $R/exp
B=$R/sevennet_amd/csrc/build_libx_$NAME
if [ ! -d $B ]; then mkdir -p $B; cp -p $R/sevennet_amd/csrc/build/*.o $R/sevennet_amd/csrc/build/*.stamp $B/ 2>/dev/null || true; fi
cd $R && SNET_BUILD_LIB=$R/exp/libx_$NAME.so SNET_CODEGEN_OPTS="$OPTS" SNET_BUILD_DEFS="$DEFS" python -m sevennet_amd.build -j 16 | tail -1
