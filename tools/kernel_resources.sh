#!/bin/bash
# VGPR / spill / LDS / scratch figures of every kernel in one object of sevennet_amd/csrc/build (gfx950 code object notes)
#   tools/kernel_resources.sh convf_22d6a77ad5ac [name filter]
set -e
OBJ=${1:?object name without .o}; FILT=${2:-.}
D=$(mktemp -d); B=$(dirname $0)/../sevennet_amd/csrc/build${SNET_BUILD_SFX:-}
/opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin=$D/fat.bin $B/$OBJ.o
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$D/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$D/dev.co
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $D/dev.co | python3 -c "
import sys, re
txt = sys.stdin.read()
for blk in txt.split('- .agpr_count')[1:]:
    g = lambda k: (re.search(r'\.' + k + r':\s+(\S+)', blk) or [None, '?'])[1]
    name = g('name')
    if re.search(r'$FILT', name):
        short = re.sub(r'^_ZN12_GLOBAL__N_1\d+', '', name)[:60]
        print(f'{short:60s} vgpr {g(\"vgpr_count\"):>4s} spill {g(\"vgpr_spill_count\"):>4s} sgpr {g(\"sgpr_count\"):>4s} lds {g(\"group_segment_fixed_size\"):>6s} scratch {g(\"private_segment_fixed_size\"):>5s}')
"
rm -rf $D
