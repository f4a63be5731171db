#!/bin/bash
# Add the engine-backed pair styles `e3gnn`, `e3gnn/parallel` and `d3` to a LAMMPS source tree.
#
#   bash lammps/patch_lammps_hip.sh <lammps_root> [<dir holding libsnet_hip.so>] [<cxx_standard>]
#
# Counterpart of the reference's sevenn/pair_e3gnn/patch_lammps.sh (:74-134: back up, copy the pair style sources,
# append to cmake/CMakeLists.txt) for the MI355X engine -- with less to patch: no comm_brick.{h,cpp} replacement
# (the ghost exchange is RCCL inside libsnet_hip.so), no LibTorch, no CUDA.  What LAMMPS links is the one shared
# library `libsnet_hip.so` (built by `python -m sevennet_amd.build`) and the HIP runtime.
# Tested against: nothing yet -- no LAMMPS tree exists in the development image (target: stable_2Aug2023_update3,
# docs/source/user_guide/lammps_torch.md:4 of the reference).
set -e
lammps_root=$1
SCRIPT_DIR=$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)
lib_dir=${2:-$SCRIPT_DIR/../sevennet_amd}
cxx_standard=${3:-17}

if [ "$#" -lt 1 ] || [ "$#" -gt 3 ]; then
    echo "Usage: bash patch_lammps_hip.sh {lammps_root} [{libsnet_hip_dir}] [{cxx_standard}]"
    exit 1
fi
if [ ! -d "$lammps_root" ]; then
    echo "Error: No such directory: $lammps_root"; exit 1
fi
if [ ! -d "$lammps_root/cmake" ] || [ ! -d "$lammps_root/src" ]; then
    echo "Error: Given $lammps_root is not a root of LAMMPS source"; exit 1
fi
if [ ! -f "$SCRIPT_DIR/pair_e3gnn_hip.cpp" ]; then
    echo "Error: pair_e3gnn_hip.cpp not found next to this script"; exit 1
fi
lib_dir=$(cd "$lib_dir" && pwd)
if [ ! -f "$lib_dir/libsnet_hip.so" ]; then
    echo "Error: $lib_dir/libsnet_hip.so not found (build it: python -m sevennet_amd.build)"; exit 1
fi
if grep -q "snet_hip" "$lammps_root/cmake/CMakeLists.txt"; then
    echo "Error: $lammps_root/cmake/CMakeLists.txt is already patched"; exit 1
fi

# 0. back up what is modified
backup_dir="$lammps_root/_backups"
mkdir -p "$backup_dir"
cp "$lammps_root/cmake/CMakeLists.txt" "$backup_dir/CMakeLists.txt"

# 1. the pair styles and the C ABI they call
cp "$SCRIPT_DIR"/pair_e3gnn_hip.{h,cpp} "$SCRIPT_DIR"/pair_d3_hip.{h,cpp} "$lammps_root/src/"
cp "$SCRIPT_DIR/../include/snet_hip.h" "$SCRIPT_DIR/../include/snet_d3_ref.h" "$lammps_root/src/"

# 2. cmake: C++ standard, HIP runtime, libsnet_hip.so (with an rpath so that `lmp` finds it)
sed -i "s/set(CMAKE_CXX_STANDARD 11)/set(CMAKE_CXX_STANDARD $cxx_standard)/" "$lammps_root/cmake/CMakeLists.txt"
cat >> "$lammps_root/cmake/CMakeLists.txt" << EOF2

# ---- SevenNet MI355X force engine (sevennet_amd): pair styles e3gnn, e3gnn/parallel
list(APPEND CMAKE_PREFIX_PATH /opt/rocm)
find_package(hip REQUIRED)
add_library(snet_hip SHARED IMPORTED)
set_target_properties(snet_hip PROPERTIES IMPORTED_LOCATION "$lib_dir/libsnet_hip.so")
target_link_libraries(lammps PUBLIC snet_hip hip::host)
set_property(TARGET lammps APPEND PROPERTY BUILD_RPATH "$lib_dir")
set_property(TARGET lammps APPEND PROPERTY INSTALL_RPATH "$lib_dir")
EOF2

echo "Patched $lammps_root: pair styles e3gnn, e3gnn/parallel and d3 (libsnet_hip.so from $lib_dir)."
echo "Build:  cd $lammps_root && mkdir -p build && cd build && cmake ../cmake -DCMAKE_CXX_COMPILER=hipcc -DBUILD_MPI=yes && make -j"
echo "Model:  python -m sevennet_amd.deploy <checkpoint.pth> -o model.snet    (pair_coeff * * model.snet <elements>)"
