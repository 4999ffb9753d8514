#!/bin/bash
# usage (here, CPU): bash tools/build_ab.sh TAG [-DFLAG=VALUE ...] -- builds build_ab/libfem_hip_TAG.so from the tree's
# device.hip with extra compiler flags (the host objects of the regular build are reused); for tools/ab_libs*.sh
set -e
tag=$1; shift
cd "$(dirname "$0")/../fem_amd/csrc"
make -s build/mesh.o build/mesh_stats.o build/partition.o build/partition_ml.o build/pack.o build/shard.o build/capi.o
mkdir -p ../../build_ab
kb=$(cat kernels_common.hpp kernels_step.hpp kernels_assembly.hpp kernels_implicit.hpp kernels_output.hpp device.hip | sha1sum | cut -c1-12)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fno-slp-vectorize "$@" -DFEM_KERNEL_BUILD=\"$kb-$tag\" --offload-arch=gfx950 -c device.hip -o ../../build_ab/device_$tag.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../build_ab/libfem_hip_$tag.so build/mesh.o build/mesh_stats.o build/partition.o build/partition_ml.o build/pack.o build/shard.o build/capi.o ../../build_ab/device_$tag.o
ls -la ../../build_ab/libfem_hip_$tag.so
