#!/bin/bash
# Builds a TEST variant of the library with extra defines for oi_kernels.hip (A/B runs through OI_LIB_PATH on the GPU box):
#   tools/variant_build.sh <tag> -DOI_CG1B_EARLY=0      ->  tools/libopenimpala_amd_<tag>.so   (git-ignored, travels with gpurun)
set -e
tag=$1; shift
cd "$(dirname "$0")/../openimpala_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -Wall -Wextra -Wno-unused-parameter --offload-arch=gfx950 -ffp-contract=off "$@" -c oi_kernels.hip -o /tmp/oi_kernels_$tag.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../tools/libopenimpala_amd_$tag.so /tmp/oi_kernels_$tag.o oi_tort.o oi_comm.o oi_io.o oi_hdf5.o \
    oi_effdiff.o oi_mg.o oi_pool.o -L/opt/rocm/lib -lrccl -lz -Wl,-rpath,/opt/rocm/lib
ls -la ../../tools/libopenimpala_amd_$tag.so
