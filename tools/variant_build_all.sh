#!/bin/bash
# Builds a TEST variant of the WHOLE library with extra defines on every device / host source that includes oi_device.h
# (layout constants such as OI_LINE_VOX are seen by the kernel files oi_k_*.hip, oi_tort.cpp, oi_mg.hip, oi_effdiff.hip and oi_comm.cpp):
#   tools/variant_build_all.sh <tag> -DOI_LINE_VOX=4   ->  tools/libopenimpala_amd_<tag>.so  (git-ignored, travels with gpurun)
set -e
tag=$1; shift
cd "$(dirname "$0")/../openimpala_amd/csrc"
out=/tmp/oi_variant_$tag
mkdir -p $out
F="-O3 -std=c++17 -fPIC -Wall -Wextra -Wno-unused-parameter --offload-arch=gfx950 -ffp-contract=off"
pids=()
for src in oi_k_setup.hip oi_k_dense.hip oi_k_cg1b.hip oi_k_reduce.hip oi_tort.cpp oi_mg.hip oi_effdiff.hip oi_comm.cpp; do
  /opt/rocm/bin/hipcc $F "$@" -c $src -o $out/${src%.*}.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../tools/libopenimpala_amd_$tag.so $out/oi_k_setup.o $out/oi_k_dense.o $out/oi_k_cg1b.o $out/oi_k_reduce.o $out/oi_tort.o $out/oi_comm.o \
    oi_io.o oi_hdf5.o $out/oi_effdiff.o $out/oi_mg.o oi_pool.o -L/opt/rocm/lib -lrccl -lz -Wl,-rpath,/opt/rocm/lib
ls -la ../../tools/libopenimpala_amd_$tag.so
