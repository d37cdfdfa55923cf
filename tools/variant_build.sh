#!/bin/bash
# Builds a TEST variant of the library with extra defines for the kernel files oi_k_*.hip (A/B runs through OI_LIB_PATH on the GPU box):
#   tools/variant_build.sh <tag> -DOI_CG1B_EARLY=0      ->  tools/libopenimpala_amd_<tag>.so   (git-ignored, travels with gpurun)
# (-DOI_XCHECK=1 among the defines adds the superseded kernels, as `make xcheck` does)
set -e
tag=$1; shift
cd "$(dirname "$0")/../openimpala_amd/csrc"
out=/tmp/oi_variant_$tag
mkdir -p $out
F="-O3 -std=c++17 -fPIC -Wall -Wextra -Wno-unused-parameter --offload-arch=gfx950 -ffp-contract=off"
srcs="oi_k_setup.hip oi_k_dense.hip oi_k_cg1b.hip oi_k_reduce.hip"
case " $* " in *" -DOI_XCHECK=1 "*) srcs="$srcs oi_k_xcheck.hip oi_tort.cpp oi_effdiff.hip";; esac
pids=(); objs=""
for src in $srcs; do
  /opt/rocm/bin/hipcc $F "$@" -c $src -o $out/${src%.*}.o &
  pids+=($!); objs="$objs $out/${src%.*}.o"
done
for p in "${pids[@]}"; do wait $p; done
case " $* " in *" -DOI_XCHECK=1 "*) rest="oi_comm.o oi_mg.o";; *) rest="oi_tort.o oi_comm.o oi_effdiff.o oi_mg.o";; esac
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../tools/libopenimpala_amd_$tag.so $objs $rest oi_io.o oi_hdf5.o oi_pool.o \
    -L/opt/rocm/lib -lrccl -lz -Wl,-rpath,/opt/rocm/lib
ls -la ../../tools/libopenimpala_amd_$tag.so
