#!/bin/bash
# Builds two TEST variants of the library whose k_cg1b passes the row offset of its x accesses as the buffer instruction's
# SCALAR offset (the round-2 form that read x from wrong rows now and then): A as it was, B with 32 wait states after each such
# instruction.  Run where hipcc is (no GPU needed); the .so files land in tools/ and travel with gpurun.  On the GPU:
#   OI_LIB_PATH=tools/libopenimpala_amd_soffA.so python tools/cmp_cg1b.py     (and ..._soffB.so)
set -e
cd "$(dirname "$0")/../openimpala_amd/csrc"
for v in A:1 B:2; do
  tag=${v%%:*}; val=${v##*:}
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -Wall -Wextra -Wno-unused-parameter --offload-arch=gfx950 -ffp-contract=off -DOI_SOFFSET_TEST=$val \
      -c oi_k_cg1b.hip -o /tmp/oi_k_cg1b_soff$tag.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../tools/libopenimpala_amd_soff$tag.so /tmp/oi_k_cg1b_soff$tag.o oi_k_setup.o oi_k_dense.o \
      oi_k_reduce.o oi_tort.o oi_comm.o oi_io.o oi_hdf5.o oi_effdiff.o oi_mg.o oi_pool.o -L/opt/rocm/lib -lrccl -lz -Wl,-rpath,/opt/rocm/lib
done
ls -la ../../tools/libopenimpala_amd_soff*.so
