# same-box A/B of two builds of the library on the single-kernel iteration: tools/ab_libs.sh [ab_bench args]
for rep in 1 2 3; do
for lib in libopenimpala_amd_prev.so libopenimpala_amd.so; do
  OI_LIB_PATH=$PWD/openimpala_amd/$lib python tools/ab_bench.py ${1:-1024} ${2:-1}
done; done
