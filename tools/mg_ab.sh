for rep in 1 2; do
for lib in libopenimpala_amd_prev.so libopenimpala_amd.so; do
  OI_LIB_PATH=$PWD/openimpala_amd/$lib python tools/mg_ab.py 1024 | tail -1
  OI_LIB_PATH=$PWD/openimpala_amd/$lib python tools/mg_ab.py 1024 1 | tail -1 | sed 's/^/dense /'
done; done
