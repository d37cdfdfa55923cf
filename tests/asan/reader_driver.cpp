// Test-only driver: the file readers (oi_io.cpp, oi_hdf5.cpp) compiled with AddressSanitizer + UBSan and fed files
// from the command line.  The product library is never built this way; this executable exists so that the parsers of
// untrusted input run under sanitizers on the CPU box.
// usage: reader_driver tiff|hdf5|raw|dat <dataset or raw "nx,ny,nz,dtype"> files...
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/openimpala_amd.h"

namespace oi {  // the library keeps these in oi_comm.cpp (RCCL side); the readers only need the message slot
static thread_local char g_err[1024];
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
}
const char* last_error() { return g_err; }
}  // namespace oi

int main(int argc, char** argv) {
  if (argc < 4) return 2;
  const char* kind = argv[1];
  const char* arg = argv[2];
  for (int a = 3; a < argc; ++a) {
    uint8_t* phase = nullptr;
    int nx = 0, ny = 0, nz = 0, bps = 0;
    oi_status st = OI_BAD_ARG;
    if (!strcmp(kind, "tiff")) {
      st = oi_read_tiff(argv[a], 0.5, &phase, &nx, &ny, &nz, &bps);
      if (st == OI_OK && nz > 2) {  // a z-slab too
        uint8_t* part = nullptr;
        int px, py, pz, pb;
        if (oi_read_tiff_planes(argv[a], 0.5, 1, nz - 2, &part, &px, &py, &pz, &pb) == OI_OK) oi_free_host(part);
      }
    } else if (!strcmp(kind, "hdf5")) {
      st = oi_read_hdf5(argv[a], arg, 0.5, &phase, &nx, &ny, &nz);
    } else if (!strcmp(kind, "dat")) {
      st = oi_read_dat(argv[a], 0.5, &phase, &nx, &ny, &nz);
    } else if (!strcmp(kind, "raw")) {
      int dt = 1;
      if (sscanf(arg, "%d,%d,%d,%d", &nx, &ny, &nz, &dt) != 4) return 2;
      st = oi_read_raw(argv[a], nx, ny, nz, dt, 0.5, &phase);
    }
    if (st == OI_OK) {
      unsigned long long ones = 0;
      for (size_t v = 0; v < (size_t)nx * ny * nz; ++v) ones += phase[v];  // touch every byte the reader claims
      printf("ok %d %d %d %llu\n", nx, ny, nz, ones);
      oi_free_host(phase);
    } else {
      printf("err %d\n", (int)st);
    }
  }
  printf("batch done\n");
  return 0;
}
