// What can a trivially simple kernel sustain when its 16-byte loads/stores are predicated on a mask
// with the spatial statistics of the bench structure?  Reads a code file (1 byte per voxel, e.g. dumped
// by tools/dump_codes.py), then times  c = a + s*b ; d = a - s*b  (3 reads incl. the mask byte, 2 writes)
//   (1) unpredicated, (2) predicated on the lane's code pair.
// Reported: ms, dense GB/s, and GB/s of the bytes a 128-byte-line-granular memory system must move.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/stream_bench_masked tools/stream_bench_masked.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <bool PRED, int U>
__global__ void __launch_bounds__(256) k_mix(const unsigned char* __restrict__ code, const double* __restrict__ a,
                                              const double* __restrict__ b, double* __restrict__ c, double* __restrict__ d,
                                              double s, long long npairs) {
  const long long per = (npairs + gridDim.x - 1) / gridDim.x;
  const long long b0 = per * blockIdx.x, b1 = b0 + per < npairs ? b0 + per : npairs;
  for (long long t = b0 + threadIdx.x; t < b1; t += U * 256) {
    double2 av[U], bv[U];
    unsigned short cc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long e = t + u * 256;
      cc[u] = e < b1 ? *reinterpret_cast<const unsigned short*>(code + 2 * e) : 0;
      if (!PRED && e < b1) cc[u] = 1;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long e = t + u * 256;
      if (cc[u]) { av[u] = *reinterpret_cast<const double2*>(a + 2 * e); bv[u] = *reinterpret_cast<const double2*>(b + 2 * e); }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long e = t + u * 256;
      if (cc[u]) {
        *reinterpret_cast<double2*>(c + 2 * e) = make_double2(av[u].x + s * bv[u].x, av[u].y + s * bv[u].y);
        *reinterpret_cast<double2*>(d + 2 * e) = make_double2(av[u].x - s * bv[u].x, av[u].y - s * bv[u].y);
      }
    }
  }
}

int main(int argc, char** argv) {
  if (argc < 2) { printf("usage: stream_bench_masked <code file>\n"); return 1; }
  FILE* f = fopen(argv[1], "rb");
  if (!f) { printf("cannot open %s\n", argv[1]); return 1; }
  fseek(f, 0, SEEK_END);
  const long long n = ftell(f) & ~1ll;
  fseek(f, 0, SEEK_SET);
  std::vector<unsigned char> h(n);
  if (fread(h.data(), 1, n, f) != (size_t)n) { printf("short read\n"); return 1; }
  fclose(f);
  long long live_lines = 0, lines = (n + 15) / 16, free_vox = 0;
  for (long long l = 0; l < lines; ++l) {
    bool any = false;
    for (long long v = 16 * l; v < 16 * l + 16 && v < n; ++v) { any |= h[v] != 0; free_vox += h[v] != 0; }
    live_lines += any;
  }
  printf("voxels %lld, free %.4f, live 16-voxel lines %.4f\n", n, (double)free_vox / n, (double)live_lines / lines);
  unsigned char* code; double *a, *b, *c, *d;
  CK(hipMalloc(&code, n)); CK(hipMalloc(&a, 8 * n)); CK(hipMalloc(&b, 8 * n)); CK(hipMalloc(&c, 8 * n)); CK(hipMalloc(&d, 8 * n));
  CK(hipMemcpy(code, h.data(), n, hipMemcpyHostToDevice));
  CK(hipMemset(a, 0, 8 * n)); CK(hipMemset(b, 0, 8 * n)); CK(hipMemset(c, 0, 8 * n)); CK(hipMemset(d, 0, 8 * n));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const long long npairs = n / 2;
  const double dense = 33.0 * n, moved = 1.0 * n + 32.0 * 16.0 * live_lines;
  for (int pred = 0; pred < 2; ++pred)
    for (int blocks : {2048, 4096, 8192}) {
      float best = 1e30f;
      for (int rep = 0; rep < 6; ++rep) {
        CK(hipEventRecord(e0));
        if (pred) hipLaunchKernelGGL((k_mix<true, 4>), dim3(blocks), dim3(256), 0, 0, code, a, b, c, d, 0.5, npairs);
        else hipLaunchKernelGGL((k_mix<false, 4>), dim3(blocks), dim3(256), 0, 0, code, a, b, c, d, 0.5, npairs);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
      }
      printf("%s blocks %5d: %.3f ms  dense %.0f GB/s  line-granular moved %.0f GB/s\n", pred ? "predicated  " : "unpredicated",
             blocks, best, dense / best / 1e6, (pred ? moved : dense) / best / 1e6);
    }
  return 0;
}
