// Can the vector arenas of the placement search be backed by the HIP virtual-memory API, so that the slots the search
// does not choose are RELEASED afterwards?  Probe of what that costs and whether the physical layout behaves like hipMalloc's:
//   1. support / granularity; time of hipMalloc + hipFree and hipMemCreate + hipMemRelease for large sizes (the spacer);
//   2. an arena built from chunks of G bytes created back to back and mapped into one reserved range: creation, mapping,
//      access, memset times; a copy between two windows of it against the same copy inside a hipMalloc allocation
//      (page-table fragment effects); the distance sweep of tools/hbm_window.hip (copy bandwidth against the distance of
//      source and destination) on both kinds of arena: the same structure = the chunks lie in allocation order;
//   3. unmap + release of some chunks: hipMemGetInfo must report them free while the rest stays usable.
// hipcc --offload-arch=gfx950 -O3 -o vmm_probe vmm_probe.hip ; ./vmm_probe [arena GiB] [chunk MiB]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef unsigned int u4 __attribute__((ext_vector_type(4)));
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ __launch_bounds__(256) void k_copy(char* dst, const char* src, long long units) {
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < units; i += stride)
    __builtin_nontemporal_store(__builtin_nontemporal_load((const u4*)src + i), (u4*)dst + i);
}

static double copy_tbs(char* dst, const char* src, size_t bytes, hipStream_t s) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const long long units = (long long)(bytes / 16);
  hipLaunchKernelGGL(k_copy, dim3(256 * 8), dim3(256), 0, s, dst, src, units);
  CK(hipEventRecord(a, s));
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k_copy, dim3(256 * 8), dim3(256), 0, s, dst, src, units);
  CK(hipEventRecord(b, s));
  CK(hipEventSynchronize(b));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, a, b));
  CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
  return 3.0 * 2.0 * (double)bytes / (ms * 1e-3) / 1e12;
}

static size_t free_mib() { size_t f = 0, t = 0; CK(hipMemGetInfo(&f, &t)); return f >> 20; }

int main(int argc, char** argv) {
  const size_t arena = (size_t)(argc > 1 ? atof(argv[1]) : 24.0) << 30;
  const size_t G = (size_t)(argc > 2 ? atoi(argv[2]) : 128) << 20;
  int dev = 0, vmm = 0;
  CK(hipSetDevice(dev));
  CK(hipDeviceGetAttribute(&vmm, hipDeviceAttributeVirtualMemoryManagementSupported, dev));
  hipMemAllocationProp prop{};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = dev;
  size_t gmin = 0, grec = 0;
  CK(hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum));
  CK(hipMemGetAllocationGranularity(&grec, &prop, hipMemAllocationGranularityRecommended));
  printf("virtual memory management supported: %d, granularity min %zu recommended %zu; free %zu MiB\n", vmm, gmin, grec, free_mib());
  if (!vmm) return 0;
  hipStream_t s;
  CK(hipStreamCreate(&s));

  // 1. large allocations, both ways
  for (size_t gb : {8, 48, 48}) {
    double t0 = now();
    void* p = nullptr;
    CK(hipMalloc(&p, gb << 30));
    double t1 = now();
    CK(hipFree(p));
    double t2 = now();
    hipMemGenericAllocationHandle_t h;
    CK(hipMemCreate(&h, gb << 30, &prop, 0));
    double t3 = now();
    CK(hipMemRelease(h));
    double t4 = now();
    printf("%2zu GiB: hipMalloc %.1f ms, hipFree %.1f ms | hipMemCreate %.1f ms, hipMemRelease %.1f ms\n", gb, 1e3 * (t1 - t0), 1e3 * (t2 - t1),
           1e3 * (t3 - t2), 1e3 * (t4 - t3));
  }

  // 2. an arena of chunks
  const size_t nchunk = arena / G;
  std::vector<hipMemGenericAllocationHandle_t> hs(nchunk);
  char* va = nullptr;
  double t0 = now();
  CK(hipMemAddressReserve((void**)&va, nchunk * G, 0, nullptr, 0));
  double t1 = now();
  for (size_t c = 0; c < nchunk; ++c) CK(hipMemCreate(&hs[c], G, &prop, 0));
  double t2 = now();
  for (size_t c = 0; c < nchunk; ++c) CK(hipMemMap(va + c * G, G, 0, hs[c], 0));
  double t3 = now();
  hipMemAccessDesc acc{};
  acc.location.type = hipMemLocationTypeDevice;
  acc.location.id = dev;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  CK(hipMemSetAccess(va, nchunk * G, &acc, 1));
  double t4 = now();
  CK(hipMemsetAsync(va, 0, nchunk * G, s));
  CK(hipStreamSynchronize(s));
  double t5 = now();
  printf("arena of %zu chunks x %zu MiB: reserve %.1f ms, create %.1f ms, map %.1f ms, set access %.1f ms, memset %.1f ms; free %zu MiB\n", nchunk,
         G >> 20, 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), 1e3 * (t4 - t3), 1e3 * (t5 - t4), free_mib());
  char* ma = nullptr;
  double t6 = now();
  CK(hipMalloc((void**)&ma, nchunk * G));
  double t7 = now();
  CK(hipMemsetAsync(ma, 0, nchunk * G, s));
  CK(hipStreamSynchronize(s));
  printf("hipMalloc of the same size: %.1f ms, memset %.1f ms\n", 1e3 * (t7 - t6), 1e3 * (now() - t7));
  // copy bandwidth of a 1 GiB window against the distance of its destination, both arenas
  const size_t W = (size_t)1 << 30;
  printf("distance GiB   chunked TB/s   hipMalloc TB/s\n");
  for (size_t d = W; d + W <= nchunk * G; d += (d < ((size_t)8 << 30) ? W : (size_t)2 << 30)) {
    const double a = copy_tbs(va + d, va, W, s), b = copy_tbs(ma + d, ma, W, s);
    printf("%8.2f %12.3f %12.3f\n", d / 1073741824.0, a, b);
  }
  // fine structure: period 128 MiB
  printf("distance MiB (+2 GiB)   chunked TB/s   hipMalloc TB/s\n");
  for (size_t d = (size_t)2 << 30; d < ((size_t)2 << 30) + ((size_t)320 << 20); d += (size_t)32 << 20) {
    const double a = copy_tbs(va + d, va, W, s), b = copy_tbs(ma + d, ma, W, s);
    printf("%8zu %12.3f %12.3f\n", (d - ((size_t)2 << 30)) >> 20, a, b);
  }
  CK(hipFree(ma));

  // 3. release every second quarter
  const size_t before = free_mib();
  double t8 = now();
  size_t released = 0;
  for (size_t c = 0; c < nchunk; ++c)
    if ((c * 4 / nchunk) & 1) { CK(hipMemUnmap(va + c * G, G)); CK(hipMemRelease(hs[c])); hs[c] = nullptr; released += G; }
  double t9 = now();
  printf("unmap + release of %zu MiB: %.1f ms; free memory %zu -> %zu MiB\n", released >> 20, 1e3 * (t9 - t8), before, free_mib());
  printf("copy inside the first quarter after the release: %.3f TB/s\n", copy_tbs(va + (nchunk / 8) * G, va, std::min(W, (nchunk / 8) * G), s));
  for (size_t c = 0; c < nchunk; ++c)
    if (hs[c]) { CK(hipMemUnmap(va + c * G, G)); CK(hipMemRelease(hs[c])); }
  CK(hipMemAddressFree(va, nchunk * G));
  printf("all released: free %zu MiB\n", free_mib());
  return 0;
}
