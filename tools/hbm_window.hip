// Where do the timing modes of the CG kernels come from?  Probe of the HBM address mapping: read bandwidth of a moving
// window of W bytes (larger than the 256 MB Infinity Cache) against its offset in one large allocation, and of TWO
// windows read together against their distance.  hipcc --offload-arch=gfx950 -O3 -o hbm_window hbm_window.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef unsigned int u4 __attribute__((ext_vector_type(4)));

// nwin windows of `bytes` each (16-byte units); every block walks its share of every window at the same pace
__global__ __launch_bounds__(256) void k_read(const char* base, const long long* off, int nwin, long long units, unsigned* sink) {
  const long long stride = (long long)gridDim.x * 256;
  u4 acc = {0, 0, 0, 0};
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < units; i += stride)
    for (int w = 0; w < nwin; ++w) {
      const u4 v = __builtin_nontemporal_load((const u4*)(base + off[w]) + i);
      acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
    }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) *sink = 1;
}

// the same, but the windows are swept like the CG kernel sweeps z: block b owns a contiguous piece (so the set of
// addresses in flight at one time is `resident blocks` pieces spread over the window, not one dense front)
__global__ __launch_bounds__(256) void k_read_pieces(const char* base, const long long* off, int nwin, long long units, unsigned* sink) {
  const long long per = units / gridDim.x;
  const long long lo = (long long)blockIdx.x * per;
  u4 acc = {0, 0, 0, 0};
  for (long long i = lo + threadIdx.x; i < lo + per; i += 256)
    for (int w = 0; w < nwin; ++w) {
      const u4 v = __builtin_nontemporal_load((const u4*)(base + off[w]) + i);
      acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
    }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) *sink = 1;
}

// first nr windows read, the other nwin - nr written (the CG kernels read 3 vectors and write 3 at the same offsets)
__global__ __launch_bounds__(256) void k_rw(char* base, const long long* off, int nr, int nwin, long long units, int pieces) {
  long long lo, hi, step;
  if (pieces) { const long long per = units / gridDim.x; lo = (long long)blockIdx.x * per + threadIdx.x; hi = (long long)(blockIdx.x + 1) * per; step = 256; }
  else { lo = (long long)blockIdx.x * 256 + threadIdx.x; hi = units; step = (long long)gridDim.x * 256; }
  for (long long i = lo; i < hi; i += step) {
    u4 acc = {1, 2, 3, 4};
    for (int w = 0; w < nr; ++w) {
      const u4 v = __builtin_nontemporal_load((const u4*)(base + off[w]) + i);
      acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
    }
    for (int w = nr; w < nwin; ++w) __builtin_nontemporal_store(acc, (u4*)(base + off[w]) + i);
  }
}

// half-empty wave instructions: every lane group of 4 (one 64-byte line) takes part or not by `pattern` (bit g of the
// 16-bit pattern, rotated per wave-row so that memory is still covered evenly); useful bytes only are counted.
// mode 0: the inactive lanes are switched off by a branch; 1: they get an out-of-range buffer offset
__global__ __launch_bounds__(256) void k_read_sparse(const char* base, long long units, unsigned pattern, int mode, unsigned* sink) {
  const int lane = threadIdx.x & 63;
  const long long stride = (long long)gridDim.x * 256;
  u4 acc = {0, 0, 0, 0};
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fff0000, 0x00020000);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < units; i += stride) {
    const unsigned rot = (unsigned)((i >> 6) * 5) & 15;
    const unsigned pat = ((pattern >> rot) | (pattern << (16 - rot))) & 0xffffu;
    const bool on = (pat >> (lane >> 2)) & 1;
    u4 v = {0, 0, 0, 0};
    if (mode == 0) {
      if (on) v = __builtin_nontemporal_load((const u4*)base + i);
    } else {
      const char* wbase = base + ((i >> 6) << 10);  // this wave-row's 1 KiB (uniform), lane offset below
      rs = __builtin_amdgcn_make_buffer_rsrc((void*)__builtin_amdgcn_readfirstlane((unsigned long long)wbase & 0xffffffffu) == 0 ? (void*)wbase : (void*)wbase, 0, 1024, 0x00020000);
      v = __builtin_amdgcn_raw_buffer_load_b128(rs, on ? lane * 16 : 0x7ffff000, 0, 0);
    }
    acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) *sink = 1;
}

static char* big;
static long long* d_off;
static unsigned* sink;

static double bw(int pieces, const std::vector<long long>& off, long long bytes, int reps) {
  CK(hipMemcpy(d_off, off.data(), off.size() * 8, hipMemcpyHostToDevice));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int blocks = pieces ? 4096 : 2048;
  auto go = [&] {
    if (pieces) hipLaunchKernelGGL(k_read_pieces, dim3(blocks), dim3(256), 0, 0, big, d_off, (int)off.size(), bytes / 16, sink);
    else hipLaunchKernelGGL(k_read, dim3(blocks), dim3(256), 0, 0, big, d_off, (int)off.size(), bytes / 16, sink);
  };
  go();
  CK(hipEventRecord(a));
  for (int r = 0; r < reps; ++r) go();
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
  return (double)bytes * off.size() * reps / (ms * 1e-3) / 1e12;
}

static double bw_rw(int pieces, int nr, const std::vector<long long>& off, long long bytes, int reps) {
  CK(hipMemcpy(d_off, off.data(), off.size() * 8, hipMemcpyHostToDevice));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  auto go = [&] { hipLaunchKernelGGL(k_rw, dim3(pieces ? 4096 : 2048), dim3(256), 0, 0, big, d_off, nr, (int)off.size(), bytes / 16, pieces); };
  go();
  CK(hipEventRecord(a));
  for (int r = 0; r < reps; ++r) go();
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
  return (double)bytes * off.size() * reps / (ms * 1e-3) / 1e12;
}

int main(int argc, char** argv) {
  const long long GiB = 1ll << 30, MiB = 1ll << 20;
  const long long total = (argc > 1 ? atoll(argv[1]) : 200) * GiB;
  CK(hipMalloc(&big, total)); CK(hipMalloc(&d_off, 64 * 8)); CK(hipMalloc(&sink, 4));
  CK(hipMemset(big, 1, total)); CK(hipDeviceSynchronize());
  printf("arena of %lld GiB at %p\n", total / GiB, (void*)big);
  if (argc > 2 && argv[2][0] == 'd') {  // copy bandwidth against the distance, finely
    const long long step = (argc > 3 ? atoll(argv[3]) : 256) * MiB, w = 512 * MiB;
    printf("-- copy of a 512 MiB window at 0 to the window at distance D = m x %lld MiB: TB/s, 16 values per line\n", step / MiB);
    for (long long m = (w + step - 1) / step; (m * step + w) <= total; ++m) {
      if ((m % 16) == 0 || m == (w + step - 1) / step) printf("\n  D %7.2f GiB:", (double)(m * step) / GiB);
      printf(" %.2f", bw_rw(0, 1, {0, m * step}, w, 6));
      fflush(stdout);
    }
    printf("\n");
    return 0;
  }
  if (argc > 2 && argv[2][0] == 's') {  // sparse wave instructions
    printf("-- read of a 16 GiB window, 64-byte lines taking part by pattern: useful TB/s (lines fetched / all lines)\n");
    for (unsigned pattern : {0xffffu, 0xfff0u, 0xff00u, 0xf0f0u, 0xaaaau, 0xccccu, 0x8888u, 0x0101u}) {
      const int live = __builtin_popcount(pattern);
      for (int mode = 0; mode < 1; ++mode) {
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        const long long units = 16 * GiB / 16;
        auto go = [&] { hipLaunchKernelGGL(k_read_sparse, dim3(2048), dim3(256), 0, 0, big, units, pattern, mode, sink); };
        go();
        CK(hipEventRecord(a));
        for (int r = 0; r < 4; ++r) go();
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("  pattern %04x (%2d of 16 lines)  %s  %7.3f ms  useful %.3f TB/s\n", pattern, live, mode ? "range-checked" : "branch       ",
               ms / 4, 16.0 * GiB * live / 16 * 4 / (ms * 1e-3) / 1e12);
      }
    }
    return 0;
  }
  if (argc > 2 && argv[2][0] == 'b') {  // which address bits matter between a read and a written stream?
    printf("-- copy of a 512 MiB window at O to O + 2^k (+ 512 MiB when they would overlap), TB/s\n        ");
    const long long Os[4] = {0, 5 * GiB, 77 * GiB + 300 * MiB, 130 * GiB};
    for (long long o : Os) printf("   O=%-6.1f", (double)o / GiB);
    printf(" GiB\n");
    for (int k = 12; k <= 37; ++k) {
      const long long d = 1ll << k, w = 512 * MiB;
      printf("  k %2d  ", k);
      for (long long o : Os) {
        const long long dst = d >= w ? o + d : o + w + d;  // small k: the window after, shifted by 2^k
        if (dst + w > total) { printf("     -    "); continue; }
        printf("  %7.3f ", bw_rw(0, 1, {o, dst}, w, 16));
      }
      printf("\n");
    }
    printf("-- the same with 3 read windows at O, O+s, O+2s and 3 written at O+D, O+D+s, O+D+2s;  s = 8.016 GiB\n");
    const long long st = 8208 * MiB;
    for (long long dd : {24624 * MiB /* = 3 s: the plain layout */, 32 * GiB, 32832 * MiB /* 4 s */, 36 * GiB, 48 * GiB, 64 * GiB, 96 * GiB}) {
      printf("  D %7.3f GiB ", (double)dd / GiB);
      for (long long o : {0ll, 5 * GiB, 20 * GiB, 41 * GiB}) {
        std::vector<long long> off = {o, o + st, o + 2 * st, o + dd, o + dd + st, o + dd + 2 * st};
        if (off.back() + GiB > total) { printf("     -    "); continue; }
        printf("  %7.3f ", bw_rw(0, 3, off, GiB, 8));
      }
      printf("\n");
    }
    return 0;
  }
  if (argc > 2) {  // read/write mixes
    printf("-- copy of a 1 GiB window at 0 to a window at distance D (dense front | block pieces), TB/s read + written\n");
    for (long long d : {1, 2, 3, 4, 6, 8, 12, 16, 20, 24, 28, 32, 36, 40, 48, 56, 64, 72, 80, 96, 112, 128, 160, 190})
      if ((d + 1) * GiB <= total) printf("  D %4lld GiB  %.3f | %.3f\n", d, bw_rw(0, 1, {0, d * GiB}, GiB, 16), bw_rw(1, 1, {0, d * GiB}, GiB, 16));
    printf("-- copy of a whole 8 GiB array to the array at distance D\n");
    for (long long d : {8, 9, 12, 16, 24, 32, 40, 64, 96, 128, 160})
      if ((d + 8) * GiB <= total) printf("  D %4lld GiB  %.3f | %.3f\n", d, bw_rw(0, 1, {0, d * GiB}, 8 * GiB, 3), bw_rw(1, 1, {0, d * GiB}, 8 * GiB, 3));
    printf("-- 3 read + 3 written windows of 1 GiB at base + k * 8.016 GiB against the base\n");
    const long long st = 8208 * MiB;
    for (long long b = 0; b + 5 * st + GiB <= total; b += 6 * GiB) {
      std::vector<long long> off;
      for (int k = 0; k < 6; ++k) off.push_back(b + k * st);
      printf("  base %4lld GiB  %.3f | %.3f\n", b / GiB, bw_rw(0, 3, off, GiB, 8), bw_rw(1, 3, off, GiB, 8));
    }
    printf("-- the same at base 0 against the stride\n");
    for (long long sx : {8208 * MiB, 8464 * MiB, 9 * GiB, 10 * GiB, 12 * GiB, 16 * GiB, 20 * GiB, 24 * GiB, 28 * GiB, 32 * GiB, 36 * GiB}) {
      std::vector<long long> off;
      for (int k = 0; k < 6; ++k) off.push_back(k * sx);
      if (off.back() + GiB > total) break;
      printf("  stride %6lld MiB  %.3f | %.3f\n", sx / MiB, bw_rw(0, 3, off, GiB, 8), bw_rw(1, 3, off, GiB, 8));
    }
    printf("-- whole-array 3 read + 3 written (8 GiB each) at base + k * 8.016 GiB against the base\n");
    for (long long b = 0; b + 6 * st <= total; b += 12 * GiB) {
      std::vector<long long> off;
      for (int k = 0; k < 6; ++k) off.push_back(b + k * st);
      printf("  base %4lld GiB  %.3f | %.3f\n", b / GiB, bw_rw(0, 3, off, 8 * GiB, 2), bw_rw(1, 3, off, 8 * GiB, 2));
    }
    return 0;
  }
  printf("-- one window at offset 0, dense front: TB/s against the window size\n");
  for (long long w : {64 * MiB, 512 * MiB, 1 * GiB, 2 * GiB, 4 * GiB, 8 * GiB, 16 * GiB, 64 * GiB}) {
    const int reps = (int)(64 * GiB / w) < 2 ? 2 : (int)(64 * GiB / w);
    printf("  W %6lld MiB  %.3f\n", w / MiB, bw(0, {0}, w, reps));
  }
  printf("-- one window of 1 GiB against its offset (dense front | block pieces)\n");
  for (long long o = 0; o + GiB <= total; o += 8 * GiB) printf("  O %4lld GiB  %.3f | %.3f\n", o / GiB, bw(0, {o}, GiB, 32), bw(1, {o}, GiB, 32));
  printf("-- fine: one window of 256 MiB... skipped (Infinity Cache)\n");
  printf("-- two windows of 1 GiB, first at 0, against their distance\n");
  for (long long d : {1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 14, 16, 20, 24, 28, 32, 40, 48, 56, 64, 96, 128})
    if ((d + 1) * GiB <= total) printf("  D %4lld GiB  %.3f | %.3f\n", d, bw(0, {0, d * GiB}, GiB, 16), bw(1, {0, d * GiB}, GiB, 16));
  printf("-- six windows of 1 GiB at base + k * 8.016 GiB (the CG layout) against the base\n");
  const long long st = 8208 * MiB;
  for (long long b = 0; b + 5 * st + GiB <= total; b += 12 * GiB) {
    std::vector<long long> off;
    for (int k = 0; k < 6; ++k) off.push_back(b + k * st);
    printf("  base %4lld GiB  %.3f | %.3f\n", b / GiB, bw(0, off, GiB, 8), bw(1, off, GiB, 8));
  }
  printf("-- six windows of 128 MiB (one resident wave of blocks' worth) at base + k * 8.016 GiB, sliding 1 GiB through the arrays\n");
  for (long long b : {0ll, 36 * GiB, 120 * GiB})
    for (long long t = 0; t < 8 * GiB; t += 2 * GiB) {
      std::vector<long long> off;
      for (int k = 0; k < 6; ++k) off.push_back(b + k * st + t);
      if (off.back() + GiB > total) break;
      printf("  base %4lld GiB  t %lld GiB  %.3f\n", b / GiB, t / GiB, bw(0, off, 128 * MiB, 64));
    }
  return 0;
}
