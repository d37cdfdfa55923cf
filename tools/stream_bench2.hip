// Write-path micro-benchmark: which store flavour / shape streams fastest on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

typedef double d2 __attribute__((ext_vector_type(2)));

template <int MODE>  // 0 plain, 1 nontemporal builtin on vector, 2 two consecutive 16B per thread, 3 blocked rows
__global__ void __launch_bounds__(256) k_copy(double* __restrict__ d, const double* __restrict__ s, long long n2) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  if (MODE == 2) {
    // each wave covers 2 KiB contiguous per iteration: lane handles 16 B at +0 and +1024
    const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    const long long nw = stride >> 6;
    for (long long w = wave; w * 128 + 64 + lane < n2; w += nw) {
      const long long t = w * 128 + lane;
      d2 a = *reinterpret_cast<const d2*>(s + 2 * t), b = *reinterpret_cast<const d2*>(s + 2 * (t + 64));
      __builtin_nontemporal_store(a, reinterpret_cast<d2*>(d + 2 * t));
      __builtin_nontemporal_store(b, reinterpret_cast<d2*>(d + 2 * (t + 64)));
    }
    return;
  }
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n2; t += stride) {
    if (MODE == 0) {
      *reinterpret_cast<d2*>(d + 2 * t) = *reinterpret_cast<const d2*>(s + 2 * t);
    } else if (MODE == 1) {
      d2 v = __builtin_nontemporal_load(reinterpret_cast<const d2*>(s + 2 * t));
      __builtin_nontemporal_store(v, reinterpret_cast<d2*>(d + 2 * t));
    } else {
      d2 v = *reinterpret_cast<const d2*>(s + 2 * t);
      __builtin_nontemporal_store(v, reinterpret_cast<d2*>(d + 2 * t));
    }
  }
}

template <int MODE>
__global__ void __launch_bounds__(256) k_write(double* __restrict__ d, long long n2) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  d2 v = {1.0, 2.0};
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n2; t += stride) {
    if (MODE == 0) *reinterpret_cast<d2*>(d + 2 * t) = v;
    else __builtin_nontemporal_store(v, reinterpret_cast<d2*>(d + 2 * t));
  }
}

// 17 B/voxel-like pattern: read 1 array + write 1 array (+ 1 byte array), y = 2*x
template <int MODE>
__global__ void __launch_bounds__(256) k_scale(double* __restrict__ d, const double* __restrict__ s, const unsigned char* __restrict__ c, long long n2) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n2; t += stride) {
    d2 v = *reinterpret_cast<const d2*>(s + 2 * t);
    unsigned short cc = *reinterpret_cast<const unsigned short*>(c + 2 * t);
    d2 o = {(cc & 0xff) * v.x, (cc >> 8) * v.y};
    if (MODE == 0) *reinterpret_cast<d2*>(d + 2 * t) = o;
    else __builtin_nontemporal_store(o, reinterpret_cast<d2*>(d + 2 * t));
  }
}

template <typename F>
double timeit(F f, int reps) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f();
  CK(hipEventRecord(a));
  for (int i = 0; i < reps; ++i) f();
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms / reps;
}

int main() {
  const long long n = 1LL << 30;
  const long long n2 = n / 2;
  double *r, *q; unsigned char* c;
  CK(hipMalloc(&r, n * 8)); CK(hipMalloc(&q, n * 8)); CK(hipMalloc(&c, n));
  CK(hipMemset(r, 0, n * 8)); CK(hipMemset(q, 0, n * 8)); CK(hipMemset(c, 1, n));
  const int reps = 5;
  for (int blocks : {1024, 2048, 8192, 65536}) {
    double ms;
#define RUN(name, bytes, ...) ms = timeit([&] { __VA_ARGS__; }, reps); printf("blocks %6d %-16s %7.3f ms %7.1f GB/s\n", blocks, name, ms, bytes * n / ms / 1e6);
    RUN("copy plain", 16.0, hipLaunchKernelGGL((k_copy<0>), dim3(blocks), dim3(256), 0, 0, r, q, n2))
    RUN("copy nt ld+st", 16.0, hipLaunchKernelGGL((k_copy<1>), dim3(blocks), dim3(256), 0, 0, r, q, n2))
    RUN("copy nt st", 16.0, hipLaunchKernelGGL((k_copy<3>), dim3(blocks), dim3(256), 0, 0, r, q, n2))
    RUN("copy 2x16 nt", 16.0, hipLaunchKernelGGL((k_copy<2>), dim3(blocks), dim3(256), 0, 0, r, q, n2))
    RUN("write plain", 8.0, hipLaunchKernelGGL((k_write<0>), dim3(blocks), dim3(256), 0, 0, r, n2))
    RUN("write nt", 8.0, hipLaunchKernelGGL((k_write<1>), dim3(blocks), dim3(256), 0, 0, r, n2))
    RUN("scale17 plain", 17.0, hipLaunchKernelGGL((k_scale<0>), dim3(blocks), dim3(256), 0, 0, r, q, c, n2))
    RUN("scale17 nt", 17.0, hipLaunchKernelGGL((k_scale<1>), dim3(blocks), dim3(256), 0, 0, r, q, c, n2))
    fflush(stdout);
  }
  // hipMemcpy D2D as the runtime's own copy
  double ms = timeit([&] { CK(hipMemcpyAsync(r, q, n * 8, hipMemcpyDeviceToDevice, 0)); }, reps);
  printf("hipMemcpyDtoD            %7.3f ms %7.1f GB/s\n", ms, 16.0 * n / ms / 1e6);
  return 0;
}
