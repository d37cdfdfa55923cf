// Streaming micro-benchmark used to pick the access pattern of the CG vector kernels:
// r = r - a*q (read 16 B, write 8 B per element) over n doubles, several variants.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/stream_bench tools/stream_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int U, bool NT>
__global__ void __launch_bounds__(256) k_triad(double* __restrict__ r, const double* __restrict__ q, double a, long long n2) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  for (; t + (U - 1) * stride < n2; t += U * stride) {
    double2 rv[U], qv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (NT) {
        rv[u].x = __builtin_nontemporal_load(&r[2 * (t + u * stride)]);
        rv[u].y = __builtin_nontemporal_load(&r[2 * (t + u * stride) + 1]);
        qv[u].x = __builtin_nontemporal_load(&q[2 * (t + u * stride)]);
        qv[u].y = __builtin_nontemporal_load(&q[2 * (t + u * stride) + 1]);
      } else {
        rv[u] = *reinterpret_cast<const double2*>(r + 2 * (t + u * stride));
        qv[u] = *reinterpret_cast<const double2*>(q + 2 * (t + u * stride));
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      double2 o = make_double2(rv[u].x - a * qv[u].x, rv[u].y - a * qv[u].y);
      if (NT) {
        __builtin_nontemporal_store(o.x, &r[2 * (t + u * stride)]);
        __builtin_nontemporal_store(o.y, &r[2 * (t + u * stride) + 1]);
      } else {
        *reinterpret_cast<double2*>(r + 2 * (t + u * stride)) = o;
      }
    }
  }
  for (; t < n2; t += stride) {
    double2 rv = *reinterpret_cast<const double2*>(r + 2 * t), qv = *reinterpret_cast<const double2*>(q + 2 * t);
    *reinterpret_cast<double2*>(r + 2 * t) = make_double2(rv.x - a * qv.x, rv.y - a * qv.y);
  }
}

// blocked: each block owns one contiguous chunk
template <int U>
__global__ void __launch_bounds__(256) k_triad_blocked(double* __restrict__ r, const double* __restrict__ q, double a, long long n2) {
  const long long per = (n2 + gridDim.x - 1) / gridDim.x;
  const long long b0 = per * blockIdx.x, b1 = b0 + per < n2 ? b0 + per : n2;
  long long t = b0 + threadIdx.x;
  for (; t + (U - 1) * 256 < b1; t += U * 256) {
    double2 rv[U], qv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      rv[u] = *reinterpret_cast<const double2*>(r + 2 * (t + u * 256));
      qv[u] = *reinterpret_cast<const double2*>(q + 2 * (t + u * 256));
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      *reinterpret_cast<double2*>(r + 2 * (t + u * 256)) = make_double2(rv[u].x - a * qv[u].x, rv[u].y - a * qv[u].y);
  }
  for (; t < b1; t += 256) {
    double2 rv = *reinterpret_cast<const double2*>(r + 2 * t), qv = *reinterpret_cast<const double2*>(q + 2 * t);
    *reinterpret_cast<double2*>(r + 2 * t) = make_double2(rv.x - a * qv.x, rv.y - a * qv.y);
  }
}

__global__ void __launch_bounds__(256) k_copy(double* __restrict__ d, const double* __restrict__ s, long long n2) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n2; t += stride)
    *reinterpret_cast<double2*>(d + 2 * t) = *reinterpret_cast<const double2*>(s + 2 * t);
}

__global__ void __launch_bounds__(256) k_read(const double* __restrict__ s, double* out, long long n2) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  double acc = 0;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n2; t += stride) {
    double2 v = *reinterpret_cast<const double2*>(s + 2 * t);
    acc += v.x + v.y;
  }
  if (acc == 123.456) out[0] = acc;
}

__global__ void __launch_bounds__(256) k_write(double* __restrict__ d, long long n2) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n2; t += stride)
    *reinterpret_cast<double2*>(d + 2 * t) = make_double2(1.0, 2.0);
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
  const long long n = 1LL << 30;  // doubles per array (8 GiB)
  const long long n2 = n / 2;
  double *r, *q;
  CK(hipMalloc(&r, n * 8)); CK(hipMalloc(&q, n * 8));
  CK(hipMemset(r, 0, n * 8)); CK(hipMemset(q, 0, n * 8));
  const int reps = 5;
  for (int blocks : {1024, 2048, 4096, 8192, 16384, 65536}) {
    double ms;
    ms = timeit([&] { hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, 0, r, q, n2); }, reps);
    printf("blocks %6d copy        %7.3f ms %7.1f GB/s\n", blocks, ms, 16.0 * n / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL(k_read, dim3(blocks), dim3(256), 0, 0, q, r, n2); }, reps);
    printf("blocks %6d read        %7.3f ms %7.1f GB/s\n", blocks, ms, 8.0 * n / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL(k_write, dim3(blocks), dim3(256), 0, 0, r, n2); }, reps);
    printf("blocks %6d write       %7.3f ms %7.1f GB/s\n", blocks, ms, 8.0 * n / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL((k_triad<1, false>), dim3(blocks), dim3(256), 0, 0, r, q, 0.5, n2); }, reps);
    printf("blocks %6d triad U1    %7.3f ms %7.1f GB/s\n", blocks, ms, 24.0 * n / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL((k_triad<2, false>), dim3(blocks), dim3(256), 0, 0, r, q, 0.5, n2); }, reps);
    printf("blocks %6d triad U2    %7.3f ms %7.1f GB/s\n", blocks, ms, 24.0 * n / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL((k_triad<4, false>), dim3(blocks), dim3(256), 0, 0, r, q, 0.5, n2); }, reps);
    printf("blocks %6d triad U4    %7.3f ms %7.1f GB/s\n", blocks, ms, 24.0 * n / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL((k_triad<2, true>), dim3(blocks), dim3(256), 0, 0, r, q, 0.5, n2); }, reps);
    printf("blocks %6d triad U2 nt %7.3f ms %7.1f GB/s\n", blocks, ms, 24.0 * n / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL((k_triad_blocked<4>), dim3(blocks), dim3(256), 0, 0, r, q, 0.5, n2); }, reps);
    printf("blocks %6d triad blk4  %7.3f ms %7.1f GB/s\n", blocks, ms, 24.0 * n / ms / 1e6);
    fflush(stdout);
  }
  return 0;
}
