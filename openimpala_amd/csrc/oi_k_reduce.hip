// oi_k_reduce.hip -- final sums of the per-block partials, scalar records of the CG variants, stop test, rank sums.  See oi_kern.h.
#include "oi_kern.h"

namespace oi {

// after the 5 sums of kernel k are in rec[k] (all-reduced in multi-rank runs): convergence
// test on the exact ||r_k||, alpha_k, and beta_{k+1} from the one-step extrapolation
__device__ __forceinline__ void cg1_scalars(double* rk, Ctrl* ctrl, int k) {
  const Cg1Scalars sc = cg1_scalars_of(rk, ctrl->tol2);
  if (!ctrl->done && sc.converged) { ctrl->done = k + 1; }
  rk[R1_ALPHA] = sc.alpha;
  rk[R1_BETA] = sc.beta;
}

// three waves per component (15 waves): lane L = 0 .. 191 of a component adds partials L, L + 192, ... with four
// independent running sums (the loads of a round are in flight together: the descriptor-addressed CG kernel launches
// 32768 blocks at 1024^3 and one wave per component took 0.2 ms over them), then a fixed tree -- the order of
// summation depends on the number of blocks only, never on timing
__global__ void __launch_bounds__(960) k_finalize1(const double* __restrict__ partials, int nblocks, double* rk,
                                                    Ctrl* ctrl, int k, int with_scalars) {
  if (ctrl->done) return;
  __shared__ double part[15];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int c = wv % 5, L = (wv / 5) * 64 + lane;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  int b = L;
  for (; b + 576 < nblocks; b += 768) {
    a0 += partials[(long long)b * 5 + c];
    a1 += partials[(long long)(b + 192) * 5 + c];
    a2 += partials[(long long)(b + 384) * 5 + c];
    a3 += partials[(long long)(b + 576) * 5 + c];
  }
  for (; b < nblocks; b += 192) a0 += partials[(long long)b * 5 + c];
  const double v = wave_sum((a0 + a1) + (a2 + a3));
  if (lane == 0) part[wv] = v;
  __syncthreads();
  if (threadIdx.x < 5) rk[threadIdx.x] = (part[threadIdx.x] + part[threadIdx.x + 5]) + part[threadIdx.x + 10];
  __syncthreads();
  if (threadIdx.x == 0 && with_scalars) cg1_scalars(rk, ctrl, k);
}

// The same sums by kFinSlices blocks (large grids): block j adds the records of slice j -- wave c component c, four running sums
// per lane -- and leaves five slice sums in `scratch`; the block that draws the last ticket adds the slices in slice order.  Which
// block that is depends on timing, what it computes does not: the order of summation is fixed by (nblocks, kFinSlices).
// scratch: 5 * kFinSlices doubles + the ticket (zero between launches: the last block resets it).
__global__ void __launch_bounds__(320) k_finalize1_sliced(const double* __restrict__ partials, int nblocks, double* scratch, double* rk,
                                                           Ctrl* ctrl, int k, int with_scalars) {
  if (ctrl->done) return;  // (set by the CG kernel before this launch, or by the last block below after every block has passed here)
  __shared__ int last;
  const int c = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int S = (nblocks + kFinSlices - 1) / kFinSlices;
  const int b0 = blockIdx.x * S, b1 = min(nblocks, b0 + S);
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  int b = b0 + lane;
  for (; b + 192 < b1; b += 256) {
    a0 += partials[(long long)b * 5 + c];
    a1 += partials[(long long)(b + 64) * 5 + c];
    a2 += partials[(long long)(b + 128) * 5 + c];
    a3 += partials[(long long)(b + 192) * 5 + c];
  }
  for (; b < b1; b += 64) a0 += partials[(long long)b * 5 + c];
  const double v = wave_sum((a0 + a1) + (a2 + a3));
  unsigned* ticket = reinterpret_cast<unsigned*>(scratch + 5 * kFinSlices);
  if (lane == 0) __hip_atomic_store(&scratch[blockIdx.x * 5 + c], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) last = atomicAdd(ticket, 1u) == (unsigned)kFinSlices - 1u;
  __syncthreads();
  if (!last) return;
  __threadfence();
  // (agent-scope loads: the slice sums come from other CUs)
  const double sl = lane < kFinSlices ? __hip_atomic_load(&scratch[lane * 5 + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
  static_assert(kFinSlices <= 64, "one slice per lane");
  const double t = wave_sum(sl);
  if (lane == 0) rk[c] = t;
  __syncthreads();
  if (threadIdx.x == 0) {
    *ticket = 0u;
    if (with_scalars) cg1_scalars(rk, ctrl, k);
  }
}

__global__ void k_scalars1(double* rk, Ctrl* ctrl, int k) {
  if (!ctrl->done) cg1_scalars(rk, ctrl, k);
}

void launch_finalize1(const double* partials, int nblocks, double* rec_k, Ctrl* ctrl, int k, int with_scalars,
                      hipStream_t s, double* scratch) {
  if (scratch && nblocks >= 4096) {
    hipLaunchKernelGGL(k_finalize1_sliced, dim3(kFinSlices), dim3(320), 0, s, partials, nblocks, scratch, rec_k, ctrl, k, with_scalars);
    return;
  }
  hipLaunchKernelGGL(k_finalize1, dim3(1), dim3(960), 0, s, partials, nblocks, rec_k, ctrl, k, with_scalars);
}
void launch_scalars1(double* rec_k, Ctrl* ctrl, int k, hipStream_t s) {
  hipLaunchKernelGGL(k_scalars1, dim3(1), dim3(1), 0, s, rec_k, ctrl, k);
}

// --------------------------------------------------------------------------------------
// Deterministic final sums: one block adds the per-block partials in a fixed order.
// --------------------------------------------------------------------------------------
// (1024 threads, four independent running sums per thread: the descriptor-addressed kernels of the two-kernel iteration launch
// 32768 blocks at 1024^3, and 256 threads with one running sum took 57 us over them, three times per multigrid-PCG iteration;
// the order of summation depends on the number of blocks only)
__global__ void __launch_bounds__(1024) k_finalize(const double* __restrict__ partials, int nblocks, int ncomp,
                                                    double* __restrict__ dst, Ctrl* ctrl, int check, int iter_done) {
  if (ctrl && ctrl->done) return;
  __shared__ double red[16];
  const int T = blockDim.x;
  for (int c = 0; c < ncomp; ++c) {
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int b = threadIdx.x;
    for (; b + 3 * T < nblocks; b += 4 * T) {
      a0 += partials[(long long)b * ncomp + c];
      a1 += partials[(long long)(b + T) * ncomp + c];
      a2 += partials[(long long)(b + 2 * T) * ncomp + c];
      a3 += partials[(long long)(b + 3 * T) * ncomp + c];
    }
    for (; b < nblocks; b += T) a0 += partials[(long long)b * ncomp + c];
    const double t = block_sum((a0 + a1) + (a2 + a3), red);
    if (threadIdx.x == 0) {
      dst[c] = t;
      if (check && c == 0 && t <= ctrl->tol2) ctrl->done = iter_done;
    }
  }
}

void launch_finalize(const double* partials, int nblocks, int ncomp, double* dst, Ctrl* ctrl, int check,
                     int iter_done, hipStream_t s) {
  // (small grids: a block of 256 has the partials in flight just as well)
  hipLaunchKernelGGL(k_finalize, dim3(1), dim3(nblocks > 4096 ? 1024 : 256), 0, s, partials, nblocks, ncomp, dst, ctrl, check, iter_done);
}

__global__ void k_check_done(const double* __restrict__ rr, Ctrl* ctrl, int iter_done) {
  if (!ctrl->done && *rr <= ctrl->tol2) ctrl->done = iter_done;
}

void launch_check_done(const double* rr, Ctrl* ctrl, int iter_done, hipStream_t s) {
  hipLaunchKernelGGL(k_check_done, dim3(1), dim3(1), 0, s, rr, ctrl, iter_done);
}

// dst[i] = sum over sources of srcs[s][i]  (in-process "all-reduce" of the local transport)
__global__ void k_sum_scalars(double* dst, const double* const* srcs, int nsrc, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    double v = 0.0;
    for (int s = 0; s < nsrc; ++s) v += srcs[s][i];
    dst[i] = v;
  }
}

void launch_sum_scalars(double* dst, const double* const* srcs, int nsrc, int n, hipStream_t s) {
  hipLaunchKernelGGL(k_sum_scalars, dim3((n + 63) / 64), dim3(64), 0, s, dst, srcs, nsrc, n);
}

struct SumSrcs { const float* p[kMaxSumRanks]; };
__global__ void __launch_bounds__(256) k_sum_f32_ranks(float* __restrict__ dst, SumSrcs srcs, int nsrc, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float v = srcs.p[0][i];
    for (int s = 1; s < nsrc; ++s) v += srcs.p[s][i];
    dst[i] = v;
  }
}

void launch_sum_f32_ranks(float* dst, const float* const* srcs_host, int nsrc, size_t n, hipStream_t s) {
  SumSrcs a = {};
  for (int r = 0; r < nsrc && r < kMaxSumRanks; ++r) a.p[r] = srcs_host[r];
  const size_t blocks = std::min<size_t>((n + 255) / 256, 4096);
  hipLaunchKernelGGL(k_sum_f32_ranks, dim3((unsigned)std::max<size_t>(blocks, 1)), dim3(256), 0, s, dst, a, nsrc, n);
}

// --------------------------------------------------------------------------------------

}  // namespace oi
