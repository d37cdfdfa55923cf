// oi_effdiff.hip -- homogenization path (the reference's default calculation_method):
// periodic cell problems chi_k and the D_eff tensor, on the same HBM layout and CG
// building blocks as the tortuosity path: unfused iteration (periodic apply +
// the flat vector kernels of oi_k_dense.hip); z-slabs over a RING of ranks (rank 0 and the last
// rank are neighbours: periodic z), one p-plane ring exchange and two scalar all-reduces per iteration.
//
// Replaces OpenImpala::EffectiveDiffusivityHypre (src/props/EffectiveDiffusivityHypre.cpp:
// ctor/generateActiveMask :213-323, solve :543-687, getChiSolution :689-) with its Fortran
// assembly kernel effdiff_fillmtx (src/props/EffDiffFillMtx.F90:42-264) and the tensor
// integrator calculate_Deff_tensor_homogenization (src/props/Diffusion.cpp:60-167).
//
// System restated from EffDiffFillMtx.F90 (dx = 1, fully periodic): active = phase == id;
// row of an active cell: -1 towards every active neighbour, diagonal 6 regardless of the
// neighbours (both branches of F90:126-205 add 1/dx^2), rhs_k = (D(+k) - D(-k)) / 2;
// inactive cells: identity, rhs 0.  With vectors that vanish on inactive cells,
// (A v)[m] = 6 v[m] - (sum of the six periodic neighbours): the tortuosity kernel form with
// code = 6.  SPD (strictly diagonally dominant at every interface), so plain CG applies.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <memory>
#include <vector>

#include "oi_comm.h"
#include "oi_device.h"
#include "oi_pool.h"
#include "oi_mg.h"

using namespace oi;

#define HIPC(call)                                                                       \
  do {                                                                                   \
    hipError_t e_ = (call);                                                              \
    if (e_ != hipSuccess) {                                                              \
      oi::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
      (void)hipGetLastError(); /* reported: do not leave it for the next launch check to find */ \
      return OI_HIP_ERROR;                                                               \
    }                                                                                    \
  } while (0)
#define OIC(call)                \
  do {                           \
    oi_status s_ = (call);       \
    if (s_ != OI_OK) return s_;  \
  } while (0)

namespace {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}
__device__ __forceinline__ double block_sum(double v, double* lds) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  v = wave_sum(v);
  __syncthreads();
  if (lane == 0) lds[wave] = v;
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x == 0)
    for (int w = 0; w < nw; ++w) t += lds[w];
  return t;
}

// element index of (plane g incl. ghosts, row j, column i) with periodic wrap in x and y
__device__ __forceinline__ long long eidx(const Geom& g, int gz, int j, int i) {
  i = i < 0 ? i + g.nx : (i >= g.nx ? i - g.nx : i);
  j = j < 0 ? j + g.ny : (j >= g.ny ? j - g.ny : j);
  return (long long)gz * g.plane + (long long)j * g.pitch + i;
}

// phase bytes -> code 6 (active) / 0 on the owned planes; count of active cells
__global__ void __launch_bounds__(256) k_ed_codes(Geom g, const uint8_t* __restrict__ phase, int phase_id,
                                                   uint8_t* __restrict__ code, unsigned long long* n_active) {
  const long long n = (long long)g.nzl * g.ny * g.nx;
  unsigned long long cnt = 0;
  for (long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (long long)gridDim.x * blockDim.x) {
    const int i = (int)(v % g.nx), j = (int)((v / g.nx) % g.ny), gz = 1 + (int)(v / ((long long)g.nx * g.ny));
    const bool act = phase[v] == (uint8_t)phase_id;
    code[(long long)gz * g.plane + (long long)j * g.pitch + i] = act ? 6 : 0;
    cnt += act;
  }
  __shared__ double red[4];
  const double t = block_sum((double)cnt, red);
  if (threadIdx.x == 0 && t != 0.0) atomicAdd(n_active, (unsigned long long)t);
}

// r = b_k = (D(+k) - D(-k)) / 2 on active cells (EffDiffFillMtx.F90:208-217 + :131-201), p = r,
// chi = 0; partial r.r.  Needs the periodic ghost planes of `code`.
__global__ void __launch_bounds__(256) k_ed_rhs(Geom g, const uint8_t* __restrict__ code, int dir_k,
                                                 double* __restrict__ chi, double* __restrict__ r,
                                                 double* __restrict__ p, double* __restrict__ partials) {
  const long long n = (long long)g.nzl * g.ny * g.nx;
  double rr = 0.0;
  for (long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (long long)gridDim.x * blockDim.x) {
    const int i = (int)(v % g.nx), j = (int)((v / g.nx) % g.ny), gz = 1 + (int)(v / ((long long)g.nx * g.ny));
    const long long e = (long long)gz * g.plane + (long long)j * g.pitch + i;
    double b = 0.0;
    if (code[e]) {
      const int di = dir_k == 0, dj = dir_k == 1, dk = dir_k == 2;
      const double dm = code[eidx(g, gz - dk, j - dj, i - di)] ? 1.0 : 0.0;
      const double dp = code[eidx(g, gz + dk, j + dj, i + di)] ? 1.0 : 0.0;
      b = (dp - dm) * 0.5;
    }
    chi[e] = 0.0;
    r[e] = b;
    p[e] = b;
    rr += b * b;
  }
  __shared__ double red[4];
  const double t = block_sum(rr, red);
  if (threadIdx.x == 0) { partials[2 * blockIdx.x] = t; partials[2 * blockIdx.x + 1] = t; }  // {r.r, r.z}
}

// the same right-hand side into the COMPACTED line layout (launch_build_lines): r only -- the rotated iteration starts from
// p = 0 and chi = 0 (memsets) -- and the partial sums {r.r, r.r}.  A line without an active cell is not stored and has b = 0.
__global__ void __launch_bounds__(256) k_ed_rhs_lines(Geom g, const uint8_t* __restrict__ code, int dir_k, double* __restrict__ r,
                                                       const int* __restrict__ lineoff, int LW, double* __restrict__ partials) {
  // a block takes whole rows (two scalar divisions per row instead of three 64-bit ones per voxel), its threads the row's voxels
  const long long nrows = (long long)g.nzl * g.ny;
  const int di = dir_k == 0, dj = dir_k == 1, dk = dir_k == 2;
  double rr = 0.0;
  for (long long row = blockIdx.x; row < nrows; row += gridDim.x) {
    const int gz = 1 + (int)(row / g.ny), j = (int)(row - (long long)(gz - 1) * g.ny);
    const long long e0 = (long long)gz * g.plane + (long long)j * g.pitch;
    const int* lrow = lineoff + ((long long)gz * g.ny + j) * LW;
    for (int i = threadIdx.x; i < g.nx; i += blockDim.x) {
      const int lo = lrow[i >> kLVs];
      if (lo < 0) continue;
      double b = 0.0;
      if (code[e0 + i]) {
        const double dm = code[eidx(g, gz - dk, j - dj, i - di)] ? 1.0 : 0.0;
        const double dp = code[eidx(g, gz + dk, j + dj, i + di)] ? 1.0 : 0.0;
        b = (dp - dm) * 0.5;
      }
      r[(long long)kLV * lo + (i & (kLV - 1))] = b;
      rr += b * b;
    }
  }
  __shared__ double red[4];
  const double t = block_sum(rr, red);
  if (threadIdx.x == 0) { partials[2 * blockIdx.x] = t; partials[2 * blockIdx.x + 1] = t; }
}

// q = A v (periodic), partial v.q; same summation order as the oracle
__global__ void __launch_bounds__(256) k_ed_apply(Geom g, const uint8_t* __restrict__ code,
                                                   const double* __restrict__ v, double* __restrict__ q,
                                                   double* __restrict__ partials, const Ctrl* __restrict__ ctrl) {
  if (ctrl->done) return;
  const long long n = (long long)g.nzl * g.ny * g.nx;
  double pq = 0.0;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x) {
    const int i = (int)(t % g.nx), j = (int)((t / g.nx) % g.ny), gz = 1 + (int)(t / ((long long)g.nx * g.ny));
    const long long e = (long long)gz * g.plane + (long long)j * g.pitch + i;
    // q vanishes on inactive cells and its buffer holds zeros there: groups of 8 lanes (64 B of q) without an
    // active cell are skipped, the others are written whole
    const bool act = code[e] != 0;
    const unsigned long long mlive = __ballot(act);
    if (((mlive >> (threadIdx.x & 56)) & 0xffull) == 0) continue;
    double out = 0.0;
    if (act) {
      const double c = v[e];
      const double s = ((((v[eidx(g, gz, j, i - 1)] + v[eidx(g, gz, j, i + 1)]) + v[eidx(g, gz, j - 1, i)]) +
                         v[eidx(g, gz, j + 1, i)]) + v[e - g.plane]) + v[e + g.plane];
      out = 6.0 * c - s;
      pq += c * out;
    }
    q[e] = out;
  }
  __shared__ double red[4];
  const double t = block_sum(pq, red);
  if (threadIdx.x == 0) partials[blockIdx.x] = t;
}

// D[row][col] partial sums: over active cells, (row == col) - d chi_col / d x_row  (Diffusion.cpp:108-140)
__global__ void __launch_bounds__(256) k_ed_tensor(Geom g, const uint8_t* __restrict__ code,
                                                    const double* __restrict__ cx, const double* __restrict__ cy,
                                                    const double* __restrict__ cz, double* __restrict__ partials) {
  const long long n = (long long)g.nzl * g.ny * g.nx;
  double s[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  const double* chi[3] = {cx, cy, cz};
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x) {
    const int i = (int)(t % g.nx), j = (int)((t / g.nx) % g.ny), gz = 1 + (int)(t / ((long long)g.nx * g.ny));
    const long long e = (long long)gz * g.plane + (long long)j * g.pitch + i;
    if (!code[e]) continue;
    const long long nb[6] = {eidx(g, gz, j, i - 1), eidx(g, gz, j, i + 1), eidx(g, gz, j - 1, i), eidx(g, gz, j + 1, i),
                             e - g.plane, e + g.plane};
#pragma unroll
    for (int row = 0; row < 3; ++row)
#pragma unroll
      for (int col = 0; col < 3; ++col) {
        const double grad = (chi[col][nb[2 * row + 1]] - chi[col][nb[2 * row]]) * 0.5;
        s[3 * row + col] += (row == col ? 1.0 : 0.0) - grad;
      }
  }
  __shared__ double red[4];
  for (int c = 0; c < 9; ++c) {
    const double t = block_sum(s[c], red);
    if (threadIdx.x == 0) partials[9 * blockIdx.x + c] = t;
  }
}

int nblocks(long long items, int cap) {
  long long b = (items + 255) / 256;
  return (int)std::max<long long>(1, std::min<long long>(b, cap));
}

}  // namespace

struct oi_effdiff {
  Geom g{};
  oi_tort_opts opts{};
  int device = 0, phase_id = 0;
  oi_comm* comm = nullptr;  // ring of z-slabs (null: the whole volume lives here)
  hipStream_t stream = nullptr;
  uint8_t* code = nullptr;
  double *chi[3] = {nullptr, nullptr, nullptr}, *r = nullptr, *p = nullptr, *q = nullptr;
  double* z = nullptr;              // preconditioned residual (OI_PRECOND_MG)
  double* p2 = nullptr;             // second search-direction buffer of the fused two-kernel iteration
  bool fused = false;               // multigrid PCG on the tiled kernels of the flow-through path (even extents)
  // round 4: the multigrid-preconditioned solves (one GPU, or a ring whose slabs all start on an even plane) take the flow-through path's COMPACTED vectors (r, q, p, z: only the 8-voxel lines
  // that hold an active cell) and its rotated multigrid iteration on k_cg1b's forms (oi_tort.cpp: cg_one_iteration, h->rot)
  bool rot = false;
  int* lineoff = nullptr;
  long long* plane_first_dev = nullptr;
  void* seg = nullptr;
  std::vector<long long> plane_first;
  long long nvec_cg = 0;            // elements of r / q / p / z (= g.nvec on dense vectors)
  double* r2 = nullptr;             // second residual buffer of the rotated iteration
  oi::MgHierarchy* mg = nullptr;    // periodic aggregation-multigrid hierarchy of the cell problems
  double *partials = nullptr, *hist = nullptr, *scal = nullptr;
  Ctrl* ctrl = nullptr;
  unsigned long long* counter = nullptr;
  int maxiter = 0;
  long long n_active = 0;
  bool solved[3] = {false, false, false};
  int iters[3] = {0, 0, 0};
  double relres[3] = {0, 0, 0};
};

extern "C" void oi_effdiff_destroy(oi_effdiff* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  if (h->mg) oi::mg_destroy(h->mg);
  void* dev[] = {h->code, h->chi[0], h->chi[1], h->chi[2], h->r, h->r2, h->p, h->p2, h->q, h->z, h->partials, h->hist, h->scal, h->ctrl, h->counter,
                 h->lineoff, h->plane_first_dev, h->seg};
  for (void* d : dev)
    if (d) (void)oi::dev_free(d);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

// periodic images of the first / last owned plane into the ghost planes (FillBoundary with a
// periodic geometry, Diffusion.cpp:306-309)
static oi_status wrap_z(oi_effdiff* h, void* base, size_t plane_bytes) {
  char* b = (char*)base;
  const size_t nzl = (size_t)h->g.nzl;
  if (h->comm)  // my first plane -> previous rank's upper ghost, my last plane -> next rank's lower ghost
    return h->comm->exchange_ring(b + plane_bytes, b, b + plane_bytes * nzl, b + plane_bytes * (nzl + 1), plane_bytes,
                                  h->stream);
  HIPC(hipMemcpyAsync(b, b + plane_bytes * nzl, plane_bytes, hipMemcpyDeviceToDevice, h->stream));
  HIPC(hipMemcpyAsync(b + plane_bytes * (nzl + 1), b + plane_bytes, plane_bytes, hipMemcpyDeviceToDevice, h->stream));
  return OI_OK;
}

// the same images for a COMPACTED vector: the stored lines of plane nzl are those of ghost plane 0 (same code bytes, same line
// list), those of plane 1 are ghost plane nzl + 1's
static oi_status wrap_lines(oi_effdiff* h, double* v) {
  const std::vector<long long>& pf = h->plane_first;
  const size_t nzl = (size_t)h->g.nzl, le = (size_t)kLV * sizeof(double);
  if (h->comm)  // a ring of slabs: my first plane's lines -> the previous rank's upper ghost, my last plane's -> the next rank's lower ghost
    return h->comm->exchange_ring_v(v + kLV * pf[1], le * (size_t)(pf[2] - pf[1]), v + kLV * pf[0], le * (size_t)(pf[1] - pf[0]),
                                    v + kLV * pf[nzl], le * (size_t)(pf[nzl + 1] - pf[nzl]), v + kLV * pf[nzl + 1],
                                    le * (size_t)(pf[nzl + 2] - pf[nzl + 1]), h->stream);
  HIPC(hipMemcpyAsync(v + kLV * pf[0], v + kLV * pf[nzl], le * (size_t)(pf[nzl + 1] - pf[nzl]), hipMemcpyDeviceToDevice, h->stream));
  HIPC(hipMemcpyAsync(v + kLV * pf[nzl + 1], v + kLV * pf[1], le * (size_t)(pf[2] - pf[1]), hipMemcpyDeviceToDevice, h->stream));
  return OI_OK;
}

// Ring neighbours prove to each other, once, that they agree on the stored-line counts of the planes they exchange (every rank
// derives them from its OWN copy of the code bytes; exchange_ring_v skips zero-length messages on each side independently, so a
// disagreement would dead-lock the send / receive group -- oi_tort.cpp: check_plane_message_lengths).  *ok on every rank alike.
static oi_status ring_lengths_agree(oi_effdiff* h, unsigned long long* d /* 8 device words */, int* ok) {
  const std::vector<long long>& pf = h->plane_first;
  const int nzl = h->g.nzl;
  hipStream_t s = h->stream;
  const unsigned long long host[8] = {(unsigned long long)(pf[2] - pf[1]), (unsigned long long)(pf[1] - pf[0]), 0, 0,
                                      (unsigned long long)(pf[nzl + 1] - pf[nzl]), (unsigned long long)(pf[nzl + 2] - pf[nzl + 1]), 0, 0};
  HIPC(hipMemcpyAsync(d, host, sizeof host, hipMemcpyHostToDevice, s));
  // {what I send down, what I expect from below} -> the previous rank's words 6, 7; {what I send up, what I expect from above} ->
  // the next rank's words 2, 3
  OIC(h->comm->exchange_ring(d + 0, d + 2, d + 4, d + 6, 2 * sizeof(unsigned long long), s));
  unsigned long long back[8];
  HIPC(hipMemcpyAsync(back, d, sizeof back, hipMemcpyDeviceToHost, s));
  HIPC(hipStreamSynchronize(s));
  int bad = (back[2] != host[1] || back[3] != host[0] || back[6] != host[5] || back[7] != host[4]) ? 1 : 0;
  OIC(h->comm->allreduce_max_i32_host(&bad, 1, s));
  *ok = bad ? 0 : 1;
  return OI_OK;
}

// sum of per-block partials into dst (deterministic), summed over the ranks, optional convergence test
static oi_status reduce_to(oi_effdiff* h, int nblocks, int ncomp, double* dst, bool check, int iter_done) {
  launch_finalize(h->partials, nblocks, ncomp, dst, h->ctrl, (check && !h->comm) ? 1 : 0, iter_done, h->stream);
  if (h->comm) {
    OIC(h->comm->allreduce_sum_f64(dst, ncomp, h->stream));
    if (check) launch_check_done(dst, h->ctrl, iter_done, h->stream);
  }
  return OI_OK;
}

static oi_status effdiff_create_common(oi_effdiff** out, oi_comm* comm, const uint8_t* phase, bool on_device, int nx,
                                       int ny, int nz, int z0, int nzl, int phase_id, const oi_tort_opts* opts) {
  if (!out) { oi::set_error("%s: null handle or argument out of range", __func__); return OI_BAD_ARG; }
  *out = nullptr;
  if (!phase || nx < 1 || ny < 1 || nz < 1 || nzl < 1 || z0 < 0 || z0 + nzl > nz) { oi::set_error("%s: null handle or argument out of range", __func__); return OI_BAD_ARG; }
  oi_tort_opts o;
  oi_tort_default_opts(&o);
  if (opts) o = *opts;
  if (!(o.eps > 0.0)) { oi::set_error("eps must be positive"); return OI_BAD_ARG; }  // EffectiveDiffusivityHypre.cpp:156
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
    oi::set_error("no HIP device visible: this library has no CPU fallback");
    return OI_NO_DEVICE;
  }
  if (comm) HIPC(hipSetDevice(comm->device));
  else if (o.device >= 0) HIPC(hipSetDevice(o.device));
  std::unique_ptr<oi_effdiff, void (*)(oi_effdiff*)> h(new oi_effdiff(), oi_effdiff_destroy);
  HIPC(hipGetDevice(&h->device));
  h->opts = o;
  h->phase_id = phase_id;
  h->comm = (comm && comm->size > 1) ? comm : nullptr;
  Geom& g = h->g;
  g.nx = nx; g.ny = ny; g.nz = nz; g.z0 = z0; g.nzl = nzl; g.dir = 0;
  g.pitch = (nx + 1) & ~1;
  g.W = (nx + 63) / 64;
  g.plane = (long long)g.pitch * ny;
  g.wplane = (long long)g.W * ny;
  g.nvec = (long long)(nzl + 2) * g.plane;
  g.nwords = 0;
  g.vlo = 0; g.vhi = 0;
  g.periodic = 1;  // the tiled kernels wrap x and y by index
  g.tune = oi::default_tune();  // this handle's launch shapes: a snapshot of the process defaults
  g.tune.max_blocks = oi::partial_blocks_bound(g.pitch, ny, nzl);
  // the reference's default budget is 1000 multigrid-preconditioned iterations (:120); CG on this
  // well-conditioned system needs ~100, so the same default is generous
  h->maxiter = o.maxiter > 0 ? o.maxiter : 1000 + 4 * (nx + ny + nz);
  // A ring of slabs builds in lock-step: every stretch of rank-local work (allocations, set-up kernels) ends in oi::agree_status,
  // so a slab that cannot get its memory takes its neighbours out with an error instead of leaving them in the next collective.
  const hipError_t es = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
  hipStream_t s = h->stream;
  const size_t n = (size_t)nx * ny * nzl;
  auto local_codes = [&]() -> oi_status {
    HIPC(es);
    HIPC(oi::dev_malloc((void**)&h->code, (size_t)g.nvec));
    for (int k = 0; k < 3; ++k) HIPC(oi::dev_malloc((void**)&h->chi[k], sizeof(double) * g.nvec));
    HIPC(oi::dev_malloc((void**)&h->partials, sizeof(double) * 9 * g.tune.max_blocks));
    HIPC(oi::dev_malloc((void**)&h->hist, sizeof(double) * H_STRIDE * (h->maxiter + 2)));
    HIPC(oi::dev_malloc((void**)&h->scal, sizeof(double) * 16));
    HIPC(oi::dev_malloc((void**)&h->ctrl, sizeof(Ctrl)));
    HIPC(oi::dev_malloc((void**)&h->counter, sizeof(unsigned long long)));
    HIPC(hipMemsetAsync(h->code, 0, (size_t)g.nvec, s));
    HIPC(hipMemsetAsync(h->counter, 0, sizeof(unsigned long long), s));
    for (int k = 0; k < 3; ++k) HIPC(hipMemsetAsync(h->chi[k], 0, sizeof(double) * g.nvec, s));
    uint8_t* tmp = nullptr;
    const uint8_t* dphase = phase;
    if (!on_device) {
      HIPC(oi::dev_malloc((void**)&tmp, n));
      const hipError_t eu = hipMemcpyAsync(tmp, phase, n, hipMemcpyHostToDevice, s);
      if (eu != hipSuccess) { (void)oi::dev_free(tmp); HIPC(eu); }
      dphase = tmp;
    }
    hipLaunchKernelGGL(k_ed_codes, dim3(nblocks((long long)n, 8192)), dim3(256), 0, s, g, dphase, phase_id, h->code, h->counter);
    const hipError_t ek = hipStreamSynchronize(s);
    if (tmp) (void)oi::dev_free(tmp);
    HIPC(ek);
    return OI_OK;
  };
  OIC(oi::agree_status(h->comm, local_codes(), s, "the allocation of the code bytes and the solution fields"));
  OIC(wrap_z(h.get(), h->code, (size_t)g.plane));
  unsigned long long cnt = 0;
  HIPC(hipMemcpyAsync(&cnt, h->counter, sizeof(cnt), hipMemcpyDeviceToHost, s));
  HIPC(hipStreamSynchronize(s));
  if (h->comm) OIC(h->comm->allreduce_sum_u64_host(&cnt, 1, s));
  h->n_active = (long long)cnt;
  // one multigrid cycle per PCG iteration (what PFMG is to the reference's FlexGMRES, EffectiveDiffusivityHypre.cpp:
  // 578-600); tiny or empty volumes keep plain CG
  const bool want_mg = (o.precond == OI_PRECOND_MG || o.precond == OI_PRECOND_DEFAULT) && h->n_active > 0 && nx >= 4 && ny >= 4 && nz >= 4;
  h->nvec_cg = g.nvec;
  // Compacted vectors + the rotated iteration of the flow-through path (k_cg1b FORM 2 -> coarse levels, FORM 3 -> FORM 1, their
  // periodic instantiations): one GPU, any extents
  // (a ring of slabs: every rank takes the same decision -- the iteration has collectives inside -- and the neighbours agree on
  // the line lists they will exchange)
  // (over RCCL the compacted ring exchange stays off until it has run on more than one GPU: Tune::ring_lines_rccl)
  const bool ring_lines = !h->comm || h->comm->kind != 1 || g.tune.ring_lines_rccl;
  // (k_fused_a on compacted vectors is a cross-check kernel of the test library: the product compacts where k_cg1b FORM 1 runs)
  const bool descr_first = (g.tune.fused_kernel == 0 && g.tune.fused_rows == 2 && 8ll * g.plane < 0x7fff0000ll) || oi::xcheck_built();
  int can = (want_mg && ring_lines && descr_first && !g.tune.mg_plain && g.tune.cg1_compact && !g.tune.mg_norot && periodic_tiles_ok(g) &&
             (long long)(nzl + 2) * ny * ((g.pitch + kLV - 1) / kLV) < (1ll << 31)) ? 1 : 0;
  if (h->comm) {
    int cannot = can ? 0 : 1;
    OIC(h->comm->allreduce_max_i32_host(&cannot, 1, s));
    can = cannot ? 0 : 1;
  }
  if (can) {
    const int LW = (g.pitch + kLV - 1) / kLV;
    const long long nlines = (long long)(nzl + 2) * ny * LW;
    unsigned long long* d8 = nullptr;  // scratch of the ring's length proof
    bool images = false;
    int rot = 0;
    auto local_lines = [&]() -> oi_status {
      HIPC(oi::dev_malloc((void**)&h->lineoff, sizeof(int) * (size_t)nlines));
      HIPC(oi::dev_malloc((void**)&h->plane_first_dev, sizeof(long long) * ((size_t)nzl + 3)));
      HIPC(oi::dev_malloc(&h->seg, sizeof(uint2) * (size_t)(nzl + 2) * ny * ((g.pitch + 127) / 128)));
      if (h->comm) HIPC(oi::dev_malloc((void**)&d8, 8 * sizeof(unsigned long long)));
      HIPC(launch_build_lines(g, h->code, h->lineoff, h->plane_first_dev, h->seg, s));
      h->plane_first.resize((size_t)nzl + 3);
      HIPC(hipMemcpy(h->plane_first.data(), h->plane_first_dev, sizeof(long long) * (nzl + 3), hipMemcpyDeviceToHost));
      const std::vector<long long>& pf = h->plane_first;
      // (the ghost planes carry the code bytes of their images: the same line lists, or the layout is not what wrap_lines assumes)
      images = pf[1] - pf[0] == pf[nzl + 1] - pf[nzl] && pf[nzl + 2] - pf[nzl + 1] == pf[2] - pf[1];
      g.live_lines = (float)((double)pf[nzl + 2] / (double)nlines);
      rot = fused_descr_ok(g, h->lineoff, h->plane_first_dev, h->seg, 0) && mg_post_descr(g, h->lineoff, h->plane_first_dev, h->seg) ? 1 : 0;
      return OI_OK;
    };
    oi_status st = oi::agree_status(h->comm, local_lines(), s, "the set-up of the compacted line lists");
    if (st == OI_OK && h->comm) {
      int agree = 0;
      st = ring_lengths_agree(h.get(), d8, &agree);
      images = agree != 0;
      if (st == OI_OK) {
        int no = (rot && images) ? 0 : 1;  // (the live-line share drives launch shapes: one value on every rank)
        st = h->comm->allreduce_max_i32_host(&no, 1, s);
        rot = no ? 0 : 1;
      }
    }
    if (d8) (void)oi::dev_free(d8);
    OIC(st);
    h->rot = images && rot;
    if (h->rot) {
      h->nvec_cg = std::max<long long>(16, (long long)kLV * h->plane_first[nzl + 2]);
    } else {
      (void)oi::dev_free(h->lineoff); (void)oi::dev_free(h->plane_first_dev); (void)oi::dev_free(h->seg);
      h->lineoff = nullptr; h->plane_first_dev = nullptr; h->seg = nullptr;
      g.live_lines = 0.f;
    }
  }
  // whether the fused two-kernel iteration runs (and needs a second p buffer) is known before the vectors are allocated:
  // the LDS-tiled kernels of the flow-through path wrap x / y by index -> their fused two-kernel iteration.  Odd extents included
  // (round 3): the coarse parent of the image planes beyond an odd nz is the coarse ghost plane (cplane_of in k_mg_fine), and with an
  // odd nx the pair (nx - 1, pad) keeps the wrapped column 0 in its second LDS slot.  Without multigrid: plain CG (precond = none /
  // jacobi is not offered here), the same two-kernel iteration with z = r
  h->fused = want_mg ? !g.tune.mg_plain : (h->n_active > 0 && nx >= 4 && ny >= 2 && !g.tune.mg_plain);
  auto local_vectors = [&]() -> oi_status {
    HIPC(oi::dev_malloc((void**)&h->r, sizeof(double) * h->nvec_cg));
    HIPC(oi::dev_malloc((void**)&h->p, sizeof(double) * h->nvec_cg));
    HIPC(oi::dev_malloc((void**)&h->q, sizeof(double) * h->nvec_cg));
    if (h->rot) HIPC(oi::dev_malloc((void**)&h->r2, sizeof(double) * h->nvec_cg));
    if (want_mg) {
      HIPC(oi::dev_malloc((void**)&h->z, sizeof(double) * h->nvec_cg));
      HIPC(hipMemsetAsync(h->z, 0, sizeof(double) * h->nvec_cg, s));
    }
    if (h->fused) {
      HIPC(oi::dev_malloc((void**)&h->p2, sizeof(double) * h->nvec_cg));
      HIPC(hipMemsetAsync(h->p2, 0, sizeof(double) * h->nvec_cg, s));
    }
    return OI_OK;
  };
  OIC(oi::agree_status(h->comm, local_vectors(), s, "the allocation of the CG vectors"));
  if (want_mg) {
    size_t mg_bytes = 0;
    h->mg = oi::mg_setup(g, h->code, h->comm, 1, s, &mg_bytes);  // (agrees on its own allocations)
    if (!h->mg) return OI_HIP_ERROR;
  }
  *out = h.release();
  return OI_OK;
}

extern "C" oi_status oi_effdiff_create(oi_effdiff** out, const uint8_t* phase, int nx, int ny, int nz, int phase_id,
                                       const oi_tort_opts* opts) {
  return effdiff_create_common(out, nullptr, phase, false, nx, ny, nz, 0, nz, phase_id, opts);
}

// one z-slab [z0, z0 + nz_local) of a volume spread over the ranks of `comm` (collective)
extern "C" oi_status oi_effdiff_create_slab(oi_effdiff** out, oi_comm* comm, const uint8_t* phase, int on_device, int nx,
                                            int ny, int nz_global, int z0, int nz_local, int phase_id,
                                            const oi_tort_opts* opts) {
  return effdiff_create_common(out, comm, phase, on_device != 0, nx, ny, nz_global, z0, nz_local, phase_id, opts);
}

// EffectiveDiffusivityHypre::solve (EffectiveDiffusivityHypre.cpp:543-687) for direction k.
// *converged = 0 leaves chi_k = 0 as the reference does (:641-645).
extern "C" oi_status oi_effdiff_solve(oi_effdiff* h, int dir_k, int* converged, int* iterations, double* relres) {
  if (!h || dir_k < 0 || dir_k > 2) { oi::set_error("%s: null handle or argument out of range", __func__); return OI_BAD_ARG; }
  HIPC(hipSetDevice(h->device));
  const Geom& g = h->g;
  hipStream_t s = h->stream;
  double* chi = h->chi[dir_k];
  if (converged) *converged = 0;
  if (h->n_active == 0) {  // :558-575: nothing to solve, chi = 0, reported as converged
    HIPC(hipMemsetAsync(chi, 0, sizeof(double) * g.nvec, s));
    HIPC(hipStreamSynchronize(s));
    h->solved[dir_k] = true; h->iters[dir_k] = 0; h->relres[dir_k] = 0.0;
    if (converged) *converged = 1;
    if (iterations) *iterations = 0;
    if (relres) *relres = 0.0;
    return OI_OK;
  }
  const long long n = (long long)g.nzl * g.ny * g.nx;
  const int nb0 = nblocks(n, 2048);
  // (rank-local start of a solve; its outcome is agreed on before the first collective, so that a slab whose device has failed
  // does not leave its neighbours waiting in the all-reduce of ||b||)
  auto local_start = [&]() -> oi_status {
    HIPC(hipMemsetAsync(h->hist, 0, sizeof(double) * H_STRIDE * (h->maxiter + 2), s));
    // (rotated iteration: r2 and p are written whole by launch 0 before anything reads them -- checked with both filled with NaNs
    // before 450 solves of 150 shapes, tools/homog_oddnx_probe.py; r is cleared because k_ed_rhs_lines leaves the pad voxel of an odd
    // nx alone -- garbage there enters ||r||^2 --, and what launch 0 READS beside r -- q_{-1} and p_{-1}, multiplied by alpha = beta = 0
    // -- must be finite)
    HIPC(hipMemsetAsync(h->r, 0, sizeof(double) * h->nvec_cg, s));
    if (!h->rot) HIPC(hipMemsetAsync(h->p, 0, sizeof(double) * h->nvec_cg, s));
    if (h->p2) HIPC(hipMemsetAsync(h->p2, 0, sizeof(double) * h->nvec_cg, s));
    HIPC(hipMemsetAsync(h->q, 0, sizeof(double) * h->nvec_cg, s));
    HIPC(hipMemsetAsync(chi, 0, sizeof(double) * g.nvec, s));
    if (h->rot) hipLaunchKernelGGL(k_ed_rhs_lines, dim3(nb0), dim3(256), 0, s, g, h->code, dir_k, h->r, h->lineoff, (g.pitch + kLV - 1) / kLV, h->partials);
    else hipLaunchKernelGGL(k_ed_rhs, dim3(nb0), dim3(256), 0, s, g, h->code, dir_k, chi, h->r, h->p, h->partials);
    HIPC(hipGetLastError());
    return OI_OK;
  };
  OIC(oi::agree_status(h->comm, local_start(), s, "the start of a cell-problem solve"));
  Ctrl c;
  c.done = 0; c.pad = 0; c.tol2 = std::numeric_limits<double>::infinity();  // set once ||b|| is known
  HIPC(hipMemcpyAsync(h->ctrl, &c, sizeof(Ctrl), hipMemcpyHostToDevice, s));
  OIC(reduce_to(h, nb0, 2, h->hist + H_RR, false, 0));
  double bb = 0.0;
  HIPC(hipMemcpyAsync(&bb, h->hist + H_RR, sizeof(double), hipMemcpyDeviceToHost, s));
  HIPC(hipStreamSynchronize(s));
  const double bnorm = std::sqrt(bb);
  int K = 0;
  double rel = 0.0;
  bool conv = bnorm == 0.0;  // b = 0 (e.g. no interface along k): chi = 0 is exact
  if (!conv) {
    c.tol2 = (h->opts.eps * bnorm) * (h->opts.eps * bnorm);
    HIPC(hipMemcpyAsync(h->ctrl, &c, sizeof(Ctrl), hipMemcpyHostToDevice, s));
    const int batch = h->opts.check_every > 0 ? h->opts.check_every : (h->mg ? 4 : 16);
    int issued = 0;
    Ctrl seen = c;
    const int limit = h->maxiter + (h->rot ? 1 : 0);  // launch k of the rotated iteration completes iteration k - 1
    if (h->rot) OIC(wrap_lines(h, h->r));
    while (!seen.done && issued < limit) {
      for (int b = 0; b < batch && issued < limit; ++b, ++issued) {
        const int k = issued;
        if (h->rot) {
          // oi_tort.cpp, cg_one_iteration (h->rot), with the plane exchange of a slab replaced by the periodic images.
          // A: r_k = r_{k-1} - alpha_{k-1} q_{k-1}, ||r_k||^2 and the stop test, pre-smoother + restriction; B: the rest of the
          // cycle, z_k, r_k.z_k and (even launches) chi += alpha_{k-2} p_{k-2} + alpha_{k-1} p_{k-1}; C: p_k, q_k, p_k.q_k
          double* pbuf[2] = {h->p, h->p2};
          double* p_in = pbuf[(k + 1) & 1];
          double* p_out = pbuf[k & 1];
          const double* r_in = (k & 1) ? h->r2 : h->r;
          double* r_out = (k & 1) ? h->r : h->r2;
          MgCoarse cref{};
          float* bcoarse = nullptr;
          double omega = 0.0;
          if (!oi::mg_coarse_target(h->mg, &cref, &bcoarse, &omega, s)) return OI_HIP_ERROR;
          int nb = launch_mg_pre_update(g, h->code, r_in, r_out, h->q, h->hist, k, cref, bcoarse, omega, h->partials, h->ctrl, h->lineoff,
                                        h->plane_first_dev, h->seg, s);
          if (nb < 0) { oi::set_error("launch grid exceeds the handle's partial-sum capacity (%d blocks)", g.tune.max_blocks); return OI_BAD_ARG; }
          OIC(reduce_to(h, nb, 2, h->hist + H_STRIDE * k + H_RR, true, k + 1));
          const oi::MgXUpdate xu{chi, p_in, 0, h->hist, k, p_out};
          nb = oi::mg_vcycle_after_restriction(h->mg, g, h->code, r_out, h->z, h->partials, h->ctrl, h->lineoff, h->plane_first_dev, h->seg, s, &xu);
          if (nb < 0) return OI_RCCL_ERROR;
          OIC(reduce_to(h, nb, 1, h->hist + H_STRIDE * k + H_RZ, false, 0));
          OIC(wrap_lines(h, h->z));
          nb = launch_fused_a(g, h->code, chi, h->z, p_in, p_out, 0, h->q, 0, h->hist, k, h->partials, h->ctrl, h->lineoff, h->plane_first_dev,
                              h->seg, s, 1);
          if (nb < 0) { oi::set_error("launch grid exceeds the handle's partial-sum capacity (%d blocks)", g.tune.max_blocks); return OI_BAD_ARG; }
          OIC(reduce_to(h, nb, 1, h->hist + H_STRIDE * k + H_PQ, false, 0));
          OIC(wrap_lines(h, h->q));  // (kernel A of the next launch forms r on the image planes from r and q there)
          continue;
        }
        if (h->fused) {
          // the flow-through path's fused two-kernel PCG iteration (k_fused_a wraps x / y by index):
          // z = M^-1 r; p_k = z + beta p_{k-1}, q = A p_k, p.q, chi += alpha_{k-1} p_{k-1}; r -= alpha q
          double* pbuf[2] = {h->p, h->p2};
          OIC(wrap_z(h, h->r, sizeof(double) * g.plane));
          const double* zsrc = h->r;  // plain CG: z = r, and r.z = r.r is already in hist[k] (k_ed_rhs / launch_update_r)
          if (h->mg) {
            const int nbz = oi::mg_vcycle(h->mg, g, h->code, h->r, h->q, h->z, h->partials, h->ctrl, 1, nullptr, s);
            if (nbz < 0) return OI_RCCL_ERROR;
            OIC(reduce_to(h, nbz, 1, h->hist + H_STRIDE * k + H_RZ, false, 0));
            OIC(wrap_z(h, h->z, sizeof(double) * g.plane));
            zsrc = h->z;
          }
          int nb = launch_fused_a(g, h->code, chi, zsrc, pbuf[(k + 1) & 1], pbuf[k & 1], 0, h->q, 0, h->hist, k, h->partials,
                                  h->ctrl, nullptr, nullptr, nullptr, s);
          OIC(reduce_to(h, nb, 1, h->hist + H_STRIDE * k + H_PQ, false, 0));
          nb = launch_update_r(g, h->code, h->r, h->q, 0, h->hist, k, h->partials, h->ctrl, 0, -1, s);
          OIC(reduce_to(h, nb, 2, h->hist + H_STRIDE * (k + 1) + H_RR, true, k + 2));
          continue;
        }
        if (h->mg) {  // z = M^-1 r, r.z into hist[k], p = z + beta p
          OIC(wrap_z(h, h->r, sizeof(double) * g.plane));
          const int nbz = oi::mg_vcycle(h->mg, g, h->code, h->r, h->q, h->z, h->partials, h->ctrl, g.tune.mg_plain ? 0 : 1, nullptr, s);
          if (nbz < 0) return OI_RCCL_ERROR;
          OIC(reduce_to(h, nbz, 1, h->hist + H_STRIDE * k + H_RZ, false, 0));
          if (k > 0) launch_update_p(g, h->code, h->z, h->p, 0, h->hist, k, h->ctrl, s);
          else HIPC(hipMemcpyAsync(h->p, h->z, sizeof(double) * g.nvec, hipMemcpyDeviceToDevice, s));
        } else if (k > 0) launch_update_p(g, h->code, h->r, h->p, 0, h->hist, k, h->ctrl, s);
        OIC(wrap_z(h, h->p, sizeof(double) * g.plane));
        hipLaunchKernelGGL(k_ed_apply, dim3(nb0), dim3(256), 0, s, g, h->code, h->p, h->q, h->partials, h->ctrl);
        OIC(reduce_to(h, nb0, 1, h->hist + H_STRIDE * k + H_PQ, false, 0));
        const int nb = launch_update_xr(g, h->code, chi, h->r, h->p, h->q, 0, h->hist, k, h->partials, h->ctrl, s);
        OIC(reduce_to(h, nb, 2, h->hist + H_STRIDE * (k + 1) + H_RR, true, k + 2));
      }
      HIPC(hipMemcpyAsync(&seen, h->ctrl, sizeof(Ctrl), hipMemcpyDeviceToHost, s));
      HIPC(hipStreamSynchronize(s));
    }
    K = seen.done ? seen.done - 1 : (h->rot ? std::max(0, issued - 1) : issued);
    if (h->rot) {
      // pairwise update in the post-smoother (oi_tort_cg_finish): after launch m, chi = chi_e with e the largest even number <= m;
      // stopped at convergence (kernels B and C of launch K returned at once) m = K - 1, else m = K.  What chi still lacks, oldest
      // term first; p_j lives in buffer j & 1 and neither has been overwritten since
      double* pbuf[2] = {h->p, h->p2};
      const int m = seen.done ? K - 1 : K;
      const int e = m < 0 ? 0 : (m & ~1);
      if (K - e == 2) launch_axpy_hist(g, chi, pbuf[e & 1], 0, h->hist, e, h->lineoff, s, pbuf[(e + 1) & 1]);  // (two terms: one pass over chi)
      else for (int j = e; j < K; ++j) launch_axpy_hist(g, chi, pbuf[j & 1], 0, h->hist, j, h->lineoff, s);
    } else if (h->fused && K > 0) {  // the fused iteration applies chi += alpha_k p_k inside iteration k + 1: flush the last one
      double* pbuf[2] = {h->p, h->p2};
      launch_axpy_hist(g, chi, pbuf[(K - 1) & 1], 0, h->hist, K - 1, nullptr, s);
    }
    double rr = 0.0;
    HIPC(hipMemcpyAsync(&rr, h->hist + H_STRIDE * K + H_RR, sizeof(double), hipMemcpyDeviceToHost, s));
    HIPC(hipStreamSynchronize(s));
    rel = std::sqrt(rr) / bnorm;
    conv = seen.done != 0 && std::isfinite(rel) && rel <= h->opts.eps;  // :608-612
    if (!conv) HIPC(hipMemsetAsync(chi, 0, sizeof(double) * g.nvec, s));  // :641
  }
  OIC(wrap_z(h, chi, sizeof(double) * g.plane));  // periodic ghost cells of chi for the tensor
  HIPC(hipStreamSynchronize(s));
  h->solved[dir_k] = true; h->iters[dir_k] = K; h->relres[dir_k] = rel;
  if (converged) *converged = conv ? 1 : 0;
  if (iterations) *iterations = K;
  if (relres) *relres = rel;
  return OI_OK;
}

// getChiSolution (EffectiveDiffusivityHypre.cpp:689-): nx*ny*nz doubles, x fastest
extern "C" oi_status oi_effdiff_get_chi(oi_effdiff* h, int dir_k, double* chi_out) {
  if (!h || dir_k < 0 || dir_k > 2 || !chi_out) { oi::set_error("%s: null handle or argument out of range", __func__); return OI_BAD_ARG; }
  HIPC(hipSetDevice(h->device));
  const Geom& g = h->g;
  const size_t n = (size_t)g.nx * g.ny * g.nzl;
  double* dense = nullptr;
  HIPC(oi::dev_malloc((void**)&dense, n * sizeof(double)));
  launch_pitch_copy_out(g, h->chi[dir_k], dense, h->stream);
  hipError_t e = hipMemcpyAsync(chi_out, dense, n * sizeof(double), hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  (void)oi::dev_free(dense);
  HIPC(e);
  return OI_OK;
}

// calculate_Deff_tensor_homogenization (Diffusion.cpp:60-167): row-major 3x3, D_eff / D_material
extern "C" oi_status oi_effdiff_tensor(oi_effdiff* h, double* deff9) {
  if (!h || !deff9) { oi::set_error("%s: null handle or argument out of range", __func__); return OI_BAD_ARG; }
  if (!(h->solved[0] && h->solved[1] && h->solved[2])) {
    oi::set_error("solve chi_x, chi_y and chi_z first");
    return OI_BAD_ARG;
  }
  HIPC(hipSetDevice(h->device));
  const Geom& g = h->g;
  const long long n = (long long)g.nzl * g.ny * g.nx;
  const int nb = nblocks(n, 2048);
  hipLaunchKernelGGL(k_ed_tensor, dim3(nb), dim3(256), 0, h->stream, g, h->code, h->chi[0], h->chi[1], h->chi[2],
                     h->partials);
  launch_finalize(h->partials, nb, 9, h->scal, nullptr, 0, 0, h->stream);
  if (h->comm) OIC(h->comm->allreduce_sum_f64(h->scal, 9, h->stream));
  double sums[9];
  HIPC(hipMemcpyAsync(sums, h->scal, sizeof(sums), hipMemcpyDeviceToHost, h->stream));
  HIPC(hipStreamSynchronize(h->stream));
  const double ntot = (double)g.nx * g.ny * g.nz;
  for (int t = 0; t < 9; ++t) deff9[t] = sums[t] / ntot;  // :150-156
  return OI_OK;
}

extern "C" oi_status oi_effdiff_info(const oi_effdiff* h, int64_t* n_active, int* iterations3, double* relres3) {
  if (!h) { oi::set_error("%s: null handle or argument out of range", __func__); return OI_BAD_ARG; }
  if (n_active) *n_active = h->n_active;
  for (int k = 0; k < 3; ++k) {
    if (iterations3) iterations3[k] = h->iters[k];
    if (relres3) relres3[k] = h->relres[k];
  }
  return OI_OK;
}

// which vectors the solves run on: *compacted = 1 when r, q, p, z hold only the 8-voxel lines with an active cell and the solves
// take the rotated multigrid iteration (multigrid preconditioner; one GPU or a ring of slabs with even first planes); *vector_elements = doubles per such vector
extern "C" oi_status oi_effdiff_layout(const oi_effdiff* h, int* compacted, int64_t* vector_elements) {
  if (!h) { oi::set_error("%s: null handle or argument out of range", __func__); return OI_BAD_ARG; }
  if (compacted) *compacted = h->rot ? 1 : 0;
  if (vector_elements) *vector_elements = h->nvec_cg;
  return OI_OK;
}
