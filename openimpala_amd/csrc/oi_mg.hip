// oi_mg.hip -- aggregation-multigrid cycle used as the preconditioner of the CG solve
// (OI_PRECOND_MG).  What HYPRE's SMG buys the reference (TortuosityHypre.cpp:664-693: one
// multigrid V-cycle per Krylov iteration, <= 200 iterations), rebuilt matrix-free for the
// masked voxel operator:
//
//   * aggregates = 2x2x2 blocks of cells (aligned to GLOBAL even coordinates), piecewise-constant
//     prolongation P, restriction P^T;
//   * Galerkin coarse operators A_{l+1} = P^T A_l P are again 7-point operators whose entries
//     are INTEGER link counts: w_x(I) = number of fine links crossing the +x face of block I,
//     diag(I) = sum of the children's diagonals - 2 * (links inside the block); built level
//     by level on the device; the first three coarse levels store them as one byte per link and
//     two per diagonal (5 B/cell), deeper ones as floats (exact below 2^24);
//   * per level: one damped-Jacobi sweep from zero, residual, restriction, recursion, scaled
//     coarse correction, one damped-Jacobi sweep; coarse levels 1-3 are visited twice per
//     parent visit (W-cycle: unsmoothed aggregation needs it), the others once; the coarsest level
//     (<= 512 cells) gets 16 sweeps.  Every step is a fixed symmetric linear map, so the cycle
//     is an SPD preconditioner and plain PCG applies;
//   * coarse vectors are fp32 (the cycle only preconditions; the outer CG, its residual and
//     the fine-level smoothing stay fp64).
//
// Slabs: every level keeps the z-slab partition of the fine grid (ghost plane below and above,
// one neighbour exchange per produced right-hand side / correction) while the slab boundaries
// stay on even planes and the level is large; from the first level that is small (<= 2^18
// cells), thin or misaligned on some rank, the rest of the hierarchy is REPLICATED: its
// right-hand side is summed across ranks (one all-reduce per visit) and every rank runs the
// same small cycle.  The arithmetic per cell does not depend on the partition.
//
// Periodic variant (homogenization cell problems, oi_effdiff.hip): levels wrap in x and y by index and in z through
// the ghost planes (self-images, or the ring of slabs); the fine level then uses the plain kernels below.
//
// Measured on the oracle prototype (tests/experiments/mg_prototype.py, profiles/r01/mg_prototype.txt):
// 20-61 PCG iterations to 1e-6 against 474-1777 with Jacobi, independent of the grid size.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "oi_comm.h"
#include "oi_device.h"
#include "oi_pool.h"
#include "oi_mg.h"

namespace oi {

namespace {

constexpr float kOmega = 0.9f;   // Jacobi damping (default; option mg_omega).  Round 3: 0.8 -> 0.9 after a sweep on the product
                                 // (profiles/r03/mg_omega_*.txt: 23 -> 22 iterations at 1024^3 spheres, 48 -> 46 on 768^3 blobs; the
                                 // prototype agrees: 21 -> 20, 28 -> 27); any value < 1 keeps the cycle positive definite
constexpr float kScale = 1.5f;   // over-correction of the piecewise-constant coarse correction
constexpr int kCoarsestSweeps = 16;
constexpr long long kCoarsestCells = 512;
constexpr unsigned kWCycleMask = 14u;  // bit l: coarse level l (0 = first) is visited twice per parent visit;
                                       // levels 1-3 (profiles/r01/mg_cycle_sweep.txt)
constexpr int kCompactLevels = 3;      // link weights <= 4^(l+1) fit a byte, diagonals <= 6 * 4^(l+1) 16 bits
constexpr long long kReplicateCells = 1ll << 18;

struct Dims {
  int nx, ny, nzl, z0;  // owned planes [z0, z0 + nzl) of the level's global z range
  int nzg;              // planes of the whole level
  int periodic;         // 1: the homogenization cell problem (x, y wrap by index, z through the ghost planes)
  int zimg;             // 1: periodic AND the whole level lives on this rank: the ghost planes are the images of planes nzl and 1, and
                        // whoever writes a cell of plane 1 / nzl writes its image too (OI_MG_IMG) -- no image-plane copy afterwards
  long long sz;         // plane stride = nx * ny; arrays hold nzl + 2 planes (ghosts at 0 and nzl + 1)
  float omega;          // Jacobi damping of the cycle (MgHierarchy::omega, the same on every level)
};

template <class W, class D>
struct LevelT : Dims {
  const W *wx, *wy, *wz;  // link weight towards +x / +y / +z (0 = no link; 0 on the domain faces)
  const D* diag;          // 0 = cell does not exist
};

// one thread per cell of the local planes [kfirst, kfirst + nplanes); blocks are x-segments of
// rows, dealt to the 8 XCDs in contiguous z-chunks so that the +-z neighbours share an L2
#define OI_MG_CELL(L)                                                                          \
  int t_ = blockIdx.x;                                                                         \
  if ((gridDim.x & 7) == 0) t_ = (t_ & 7) * (gridDim.x >> 3) + (t_ >> 3);                      \
  const int xb_ = (L.nx + blockDim.x - 1) / blockDim.x;                                        \
  const int i = (t_ % xb_) * blockDim.x + threadIdx.x, j = (t_ / xb_) % L.ny, k = kfirst + t_ / (xb_ * L.ny); \
  if (i >= L.nx) return;                                                                       \
  const long long sy = L.nx, sz = L.sz;                                                        \
  const long long m = (long long)i + sy * j + sz * k;

// 1/d for the (small-integer) diagonals: one v_rcp_f32 (1 ulp) instead of the IEEE division sequence; the
// smoother only needs a fixed positive diagonal scaling
__device__ __forceinline__ float rcp(float d) { return __builtin_amdgcn_rcpf(d); }

// the image of a value just stored at cell m of local plane k (Dims::zimg; one plane only: both images)
#define OI_MG_IMG(L, arr, m, k, T, v)                                                       \
  do {                                                                                      \
    if (L.zimg) {                                                                           \
      if ((k) == 1) *reinterpret_cast<T*>((arr) + (m) + L.sz * L.nzl) = (v);                \
      if ((k) == L.nzl) *reinterpret_cast<T*>((arr) + (m) - L.sz * L.nzl) = (v);            \
    }                                                                                       \
  } while (0)

// indices of the -x / +x / -y / +y neighbours of cell (i, j) at m.  Non-periodic levels need no edge tests (the
// weights on the domain faces are zero and every index stays inside the ghosted array); periodic levels wrap.
#define OI_MG_NBRS(L)                                                                              \
  const long long mxm = (L.periodic && i == 0) ? m + (L.nx - 1) : m - 1;                            \
  const long long mxp = (L.periodic && i == L.nx - 1) ? m - (L.nx - 1) : m + 1;                     \
  const long long mym = (L.periodic && j == 0) ? m + sy * (L.ny - 1) : m - sy;                      \
  const long long myp = (L.periodic && j == L.ny - 1) ? m - sy * (L.ny - 1) : m + sy;

// sum over the six links of weight * neighbour value (the weight of a link lives with its lower cell)
template <class W, class D>
__device__ __forceinline__ float offdiag(const LevelT<W, D>& L, long long m, long long mxm, long long mym, float vxm,
                                         float vxp, float vym, float vyp, float vzm, float vzp) {
  float s = (float)L.wx[mxm] * vxm;
  s += (float)L.wx[m] * vxp;
  s += (float)L.wy[mym] * vym;
  s += (float)L.wy[m] * vyp;
  s += (float)L.wz[m - L.sz] * vzm;
  s += (float)L.wz[m] * vzp;
  return s;
}

// local plane of the coarse level C that holds the parent of plane kk (local, ghosts 0 and nzl + 1) of level L
__device__ __forceinline__ int parent_plane(const Dims& L, const Dims& C, int kk) {
  const int kg = L.z0 + kk - 1;
  if (L.periodic) {  // the planes beyond the two ends are images: their parents sit in C's ghost planes
    if (kg < 0) return 0;
    if (kg >= L.nzg) return C.nzl + 1;
  }
  return (kg >> 1) - C.z0 + 1;
}

// ------------------------------ hierarchy construction ------------------------------
// First coarse level from the fine codes: a fine cell is free iff code != 0; a fine link joins
// two free face neighbours; diag_fine = code (active neighbours incl. Dirichlet cells).  Only
// OWNED fine planes contribute (an aggregate straddling two slabs is summed across ranks).
__global__ void __launch_bounds__(256) k_mg_build1(Geom g, const uint8_t* __restrict__ code, Dims L, int kfirst,
                                                    float* __restrict__ wx, float* __restrict__ wy,
                                                    float* __restrict__ wz, float* __restrict__ diag) {
  OI_MG_CELL(L)
  (void)sy;
  auto fr = [&](int ii, int jj, int gz) -> int {  // free?  gz: local plane incl. the ghosts 0 and nzl + 1
    if (L.periodic) {
      ii = ii < 0 ? ii + g.nx : (ii >= g.nx ? ii - g.nx : ii);
      jj = jj < 0 ? jj + g.ny : (jj >= g.ny ? jj - g.ny : jj);
    }
    if (ii < 0 || jj < 0 || ii >= g.nx || jj >= g.ny) return 0;
    return code[(long long)gz * g.plane + (long long)jj * g.pitch + ii] != 0;
  };
  const int Kg = L.z0 + k - 1;
  int lx = 0, ly = 0, lz = 0, d = 0;
  for (int c = 0; c < 2; ++c) {
    const int kgf = 2 * Kg + c, gz = kgf - g.z0 + 1;
    if (gz < 1 || gz > g.nzl) continue;
    for (int b = 0; b < 2; ++b)
      for (int a = 0; a < 2; ++a) {
        const int ii = 2 * i + a, jj = 2 * j + b;
        if (ii >= g.nx || jj >= g.ny || !fr(ii, jj, gz)) continue;
        const int cd = code[(long long)gz * g.plane + (long long)jj * g.pitch + ii];
        // a child's partner inside the block (missing for the single child of an odd-sized level) keeps the link
        // off the coarse diagonal; the link through the block's + face becomes the coarse link
        const bool px = a == 1 || ii + 1 < g.nx, py = b == 1 || jj + 1 < g.ny, pz = c == 1 || kgf + 1 < g.nz;
        const int inside = (px ? fr(a ? ii - 1 : ii + 1, jj, gz) : 0) + (py ? fr(ii, b ? jj - 1 : jj + 1, gz) : 0) +
                           (pz ? fr(ii, jj, c ? gz - 1 : gz + 1) : 0);
        d += cd - inside;
        if (a == 1 || !px) lx += fr(ii + 1, jj, gz);
        if (b == 1 || !py) ly += fr(ii, jj + 1, gz);
        if (c == 1 || !pz) lz += fr(ii, jj, gz + 1);
      }
  }
  wx[m] = (float)lx; wy[m] = (float)ly; wz[m] = (float)lz; diag[m] = (float)d;
}

// level l+1 from level l (both coarse); again only owned children contribute
template <class W, class D>
__global__ void __launch_bounds__(256) k_mg_build(LevelT<W, D> F, Dims L, int kfirst, float* __restrict__ wx,
                                                   float* __restrict__ wy, float* __restrict__ wz,
                                                   float* __restrict__ diag) {
  OI_MG_CELL(L)
  (void)sy;
  const int Kg = L.z0 + k - 1;
  float lx = 0.f, ly = 0.f, lz = 0.f, d = 0.f;
  for (int c = 0; c < 2; ++c) {
    const int kl = 2 * Kg + c - F.z0 + 1;
    if (kl < 1 || kl > F.nzl) continue;
    for (int b = 0; b < 2; ++b)
      for (int a = 0; a < 2; ++a) {
        const int ii = 2 * i + a, jj = 2 * j + b;
        if (ii >= F.nx || jj >= F.ny) continue;
        const long long fm = (long long)ii + (long long)F.nx * jj + F.sz * kl;
        d += (float)F.diag[fm];
        // links towards +x/+y/+z: the ones inside the block (a/b/c == 0 with a partner in the block) leave the
        // diagonal twice, the ones through the block's + face become the coarse link -- unless the level is a
        // single periodic block in that direction: then the wrap link is inside the block as well
        const bool px = a == 0 && ii + 1 < F.nx, py = b == 0 && jj + 1 < F.ny, pz = c == 0 && (2 * Kg + c) + 1 < F.nzg;
        const float fx = (float)F.wx[fm], fy = (float)F.wy[fm], fz = (float)F.wz[fm];
        if (px || (L.periodic && L.nx == 1)) d -= 2.f * fx; else lx += fx;
        if (py || (L.periodic && L.ny == 1)) d -= 2.f * fy; else ly += fy;
        if (pz || (L.periodic && L.nzg == 1)) d -= 2.f * fz; else lz += fz;
      }
  }
  wx[m] = lx; wy[m] = ly; wz[m] = lz; diag[m] = d;
}

__global__ void __launch_bounds__(256) k_mg_compact(long long n, const float* __restrict__ wx, const float* __restrict__ wy,
                                                     const float* __restrict__ wz, const float* __restrict__ diag,
                                                     uint8_t* __restrict__ ox, uint8_t* __restrict__ oy,
                                                     uint8_t* __restrict__ oz, uint16_t* __restrict__ od) {
  for (long long m = (long long)blockIdx.x * blockDim.x + threadIdx.x; m < n; m += (long long)gridDim.x * blockDim.x) {
    ox[m] = (uint8_t)wx[m]; oy[m] = (uint8_t)wy[m]; oz[m] = (uint8_t)wz[m]; od[m] = (uint16_t)diag[m];
  }
}

// ------------------------------ coarse-level cycle kernels ------------------------------
// r = b - A (w b / D)   (first Jacobi sweep from zero is never stored; the post-smoother recomputes it)
template <class W, class D>
__global__ void __launch_bounds__(256) k_mg_pre(LevelT<W, D> L, int kfirst, const float* __restrict__ b,
                                                 float* __restrict__ r) {
  OI_MG_CELL(L)
  const float d = (float)L.diag[m];
  float res = 0.f;
  if (d > 0.f) {
    auto xs = [&](long long mm) -> float { const float dd = (float)L.diag[mm]; return dd > 0.f ? L.omega * b[mm] * rcp(dd) : 0.f; };
    OI_MG_NBRS(L)
    const float s = offdiag(L, m, mxm, mym, xs(mxm), xs(mxp), xs(mym), xs(myp), xs(m - sz), xs(m + sz));
    res = b[m] - (d * xs(m) - s);
  }
  r[m] = res;
}

// residual of a finished visit (the right-hand side of the second W-cycle visit): r = b - A x
template <class W, class D>
__global__ void __launch_bounds__(256) k_mg_resid(LevelT<W, D> L, int kfirst, const float* __restrict__ b,
                                                   const float* __restrict__ x, float* __restrict__ r) {
  OI_MG_CELL(L)
  const float d = (float)L.diag[m];
  float res = 0.f;
  if (d > 0.f) {
    OI_MG_NBRS(L)
    res = b[m] - (d * x[m] - offdiag(L, m, mxm, mym, x[mxm], x[mxp], x[mym], x[myp], x[m - sz], x[m + sz]));
  }
  r[m] = res;
  OI_MG_IMG(L, r, m, k, float, res);
}

// coarse right-hand side = P^T r (sum over the owned children); L is the COARSE level
__global__ void __launch_bounds__(256) k_mg_restrict(Dims F, const float* __restrict__ r, Dims L, int kfirst,
                                                      float* __restrict__ bc) {
  OI_MG_CELL(L)
  (void)sy;
  const int Kg = L.z0 + k - 1;
  float s = 0.f;
  for (int c = 0; c < 2; ++c) {
    const int kl = 2 * Kg + c - F.z0 + 1;
    if (kl < 1 || kl > F.nzl) continue;
    for (int bb = 0; bb < 2; ++bb) {
      const int fj = 2 * j + bb;
      if (fj >= F.ny) continue;
      const long long row = F.sz * kl + (long long)F.nx * fj;
      s += r[row + 2 * i];
      if (2 * i + 1 < F.nx) s += r[row + 2 * i + 1];
    }
  }
  bc[m] = s;
  OI_MG_IMG(L, bc, m, k, float, s);
}

__global__ void __launch_bounds__(256) k_mg_fine_restrict(Geom g, const uint8_t* __restrict__ code,
                                                           const double* __restrict__ r1, Dims L, int kfirst,
                                                           float* __restrict__ bc) {
  OI_MG_CELL(L)
  (void)sy;
  const int Kg = L.z0 + k - 1;
  double s = 0.0;
  for (int c = 0; c < 2; ++c) {
    const int gz = 2 * Kg + c - g.z0 + 1;
    if (gz < 1 || gz > g.nzl) continue;
    for (int bb = 0; bb < 2; ++bb) {
      const int fj = 2 * j + bb;
      if (fj >= g.ny) continue;
      const long long row = (long long)gz * g.plane + (long long)fj * g.pitch;
      // r1 vanishes on non-free voxels; the pad column holds code 0, so the pair is always readable
      const unsigned short cc = *reinterpret_cast<const unsigned short*>(code + row + 2 * i);
      if (cc & 0xff) s += r1[row + 2 * i];
      if (cc >> 8) s += r1[row + 2 * i + 1];
    }
  }
  bc[m] = (float)s;
}

// out = x2 + w (b - A x2) / D with x2 = w b / D + kScale * P (xc + xc2)   (prolongation + post-smoothing)
template <class W, class D>
__global__ void __launch_bounds__(256) k_mg_post(LevelT<W, D> L, int kfirst, const float* __restrict__ b,
                                                  const float* __restrict__ xc, const float* __restrict__ xc2, Dims C,
                                                  float scale, const float* __restrict__ xpre, float* __restrict__ out) {
  OI_MG_CELL(L)
  auto x2 = [&](long long mm, int ii, int jj, int kk) -> float {
    const float dd = (float)L.diag[mm];
    if (!(dd > 0.f)) return 0.f;
    const long long cm = (long long)(ii >> 1) + (long long)C.nx * (jj >> 1) + C.sz * parent_plane(L, C, kk);
    float c = xc[cm];
    if (xc2) c += xc2[cm];
    return xpre ? xpre[mm] + scale * c : L.omega * b[mm] * rcp(dd) + scale * c;
  };
  float o = 0.f;
  const float d = (float)L.diag[m];
  if (d > 0.f) {
    OI_MG_NBRS(L)
    const bool per = L.periodic != 0;
    const int ixm = i > 0 ? i - 1 : L.nx - 1, ixp = i < L.nx - 1 ? i + 1 : 0, jym = j > 0 ? j - 1 : L.ny - 1, jyp = j < L.ny - 1 ? j + 1 : 0;
    const float c = x2(m, i, j, k);
    const float s = offdiag(L, m, mxm, mym, (per || i > 0) ? x2(mxm, ixm, j, k) : 0.f, (per || i < L.nx - 1) ? x2(mxp, ixp, j, k) : 0.f,
                            (per || j > 0) ? x2(mym, i, jym, k) : 0.f, (per || j < L.ny - 1) ? x2(myp, i, jyp, k) : 0.f,
                            x2(m - sz, i, j, k - 1), x2(m + sz, i, j, k + 1));
    o = c + L.omega * (b[m] - (d * c - s)) * rcp(d);
  }
  out[m] = o;
  OI_MG_IMG(L, out, m, k, float, o);
}

// ---- the same two kernels, FOUR cells along x per thread (byte / 16-bit levels whose nx is a multiple of 4) ----
// The one-cell kernels issue ~20 (pre) / ~35 (post) scalar loads per cell, nearly all of them cache hits: at 512^3
// they run at 1 TB/s, bound by the number of load instructions.  Four cells per thread share 16-byte row loads and
// keep the x neighbours in registers: the same arithmetic in the same order (bit-identical results) from a
// quarter of the loads.
#define OI_MG_CELL4(L)                                                                          \
  int t_ = blockIdx.x;                                                                          \
  if ((gridDim.x & 7) == 0) t_ = (t_ & 7) * (gridDim.x >> 3) + (t_ >> 3);                       \
  const int gpr_ = L.nx >> 2;                                                                   \
  const long long id_ = (long long)t_ * blockDim.x + threadIdx.x;                               \
  const long long row_ = id_ / gpr_;                                                            \
  const int i = (int)(id_ - row_ * gpr_) << 2, j = (int)(row_ % L.ny), k = kfirst + (int)(row_ / L.ny); \
  if (k >= kfirst + nplanes) return;                                                            \
  const long long sy = L.nx, sz = L.sz;                                                         \
  const long long m = (long long)i + sy * j + sz * k;

__device__ __forceinline__ void ld4(const float* p, float (&v)[4]) {
  const float4 t = *reinterpret_cast<const float4*>(p);
  v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
__device__ __forceinline__ void ld4(const uint16_t* p, float (&v)[4]) {
  const uint2 t = *reinterpret_cast<const uint2*>(p);
  v[0] = (float)(t.x & 0xffffu); v[1] = (float)(t.x >> 16); v[2] = (float)(t.y & 0xffffu); v[3] = (float)(t.y >> 16);
}
__device__ __forceinline__ void ld4(const uint8_t* p, float (&v)[4]) {
  const unsigned t = *reinterpret_cast<const unsigned*>(p);
  v[0] = (float)(t & 0xffu); v[1] = (float)((t >> 8) & 0xffu); v[2] = (float)((t >> 16) & 0xffu); v[3] = (float)(t >> 24);
}

// k_mg_lin4<SRC, OUT>: one linear step on a level, four cells per thread.  u = the first Jacobi iterate w b / D formed on the
// fly (SRC 0) or a stored vector x (SRC 1); res = b - A u; the result is res (OUT 0) or the smoothed u + w res / D (OUT 1):
//   <0, 0> pre-smoothing residual (one sweep from zero)          <1, 0> residual of a stored iterate
//   <0, 1> second Jacobi iterate (two sweeps from zero)          <1, 1> one Jacobi sweep on a stored iterate
// Same arithmetic in the same order as the one-cell kernels k_mg_pre / k_mg_resid / k_mg_jacobi (bit-identical results).
template <class W, class D, int SRC, int OUT>
__global__ void __launch_bounds__(256) k_mg_lin4(LevelT<W, D> L, int kfirst, int nplanes, const float* __restrict__ b,
                                                  const float* __restrict__ x, float* __restrict__ r) {
  OI_MG_CELL4(L)
  auto xs1 = [&](float bb, float dd) -> float { return dd > 0.f ? L.omega * bb * rcp(dd) : 0.f; };
  auto u4 = [&](long long mm, float (&o)[4]) {
    if (SRC == 1) { ld4(x + mm, o); return; }
    float bb[4], dd[4];
    ld4(b + mm, bb); ld4(L.diag + mm, dd);
#pragma unroll
    for (int c = 0; c < 4; ++c) o[c] = xs1(bb[c], dd[c]);
  };
  auto u1 = [&](long long mm) -> float { return SRC == 1 ? x[mm] : xs1(b[mm], (float)L.diag[mm]); };
  const long long mxm = (L.periodic && i == 0) ? m + (L.nx - 1) : m - 1;                 // left of cell i
  const long long mxp = (L.periodic && i + 3 == L.nx - 1) ? m + 3 - (L.nx - 1) : m + 4;  // right of cell i + 3
  const long long mym = (L.periodic && j == 0) ? m + sy * (L.ny - 1) : m - sy;
  const long long myp = (L.periodic && j == L.ny - 1) ? m - sy * (L.ny - 1) : m + sy;
  float bc[4], dc[4], xc_[4], xym[4], xyp[4], xzm[4], xzp[4], wxc[4], wyc[4], wym[4], wzc[4], wzm[4];
  ld4(L.diag + m, dc);
  // four cells without an operator row (solid all around): zeros out, none of the twenty neighbour loads (a sphere pack leaves
  // 40-45 % of the first coarse level's cells dead, in runs)
  if (!(dc[0] > 0.f || dc[1] > 0.f || dc[2] > 0.f || dc[3] > 0.f)) {
    *reinterpret_cast<float4*>(r + m) = make_float4(0.f, 0.f, 0.f, 0.f);
    OI_MG_IMG(L, r, m, k, float4, make_float4(0.f, 0.f, 0.f, 0.f));
    return;
  }
  ld4(b + m, bc);
  if (SRC == 1) {
    ld4(x + m, xc_);
  } else {
#pragma unroll
    for (int c = 0; c < 4; ++c) xc_[c] = xs1(bc[c], dc[c]);
  }
  const float xl = u1(mxm), xr = u1(mxp);
  u4(mym, xym); u4(myp, xyp); u4(m - sz, xzm); u4(m + sz, xzp);
  ld4(L.wx + m, wxc); ld4(L.wy + m, wyc); ld4(L.wy + mym, wym); ld4(L.wz + m, wzc); ld4(L.wz + (m - sz), wzm);
  const float wxl = (float)L.wx[mxm];
  float res[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    res[c] = 0.f;
    if (dc[c] > 0.f) {
      float s = (c == 0 ? wxl : wxc[c > 0 ? c - 1 : 0]) * (c == 0 ? xl : xc_[c > 0 ? c - 1 : 0]);
      s += wxc[c] * (c == 3 ? xr : xc_[c < 3 ? c + 1 : 3]);
      s += wym[c] * xym[c];
      s += wyc[c] * xyp[c];
      s += wzm[c] * xzm[c];
      s += wzc[c] * xzp[c];
      const float rr = bc[c] - (dc[c] * xc_[c] - s);
      res[c] = OUT == 0 ? rr : xc_[c] + L.omega * rr * rcp(dc[c]);
    }
  }
  *reinterpret_cast<float4*>(r + m) = make_float4(res[0], res[1], res[2], res[3]);
  OI_MG_IMG(L, r, m, k, float4, make_float4(res[0], res[1], res[2], res[3]));
}

template <class W, class D>
__global__ void __launch_bounds__(256) k_mg_post4(LevelT<W, D> L, int kfirst, int nplanes, const float* __restrict__ b,
                                                   const float* __restrict__ xc, const float* __restrict__ xc2, Dims C,
                                                   float scale, const float* __restrict__ xpre, float* __restrict__ out) {
  OI_MG_CELL4(L)
  const bool per = L.periodic != 0;
  // xpre != null: the iterate before the coarse correction is a stored vector (two pre-smoothing sweeps), not w b / D
  auto x21 = [&](float bb, float dd, float cc) -> float { return dd > 0.f ? L.omega * bb * rcp(dd) + scale * cc : 0.f; };
  auto x21p = [&](float xp, float dd, float cc) -> float { return dd > 0.f ? xp + scale * cc : 0.f; };
  auto par = [&](long long cm) -> float {
    float c = xc[cm];
    if (xc2) c += xc2[cm];
    return c;
  };
  // x2 of the four cells i .. i + 3 of row jj in plane kk (array offset mm); their parents are (i >> 1) and (i >> 1) + 1
  auto x24 = [&](long long mm, int jj, int kk, float (&o)[4]) {
    float bb[4], dd[4];
    ld4((xpre ? xpre : b) + mm, bb); ld4(L.diag + mm, dd);
    const long long cm = (long long)(i >> 1) + (long long)C.nx * (jj >> 1) + C.sz * parent_plane(L, C, kk);
    float2 p = *reinterpret_cast<const float2*>(xc + cm);
    if (xc2) { const float2 q = *reinterpret_cast<const float2*>(xc2 + cm); p.x += q.x; p.y += q.y; }
#pragma unroll
    for (int c = 0; c < 4; ++c) o[c] = xpre ? x21p(bb[c], dd[c], c < 2 ? p.x : p.y) : x21(bb[c], dd[c], c < 2 ? p.x : p.y);
  };
  auto zero4 = [](float (&o)[4]) { o[0] = o[1] = o[2] = o[3] = 0.f; };
  const long long mxm = (L.periodic && i == 0) ? m + (L.nx - 1) : m - 1;
  const long long mxp = (L.periodic && i + 3 == L.nx - 1) ? m + 3 - (L.nx - 1) : m + 4;
  const long long mym = (L.periodic && j == 0) ? m + sy * (L.ny - 1) : m - sy;
  const long long myp = (L.periodic && j == L.ny - 1) ? m - sy * (L.ny - 1) : m + sy;
  const int ixm = i > 0 ? i - 1 : L.nx - 1, ixp = i + 3 < L.nx - 1 ? i + 4 : 0;
  const int jym = j > 0 ? j - 1 : L.ny - 1, jyp = j < L.ny - 1 ? j + 1 : 0;
  float dc[4], bc[4], cc[4], xym[4], xyp[4], xzm[4], xzp[4], wxc[4], wyc[4], wym[4], wzc[4], wzm[4];
  ld4(L.diag + m, dc);
  if (!(dc[0] > 0.f || dc[1] > 0.f || dc[2] > 0.f || dc[3] > 0.f)) {  // (see k_mg_lin4)
    *reinterpret_cast<float4*>(out + m) = make_float4(0.f, 0.f, 0.f, 0.f);
    OI_MG_IMG(L, out, m, k, float4, make_float4(0.f, 0.f, 0.f, 0.f));
    return;
  }
  ld4(b + m, bc);
  x24(m, j, k, cc);
  const long long prow = (long long)C.nx * (j >> 1) + C.sz * parent_plane(L, C, k);
  auto x2e = [&](long long mm, long long cm) -> float {
    return xpre ? x21p(xpre[mm], (float)L.diag[mm], par(cm)) : x21(b[mm], (float)L.diag[mm], par(cm));
  };
  const float xl = (per || i > 0) ? x2e(mxm, (long long)(ixm >> 1) + prow) : 0.f;
  const float xr = (per || i + 3 < L.nx - 1) ? x2e(mxp, (long long)(ixp >> 1) + prow) : 0.f;
  if (per || j > 0) x24(mym, jym, k, xym); else zero4(xym);
  if (per || j < L.ny - 1) x24(myp, jyp, k, xyp); else zero4(xyp);
  x24(m - sz, j, k - 1, xzm); x24(m + sz, j, k + 1, xzp);
  ld4(L.wx + m, wxc); ld4(L.wy + m, wyc); ld4(L.wy + mym, wym); ld4(L.wz + m, wzc); ld4(L.wz + (m - sz), wzm);
  const float wxl = (float)L.wx[mxm];
  float o[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    o[c] = 0.f;
    if (dc[c] > 0.f) {
      float s = (c == 0 ? wxl : wxc[c > 0 ? c - 1 : 0]) * (c == 0 ? xl : cc[c > 0 ? c - 1 : 0]);
      s += wxc[c] * (c == 3 ? xr : cc[c < 3 ? c + 1 : 3]);
      s += wym[c] * xym[c];
      s += wyc[c] * xyp[c];
      s += wzm[c] * xzm[c];
      s += wzc[c] * xzp[c];
      o[c] = cc[c] + L.omega * (bc[c] - (dc[c] * cc[c] - s)) * rcp(dc[c]);
    }
  }
  *reinterpret_cast<float4*>(out + m) = make_float4(o[0], o[1], o[2], o[3]);
  OI_MG_IMG(L, out, m, k, float4, make_float4(o[0], o[1], o[2], o[3]));
}

// ---- round 4 (late): the correction + post-smoothing step of a big byte / 16-bit level through an LDS TILE ----
// k_mg_post4 forms the corrected iterate x2 = w b / D + s e (e = the parent's correction) of its four cells AND of their eighteen
// face neighbours from global memory (b, D and e of every neighbour: ~27 load instructions per four cells, cache hits, and the
// kernel is bound by them: 0.65 ms on the 512^3 level of a 1024^3 solve against 0.3 ms for its bytes).  Here a block of 512 threads
// owns a tile of 64 x 8 x 8 cells: x2 of the tile and of its six faces goes to LDS once (1.53 evaluations per cell instead of
// 5.5), the stencil reads its neighbours there.  Same arithmetic in the same order: bit-identical to k_mg_post4 / k_mg_post (option
// mg_coarse_tiled = 0 keeps those kernels as the cross-check).  Out-of-domain positions hold zeros -- the value k_mg_post4 uses
// there; a periodic level finds the wrapped row / quad / column there, the image planes in the ghost planes as ever.
// ANY nx (>= 64): the rows of a level whose nx is not a multiple of four start anywhere, so the quads are read and written through
// under-aligned vector types (one instruction on this target), and the last quad of a row that ends past it moves the cells that
// exist one by one -- k_mg_post4 needs nx % 4 == 0, and such levels (a 500^3 or 900^3 volume's first coarse level) fell back to
// the one-cell kernels: 2.8 instead of 2.1 ms per iteration at 500^3 against 512^3.
constexpr int kTX = 64, kTY = 8, kTZ = 8;       // cells of a tile
constexpr int kTRP = kTX + 8;                    // LDS row pitch in floats: [3 pad | left halo | 64 cells | right halo | 3 pad]
constexpr int kTPP = (kTY + 2) * kTRP;           // LDS plane pitch
typedef float f4u_t __attribute__((ext_vector_type(4), aligned(4)));
typedef float f2u_t __attribute__((ext_vector_type(2), aligned(4)));
typedef unsigned short us4u_t __attribute__((ext_vector_type(4), aligned(2)));
typedef unsigned char uc4u_t __attribute__((ext_vector_type(4), aligned(1)));
// the first nv (1 .. 4) of four consecutive values, the others 0
__device__ __forceinline__ void ldq(const float* p, int nv, float (&v)[4]) {
  if (nv == 4) { const f4u_t t = *reinterpret_cast<const f4u_t*>(p); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; return; }
#pragma unroll
  for (int c = 0; c < 4; ++c) v[c] = c < nv ? p[c] : 0.f;
}
__device__ __forceinline__ void ldq(const uint16_t* p, int nv, float (&v)[4]) {
  if (nv == 4) { const us4u_t t = *reinterpret_cast<const us4u_t*>(p); v[0] = (float)t.x; v[1] = (float)t.y; v[2] = (float)t.z; v[3] = (float)t.w; return; }
#pragma unroll
  for (int c = 0; c < 4; ++c) v[c] = c < nv ? (float)p[c] : 0.f;
}
__device__ __forceinline__ void ldq(const uint8_t* p, int nv, float (&v)[4]) {
  if (nv == 4) { const uc4u_t t = *reinterpret_cast<const uc4u_t*>(p); v[0] = (float)t.x; v[1] = (float)t.y; v[2] = (float)t.z; v[3] = (float)t.w; return; }
#pragma unroll
  for (int c = 0; c < 4; ++c) v[c] = c < nv ? (float)p[c] : 0.f;
}
__device__ __forceinline__ void stq(float* p, int nv, float a, float b, float c, float d) {
  if (nv == 4) { f4u_t t; t.x = a; t.y = b; t.z = c; t.w = d; *reinterpret_cast<f4u_t*>(p) = t; return; }
  p[0] = a;
  if (nv > 1) p[1] = b;
  if (nv > 2) p[2] = c;
}
// the corrections of the two parents of a quad (the second one only where the quad reaches it)
__device__ __forceinline__ float2 ldpar(const float* xc, const float* xc2, long long cm, int nv) {
  float2 p;
  if (nv > 2) {
    const f2u_t t = *reinterpret_cast<const f2u_t*>(xc + cm);
    p = make_float2(t.x, t.y);
    if (xc2) { const f2u_t q = *reinterpret_cast<const f2u_t*>(xc2 + cm); p.x += q.x; p.y += q.y; }
  } else {
    p = make_float2(xc[cm], 0.f);
    if (xc2) p.x += xc2[cm];
  }
  return p;
}
// AL (a template parameter of the tiled kernels): nx % 4 == 0 -- every quad is whole and 16-byte aligned, the loads and stores are the
// aligned vector ones (the under-aligned types above compile to one instruction per ELEMENT: fine for the levels that would otherwise
// run the one-cell kernels, 10 % of the cycle lost on the levels that need none of it)
template <bool AL, class P>
__device__ __forceinline__ void ldqa(const P* p, int nv, float (&v)[4]) {
  if (AL) ld4(p, v); else ldq(p, nv, v);
}
template <bool AL>
__device__ __forceinline__ void stqa(float* p, int nv, float a, float b, float c, float d) {
  if (AL) *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d); else stq(p, nv, a, b, c, d);
}
template <bool AL>
__device__ __forceinline__ float2 ldpara(const float* xc, const float* xc2, long long cm, int nv) {
  if (!AL) return ldpar(xc, xc2, cm, nv);
  float2 p = *reinterpret_cast<const float2*>(xc + cm);
  if (xc2) { const float2 q = *reinterpret_cast<const float2*>(xc2 + cm); p.x += q.x; p.y += q.y; }
  return p;
}
template <class W, class D, bool AL>
__global__ void __launch_bounds__(512) k_mg_post4t(LevelT<W, D> L, const float* __restrict__ b, const float* __restrict__ xc,
                                                    const float* __restrict__ xc2, Dims C, float scale, const float* __restrict__ xpre,
                                                    float* __restrict__ out, int tiles_x, int tiles_y) {
  __shared__ __attribute__((aligned(16))) float T[(kTZ + 2) * kTPP];
  int t_ = blockIdx.x;
  if ((gridDim.x & 7) == 0) t_ = (t_ & 7) * (gridDim.x >> 3) + (t_ >> 3);  // (every XCD a contiguous z range of tiles)
  const int bx = t_ % tiles_x, by = (t_ / tiles_x) % tiles_y, bz = t_ / (tiles_x * tiles_y);
  const int i0 = bx * kTX, j0 = by * kTY, k0 = 1 + bz * kTZ;  // first cell of the tile (k: local plane, ghosts 0 and nzl + 1)
  const long long sy = L.nx, sz = L.sz;
  const bool per = L.periodic != 0;
  auto x21 = [&](float bb, float dd, float cc) -> float { return dd > 0.f ? L.omega * bb * rcp(dd) + scale * cc : 0.f; };
  auto x21p = [&](float xp, float dd, float cc) -> float { return dd > 0.f ? xp + scale * cc : 0.f; };
  // where a position of the tile or of its faces lives: the wrapped index of a periodic level one step beyond an end, -1 = nowhere
  auto row_of = [&](int jj) { return (jj >= 0 && jj < L.ny) ? jj : (per && jj == -1 ? L.ny - 1 : (per && jj == L.ny ? 0 : -1)); };
  auto col_of = [&](int ii) { return (ii >= 0 && ii < L.nx) ? ii : (per && ii == -1 ? L.nx - 1 : (per && ii == L.nx ? 0 : -1)); };
  auto nv_of = [&](int ii) { return L.nx - ii < 4 ? L.nx - ii : 4; };  // cells of the quad that starts at column ii (< nx)
  // x2 of the cells ii .. ii + nv - 1 of row jj in plane kk (all inside the arrays), zeros behind them
  auto x24 = [&](int ii, int jj, int kk, int nv, float (&o)[4]) {
    const long long mm = (long long)ii + sy * jj + sz * kk;
    float bb[4], dd[4];
    ldqa<AL>(L.diag + mm, nv, dd);
    if (!(dd[0] > 0.f || dd[1] > 0.f || dd[2] > 0.f || dd[3] > 0.f)) { o[0] = o[1] = o[2] = o[3] = 0.f; return; }
    ldqa<AL>((xpre ? xpre : b) + mm, nv, bb);
    const long long cm = (long long)(ii >> 1) + (long long)C.nx * (jj >> 1) + C.sz * parent_plane(L, C, kk);
    const float2 p = ldpara<AL>(xc, xc2, cm, nv);
#pragma unroll
    for (int c = 0; c < 4; ++c) o[c] = xpre ? x21p(bb[c], dd[c], c < 2 ? p.x : p.y) : x21(bb[c], dd[c], c < 2 ? p.x : p.y);
  };
  auto x2one = [&](int ii, int jj, int kk) -> float {
    const long long mm = (long long)ii + sy * jj + sz * kk;
    const float dd = (float)L.diag[mm];
    if (!(dd > 0.f)) return 0.f;
    const long long cm = (long long)(ii >> 1) + (long long)C.nx * (jj >> 1) + C.sz * parent_plane(L, C, kk);
    float c = xc[cm];
    if (xc2) c += xc2[cm];
    return xpre ? x21p(xpre[mm], dd, c) : x21(b[mm], dd, c);
  };
  const int tid = threadIdx.x;
  const int tq = tid & 15, ty = (tid >> 4) & 7, tz = tid >> 7;  // own quads: (tq, ty, tz) and (tq, ty, tz + 4)
  auto lds_at = [&](int lq, int ly, int lz) { return (lz + 1) * kTPP + (ly + 1) * kTRP + 4 + 4 * lq; };  // quad lq of tile row ly, plane lz
  // ---- own quads: x2 to registers and LDS (a position outside the level: zeros, or what a periodic level wraps there) ----
  float dc[2][4], bc[2][4], cc[2][4];
  bool mine[2];
  const int ii = i0 + 4 * tq;
  const int nvq = ii < L.nx ? nv_of(ii) : 4;  // (a quad that starts exactly at the end of a periodic level is the wrapped quad 0: whole)
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int lz = tz + 4 * u;
    const int jj = j0 + ty, kk = k0 + lz;
    mine[u] = ii < L.nx && jj < L.ny && kk <= L.nzl;
    const int iw = ii < L.nx ? ii : col_of(ii), jw = row_of(jj);  // (a quad starts inside the level or exactly at its end)
    cc[u][0] = cc[u][1] = cc[u][2] = cc[u][3] = 0.f;
    dc[u][0] = dc[u][1] = dc[u][2] = dc[u][3] = 0.f;
    if (iw >= 0 && jw >= 0 && kk <= L.nzl + 1) {
      if (mine[u]) {
        const long long m = (long long)ii + sy * jj + sz * kk;
        ldqa<AL>(L.diag + m, nvq, dc[u]);
        if (dc[u][0] > 0.f || dc[u][1] > 0.f || dc[u][2] > 0.f || dc[u][3] > 0.f) {
          ldqa<AL>(b + m, nvq, bc[u]);
          float xs[4] = {0.f, 0.f, 0.f, 0.f};
          if (xpre) ldqa<AL>(xpre + m, nvq, xs);
          const long long cm = (long long)(ii >> 1) + (long long)C.nx * (jj >> 1) + C.sz * parent_plane(L, C, kk);
          const float2 p = ldpara<AL>(xc, xc2, cm, nvq);
#pragma unroll
          for (int c = 0; c < 4; ++c)
            cc[u][c] = xpre ? x21p(xs[c], dc[u][c], c < 2 ? p.x : p.y) : x21(bc[u][c], dc[u][c], c < 2 ? p.x : p.y);
        }
        // a row that ends inside this quad: the slot behind its last cell is that cell's +x neighbour -- column 0 of a periodic level
        // (the open level keeps the zero k_mg_post4 uses there)
        if (per && nvq < 4) {
          const float wv = x2one(0, jj, kk);
          if (nvq == 1) cc[u][1] = wv; else if (nvq == 2) cc[u][2] = wv; else cc[u][3] = wv;
        }
      } else {
        x24(iw, jw, kk, iw == ii ? nvq : 4, cc[u]);
      }
    }
    *reinterpret_cast<float4*>(&T[lds_at(tq, ty, lz)]) = make_float4(cc[u][0], cc[u][1], cc[u][2], cc[u][3]);
  }
  // ---- the six faces ----
  {
    float h[4];
    // y faces: 8 planes x 2 sides x 16 quads (threads 0 .. 255); z faces: 8 rows x 2 sides x 16 quads (threads 256 .. 511)
    const int hq = tid & 15, side = (tid >> 4) & 1, a = (tid >> 5) & 7;
    const bool zface = tid >= 256;
    const int ly = zface ? a : (side ? kTY : -1), lz = zface ? (side ? kTZ : -1) : a;
    const int hi = i0 + 4 * hq, jj = j0 + ly, kk = k0 + lz;
    const int iw = hi < L.nx ? hi : col_of(hi), jw = row_of(jj);
    h[0] = h[1] = h[2] = h[3] = 0.f;
    if (iw >= 0 && jw >= 0 && kk >= 0 && kk <= L.nzl + 1) x24(iw, jw, kk, nv_of(iw), h);
    *reinterpret_cast<float4*>(&T[lds_at(hq, ly, lz)]) = make_float4(h[0], h[1], h[2], h[3]);
    // x faces: 8 planes x 8 rows x 2 sides, one cell each (threads 0 .. 127)
    if (tid < 128) {
      const int xs_ = tid & 1, xy = (tid >> 1) & 7, xz = tid >> 4;
      const int xi = xs_ ? i0 + kTX : i0 - 1, xj = j0 + xy, xk = k0 + xz;
      const int xiw = col_of(xi);
      float v = 0.f;
      if (xiw >= 0 && xj < L.ny && xk <= L.nzl + 1) v = x2one(xiw, xj, xk);
      T[(xz + 1) * kTPP + (xy + 1) * kTRP + (xs_ ? 4 + kTX : 3)] = v;
    }
  }
  __syncthreads();
  // ---- the stencil of the own quads ----
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    if (!mine[u]) continue;
    const int lz = tz + 4 * u;
    const int i = ii, j = j0 + ty, k = k0 + lz;
    const long long m = (long long)i + sy * j + sz * k;
    if (!(dc[u][0] > 0.f || dc[u][1] > 0.f || dc[u][2] > 0.f || dc[u][3] > 0.f)) {
      stqa<AL>(out + m, nvq, 0.f, 0.f, 0.f, 0.f);
      if (L.zimg) {
        if (k == 1) stqa<AL>(out + m + L.sz * L.nzl, nvq, 0.f, 0.f, 0.f, 0.f);
        if (k == L.nzl) stqa<AL>(out + m - L.sz * L.nzl, nvq, 0.f, 0.f, 0.f, 0.f);
      }
      continue;
    }
    const int pos = lds_at(tq, ty, lz);
    const float xl = T[pos - 1], xr = T[pos + 4];
    const float4 ym4 = *reinterpret_cast<const float4*>(&T[pos - kTRP]), yp4 = *reinterpret_cast<const float4*>(&T[pos + kTRP]);
    const float4 zm4 = *reinterpret_cast<const float4*>(&T[pos - kTPP]), zp4 = *reinterpret_cast<const float4*>(&T[pos + kTPP]);
    const float xym[4] = {ym4.x, ym4.y, ym4.z, ym4.w}, xyp[4] = {yp4.x, yp4.y, yp4.z, yp4.w};
    const float xzm[4] = {zm4.x, zm4.y, zm4.z, zm4.w}, xzp[4] = {zp4.x, zp4.y, zp4.z, zp4.w};
    const long long mxm = (per && i == 0) ? m + (L.nx - 1) : m - 1;
    const long long mym = (per && j == 0) ? m + sy * (L.ny - 1) : m - sy;
    float wxc[4], wyc[4], wym[4], wzc[4], wzm[4];
    ldqa<AL>(L.wx + m, nvq, wxc); ldqa<AL>(L.wy + m, nvq, wyc); ldqa<AL>(L.wy + mym, nvq, wym); ldqa<AL>(L.wz + m, nvq, wzc); ldqa<AL>(L.wz + (m - sz), nvq, wzm);
    const float wxl = (float)L.wx[mxm];
    float o[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      o[c] = 0.f;
      if (dc[u][c] > 0.f) {
        float s = (c == 0 ? wxl : wxc[c > 0 ? c - 1 : 0]) * (c == 0 ? xl : cc[u][c > 0 ? c - 1 : 0]);
        s += wxc[c] * (c == 3 ? xr : cc[u][c < 3 ? c + 1 : 3]);
        s += wym[c] * xym[c];
        s += wyc[c] * xyp[c];
        s += wzm[c] * xzm[c];
        s += wzc[c] * xzp[c];
        o[c] = cc[u][c] + L.omega * (bc[u][c] - (dc[u][c] * cc[u][c] - s)) * rcp(dc[u][c]);
      }
    }
    stqa<AL>(out + m, nvq, o[0], o[1], o[2], o[3]);
    if (L.zimg) {
      if (k == 1) stqa<AL>(out + m + L.sz * L.nzl, nvq, o[0], o[1], o[2], o[3]);
      if (k == L.nzl) stqa<AL>(out + m - L.sz * L.nzl, nvq, o[0], o[1], o[2], o[3]);
    }
  }
}

// The pre-smoothing residual through the same tile: u = w b / D of the tile and its faces in LDS, res = b - A u of the own cells;
// RESTRICT: the residuals stay in LDS and 256 threads sum the eight children of the tile's 32 x 4 x 4 coarse cells in
// k_mg_restrict's order (plane, row, x) -- no residual array, no restriction launch, what k_mg_lin4r does with 4 x 2 x 2 cells per
// thread, on open AND periodic levels (even extents, the whole level and its child on this rank).  Bit-identical to k_mg_lin4<0, 0>
// (+ k_mg_restrict) and to k_mg_lin4r.
// SRC / OUT as in k_mg_lin4: u = w b / D formed on the fly (SRC 0) or a stored vector x (SRC 1); the result is res (OUT 0) or the
// smoothed u + w res / D (OUT 1) -- the four steps of a level that runs two sweeps before and after the correction.
template <class W, class D, bool AL, bool RESTRICT, int SRC = 0, int OUT = 0>
__global__ void __launch_bounds__(512) k_mg_lin4t(LevelT<W, D> L, const float* __restrict__ b, float* __restrict__ r, Dims C,
                                                   float* __restrict__ bcoarse, int tiles_x, int tiles_y, const float* __restrict__ x = nullptr) {
  static_assert(!RESTRICT || (SRC == 0 && OUT == 0), "the restriction takes the pre-smoothing residual");
  __shared__ __attribute__((aligned(16))) float T[(kTZ + 2) * kTPP];
  __shared__ __attribute__((aligned(16))) float R[RESTRICT ? kTZ * kTY * kTX : 4];
  int t_ = blockIdx.x;
  if ((gridDim.x & 7) == 0) t_ = (t_ & 7) * (gridDim.x >> 3) + (t_ >> 3);
  const int bx = t_ % tiles_x, by = (t_ / tiles_x) % tiles_y, bz = t_ / (tiles_x * tiles_y);
  const int i0 = bx * kTX, j0 = by * kTY, k0 = 1 + bz * kTZ;
  const long long sy = L.nx, sz = L.sz;
  const bool per = L.periodic != 0;
  auto xs1 = [&](float bb, float dd) -> float { return dd > 0.f ? L.omega * bb * rcp(dd) : 0.f; };
  auto row_of = [&](int jj) { return (jj >= 0 && jj < L.ny) ? jj : (per && jj == -1 ? L.ny - 1 : (per && jj == L.ny ? 0 : -1)); };
  auto col_of = [&](int ii) { return (ii >= 0 && ii < L.nx) ? ii : (per && ii == -1 ? L.nx - 1 : (per && ii == L.nx ? 0 : -1)); };
  auto nv_of = [&](int ii) { return L.nx - ii < 4 ? L.nx - ii : 4; };
  auto u4 = [&](int ii, int jj, int kk, int nv, float (&o)[4]) {
    const long long mm = (long long)ii + sy * jj + sz * kk;
    if (SRC == 1) { ldqa<AL>(x + mm, nv, o); return; }
    float bb[4], dd[4];
    ldqa<AL>(L.diag + mm, nv, dd);
    if (!(dd[0] > 0.f || dd[1] > 0.f || dd[2] > 0.f || dd[3] > 0.f)) { o[0] = o[1] = o[2] = o[3] = 0.f; return; }
    ldqa<AL>(b + mm, nv, bb);
#pragma unroll
    for (int c = 0; c < 4; ++c) o[c] = xs1(bb[c], dd[c]);
  };
  auto u1 = [&](int ii, int jj, int kk) -> float {
    const long long mm = (long long)ii + sy * jj + sz * kk;
    return SRC == 1 ? x[mm] : xs1(b[mm], (float)L.diag[mm]);
  };
  const int tid = threadIdx.x;
  const int tq = tid & 15, ty = (tid >> 4) & 7, tz = tid >> 7;
  auto lds_at = [&](int lq, int ly, int lz) { return (lz + 1) * kTPP + (ly + 1) * kTRP + 4 + 4 * lq; };
  float dc[2][4], bc[2][4], uc[2][4];
  bool mine[2];
  const int ii = i0 + 4 * tq;
  const int nvq = ii < L.nx ? nv_of(ii) : 4;
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int lz = tz + 4 * u;
    const int jj = j0 + ty, kk = k0 + lz;
    mine[u] = ii < L.nx && jj < L.ny && kk <= L.nzl;
    const int iw = ii < L.nx ? ii : col_of(ii), jw = row_of(jj);
    uc[u][0] = uc[u][1] = uc[u][2] = uc[u][3] = 0.f;
    dc[u][0] = dc[u][1] = dc[u][2] = dc[u][3] = 0.f;
    bc[u][0] = bc[u][1] = bc[u][2] = bc[u][3] = 0.f;
    if (iw >= 0 && jw >= 0 && kk <= L.nzl + 1) {
      if (mine[u]) {
        const long long m = (long long)ii + sy * jj + sz * kk;
        ldqa<AL>(L.diag + m, nvq, dc[u]);
        if (dc[u][0] > 0.f || dc[u][1] > 0.f || dc[u][2] > 0.f || dc[u][3] > 0.f) {
          ldqa<AL>(b + m, nvq, bc[u]);
          if (SRC == 1) {
            ldqa<AL>(x + m, nvq, uc[u]);
          } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) uc[u][c] = xs1(bc[u][c], dc[u][c]);
          }
        } else if (SRC == 1) {
          ldqa<AL>(x + m, nvq, uc[u]);  // (a stored iterate is what it is, also where no cell has a row: the neighbours read it from the tile)
        }
        if (per && nvq < 4) {  // (the slot behind the row's last cell: column 0 of a periodic level, see k_mg_post4t)
          const float wv = u1(0, jj, kk);
          if (nvq == 1) uc[u][1] = wv; else if (nvq == 2) uc[u][2] = wv; else uc[u][3] = wv;
        }
      } else {
        u4(iw, jw, kk, iw == ii ? nvq : 4, uc[u]);
      }
    }
    *reinterpret_cast<float4*>(&T[lds_at(tq, ty, lz)]) = make_float4(uc[u][0], uc[u][1], uc[u][2], uc[u][3]);
  }
  {
    float h[4];
    const int hq = tid & 15, side = (tid >> 4) & 1, a = (tid >> 5) & 7;
    const bool zface = tid >= 256;
    const int ly = zface ? a : (side ? kTY : -1), lz = zface ? (side ? kTZ : -1) : a;
    const int hi = i0 + 4 * hq, jj = j0 + ly, kk = k0 + lz;
    const int iw = hi < L.nx ? hi : col_of(hi), jw = row_of(jj);
    h[0] = h[1] = h[2] = h[3] = 0.f;
    if (iw >= 0 && jw >= 0 && kk >= 0 && kk <= L.nzl + 1) u4(iw, jw, kk, nv_of(iw), h);
    *reinterpret_cast<float4*>(&T[lds_at(hq, ly, lz)]) = make_float4(h[0], h[1], h[2], h[3]);
    if (tid < 128) {
      const int xs_ = tid & 1, xy = (tid >> 1) & 7, xz = tid >> 4;
      const int xi = xs_ ? i0 + kTX : i0 - 1, xj = j0 + xy, xk = k0 + xz;
      const int xiw = col_of(xi);
      float v = 0.f;
      if (xiw >= 0 && xj < L.ny && xk <= L.nzl + 1) v = u1(xiw, xj, xk);
      T[(xz + 1) * kTPP + (xy + 1) * kTRP + (xs_ ? 4 + kTX : 3)] = v;
    }
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int lz = tz + 4 * u;
    float res[4] = {0.f, 0.f, 0.f, 0.f};
    const int i = ii, j = j0 + ty, k = k0 + lz;
    const long long m = (long long)i + sy * j + sz * k;
    const bool live = mine[u] && (dc[u][0] > 0.f || dc[u][1] > 0.f || dc[u][2] > 0.f || dc[u][3] > 0.f);
    if (live) {
      const int pos = lds_at(tq, ty, lz);
      const float xl = T[pos - 1], xr = T[pos + 4];
      const float4 ym4 = *reinterpret_cast<const float4*>(&T[pos - kTRP]), yp4 = *reinterpret_cast<const float4*>(&T[pos + kTRP]);
      const float4 zm4 = *reinterpret_cast<const float4*>(&T[pos - kTPP]), zp4 = *reinterpret_cast<const float4*>(&T[pos + kTPP]);
      const float xym[4] = {ym4.x, ym4.y, ym4.z, ym4.w}, xyp[4] = {yp4.x, yp4.y, yp4.z, yp4.w};
      const float xzm[4] = {zm4.x, zm4.y, zm4.z, zm4.w}, xzp[4] = {zp4.x, zp4.y, zp4.z, zp4.w};
      const long long mxm = (per && i == 0) ? m + (L.nx - 1) : m - 1;
      const long long mym = (per && j == 0) ? m + sy * (L.ny - 1) : m - sy;
      float wxc[4], wyc[4], wym[4], wzc[4], wzm[4];
      ldqa<AL>(L.wx + m, nvq, wxc); ldqa<AL>(L.wy + m, nvq, wyc); ldqa<AL>(L.wy + mym, nvq, wym); ldqa<AL>(L.wz + m, nvq, wzc); ldqa<AL>(L.wz + (m - sz), nvq, wzm);
      const float wxl = (float)L.wx[mxm];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (dc[u][c] > 0.f) {
          float s = (c == 0 ? wxl : wxc[c > 0 ? c - 1 : 0]) * (c == 0 ? xl : uc[u][c > 0 ? c - 1 : 0]);
          s += wxc[c] * (c == 3 ? xr : uc[u][c < 3 ? c + 1 : 3]);
          s += wym[c] * xym[c];
          s += wyc[c] * xyp[c];
          s += wzm[c] * xzm[c];
          s += wzc[c] * xzp[c];
          const float rr = bc[u][c] - (dc[u][c] * uc[u][c] - s);
          res[c] = OUT == 0 ? rr : uc[u][c] + L.omega * rr * rcp(dc[u][c]);
        }
      }
    }
    if (RESTRICT) *reinterpret_cast<float4*>(&R[((lz * kTY) + ty) * kTX + 4 * tq]) = make_float4(res[0], res[1], res[2], res[3]);
    else if (mine[u]) {
      stqa<AL>(r + m, nvq, res[0], res[1], res[2], res[3]);
      if (L.zimg) {
        if (k == 1) stqa<AL>(r + m + L.sz * L.nzl, nvq, res[0], res[1], res[2], res[3]);
        if (k == L.nzl) stqa<AL>(r + m - L.sz * L.nzl, nvq, res[0], res[1], res[2], res[3]);
      }
    }
  }
  if (RESTRICT) {
    __syncthreads();
    if (tid < 256) {
      const int cq = tid & 15, cy = (tid >> 4) & 3, cz = tid >> 6;  // coarse cells 2 cq, 2 cq + 1 of coarse row cy, coarse plane cz of the tile
      const int I = (i0 >> 1) + 2 * cq, J = (j0 >> 1) + cy, kf = k0 + 2 * cz;  // kf: the first of the two fine planes (local)
      if (I < C.nx && J < C.ny && kf <= L.nzl) {
        float o[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          float t = 0.f;
#pragma unroll
          for (int s = 0; s < 4; ++s) {  // k_mg_restrict's order: plane, row, x
            const float* q = &R[(((2 * cz + (s >> 1)) * kTY) + 2 * cy + (s & 1)) * kTX + 4 * cq + 2 * e];
            t += q[0]; t += q[1];
          }
          o[e] = t;
        }
        const int cp = ((L.z0 + kf - 1) >> 1) - C.z0 + 1;
        const long long cm = (long long)I + (long long)C.nx * J + C.sz * cp;
        const bool two_cells = I + 1 < C.nx;  // (nx % 4 == 2: the row's last quad holds one coarse cell)
        bcoarse[cm] = o[0];
        if (two_cells) bcoarse[cm + 1] = o[1];
        if (C.zimg) {
          if (cp == 1) { bcoarse[cm + C.sz * C.nzl] = o[0]; if (two_cells) bcoarse[cm + 1 + C.sz * C.nzl] = o[1]; }
          if (cp == C.nzl) { bcoarse[cm - C.sz * C.nzl] = o[0]; if (two_cells) bcoarse[cm + 1 - C.sz * C.nzl] = o[1]; }
        }
      }
    }
  }
}

// ---- round 4: pre-smoothing residual + restriction on BLOCKS of 4 x 2 x 2 cells per thread (open problems on one rank, even extents) ----
// A thread that owns rows 2J, 2J+1 of planes 2K, 2K+1 holds all eight children of its two coarse cells: the pre-smoothing residual is
// restricted in registers -- no residual array, no restriction launch (0.22 ms of a 13.5 ms iteration at 1024^3; the stencil part itself
// takes what k_mg_lin4 took: fewer loads per cell, a quarter of the threads).  Same arithmetic in the same order as k_mg_lin4<0, 0> +
// k_mg_restrict (bit-identical; option mg_blocked = 0 keeps the pair as the cross-check).  The same blocking of the prolongation +
// post-smoother was built and measured 10 % SLOWER than k_mg_post4 (132 registers, profiles/r04/mg_blocked_coarse_kernels.txt): not kept.
#define OI_MG_BLOCK8(L)                                                                         \
  int t_ = blockIdx.x;                                                                          \
  if ((gridDim.x & 7) == 0) t_ = (t_ & 7) * (gridDim.x >> 3) + (t_ >> 3);                       \
  const int gpr_ = L.nx >> 2, rows_ = L.ny >> 1;                                                \
  const long long id_ = (long long)t_ * blockDim.x + threadIdx.x;                               \
  const long long row_ = id_ / gpr_;                                                            \
  const int i = (int)(id_ - row_ * gpr_) << 2, J = (int)(row_ % rows_), Kp = (int)(row_ / rows_); \
  if (Kp >= (L.nzl >> 1)) return;                                                               \
  const int j = 2 * J, k = 1 + 2 * Kp;                                                          \
  const long long sy = L.nx, sz = L.sz;                                                         \
  const long long m00 = (long long)i + sy * j + sz * k;

// bc = P^T (b - A (w b / D)) on the coarse level C; L: open, nx % 4 == 0, ny and nzl even, z0 even
template <class W, class D>
__global__ void __launch_bounds__(256) k_mg_lin4r(LevelT<W, D> L, const float* __restrict__ b, Dims C, float* __restrict__ bc) {
  OI_MG_BLOCK8(L)
  auto xs1 = [&](float bb, float dd) -> float { return dd > 0.f ? L.omega * bb * rcp(dd) : 0.f; };
  auto u4 = [&](long long mm, float (&o)[4]) {
    float bb[4], dd[4];
    ld4(b + mm, bb); ld4(L.diag + mm, dd);
#pragma unroll
    for (int c = 0; c < 4; ++c) o[c] = xs1(bb[c], dd[c]);
  };
  auto u1 = [&](long long mm) -> float { return xs1(b[mm], (float)L.diag[mm]); };
  // own cells: segment s = 2 * c + bb  (plane k + c, row j + bb), the order the restriction sums them in
  float dc[4][4], bcv[4][4], u[4][4];
  bool any = false;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const long long m = m00 + sy * (s & 1) + sz * (s >> 1);
    ld4(L.diag + m, dc[s]);
    any = any || dc[s][0] > 0.f || dc[s][1] > 0.f || dc[s][2] > 0.f || dc[s][3] > 0.f;
  }
  const long long cm = (long long)(i >> 1) + (long long)C.nx * J + C.sz * (((L.z0 + k - 1) >> 1) - C.z0 + 1);
  if (!any) {  // sixteen cells without an operator row: both block sums are sums of zeros
    *reinterpret_cast<float2*>(bc + cm) = make_float2(0.f, 0.f);
    return;
  }
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const long long m = m00 + sy * (s & 1) + sz * (s >> 1);
    ld4(b + m, bcv[s]);
#pragma unroll
    for (int c = 0; c < 4; ++c) u[s][c] = xs1(bcv[s][c], dc[s][c]);
  }
  float res[4][4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int bb = s & 1, cz = s >> 1;
    const long long m = m00 + sy * bb + sz * cz;
    // a row segment whose four cells have no operator row: zeros, none of its neighbour loads (k_mg_lin4's early return)
    if (!(dc[s][0] > 0.f || dc[s][1] > 0.f || dc[s][2] > 0.f || dc[s][3] > 0.f)) {
      res[s][0] = res[s][1] = res[s][2] = res[s][3] = 0.f;
      continue;
    }
    float xym[4], xyp[4], xzm[4], xzp[4], wxc[4], wyc[4], wym[4], wzc[4], wzm[4];
    const float xl = u1(m - 1), xr = u1(m + 4);
    // y and z neighbours: the other segment of the block where there is one, a load where the block ends
    if (bb == 0) { u4(m - sy, xym); _Pragma("unroll") for (int c = 0; c < 4; ++c) xyp[c] = u[(s + 1) & 3][c]; }
    else { _Pragma("unroll") for (int c = 0; c < 4; ++c) xym[c] = u[(s - 1) & 3][c]; u4(m + sy, xyp); }
    if (cz == 0) { u4(m - sz, xzm); _Pragma("unroll") for (int c = 0; c < 4; ++c) xzp[c] = u[(s + 2) & 3][c]; }
    else { _Pragma("unroll") for (int c = 0; c < 4; ++c) xzm[c] = u[(s - 2) & 3][c]; u4(m + sz, xzp); }
    ld4(L.wx + m, wxc); ld4(L.wy + m, wyc); ld4(L.wy + (m - sy), wym); ld4(L.wz + m, wzc); ld4(L.wz + (m - sz), wzm);
    const float wxl = (float)L.wx[m - 1];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      res[s][c] = 0.f;
      if (dc[s][c] > 0.f) {
        float t = (c == 0 ? wxl : wxc[c > 0 ? c - 1 : 0]) * (c == 0 ? xl : u[s][c > 0 ? c - 1 : 0]);
        t += wxc[c] * (c == 3 ? xr : u[s][c < 3 ? c + 1 : 3]);
        t += wym[c] * xym[c];
        t += wyc[c] * xyp[c];
        t += wzm[c] * xzm[c];
        t += wzc[c] * xzp[c];
        res[s][c] = bcv[s][c] - (dc[s][c] * u[s][c] - t);
      }
    }
  }
  // P^T: the eight children of a coarse cell in k_mg_restrict's order -- plane, row, x
  float o[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    float t = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) { t += res[s][2 * e]; t += res[s][2 * e + 1]; }
    o[e] = t;
  }
  *reinterpret_cast<float2*>(bc + cm) = make_float2(o[0], o[1]);
}

// coarsest level: x_out = x_in + w (b - A x_in) / D   (first = 1: x_in = 0)
template <class W, class D>
__global__ void __launch_bounds__(256) k_mg_jacobi(LevelT<W, D> L, int kfirst, const float* __restrict__ b,
                                                    const float* __restrict__ xin, float* __restrict__ xout, int first) {
  OI_MG_CELL(L)
  const float d = (float)L.diag[m];
  float o = 0.f;
  if (d > 0.f) {
    if (first) {
      o = L.omega * b[m] * rcp(d);
    } else {
      const float c = xin[m];
      OI_MG_NBRS(L)
      o = c + L.omega * (b[m] - (d * c - offdiag(L, m, mxm, mym, xin[mxm], xin[mxp], xin[mym], xin[myp], xin[m - sz], xin[m + sz]))) * rcp(d);
    }
  }
  xout[m] = o;
  OI_MG_IMG(L, xout, m, k, float, o);
}

// coarsest level, all sweeps in ONE launch: <= 512 cells, one thread per cell, weights and right-hand side in registers, the
// iterate in two LDS planes, a barrier between sweeps (the per-sweep launches of k_mg_jacobi -- 16 sweeps x 8 visits per cycle --
// were 0.9 ms of every cycle at 1024^3, all of it launch latency).  Same arithmetic in the same order as k_mg_jacobi: bit-identical.
// Levels that live whole on this rank only.  Round 4: also the PERIODIC level of the homogenization cell problems -- x, y and z
// neighbours wrap by index inside the LDS iterate (what the ghost planes of the per-sweep launches held: the images of the first /
// last plane), which also saves the two plane copies after every sweep: 128 launches + 256 copies per cycle, 1.5 ms of a 19.7 ms
// iteration at 1024^3.
constexpr int kCoarsestLds = 512 + 2 * 512;
template <class W, class D>
__global__ void __launch_bounds__(512) k_mg_coarsest(LevelT<W, D> L, const float* __restrict__ b, float* __restrict__ out, int sweeps) {
  __shared__ float xb[2][kCoarsestLds];
  const int n = (int)(L.sz * (L.nzl + 2)), cells = (int)(L.sz * L.nzl);
  for (int t = threadIdx.x; t < n; t += blockDim.x) { xb[0][t] = 0.f; xb[1][t] = 0.f; }
  const int t = threadIdx.x;
  const bool act = t < cells;
  const int sz = (int)L.sz, sy = L.nx;
  const int m = sz + t;
  int mxm = m - 1, mxp = m + 1, mym = m - sy, myp = m + sy, mzm = m - sz, mzp = m + sz;
  if (L.periodic && act) {  // (OI_MG_NBRS; z: the image planes 0 and nzl + 1 ARE planes nzl and 1)
    const int i = t % L.nx, j = (t / L.nx) % L.ny, kk = t / sz;
    if (i == 0) mxm = m + (L.nx - 1);
    if (i == L.nx - 1) mxp = m - (L.nx - 1);
    if (j == 0) mym = m + sy * (L.ny - 1);
    if (j == L.ny - 1) myp = m - sy * (L.ny - 1);
    if (kk == 0) mzm = m + sz * (L.nzl - 1);
    if (kk == L.nzl - 1) mzp = m - sz * (L.nzl - 1);
  }
  float d = 0.f, bb = 0.f, wxl = 0.f, wxc = 0.f, wyl = 0.f, wyc = 0.f, wzl = 0.f, wzc = 0.f;
  if (act) {
    d = (float)L.diag[m]; bb = b[m];
    // (the weight of the -z link is read where k_mg_jacobi reads it: from the plane below, the ghost plane for the first one)
    wxl = (float)L.wx[mxm]; wxc = (float)L.wx[m]; wyl = (float)L.wy[mym]; wyc = (float)L.wy[m]; wzl = (float)L.wz[m - sz]; wzc = (float)L.wz[m];
  }
  __syncthreads();
  float c = (act && d > 0.f) ? L.omega * bb * rcp(d) : 0.f;  // sweep 1, from zero
  if (act) xb[0][m] = c;
  __syncthreads();
  for (int it = 2; it <= sweeps; ++it) {
    const float* xi = xb[it & 1];      // written by sweep it - 1
    float* xo = xb[(it & 1) ^ 1];
    float o = 0.f;
    if (act && d > 0.f) {
      float s_ = wxl * xi[mxm];
      s_ += wxc * xi[mxp];
      s_ += wyl * xi[mym];
      s_ += wyc * xi[myp];
      s_ += wzl * xi[mzm];
      s_ += wzc * xi[mzp];
      o = c + L.omega * (bb - (d * c - s_)) * rcp(d);
    }
    if (act) xo[m] = o;
    c = o;
    __syncthreads();
  }
  if (act) {
    out[m] = c;
    const int kk1 = t / sz + 1;  // local plane of this cell
    OI_MG_IMG(L, out, m, kk1, float, c);
  }
}

// ------------------------------ plain fine-level kernels (cross-check of the tiled ones) ------------------------------
// fine vectors live in the pitched / ghost-plane layout; b vanishes on non-free cells and its
// ghost planes hold the neighbours' values.   r1 = b - A (w b / code)
__global__ void __launch_bounds__(256) k_mg_fine_pre(Geom g, const uint8_t* __restrict__ code, const double* __restrict__ b,
                                                      double* __restrict__ r1, const Ctrl* __restrict__ ctrl, int periodic, double omega_) {
  if (ctrl && ctrl->done) return;
  const long long n = (long long)g.nzl * g.ny * g.nx;
  const double w = omega_;
  auto xs = [&](long long e) -> double { const unsigned c = code[e]; return c ? w * b[e] / (double)c : 0.0; };
  for (long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (long long)gridDim.x * blockDim.x) {
    const int i = (int)(v % g.nx), j = (int)((v / g.nx) % g.ny), gz = 1 + (int)(v / ((long long)g.nx * g.ny));
    const long long e = (long long)gz * g.plane + (long long)j * g.pitch + i;
    const unsigned c = code[e];
    double res = 0.0;
    if (c) {
      double s;
      if (periodic) {  // x, y wrap by index; the ghost planes hold the periodic images in z
        const long long exm = i > 0 ? e - 1 : e + (g.nx - 1), exp_ = i < g.nx - 1 ? e + 1 : e - (g.nx - 1);
        const long long eym = j > 0 ? e - g.pitch : e + (long long)g.pitch * (g.ny - 1);
        const long long eyp = j < g.ny - 1 ? e + g.pitch : e - (long long)g.pitch * (g.ny - 1);
        s = (((xs(exm) + xs(exp_)) + xs(eym)) + xs(eyp)) + xs(e - g.plane) + xs(e + g.plane);
      } else {
        s = ((((i > 0 ? xs(e - 1) : 0.0) + (i < g.nx - 1 ? xs(e + 1) : 0.0)) + (j > 0 ? xs(e - g.pitch) : 0.0)) +
             (j < g.ny - 1 ? xs(e + g.pitch) : 0.0)) + xs(e - g.plane) + xs(e + g.plane);
      }
      res = b[e] - ((double)c * xs(e) - s);
    }
    r1[e] = res;
  }
}

// z = x2 + w (b - A x2) / code,  x2 = w b / code + kScale * P (xc + xc2)
__global__ void __launch_bounds__(256) k_mg_fine_post(Geom g, const uint8_t* __restrict__ code, const double* __restrict__ b,
                                                       MgCoarse C, double sc, double* __restrict__ z, const Ctrl* __restrict__ ctrl, double omega_) {
  if (ctrl && ctrl->done) return;
  const long long n = (long long)g.nzl * g.ny * g.nx;
  const double w = omega_;
  auto x2 = [&](int i, int j, int gz) -> double {  // gz: plane incl. ghost offset
    const long long e = (long long)gz * g.plane + (long long)j * g.pitch + i;
    const unsigned c = code[e];
    if (!c) return 0.0;
    const int kg = g.z0 + gz - 1;
    int cp = (kg >> 1) - C.z0c + 1;
    if (C.periodic) cp = kg < 0 ? 0 : (kg >= g.nz ? C.cnzl + 1 : cp);  // images beyond the ends: parents in the coarse ghosts
    const long long cm = (long long)(i >> 1) + (long long)C.cnx * (j >> 1) + C.csz * cp;
    float cv = C.xc[cm];
    if (C.xc2) cv += C.xc2[cm];
    return w * b[e] / (double)c + sc * (double)cv;
  };
  for (long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (long long)gridDim.x * blockDim.x) {
    const int i = (int)(v % g.nx), j = (int)((v / g.nx) % g.ny), gz = 1 + (int)(v / ((long long)g.nx * g.ny));
    const long long e = (long long)gz * g.plane + (long long)j * g.pitch + i;
    const unsigned c = code[e];
    double out = 0.0;
    if (c) {
      const double xc0 = x2(i, j, gz);
      double s;
      if (C.periodic)
        s = (((x2(i > 0 ? i - 1 : g.nx - 1, j, gz) + x2(i < g.nx - 1 ? i + 1 : 0, j, gz)) + x2(i, j > 0 ? j - 1 : g.ny - 1, gz)) +
             x2(i, j < g.ny - 1 ? j + 1 : 0, gz)) + x2(i, j, gz - 1) + x2(i, j, gz + 1);
      else
        s = ((((i > 0 ? x2(i - 1, j, gz) : 0.0) + (i < g.nx - 1 ? x2(i + 1, j, gz) : 0.0)) +
              (j > 0 ? x2(i, j - 1, gz) : 0.0)) + (j < g.ny - 1 ? x2(i, j + 1, gz) : 0.0)) +
            x2(i, j, gz - 1) + x2(i, j, gz + 1);
      out = xc0 + w * (b[e] - ((double)c * xc0 - s)) / (double)c;
    }
    z[e] = out;
  }
}

// partial r.z over the owned planes
__global__ void __launch_bounds__(256) k_mg_dot(Geom g, const double* __restrict__ a, const double* __restrict__ b,
                                                 double* __restrict__ partials) {
  const long long npairs = (long long)g.nzl * g.plane / 2;
  double s = 0.0;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < npairs; t += (long long)gridDim.x * blockDim.x) {
    const long long e = g.plane + 2 * t;
    const double2 av = *reinterpret_cast<const double2*>(a + e), bv = *reinterpret_cast<const double2*>(b + e);
    s += av.x * bv.x; s += av.y * bv.y;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  __shared__ double red[4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) red[w] = s;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

inline int nblk(long long items, int cap = 8192) {
  long long b = (items + 255) / 256;
  return (int)std::max<long long>(1, std::min<long long>(b, cap));
}

}  // namespace

struct MgLevelHost {
  Dims d{};
  int nz = 0;               // global plane count of the level
  bool replicated = false;  // every rank holds (and solves) the whole level
  bool compact = false;     // byte weights / 16-bit diagonal
  long long n = 0;          // cells incl. the two ghost planes
  void *wx = nullptr, *wy = nullptr, *wz = nullptr, *diag = nullptr;
  float *b = nullptr, *b2 = nullptr, *r = nullptr, *x2 = nullptr, *xs = nullptr;
  float *t = nullptr, *t2 = nullptr;  // two smoothing sweeps (MgHierarchy::nu_coarse = 2): the pre-smoothed iterate, a scratch
  template <class W, class D>
  LevelT<W, D> dev() const {
    LevelT<W, D> L;
    static_cast<Dims&>(L) = d;
    L.wx = (const W*)wx; L.wy = (const W*)wy; L.wz = (const W*)wz; L.diag = (const D*)diag;
    return L;
  }
  int bx() const { return d.nx >= 256 ? 256 : (d.nx >= 128 ? 128 : 64); }
  int blocks(int nplanes) const { return ((d.nx + bx() - 1) / bx()) * d.ny * nplanes; }
};

struct MgHierarchy {
  std::vector<MgLevelHost> levels;  // levels[0] = first COARSE level
  oi_comm* comm = nullptr;          // neighbours of the slab partition (null: whole volume)
  bool periodic = false;            // homogenization cell problem: x, y, z wrap (z-slabs then form a ring)
  size_t bytes = 0;
  unsigned wmask = kWCycleMask;
  float scale = kScale;
  float omega = kOmega;
  bool novec4 = false;  // cross-check: one-cell coarse kernels instead of the four-cells-per-thread ones
  bool blocked = true;  // round 4: pre-smoothing residual + restriction as k_mg_lin4r (4 x 2 x 2 cells per thread) where a level allows it; false: cross-check
  bool coarsest_plain = false;  // cross-check: one launch per sweep on the coarsest level instead of k_mg_coarsest
  bool fused_images = true;     // periodic levels on one rank: no image-plane copy after a kernel that wrote the images itself; false: cross-check
  int coarse_tiled = 1;         // levels of >= 2^21 cells: correction + post-smoothing through an LDS tile (k_mg_post4t); 0: k_mg_post4 (cross-check); 2: every four-cell level (tests)
  int nu_coarse = 1;    // damped-Jacobi sweeps before and after the coarse correction on the COARSE levels (1 or 2)
  bool failed = false;
};

void mg_destroy(MgHierarchy* mg) {
  if (!mg) return;
  for (auto& L : mg->levels) {
    void* p[] = {L.wx, L.wy, L.wz, L.diag, L.b, L.b2, L.r, L.x2, L.xs, L.t, L.t2};
    for (void* q : p)
      if (q) (void)oi::dev_free(q);
  }
  delete mg;
}

// typed launch of a level kernel: KERN<uint8_t, uint16_t> on the compact levels, <float, float> below
#define OI_MG_LAUNCH(KERN, LV, NPLANES, S, ...)                                                                      \
  do {                                                                                                               \
    if ((LV).blocks(NPLANES) > 0) {                                                                                  \
      if ((LV).compact)                                                                                              \
        hipLaunchKernelGGL((KERN<uint8_t, uint16_t>), dim3((LV).blocks(NPLANES)), dim3((LV).bx()), 0, S,             \
                           (LV).dev<uint8_t, uint16_t>(), __VA_ARGS__);                                              \
      else                                                                                                           \
        hipLaunchKernelGGL((KERN<float, float>), dim3((LV).blocks(NPLANES)), dim3((LV).bx()), 0, S,                   \
                           (LV).dev<float, float>(), __VA_ARGS__);                                                   \
    }                                                                                                                \
  } while (0)

// periodic images of a level that lives whole on this rank, both in ONE launch: ghost plane 0 <- plane nzl, ghost plane nzl + 1 <-
// plane 1 (round 4: two hipMemcpyAsync per exchange were ~150 blit launches per cycle of the homogenization solve)
__global__ void __launch_bounds__(256) k_wrap_planes(uint32_t* __restrict__ p, long long plane_words, int nzl) {
  const long long n = 2 * plane_words;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x) {
    const bool hi = t >= plane_words;
    const long long w = hi ? t - plane_words : t;
    if (hi) p[plane_words * (nzl + 1) + w] = p[plane_words + w];
    else p[w] = p[plane_words * nzl + w];
  }
}

// ghost planes of a per-level array from the z-neighbours (no-op without neighbours / on replicated levels)
// imaged: the kernel that produced the array wrote the images itself (OI_MG_IMG on a level with Dims::zimg)
static bool ghost_exchange(MgHierarchy* mg, const MgLevelHost& L, void* base, size_t elem, hipStream_t s, bool imaged = false) {
  char* p = (char*)base;
  const size_t pb = (size_t)L.d.sz * elem;
  if (!mg->comm || L.replicated) {
    if (!mg->periodic) return true;
    if (imaged && L.d.zimg && mg->fused_images) return true;
    // the whole level lives here: its first / last plane are each other's periodic images
    if ((pb & 3) == 0 && L.d.nzl >= 2) {  // (one plane only: source and destination of the two copies are the same pair, keep the ordered copies)
      const long long words = (long long)(pb >> 2);
      hipLaunchKernelGGL(k_wrap_planes, dim3(nblk(2 * words, 1024)), dim3(256), 0, s, reinterpret_cast<uint32_t*>(p), words, L.d.nzl);
      return true;
    }
    if (hipMemcpyAsync(p, p + pb * (size_t)L.d.nzl, pb, hipMemcpyDeviceToDevice, s) != hipSuccess ||
        hipMemcpyAsync(p + pb * (size_t)(L.d.nzl + 1), p + pb, pb, hipMemcpyDeviceToDevice, s) != hipSuccess) {
      mg->failed = true;
      return false;
    }
    return true;
  }
  const oi_status st = mg->periodic
                           ? mg->comm->exchange_ring(p + pb, p, p + pb * (size_t)L.d.nzl, p + pb * (size_t)(L.d.nzl + 1), pb, s)
                           : mg->comm->exchange(p + pb, p, p + pb * (size_t)L.d.nzl, p + pb * (size_t)(L.d.nzl + 1), pb, s);
  if (st != OI_OK) {
    mg->failed = true;
    return false;
  }
  return true;
}

// coarse planes (local indices) that receive contributions from the owned planes [z0, z0 + nzl) of the parent
static void child_span(const MgLevelHost& C, int pz0, int pnzl, int* kfirst, int* nplanes) {
  const int ka = pz0 >> 1, kb = (pz0 + pnzl - 1) >> 1;  // global
  *kfirst = ka - C.d.z0 + 1;
  *nplanes = kb - ka + 1;
}

// after the owned contributions of a right-hand side / operator array are in place: sum across ranks
// (first replicated level) or fetch the ghost planes (slab level)
static bool finish_child_array(MgHierarchy* mg, const MgLevelHost& C, bool parent_replicated, float* a, hipStream_t s, bool imaged = false) {
  if (mg->comm && !parent_replicated && C.replicated) {
    if (mg->comm->allreduce_sum_f32(a + C.d.sz, (size_t)C.d.sz * C.d.nzl, s) != OI_OK) {
      mg->failed = true;
      return false;
    }
    imaged = false;  // (the sum over the ranks came after the producer's images)
  }
  return ghost_exchange(mg, C, a, sizeof(float), s, imaged);  // neighbours' planes, or the periodic images, or nothing to do
}

#define MGS(call)                                                            \
  do {                                                                       \
    hipError_t e_ = (call);                                                  \
    if (e_ != hipSuccess) {                                                  \
      oi::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
      (void)hipGetLastError();                                               \
      return OI_HIP_ERROR;                                                   \
    }                                                                        \
  } while (0)

MgHierarchy* mg_setup(const Geom& g, const uint8_t* code, oi_comm* comm, int periodic, hipStream_t s, size_t* bytes_out) {
  MgHierarchy* mg = new MgHierarchy();
  mg->comm = (comm && comm->size > 1) ? comm : nullptr;
  mg->periodic = periodic != 0;
  long long replicate_cells = kReplicateCells;
  if (g.tune.mg_replicate_cells > 0) replicate_cells = g.tune.mg_replicate_cells;  // tests: distribute small levels too
  if (g.tune.mg_wmask >= 0) mg->wmask = (unsigned)g.tune.mg_wmask;                   // experiments with the cycle shape
  if (g.tune.mg_scale > 0.f) mg->scale = g.tune.mg_scale;
  if (g.tune.mg_omega > 0.f && g.tune.mg_omega <= 1.f) mg->omega = g.tune.mg_omega;
  mg->novec4 = g.tune.mg_novec4 != 0;
  mg->blocked = g.tune.mg_blocked != 0;
  mg->nu_coarse = g.tune.mg_nu_coarse >= 2 ? 2 : 1;
  mg->coarsest_plain = g.tune.mg_coarsest_plain != 0;
  mg->coarse_tiled = g.tune.mg_coarse_tiled;
  mg->fused_images = g.tune.mg_fused_images != 0;
  // ---- 1. plan ----
  {
    int pnx = g.nx, pny = g.ny, pnz = g.nz, pz0 = g.z0, pnzl = g.nzl;
    bool prep = false;  // parent replicated?
    for (int l = 0; l < 24; ++l) {
      MgLevelHost L;
      L.d.nx = (pnx + 1) / 2; L.d.ny = (pny + 1) / 2; L.nz = (pnz + 1) / 2;
      L.d.sz = (long long)L.d.nx * L.d.ny;
      L.d.nzg = L.nz;
      L.d.periodic = periodic ? 1 : 0;
      L.d.omega = (g.tune.mg_omega_coarse > 0.f && g.tune.mg_omega_coarse <= 1.f) ? g.tune.mg_omega_coarse : mg->omega;
      L.compact = l < kCompactLevels;
      // keep the slab partition while every rank's slab starts on an even plane, holds at least two
      // planes and the level is worth distributing; otherwise this and all deeper levels are replicated
      L.replicated = prep;
      if (mg->comm && !prep) {
        int viol = ((pz0 & 1) || pnzl < 2 || L.d.sz * L.nz <= replicate_cells) ? 1 : 0;
        if (mg->comm->allreduce_max_i32_host(&viol, 1, s) != OI_OK) { mg_destroy(mg); return nullptr; }
        L.replicated = viol != 0;
      }
      if (mg->comm && !L.replicated) {
        L.d.z0 = pz0 >> 1;
        L.d.nzl = ((pz0 + pnzl + 1) >> 1) - L.d.z0;
      } else {
        L.d.z0 = 0;
        L.d.nzl = L.nz;
      }
      L.d.zimg = (periodic && (!mg->comm || L.replicated)) ? 1 : 0;  // the cycle's kernels write the z images of what they store
      L.n = L.d.sz * (L.d.nzl + 2);
      mg->levels.push_back(L);
      pnx = L.d.nx; pny = L.d.ny; pnz = L.nz; pz0 = L.d.z0; pnzl = L.d.nzl; prep = L.replicated;
      const long long cells = L.d.sz * L.nz;
      if (cells <= kCoarsestCells || (L.d.nx == 1 && L.d.ny == 1 && L.nz == 1)) break;
    }
  }
  // ---- 2. allocate (rank-local), then agree ----
  float* t[4] = {nullptr, nullptr, nullptr, nullptr};  // float scratch of the compact levels' operators (sized for the largest)
  auto free_scratch = [&] { for (auto& p : t) { if (p) (void)oi::dev_free(p); p = nullptr; } };
  auto allocate = [&]() -> oi_status {
    long long tn = 0;
    for (MgLevelHost& L : mg->levels) {
      const size_t wb = L.compact ? 1 : 4, db = L.compact ? 2 : 4;
      MGS(oi::dev_malloc(&L.wx, wb * L.n)); MGS(oi::dev_malloc(&L.wy, wb * L.n)); MGS(oi::dev_malloc(&L.wz, wb * L.n));
      MGS(oi::dev_malloc(&L.diag, db * L.n));
      float** vecs[] = {&L.b, &L.b2, &L.r, &L.x2, &L.xs};
      for (float** v : vecs) {
        MGS(oi::dev_malloc((void**)v, sizeof(float) * L.n));
        MGS(hipMemsetAsync(*v, 0, sizeof(float) * L.n, s));
      }
      mg->bytes += (size_t)L.n * (3 * wb + db + 5 * sizeof(float));
      if (mg->nu_coarse > 1) {
        float** more[] = {&L.t, &L.t2};
        for (float** v : more) {
          MGS(oi::dev_malloc((void**)v, sizeof(float) * L.n));
          MGS(hipMemsetAsync(*v, 0, sizeof(float) * L.n, s));
        }
        mg->bytes += (size_t)L.n * 2 * sizeof(float);
      }
      if (L.compact) tn = std::max(tn, L.n);
    }
    if (tn > 0)
      for (auto& p : t) MGS(oi::dev_malloc((void**)&p, sizeof(float) * tn));
    return OI_OK;
  };
  if (oi::agree_status(mg->comm, allocate(), s, "the allocation of the multigrid hierarchy") != OI_OK) {
    free_scratch();
    mg_destroy(mg);
    return nullptr;
  }
  // ---- 3. build ----
  bool ok = true;
  for (size_t l = 0; l < mg->levels.size() && ok; ++l) {
    MgLevelHost& L = mg->levels[l];
    const bool prep = l > 0 && mg->levels[l - 1].replicated;
    const int pz0 = l > 0 ? mg->levels[l - 1].d.z0 : g.z0, pnzl = l > 0 ? mg->levels[l - 1].d.nzl : g.nzl;
    // operator: owned contributions into float arrays, summed / exchanged, then (compact levels) narrowed
    float* a[4] = {t[0], t[1], t[2], t[3]};
    if (!L.compact) { a[0] = (float*)L.wx; a[1] = (float*)L.wy; a[2] = (float*)L.wz; a[3] = (float*)L.diag; }
    for (auto p : a) ok = ok && hipMemsetAsync(p, 0, sizeof(float) * L.n, s) == hipSuccess;
    if (!ok) break;
    int kfirst = 1, nplanes = L.d.nzl;
    if (L.replicated && !prep && mg->comm) child_span(L, pz0, pnzl, &kfirst, &nplanes);
    if (L.blocks(nplanes) > 0) {
      if (l == 0) {
        hipLaunchKernelGGL(k_mg_build1, dim3(L.blocks(nplanes)), dim3(L.bx()), 0, s, g, code, L.d, kfirst, a[0], a[1], a[2], a[3]);
      } else {
        const MgLevelHost& P = mg->levels[l - 1];
        if (P.compact)
          hipLaunchKernelGGL((k_mg_build<uint8_t, uint16_t>), dim3(L.blocks(nplanes)), dim3(L.bx()), 0, s,
                             P.dev<uint8_t, uint16_t>(), L.d, kfirst, a[0], a[1], a[2], a[3]);
        else
          hipLaunchKernelGGL((k_mg_build<float, float>), dim3(L.blocks(nplanes)), dim3(L.bx()), 0, s, P.dev<float, float>(),
                             L.d, kfirst, a[0], a[1], a[2], a[3]);
      }
    }
    if (L.replicated) {
      for (auto p : a) ok = ok && finish_child_array(mg, L, prep, p, s);
    } else {  // the cycle reads the diagonal and the +z weights of the ghost planes
      ok = finish_child_array(mg, L, prep, a[2], s) && finish_child_array(mg, L, prep, a[3], s);
    }
    if (L.compact && ok)
      hipLaunchKernelGGL(k_mg_compact, dim3(nblk(L.n)), dim3(256), 0, s, L.n, a[0], a[1], a[2], a[3], (uint8_t*)L.wx,
                         (uint8_t*)L.wy, (uint8_t*)L.wz, (uint16_t*)L.diag);
  }
  if (hipStreamSynchronize(s) != hipSuccess) {
    if (ok) oi::set_error("multigrid set-up: HIP failure: %s", hipGetErrorString(hipGetLastError()));
    ok = false;
  }
  free_scratch();
  if (!ok) { mg_destroy(mg); return nullptr; }
  if (bytes_out) *bytes_out = mg->bytes;
  return mg;
}

// right-hand side of child C from the residual of its parent (fine parent: r1 in the solver layout)
static bool restrict_to(MgHierarchy* mg, const Geom* g, const uint8_t* code, const double* r1, const MgLevelHost* P,
                        MgLevelHost& C, float* bc, hipStream_t s) {
  const bool prep = P ? P->replicated : false;
  const int pz0 = P ? P->d.z0 : g->z0, pnzl = P ? P->d.nzl : g->nzl;
  int kfirst = 1, nplanes = C.d.nzl;
  if (C.replicated && !prep && mg->comm) {
    (void)hipMemsetAsync(bc, 0, sizeof(float) * C.n, s);
    child_span(C, pz0, pnzl, &kfirst, &nplanes);
  }
  if (C.blocks(nplanes) > 0) {
    if (P) hipLaunchKernelGGL(k_mg_restrict, dim3(C.blocks(nplanes)), dim3(C.bx()), 0, s, P->d, P->r, C.d, kfirst, bc);
    else hipLaunchKernelGGL(k_mg_fine_restrict, dim3(C.blocks(nplanes)), dim3(C.bx()), 0, s, *g, code, r1, C.d, kfirst, bc);
  }
  return finish_child_array(mg, C, prep, bc, s, P != nullptr);  // (k_mg_restrict writes the images of its level; the fine-level kernels do not)
}

// Approximate solve of level l: right-hand side `rhs` (ghost planes valid), result in `out` (ghost
// planes valid on return).  Levels in kWCycleMask are visited twice per parent visit (the
// second visit solves for the residual of the first), the others once.
static bool coarse_cycle(MgHierarchy* mg, int l, const float* rhs, float* out, hipStream_t s) {
  auto& Ls = mg->levels;
  MgLevelHost& L = Ls[l];
  const int kfirst = 1;
  if (l == (int)Ls.size() - 1) {  // coarsest: a fixed number of damped-Jacobi sweeps, the last one lands in `out`
    // (periodic levels: on one rank only -- a ring of slabs exchanges the planes of a replicated level through the transport)
    // (a hierarchy of three levels or fewer -- volumes up to 64^3, REV samples -- ends on a byte / 16-bit level: the same kernel on its types)
    if ((mg->periodic ? !mg->comm : (!mg->comm || L.replicated)) && !mg->coarsest_plain && L.d.sz * L.d.nzl <= 512 && L.d.sz <= 512) {
      if (L.compact) hipLaunchKernelGGL((k_mg_coarsest<uint8_t, uint16_t>), dim3(1), dim3(512), 0, s, L.dev<uint8_t, uint16_t>(), rhs, out, kCoarsestSweeps);
      else hipLaunchKernelGGL((k_mg_coarsest<float, float>), dim3(1), dim3(512), 0, s, L.dev<float, float>(), rhs, out, kCoarsestSweeps);
      return mg->periodic ? ghost_exchange(mg, L, out, sizeof(float), s, true) : true;  // (the parent reads the image planes of the correction)
    }
    for (int it = 1; it <= kCoarsestSweeps; ++it) {
      const bool to_out = ((kCoarsestSweeps - it) & 1) == 0;
      float* dst = to_out ? out : L.r;
      const float* src = to_out ? L.r : out;
      OI_MG_LAUNCH(k_mg_jacobi, L, L.d.nzl, s, kfirst, rhs, src, dst, it == 1 ? 1 : 0);
      if (!ghost_exchange(mg, L, dst, sizeof(float), s, true)) return false;
    }
    return true;
  }
  MgLevelHost& C = Ls[l + 1];
  // byte / 16-bit levels whose rows split into groups of four cells take the four-cells-per-thread kernels
  const bool vec4 = L.compact && (L.d.nx & 3) == 0 && L.d.nx >= 64 && !mg->novec4;
  // (the LDS-tiled kernels take any nx: under-aligned quads, the last one of a row cell by cell)
  const bool quads = L.compact && L.d.nx >= 8 && !mg->novec4 && L.d.nzl >= 1;  // (a narrow level fills part of the tile's 16 quads per row: still far
                                                                                 // fewer loads than the one-cell kernels, whose blocks are row segments of 256)
  // from how many cells on: where the four-cell kernels run (nx % 4 == 0) they hold up to 2^21 cells; against the one-cell kernels
  // the tile pays from 2^17 cells on (a 500^3 volume's second coarse level, 125^3: profiles/r04/size_scan.txt)
  const bool al4 = (L.d.nx & 3) == 0;  // whole, aligned quads: the tiled kernels' aligned instantiations
  const long long tile_from = mg->coarse_tiled > 1 ? 0 : (vec4 ? (1ll << 21) : (1ll << 17));
  const int blocks4 = (int)((((long long)(L.d.nx >> 2) * L.d.ny * L.d.nzl) + 255) / 256);
  const bool two = mg->nu_coarse > 1;  // two sweeps before and two after the correction
  // blocks of 4 x 2 x 2 cells per thread, the restriction in registers: open problem, the whole level and its child on this rank,
  // even extents (the blocks are the aggregates)
  const bool blk = vec4 && !two && mg->blocked && !mg->comm && !mg->periodic && (L.d.ny & 1) == 0 && (L.d.nzl & 1) == 0 && (L.d.z0 & 1) == 0 &&
                   !C.replicated && C.d.nx * 2 == L.d.nx && C.d.ny * 2 == L.d.ny && C.d.nzl * 2 == L.d.nzl && C.d.z0 * 2 == L.d.z0;
  const int blocks8 = (int)((((long long)(L.d.nx >> 2) * (L.d.ny >> 1) * (L.d.nzl >> 1)) + 255) / 256);
#define OI_MG_LIN4(SRC, OUT, B, X, DST)                                                                                      \
  hipLaunchKernelGGL((k_mg_lin4<uint8_t, uint16_t, SRC, OUT>), dim3(blocks4), dim3(256), 0, s, L.dev<uint8_t, uint16_t>(), \
                     kfirst, L.d.nzl, B, X, DST)
  // big levels: through the LDS tile (k_mg_lin4t), with the restriction inside where the level's extents allow it -- periodic levels too
  const bool tiled_pre = quads && !two && mg->coarse_tiled && (long long)L.d.sz * L.d.nzl >= tile_from && kfirst == 1;
  const bool blkt = tiled_pre && mg->blocked && !mg->comm && (L.d.ny & 1) == 0 && (L.d.nzl & 1) == 0 && (L.d.z0 & 1) == 0 && !C.replicated &&
                    C.d.nx * 2 == L.d.nx && C.d.ny * 2 == L.d.ny && C.d.nzl * 2 == L.d.nzl && C.d.z0 * 2 == L.d.z0;
  const int ttx = (L.d.nx + kTX - 1) / kTX, tty = (L.d.ny + kTY - 1) / kTY, ttz = (L.d.nzl + kTZ - 1) / kTZ;
  // (two sweeps before and after the correction: the same tile for the three k_mg_lin4 steps of such a level)
  const bool tiled2 = quads && two && mg->coarse_tiled && (long long)L.d.sz * L.d.nzl >= tile_from && kfirst == 1;
#define OI_MG_LIN4T(SRC, OUT, B, X, DST)                                                                                              \
  do {                                                                                                                                \
    if (al4) hipLaunchKernelGGL((k_mg_lin4t<uint8_t, uint16_t, true, false, SRC, OUT>), dim3(ttx * tty * ttz), dim3(512), 0, s,        \
                                L.dev<uint8_t, uint16_t>(), B, DST, C.d, (float*)nullptr, ttx, tty, X);                                \
    else hipLaunchKernelGGL((k_mg_lin4t<uint8_t, uint16_t, false, false, SRC, OUT>), dim3(ttx * tty * ttz), dim3(512), 0, s,          \
                            L.dev<uint8_t, uint16_t>(), B, DST, C.d, (float*)nullptr, ttx, tty, X);                                    \
  } while (0)
  if (blkt) {
    if (al4) hipLaunchKernelGGL((k_mg_lin4t<uint8_t, uint16_t, true, true>), dim3(ttx * tty * ttz), dim3(512), 0, s, L.dev<uint8_t, uint16_t>(), rhs, (float*)nullptr, C.d, C.b, ttx, tty);
    else hipLaunchKernelGGL((k_mg_lin4t<uint8_t, uint16_t, false, true>), dim3(ttx * tty * ttz), dim3(512), 0, s, L.dev<uint8_t, uint16_t>(), rhs, (float*)nullptr, C.d, C.b, ttx, tty);
  } else if (blk) {  // pre-smoothing residual and its restriction in one launch; no residual array
    hipLaunchKernelGGL((k_mg_lin4r<uint8_t, uint16_t>), dim3(blocks8), dim3(256), 0, s, L.dev<uint8_t, uint16_t>(), rhs, C.d, C.b);
  } else if (!two) {
    if (tiled_pre && al4) hipLaunchKernelGGL((k_mg_lin4t<uint8_t, uint16_t, true, false>), dim3(ttx * tty * ttz), dim3(512), 0, s, L.dev<uint8_t, uint16_t>(), rhs, L.r, C.d, (float*)nullptr, ttx, tty);
    else if (tiled_pre) hipLaunchKernelGGL((k_mg_lin4t<uint8_t, uint16_t, false, false>), dim3(ttx * tty * ttz), dim3(512), 0, s, L.dev<uint8_t, uint16_t>(), rhs, L.r, C.d, (float*)nullptr, ttx, tty);
    else if (vec4 && blocks4 > 0) OI_MG_LIN4(0, 0, rhs, (const float*)nullptr, L.r);
    else OI_MG_LAUNCH(k_mg_pre, L, L.d.nzl, s, kfirst, rhs, L.r);
  } else {
    // x2 = two damped-Jacobi sweeps from zero (kept in L.t for the post-smoother), r = rhs - A x2
    if (tiled2) {
      OI_MG_LIN4T(0, 1, rhs, (const float*)nullptr, L.t);
    } else if (vec4 && blocks4 > 0) {
      OI_MG_LIN4(0, 1, rhs, (const float*)nullptr, L.t);
    } else {
      OI_MG_LAUNCH(k_mg_jacobi, L, L.d.nzl, s, kfirst, rhs, (const float*)nullptr, L.t2, 1);
      if (!ghost_exchange(mg, L, L.t2, sizeof(float), s, true)) return false;
      OI_MG_LAUNCH(k_mg_jacobi, L, L.d.nzl, s, kfirst, rhs, (const float*)L.t2, L.t, 0);
    }
    if (!ghost_exchange(mg, L, L.t, sizeof(float), s, true)) return false;
    if (tiled2) OI_MG_LIN4T(1, 0, rhs, (const float*)L.t, L.r);
    else if (vec4 && blocks4 > 0) OI_MG_LIN4(1, 0, rhs, (const float*)L.t, L.r);
    else OI_MG_LAUNCH(k_mg_resid, L, L.d.nzl, s, kfirst, rhs, (const float*)L.t, L.r);
  }
  if (blk || blkt) { if (!finish_child_array(mg, C, false, C.b, s, blkt)) return false; }
  else if (!restrict_to(mg, nullptr, nullptr, nullptr, &L, C, C.b, s)) return false;
  if (!coarse_cycle(mg, l + 1, C.b, C.x2, s)) return false;
  const float* second = nullptr;
  if ((mg->wmask >> (l + 1)) & 1u) {
    OI_MG_LAUNCH(k_mg_resid, C, C.d.nzl, s, kfirst, C.b, C.x2, C.b2);
    if (!ghost_exchange(mg, C, C.b2, sizeof(float), s, true)) return false;
    if (!coarse_cycle(mg, l + 1, C.b2, C.xs, s)) return false;
    second = C.xs;
  }
  // correction + post-smoothing: one sweep fused with the prolongation; with two sweeps a plain one follows
  const float* xpre = two ? L.t : nullptr;
  float* dst = two ? L.t2 : out;
  // (big levels: through an LDS tile of 64 x 8 x 8 cells, k_mg_post4t -- bit-identical)
  const bool tiled = quads && mg->coarse_tiled && (long long)L.d.sz * L.d.nzl >= tile_from && kfirst == 1;
  if (tiled) {
    const int tx_ = (L.d.nx + kTX - 1) / kTX, ty_ = (L.d.ny + kTY - 1) / kTY, tz_ = (L.d.nzl + kTZ - 1) / kTZ;
    if (al4) hipLaunchKernelGGL((k_mg_post4t<uint8_t, uint16_t, true>), dim3(tx_ * ty_ * tz_), dim3(512), 0, s, L.dev<uint8_t, uint16_t>(), rhs, C.x2, second, C.d,
                       mg->scale, xpre, dst, tx_, ty_);
    else hipLaunchKernelGGL((k_mg_post4t<uint8_t, uint16_t, false>), dim3(tx_ * ty_ * tz_), dim3(512), 0, s, L.dev<uint8_t, uint16_t>(), rhs, C.x2, second, C.d,
                       mg->scale, xpre, dst, tx_, ty_);
  } else if (vec4 && blocks4 > 0)
    hipLaunchKernelGGL((k_mg_post4<uint8_t, uint16_t>), dim3(blocks4), dim3(256), 0, s, L.dev<uint8_t, uint16_t>(), kfirst, L.d.nzl, rhs,
                       C.x2, second, C.d, mg->scale, xpre, dst);
  else
    OI_MG_LAUNCH(k_mg_post, L, L.d.nzl, s, kfirst, rhs, C.x2, second, C.d, mg->scale, xpre, dst);
  if (two) {
    if (!ghost_exchange(mg, L, L.t2, sizeof(float), s, true)) return false;
    if (tiled2) OI_MG_LIN4T(1, 1, rhs, (const float*)L.t2, out);
    else if (vec4 && blocks4 > 0) OI_MG_LIN4(1, 1, rhs, (const float*)L.t2, out);
    else OI_MG_LAUNCH(k_mg_jacobi, L, L.d.nzl, s, kfirst, rhs, (const float*)L.t2, out, 0);
  }
#undef OI_MG_LIN4
#undef OI_MG_LIN4T
  return ghost_exchange(mg, L, out, sizeof(float), s, true);
}

// z = M^-1 r on the fine level; `scratch` is a fine vector the caller does not need meanwhile.
// r's ghost planes must hold the neighbours' values.  Returns the number of z.r partials, -1 on a
// transport failure.
// the fine post-smoother with the partial sums of z.r: compacted vectors with their segment records at hand take the
// descriptor-addressed form (k_cg1b FORM 3), everything else k_mg_fine<2, 2>; same chunks, same sums
static int post_smooth(MgHierarchy* mg, const Geom& g, const uint8_t* code, const double* r, const MgCoarse& cref, double* z, double* partials,
                       const Ctrl* ctrl, const int* lineoff, const long long* plane_first, const void* seg, hipStream_t s,
                       const MgXUpdate* xu = nullptr) {
  if (mg_post_descr(g, lineoff, plane_first, seg))
    return launch_mg_post(g, code, r, cref, (double)mg->omega, (double)mg->scale, z, partials, ctrl, lineoff, plane_first, seg, s,
                          xu ? xu->x : nullptr, xu ? xu->p_prev : nullptr, xu ? xu->p_f32 : 0, xu ? xu->hist : nullptr, xu ? xu->k : 0,
                          xu ? xu->p_prev2 : nullptr);
  return launch_mg_fine(2, g, code, r, cref, (double)mg->omega, (double)mg->scale, z, nullptr, partials, ctrl, lineoff, s);
}
static int mg_vcycle_launch(MgHierarchy* mg, const Geom& g, const uint8_t* code, const double* r, double* scratch, double* z,
                            double* partials, const Ctrl* ctrl, int tiled, const int* lineoff, hipStream_t s,
                            const long long* plane_first, const void* seg) {
  auto& Ls = mg->levels;
  const long long nfine = (long long)g.nzl * g.ny * g.nx;
  MgLevelHost& C = Ls[0];
  MgCoarse cref{C.x2, nullptr, C.d.nx, C.d.ny, C.d.z0, C.d.sz, C.d.nzl, mg->periodic ? 1 : 0};
  const bool transition = mg->comm && C.replicated;  // owned contributions are summed across ranks afterwards
  // periodic problem: the LDS-tiled fine kernels wrap x / y by index; odd extents leave a single column / row / plane in the
  // last aggregates as in the open problem, and with an odd nx the pair (nx - 1, pad) keeps the wrapped column 0 in its
  // second LDS slot (k_fused_a, k_mg_fine)
  if (mg->periodic && !g.periodic) tiled = 0;
  if (tiled && (g.z0 & 1) == 0) {
    // pre-smoothing residual and its restriction in one pass: the residual itself is never stored
    if (transition) (void)hipMemsetAsync(C.b, 0, sizeof(float) * C.n, s);
    launch_mg_fine(3, g, code, r, cref, (double)mg->omega, (double)mg->scale, nullptr, C.b, partials, ctrl, lineoff, s);
    if (!finish_child_array(mg, C, false, C.b, s)) return -1;
  } else {
    if (lineoff) { oi::set_error("compacted vectors need the fused fine-level kernels (even first plane)"); return -1; }
    if (tiled) launch_mg_fine(1, g, code, r, cref, (double)mg->omega, (double)mg->scale, scratch, nullptr, partials, ctrl, nullptr, s);
    else hipLaunchKernelGGL(k_mg_fine_pre, dim3(nblk(nfine)), dim3(256), 0, s, g, code, r, scratch, ctrl, mg->periodic ? 1 : 0, (double)mg->omega);
    if (!restrict_to(mg, &g, code, scratch, nullptr, C, C.b, s)) return -1;
  }
  if (!coarse_cycle(mg, 0, C.b, C.x2, s)) return -1;
  if (mg->wmask & 1u) {
    OI_MG_LAUNCH(k_mg_resid, C, C.d.nzl, s, 1, C.b, C.x2, C.b2);
    if (!ghost_exchange(mg, C, C.b2, sizeof(float), s, true)) return -1;
    if (!coarse_cycle(mg, 0, C.b2, C.xs, s)) return -1;
    cref.xc2 = C.xs;
  }
  if (tiled)  // also leaves the per-block partial sums of z.r
    return post_smooth(mg, g, code, r, cref, z, partials, ctrl, lineoff, plane_first, seg, s);
  hipLaunchKernelGGL(k_mg_fine_post, dim3(nblk(nfine)), dim3(256), 0, s, g, code, r, cref, (double)mg->scale, z, ctrl, (double)mg->omega);
  return launch_dot(g, r, z, partials, s);
}

// ---- the cycle in two halves, for a caller that produces the first coarse right-hand side itself ----
// (the two-kernel iteration on compacted vectors forms r_k = r_{k-1} - alpha q, pre-smooths it and restricts it in ONE
// kernel, launch_mg_pre_update; it needs to know where the block sums go, and then runs everything after the restriction)
// where the fp32 right-hand side of the first coarse level lives, and its layout; *omega / the cycle's fixed parameters.
// `prepare` zeroes the target first where the owned contributions are summed across ranks afterwards.
bool mg_coarse_target(MgHierarchy* mg, MgCoarse* c, float** bcoarse, double* omega, hipStream_t s) {
  MgLevelHost& C = mg->levels[0];
  *c = MgCoarse{C.x2, nullptr, C.d.nx, C.d.ny, C.d.z0, C.d.sz, C.d.nzl, mg->periodic ? 1 : 0};
  *bcoarse = C.b;
  *omega = (double)mg->omega;
  if (mg->comm && C.replicated) return hipMemsetAsync(C.b, 0, sizeof(float) * C.n, s) == hipSuccess;
  return true;
}
// everything after the restriction: ghost planes / rank sum of the coarse right-hand side, the coarse cycle, and the fine
// post-smoother z = x2 + w (r - A x2) / c with the partial sums of z.r (returns their number, -1 on a transport failure)
int mg_vcycle_after_restriction(MgHierarchy* mg, const Geom& g, const uint8_t* code, const double* r, double* z, double* partials,
                                const Ctrl* ctrl, const int* lineoff, const long long* plane_first, const void* seg, hipStream_t s,
                                const MgXUpdate* xu, hipEvent_t before_post) {
  MgLevelHost& C = mg->levels[0];
  MgCoarse cref{C.x2, nullptr, C.d.nx, C.d.ny, C.d.z0, C.d.sz, C.d.nzl, mg->periodic ? 1 : 0};
  if (!finish_child_array(mg, C, false, C.b, s)) return -1;
  if (!coarse_cycle(mg, 0, C.b, C.x2, s)) return -1;
  if (mg->wmask & 1u) {
    OI_MG_LAUNCH(k_mg_resid, C, C.d.nzl, s, 1, C.b, C.x2, C.b2);
    if (!ghost_exchange(mg, C, C.b2, sizeof(float), s, true)) return -1;
    if (!coarse_cycle(mg, 0, C.b2, C.xs, s)) return -1;
    cref.xc2 = C.xs;
  }
  if (before_post) (void)hipEventRecord(before_post, s);
  return post_smooth(mg, g, code, r, cref, z, partials, ctrl, lineoff, plane_first, seg, s, xu);
}
bool mg_post_descr(const Geom& g, const int* lineoff, const long long* plane_first, const void* seg) {
  return !g.tune.mg_post_kernel && fused_descr_ok(g, lineoff, plane_first, seg, 0);
}

// (A hipGraph replay of the cycle's ~250 launches was measured in round 1: no gain on ROCm 7.2 -- back-to-back dependent
// launches already run at the dispatch latency -- and it did not survive rocprofv3 --kernel-trace; removed.)
int mg_vcycle(MgHierarchy* mg, const Geom& g, const uint8_t* code, const double* r, double* scratch, double* z,
              double* partials, const Ctrl* ctrl, int tiled, const int* lineoff, hipStream_t s, const long long* plane_first, const void* seg) {
  return mg_vcycle_launch(mg, g, code, r, scratch, z, partials, ctrl, tiled, lineoff, s, plane_first, seg);
}

int mg_levels(const MgHierarchy* mg) { return mg ? (int)mg->levels.size() + 1 : 0; }
int mg_replicated_from(const MgHierarchy* mg) {
  if (!mg) return -1;
  for (size_t l = 0; l < mg->levels.size(); ++l)
    if (mg->levels[l].replicated) return (int)l + 1;
  return -1;
}

int launch_dot(const Geom& g, const double* a, const double* b, double* partials, hipStream_t s) {
  const int blocks = nblk((long long)g.nzl * g.plane / 2, 2048);
  hipLaunchKernelGGL(k_mg_dot, dim3(blocks), dim3(256), 0, s, g, a, b, partials);
  return blocks;
}

}  // namespace oi
