// oi_kern.h -- device helpers and launch-shape arithmetic shared by the kernel translation units of the flow-through path:
//   oi_k_setup.hip   mask (flood fill), code bytes, compacted line lists, fluxes, synthetic structures, copies, remspot
//   oi_k_dense.hip   kernels of the DENSE vector layout: unfused CG passes, the LDS-tiled two-kernel iteration (k_fused_a, also the
//                    capacity-mode passes and the periodic apply), the wave-autonomous stand-alone apply (k_apply7), the fine halves
//                    of the multigrid cycle (k_mg_fine), the copy streams of the measurements
//   oi_k_cg1b.hip    the descriptor-addressed kernel of the COMPACTED layout in its four forms (single-kernel CG iteration; first
//                    kernel / residual update + pre-smoother + restriction / post-smoother of the multigrid-preconditioned iteration)
//                    and the launchers that choose between it and the LDS-tiled kernels
//   oi_k_reduce.hip  the final sums of the per-block partials, scalar records, stop test
//   oi_k_xcheck.hip  SUPERSEDED kernels kept as bit-identity references (k_cg1: the branching first form of the single-kernel
//                    iteration, dense or compacted) -- compiled into libopenimpala_amd_xcheck.so only (make xcheck; -DOI_XCHECK=1)
//
// Everything here is HBM-bandwidth-bound integer/fp64 streaming work: 64-wide wavefronts read x-contiguous rows with 16-byte
// lanes, the stencil's x/y reuse goes through LDS tiles, z reuse through registers, dot products through wavefront shuffles +
// one LDS slot per wave + per-block partials summed in a fixed order (no float atomics: bit-reproducible).  No MFMA: 0.26 flop/B.
// Compiled with -ffp-contract=off: every product/sum is rounded exactly as in the CPU oracle, so single-kernel outputs are
// bit-identical to it.
#pragma once
#include "oi_device.h"
#include "oi_pool.h"

#include <algorithm>
#include <cstdlib>
#include <mutex>

#ifndef OI_XCHECK
#define OI_XCHECK 0  // 1: the test library, with the superseded cross-check kernels (oi_k_xcheck.hip, k_fused_a on compacted vectors)
#endif

namespace oi {

// --------------------------------------------------------------------------------------
// small device helpers
// --------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// Sum over the block; result valid in thread 0.  `lds` holds >= blockDim/64 doubles.
__device__ __forceinline__ double block_sum(double v, double* lds) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  v = wave_sum(v);
  __syncthreads();  // protects lds reuse between consecutive calls
  if (lane == 0) lds[wave] = v;
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x == 0)
    for (int w = 0; w < nw; ++w) t += lds[w];
  return t;
}

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

__device__ __forceinline__ bool bit_at(const uint64_t* bits, const Geom& g, int gz, int j, int i) {
  return (bits[(long long)gz * g.wplane + (long long)j * g.W + (i >> 6)] >> (i & 63)) & 1ULL;
}

// Block -> logical tile (x fastest, then y, then z chunk) of the z-marching kernels.  HIP deals blocks b, b + 8, ... to
// one XCD (one L2).  map 2 (default): inside EVERY z chunk each XCD owns a contiguous band of tile rows, and chunks
// follow launch order, so the whole chip sweeps the volume in z like one linear stream -- the planes in flight are a
// few hundred MB that stay within DRAM-page / TLB reach and the chunk-boundary planes are re-read while still cached
// (1024^3 stand-alone apply: 3.00 ms against 3.40 ms with map 0, profiles/r02/apply_proto_*.txt).  map 0: every XCD
// one contiguous range of ALL tiles (eight distant regions at once); map 1: identity.
__device__ __forceinline__ int tile_of_block(int b, int tiles_xy, int nzc, int map) {
  if (map == 2 && (tiles_xy & 7) == 0) {
    const int per = tiles_xy >> 3, idx = b >> 3;
    return (idx / per) * tiles_xy + (b & 7) * per + idx % per;
  }
  const int ntiles = tiles_xy * nzc;
  if (map == 0 && (ntiles & 7) == 0) return (b & 7) * (ntiles >> 3) + (b >> 3);
  return b;
}


static inline int cap_blocks(long long work_items, int per_block, int cap) {
  long long b = (work_items + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (int)b;
}



// --------------------------------------------------------------------------------------
// Direct (cache-served) masked 7-point apply on voxel pairs.  Summation order is the
// oracle's: ((((xm + xp) + ym) + yp) + zm) + zp, then code*v - sum.
// --------------------------------------------------------------------------------------
struct Pair {
  double a, b;
};

__device__ __forceinline__ double2 ld2(const double* p) { return *reinterpret_cast<const double2*>(p); }
__device__ __forceinline__ void st2(double* p, double a, double b) {
  *reinterpret_cast<double2*>(p) = make_double2(a, b);
}
// the same store with a cache policy chosen at run time (Tune::dense_nt, a wave-uniform flag): nontemporal for a streamed array
typedef double v2dn_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void st2p(double* p, double a, double b, bool nt) {
  if (nt) { v2dn_t v = {a, b}; __builtin_nontemporal_store(v, reinterpret_cast<v2dn_t*>(p)); }
  else st2(p, a, b);
}

// (A v) for the pair at element e = (gz, j, i..i+1); i even.  Needs the row/plane bounds.
__device__ __forceinline__ Pair apply_pair(const Geom& g, const double* __restrict__ v, long long e, int i, int j,
                                           unsigned c0, unsigned c1) {
  const double2 c = ld2(v + e);
  const double xm = (i > 0) ? v[e - 1] : 0.0;
  const double xp = (i + 2 < g.pitch) ? v[e + 2] : 0.0;  // columns >= nx hold zeros
  double2 ym = make_double2(0.0, 0.0), yp = make_double2(0.0, 0.0);
  if (j > 0) ym = ld2(v + e - g.pitch);
  if (j < g.ny - 1) yp = ld2(v + e + g.pitch);
  const double2 zm = ld2(v + e - g.plane), zp = ld2(v + e + g.plane);
  Pair q;
  const double s0 = ((((xm + c.y) + ym.x) + yp.x) + zm.x) + zp.x;
  const double s1 = ((((c.x + xp) + ym.y) + yp.y) + zm.y) + zp.y;
  q.a = c0 ? (double)c0 * c.x - s0 : 0.0;
  q.b = c1 ? (double)c1 * c.y - s1 : 0.0;
  return q;
}

#define OI_PAIR_LOOP_BEGIN                                                                   \
  const int hp = g.pitch >> 1;                                                               \
  const long long npairs = (long long)g.nzl * g.ny * hp;                                     \
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < npairs;           \
       t += (long long)gridDim.x * blockDim.x) {                                             \
    const int i = 2 * (int)(t % hp);                                                         \
    const int j = (int)((t / hp) % g.ny);                                                    \
    const long long e = g.plane + 2 * t; /* owned planes start after the lo ghost plane */ \
    const unsigned short cc = *reinterpret_cast<const unsigned short*>(code + e);            \
    const unsigned c0 = cc & 0xffu, c1 = cc >> 8;
#define OI_PAIR_LOOP_END }

// --------------------------------------------------------------------------------------
// CG vector kernels (unfused variant): flat 16-byte streams over the owned planes.
// alpha_k = rz_k / pq_k and beta_k = rz_k / rz_{k-1} are formed in registers from the
// device-resident history, so the host never sits in the loop.  A vanishing
// denominator (exact convergence) turns the step into a no-op instead of a NaN.
// --------------------------------------------------------------------------------------
__device__ __forceinline__ double safe_div(double a, double b) { return b != 0.0 ? a / b : 0.0; }

// CG vectors vanish on non-free voxels, so a 128-byte line (8 lanes x 16 B) none of whose lanes holds a free
// voxel is neither read nor written.  Inside a LIVE line every lane takes part (a pair without a free voxel
// moves zeros): HBM then never has to merge a partial-line store, which measures 8-15 % faster than
// predicating pair by pair.  Must be called by all lanes of the wave that are in the loop.
__device__ __forceinline__ bool line_live(bool pair_live) {
  const unsigned long long m = __ballot(pair_live);
  return ((m >> (threadIdx.x & (64 - kDenseLG))) & ((1ull << kDenseLG) - 1)) != 0;
}

// typed loads/stores of the search direction: p may be STORED as fp32 (opts.p_storage),
// all arithmetic stays fp64
template <typename PT> struct PIo;
template <> struct PIo<double> {
  static __device__ __forceinline__ double2 ld_pair(const double* p) { return ld2(p); }
  static __device__ __forceinline__ void st_pair(double* p, double a, double b) { st2(p, a, b); }
  static __device__ __forceinline__ double rnd(double v) { return v; }
};
template <> struct PIo<float> {
  static __device__ __forceinline__ double2 ld_pair(const float* p) {
    const float2 f = *reinterpret_cast<const float2*>(p);
    return make_double2((double)f.x, (double)f.y);
  }
  static __device__ __forceinline__ void st_pair(float* p, double a, double b) {
    *reinterpret_cast<float2*>(p) = make_float2((float)a, (float)b);
  }
  // the value the next iteration will read back: the stencil, p.q and the x update all use it
  static __device__ __forceinline__ double rnd(double v) { return (double)(float)v; }
};

// p2 != null: a second term, x = (x + alpha_k p) + alpha_{k+1} p2 -- the two dependent additions two launches would make, x moved once

// tile of the z-marching LDS kernels (k_fused_a, k_mg_fine, k_cg1, k_cg1b)
constexpr int kBX = 128;            // tile width in voxels = 64 lanes x 2
constexpr int kLP = kBX + 4;        // LDS row pitch in doubles
constexpr int kNW = 8;              // waves per block
constexpr int kBX_lines = 128 / kLV;  // stored lines per row of a 128-voxel tile column (segment records)

// Defaults from the 1024^3 sweep in profiles/r01/tile_sweep.txt: 16-row tiles for the CG
// kernel (32 rows need 190 VGPRs -> one block per CU), 32-row tiles for the stand-alone apply.
// (tile shapes and chunk lengths are per-handle state: Geom::tune)

// Planes per block of the z-marching kernels.  Large volumes: >= ~4096 blocks, but chunks of >= 64 planes so the
// two redundant halo planes per chunk stay near 3 %.  Small volumes (the reference's 100^3 sample gives 7 tile
// columns) would leave most of the 256 CUs idle behind a few long marches: there the chunks shrink, down to 8
// planes, until ~1024 blocks exist -- the march is latency-bound, so halo planes are cheap next to idle CUs.
static inline int auto_zchunk(const Geom& g, long long cols) {
  long long want = (4096 + cols - 1) / cols;
  if (want < 1) want = 1;
  long long zc = (g.nzl + want - 1) / want;
  if (zc < 64) {
    const long long blocks64 = cols * ((g.nzl + 63) / 64);
    if (blocks64 >= 1024) {
      zc = 64;
    } else {
      zc = ((long long)g.nzl * cols + 1023) / 1024;
      if (zc < 8) zc = 8;
      if (zc > 64) zc = 64;
      zc = (zc + 1) & ~1ll;  // even: z pairs of the fused restriction stay inside a chunk
    }
  }
  return (int)zc;
}

// planes per block and number of z chunks of a z-marching launch over `cols` tile columns: the handle's fixed choice
// (Tune::zchunk) or the automatic one, kept within the handle's partial-sum capacity (Tune::max_blocks)
static inline void pick_chunks(const Geom& g, long long cols, bool even, int* zchunk_out, int* nzc_out) {
  int zchunk = g.tune.zchunk > 0 ? g.tune.zchunk : auto_zchunk(g, cols);
  if (zchunk > g.nzl) zchunk = g.nzl;
  if (zchunk < 1) zchunk = 1;
  long long nzc = (g.nzl + zchunk - 1) / zchunk;
  const long long cap = g.tune.max_blocks > 0 ? g.tune.max_blocks : kMaxPartials;
  if (cols * nzc > cap) {  // the partial-sum array bounds the grid
    const long long maxc = cap / cols > 0 ? cap / cols : 1;
    zchunk = (int)((g.nzl + maxc - 1) / maxc);
    nzc = (g.nzl + zchunk - 1) / zchunk;
  }
  if (even && (zchunk & 1) && zchunk < g.nzl) {  // chunks must hold whole z pairs
    ++zchunk;
    nzc = (g.nzl + zchunk - 1) / zchunk;
  }
  *zchunk_out = zchunk;
  *nzc_out = (int)nzc;
}
// a launch whose tile columns alone exceed the partial-sum capacity cannot run (create sizes the array by
// partial_blocks_bound, so this only guards against a handle driven with a foreign geometry)
static inline bool grid_fits(const Geom& g, long long cols, int nzc) {
  return cols * nzc <= (g.tune.max_blocks > 0 ? g.tune.max_blocks : kMaxPartials);
}


// The first kernel of the two-kernel iteration.  Compacted vectors with their segment records at hand (the multigrid-
// preconditioned solve): the descriptor-addressed form, k_cg1b<.., FORM 1>; everything else: k_fused_a.  BOTH take the
// chunking of fused_chunks, so that the per-block partial sums -- and with them the iterates -- are the same bit for bit
// whichever kernel and layout runs (tests: ..._layouts_compacted_and_dense, ..._two_solver_paths_agree).
static inline void fused_chunks(const Geom& g, long long cols, bool even, int* zchunk, int* nzc) {
  Geom gg = g;
  // (a periodic problem on dense vectors keeps the chunks it always had; on compacted vectors -- live_lines is set -- it takes these)
  if (g.tune.zchunk <= 0 && g.tune.fused_zchunk > 0 && (!g.periodic || g.live_lines > 0.f) &&
      cols * ((g.nzl + g.tune.fused_zchunk - 1) / g.tune.fused_zchunk) >= 2048)
    gg.tune.zchunk = g.tune.fused_zchunk;
  pick_chunks(gg, cols, even, zchunk, nzc);
}

enum { R1_RR = 0, R1_RZ, R1_PQ, R1_ZQ, R1_QDQ, R1_ALPHA, R1_BETA, R1_STRIDE = 8 };

// From the five sums of record k (all-reduced in multi-rank runs): alpha_k, the one-step extrapolation beta_{k+1}, and the
// convergence test on the exact ||r_k||.  Evaluated (a) by EVERY block of kernel k + 1 in its prologue -- same inputs, same
// arithmetic, same result everywhere, so no launch sits between the reduction and the next iteration -- and (b) by
// k_scalars1 once per host batch, so that the last record of a batch is closed too.
struct Cg1Scalars { double alpha, beta; bool converged; };
__device__ __forceinline__ Cg1Scalars cg1_scalars_of(const double* rk, double tol2) {
  Cg1Scalars o;
  o.converged = rk[R1_RR] <= tol2;
  o.alpha = safe_div(rk[R1_RZ], rk[R1_PQ]);
  double rz_next = rk[R1_RZ] - 2.0 * o.alpha * rk[R1_ZQ] + o.alpha * o.alpha * rk[R1_QDQ];
  if (!(rz_next > 0.0)) rz_next = 0.0;  // cancellation at exact convergence: restart direction
  o.beta = safe_div(rz_next, rk[R1_RZ]);
  return o;
}

// COMPACT: r, q and p live in the COMPACTED layout -- only the aligned kLV-voxel lines (8 voxels = 64 B) that hold a free
// voxel are stored, in (plane, row, line) order; lineoff[(gz * ny + j) * LW + line] is a line's slot (-1:
// not stored).  Same bytes as the predicated dense layout, but contiguous in HBM.  x stays dense.

// ---- launch entry points that cross translation units (not part of oi_device.h: nothing outside the kernel files calls them) ----
// the LDS-tiled first kernel of the two-kernel iteration (k_fused_a); lineoff != null (compacted vectors) only in the test library
int launch_fused_a_lds(const Geom& g, const uint8_t* code, double* x, const double* r, const void* p_in, void* p_out, int p_f32, double* q,
                       int jacobi, const double* hist, int k, double* partials, const Ctrl* ctrl, const int* lineoff, hipStream_t s,
                       int blocks, int zchunk, int tiles_x, int tiles_y, int nzc);
// the branching first form of the single-kernel iteration (k_cg1; oi_k_xcheck.hip).  -2: not in this library
int launch_cg1_first_form(int R, int NW, int blocks, const Geom& g, const uint8_t* code, double* x, const double* r_in, double* r_out,
                          const double* q_in, double* q_out, const void* p_in, void* p_out, int p_f32, int jacobi, double* rec, int k, int zchunk,
                          int tiles_x, int tiles_y, int nzc, int part, int partial_base, double* partials, Ctrl* ctrl, const int* lineoff,
                          hipStream_t s);

}  // namespace oi
