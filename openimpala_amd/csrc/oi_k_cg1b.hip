// oi_k_cg1b.hip -- the descriptor-addressed kernel of the compacted layout (four forms) and its launchers.  See oi_kern.h.
#include <type_traits>
#include "oi_kern.h"

namespace oi {


// --------------------------------------------------------------------------------------
// k_cg1b: the same iteration (same arithmetic, same order of operations, bit-identical results) with the memory side
// rebuilt.  What the shader counters said about k_cg1 (profiles/r02/k_cg1_sq_counters.txt): waves wait 83 % of their
// time, the vector ALUs are busy 20 %, HBM is not pushing back -- and its ISA shows why: every `if (line stored)` is a
// divergent branch with its loads inside, so a wave walks through up to FIVE dependent HBM round trips per plane (row 0's
// r/q/p, row 0's x and p_{k-2}, row 1's r/q/p, row 1's x, the halo).  Here every access goes through a buffer
// descriptor rebuilt per plane from scalars (plane base + 32-bit lane offset; a lane that has nothing to load or store
// gets an out-of-range offset and the hardware drops it: predication without a branch).  All loads of a plane -- code
// bytes and line slots two planes ahead, r/q/p, x, p_{k-2}, the halo's r/q/p -- are issued back to back in straight-line
// code and waited for once.  Compacted layout only (the dense layout keeps k_cg1).
// --------------------------------------------------------------------------------------
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
constexpr int kOob = 0x7ffff000;          // lane offset that every descriptor's range check rejects
#ifndef OI_TIMING_KNOCK
#define OI_TIMING_KNOCK 0  // TIMING BUILDS ONLY (tools/variant_build.sh + tools/knock_bench.py, wrong results): 1 no barrier per plane, 2 no LDS
                           // reads, 4 no LDS writes, 8 no vector traffic (every offset out of range), 16 code bytes / records of one plane
#endif
#ifndef OI_CG1B_EARLY
#define OI_CG1B_EARLY 1   // 0: the first form (requests at the top of the iteration), kept for A/B builds (tools/early_issue_ab.sh)
#endif
constexpr long long kBufMax = 0x7fff0000;  // largest range (bytes) a descriptor may cover here

__device__ __forceinline__ __amdgpu_buffer_rsrc_t buf_of(const void* p, long long bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
template <int AUX = 0>
__device__ __forceinline__ double2 bld2(__amdgpu_buffer_rsrc_t r, int off) {
  return __builtin_bit_cast(double2, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, AUX));
}
__device__ __forceinline__ double bld1(__amdgpu_buffer_rsrc_t r, int off) {
  return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0));
}
// AUX = the store's cache policy (the instruction's aux operand): 0 default, 2 nontemporal
template <int AUX = 0>
__device__ __forceinline__ void bst2(__amdgpu_buffer_rsrc_t r, int off, double a, double b) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, make_double2(a, b)), r, off, 0, AUX);
}
// the search direction through a descriptor; `off` is the byte offset the fp64 vectors use (fp32 storage: half of it).
// Loads return the bits as they were stored (Pair / One); they are widened when they are used, not when they are issued,
// so that no conversion sits between the loads of a plane.
template <typename PT> struct PBuf;
template <> struct PBuf<double> {
  typedef double2 Pair;
  typedef double One;
  static constexpr int kLineBytes = 8 * kLV;
  template <int AUX = 0>
  static __device__ __forceinline__ Pair ld_pair(__amdgpu_buffer_rsrc_t r, int off) { return bld2<AUX>(r, off); }
  static __device__ __forceinline__ One ld_one(__amdgpu_buffer_rsrc_t r, int off) { return bld1(r, off); }
  static __device__ __forceinline__ double2 wide(Pair v) { return v; }
  static __device__ __forceinline__ double wide1(One v) { return v; }
  template <int AUX = 0>
  static __device__ __forceinline__ void st_pair(__amdgpu_buffer_rsrc_t r, int off, double a, double b) { bst2<AUX>(r, off, a, b); }
};
template <> struct PBuf<float> {
  typedef float2 Pair;
  typedef float One;
  static constexpr int kLineBytes = 4 * kLV;
  template <int AUX = 0>
  static __device__ __forceinline__ Pair ld_pair(__amdgpu_buffer_rsrc_t r, int off) {
    return __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(r, off >> 1, 0, AUX));
  }
  static __device__ __forceinline__ One ld_one(__amdgpu_buffer_rsrc_t r, int off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off >> 1, 0, 0));
  }
  static __device__ __forceinline__ double2 wide(Pair f) { return make_double2((double)f.x, (double)f.y); }
  static __device__ __forceinline__ double wide1(One f) { return (double)f; }
  template <int AUX = 0>
  static __device__ __forceinline__ void st_pair(__amdgpu_buffer_rsrc_t r, int off, double a, double b) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, make_float2((float)a, (float)b)), r, off >> 1, 0, AUX);
  }
};

// STAUX / LDAUX: which vectors leave / arrive with NONTEMPORAL accesses (bit masks; stores: 1 r_out, 2 q_out (FORM 3: z), 4 p_out,
// 8 x; loads of the tile's own rows: 1 r_in, 2 q_in, 4 p_in, 8 x, 16 p_{k-2}; the halo and the code bytes always take the default
// policy: they are what the L2 is for).  Chosen per form by in-process A/B on one handle (tools/pol_ab.py, profiles/r04/
// k_cg1b_cache_policies_*.txt): r, p and x written nontemporally and x, p_{k-2} read nontemporally take 7.6 % off FORM 0, while
// nontemporal loads of r, q, p (re-read by the neighbouring tiles' halos) cost 10 %.
// a / c for a neighbour count c = 1 .. 6 without the division instruction sequence: yc = {RN(1 / c), c};
// q = RN(a y), r = a - c q (exact in one fma: q is a faithful quotient, |1/c - y| <= 2^-54 / c for these six divisors), q' = RN(q + r y)
// is the CORRECTLY ROUNDED quotient (Markstein) -- the bits of a / c, in three instructions instead of about fourteen.  Checked
// against the division for 2.5e9 operands per divisor set (random and mantissa edge patterns); the Jacobi-preconditioned
// single-kernel iteration divides four times per voxel and was instruction-bound on it (config 5's share).
__device__ __forceinline__ double div_small(double a, double2 yc) {
  const double q = a * yc.x;
  const double r = __builtin_fma(-yc.y, q, a);
  return __builtin_fma(r, yc.x, q);
}
constexpr int kNtSt0 = 13, kNtLd0 = 24;
// (the light forms are instruction-bound and gain less: FORM 1 p_out 1.9 %, FORM 2 r_out 0.7 %, FORM 3 x written and x, p_{k-1} read 1.4 %
// of the cycle; q_out / z nontemporal is a loss everywhere: the next kernel of the same iteration reads it first)
constexpr int kNtSt1 = 4, kNtLd1 = 0, kNtSt2 = 1, kNtLd2 = 0, kNtSt3 = 8, kNtLd3 = 12 + 16;
// FORM 0: the single-kernel CG iteration described above.
// FORM 1: the first kernel of the TWO-kernel iteration (what k_fused_a<.., compacted> computes, for the multigrid-preconditioned
// solve: x += alpha_{k-1} p_{k-1}; p_k = z_k + beta_k p_{k-1}; q_k = A p_k; partial p_k.q_k) on the same memory side --
// `r_in` is the preconditioned residual z, q_in / r_out are not touched, `rec` is the device history of the two-kernel
// iteration (hist[3k + {p.q, r.r, r.z}]), one partial sum per block.  Same arithmetic in the same order as k_fused_a.
// FORM 2: the residual update of the two-kernel iteration FUSED with the fine-level pre-smoother and the restriction of the
// multigrid cycle (k_update_r_lines + k_mg_fine MODE 3 in one pass): r_k = r_{k-1} - alpha_{k-1} q_{k-1} (read from r_in / q_in,
// stored to r_out -- double-buffered, the halo ring recomputes it from the inputs), partial r_k.r_k, v = omega r_k / c staged,
// res = r_k - A v, and the 2x2x2 block sums of res as the fp32 right-hand side of the first coarse level (`bcoarse`, `C`).
// Chunks hold whole z pairs (even first plane, even length).  Same arithmetic in the same order as the two kernels it replaces.
template <int R, int NW, typename PT, bool JACOBI, bool SEG, int FORM = 0, int STAUX = 0, int LDAUX = 0, int VAR = 0, bool NOX = false>
__global__ void __launch_bounds__(64 * NW, (R <= 2 ? 4 : 2)) k_cg1b(Geom g, const uint8_t* __restrict__ code, double* __restrict__ x,
                                                    const double* __restrict__ r_in, double* __restrict__ r_out,
                                                    const double* __restrict__ q_in, double* __restrict__ q_out,
                                                    const PT* __restrict__ p_in, PT* __restrict__ p_out,
                                                    double* rec, int k, int xskip, int zchunk,
                                                    int tiles_x, int tiles_y, int nzc, int part, int partial_base,
                                                    double* __restrict__ partials, Ctrl* ctrl,
                                                    const int* __restrict__ lineoff, const long long* __restrict__ plane_first,
                                                    const uint2* __restrict__ seg, int LW, int fullline_arg,
                                                    float* __restrict__ bcoarse, MgCoarse C, double omega, double scale) {
  constexpr int BY = NW * R;
  const int fullline = (VAR & 1) ? 1 : fullline_arg;  // (VAR: EXPERIMENT instantiations only -- 1 whole lines as a compile-time fact, 2 / 4 no halo reads)
  constexpr int TS = (BY + 2) * kLP;  // doubles per staged plane
  static_assert(2 * BY <= 64, "wave 2 loads the 2*BY x-halo values with one lane each");
  if (!(OI_TIMING_KNOCK & 8) && ctrl && ctrl->done) return;
  // p_k of three consecutive planes (ring): the z neighbours are read from here, not carried in registers
  __shared__ __attribute__((aligned(16))) double tile[3 * TS];
  __shared__ double red[NW];
  // FORM 2: y-pair sums of the restriction (odd waves publish, even waves own the aggregates), omega / c for c = 0..6
  __shared__ double ysum[FORM == 2 ? 2 : 1][FORM == 2 ? R * (NW / 2) * 64 : 1];
  __shared__ double wtab[8];
  __shared__ double2 jtab[JACOBI ? 8 : 1];  // {1 / c, c} for div_small
  if (FORM >= 2) {
    static_assert(FORM != 2 || (NW & 1) == 0, "rows 2m and 2m+1 of a tile belong to waves w and w+1");
    if (threadIdx.x < 8) wtab[threadIdx.x] = threadIdx.x ? omega / (double)threadIdx.x : 0.0;
    __syncthreads();
  }
  if (JACOBI) {
    if (threadIdx.x < 8) jtab[threadIdx.x] = threadIdx.x ? make_double2(1.0 / (double)threadIdx.x, (double)threadIdx.x) : make_double2(0.0, 0.0);
    __syncthreads();
  }

  const int nsel = part == 0 ? nzc : (part == 1 ? min(2, nzc) : nzc - 2);
  const int L = tile_of_block(blockIdx.x, tiles_x * tiles_y, nsel, g.tune.tile_map);
  const int tx = L % tiles_x, ty = (L / tiles_x) % tiles_y, zsel = L / (tiles_x * tiles_y);
  const int zc = part == 0 ? zsel : (part == 1 ? (zsel == 0 ? 0 : nzc - 1) : zsel + 1);

  double beta = 0.0, alpha_prev = 0.0, alpha_prev2 = 0.0;
  if (FORM == 1) {
    if (k > 0) {
      beta = safe_div(rec[H_STRIDE * k + H_RZ], rec[H_STRIDE * (k - 1) + H_RZ]);
      alpha_prev = safe_div(rec[H_STRIDE * (k - 1) + H_RZ], rec[H_STRIDE * (k - 1) + H_PQ]);
    }
  } else if (FORM == 2) {
    if (k > 0) alpha_prev = safe_div(rec[H_STRIDE * (k - 1) + H_RZ], rec[H_STRIDE * (k - 1) + H_PQ]);
  } else if (FORM == 3) {  // (x += alpha_{k-1} p_{k-1} rides along when the caller hands x over: see do_x)
    if (k > 0 && x) alpha_prev = safe_div(rec[H_STRIDE * (k - 1) + H_RZ], rec[H_STRIDE * (k - 1) + H_PQ]);
  } else if (k > 0) {
    double* rk1 = rec + R1_STRIDE * (k - 1);
    const Cg1Scalars sc = cg1_scalars_of(rk1, ctrl->tol2);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      rk1[R1_ALPHA] = sc.alpha;
      rk1[R1_BETA] = sc.beta;
      if (!(OI_TIMING_KNOCK & 8) && sc.converged) ctrl->done = k;
    }
    if (!(OI_TIMING_KNOCK & 8) && sc.converged) return;
    alpha_prev = sc.alpha;
    beta = sc.beta;
  }
  // FORM 1: x takes alpha_{k-1} p_{k-1} in every launch -- unless the post-smoother of the same iteration (FORM 3, launched just
  // before with x and p_{k-1}) has applied it already: that kernel is instruction-bound and carries the 24 bytes per voxel for
  // less than they cost here (xskip < 0 says so)
  const bool x_elsewhere = FORM == 1 && xskip < 0;
  // FORM 3 (round 4): xskip = 1 makes its x update PAIRWISE as in FORM 0 -- every second launch x takes alpha_{k-2} p_{k-2} and then
  // alpha_{k-1} p_{k-1} (p_{k-2} is what the other p buffer, `p_out`, still holds: the first kernel of THIS iteration overwrites it
  // only after this kernel).  Two dependent additions in that order: the same bits as two launches that add one term each.
  if (FORM != 0 && FORM != 3) xskip = 0;
  // NOX: a launch the HOST knows will not touch x (FORM 1 when the post-smoother carries the update, FORM 3 in the launches
  // between two pairwise updates) is its own instantiation: x, p_{k-1} / p_{k-2} and the alphas are not even live -- the light
  // forms are instruction-bound and FORM 3 reloaded 74 spilled scalars per plane step with the x path compiled in
  const bool do_x = (NOX || FORM == 2) ? false
                    : FORM == 3 ? (x != nullptr && (xskip ? (k >= 2 && (k & 1) == 0) : k > 0))
                    : x_elsewhere ? false : (xskip ? (k >= 2 && (k & 1) == 0) : (k > 0));
  if (xskip && do_x)
    alpha_prev2 = FORM == 3 ? safe_div(rec[H_STRIDE * (k - 2) + H_RZ], rec[H_STRIDE * (k - 2) + H_PQ]) : rec[R1_STRIDE * (k - 2) + R1_ALPHA];

  const int lane = threadIdx.x & 63;
  const int wu = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // the wave index as a SCALAR: rows are scalars too
  const int i = tx * kBX + 2 * lane, j0 = ty * BY;
  const bool in_x = i < g.pitch;
  const int gs = 1 + zc * zchunk, ge = min(g.nzl + 1, gs + zchunk);
  // per-lane parts of the offsets inside a plane (the row part is a scalar, added when the access is formed)
  const int ivox = in_x ? i : kOob;                          // code bytes; x 8: x
  const int ivox8 = in_x ? 8 * i : kOob;
  const int ilin = in_x ? (i >> kLVs) * 4 : kOob;            // slot table
  const int sub = (i & (kLV - 1)) * 8;                       // this lane's pair inside its line (fp64 bytes)
  // halo lanes: waves 0 and 1 the rows below / above the tile (a voxel pair per lane), wave 2 the columns left / right
  // VAR & 8: a PERIODIC problem (the homogenization cell problems, Geom::periodic; its own instantiations, nothing of it in the
  // others).  The halo of the domain's first row / column is its last one and vice versa; where the domain ends INSIDE a tile
  // the wrapped row / column is staged right behind the last one (hlds / xcol), in the slot of the first row / pair outside the
  // domain, which stage_rows then leaves alone (k_fused_a does the same); z wraps through the ghost planes, which hold the images
  constexpr bool per = (VAR & 8) != 0;
  const bool ywrap_hi = per && j0 + BY >= g.ny, xwrap_hi = per && tx * kBX + kBX >= g.nx;
  const int hrow = (wu == 0) ? ((per && j0 == 0) ? g.ny - 1 : j0 - 1) : (ywrap_hi ? 0 : j0 + BY);
  const int hlds = (wu == 0) ? 0 : (ywrap_hi ? g.ny - j0 + 1 : BY + 1);   // LDS row of this wave's halo row (scalar)
  const int xcol = xwrap_hi ? 2 + g.nx - tx * kBX : 2 + kBX;               // LDS column of the right halo column (scalar)
  const bool yhalo = (wu < 2) && in_x && hrow >= 0 && hrow < g.ny;
  const int xh_row = j0 + (lane >> 1);
  const int xh_i = (lane & 1) ? (xwrap_hi ? 0 : tx * kBX + kBX) : ((per && tx == 0) ? g.nx - 1 : tx * kBX - 1);
  const bool xhalo = (wu == 2) && lane < 2 * BY && xh_row < g.ny && xh_i >= 0 && xh_i < g.pitch;
  const int hvox = yhalo ? hrow * g.pitch + i : (xhalo ? xh_row * g.pitch + xh_i : kOob);
  const int hlin = yhalo ? (hrow * LW + (i >> kLVs)) * 4 : (xhalo ? (xh_row * LW + (xh_i >> kLVs)) * 4 : kOob);
  const int hsub = yhalo ? sub : (xh_i & (kLV - 1)) * 8;
  const bool is_halo = yhalo || xhalo;

  const long long plane_lines4 = 4ll * g.ny * LW;
  // SEG: the slots of a row segment come from its segment record (k_line_finalize) -- one scalar load per row instead of a
  // table entry fetched by every lane.  `lsel` = this lane's line inside the 128-voxel segment
  // (the bit masks are re-formed from an opaque copy of the lane index where they are used: two VGPRs held across the loop
  // were two spills at the 128-register budget, see stage_halo)
  auto slot_from = [&](uint2 rec, unsigned bit, unsigned below) { return (rec.y & bit) ? (int)rec.x + __builtin_popcount(rec.y & below) : -1; };
  auto slot_own = [&](uint2 rec) {
    int l_ = threadIdx.x & 63;
    asm volatile("" : "+v"(l_));
    const unsigned bit = 1u << (l_ >> (kLVs - 1));
    return slot_from(rec, bit, bit - 1u);
  };
  const long long seg_plane = (long long)g.ny * tiles_x;
  // Code bytes, records and table entries are fetched two planes ahead of the march.  Inside the loop their bases RUN -- one 64-bit
  // addition per step instead of a 64-bit product per use (round 4: the light forms are instruction-bound, a third of their
  // ~280 scalar instructions per plane step was address arithmetic); the prologue's few calls form theirs from the plane number
  const uint8_t* code_run = code;                                   // -> plane gz + 2 while the loop runs (set before it)
  const uint2* seg_run = seg;                                       // -> its records of this tile column (+ tx)
  const char* slot_run = reinterpret_cast<const char*>(lineoff);    // -> its table entries
  auto code_buf = [&](int p, bool run) { return buf_of(run ? code_run : code + (long long)p * g.plane, g.plane); };
  auto slot_buf = [&](int p, bool run) {
    return buf_of(run ? slot_run : reinterpret_cast<const char*>(lineoff) + (long long)p * plane_lines4, plane_lines4);
  };
  auto seg_row = [&](int p, int j, bool run) { return run ? seg_run[(long long)j * tiles_x] : seg[(long long)p * seg_plane + (long long)j * tiles_x + tx]; };
  auto x_buf = [&](const double* xplane) { return buf_of(xplane, 8ll * g.plane); };
  auto x_plane = [&](int p) { return x + (long long)p * g.plane; };
  auto row_of = [&](int rr_) { return j0 + wu + rr_ * NW; };  // scalar

  // ---- code bytes and line slots of plane p (they arrive one iteration before they are used) ----
  // SEG, main loop: the records are only REQUESTED here (scalar loads, `recs`); their slots are formed at the end of the
  // iteration (slots_of) -- forming them here would wait for the scalar loads in the middle of the issue phase
  uint2 rec_nn[R];
  auto codes_issue = [&](int p, unsigned short (&cn)[R], int (&ln)[R], bool defer) {
#pragma unroll
    for (int rr_ = 0; rr_ < R; ++rr_) { cn[rr_] = 0; ln[rr_] = -1; rec_nn[rr_] = make_uint2(0u, 0u); }
#if OI_TIMING_KNOCK & 16  // TIMING BUILDS ONLY: the code bytes and records of the chunk's first plane for every plane (cache hits)
    if (defer) p = gs;
#endif
    if (p <= g.nzl + 1) {
#if OI_TIMING_KNOCK & 16
      const bool run = false;
#else
      const bool run = defer;  // (the loop's calls: plane gz + 2, where the running bases point)
#endif
      const __amdgpu_buffer_rsrc_t cb = code_buf(p, run), sb = slot_buf(p, run);
#pragma unroll
      for (int rr_ = 0; rr_ < R; ++rr_) {
        const int j = row_of(rr_);
        if (j < g.ny) {  // (scalar)
          // (the row's offset is added into the LANE offset, not passed as the instruction's scalar offset: with the scalar
          // offset the fp32-p build of this kernel read x from a wrong row now and then -- the compiler reuses the scalar
          // register two instructions after the load that names it; profiles/r02/k_cg1b_soffset_hazard.txt)
          cn[rr_] = __builtin_amdgcn_raw_buffer_load_b16(cb, ivox + j * g.pitch, 0, 0);
          // a lane outside the row reads 0 from the range check, which would be slot 0: such lanes have code 0 and the
          // slot is only looked at together with in_x
          if (SEG) {
            rec_nn[rr_] = seg_row(p, j, run);  // (uniform address: a scalar load)
            if (!defer) ln[rr_] = slot_own(rec_nn[rr_]);
          } else {
            ln[rr_] = (int)__builtin_amdgcn_raw_buffer_load_b32(sb, ilin + j * LW * 4, 0, 0);
          }
        }
      }
    }
  };
  unsigned hcode = 0;  // halo lanes: code byte(s) and slot of their halo voxel (pair) in the plane staged next
  int hslot = 0;
  uint2 hrec_s = make_uint2(0u, 0u);  // SEG: the record of the row above / below the tile as requested (waves 0, 1)
  auto halo_codes_issue = [&](int p, bool defer) {
    hcode = 0; hslot = 0;
    hrec_s = make_uint2(0u, 0u);
#if OI_TIMING_KNOCK & 16
    if (defer) p = gs;
#endif
    if (p <= g.nzl + 1 && wu < 3) {
#if OI_TIMING_KNOCK & 16
      const bool run = false;
#else
      const bool run = defer;
#endif
      const __amdgpu_buffer_rsrc_t cb = code_buf(p, run), sb = slot_buf(p, run);
      if (wu < 2) hcode = __builtin_amdgcn_raw_buffer_load_b16(cb, hvox, 0, 0);
      else hcode = __builtin_amdgcn_raw_buffer_load_b8(cb, hvox, 0, 0);
      // SEG: the row above / below the tile has a record, read by the whole wave (the slot only matters where hcode != 0); the
      // lanes of wave 2 (columns left / right of the tile, a different row each) keep their 4-byte table entries
      if (!SEG || wu == 2) {
        hslot = (int)__builtin_amdgcn_raw_buffer_load_b32(sb, hlin, 0, 0);
      } else {
        const int hr_ = min(max(hrow, 0), g.ny - 1);
        hrec_s = seg_row(p, hr_, run);
        if (!defer) hslot = slot_own(hrec_s);
      }
    }
  };
  typedef typename PBuf<PT>::Pair PPair;
  struct Raw { double2 r[R], q[R], xv[R]; PPair p[R], pold[R]; int off[R], xoff[R]; float c[R]; };
  struct HRaw { double2 r, q; PPair p; typename PBuf<PT>::One p1; float c; };
  // FORM 3: the fp32 coarse correction of the aggregate that holds a voxel pair (x pairs share it: i is even); the offset inside
  // the coarse plane does not change along the march, the plane comes with the descriptor
  // (periodic instantiations: the planes beyond the two ends are images whose parents sit in the coarse GHOST planes; an odd nz does
  // not give that by the arithmetic -- k_mg_fine's rule)
  auto cplane_of = [&](int gz) {
    const int kg = g.z0 + gz - 1;
    if ((VAR & 8) && kg >= g.nz) return C.cnzl + 1;
    if ((VAR & 8) && kg < 0) return 0;
    return (kg >> 1) - C.z0c + 1;
  };
  auto corr_buf = [&](const float* a, int gz) { return buf_of(a + C.csz * cplane_of(gz), 4ll * C.csz); };
  auto corr_ld = [&](int gz, int off) {
    float cv = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(corr_buf(C.xc, gz), off, 0, 0));
    if (C.xc2) cv += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(corr_buf(C.xc2, gz), off, 0, 0));
    return cv;
  };
  const int hcoff = yhalo ? ((i >> 1) + C.cnx * (hrow >> 1)) * 4 : (xhalo ? ((xh_i >> 1) + C.cnx * (xh_row >> 1)) * 4 : kOob);

  // ---- issue every load of plane p: r, q, p of the tile rows; x and p_{k-2} where x is updated ----
  auto plane_issue = [&](int p, const unsigned short (&cn)[R], const int (&ln)[R], long long pfa, long long pfb, Raw& raw, int mode, const double* xpl) {
    const long long nb = (long long)(8 * kLV) * (pfb - pfa);
    const __amdgpu_buffer_rsrc_t rb = buf_of(r_in + kLV * pfa, nb), qb = buf_of(q_in + kLV * pfa, nb);
    const __amdgpu_buffer_rsrc_t pb = buf_of(p_in + kLV * pfa, (long long)PBuf<PT>::kLineBytes * (pfb - pfa));
    const bool with_x = mode == 1 && do_x;
#pragma unroll
    for (int rr_ = 0; rr_ < R; ++rr_) {
      const bool st = in_x && (fullline ? ln[rr_] >= 0 : cn[rr_] != 0);  // (ln = -1, cn = 0 on rows outside the volume)
#if OI_TIMING_KNOCK & 8  // TIMING BUILDS ONLY: no vector traffic at all (every offset out of range), the instructions stay
      raw.off[rr_] = (st && ln[rr_] == -12345) ? (ln[rr_] - (int)pfa) * (8 * kLV) + sub : kOob;
#else
      raw.off[rr_] = st ? (ln[rr_] - (int)pfa) * (8 * kLV) + sub : kOob;
#endif
      // LDAUX & 32: the tile's INTERIOR rows arrive nontemporally (no neighbouring tile reads them as its halo rows); the
      // first and last row keep the default policy.  (wave-uniform choice between two instructions: the policy is an immediate)
      const bool inner = (LDAUX & 32) && !((wu == 0 && rr_ == 0) || (wu == NW - 1 && rr_ == R - 1));
      if (inner) {
        raw.r[rr_] = bld2<2>(rb, raw.off[rr_]);
        if (FORM == 0 || FORM == 2) raw.q[rr_] = bld2<2>(qb, raw.off[rr_]);
        else raw.q[rr_] = make_double2(0.0, 0.0);
        if (FORM < 2 || (FORM == 3 && with_x)) raw.p[rr_] = PBuf<PT>::template ld_pair<2>(pb, raw.off[rr_]);
        else raw.p[rr_] = PPair();
      } else {
      raw.r[rr_] = bld2<(LDAUX & 1) ? 2 : 0>(rb, raw.off[rr_]);
      if (FORM == 0 || FORM == 2) raw.q[rr_] = bld2<(LDAUX & 2) ? 2 : 0>(qb, raw.off[rr_]);
      else raw.q[rr_] = make_double2(0.0, 0.0);
      if (FORM < 2 || (FORM == 3 && with_x)) raw.p[rr_] = PBuf<PT>::template ld_pair<(LDAUX & 4) ? 2 : 0>(pb, raw.off[rr_]);
      else raw.p[rr_] = PPair();
      }
      raw.c[rr_] = 0.f;
      if (FORM == 3) raw.c[rr_] = corr_ld(p, (in_x && cn[rr_]) ? ((i >> 1) + C.cnx * (row_of(rr_) >> 1)) * 4 : kOob);
    }
    if (with_x) {
      const __amdgpu_buffer_rsrc_t xb = x_buf(xpl);
      const __amdgpu_buffer_rsrc_t ob = buf_of(p_out + kLV * pfa, (long long)PBuf<PT>::kLineBytes * (pfb - pfa));
#pragma unroll
      for (int rr_ = 0; rr_ < R; ++rr_) {
#ifdef OI_SOFFSET_TEST  // TEST BUILD ONLY (tools/soffset_experiment.sh): the row offset of the x accesses as the instruction's scalar
        // offset, the form that read x from wrong rows in round 2 (profiles/r02/k_cg1b_soffset_hazard.txt); 2 = with wait states
        raw.xoff[rr_] = raw.off[rr_] == kOob ? kOob : ivox8;
        raw.xv[rr_] = __builtin_bit_cast(double2, __builtin_amdgcn_raw_buffer_load_b128(xb, raw.xoff[rr_], row_of(rr_) * g.pitch * 8, 0));
#if OI_SOFFSET_TEST == 2
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#endif
#else
        raw.xoff[rr_] = raw.off[rr_] == kOob ? kOob : ivox8 + row_of(rr_) * g.pitch * 8;
        raw.xv[rr_] = __builtin_bit_cast(double2, __builtin_amdgcn_raw_buffer_load_b128(xb, raw.xoff[rr_], 0, (LDAUX & 8) ? 2 : 0));
#endif
        if (xskip) raw.pold[rr_] = PBuf<PT>::template ld_pair<(LDAUX & 16) ? 2 : 0>(ob, raw.off[rr_]);  // p_{k-2}
      }
    }
  };
  // mode 0: a halo plane owned by another block (nothing stored); 1: an output plane of this block (r_k, p_k stored, x
  // updated); 2: a ghost plane of the slab, kept current by the block that marches over it (r_k, p_k)
  auto plane_finish = [&](int p, const unsigned short (&cn)[R], long long pfa, long long pfb, const Raw& raw, double2 (&pn)[R],
                          double2 (&rn)[R], int mode, const double* xpl) {
    const long long nb = (long long)(8 * kLV) * (pfb - pfa);
    const __amdgpu_buffer_rsrc_t rob = buf_of(r_out + kLV * pfa, nb);
    const __amdgpu_buffer_rsrc_t pob = buf_of(p_out + kLV * pfa, (long long)PBuf<PT>::kLineBytes * (pfb - pfa));
    const __amdgpu_buffer_rsrc_t xb = x_buf(xpl);
    const bool with_x = mode == 1 && do_x;
#pragma unroll
    for (int rr_ = 0; rr_ < R; ++rr_) {
      // (a lane without a stored line read zeros: r0 = r1 = 0, p = 0, exactly what the branch of k_cg1 leaves there)
      const double2 rv = raw.r[rr_], qv = raw.q[rr_], pin = PBuf<PT>::wide(raw.p[rr_]);
      const double r0 = (FORM == 0 || FORM == 2) ? rv.x - alpha_prev * qv.x : rv.x, r1 = (FORM == 0 || FORM == 2) ? rv.y - alpha_prev * qv.y : rv.y;
      rn[rr_] = make_double2(r0, r1);
      if (FORM == 3) {  // what is staged is v = omega b / c + s * (coarse correction of the pair's aggregate)   (k_mg_fine, MODE 2)
        const unsigned short cc = cn[rr_];
        double v0 = r0 * wtab[cc & 0xffu], v1 = r1 * wtab[cc >> 8];
        if (cc) {
          const double cr = scale * (double)raw.c[rr_];
          if (cc & 0xffu) v0 += cr;
          if (cc >> 8) v1 += cr;
        }
        pn[rr_] = make_double2(v0, v1);
        if (with_x && mode != 0) {  // x_k = x_{k-1} + alpha_{k-1} p_{k-1}, the update the first kernel of this iteration leaves out
          double x0 = raw.xv[rr_].x, x1 = raw.xv[rr_].y;
          if (xskip) {  // (pairwise: x_{k-2} + alpha_{k-2} p_{k-2} first)
            const double2 pold = PBuf<PT>::wide(raw.pold[rr_]);
            x0 = x0 + alpha_prev2 * pold.x; x1 = x1 + alpha_prev2 * pold.y;
          }
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, make_double2(x0 + alpha_prev * pin.x, x1 + alpha_prev * pin.y)),
                                                 xb, raw.xoff[rr_], 0, (STAUX & 8) ? 2 : 0);
        }
        continue;
      }
      if (FORM == 2) {  // what is staged is the first damped-Jacobi iterate v = omega r / c of the new residual
        const unsigned short cc = cn[rr_];
        pn[rr_] = make_double2(r0 * wtab[cc & 0xffu], r1 * wtab[cc >> 8]);
        if (mode != 0) bst2<(STAUX & 1) ? 2 : 0>(rob, raw.off[rr_], r0, r1);
        continue;
      }
      double z0 = r0, z1 = r1;
      if (JACOBI) {
        const unsigned short cc = cn[rr_];
        if (cc & 0xff) z0 = div_small(r0, jtab[cc & 0xff]);
        if (cc >> 8) z1 = div_small(r1, jtab[cc >> 8]);
      }
      pn[rr_] = make_double2(PIo<PT>::rnd(z0 + beta * pin.x), PIo<PT>::rnd(z1 + beta * pin.y));
      if (mode != 0) {
        if (FORM == 0) bst2<(STAUX & 1) ? 2 : 0>(rob, raw.off[rr_], r0, r1);
        if (with_x) {
          const double2 xv = raw.xv[rr_];
#ifdef OI_DEBUG_X
          if (raw.xoff[rr_] != kOob) {
            const double2 xv2 = ld2(x + (long long)p * g.plane + (long long)row_of(rr_) * g.pitch + i);
            if (xv.x != xv2.x || xv.y != xv2.y)
              printf("x mismatch k %d plane %d row %d i %d: buffer %g %g plain %g %g  xoff %d soff %d\n", k, p, row_of(rr_), i, xv.x, xv.y,
                     xv2.x, xv2.y, raw.xoff[rr_], row_of(rr_) * g.pitch * 8);
          }
#endif
          double a0 = alpha_prev * pin.x, a1 = alpha_prev * pin.y;
          if (xskip) {
            const double2 pold = PBuf<PT>::wide(raw.pold[rr_]);
            a0 = alpha_prev2 * pold.x + a0; a1 = alpha_prev2 * pold.y + a1;
          }
#ifdef OI_SOFFSET_TEST
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, make_double2(xv.x + a0, xv.y + a1)), xb,
                                                 raw.xoff[rr_], row_of(rr_) * g.pitch * 8, 0);
#if OI_SOFFSET_TEST == 2
          asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#endif
#else
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, make_double2(xv.x + a0, xv.y + a1)), xb,
                                                 raw.xoff[rr_], 0, (STAUX & 8) ? 2 : 0);
#endif
        }
        PBuf<PT>::template st_pair<(STAUX & 4) ? 2 : 0>(pob, raw.off[rr_], pn[rr_].x, pn[rr_].y);
      }
    }
  };
  // ---- the halo of plane p (its code bytes and slots came an iteration ahead) ----
#if OI_TIMING_KNOCK & 8
  auto halo_offset = [&](long long pfa) { return (is_halo && hcode && hslot == -12345) ? (hslot - (int)pfa) * (8 * kLV) + hsub : kOob; };
#elif defined(OI_TIMING_NOHALO)  // TIMING BUILDS ONLY (wrong results; tools/variant_build.sh): what the halo's vector reads cost. 1: rows, 2: columns
  auto halo_offset = [&](long long pfa) {
    if (((OI_TIMING_NOHALO & 1) && wu < 2) || ((OI_TIMING_NOHALO & 2) && wu == 2)) return kOob;
    return (is_halo && hcode) ? (hslot - (int)pfa) * (8 * kLV) + hsub : kOob;
  };
#else
  // (VAR & 2 / VAR & 4, EXPERIMENT instantiations with wrong results: what the vector reads of the halo rows / columns cost)
  auto halo_offset = [&](long long pfa) {
    if (((VAR & 2) && wu < 2) || ((VAR & 4) && wu == 2)) return kOob;
    return (is_halo && hcode) ? (hslot - (int)pfa) * (8 * kLV) + hsub : kOob;
  };
#endif
  auto halo_issue = [&](int hp, long long pfa, long long pfb, int off, HRaw& h) {
    h.r = h.q = make_double2(0.0, 0.0);
    h.p = PPair();
    h.p1 = 0;
    h.c = 0.f;
    if (wu < 3) {
      const long long nb = (long long)(8 * kLV) * (pfb - pfa);
      const __amdgpu_buffer_rsrc_t rb = buf_of(r_in + kLV * pfa, nb), qb = buf_of(q_in + kLV * pfa, nb);
      const __amdgpu_buffer_rsrc_t pb = buf_of(p_in + kLV * pfa, (long long)PBuf<PT>::kLineBytes * (pfb - pfa));
      if (wu < 2) {
        h.r = bld2<0>(rb, off);
        if (FORM < 2) h.p = PBuf<PT>::template ld_pair<0>(pb, off);
        if (FORM == 0 || FORM == 2) h.q = bld2<0>(qb, off);
      } else {
        h.r.x = bld1(rb, off);
        if (FORM < 2) h.p1 = PBuf<PT>::ld_one(pb, off);
        if (FORM == 0 || FORM == 2) h.q.x = bld1(qb, off);
      }
      if (FORM == 3) h.c = corr_ld(hp, off == kOob ? kOob : hcoff);
    }
  };
  auto stage_rows = [&](double* T, const double2 (&pn)[R]) {
#pragma unroll
    for (int rr_ = 0; rr_ < R; ++rr_) {
      if (per && (i == g.nx || row_of(rr_) == g.ny)) continue;  // (that slot holds the wrapped column / row)
      // odd nx, periodic: the pair (nx - 1, pad) straddles the seam -- its second slot holds the wrapped column 0 (wave 2), which the
      // stencil then reads as this pair's second value: the +x neighbour of voxel nx - 1 (the pad voxel itself has code 0)
      if (per && i + 1 == g.nx) T[(wu + rr_ * NW + 1) * kLP + 2 + 2 * lane] = pn[rr_].x;
      else *reinterpret_cast<double2*>(&T[(wu + rr_ * NW + 1) * kLP + 2 + 2 * lane]) = pn[rr_];
    }
  };
  auto stage_halo = [&](double* T, const HRaw& hr) {
    // (the LDS addresses of the halo are formed from an opaque copy of the lane index: hoisted out of the loop they were
    // spilled to scratch at the 128-VGPR budget, and a scratch reload waits for every store the wave has in flight)
    int lane = threadIdx.x & 63;
    asm volatile("" : "+v"(lane));
    if (wu < 2) {
      double2 h = make_double2(0.0, 0.0);
      if (hcode) {
        double z0 = (FORM == 0 || FORM == 2) ? hr.r.x - alpha_prev * hr.q.x : hr.r.x, z1 = (FORM == 0 || FORM == 2) ? hr.r.y - alpha_prev * hr.q.y : hr.r.y;
        if (FORM == 2) {
          if (hcode & 0xff) h.x = z0 * wtab[hcode & 0xff];
          if (hcode >> 8) h.y = z1 * wtab[hcode >> 8];
        }
        if (FORM == 3) {
          const double cr = scale * (double)hr.c;
          if (hcode & 0xff) h.x = z0 * wtab[hcode & 0xff] + cr;
          if (hcode >> 8) h.y = z1 * wtab[hcode >> 8] + cr;
        }
        if (JACOBI) {
          if (hcode & 0xff) z0 = div_small(z0, jtab[hcode & 0xff]);
          if (hcode >> 8) z1 = div_small(z1, jtab[hcode >> 8]);
        }
        if (FORM < 2) {
          const double2 hp = PBuf<PT>::wide(hr.p);
          h.x = PIo<PT>::rnd(z0 + beta * hp.x); h.y = PIo<PT>::rnd(z1 + beta * hp.y);
        }
      }
      *reinterpret_cast<double2*>(&T[(per ? hlds : ((wu == 0) ? 0 : BY + 1)) * kLP + 2 + 2 * lane]) = h;
    } else if (wu == 2 && lane < 2 * BY) {
      double h = 0.0;
      if (hcode) {
        double z = (FORM == 0 || FORM == 2) ? hr.r.x - alpha_prev * hr.q.x : hr.r.x;
        if (JACOBI) z = div_small(z, jtab[hcode & 7]);
        if (FORM == 2) h = z * wtab[hcode];
        else if (FORM == 3) { h = z * wtab[hcode]; h += scale * (double)hr.c; }
        else h = PIo<PT>::rnd(z + beta * PBuf<PT>::wide1(hr.p1));
      }
      T[((lane >> 1) + 1) * kLP + ((lane & 1) ? (per ? xcol : 2 + kBX) : 1)] = h;
    }
  };

  double2 pn_nxt[R], rn_cur[R], rn_nxt[R];
  unsigned short cc_cur[R], cc_nxt[R], cc_nn[R];
  int lo_cur[R], lo_nxt[R], lo_nn[R];
  // first slots of planes gz - 1 .. gz + 2 (scalars, a plane ahead of their use)
  auto pf_at = [&](int p) { return plane_first[min(p, g.nzl + 2)]; };
  long long pfm = pf_at(gs - 1), pf0 = pf_at(gs), pf1 = pf_at(gs + 1), pf2 = pf_at(gs + 2);
  int b_prev = 0, b_cur = TS, b_nxt = 2 * TS;  // ring positions (doubles) of planes gz - 1, gz, gz + 1
  // OI_CG1B_EARLY: the vectors of plane gz + 2 (and its halo) are requested at the END of iteration gz, before the barrier, not
  // at the top of iteration gz + 1: they are in flight while the block waits at the barrier and while the next iteration
  // requests its code bytes and records -- the one memory round trip of a plane step starts that much earlier.  The raw
  // values live across the barrier; the peak register demand (the stencil of plane gz) lies before the requests.
  // (FORM 0 has no registers to spare for it: 86 spills at the 128-register budget; FORM 1 / 2 take it with 109 / 111)
  constexpr bool kEarly = (OI_CG1B_EARLY != 0) && FORM != 0;
  Raw raw_e;   // (kEarly only: the requests of a plane outlive the iteration that makes them; FORM 0 keeps iteration-local
  HRaw hr_e;   // objects, whose registers are dead at the barrier)
  {
    Raw raw_p;
    HRaw hr_p;
    Raw& raw = raw_p;
    HRaw& hr = hr_p;
    halo_codes_issue(gs, false);
    codes_issue(gs - 1, cc_nn, lo_nn, false);
    codes_issue(gs, cc_cur, lo_cur, false);
    codes_issue(gs + 1, cc_nxt, lo_nxt, false);
    plane_issue(gs - 1, cc_nn, lo_nn, pfm, pf0, raw, gs == 1 ? 2 : 0, x_plane(gs - 1));
    plane_finish(gs - 1, cc_nn, pfm, pf0, raw, pn_nxt, rn_nxt, gs == 1 ? 2 : 0, x_plane(gs - 1));
    stage_rows(tile + b_prev, pn_nxt);
    const int hoff = halo_offset(pf0);
    plane_issue(gs, cc_cur, lo_cur, pf0, pf1, raw, 1, x_plane(gs));
    halo_issue(gs, pf0, pf1, hoff, hr);
    plane_finish(gs, cc_cur, pf0, pf1, raw, pn_nxt, rn_cur, 1, x_plane(gs));
    stage_rows(tile + b_cur, pn_nxt);
    stage_halo(tile + b_cur, hr);
    halo_codes_issue(gs + 1, false);
    if (kEarly) {  // plane gs + 1 for the first iteration
      const int mode1 = gs + 1 < ge ? 1 : (ge == g.nzl + 1 ? 2 : 0);
      const int hoff1 = halo_offset(pf1);
      plane_issue(gs + 1, cc_nxt, lo_nxt, pf1, pf2, raw_e, mode1, x_plane(gs + 1));
      if (gs + 1 < ge) halo_issue(gs + 1, pf1, pf2, hoff1, hr_e);
    }
  }
  __syncthreads();
  code_run = code + (long long)(gs + 2) * g.plane;
  if (SEG) seg_run = seg + (long long)(gs + 2) * seg_plane + tx;
  slot_run = reinterpret_cast<const char*>(lineoff) + (long long)(gs + 2) * plane_lines4;
  const double* x_run = x_plane(gs + 1);  // -> plane gz + 1 while the loop runs

  // FORM 2: even waves own the 2x2x2 block sums of the pre-smoothing residual; `own` = their x-pair sum of the plane just
  // finished, `acc` the running 2 x 2 x (1 or 2) sum; plane gzp is consumed one barrier after it was computed (k_mg_fine, MODE 3)
  double own[R], acc[R];
#pragma unroll
  for (int rr_ = 0; rr_ < R; ++rr_) { own[rr_] = 0.0; acc[rr_] = 0.0; }
  auto consume = [&](int gzp) {
    if (FORM != 2 || (wu & 1)) return;
    int lane = threadIdx.x & 63;
    asm volatile("" : "+v"(lane));
    const double* Y = ysum[FORM == 2 ? (gzp & 1) : 0];
    const int kg = g.z0 + gzp - 1;  // global fine plane
    const bool store = (kg & 1) != 0 || gzp == ge - 1;
#pragma unroll
    for (int rr_ = 0; rr_ < R; ++rr_) {
      acc[rr_] += own[rr_] + Y[(rr_ * (NW / 2) + (wu >> 1)) * 64 + lane];
      if (store) {
        const int I = tx * (kBX / 2) + lane, J = (j0 + wu + rr_ * NW) >> 1;
        if (I < C.cnx && J < C.cny) bcoarse[(long long)I + (long long)C.cnx * J + C.csz * ((kg >> 1) - C.z0c + 1)] = (float)acc[rr_];
        acc[rr_] = 0.0;
      }
    }
  };
  double s_rr = 0.0, s_rz = 0.0, s_pq = 0.0, s_zq = 0.0, s_qdq = 0.0;
  // q = A p of plane gz (FORM 2 / 3: the smoother's stencil) from the LDS ring and the next plane's registers, its stores and sums
  auto stencil = [&](const int gz) {
    const double* T = tile + b_cur;
    const double* Tp = tile + b_prev;
    const long long nbq = (long long)(8 * kLV) * (pf1 - pf0);
    // (FORM 3: what is written is z, through r_out)
    const __amdgpu_buffer_rsrc_t qob = buf_of(FORM == 2 ? (double*)nullptr : (FORM == 3 ? r_out : q_out) + kLV * pf0, FORM == 2 ? 0 : nbq);
    double xsum[R];
#pragma unroll
    for (int rr_ = 0; rr_ < R; ++rr_) {
      const unsigned short cc = cc_cur[rr_];
      const bool stored = in_x && lo_cur[rr_] >= 0;
#if OI_TIMING_KNOCK & 8
      const int qoff = lo_cur[rr_] == -12345 ? (lo_cur[rr_] - (int)pf0) * (8 * kLV) + sub : kOob;
#else
      const int qoff = (lo_cur[rr_] - (int)pf0) * (8 * kLV) + sub;
#endif
      xsum[rr_] = 0.0;
      if (FORM != 2 && fullline && !cc && stored) bst2<(STAUX & 2) ? 2 : 0>(qob, qoff, 0.0, 0.0);  // whole lines
      if (cc) {  // q and every partial sum vanish where both voxels are non-free
        const int base = (wu + rr_ * NW + 1) * kLP + 2 + 2 * lane;
#if OI_TIMING_KNOCK & 2  // TIMING BUILDS ONLY (wrong results): the stencil without its LDS reads
        const double2 lf = pn_nxt[rr_], pc = rn_cur[rr_], rt = pn_nxt[rr_], ym = rn_cur[rr_], yp = pn_nxt[rr_], zm = rn_cur[rr_];
        (void)T; (void)Tp; (void)base;
#else
        const double2 lf = *reinterpret_cast<const double2*>(&T[base - 2]);
        const double2 pc = *reinterpret_cast<const double2*>(&T[base]);
        const double2 rt = *reinterpret_cast<const double2*>(&T[base + 2]);
        const double2 ym = *reinterpret_cast<const double2*>(&T[base - kLP]);
        const double2 yp = *reinterpret_cast<const double2*>(&T[base + kLP]);
        const double2 zm = *reinterpret_cast<const double2*>(&Tp[base]);
#endif
        const unsigned c0 = cc & 0xffu, c1 = cc >> 8;
        const double2 rc = rn_cur[rr_];
        const double s0 = ((((lf.y + pc.y) + ym.x) + yp.x) + zm.x) + pn_nxt[rr_].x;
        const double s1 = ((((pc.x + rt.x) + ym.y) + yp.y) + zm.y) + pn_nxt[rr_].y;
        const double q0 = c0 ? (double)c0 * pc.x - s0 : 0.0;
        const double q1 = c1 ? (double)c1 * pc.y - s1 : 0.0;
        if (FORM == 2) {  // residual of the first Jacobi iterate: r - A v; its x-pair sum goes to the restriction
          const double o0 = c0 ? rc.x - q0 : 0.0, o1 = c1 ? rc.y - q1 : 0.0;
          xsum[rr_] = o0 + o1;
          s_rr += rc.x * rc.x; s_rr += rc.y * rc.y;
          continue;
        }
        if (FORM == 3) {  // z = v + omega (b - A v) / c and the partial sum of z.b
          const double o0 = c0 ? pc.x + (rc.x - q0) * wtab[c0] : 0.0, o1 = c1 ? pc.y + (rc.y - q1) * wtab[c1] : 0.0;
          bst2<(STAUX & 2) ? 2 : 0>(qob, qoff, o0, o1);
          s_pq += o0 * rc.x; s_pq += o1 * rc.y;
          continue;
        }
        bst2<(STAUX & 2) ? 2 : 0>(qob, qoff, q0, q1);
        double z0 = rc.x, z1 = rc.y, d0 = q0, d1 = q1;
        if (JACOBI) {
          if (c0) { const double2 yc = jtab[c0]; z0 = div_small(rc.x, yc); d0 = div_small(q0, yc); }
          if (c1) { const double2 yc = jtab[c1]; z1 = div_small(rc.y, yc); d1 = div_small(q1, yc); }
        }
        s_pq += pc.x * q0; s_pq += pc.y * q1;
        if (FORM == 0) {
          s_rr += rc.x * rc.x; s_rr += rc.y * rc.y;
          if (JACOBI) { s_rz += rc.x * z0; s_rz += rc.y * z1; }  // (without Jacobi z = r: the same products in the same order as s_rr)
          s_zq += z0 * q0; s_zq += z1 * q1;
          s_qdq += q0 * d0; s_qdq += q1 * d1;
        }
      }
    }
    if (FORM == 2) {  // odd waves publish their x-pair sums, even waves keep theirs until the barrier has passed
      int lane = threadIdx.x & 63;
      asm volatile("" : "+v"(lane));
      double* Y = ysum[FORM == 2 ? (gz & 1) : 0];
#pragma unroll
      for (int rr_ = 0; rr_ < R; ++rr_) {
        if (wu & 1) Y[(rr_ * (NW / 2) + (wu >> 1)) * 64 + lane] = xsum[rr_];
        else own[rr_] = xsum[rr_];
      }
    }
  };
  auto step = [&](const int gz, Raw& raw, HRaw& hr) {
    // pf0 .. pf2 = first slots of planes gz, gz + 1, gz + 2
    const long long pf3 = pf_at(gz + 3);
    const int mode = gz + 1 < ge ? 1 : (ge == g.nzl + 1 ? 2 : 0);
    const bool more = gz + 1 < ge;
    if (!kEarly) {
      // every load of this iteration, back to back: plane gz + 1's vectors and halo, plane gz + 2's codes and slots
      const int hoff = halo_offset(pf1);  // (first: the slot came with the previous iteration's loads, nothing newer is waited for)
      __builtin_amdgcn_sched_barrier(0);
      plane_issue(gz + 1, cc_nxt, lo_nxt, pf1, pf2, raw, mode, x_run);
      if (more) halo_issue(gz + 1, pf1, pf2, hoff, hr);
    } else {
      __builtin_amdgcn_sched_barrier(0);  // (plane gz + 1's vectors and halo were requested before the barrier)
    }
    const unsigned hcode_now = hcode;
    codes_issue(gz + 2, cc_nn, lo_nn, true);
    if (more) halo_codes_issue(gz + 2, true);
    const unsigned hcode_next = hcode;
    const int hslot_next = hslot;
    __builtin_amdgcn_sched_barrier(0);
    plane_finish(gz + 1, cc_nxt, pf1, pf2, raw, pn_nxt, rn_nxt, mode, x_run);
#if OI_TIMING_KNOCK & 4  // TIMING BUILDS ONLY: nothing staged in LDS
    if (more && hr.r.x == 1.2345e300) {
#else
    if (more) {
#endif
      stage_rows(tile + b_nxt, pn_nxt);
      hcode = hcode_now;
      stage_halo(tile + b_nxt, hr);
    }
    hcode = hcode_next; hslot = hslot_next;
    if (FORM == 2 && gz > gs) consume(gz - 1);
    stencil(gz);
    if (SEG) {  // the records requested at the top of this iteration have long arrived: their slots
#pragma unroll
      for (int rr_ = 0; rr_ < R; ++rr_) lo_nn[rr_] = slot_own(rec_nn[rr_]);
      if (more && wu < 2) hslot = slot_own(hrec_s);
    }
#pragma unroll
    for (int rr_ = 0; rr_ < R; ++rr_) {
      rn_cur[rr_] = rn_nxt[rr_];
      cc_cur[rr_] = cc_nxt[rr_]; cc_nxt[rr_] = cc_nn[rr_];
      lo_cur[rr_] = lo_nxt[rr_]; lo_nxt[rr_] = lo_nn[rr_];
    }
    pf0 = pf1; pf1 = pf2; pf2 = pf3;
    code_run += g.plane; slot_run += plane_lines4; x_run += g.plane;
    if (SEG) seg_run += seg_plane;
    { const int t = b_prev; b_prev = b_cur; b_cur = b_nxt; b_nxt = t; }
    if (kEarly && gz + 1 < ge) {  // the next iteration's plane (gz + 2): its code bytes and slots have arrived and been rotated into *_nxt
      const int mode2 = gz + 2 < ge ? 1 : (ge == g.nzl + 1 ? 2 : 0);
      const int hoff2 = halo_offset(pf1);
      __builtin_amdgcn_sched_barrier(0);
      plane_issue(gz + 2, cc_nxt, lo_nxt, pf1, pf2, raw, mode2, x_run);  // (x_run has moved on to plane gz + 2)
      if (gz + 2 < ge) halo_issue(gz + 2, pf1, pf2, hoff2, hr);
      __builtin_amdgcn_sched_barrier(0);
    }
#if !(OI_TIMING_KNOCK & 1)  // (TIMING BUILDS ONLY: 1 = no barrier per plane)
    __syncthreads();
#endif
  };
  if (kEarly) {
    for (int gz = gs; gz < ge; ++gz) step(gz, raw_e, hr_e);
  } else {
    for (int gz = gs; gz < ge; ++gz) {
      Raw raw;
      HRaw hr;
      step(gz, raw, hr);
    }
  }
  if (FORM == 1 || FORM == 3) {
    const double t = block_sum(s_pq, red);
    if (threadIdx.x == 0) partials[(long long)partial_base + blockIdx.x] = t;
    return;
  }
  if (FORM == 2) {  // {r.r, r.z} with z = r, the two sums launch_update_r leaves
    consume(ge - 1);
    const double t = block_sum(s_rr, red);
    if (threadIdx.x == 0) { partials[2 * ((long long)partial_base + blockIdx.x)] = t; partials[2 * ((long long)partial_base + blockIdx.x) + 1] = t; }
    return;
  }
  const double t0 = block_sum(s_rr, red);
  const double t1 = JACOBI ? block_sum(s_rz, red) : t0;
  const double t2 = block_sum(s_pq, red);
  const double t3 = block_sum(s_zq, red);
  const double t4 = block_sum(s_qdq, red);
  if (threadIdx.x == 0) {
    double* o = partials + 5 * ((long long)partial_base + blockIdx.x);
    o[0] = t0; o[1] = t1; o[2] = t2; o[3] = t3; o[4] = t4;
  }
}


// k_cg1 tile shapes (Tune::cg1_rows x cg1_waves): 8 rows = 8 waves x 1 row per thread, 16 = 8 x 2, 32 = 16 waves x 2
// rows (the 8-wave x 4-row form needs 194 VGPRs and fits only one block per CU; tile_rows 33 selects it)
void tune_set_cg1_tile(Tune& t, int tile_rows) {
  t.cg1_auto = tile_rows <= 0;
  if (tile_rows == 8) { t.cg1_rows = 1; t.cg1_waves = 8; }
  else if (tile_rows == 16) { t.cg1_rows = 2; t.cg1_waves = 8; }
  else if (tile_rows == 32) { t.cg1_rows = 2; t.cg1_waves = 16; }
  else if (tile_rows == 33) { t.cg1_rows = 4; t.cg1_waves = 8; }
}

template <int R, int NW, typename PT, bool JAC>
static void launch_cg1b_k(int blocks, const Geom& g, const uint8_t* code, double* x, const double* r_in, double* r_out,
                          const double* q_in, double* q_out, const void* p_in, void* p_out, double* rec, int k,
                          int zchunk, int tiles_x, int tiles_y, int nzc, int part, int partial_base, double* partials,
                          Ctrl* ctrl, const int* lineoff, const long long* plane_first, const void* seg, hipStream_t s) {
#ifdef OI_EXP_CACHEPOL  // EXPERIMENT BUILDS: cg1_nt selects a (load, store) cache-policy pair of the default shape
  if constexpr (R == 2 && NW == 8 && !JAC && std::is_same<PT, double>::value) {
    if (seg && g.tune.cg1_seg && g.tune.cg1_nt >= 2) {
#define OI_POL(ID, LD, ST, VR)                                                                                                    \
  if (g.tune.cg1_nt == ID) {                                                                                                   \
    hipLaunchKernelGGL((k_cg1b<R, NW, PT, JAC, true, 0, ST, LD, VR>), dim3(blocks), dim3(64 * NW), 0, s, g, code, x, r_in, r_out, q_in, q_out, \
                       (const PT*)p_in, (PT*)p_out, rec, k, g.tune.cg1_xskip, zchunk, tiles_x, tiles_y, nzc, part, partial_base, \
                       partials, ctrl, lineoff, plane_first, (const uint2*)seg, (g.pitch + kLV - 1) / kLV, g.tune.cg1_fullline, \
                       (float*)nullptr, MgCoarse{}, 0.0, 0.0);                                                                 \
    return;                                                                                                                    \
  }
      OI_POL(2, 24, 13, 0) OI_POL(3, 24, 13, 1) OI_POL(4, 24, 13, 2) OI_POL(5, 24, 13, 4) OI_POL(6, 24, 13, 6)
      OI_POL(7, 24 + 1, 13, 0) OI_POL(8, 24 + 2, 13, 0) OI_POL(9, 24 + 4, 13, 0)
#undef OI_POL
    }
  }
#endif
  if (seg && g.tune.cg1_seg && g.tune.cg1_nt)
    hipLaunchKernelGGL((k_cg1b<R, NW, PT, JAC, true, 0, kNtSt0, kNtLd0>), dim3(blocks), dim3(64 * NW), 0, s, g, code, x, r_in, r_out, q_in, q_out,
                       (const PT*)p_in, (PT*)p_out, rec, k, g.tune.cg1_xskip, zchunk, tiles_x, tiles_y, nzc, part, partial_base,
                       partials, ctrl, lineoff, plane_first, (const uint2*)seg, (g.pitch + kLV - 1) / kLV, g.tune.cg1_fullline,
                       (float*)nullptr, MgCoarse{}, 0.0, 0.0);
  else if (seg && g.tune.cg1_seg)
    hipLaunchKernelGGL((k_cg1b<R, NW, PT, JAC, true>), dim3(blocks), dim3(64 * NW), 0, s, g, code, x, r_in, r_out, q_in, q_out,
                       (const PT*)p_in, (PT*)p_out, rec, k, g.tune.cg1_xskip, zchunk, tiles_x, tiles_y, nzc, part, partial_base,
                       partials, ctrl, lineoff, plane_first, (const uint2*)seg, (g.pitch + kLV - 1) / kLV, g.tune.cg1_fullline,
                       (float*)nullptr, MgCoarse{}, 0.0, 0.0);
  else
    hipLaunchKernelGGL((k_cg1b<R, NW, PT, JAC, false>), dim3(blocks), dim3(64 * NW), 0, s, g, code, x, r_in, r_out, q_in, q_out,
                       (const PT*)p_in, (PT*)p_out, rec, k, g.tune.cg1_xskip, zchunk, tiles_x, tiles_y, nzc, part, partial_base,
                       partials, ctrl, lineoff, plane_first, (const uint2*)nullptr, (g.pitch + kLV - 1) / kLV, g.tune.cg1_fullline,
                       (float*)nullptr, MgCoarse{}, 0.0, 0.0);
}
template <int R, int NW, typename PT>
static void launch_cg1_r(int blocks, const Geom& g, const uint8_t* code, double* x, const double* r_in, double* r_out,
                         const double* q_in, double* q_out, const void* p_in, void* p_out, int jacobi,
                         double* rec, int k, int zchunk, int tiles_x, int tiles_y, int nzc, int part,
                         int partial_base, double* partials, Ctrl* ctrl, const int* lineoff, const long long* plane_first,
                         const void* seg, hipStream_t s) {
  // compacted layout: the descriptor-addressed kernel, unless a plane is too large for a 31-bit range or the tuning
  // asks for the first form (cross-checks)
  if (lineoff && plane_first && g.tune.cg1_kernel == 0 && 8ll * g.plane < kBufMax) {
    if (jacobi)
      launch_cg1b_k<R, NW, PT, true>(blocks, g, code, x, r_in, r_out, q_in, q_out, p_in, p_out, rec, k, zchunk, tiles_x, tiles_y,
                                     nzc, part, partial_base, partials, ctrl, lineoff, plane_first, seg, s);
    else
      launch_cg1b_k<R, NW, PT, false>(blocks, g, code, x, r_in, r_out, q_in, q_out, p_in, p_out, rec, k, zchunk, tiles_x, tiles_y,
                                      nzc, part, partial_base, partials, ctrl, lineoff, plane_first, seg, s);
    return;
  }
  // everything else is the branching first form (k_cg1): dense vectors, planes beyond a 31-bit range, the cross-check
  // option -- a superseded kernel, in the test library only (oi_k_xcheck.hip).  The host refuses such handles in the product.
  (void)launch_cg1_first_form(R, NW, blocks, g, code, x, r_in, r_out, q_in, q_out, p_in, p_out, std::is_same<PT, float>::value ? 1 : 0, jacobi, rec, k,
                              zchunk, tiles_x, tiles_y, nzc, part, partial_base, partials, ctrl, lineoff, s);
}


#if !OI_XCHECK
int launch_cg1_first_form(int, int, int, const Geom&, const uint8_t*, double*, const double*, double*, const double*, double*, const void*, void*, int,
                          int, double*, int, int, int, int, int, int, int, double*, Ctrl*, const int*, hipStream_t) {
  return -2;  // (the first form lives in the test library: oi_k_xcheck.hip)
}
#endif
bool xcheck_built() { return OI_XCHECK != 0; }

// Launch shape of the single-kernel iteration (host arithmetic only; oi_debug_cg1_grid exposes it to the CPU tests).
// blocks = -1: the grid would not fit the handle's partial-sum array (Tune::max_blocks).
Cg1Grid cg1_grid(const Geom& g, bool descr, int split, int part) {
  Cg1Grid o{};
  // automatic tile shape: the descriptor-addressed kernel (k_cg1b) takes 8-wave, 16-row tiles (two blocks per CU out of
  // step with each other) and 16-plane chunks (profiles/r02/k_cg1b_sweep.txt: 7.87 ms against 8.04 / 8.43 with 32 / 64
  // planes, 8.57 k_cg1); k_cg1 16-wave, 32-row tiles on large planes (profiles/r01/cg1_sweep.txt)
  o.R = g.tune.cg1_auto ? 2 : g.tune.cg1_rows;
  o.NW = g.tune.cg1_auto ? ((!descr && g.ny >= 256 && g.pitch >= 256) ? 16 : 8) : g.tune.cg1_waves;
  const int by = o.NW * o.R;
  o.tiles_x = (g.pitch + kBX - 1) / kBX;
  o.tiles_y = (g.ny + by - 1) / by;
  const long long cols = (long long)o.tiles_x * o.tiles_y;
  if (!split && g.tune.zchunk <= 0 && descr && cols * ((g.nzl + 15) / 16) >= 2048) {
    Geom gs_ = g;
    // sparse structures (a quarter of the lines stored: config 5's porosity 0.2) do better with long chunks -- little to
    // overlap per plane, and the chunk-end planes cost the same (profiles/r02/k_cg1b_sweep.txt: 46.8 vs 49.0 ms on
    // config 5's share, 4.94 vs 5.07 ms at 1024^3); the sphere pack (half of the lines) with 16 planes
    // (round 4, with the nontemporal policy -- the written vectors no longer push the chunk's own planes out of the L2 --:
    // 24 / 32 / 48 planes 6.05 ms against 6.11 with 16 and 6.13 with 64, same handle, alternating:
    // profiles/r04/k_cg1b_chunk_length_tile_map_blobs_with_nontemporal.txt; 24 because the slabs of the strong-scaling
    // runs prefer it to 32: 128 planes 0.933 vs 0.960 ms, 256: 1.622 vs 1.638, 512: 3.123 vs 3.140, slab_chunk_lengths.txt)
    gs_.tune.zchunk = (g.live_lines > 0.f && g.live_lines < 0.35f) ? 64 : (g.tune.cg1_nt ? 24 : 16);
    pick_chunks(gs_, cols, false, &o.zchunk, &o.nzc);
  } else if (split && g.tune.zchunk <= 0) {
    Geom gs_ = g;
    gs_.tune.zchunk = g.nzl / 4;  // two boundary chunks + at least two interior ones
    if (gs_.tune.zchunk > 64) gs_.tune.zchunk = 64;
    if (gs_.tune.zchunk < 16 || descr) gs_.tune.zchunk = 16;  // (k_cg1b: 16-plane chunks, and the planes leave sooner)
    pick_chunks(gs_, cols, false, &o.zchunk, &o.nzc);
  } else {
    pick_chunks(g, cols, false, &o.zchunk, &o.nzc);
  }
  if (!grid_fits(g, cols, o.nzc)) { o.blocks = -1; return o; }
  const int nsel = part == 0 ? o.nzc : (part == 1 ? (o.nzc < 2 ? o.nzc : 2) : o.nzc - 2);
  o.blocks = nsel <= 0 ? 0 : (int)(cols * nsel);
  return o;
}

// part: 0 = whole slab, 1 = boundary chunks, 2 = interior chunks (0 blocks when there are none).
// `split` selects the chunking of a slab with neighbours: >= 4 chunks so that the interior
// launch can hide the plane exchange.  Returns the number of blocks launched; they write
// partials [partial_base, partial_base + blocks).
int launch_cg1(const Geom& g, const uint8_t* code, double* x, const double* r_in, double* r_out, const double* q_in,
               double* q_out, const void* p_in, void* p_out, int p_f32, int jacobi, double* rec, int k,
               double* partials, Ctrl* ctrl, int split, int part, int partial_base, const int* lineoff,
               const long long* plane_first, const void* seg, hipStream_t s) {
  // automatic tile shape (1024^3 sweep, profiles/r01/cg1_sweep.txt): 16-wave, 32-row tiles pay
  // off on large planes; locals only, handles may be driven from several host threads
  const bool descr = lineoff && plane_first && g.tune.cg1_kernel == 0 && 8ll * g.plane < kBufMax;
  const Cg1Grid gr = cg1_grid(g, descr, split, part);
  if (gr.blocks <= 0) return gr.blocks;
  const int R = gr.R, NW = gr.NW, tiles_x = gr.tiles_x, tiles_y = gr.tiles_y, zchunk = gr.zchunk, nzc = gr.nzc, blocks = gr.blocks;
#define OI_CG1(RR, WW)                                                                                             \
  do {                                                                                                             \
    if (p_f32) launch_cg1_r<RR, WW, float>(blocks, g, code, x, r_in, r_out, q_in, q_out, p_in, p_out, jacobi, rec, \
                                           k, zchunk, tiles_x, tiles_y, nzc, part, partial_base, partials, ctrl,   \
                                           lineoff, plane_first, seg, s);                                          \
    else launch_cg1_r<RR, WW, double>(blocks, g, code, x, r_in, r_out, q_in, q_out, p_in, p_out, jacobi, rec, k,   \
                                      zchunk, tiles_x, tiles_y, nzc, part, partial_base, partials, ctrl, lineoff,  \
                                      plane_first, seg, s);                                                        \
  } while (0)
  if (R == 4) OI_CG1(4, 8);
  else if (R == 2 && NW == 16) OI_CG1(2, 16);
  else if (R == 2) OI_CG1(2, 8);
  else OI_CG1(1, 8);
#undef OI_CG1
  return blocks;
}

// a periodic problem (g.periodic) on k_cg1b's forms (their VAR & 8 instantiations): an even first plane; an odd nx, ny or nz leaves a
// single column / row / plane in the last aggregates as in the open problem
bool periodic_tiles_ok(const Geom& g) {
  return !g.periodic || (g.z0 & 1) == 0;
}
// can the descriptor-addressed forms of the two-kernel iteration (k_cg1b FORM 1 / FORM 2) run on this slab?
bool fused_descr_ok(const Geom& g, const int* lineoff, const long long* plane_first, const void* seg, int jacobi) {
  return lineoff && plane_first && seg && !jacobi && g.tune.fused_kernel == 0 && g.tune.fused_rows == 2 && periodic_tiles_ok(g) && 8ll * g.plane < kBufMax;
}
// k_cg1b FORM 2: r_out = r_in - alpha_{k-1} q (alpha from hist, 0 for k = 0), partials {r.r, r.r} per block, and the fp32
// right-hand side of the first coarse level (owned planes) from the pre-smoothed residual.  Needs an even g.z0.  Returns the
// number of blocks, -1 when the grid does not fit the partial-sum array
int launch_mg_pre_update(const Geom& g, const uint8_t* code, const double* r_in, double* r_out, const double* q, const double* hist, int k,
                         MgCoarse c, float* bcoarse, double omega, double* partials, const Ctrl* ctrl, const int* lineoff,
                         const long long* plane_first, const void* seg, hipStream_t s) {
  const int by = kNW * 2;
  const int tiles_x = (g.pitch + kBX - 1) / kBX, tiles_y = (g.ny + by - 1) / by;
  const long long cols = (long long)tiles_x * tiles_y;
  int zchunk, nzc;
  fused_chunks(g, cols, true, &zchunk, &nzc);
  if (!grid_fits(g, cols, nzc)) return -1;
  const int blocks = (int)(cols * nzc);
#define OI_F2(LD, ST)                                                                                                                     \
  hipLaunchKernelGGL((k_cg1b<2, kNW, double, false, true, 2, ST, LD>), dim3(blocks), dim3(64 * kNW), 0, s, g, code, (double*)nullptr, r_in, r_out, q, \
                     (double*)nullptr, (const double*)nullptr, (double*)nullptr, const_cast<double*>(hist), k, 0, zchunk, tiles_x, tiles_y, nzc, \
                     0, 0, partials, const_cast<Ctrl*>(ctrl), lineoff, plane_first, (const uint2*)seg, (g.pitch + kLV - 1) / kLV, 1, bcoarse, c, \
                     omega, 0.0)
#ifdef OI_EXP_CACHEPOL
  switch (g.tune.cg1_nt) {
    case 1: OI_F2(0, 1); return blocks;  case 2: OI_F2(2, 1); return blocks;  case 3: OI_F2(1, 1); return blocks;  case 4: OI_F2(3, 1); return blocks;
    case 5: OI_F2(2, 0); return blocks;  case 6: OI_F2(3, 0); return blocks;  case 7: OI_F2(1, 0); return blocks;  default: OI_F2(0, 0); return blocks;
  }
#endif
  if (g.periodic) {  // (the cell problems of the homogenization path: the form's periodic instantiation, oi_effdiff.hip)
    hipLaunchKernelGGL((k_cg1b<2, kNW, double, false, true, 2, kNtSt2, kNtLd2, 8>), dim3(blocks), dim3(64 * kNW), 0, s, g, code, (double*)nullptr, r_in, r_out, q,
                       (double*)nullptr, (const double*)nullptr, (double*)nullptr, const_cast<double*>(hist), k, 0, zchunk, tiles_x, tiles_y, nzc,
                       0, 0, partials, const_cast<Ctrl*>(ctrl), lineoff, plane_first, (const uint2*)seg, (g.pitch + kLV - 1) / kLV, 1, bcoarse, c,
                       omega, 0.0);
    return blocks;
  }
  if (g.tune.cg1_nt) OI_F2(kNtLd2, kNtSt2); else OI_F2(0, 0);
#undef OI_F2
  return blocks;
}
// k_cg1b FORM 3, the fine post-smoother of the cycle on compacted vectors: z = v + omega (b - A v) / c with
// v = omega b / c + scale * (coarse correction), partial sums of z.b per block (what launch_mg_fine mode 2 leaves, chunk for
// chunk: same arithmetic in the same order).  Returns the number of blocks, -1 when the grid does not fit
int launch_mg_post(const Geom& g, const uint8_t* code, const double* b, MgCoarse c, double omega, double scale, double* z,
                   double* partials, const Ctrl* ctrl, const int* lineoff, const long long* plane_first, const void* seg, hipStream_t s,
                   double* x, const void* p_prev, int p_f32, const double* hist, int k, const void* p_prev2) {
  const int by = kNW * 2;
  const int tiles_x = (g.pitch + kBX - 1) / kBX, tiles_y = (g.ny + by - 1) / by;
  const long long cols = (long long)tiles_x * tiles_y;
  int zchunk, nzc;
  fused_chunks(g, cols, false, &zchunk, &nzc);
  if (!grid_fits(g, cols, nzc)) return -1;
  const int blocks = (int)(cols * nzc);
  // x != null: x += alpha_{k-1} p_{k-1} (alpha from hist) on the planes this launch writes -- what the first kernel of the
  // two-kernel iteration otherwise does (launch_fused_a with x_elsewhere)
#define OI_F3N(LD, ST)                                                                                                               \
  hipLaunchKernelGGL((k_cg1b<2, kNW, double, false, true, 3, ST, LD, 0, true>), dim3(blocks), dim3(64 * kNW), 0, s, g, code, (double*)nullptr, b, z, \
                     (const double*)nullptr, (double*)nullptr, (const double*)nullptr, (double*)nullptr, const_cast<double*>(hist), k, 0, zchunk,    \
                     tiles_x, tiles_y, nzc, 0, 0, partials, const_cast<Ctrl*>(ctrl), lineoff, plane_first, (const uint2*)seg,                        \
                     (g.pitch + kLV - 1) / kLV, 1, (float*)nullptr, c, omega, scale)
#define OI_F3(PT, LD, ST)                                                                                                            \
  hipLaunchKernelGGL((k_cg1b<2, kNW, PT, false, true, 3, ST, LD>), dim3(blocks), dim3(64 * kNW), 0, s, g, code, x, b, z, (const double*)nullptr, \
                     (double*)nullptr, (const PT*)p_prev, (PT*)const_cast<void*>(p_prev2), const_cast<double*>(hist), k, (x && p_prev2) ? 1 : 0, zchunk, tiles_x, tiles_y, nzc, \
                     0, 0, partials, const_cast<Ctrl*>(ctrl), lineoff, plane_first, (const uint2*)seg, (g.pitch + kLV - 1) / kLV, 1,  \
                     (float*)nullptr, c, omega, scale)
#ifdef OI_EXP_CACHEPOL
  if (!(x && p_f32)) switch (g.tune.cg1_nt) {
    case 1: OI_F3(double, 0, 2); return blocks;   case 2: OI_F3(double, 0, 8); return blocks;   case 3: OI_F3(double, 0, 10); return blocks;
    case 4: OI_F3(double, 8, 10); return blocks;  case 5: OI_F3(double, 12, 10); return blocks; case 6: OI_F3(double, 12, 8); return blocks;
    case 7: OI_F3(double, 12, 0); return blocks;  default: OI_F3(double, 0, 0); return blocks;
  }
#endif
  // (does this launch touch x?  The kernel's own rule, evaluated here: an instantiation without the x path otherwise)
  const bool touches_x = x != nullptr && (p_prev2 ? (k >= 2 && (k & 1) == 0) : k > 0);
  if (g.periodic) {  // (periodic instantiations: fp64 p, the shipped cache policy)
    if (!touches_x)
      hipLaunchKernelGGL((k_cg1b<2, kNW, double, false, true, 3, kNtSt3, kNtLd3, 8, true>), dim3(blocks), dim3(64 * kNW), 0, s, g, code, (double*)nullptr, b, z,
                         (const double*)nullptr, (double*)nullptr, (const double*)nullptr, (double*)nullptr, const_cast<double*>(hist), k, 0, zchunk,
                         tiles_x, tiles_y, nzc, 0, 0, partials, const_cast<Ctrl*>(ctrl), lineoff, plane_first, (const uint2*)seg,
                         (g.pitch + kLV - 1) / kLV, 1, (float*)nullptr, c, omega, scale);
    else
      hipLaunchKernelGGL((k_cg1b<2, kNW, double, false, true, 3, kNtSt3, kNtLd3, 8>), dim3(blocks), dim3(64 * kNW), 0, s, g, code, x, b, z, (const double*)nullptr,
                         (double*)nullptr, (const double*)p_prev, (double*)const_cast<void*>(p_prev2), const_cast<double*>(hist), k, (x && p_prev2) ? 1 : 0, zchunk,
                         tiles_x, tiles_y, nzc, 0, 0, partials, const_cast<Ctrl*>(ctrl), lineoff, plane_first, (const uint2*)seg,
                         (g.pitch + kLV - 1) / kLV, 1, (float*)nullptr, c, omega, scale);
    return blocks;
  }
  if (!touches_x) { if (g.tune.cg1_nt) OI_F3N(kNtLd3, kNtSt3); else OI_F3N(0, 0); }
  else if (g.tune.cg1_nt) { if (p_f32) OI_F3(float, kNtLd3, kNtSt3); else OI_F3(double, kNtLd3, kNtSt3); }
  else { if (p_f32) OI_F3(float, 0, 0); else OI_F3(double, 0, 0); }
#undef OI_F3N
#undef OI_F3
  return blocks;
}
int launch_fused_a(const Geom& g, const uint8_t* code, double* x, const double* r, const void* p_in, void* p_out,
                   int p_f32, double* q, int jacobi, const double* hist, int k, double* partials, const Ctrl* ctrl,
                   const int* lineoff, const long long* plane_first, const void* seg, hipStream_t s, int x_elsewhere) {
  const int R = g.tune.fused_rows;
  // (a periodic problem takes the descriptor-addressed form -- its periodic instantiation -- with fp64 p and the x update in the
  // post-smoother, what oi_effdiff.hip runs; anything else periodic stays on k_fused_a)
  const bool per_descr = g.periodic && !p_f32 && x_elsewhere && fused_descr_ok(g, lineoff, plane_first, seg, jacobi);
  if (R == 2 && (!g.periodic || per_descr)) {
    const int by = kNW * R;
    const int tiles_x = (g.pitch + kBX - 1) / kBX, tiles_y = (g.ny + by - 1) / by;
    const long long cols = (long long)tiles_x * tiles_y;
    int zchunk, nzc;
    fused_chunks(g, cols, false, &zchunk, &nzc);
    if (!grid_fits(g, cols, nzc)) return -1;
    const int blocks = (int)(cols * nzc);
    const bool descr = fused_descr_ok(g, lineoff, plane_first, seg, jacobi);
    if (descr) {
#define OI_FBN(PT, LD, ST)                                                                                                     \
  hipLaunchKernelGGL((k_cg1b<2, kNW, PT, false, true, 1, ST, LD, 0, true>), dim3(blocks), dim3(64 * kNW), 0, s, g, code, (double*)nullptr, r, \
                     (double*)nullptr, (const double*)nullptr, q, (const PT*)p_in, (PT*)p_out, const_cast<double*>(hist), k, -1, zchunk, tiles_x, \
                     tiles_y, nzc, 0, 0, partials, const_cast<Ctrl*>(ctrl), lineoff, plane_first, (const uint2*)seg,             \
                     (g.pitch + kLV - 1) / kLV, 1, (float*)nullptr, MgCoarse{}, 0.0, 0.0)
#define OI_FB(PT, LD, ST)                                                                                                      \
  hipLaunchKernelGGL((k_cg1b<2, kNW, PT, false, true, 1, ST, LD>), dim3(blocks), dim3(64 * kNW), 0, s, g, code, x, r, (double*)nullptr,   \
                     (const double*)nullptr, q, (const PT*)p_in, (PT*)p_out, const_cast<double*>(hist), k, x_elsewhere ? -1 : 0, zchunk, tiles_x,  \
                     tiles_y, nzc, 0, 0, partials, const_cast<Ctrl*>(ctrl), lineoff, plane_first, (const uint2*)seg,             \
                     (g.pitch + kLV - 1) / kLV, 1, (float*)nullptr, MgCoarse{}, 0.0, 0.0)
#ifdef OI_EXP_CACHEPOL
      if (!p_f32) switch (g.tune.cg1_nt) {
        case 1: OI_FB(double, 0, 4); return blocks;  case 2: OI_FB(double, 0, 2); return blocks;  case 3: OI_FB(double, 0, 6); return blocks;
        case 4: OI_FB(double, 1, 6); return blocks;  case 5: OI_FB(double, 1, 4); return blocks;  case 6: OI_FB(double, 1, 0); return blocks;
        case 7: OI_FB(double, 4, 4); return blocks;  default: OI_FB(double, 0, 0); return blocks;
      }
#endif
      if (g.periodic) {  // (periodic instantiation: fp64 p, x updated by the post-smoother -- what oi_effdiff.hip runs)
        hipLaunchKernelGGL((k_cg1b<2, kNW, double, false, true, 1, kNtSt1, kNtLd1, 8, true>), dim3(blocks), dim3(64 * kNW), 0, s, g, code, (double*)nullptr, r,
                           (double*)nullptr, (const double*)nullptr, q, (const double*)p_in, (double*)p_out, const_cast<double*>(hist), k, -1, zchunk, tiles_x,
                           tiles_y, nzc, 0, 0, partials, const_cast<Ctrl*>(ctrl), lineoff, plane_first, (const uint2*)seg,
                           (g.pitch + kLV - 1) / kLV, 1, (float*)nullptr, MgCoarse{}, 0.0, 0.0);
        return blocks;
      }
      if (x_elsewhere) {  // (the post-smoother of this iteration has taken the x update: the instantiation without the x path)
        if (g.tune.cg1_nt) { if (p_f32) OI_FBN(float, kNtLd1, kNtSt1); else OI_FBN(double, kNtLd1, kNtSt1); }
        else { if (p_f32) OI_FBN(float, 0, 0); else OI_FBN(double, 0, 0); }
      } else if (g.tune.cg1_nt) { if (p_f32) OI_FB(float, kNtLd1, kNtSt1); else OI_FB(double, kNtLd1, kNtSt1); }
      else { if (p_f32) OI_FB(float, 0, 0); else OI_FB(double, 0, 0); }
#undef OI_FBN
#undef OI_FB
      return blocks;
    }
    return launch_fused_a_lds(g, code, x, r, p_in, p_out, p_f32, q, jacobi, hist, k, partials, ctrl, lineoff, s, blocks, zchunk, tiles_x, tiles_y, nzc);
  }
  return launch_fused_a_lds(g, code, x, r, p_in, p_out, p_f32, q, jacobi, hist, k, partials, ctrl, lineoff, s, 0, 0, 0, 0, 0);
}

}  // namespace oi
