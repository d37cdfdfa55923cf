// oi_k_xcheck.hip -- SUPERSEDED kernels kept as bit-identity references of the shipped ones; part of the test library only
// (make xcheck -> libopenimpala_amd_xcheck.so, -DOI_XCHECK=1; tests/test_xcheck_gpu.py).  See oi_kern.h.
#include "oi_kern.h"

namespace oi {


// --------------------------------------------------------------------------------------
// Single-kernel CG iteration ("cg1").  Kernel k turns the state after k-1 iterations into
// the state after k iterations plus the next search direction and its image:
//     x_k  = x_{k-1} + alpha_{k-1} p_{k-1}
//     r_k  = r_{k-1} - alpha_{k-1} q_{k-1}
//     p_k  = z_k + beta_k p_{k-1}            (z = r or r/code)
//     q_k  = A p_k
//     partials { r_k.r_k, r_k.z_k, p_k.q_k, z_k.q_k, q_k.D^-1 q_k }
// One reduction per iteration: alpha_k = rz_k / pq and, because r_{k+1} = r_k - alpha_k q_k,
//     rz_{k+1} = rz_k - 2 alpha_k (z_k.q_k) + alpha_k^2 (q_k.D^-1 q_k)      (exact arithmetic)
// gives beta_{k+1} = rz_{k+1} / rz_k BEFORE r_{k+1} exists; the next kernel forms r_{k+1}
// and reports its exact norms, so the extrapolation never spans more than one step
// (D'Azevedo/Eijkhout/Romine-style rearrangement; measured on the 100^3 sample: same
// iteration counts as textbook CG down to 1e-11, r stays the true residual of x).
// Same z-marching LDS-tiled structure as k_fused_a; r_k and p_k on the halo ring are
// recomputed from r_{k-1}, q_{k-1}, p_{k-1}, which is why r, q and p are all
// double-buffered (a neighbouring block may already have overwritten its tile).
// Compulsory HBM traffic: reads r 8 + q 8 + p 8 + x 8 + code 1, writes r 8 + q 8 + p 8 + x 8
// = 65 B/voxel-iteration in ONE launch (57 with p stored as fp32).
// rec[k] = { rr, rz, pq, zq, qdq, alpha_k, beta_{k+1}, - }.
// --------------------------------------------------------------------------------------
template <int R, int NW, typename PT, bool JACOBI, bool COMPACT>
__global__ void __launch_bounds__(64 * NW, (R <= 2 ? 4 : 2)) k_cg1(Geom g, const uint8_t* __restrict__ code, double* __restrict__ x,
                                                   const double* __restrict__ r_in, double* __restrict__ r_out,
                                                   const double* __restrict__ q_in, double* __restrict__ q_out,
                                                   const PT* __restrict__ p_in, PT* __restrict__ p_out,
                                                   double* rec, int k, int xskip, int zchunk,
                                                   int tiles_x, int tiles_y, int nzc, int part, int partial_base,
                                                   double* __restrict__ partials, Ctrl* ctrl,
                                                   const int* __restrict__ lineoff, int LW, int fullline) {
  constexpr int BY = NW * R;
  static_assert(2 * BY <= 64, "wave 2 loads the 2*BY x-halo values with one lane each");
  if (ctrl->done) return;
  __shared__ __attribute__((aligned(16))) double tile[2][(BY + 2) * kLP];
  __shared__ double red[NW];

  // part 0: every z chunk; 1: only the first and last chunk of the slab (they produce the
  // planes the neighbours wait for); 2: the interior chunks (run while the planes travel)
  const int nsel = part == 0 ? nzc : (part == 1 ? min(2, nzc) : nzc - 2);
  const int L = tile_of_block(blockIdx.x, tiles_x * tiles_y, nsel, g.tune.tile_map);
  const int tx = L % tiles_x, ty = (L / tiles_x) % tiles_y, zsel = L / (tiles_x * tiles_y);
  const int zc = part == 0 ? zsel : (part == 1 ? (zsel == 0 ? 0 : nzc - 1) : zsel + 1);

  // record k - 1 is complete (kernel boundary / all-reduce): close it here instead of in a launch of its own
  double beta = 0.0, alpha_prev = 0.0, alpha_prev2 = 0.0;
  if (k > 0) {
    double* rk1 = rec + R1_STRIDE * (k - 1);
    const Cg1Scalars sc = cg1_scalars_of(rk1, ctrl->tol2);
    if (blockIdx.x == 0 && threadIdx.x == 0) {  // (every block holds the same values; one writes them for the host)
      rk1[R1_ALPHA] = sc.alpha;
      rk1[R1_BETA] = sc.beta;
      if (sc.converged) ctrl->done = k;  // = (k - 1) + 1: iterations after which ||r|| <= tol
    }
    if (sc.converged) return;
    alpha_prev = sc.alpha;
    beta = sc.beta;
  }
  // x is touched every iteration (xskip = 0) or every second one (xskip = 1): an even kernel
  // k >= 2 adds alpha_{k-2} p_{k-2} + alpha_{k-1} p_{k-1}; p_{k-2} is what the p_out buffer
  // still holds at the very address this thread is about to overwrite with p_k
  const bool do_x = xskip ? (k >= 2 && (k & 1) == 0) : (k > 0);
  if (xskip && do_x) alpha_prev2 = rec[R1_STRIDE * (k - 2) + R1_ALPHA];

  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int i = tx * kBX + 2 * lane, j0 = ty * BY;
  const bool in_x = i < g.pitch;
  const int gs = 1 + zc * zchunk, ge = min(g.nzl + 1, gs + zchunk);

  const int hrow = (w == 0) ? j0 - 1 : j0 + BY;
  const bool yhalo = (w < 2) && in_x && hrow >= 0 && hrow < g.ny;
  const int xh_row = j0 + (lane >> 1);
  const int xh_i = (lane & 1) ? tx * kBX + kBX : tx * kBX - 1;
  const bool xhalo = (w == 2) && lane < 2 * BY && xh_row < g.ny && xh_i >= 0 && xh_i < g.pitch;

  // r, q and p vanish on non-free voxels (code 0): their loads and stores are predicated on the code
  // byte, so a cache line without a free voxel is never fetched or written (spheres at porosity 0.4:
  // 41 % of the 128-byte lines)
  // code bytes (and, compacted layout, the line slot) of this thread's voxel pairs in plane gz, fetched
  // one plane ahead of the vectors
  auto load_codes = [&](int gz, unsigned short (&cn)[R], int (&ln)[R]) {
#pragma unroll
    for (int rr_ = 0; rr_ < R; ++rr_) {
      const int j = j0 + w + rr_ * NW;
      cn[rr_] = 0;
      ln[rr_] = -1;
      if (in_x && j < g.ny && gz <= g.nzl + 1) {
        cn[rr_] = *reinterpret_cast<const unsigned short*>(code + (long long)gz * g.plane + (long long)j * g.pitch + i);
        if (COMPACT) ln[rr_] = lineoff[((long long)gz * g.ny + j) * LW + (i >> kLVs)];
      }
      if (!COMPACT) {  // dense layout: a 16-voxel line is live when any of its 8 lanes holds a free voxel
        const unsigned long long m = __ballot(cn[rr_] != 0);
        ln[rr_] = (in_x && j < g.ny && gz <= g.nzl + 1 && ((m >> (lane & ~(kDenseLG - 1))) & ((1ull << kDenseLG) - 1))) ? 0 : -1;
      }
    }
  };
  // mode 0: a halo plane owned by another block (nothing stored); 1: an output plane of this
  // block (r_k, p_k stored, x updated at once: keeps p_{k-1} out of the register budget);
  // 2: a ghost plane of the slab, kept current by the block that marches over it (r_k, p_k)
  auto load_plane = [&](int gz, const unsigned short (&cn)[R], const int (&ln)[R], double2 (&pn)[R], double2 (&rn)[R],
                        int mode) {
#pragma unroll
    for (int rr_ = 0; rr_ < R; ++rr_) {
      const int j = j0 + w + rr_ * NW;
      pn[rr_] = make_double2(0.0, 0.0);
      rn[rr_] = make_double2(0.0, 0.0);
      // compacted layout: every lane of a stored line takes part, so lines are read AND written whole (a
      // pair without a free voxel reads and writes zeros) -- no partial-line stores for HBM to merge
      if (fullline ? ln[rr_] >= 0 : cn[rr_] != 0) {  // (implies in_x && j < g.ny)
        const long long e = (long long)gz * g.plane + (long long)j * g.pitch + i;            // dense: x
        const long long ev = COMPACT ? (long long)kLV * ln[rr_] + (i & (kLV - 1)) : e;                         // r, q, p
        const double2 rv = ld2(r_in + ev), qv = ld2(q_in + ev);
        const double2 pin = PIo<PT>::ld_pair(p_in + ev);
        const double r0 = rv.x - alpha_prev * qv.x, r1 = rv.y - alpha_prev * qv.y;
        rn[rr_] = make_double2(r0, r1);
        double z0 = r0, z1 = r1;
        if (JACOBI) {
          const unsigned short cc = cn[rr_];
          if (cc & 0xff) z0 = r0 / (double)(cc & 0xff);
          if (cc >> 8) z1 = r1 / (double)(cc >> 8);
        }
        pn[rr_] = make_double2(PIo<PT>::rnd(z0 + beta * pin.x), PIo<PT>::rnd(z1 + beta * pin.y));
        if (mode != 0) {
          st2(r_out + ev, r0, r1);
          if (mode == 1 && do_x) {
            const double2 xv = ld2(x + e);
            double a0 = alpha_prev * pin.x, a1 = alpha_prev * pin.y;
            if (xskip) {
              const double2 pold = PIo<PT>::ld_pair(p_out + ev);  // p_{k-2}
              a0 = alpha_prev2 * pold.x + a0; a1 = alpha_prev2 * pold.y + a1;
            }
            st2(x + e, xv.x + a0, xv.y + a1);
          }
          PIo<PT>::st_pair(p_out + ev, pn[rr_].x, pn[rr_].y);
        }
      }
    }
  };
  unsigned hcode = 0;  // halo lanes: code byte(s) and slot of their halo voxel (pair) in the plane staged next
  int hslot = -1;
  auto stage = [&](int b, int gz, const double2 (&pn)[R]) {
    double* T = tile[b];
#pragma unroll
    for (int rr_ = 0; rr_ < R; ++rr_)
      *reinterpret_cast<double2*>(&T[(w + rr_ * NW + 1) * kLP + 2 + 2 * lane]) = pn[rr_];
    // halo code bytes and slots were fetched a plane ahead (load_halo_codes): the halo's vector loads go out with
    // no dependent look-up in front of them (7.26 vs 7.59 ms at 1024^3; the halo waves are the block's critical path)
    {
      if (w < 2) {
        double2 h = make_double2(0.0, 0.0);
        if (hcode) {
          const long long ev = COMPACT ? (long long)kLV * hslot + (i & (kLV - 1)) : (long long)gz * g.plane + (long long)hrow * g.pitch + i;
          const double2 rv = ld2(r_in + ev), qv = ld2(q_in + ev);
          const double2 pv = PIo<PT>::ld_pair(p_in + ev);
          double z0 = rv.x - alpha_prev * qv.x, z1 = rv.y - alpha_prev * qv.y;
          if (JACOBI) {
            if (hcode & 0xff) z0 = z0 / (double)(hcode & 0xff);
            if (hcode >> 8) z1 = z1 / (double)(hcode >> 8);
          }
          h.x = PIo<PT>::rnd(z0 + beta * pv.x); h.y = PIo<PT>::rnd(z1 + beta * pv.y);
        }
        *reinterpret_cast<double2*>(&T[((w == 0) ? 0 : BY + 1) * kLP + 2 + 2 * lane]) = h;
      } else if (w == 2 && lane < 2 * BY) {
        double h = 0.0;
        if (hcode) {
          const long long ev = COMPACT ? (long long)kLV * hslot + (xh_i & (kLV - 1)) : (long long)gz * g.plane + (long long)xh_row * g.pitch + xh_i;
          double z = r_in[ev] - alpha_prev * q_in[ev];
          if (JACOBI) z = z / (double)hcode;
          h = PIo<PT>::rnd(z + beta * (double)p_in[ev]);
        }
        T[((lane >> 1) + 1) * kLP + ((lane & 1) ? 2 + kBX : 1)] = h;
      }
    }
  };
  auto load_halo_codes = [&](int gz) {  // of plane gz, for the stage() call of the NEXT iteration
    hcode = 0; hslot = -1;
    if (gz > g.nzl + 1) return;
    if (yhalo) {
      hcode = *reinterpret_cast<const unsigned short*>(code + (long long)gz * g.plane + (long long)hrow * g.pitch + i);
      if (COMPACT) hslot = lineoff[((long long)gz * g.ny + hrow) * LW + (i >> kLVs)];
    } else if (xhalo) {
      hcode = code[(long long)gz * g.plane + (long long)xh_row * g.pitch + xh_i];
      if (COMPACT) hslot = lineoff[((long long)gz * g.ny + xh_row) * LW + (xh_i >> kLVs)];
    }
  };

  double2 pn_prev[R], pn_cur[R], pn_nxt[R], rn_cur[R], rn_nxt[R];
  unsigned short cc_cur[R], cc_nxt[R], cc_nn[R];
  int lo_cur[R], lo_nxt[R], lo_nn[R];
  load_halo_codes(gs);
  load_codes(gs - 1, cc_nn, lo_nn);
  load_codes(gs, cc_cur, lo_cur);
  load_codes(gs + 1, cc_nxt, lo_nxt);
  load_plane(gs - 1, cc_nn, lo_nn, pn_prev, rn_nxt, gs == 1 ? 2 : 0);
  load_plane(gs, cc_cur, lo_cur, pn_cur, rn_cur, 1);
  stage(gs & 1, gs, pn_cur);
  load_halo_codes(gs + 1);
  __syncthreads();

  double s_rr = 0.0, s_rz = 0.0, s_pq = 0.0, s_zq = 0.0, s_qdq = 0.0;
  for (int gz = gs; gz < ge; ++gz) {
    load_codes(gz + 2, cc_nn, lo_nn);
    load_plane(gz + 1, cc_nxt, lo_nxt, pn_nxt, rn_nxt, gz + 1 < ge ? 1 : (ge == g.nzl + 1 ? 2 : 0));
    if (gz + 1 < ge) { stage((gz + 1) & 1, gz + 1, pn_nxt); load_halo_codes(gz + 2); }
    const double* T = tile[gz & 1];
#pragma unroll
    for (int rr_ = 0; rr_ < R; ++rr_) {
      const int j = j0 + w + rr_ * NW;
      const unsigned short cc = cc_cur[rr_];
      if (fullline && !cc && lo_cur[rr_] >= 0)  // whole lines
        st2(q_out + (COMPACT ? (long long)kLV * lo_cur[rr_] + (i & (kLV - 1)) : (long long)gz * g.plane + (long long)j * g.pitch + i), 0.0, 0.0);
      if (cc) {  // q and every partial sum vanish where both voxels are non-free
        const int base = (w + rr_ * NW + 1) * kLP + 2 + 2 * lane;
        const double2 lf = *reinterpret_cast<const double2*>(&T[base - 2]);
        const double2 rt = *reinterpret_cast<const double2*>(&T[base + 2]);
        const double2 ym = *reinterpret_cast<const double2*>(&T[base - kLP]);
        const double2 yp = *reinterpret_cast<const double2*>(&T[base + kLP]);
        const long long e = (long long)gz * g.plane + (long long)j * g.pitch + i;
        const unsigned c0 = cc & 0xffu, c1 = cc >> 8;
        const double2 pc = pn_cur[rr_], rc = rn_cur[rr_];
        const double s0 = ((((lf.y + pc.y) + ym.x) + yp.x) + pn_prev[rr_].x) + pn_nxt[rr_].x;
        const double s1 = ((((pc.x + rt.x) + ym.y) + yp.y) + pn_prev[rr_].y) + pn_nxt[rr_].y;
        const double q0 = c0 ? (double)c0 * pc.x - s0 : 0.0;
        const double q1 = c1 ? (double)c1 * pc.y - s1 : 0.0;
        st2(q_out + (COMPACT ? (long long)kLV * lo_cur[rr_] + (i & (kLV - 1)) : e), q0, q1);
        double z0 = rc.x, z1 = rc.y, d0 = q0, d1 = q1;
        if (JACOBI) {
          if (c0) { z0 = rc.x / (double)c0; d0 = q0 / (double)c0; }
          if (c1) { z1 = rc.y / (double)c1; d1 = q1 / (double)c1; }
        }
        s_rr += rc.x * rc.x; s_rr += rc.y * rc.y;
        s_rz += rc.x * z0; s_rz += rc.y * z1;
        s_pq += pc.x * q0; s_pq += pc.y * q1;
        s_zq += z0 * q0; s_zq += z1 * q1;
        s_qdq += q0 * d0; s_qdq += q1 * d1;
      }
    }
#pragma unroll
    for (int rr_ = 0; rr_ < R; ++rr_) {
      pn_prev[rr_] = pn_cur[rr_]; pn_cur[rr_] = pn_nxt[rr_]; rn_cur[rr_] = rn_nxt[rr_];
      cc_cur[rr_] = cc_nxt[rr_]; cc_nxt[rr_] = cc_nn[rr_];
      lo_cur[rr_] = lo_nxt[rr_]; lo_nxt[rr_] = lo_nn[rr_];
    }
    __syncthreads();
  }
  const double t0 = block_sum(s_rr, red);
  const double t1 = block_sum(s_rz, red);
  const double t2 = block_sum(s_pq, red);
  const double t3 = block_sum(s_zq, red);
  const double t4 = block_sum(s_qdq, red);
  if (threadIdx.x == 0) {
    double* o = partials + 5 * ((long long)partial_base + blockIdx.x);
    o[0] = t0; o[1] = t1; o[2] = t2; o[3] = t3; o[4] = t4;
  }
}

template <int R, int NW, typename PT, bool JAC, bool COMPACT>
static void launch_cg1_k(int blocks, const Geom& g, const uint8_t* code, double* x, const double* r_in, double* r_out,
                         const double* q_in, double* q_out, const void* p_in, void* p_out, double* rec, int k,
                         int zchunk, int tiles_x, int tiles_y, int nzc, int part, int partial_base, double* partials,
                         Ctrl* ctrl, const int* lineoff, hipStream_t s) {
  hipLaunchKernelGGL((k_cg1<R, NW, PT, JAC, COMPACT>), dim3(blocks), dim3(64 * NW), 0, s, g, code, x, r_in, r_out, q_in,
                     q_out, (const PT*)p_in, (PT*)p_out, rec, k, g.tune.cg1_xskip, zchunk, tiles_x, tiles_y, nzc, part,
                     partial_base, partials, ctrl, lineoff, (g.pitch + kLV - 1) / kLV, g.tune.cg1_fullline);
}

int launch_cg1_first_form(int R, int NW, int blocks, const Geom& g, const uint8_t* code, double* x, const double* r_in, double* r_out,
                          const double* q_in, double* q_out, const void* p_in, void* p_out, int p_f32, int jacobi, double* rec, int k, int zchunk,
                          int tiles_x, int tiles_y, int nzc, int part, int partial_base, double* partials, Ctrl* ctrl, const int* lineoff,
                          hipStream_t s) {
#define OI_CG1K(RR, WW, PT, JAC, CMP)                                                                                      \
  launch_cg1_k<RR, WW, PT, JAC, CMP>(blocks, g, code, x, r_in, r_out, q_in, q_out, p_in, p_out, rec, k, zchunk, tiles_x, \
                                     tiles_y, nzc, part, partial_base, partials, ctrl, lineoff, s)
#define OI_CG1P(RR, WW, PT)                                                                        \
  do {                                                                                             \
    if (jacobi) { if (lineoff) OI_CG1K(RR, WW, PT, true, true); else OI_CG1K(RR, WW, PT, true, false); }   \
    else { if (lineoff) OI_CG1K(RR, WW, PT, false, true); else OI_CG1K(RR, WW, PT, false, false); }        \
  } while (0)
#define OI_CG1S(RR, WW) do { if (p_f32) OI_CG1P(RR, WW, float); else OI_CG1P(RR, WW, double); } while (0)
  if (R == 4) OI_CG1S(4, 8);
  else if (R == 2 && NW == 16) OI_CG1S(2, 16);
  else if (R == 2) OI_CG1S(2, 8);
  else OI_CG1S(1, 8);
#undef OI_CG1S
#undef OI_CG1P
#undef OI_CG1K
  return blocks;
}

}  // namespace oi
