// oi_k_dense.hip -- kernels of the dense vector layout, the LDS-tiled z-marching kernels, the stand-alone apply.  See oi_kern.h.
#include "oi_kern.h"

namespace oi {

// q = A v, partials {v.q}
__global__ void __launch_bounds__(256) k_apply(Geom g, const uint8_t* __restrict__ code,
                                                const double* __restrict__ v, double* __restrict__ q,
                                                double* __restrict__ partials, const Ctrl* __restrict__ ctrl) {
  if (ctrl && ctrl->done) return;
  double pq = 0.0;
  OI_PAIR_LOOP_BEGIN
  const Pair a = apply_pair(g, v, e, i, j, c0, c1);
  st2(q + e, a.a, a.b);
  const double2 c = ld2(v + e);
  pq += c.x * a.a; pq += c.y * a.b;
  OI_PAIR_LOOP_END
  __shared__ double red[4];
  const double t = block_sum(pq, red);
  if (threadIdx.x == 0) partials[blockIdx.x] = t;
}

int launch_apply(const Geom& g, const uint8_t* code, const double* v, double* q, double* partials,
                 const Ctrl* ctrl, hipStream_t s) {
  const int blocks = cap_blocks((long long)g.nzl * g.ny * (g.pitch / 2), 256, 2048);
  hipLaunchKernelGGL(k_apply, dim3(blocks), dim3(256), 0, s, g, code, v, q, partials, ctrl);
  return blocks;
}



__global__ void __launch_bounds__(256) k_update_p(Geom g, const uint8_t* __restrict__ code,
                                                   const double* __restrict__ r, double* __restrict__ p, int jacobi,
                                                   const double* __restrict__ hist, int k,
                                                   const Ctrl* __restrict__ ctrl) {
  if (ctrl->done) return;
  const double beta = safe_div(hist[H_STRIDE * k + H_RZ], hist[H_STRIDE * (k - 1) + H_RZ]);
  OI_PAIR_LOOP_BEGIN
  (void)i; (void)j;
  if (!line_live(cc != 0)) continue;  // r and p vanish on non-free voxels: dead lines are neither read nor written
  const double2 rv = ld2(r + e), pv = ld2(p + e);
  const double z0 = (jacobi && c0) ? rv.x / (double)c0 : rv.x;
  const double z1 = (jacobi && c1) ? rv.y / (double)c1 : rv.y;
  st2(p + e, z0 + beta * pv.x, z1 + beta * pv.y);
  OI_PAIR_LOOP_END
}

void launch_update_p(const Geom& g, const uint8_t* code, const double* r, double* p, int jacobi,
                     const double* hist, int k, const Ctrl* ctrl, hipStream_t s) {
  const int blocks = cap_blocks((long long)g.nzl * g.ny * (g.pitch / 2), 256, 2048);
  hipLaunchKernelGGL(k_update_p, dim3(blocks), dim3(256), 0, s, g, code, r, p, jacobi, hist, k, ctrl);
}

// Flat streams in BLOCKED form: every block owns one contiguous chunk and keeps four
// independent 16-byte loads per array in flight (tools/stream_bench.hip: +4..6 % over a
// grid-stride loop on MI355X).
template <bool WITH_X, bool JACOBI>
__global__ void __launch_bounds__(256) k_update_xr(Geom g, const uint8_t* __restrict__ code, double* __restrict__ x,
                                                    double* __restrict__ r, const double* __restrict__ p,
                                                    const double* __restrict__ q, const double* __restrict__ hist,
                                                    int k, double* __restrict__ partials,
                                                    const Ctrl* __restrict__ ctrl) {
  if (ctrl->done) return;
  const double alpha = safe_div(hist[H_STRIDE * k + H_RZ], hist[H_STRIDE * k + H_PQ]);
  double rr = 0.0, rz = 0.0;
  constexpr int U = 4;
  const long long npairs = (long long)g.nzl * g.plane / 2;
  const long long per = (((npairs + gridDim.x - 1) / gridDim.x) + 7) & ~7ll;  // chunks start on a line boundary
  const long long b0 = per * blockIdx.x, b1 = min(b0 + per, npairs);
  // r, q, p vanish on non-free voxels (code 0): pairs without a free voxel are neither read nor written
  auto body = [&](long long t) {
    const long long e = g.plane + 2 * t;
    const unsigned short cc_ = *reinterpret_cast<const unsigned short*>(code + e);
    if (!line_live(cc_ != 0)) return;
    const double2 rv = ld2(r + e), qv = ld2(q + e);
    if (WITH_X) {
      const double2 xv = ld2(x + e), pv = ld2(p + e);
      st2p(x + e, xv.x + alpha * pv.x, xv.y + alpha * pv.y, g.tune.dense_nt & 2);
    }
    const double r0 = rv.x - alpha * qv.x, r1 = rv.y - alpha * qv.y;
    st2p(r + e, r0, r1, g.tune.dense_nt & 8);
    double z0 = r0, z1 = r1;
    if (JACOBI) {
      const unsigned short cc = *reinterpret_cast<const unsigned short*>(code + e);
      if (cc & 0xff) z0 = r0 / (double)(cc & 0xff);
      if (cc >> 8) z1 = r1 / (double)(cc >> 8);
    }
    rr += r0 * r0; rr += r1 * r1;
    rz += r0 * z0; rz += r1 * z1;
  };
  long long t = b0 + threadIdx.x;
  for (; t + (U - 1) * 256 < b1; t += U * 256) {
    double2 rv[U], qv[U];
    unsigned short cu[U];
    bool lv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      cu[u] = *reinterpret_cast<const unsigned short*>(code + g.plane + 2 * (t + u * 256));
      lv[u] = line_live(cu[u] != 0);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      rv[u] = make_double2(0.0, 0.0); qv[u] = make_double2(0.0, 0.0);
      if (lv[u]) { rv[u] = ld2(r + g.plane + 2 * (t + u * 256)); qv[u] = ld2(q + g.plane + 2 * (t + u * 256)); }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (!lv[u]) continue;
      const long long e = g.plane + 2 * (t + u * 256);
      if (WITH_X) {
        const double2 xv = ld2(x + e), pv = ld2(p + e);
        st2(x + e, xv.x + alpha * pv.x, xv.y + alpha * pv.y);
      }
      const double r0 = rv[u].x - alpha * qv[u].x, r1 = rv[u].y - alpha * qv[u].y;
      st2(r + e, r0, r1);
      double z0 = r0, z1 = r1;
      if (JACOBI) {
        const unsigned short cc = cu[u];
        if (cc & 0xff) z0 = r0 / (double)(cc & 0xff);
        if (cc >> 8) z1 = r1 / (double)(cc >> 8);
      }
      rr += r0 * r0; rr += r1 * r1;
      rz += r0 * z0; rz += r1 * z1;
    }
  }
  for (; t < b1; t += 256) body(t);
  __shared__ double red[4];
  const double a = block_sum(rr, red);
  const double b = block_sum(rz, red);
  if (threadIdx.x == 0) { partials[2 * blockIdx.x] = a; partials[2 * blockIdx.x + 1] = b; }
}

template <bool WITH_X>
static int launch_update_any(const Geom& g, const uint8_t* code, double* x, double* r, const double* p,
                             const double* q, int jacobi, const double* hist, int k, double* partials,
                             const Ctrl* ctrl, hipStream_t s) {
  const int blocks = cap_blocks((long long)g.nzl * g.plane / 2, 1024, 2048);
  if (jacobi)
    hipLaunchKernelGGL((k_update_xr<WITH_X, true>), dim3(blocks), dim3(256), 0, s, g, code, x, r, p, q, hist, k,
                       partials, ctrl);
  else
    hipLaunchKernelGGL((k_update_xr<WITH_X, false>), dim3(blocks), dim3(256), 0, s, g, code, x, r, p, q, hist, k,
                       partials, ctrl);
  return blocks;
}

int launch_update_xr(const Geom& g, const uint8_t* code, double* x, double* r, const double* p, const double* q,
                     int jacobi, const double* hist, int k, double* partials, const Ctrl* ctrl, hipStream_t s) {
  return launch_update_any<true>(g, code, x, r, p, q, jacobi, hist, k, partials, ctrl, s);
}

// r -= alpha q over COMPACTED vectors (every stored line is live: a plain stream); partials {r.r, r.r}
__global__ void __launch_bounds__(256) k_update_r_lines(long long npairs, double* __restrict__ r, const double* __restrict__ q,
                                                         const double* __restrict__ hist, int k, double* __restrict__ partials,
                                                         const Ctrl* __restrict__ ctrl) {
  if (ctrl->done) return;
  const double alpha = safe_div(hist[H_STRIDE * k + H_RZ], hist[H_STRIDE * k + H_PQ]);
  double rr = 0.0;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < npairs; t += (long long)gridDim.x * blockDim.x) {
    const double2 rv = ld2(r + 2 * t), qv = ld2(q + 2 * t);
    const double r0 = rv.x - alpha * qv.x, r1 = rv.y - alpha * qv.y;
    st2(r + 2 * t, r0, r1);
    rr += r0 * r0; rr += r1 * r1;
  }
  __shared__ double red[4];
  const double a = block_sum(rr, red);
  if (threadIdx.x == 0) { partials[2 * blockIdx.x] = a; partials[2 * blockIdx.x + 1] = a; }
}

// (first_slot / nslots: the compacted layout's owned planes -- r and q then are line lists and no Jacobi scaling applies)
int launch_update_r(const Geom& g, const uint8_t* code, double* r, const double* q, int jacobi, const double* hist,
                    int k, double* partials, const Ctrl* ctrl, long long first_slot, long long nslots, hipStream_t s) {
  if (nslots >= 0) {
    const long long npairs = nslots * (kLV / 2);
    const int blocks = cap_blocks(npairs, 1024, 2048);
    hipLaunchKernelGGL(k_update_r_lines, dim3(blocks), dim3(256), 0, s, npairs, r + kLV * first_slot, q + kLV * first_slot, hist, k,
                       partials, ctrl);
    return blocks;
  }
  return launch_update_any<false>(g, code, nullptr, r, nullptr, q, jacobi, hist, k, partials, ctrl, s);
}


template <typename PT>
__global__ void __launch_bounds__(256) k_axpy_hist(Geom g, double* __restrict__ x, const PT* __restrict__ p,
                                                    const double* __restrict__ hist, int k, const int* __restrict__ lineoff,
                                                    int LW, const PT* __restrict__ p2) {
  const double alpha = safe_div(hist[H_STRIDE * k + H_RZ], hist[H_STRIDE * k + H_PQ]);
  const double alpha2 = p2 ? safe_div(hist[H_STRIDE * (k + 1) + H_RZ], hist[H_STRIDE * (k + 1) + H_PQ]) : 0.0;
  const long long npairs = (long long)g.nzl * g.plane / 2;
  const int hp = g.pitch >> 1;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < npairs;
       t += (long long)gridDim.x * blockDim.x) {
    const long long e = g.plane + 2 * t;
    long long ev = e;
    if (lineoff) {  // p in the compacted line layout: lines that are not stored contribute nothing
      const int i = 2 * (int)(t % hp);
      const int lo = lineoff[(t / hp + g.ny) * LW + (i >> kLVs)];
      if (lo < 0) continue;
      ev = (long long)kLV * lo + (i & (kLV - 1));
    }
    const double2 xv = ld2(x + e), pv = PIo<PT>::ld_pair(p + ev);
    double x0 = xv.x + alpha * pv.x, x1 = xv.y + alpha * pv.y;
    if (p2) {
      const double2 qv = PIo<PT>::ld_pair(p2 + ev);
      x0 = x0 + alpha2 * qv.x; x1 = x1 + alpha2 * qv.y;
    }
    st2(x + e, x0, x1);
  }
}

// (lineoff != null: p in the compacted line layout, lines that are not stored contribute nothing)
template <typename PT>
__global__ void __launch_bounds__(256) k_axpy_ptr(Geom g, double* __restrict__ x, const PT* __restrict__ p,
                                                   const double* __restrict__ alpha_ptr, const int* __restrict__ lineoff,
                                                   int LW) {
  const double alpha = *alpha_ptr;
  const long long npairs = (long long)g.nzl * g.plane / 2;
  const int hp = g.pitch >> 1;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < npairs;
       t += (long long)gridDim.x * blockDim.x) {
    const long long e = g.plane + 2 * t;
    long long ev = e;
    if (lineoff) {
      const int i = 2 * (int)(t % hp);
      const long long row = t / hp;  // (gz - 1) * ny + j
      const int lo = lineoff[(row + g.ny) * LW + (i >> kLVs)];
      if (lo < 0) continue;
      ev = (long long)kLV * lo + (i & (kLV - 1));
    }
    const double2 xv = ld2(x + e), pv = PIo<PT>::ld_pair(p + ev);
    st2(x + e, xv.x + alpha * pv.x, xv.y + alpha * pv.y);
  }
}

void launch_axpy_ptr(const Geom& g, double* x, const void* p, int p_f32, const double* alpha_ptr, const int* lineoff,
                     hipStream_t s) {
  const int blocks = cap_blocks((long long)g.nzl * g.plane / 2, 256, 2048);
  const int LW = (g.pitch + kLV - 1) / kLV;
  if (p_f32) hipLaunchKernelGGL(k_axpy_ptr<float>, dim3(blocks), dim3(256), 0, s, g, x, (const float*)p, alpha_ptr, lineoff, LW);
  else hipLaunchKernelGGL(k_axpy_ptr<double>, dim3(blocks), dim3(256), 0, s, g, x, (const double*)p, alpha_ptr, lineoff, LW);
}

void launch_axpy_hist(const Geom& g, double* x, const void* p, int p_f32, const double* hist, int k, const int* lineoff,
                      hipStream_t s, const void* p_next) {
  const int blocks = cap_blocks((long long)g.nzl * g.plane / 2, 256, 2048);
  const int LW = (g.pitch + kLV - 1) / kLV;

  if (p_f32) hipLaunchKernelGGL(k_axpy_hist<float>, dim3(blocks), dim3(256), 0, s, g, x, (const float*)p, hist, k, lineoff, LW, (const float*)p_next);
  else hipLaunchKernelGGL(k_axpy_hist<double>, dim3(blocks), dim3(256), 0, s, g, x, (const double*)p, hist, k, lineoff, LW, (const double*)p_next);
}

// --------------------------------------------------------------------------------------
// Fused CG kernel A (z-marching, LDS-staged halo tiles):
//     x     += alpha_{k-1} * p_in            (deferred solution update)
//     p_out  = z + beta_k * p_in             (z = r or r/code)
//     q      = A p_out,   partial p_out.q
// A block of 8 waves owns a 128 x (8 R) column of voxels and marches over a chunk of z
// planes.  Every thread owns R x-pairs (rows w, w+8, ...); the p_out values of planes
// g-1, g, g+1 of its pairs live in registers (z reuse), plane g of the whole tile incl. a
// one-voxel halo ring lives in a double-buffered LDS plane (x/y reuse; halo values are
// recomputed from r and p_in, which the neighbouring tile's block pulls through the same
// XCD L2).  Compulsory HBM traffic: r 8 + p_in 8 + code 1 + x 8+8 + p_out 8 + q 8 = 49
// B/voxel (41 with p stored as fp32).
// LDS row: [pad | left halo | 128 tile doubles | right halo | pad]: every access is a
// conflict-free 16-byte ds_read/ds_write_b128.
// PURE = true turns the kernel into the stand-alone stencil apply q = A v (v passed as
// `r`; no p/x traffic: 8 + 1 + 8 = 17 B/voxel), partial v.q.
// --------------------------------------------------------------------------------------

// LEAN = 3 (with PURE): the stand-alone apply for a CG vector, predicated on live lines like the CG kernels.
// LEAN (with PURE) turns the apply into the two stencil passes of the q-free CG iteration, the
// capacity mode that stores x, r, p only (25 B/voxel): LEAN = 1 reduces p.Ap without storing q;
// LEAN = 2 recomputes q = A p on the fly and applies x += alpha p, r -= alpha q (`q` argument = r
// vector), partials {r.r, r.z}.  Both take a CG vector and are predicated on the code byte.
template <int R, bool PURE, typename PT, int LEAN, bool COMPACT>
__global__ void __launch_bounds__(64 * kNW, (R <= 2 ? 4 : 2)) k_fused_a(Geom g, const uint8_t* __restrict__ code, double* __restrict__ x,
                                                       const double* __restrict__ r, const PT* __restrict__ p_in,
                                                       PT* __restrict__ p_out, double* __restrict__ q, int jacobi,
                                                       const double* __restrict__ hist, int k, int zchunk,
                                                       int tiles_x, int tiles_y, int nzc,
                                                       double* __restrict__ partials, const Ctrl* __restrict__ ctrl,
                                                       const int* __restrict__ lineoff, int LW) {
  constexpr int BY = kNW * R;
  static_assert(2 * BY <= 64, "wave 2 loads the 2*BY x-halo values with one lane each");
  if (ctrl && ctrl->done) return;
  __shared__ __attribute__((aligned(16))) double tile[2][(BY + 2) * kLP];
  __shared__ double red[kNW];
  // lineoff != null (CG form only): r (z), p_in, p_out and q live in the compacted line layout of k_cg1; x is dense

  // XCD-aware block -> tile map: see tile_of_block
  const int L = tile_of_block(blockIdx.x, tiles_x * tiles_y, nzc, g.tune.tile_map);
  const int tx = L % tiles_x, ty = (L / tiles_x) % tiles_y, zc = L / (tiles_x * tiles_y);

  static_assert(LEAN == 0 || PURE, "the lean passes are forms of the stand-alone apply");
  double beta = 0.0, alpha_prev = 0.0, alpha_now = 0.0;
  if (!PURE && k > 0) {
    beta = safe_div(hist[H_STRIDE * k + H_RZ], hist[H_STRIDE * (k - 1) + H_RZ]);
    alpha_prev = safe_div(hist[H_STRIDE * (k - 1) + H_RZ], hist[H_STRIDE * (k - 1) + H_PQ]);
  }
  if (LEAN == 2) alpha_now = safe_div(hist[H_STRIDE * k + H_RZ], hist[H_STRIDE * k + H_PQ]);

  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int i = tx * kBX + 2 * lane, j0 = ty * BY;
  const bool in_x = i < g.pitch;
  const int gs = 1 + zc * zchunk, ge = min(g.nzl + 1, gs + zchunk);  // output planes [gs, ge)

  // halo duty: wave 0 -> row j0-1, wave 1 -> row j0+BY (pairs); wave 2, lanes < 2*BY -> x halos
  // periodic x / y (g.periodic, even nx): the halo of the domain's first row / column is its last one and vice versa;
  // where the domain ends INSIDE a tile the wrapped row / column is staged right behind the last one
  const bool per = g.periodic != 0;
  const bool ywrap_hi = per && j0 + BY >= g.ny, xwrap_hi = per && tx * kBX + kBX >= g.nx;
  const int hrow = (w == 0) ? ((per && j0 == 0) ? g.ny - 1 : j0 - 1) : (ywrap_hi ? 0 : j0 + BY);
  const int hlds = (w == 0) ? 0 : (ywrap_hi ? g.ny - j0 + 1 : BY + 1);  // LDS row of this wave's halo row
  const bool yhalo = (w < 2) && in_x && hrow >= 0 && hrow < g.ny;
  const int xh_row = j0 + (lane >> 1);
  const int xh_i = (lane & 1) ? (xwrap_hi ? 0 : tx * kBX + kBX) : ((per && tx == 0) ? g.nx - 1 : tx * kBX - 1);
  const int xlds = (lane & 1) ? (xwrap_hi ? 2 + g.nx - tx * kBX : 2 + kBX) : 1;  // LDS column of this lane's halo value
  const bool xhalo = (w == 2) && lane < 2 * BY && xh_row < g.ny && xh_i >= 0 && xh_i < g.pitch;

  // CG vectors (z, p, q) vanish on non-free voxels: in the CG form their loads and stores are predicated
  // on the code byte, so cache lines without a free voxel are never moved.  The PURE apply takes any v
  // (the solution with its Dirichlet values included) and reads everything.
  auto pnew_at = [&](int gz, int jj, int ii) -> double {
    const long long e = (long long)gz * g.plane + (long long)jj * g.pitch + ii;
    if (PURE && LEAN == 0) return r[e];
    if (PURE) return code[e] ? r[e] : 0.0;
    const unsigned c = code[e];
    if (!c) return 0.0;
    const long long ev = COMPACT ? (long long)kLV * lineoff[((long long)gz * g.ny + jj) * LW + (ii >> kLVs)] + (ii & (kLV - 1)) : e;
    double z = r[ev];
    if (jacobi) z = z / (double)c;
    return PIo<PT>::rnd(z + beta * (double)p_in[ev]);
  };
  // p_new of plane gz for this thread's R pairs (and the p_in values for the x update)
  // code bytes of this thread's pairs in plane gz, fetched ONE PLANE AHEAD of the vectors: the predicate of
  // the vector loads is then in registers when they are issued (one memory latency per plane, not two)
  auto load_codes = [&](int gz, unsigned short (&cn)[R], int (&ln)[R]) {
#pragma unroll
    for (int rr_ = 0; rr_ < R; ++rr_) {
      const int j = j0 + w + rr_ * kNW;
      cn[rr_] = 0;
      ln[rr_] = -1;
      if (in_x && j < g.ny && gz <= g.nzl + 1) {
        cn[rr_] = *reinterpret_cast<const unsigned short*>(code + (long long)gz * g.plane + (long long)j * g.pitch + i);
        if (!PURE && COMPACT) ln[rr_] = lineoff[((long long)gz * g.ny + j) * LW + (i >> kLVs)];
      }
    }
  };
  // mode 0: a halo plane owned by another block (nothing stored); 1: an output plane of this block (p_k stored, x
  // updated at once: keeps p_{k-1} out of the register budget); 2: a ghost plane of the slab, kept current by the
  // block that marches over it (p_k stored)
  auto load_plane = [&](int gz, double2 (&pn)[R], const unsigned short (&cn)[R], const int (&ln)[R], bool (&lv)[R], int mode) {
#pragma unroll
    for (int rr_ = 0; rr_ < R; ++rr_) {
      const int j = j0 + w + rr_ * kNW;
      pn[rr_] = make_double2(0.0, 0.0);
      lv[rr_] = false;
      // (compacted layout: rows / pairs outside the domain have no slot, so the slot test is the only one needed)
      if ((!PURE && COMPACT) || (in_x && j < g.ny)) {
        const long long e = (long long)gz * g.plane + (long long)j * g.pitch + i;
        const unsigned short cc = cn[rr_];
        if (PURE && LEAN == 0) { pn[rr_] = ld2(r + e); continue; }
        // whole lines: every lane of a live line reads (and later writes).  Compacted vectors: the table says which
        // lines exist (kLV voxels); dense vectors: a 128-byte cache line is live when any of its lanes holds a free voxel
        const bool dense_live = (!PURE && COMPACT) ? false : line_live(cc != 0);
        lv[rr_] = (!PURE && COMPACT) ? ln[rr_] >= 0 : dense_live;
        if (!lv[rr_]) continue;
        if (PURE) { pn[rr_] = ld2(r + e); continue; }
        const long long ev = COMPACT ? (long long)kLV * ln[rr_] + (i & (kLV - 1)) : e;
        const double2 rv = ld2(r + ev);
        const double2 pin = PIo<PT>::ld_pair(p_in + ev);
        double z0 = rv.x, z1 = rv.y;
        if (jacobi) {
          if (cc & 0xff) z0 = rv.x / (double)(cc & 0xff);
          if (cc >> 8) z1 = rv.y / (double)(cc >> 8);
        }
        pn[rr_] = make_double2(PIo<PT>::rnd(z0 + beta * pin.x), PIo<PT>::rnd(z1 + beta * pin.y));
        if (mode != 0) {
          if (std::is_same<PT, double>::value && (g.tune.dense_nt & 1)) st2p(reinterpret_cast<double*>(p_out) + ev, pn[rr_].x, pn[rr_].y, true);
          else PIo<PT>::st_pair(p_out + ev, pn[rr_].x, pn[rr_].y);
          if (mode == 1 && k > 0) {
            const double2 xv = ld2(x + e);
            st2p(x + e, xv.x + alpha_prev * pin.x, xv.y + alpha_prev * pin.y, g.tune.dense_nt & 2);
          }
        }
      }
    }
  };
  unsigned hcode = 0;  // halo lanes: code byte(s) and slot of their halo voxel (pair) in the plane staged next
  int hslot = -1;
  // stage plane gz (own pair values) plus the halo ring into LDS buffer b
  auto stage = [&](int b, int gz, const double2 (&pn)[R]) {
    double* T = tile[b];
#pragma unroll
    for (int rr_ = 0; rr_ < R; ++rr_)
      if (!(per && (i == g.nx || j0 + w + rr_ * kNW == g.ny))) {  // (that slot holds the wrapped column / row)
        // odd nx, periodic: the pair (nx - 1, nx) straddles the seam -- its second slot holds the wrapped column 0 (wave 2)
        if (per && i + 1 == g.nx) T[(w + rr_ * kNW + 1) * kLP + 2 + 2 * lane] = pn[rr_].x;
        else *reinterpret_cast<double2*>(&T[(w + rr_ * kNW + 1) * kLP + 2 + 2 * lane]) = pn[rr_];
      }
    if (PURE && LEAN == 0) {  // any v: nothing decides the halo loads
      if (w < 2) {
        double2 h = make_double2(0.0, 0.0);
        if (yhalo) { h.x = pnew_at(gz, hrow, i); h.y = pnew_at(gz, hrow, i + 1); }
        *reinterpret_cast<double2*>(&T[hlds * kLP + 2 + 2 * lane]) = h;
      } else if (w == 2 && lane < 2 * BY) {
        double h = 0.0;
        if (xhalo) h = pnew_at(gz, xh_row, xh_i);
        T[((lane >> 1) + 1) * kLP + xlds] = h;
      }
      return;
    }
    // halo code bytes and slots were fetched a plane ahead (load_halo_codes, see k_cg1): the halo's vector loads go
    // out with no dependent look-up in front of them
    if (w < 2) {
      double2 h = make_double2(0.0, 0.0);
      if (hcode) {
        const long long e = (long long)gz * g.plane + (long long)hrow * g.pitch + i;
        if (PURE) {
          const double2 v = ld2(r + e);
          if (hcode & 0xff) h.x = v.x;
          if (hcode >> 8) h.y = v.y;
        } else {
          const long long ev = COMPACT ? (long long)kLV * hslot + (i & (kLV - 1)) : e;
          const double2 rv = ld2(r + ev);
          const double2 pv = PIo<PT>::ld_pair(p_in + ev);
          double z0 = rv.x, z1 = rv.y;
          if (jacobi) {
            if (hcode & 0xff) z0 = rv.x / (double)(hcode & 0xff);
            if (hcode >> 8) z1 = rv.y / (double)(hcode >> 8);
          }
          h.x = PIo<PT>::rnd(z0 + beta * pv.x); h.y = PIo<PT>::rnd(z1 + beta * pv.y);
        }
      }
      *reinterpret_cast<double2*>(&T[hlds * kLP + 2 + 2 * lane]) = h;
    } else if (w == 2 && lane < 2 * BY) {
      double h = 0.0;
      if (hcode) {
        const long long e = (long long)gz * g.plane + (long long)xh_row * g.pitch + xh_i;
        if (PURE) {
          h = r[e];
        } else {
          const long long ev = COMPACT ? (long long)kLV * hslot + (xh_i & (kLV - 1)) : e;
          double z = r[ev];
          if (jacobi) z = z / (double)hcode;
          h = PIo<PT>::rnd(z + beta * (double)p_in[ev]);
        }
      }
      T[((lane >> 1) + 1) * kLP + xlds] = h;
    }
  };
  auto load_halo_codes = [&](int gz) {  // of plane gz, for the stage() call of the NEXT iteration
    hcode = 0; hslot = -1;
    if ((PURE && LEAN == 0) || gz > g.nzl + 1) return;
    if (yhalo) {
      hcode = *reinterpret_cast<const unsigned short*>(code + (long long)gz * g.plane + (long long)hrow * g.pitch + i);
      if (!PURE && COMPACT) hslot = lineoff[((long long)gz * g.ny + hrow) * LW + (i >> kLVs)];
    } else if (xhalo) {
      hcode = code[(long long)gz * g.plane + (long long)xh_row * g.pitch + xh_i];
      if (!PURE && COMPACT) hslot = lineoff[((long long)gz * g.ny + xh_row) * LW + (xh_i >> kLVs)];
    }
  };

  double2 pn_prev[R], pn_cur[R], pn_nxt[R];
  unsigned short cc_cur[R], cc_nxt[R], cc_nn[R];
  int lo_cur[R], lo_nxt[R], lo_nn[R];
  bool lv_cur[R], lv_nxt[R];
  load_halo_codes(gs);
  load_codes(gs - 1, cc_nxt, lo_nxt);
  load_codes(gs, cc_cur, lo_cur);
  load_codes(gs + 1, cc_nn, lo_nn);
  load_plane(gs - 1, pn_prev, cc_nxt, lo_nxt, lv_nxt, gs == 1 ? 2 : 0);  // (lo ghost plane of p stays current)
  load_plane(gs, pn_cur, cc_cur, lo_cur, lv_cur, 1);
#pragma unroll
  for (int rr_ = 0; rr_ < R; ++rr_) { cc_nxt[rr_] = cc_nn[rr_]; lo_nxt[rr_] = lo_nn[rr_]; }  // plane gs + 1
  stage(gs & 1, gs, pn_cur);
  load_halo_codes(gs + 1);
  __syncthreads();

  double pq = 0.0, pz = 0.0;
  for (int gz = gs; gz < ge; ++gz) {
    load_codes(gz + 2, cc_nn, lo_nn);
    load_plane(gz + 1, pn_nxt, cc_nxt, lo_nxt, lv_nxt, gz + 1 < ge ? 1 : (ge == g.nzl + 1 ? 2 : 0));
    if (gz + 1 < ge) { stage((gz + 1) & 1, gz + 1, pn_nxt); load_halo_codes(gz + 2); }
    const double* T = tile[gz & 1];
#pragma unroll
    for (int rr_ = 0; rr_ < R; ++rr_) {
      const int j = j0 + w + rr_ * kNW;
      const unsigned short cc = cc_cur[rr_];
      if ((PURE && LEAN == 0) ? (in_x && j < g.ny) : lv_cur[rr_]) {
        const int base = (w + rr_ * kNW + 1) * kLP + 2 + 2 * lane;
        const double2 lf = *reinterpret_cast<const double2*>(&T[base - 2]);
        const double2 rt = *reinterpret_cast<const double2*>(&T[base + 2]);
        const double2 ym = *reinterpret_cast<const double2*>(&T[base - kLP]);
        const double2 yp = *reinterpret_cast<const double2*>(&T[base + kLP]);
        const long long e = (long long)gz * g.plane + (long long)j * g.pitch + i;
        const long long ev = (!PURE && COMPACT) ? (long long)kLV * lo_cur[rr_] + (i & (kLV - 1)) : e;  // q, p_out (x stays at e)
        const unsigned c0 = cc & 0xffu, c1 = cc >> 8;
        const double2 pc = pn_cur[rr_];
        // odd nx, periodic: the +x neighbour of voxel nx - 1 is the wrapped column 0, staged in the pair's second slot
        const double pxp = (per && i + 1 == g.nx) ? T[base + 1] : pc.y;
        const double s0 = ((((lf.y + pxp) + ym.x) + yp.x) + pn_prev[rr_].x) + pn_nxt[rr_].x;
        const double s1 = ((((pc.x + rt.x) + ym.y) + yp.y) + pn_prev[rr_].y) + pn_nxt[rr_].y;
        const double q0 = c0 ? (double)c0 * pc.x - s0 : 0.0;
        const double q1 = c1 ? (double)c1 * pc.y - s1 : 0.0;
        if (LEAN == 2) {
          const double2 rv = ld2(q + e), xv = ld2(x + e);
          const double r0 = rv.x - alpha_now * q0, r1 = rv.y - alpha_now * q1;
          st2(q + e, r0, r1);
          st2(x + e, xv.x + alpha_now * pc.x, xv.y + alpha_now * pc.y);
          double z0 = r0, z1 = r1;
          if (jacobi) {
            if (c0) z0 = r0 / (double)c0;
            if (c1) z1 = r1 / (double)c1;
          }
          pq += r0 * r0; pq += r1 * r1;
          pz += r0 * z0; pz += r1 * z1;
          continue;
        }
        if (LEAN == 0 || LEAN == 3) st2p(q + ev, q0, q1, g.tune.dense_nt & 4);
        pq += pc.x * q0; pq += pc.y * q1;
      }
    }
#pragma unroll
    for (int rr_ = 0; rr_ < R; ++rr_) {
      pn_prev[rr_] = pn_cur[rr_]; pn_cur[rr_] = pn_nxt[rr_]; cc_cur[rr_] = cc_nxt[rr_];
      cc_nxt[rr_] = cc_nn[rr_];
      lo_cur[rr_] = lo_nxt[rr_]; lo_nxt[rr_] = lo_nn[rr_];
      lv_cur[rr_] = lv_nxt[rr_];
    }
    __syncthreads();
  }
  const double t = block_sum(pq, red);
  if (LEAN == 2) {
    const double t2 = block_sum(pz, red);
    if (threadIdx.x == 0) { partials[2 * blockIdx.x] = t; partials[2 * blockIdx.x + 1] = t2; }
  } else if (threadIdx.x == 0) {
    partials[blockIdx.x] = t;
  }
}


template <int R, bool PURE, typename PT, int LEAN>
static void launch_tiled_r(int blocks, const Geom& g, const uint8_t* code, double* x, const double* r, const void* p_in,
                           void* p_out, double* q, int jacobi, const double* hist, int k, int zchunk, int tiles_x,
                           int tiles_y, int nzc, double* partials, const Ctrl* ctrl, const int* lineoff, hipStream_t s) {
#if OI_XCHECK  // (k_fused_a on compacted vectors: the cross-check of k_cg1b FORM 1, option fused_kernel = 1)
  if (lineoff && !PURE) {
    hipLaunchKernelGGL((k_fused_a<R, PURE, PT, LEAN, !PURE>), dim3(blocks), dim3(64 * kNW), 0, s, g, code, x, r, (const PT*)p_in,
                       (PT*)p_out, q, jacobi, hist, k, zchunk, tiles_x, tiles_y, nzc, partials, ctrl, lineoff,
                       (g.pitch + kLV - 1) / kLV);
    return;
  }
#endif
    hipLaunchKernelGGL((k_fused_a<R, PURE, PT, LEAN, false>), dim3(blocks), dim3(64 * kNW), 0, s, g, code, x, r, (const PT*)p_in,
                       (PT*)p_out, q, jacobi, hist, k, zchunk, tiles_x, tiles_y, nzc, partials, ctrl, nullptr,
                       (g.pitch + kLV - 1) / kLV);
}

template <bool PURE, typename PT, int LEAN = 0>
static int launch_tiled(const Geom& g, const uint8_t* code, double* x, const double* r, const void* p_in, void* p_out,
                        double* q, int jacobi, const double* hist, int k, double* partials, const Ctrl* ctrl,
                        hipStream_t s, const int* lineoff = nullptr) {
  // the predicated PURE forms (capacity-mode passes, apply on a CG vector) run best on 16-row tiles (R = 2: 99 VGPRs,
  // 4 waves/SIMD; 10.1 vs 13.5 ms per lean iteration at 1024^3), the unpredicated stand-alone apply on 32-row tiles
  const int R = PURE ? (LEAN != 0 && g.tune.pure_rows > 2 ? 2 : g.tune.pure_rows) : g.tune.fused_rows, by = kNW * R;
  const int tiles_x = (g.pitch + kBX - 1) / kBX, tiles_y = (g.ny + by - 1) / by;
  const long long cols = (long long)tiles_x * tiles_y;
  int zchunk, nzc;
  pick_chunks(g, cols, false, &zchunk, &nzc);
  if (!grid_fits(g, cols, nzc)) return -1;
  const int blocks = (int)(cols * nzc);
  if (R == 4)
    launch_tiled_r<4, PURE, PT, LEAN>(blocks, g, code, x, r, p_in, p_out, q, jacobi, hist, k, zchunk, tiles_x, tiles_y,
                                      nzc, partials, ctrl, lineoff, s);
  else if (R == 2)
    launch_tiled_r<2, PURE, PT, LEAN>(blocks, g, code, x, r, p_in, p_out, q, jacobi, hist, k, zchunk, tiles_x, tiles_y,
                                      nzc, partials, ctrl, lineoff, s);
  else
    launch_tiled_r<1, PURE, PT, LEAN>(blocks, g, code, x, r, p_in, p_out, q, jacobi, hist, k, zchunk, tiles_x, tiles_y,
                                      nzc, partials, ctrl, lineoff, s);
  return blocks;
}

// --------------------------------------------------------------------------------------
// Stand-alone masked 7-point apply q = A v, partial v.q -- the kernel the "fraction of the HBM roofline" target of the
// north star names (17 B/voxel: v 8 + code 1 + q 8).  Barrier-free and LDS-free: one WAVE owns 128 (x) x RY (y)
// voxels and marches over a short z chunk.
//   * x neighbours: DPP wave shifts (wave_shr:1 / wave_shl:1) of the lane's own voxel pair; the two values a row needs
//     from the neighbouring tiles are fetched by 2 RY lanes with ONE 8-byte load instruction per plane and handed to
//     lanes 0 / 63 as the shift's fill value (v_readlane);
//   * y neighbours: the wave's own rows plus one halo row above and below (served by L2: the neighbouring wave reads
//     them as its centre rows at about the same time);
//   * z neighbours: a register ring of D + 2 planes, D planes loaded ahead of the one computed;
//   * addresses: one 32-bit lane offset against wave-uniform (scalar) row bases (global_load ... v_off, s[base:base+1]),
//     so a plane costs a handful of scalar adds and no 64-bit vector arithmetic;
//   * q leaves with nontemporal stores (read once by the next kernel; keeps v's halo rows in L2).
// Launch order is the point (tile_of_block, map 2; z chunks of 8 planes): the whole chip sweeps through z like a linear
// stream.  1024^3 on one MI355X: 3.00 ms = 6.1 TB/s = 0.76 of the 8 TB/s peak, at the best linear float4 copy of the
// same box (3.02 ms for 16 B/voxel); the LDS-tiled k_fused_a<PURE> with 64-plane chunks and map 0: 3.6-3.7 ms
// (profiles/r02/apply_proto_*.txt).  Summation order is the oracle's, results are bit-identical to k_apply / k_fused_a.

// the LDS-tiled first kernel of the two-kernel iteration with the launch shape its caller (launch_fused_a, oi_k_cg1b.hip) picked:
// blocks > 0 = 16-row tiles on the chunks of fused_chunks; blocks = 0 = the shape of launch_tiled (other tile heights, periodic
// problems on dense vectors).  Compacted vectors (lineoff) run on this kernel in the test library only: the product takes the
// descriptor-addressed form there (k_cg1b FORM 1) and never builds line lists it could not serve.
int launch_fused_a_lds(const Geom& g, const uint8_t* code, double* x, const double* r, const void* p_in, void* p_out, int p_f32, double* q,
                       int jacobi, const double* hist, int k, double* partials, const Ctrl* ctrl, const int* lineoff, hipStream_t s,
                       int blocks, int zchunk, int tiles_x, int tiles_y, int nzc) {
#if !OI_XCHECK
  if (lineoff) return -2;
#endif
  if (blocks > 0) {
    if (p_f32)
      launch_tiled_r<2, false, float, 0>(blocks, g, code, x, r, p_in, p_out, q, jacobi, hist, k, zchunk, tiles_x, tiles_y, nzc, partials, ctrl, lineoff, s);
    else
      launch_tiled_r<2, false, double, 0>(blocks, g, code, x, r, p_in, p_out, q, jacobi, hist, k, zchunk, tiles_x, tiles_y, nzc, partials, ctrl, lineoff, s);
    return blocks;
  }
  if (p_f32) return launch_tiled<false, float>(g, code, x, r, p_in, p_out, q, jacobi, hist, k, partials, ctrl, s, lineoff);
  return launch_tiled<false, double>(g, code, x, r, p_in, p_out, q, jacobi, hist, k, partials, ctrl, s, lineoff);
}

// --------------------------------------------------------------------------------------
typedef double v2d_t __attribute__((ext_vector_type(2)));

// shr1: lane l gets the value of lane l - 1, lane 0 gets `fill`;  shl1: lane l gets lane l + 1, lane 63 gets `fill`
__device__ __forceinline__ double dpp_shr1(double v, double fill) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(__double2loint(fill), lo, 0x138, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(__double2hiint(fill), hi, 0x138, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double dpp_shl1(double v, double fill) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(__double2loint(fill), lo, 0x130, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(__double2hiint(fill), hi, 0x130, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_bcast(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}

template <int RY>
struct ApplyPlane {
  v2d_t c[RY + 2];        // rows j0 - 1 .. j0 + RY of this lane's voxel pair
  double xh;              // lanes < 2 RY: x halo of row (lane >> 1), side (lane & 1)
  unsigned short cc[RY];  // code bytes of the centre rows
};

template <int RY, int NW, int D>
__global__ void __launch_bounds__(64 * NW) k_apply7(Geom g, const uint8_t* __restrict__ code, const double* __restrict__ v,
                                                    double* __restrict__ q, double* __restrict__ partials,
                                                    const Ctrl* __restrict__ ctrl, int zchunk, int tiles_x, int tiles_y,
                                                    int nzc) {
  constexpr int RING = D + 2;
  if (ctrl && ctrl->done) return;
  __shared__ double red[NW];
  const int L = tile_of_block(blockIdx.x, tiles_x * tiles_y, nzc, g.tune.tile_map);
  const int tx = L % tiles_x, ty = (L / tiles_x) % tiles_y, zc = L / (tiles_x * tiles_y);
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i0 = tx * kBX, j0 = (ty * NW + w) * RY;
  const bool in_x = i0 + 2 * lane < g.pitch;
  const int gs = 1 + zc * zchunk, ge = min(g.nzl + 1, gs + zchunk);  // output planes [gs, ge)

  const int hrow = j0 + (lane >> 1);
  const int hi_ = (lane & 1) ? i0 + kBX : i0 - 1;
  const bool hvalid = lane < 2 * RY && hrow < g.ny && hi_ >= 0 && hi_ < g.pitch;  // (columns >= nx hold zeros)

  const unsigned lane_b16 = 16u * lane, lane_b2 = 2u * lane;
  const long long urow = (long long)j0 * g.pitch + i0;  // wave-uniform: first voxel of the wave's row j0
  const unsigned hoff_b = hvalid ? (unsigned)(((lane >> 1) * g.pitch + (hi_ - i0 + 1)) * 8) : 0u;  // from (row j0, i0 - 1)

  ApplyPlane<RY> P[RING];
  // with_halo = false: a plane that only serves as z neighbour (centre rows, no halo, no codes)
  auto load_plane = [&](ApplyPlane<RY>& p, int gz, bool with_halo) {
    const char* vp = reinterpret_cast<const char*>(v + (long long)gz * g.plane + urow);
#pragma unroll
    for (int r = 0; r < RY + 2; ++r) {
      const int j = j0 + r - 1;
      const bool edge = (r == 0 || r == RY + 1);
      v2d_t val = {0.0, 0.0};
      if ((!edge || with_halo) && j >= 0 && j < g.ny && in_x)
        val = *reinterpret_cast<const v2d_t*>(vp + (long long)(r - 1) * g.pitch * 8 + lane_b16);
      p.c[r] = val;
    }
    double h = 0.0;
    if (with_halo && hvalid) h = *reinterpret_cast<const double*>(vp - 8 + hoff_b);
    p.xh = h;
    if (with_halo) {
      const char* cp = reinterpret_cast<const char*>(code + (long long)gz * g.plane + urow);
#pragma unroll
      for (int r = 0; r < RY; ++r) {
        unsigned short c = 0;
        if (j0 + r < g.ny && in_x) c = *reinterpret_cast<const unsigned short*>(cp + (long long)r * g.pitch + lane_b2);
        p.cc[r] = c;
      }
    }
  };

  double pq = 0.0;
  auto compute = [&](int gz, const ApplyPlane<RY>& pv, const ApplyPlane<RY>& cu, const ApplyPlane<RY>& nx) {
    char* qp = reinterpret_cast<char*>(q + (long long)gz * g.plane + urow);
#pragma unroll
    for (int r = 0; r < RY; ++r) {
      const v2d_t pc = cu.c[r + 1], ym = cu.c[r], yp = cu.c[r + 2];
      const double lf = dpp_shr1(pc.y, lane_bcast(cu.xh, 2 * r)), rt = dpp_shl1(pc.x, lane_bcast(cu.xh, 2 * r + 1));
      const unsigned c0 = cu.cc[r] & 0xffu, c1 = cu.cc[r] >> 8;
      const double s0 = ((((lf + pc.y) + ym.x) + yp.x) + pv.c[r + 1].x) + nx.c[r + 1].x;
      const double s1 = ((((pc.x + rt) + ym.y) + yp.y) + pv.c[r + 1].y) + nx.c[r + 1].y;
      const double q0 = c0 ? (double)c0 * pc.x - s0 : 0.0;
      const double q1 = c1 ? (double)c1 * pc.y - s1 : 0.0;
      if (j0 + r < g.ny && in_x) {
        const v2d_t o = {q0, q1};
        __builtin_nontemporal_store(o, reinterpret_cast<v2d_t*>(qp + (long long)r * g.pitch * 8 + lane_b16));
      }
      pq += pc.x * q0;
      pq += pc.y * q1;
    }
  };

  load_plane(P[0], gs - 1, false);
#pragma unroll
  for (int d = 0; d < D; ++d)
    if (gs + d <= ge) load_plane(P[1 + d], gs + d, gs + d < ge);
  for (int gz0 = gs; gz0 < ge; gz0 += RING) {
#pragma unroll
    for (int u = 0; u < RING; ++u) {
      const int gz = gz0 + u;
      if (gz < ge) {
        const int ahead = gz + D;
        if (ahead <= ge) load_plane(P[(u + 1 + D) % RING], ahead, ahead < ge);
        compute(gz, P[u % RING], P[(u + 1) % RING], P[(u + 2) % RING]);
      }
    }
  }
  pq = wave_sum(pq);
  if (lane == 0) red[w] = pq;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int k = 0; k < NW; ++k) t += red[k];
    partials[blockIdx.x] = t;
  }
}

static int launch_apply7(const Geom& g, const uint8_t* code, const double* v, double* q, double* partials, const Ctrl* ctrl,
                         hipStream_t s) {
  constexpr int RY = 4;
  const bool tall = g.ny > 16 * 8;  // 8-wave blocks (32 rows) on tall planes, 4-wave blocks otherwise
  const int by = RY * (tall ? 8 : 4);
  const int tiles_x = (g.pitch + kBX - 1) / kBX, tiles_y = (g.ny + by - 1) / by;
  const long long cols = (long long)tiles_x * tiles_y;
  Geom gt = g;
  if (gt.tune.zchunk <= 0) gt.tune.zchunk = kMinZChunk;  // short chunks: the launch order makes the march a z sweep
  int zchunk, nzc;
  pick_chunks(gt, cols, false, &zchunk, &nzc);
  if (!grid_fits(g, cols, nzc)) return -1;
  const int blocks = (int)(cols * nzc);
  if (tall)
    hipLaunchKernelGGL((k_apply7<RY, 8, 2>), dim3(blocks), dim3(512), 0, s, g, code, v, q, partials, ctrl, zchunk, tiles_x, tiles_y, nzc);
  else
    hipLaunchKernelGGL((k_apply7<RY, 4, 2>), dim3(blocks), dim3(256), 0, s, g, code, v, q, partials, ctrl, zchunk, tiles_x, tiles_y, nzc);
  return blocks;
}

// --------------------------------------------------------------------------------------
// Reference stream of the box: a linear copy, every wave moving contiguous 8 KiB chunks with eight 16-byte nontemporal
// loads in flight per lane and nontemporal stores -- the fastest of the copy forms tried on MI355X
// (profiles/r02/apply_proto_*_copy_ceiling*.txt: 5.6-5.7 TB/s; grid-stride and plain-policy forms 4.8-5.4).  bench.py
// reports it beside the stencil apply: what a 1 read : 1 write stream reaches on THIS box at the same moment.
// --------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_stream_copy(double* __restrict__ d, const double* __restrict__ s, long long n2) {
  constexpr int U = 8;
  const int lane = threadIdx.x & 63;
  const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long long)gridDim.x * 4;
  const long long nchunks = n2 / (64 * U);
  for (long long c = wave; c < nchunks; c += nwaves) {
    const long long t0 = c * 64 * U + lane;
    v2d_t v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(reinterpret_cast<const v2d_t*>(s + 2 * (t0 + u * 64)));
#pragma unroll
    for (int u = 0; u < U; ++u) __builtin_nontemporal_store(v[u], reinterpret_cast<v2d_t*>(d + 2 * (t0 + u * 64)));
  }
  for (long long t = nchunks * 64 * U + (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n2; t += (long long)gridDim.x * blockDim.x)
    *reinterpret_cast<v2d_t*>(d + 2 * t) = *reinterpret_cast<const v2d_t*>(s + 2 * t);
}
// the same walk over three sources and three destinations: what the memory system gives a kernel that moves the six
// double-buffered vectors of the single-kernel CG and does nothing else (d_j = s_j; the CG kernel's own ceiling is below
// this: it also streams x, the code bytes and the line table)
__global__ void __launch_bounds__(256) k_stream_3r3w(double* __restrict__ d0, double* __restrict__ d1, double* __restrict__ d2,
                                                     const double* __restrict__ s0, const double* __restrict__ s1,
                                                     const double* __restrict__ s2, long long n2) {
  constexpr int U = 4;
  const int lane = threadIdx.x & 63;
  const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long long)gridDim.x * 4;
  const long long nchunks = n2 / (64 * U);
  for (long long c = wave; c < nchunks; c += nwaves) {
    const long long t0 = c * 64 * U + lane;
    v2d_t a[U], b[U], e[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      a[u] = __builtin_nontemporal_load(reinterpret_cast<const v2d_t*>(s0 + 2 * (t0 + u * 64)));
      b[u] = __builtin_nontemporal_load(reinterpret_cast<const v2d_t*>(s1 + 2 * (t0 + u * 64)));
      e[u] = __builtin_nontemporal_load(reinterpret_cast<const v2d_t*>(s2 + 2 * (t0 + u * 64)));
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      __builtin_nontemporal_store(a[u], reinterpret_cast<v2d_t*>(d0 + 2 * (t0 + u * 64)));
      __builtin_nontemporal_store(b[u], reinterpret_cast<v2d_t*>(d1 + 2 * (t0 + u * 64)));
      __builtin_nontemporal_store(e[u], reinterpret_cast<v2d_t*>(d2 + 2 * (t0 + u * 64)));
    }
  }
}
void launch_stream_3r3w(double* d0, double* d1, double* d2, const double* s0, const double* s1, const double* s2,
                        long long ndoubles, hipStream_t s) {
  hipLaunchKernelGGL(k_stream_3r3w, dim3(8192), dim3(256), 0, s, d0, d1, d2, s0, s1, s2, ndoubles / 2);
}
void launch_stream_copy(double* dst, const double* src, long long ndoubles, hipStream_t s) {
  hipLaunchKernelGGL(k_stream_copy, dim3(8192), dim3(256), 0, s, dst, src, ndoubles / 2);
}

// q = A v for a CG vector (v vanishes on non-free voxels): the stand-alone apply with live-line predication
int launch_apply_cgvec(const Geom& g, const uint8_t* code, const double* v, double* q, double* partials, const Ctrl* ctrl,
                       hipStream_t s) {
  return launch_tiled<true, double, 3>(g, code, nullptr, v, nullptr, nullptr, q, 0, nullptr, 0, partials, ctrl, s);
}
// q-free CG passes (capacity mode): p.Ap, then x += alpha p, r -= alpha A p with partials {rr, rz}
int launch_lean_dot(const Geom& g, const uint8_t* code, const double* p, double* partials, const Ctrl* ctrl,
                    hipStream_t s) {
  return launch_tiled<true, double, 1>(g, code, nullptr, p, nullptr, nullptr, nullptr, 0, nullptr, 0, partials, ctrl, s);
}
int launch_lean_update(const Geom& g, const uint8_t* code, const double* p, double* x, double* r, int jacobi,
                       const double* hist, int k, double* partials, const Ctrl* ctrl, hipStream_t s) {
  return launch_tiled<true, double, 2>(g, code, x, p, nullptr, nullptr, r, jacobi, hist, k, partials, ctrl, s);
}

int launch_apply_tiled(const Geom& g, const uint8_t* code, const double* v, double* q, double* partials,
                       const Ctrl* ctrl, hipStream_t s) {
  if (g.tune.apply_kernel == 0 && !g.periodic) return launch_apply7(g, code, v, q, partials, ctrl, s);
  return launch_tiled<true, double>(g, code, nullptr, v, nullptr, nullptr, q, 0, nullptr, 0, partials, ctrl, s);
}

// --------------------------------------------------------------------------------------
// Fine-level halves of the multigrid V-cycle on the same z-marching LDS tile as the stencil
// apply (17-18 B/voxel each instead of seven cache-served loads per voxel):
//   MODE 1 (pre) :  v = w b / code                         out = b - A v
//   MODE 2 (post):  v = w b / code + s * xc[parent]        out = v + w (b - A v) / code,
//                   partial out.b  (= z.r of the preconditioned CG)
//   MODE 3 (pre + restriction): the residual of MODE 1 is never stored; its 2x2x2 block sums go
//                   straight into the fp32 right-hand side of the first coarse level (x pairs in the
//                   thread, y pairs through LDS between neighbouring waves, z pairs across two
//                   consecutive planes of the march) -- 9.5 B/voxel instead of 17 + 8.5
// b vanishes on non-free cells and its ghost planes hold the neighbour slabs' values; xc is the
// fp32 coarse correction of the 2x2x2 aggregates.
// --------------------------------------------------------------------------------------

// COMPACT: b and out live in the compacted line layout (lineoff != null); a compile-time switch, so that neither address
// form is computed just to be discarded by a select
template <int R, int MODE, bool COMPACT>
__global__ void __launch_bounds__(64 * kNW, (R <= 2 ? 4 : 2)) k_mg_fine(Geom g, const uint8_t* __restrict__ code,
                                                       const double* __restrict__ b, MgCoarse C, double omega, double scale,
                                                       double* __restrict__ out, float* __restrict__ bcoarse, int zchunk,
                                                       int tiles_x, int tiles_y, int nzc, double* __restrict__ partials,
                                                       const Ctrl* __restrict__ ctrl, const int* __restrict__ lineoff, int LW) {
  constexpr int BY = kNW * R;
  static_assert(2 * BY <= 64, "wave 2 loads the 2*BY x-halo values with one lane each");
  static_assert((kNW & 1) == 0, "rows 2m and 2m+1 of a tile belong to waves w and w+1");
  // lineoff != null: b and out live in the compacted line layout of k_cg1
  if (ctrl && ctrl->done) return;
  __shared__ __attribute__((aligned(16))) double tile[2][(BY + 2) * kLP];
  __shared__ double red[kNW];
  __shared__ double ysum[MODE == 3 ? 2 : 1][MODE == 3 ? R * (kNW / 2) * 64 : 1];
  __shared__ double wtab[8];  // omega / c for the neighbour counts c = 1..6 (0 for c = 0): one LDS read replaces a division
  if (threadIdx.x < 8) wtab[threadIdx.x] = threadIdx.x ? omega / (double)threadIdx.x : 0.0;
  __syncthreads();
  const int L = tile_of_block(blockIdx.x, tiles_x * tiles_y, nzc, g.tune.tile_map);
  const int tx = L % tiles_x, ty = (L / tiles_x) % tiles_y, zc = L / (tiles_x * tiles_y);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int i = tx * kBX + 2 * lane, j0 = ty * BY;
  const bool in_x = i < g.pitch;
  const int gs = 1 + zc * zchunk, ge = min(g.nzl + 1, gs + zchunk);
  // periodic x / y (g.periodic, even nx): the halo of the domain's first row / column is its last one and vice versa;
  // where the domain ends INSIDE a tile the wrapped row / column is staged right behind the last one
  const bool per = g.periodic != 0;
  const bool ywrap_hi = per && j0 + BY >= g.ny, xwrap_hi = per && tx * kBX + kBX >= g.nx;
  const int hrow = (w == 0) ? ((per && j0 == 0) ? g.ny - 1 : j0 - 1) : (ywrap_hi ? 0 : j0 + BY);
  const int hlds = (w == 0) ? 0 : (ywrap_hi ? g.ny - j0 + 1 : BY + 1);  // LDS row of this wave's halo row
  const bool yhalo = (w < 2) && in_x && hrow >= 0 && hrow < g.ny;
  const int xh_row = j0 + (lane >> 1);
  const int xh_i = (lane & 1) ? (xwrap_hi ? 0 : tx * kBX + kBX) : ((per && tx == 0) ? g.nx - 1 : tx * kBX - 1);
  const int xlds = (lane & 1) ? (xwrap_hi ? 2 + g.nx - tx * kBX : 2 + kBX) : 1;  // LDS column of this lane's halo value
  const bool xhalo = (w == 2) && lane < 2 * BY && xh_row < g.ny && xh_i >= 0 && xh_i < g.pitch;

  // v at one voxel (the pad column and the ghost planes beyond the domain hold code 0, so the coarse
  // correction is never indexed for them; ghost planes of a neighbour slab map to its ghost coarse plane)
  // scaled coarse correction of the aggregate that holds voxel (ii, jj, gz)
  // coarse plane (local, ghosts 0 and cnzl + 1) of the aggregate that holds fine plane gz; periodic problem: the planes
  // beyond the two ends are images whose parents sit in the coarse GHOST planes (an odd nz does not give that by itself)
  auto cplane_of = [&](int gz) -> int {
    const int kg = g.z0 + gz - 1;
    if (C.periodic && kg >= g.nz) return C.cnzl + 1;
    if (C.periodic && kg < 0) return 0;
    return (kg >> 1) - C.z0c + 1;
  };
  auto corr_at = [&](int ii, int jj, int gz) -> double {
    const long long cm = (long long)(ii >> 1) + (long long)C.cnx * (jj >> 1) + C.csz * cplane_of(gz);
    float cv = C.xc[cm];
    if (C.xc2) cv += C.xc2[cm];
    return scale * (double)cv;
  };
  // code bytes one plane ahead of the vectors (see k_fused_a)
  auto load_codes = [&](int gz, unsigned short (&cn)[R], int (&ln)[R]) {
#pragma unroll
    for (int rr_ = 0; rr_ < R; ++rr_) {
      const int j = j0 + w + rr_ * kNW;
      cn[rr_] = 0;
      ln[rr_] = -1;
      if (in_x && j < g.ny && gz <= g.nzl + 1) {
        cn[rr_] = *reinterpret_cast<const unsigned short*>(code + (long long)gz * g.plane + (long long)j * g.pitch + i);
        if (COMPACT) ln[rr_] = lineoff[((long long)gz * g.ny + j) * LW + (i >> kLVs)];
      }
    }
  };
  // The plane's loads are issued ONE ITERATION before they are used (issue_plane -> braw / craw, finish_plane a
  // barrier later): the kernel moves few bytes per plane and is latency-bound, two blocks per CU do not hide a load.
  double2 braw[R];
  float craw[R];
  auto issue_plane = [&](int gz, const unsigned short (&cn)[R], const int (&ln)[R]) {
#pragma unroll
    for (int rr_ = 0; rr_ < R; ++rr_) {
      const int j = j0 + w + rr_ * kNW;
      braw[rr_] = make_double2(0.0, 0.0);
      craw[rr_] = 0.f;
      // (compacted layout: rows / pairs outside the domain have no slot, so the slot test is the only one needed)
      if (COMPACT || (in_x && j < g.ny)) {
        const long long e = (long long)gz * g.plane + (long long)j * g.pitch + i;
        const unsigned short cc = cn[rr_];
        const bool dense_live = COMPACT ? false : line_live(cc != 0);
        if (COMPACT ? ln[rr_] >= 0 : dense_live) {  // b and the output vanish on non-free voxels: lines without one are not moved
          braw[rr_] = ld2(b + (COMPACT ? (long long)kLV * ln[rr_] + (i & (kLV - 1)) : e));
          if (MODE == 2 && cc) {  // voxels i and i + 1 share their aggregate (i is even)
            const long long cm = (long long)(i >> 1) + (long long)C.cnx * (j >> 1) + C.csz * cplane_of(gz);
            float cv = C.xc[cm];
            if (C.xc2) cv += C.xc2[cm];
            craw[rr_] = cv;
          }
        }
      }
    }
  };
  auto finish_plane = [&](const unsigned short (&cn)[R], double2 (&vn)[R], double2 (&bn)[R]) {
#pragma unroll
    for (int rr_ = 0; rr_ < R; ++rr_) {
      const unsigned short cc = cn[rr_];
      const double2 bv = braw[rr_];
      bn[rr_] = bv;
      double v0 = bv.x * wtab[cc & 0xffu], v1 = bv.y * wtab[cc >> 8];
      if (MODE == 2 && cc) {
        const double cr = scale * (double)craw[rr_];
        if (cc & 0xffu) v0 += cr;
        if (cc >> 8) v1 += cr;
      }
      vn[rr_] = make_double2(v0, v1);
    }
  };
  unsigned hcode = 0;  // halo lanes: code byte(s) and slot of their halo voxel (pair) in the plane staged next
  int hslot = -1;
  auto stage = [&](int bsel, int gz, const double2 (&vn)[R]) {
    double* T = tile[bsel];
#pragma unroll
    for (int rr_ = 0; rr_ < R; ++rr_)
      if (!(per && (i == g.nx || j0 + w + rr_ * kNW == g.ny))) {  // (that slot holds the wrapped column / row)
        if (per && i + 1 == g.nx) T[(w + rr_ * kNW + 1) * kLP + 2 + 2 * lane] = vn[rr_].x;  // (odd nx: see k_fused_a)
        else *reinterpret_cast<double2*>(&T[(w + rr_ * kNW + 1) * kLP + 2 + 2 * lane]) = vn[rr_];
      }
    // halo code bytes and slots were fetched a plane ahead (load_halo_codes, see k_cg1)
    if (w < 2) {
      double2 h = make_double2(0.0, 0.0);
      if (hcode) {
        const long long ev = COMPACT ? (long long)kLV * hslot + (i & (kLV - 1)) : (long long)gz * g.plane + (long long)hrow * g.pitch + i;
        const double2 bv = ld2(b + ev);
        const double cr = MODE == 2 ? corr_at(i, hrow, gz) : 0.0;  // voxels i and i + 1 share their aggregate
        if (hcode & 0xff) h.x = bv.x * wtab[hcode & 0xff] + cr;
        if (hcode >> 8) h.y = bv.y * wtab[hcode >> 8] + cr;
      }
      *reinterpret_cast<double2*>(&T[hlds * kLP + 2 + 2 * lane]) = h;
    } else if (w == 2 && lane < 2 * BY) {
      double h = 0.0;
      if (hcode) {
        const long long ev = COMPACT ? (long long)kLV * hslot + (xh_i & (kLV - 1)) : (long long)gz * g.plane + (long long)xh_row * g.pitch + xh_i;
        h = b[ev] * wtab[hcode];
        if (MODE == 2) h += corr_at(xh_i, xh_row, gz);
      }
      T[((lane >> 1) + 1) * kLP + xlds] = h;
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

  // MODE 3: even waves own the block sums; `own` is their x-pair sum of the plane just finished, `acc`
  // the running 2x2x(1 or 2) sum.  Plane gzp is consumed one barrier after it was computed.
  double own[R], acc[R];
#pragma unroll
  for (int rr_ = 0; rr_ < R; ++rr_) { own[rr_] = 0.0; acc[rr_] = 0.0; }
  auto consume = [&](int gzp) {
    if (MODE != 3 || (w & 1)) return;
    const double* Y = ysum[MODE == 3 ? (gzp & 1) : 0];
    const int kg = g.z0 + gzp - 1;  // global fine plane
    const bool store = (kg & 1) != 0 || gzp == ge - 1;
#pragma unroll
    for (int rr_ = 0; rr_ < R; ++rr_) {
      acc[rr_] += own[rr_] + Y[(rr_ * (kNW / 2) + (w >> 1)) * 64 + lane];
      if (store) {
        const int I = tx * (kBX / 2) + lane, J = (j0 + w + rr_ * kNW) >> 1;
        if (I < C.cnx && J < C.cny) bcoarse[(long long)I + (long long)C.cnx * J + C.csz * ((kg >> 1) - C.z0c + 1)] = (float)acc[rr_];
        acc[rr_] = 0.0;
      }
    }
  };

  double2 v_prev[R], v_cur[R], v_nxt[R], b_cur[R], b_nxt[R];
  unsigned short cc_cur[R], cc_nxt[R], cc_nn[R], cc_n3[R];
  int lo_cur[R], lo_nxt[R], lo_nn[R], lo_n3[R];
  load_halo_codes(gs);
  load_codes(gs - 1, cc_n3, lo_n3);
  load_codes(gs, cc_cur, lo_cur);
  load_codes(gs + 1, cc_nxt, lo_nxt);
  load_codes(gs + 2, cc_nn, lo_nn);
  issue_plane(gs - 1, cc_n3, lo_n3);
  finish_plane(cc_n3, v_prev, b_nxt);
  issue_plane(gs, cc_cur, lo_cur);
  finish_plane(cc_cur, v_cur, b_cur);
  stage(gs & 1, gs, v_cur);
  load_halo_codes(gs + 1);
  issue_plane(gs + 1, cc_nxt, lo_nxt);  // used by the first iteration
  __syncthreads();
  double dot = 0.0;
  for (int gz = gs; gz < ge; ++gz) {
    load_codes(gz + 3, cc_n3, lo_n3);
    finish_plane(cc_nxt, v_nxt, b_nxt);       // plane gz + 1, requested one iteration ago
    issue_plane(gz + 2, cc_nn, lo_nn);        // plane gz + 2, for the next iteration
    if (gz + 1 < ge) { stage((gz + 1) & 1, gz + 1, v_nxt); load_halo_codes(gz + 2); }
    if (gz > gs) consume(gz - 1);
    const double* T = tile[gz & 1];
    double xsum[R];
#pragma unroll
    for (int rr_ = 0; rr_ < R; ++rr_) {
      const int j = j0 + w + rr_ * kNW;
      xsum[rr_] = 0.0;
      if (COMPACT || (in_x && j < g.ny)) {
        const unsigned short cc = cc_cur[rr_];
        const bool dense_live = COMPACT ? false : line_live(cc != 0);
        if (!(COMPACT ? lo_cur[rr_] >= 0 : dense_live)) continue;  // (live lines are written whole; dead ones read no LDS)
        const int base = (w + rr_ * kNW + 1) * kLP + 2 + 2 * lane;
        const double2 lf = *reinterpret_cast<const double2*>(&T[base - 2]);
        const double2 rt = *reinterpret_cast<const double2*>(&T[base + 2]);
        const double2 ym = *reinterpret_cast<const double2*>(&T[base - kLP]);
        const double2 yp = *reinterpret_cast<const double2*>(&T[base + kLP]);
        const long long e = (long long)gz * g.plane + (long long)j * g.pitch + i;
        const unsigned c0 = cc & 0xffu, c1 = cc >> 8;
        const double2 vc = v_cur[rr_], bc = b_cur[rr_];
        const double vxp = (per && i + 1 == g.nx) ? T[base + 1] : vc.y;  // (odd nx, periodic: see k_fused_a)
        const double s0 = ((((lf.y + vxp) + ym.x) + yp.x) + v_prev[rr_].x) + v_nxt[rr_].x;
        const double s1 = ((((vc.x + rt.x) + ym.y) + yp.y) + v_prev[rr_].y) + v_nxt[rr_].y;
        double o0 = 0.0, o1 = 0.0;
        if (c0) { const double av = (double)c0 * vc.x - s0; o0 = MODE != 2 ? bc.x - av : vc.x + (bc.x - av) * wtab[c0]; }
        if (c1) { const double av = (double)c1 * vc.y - s1; o1 = MODE != 2 ? bc.y - av : vc.y + (bc.y - av) * wtab[c1]; }
        if (MODE == 3) { xsum[rr_] = o0 + o1; continue; }
        st2p(out + (COMPACT ? (long long)kLV * lo_cur[rr_] + (i & (kLV - 1)) : e), o0, o1, g.tune.dense_nt & 16);
        dot += o0 * bc.x; dot += o1 * bc.y;
      }
    }
    if (MODE == 3) {  // odd waves publish their x-pair sums, even waves keep theirs until the barrier has passed
      double* Y = ysum[MODE == 3 ? (gz & 1) : 0];
#pragma unroll
      for (int rr_ = 0; rr_ < R; ++rr_) {
        if (w & 1) Y[(rr_ * (kNW / 2) + (w >> 1)) * 64 + lane] = xsum[rr_];
        else own[rr_] = xsum[rr_];
      }
    }
#pragma unroll
    for (int rr_ = 0; rr_ < R; ++rr_) {
      v_prev[rr_] = v_cur[rr_]; v_cur[rr_] = v_nxt[rr_]; b_cur[rr_] = b_nxt[rr_];
      cc_cur[rr_] = cc_nxt[rr_]; cc_nxt[rr_] = cc_nn[rr_]; cc_nn[rr_] = cc_n3[rr_];
      lo_cur[rr_] = lo_nxt[rr_]; lo_nxt[rr_] = lo_nn[rr_]; lo_nn[rr_] = lo_n3[rr_];
    }
    __syncthreads();
  }
  if (MODE == 3) { consume(ge - 1); return; }
  const double t = block_sum(dot, red);
  if (threadIdx.x == 0) partials[blockIdx.x] = t;
}

// mode 1 = pre (out = b - A v), 2 = post (out = z, partial z.b); returns the number of partials
// mode 3 = pre with the restriction fused in: `bcoarse` receives the block sums of the planes this slab
// owns (needs an even g.z0: z pairs must not straddle the z chunks)
int launch_mg_fine(int mode, const Geom& g, const uint8_t* code, const double* b, MgCoarse c, double omega, double scale,
                   double* out, float* bcoarse, double* partials, const Ctrl* ctrl, const int* lineoff, hipStream_t s) {
  const int LW = (g.pitch + kLV - 1) / kLV;
  constexpr int R = 2;
  const int by = kNW * R;
  const int tiles_x = (g.pitch + kBX - 1) / kBX, tiles_y = (g.ny + by - 1) / by;
  const long long cols = (long long)tiles_x * tiles_y;
  int zchunk, nzc;
  // (the post-smoother takes the chunks of the two-kernel iteration, as its descriptor-addressed form launch_mg_post does: the
  // partial sums of z.r -- and with them the iterates -- are then the same in the dense and the compacted layout)
  if (mode == 2) fused_chunks(g, cols, false, &zchunk, &nzc);
  else pick_chunks(g, cols, mode == 3, &zchunk, &nzc);
  if (!grid_fits(g, cols, nzc)) return -1;
  const int blocks = (int)(cols * nzc);
#define OI_MGF(M, C)                                                                                                     \
  hipLaunchKernelGGL((k_mg_fine<R, M, C>), dim3(blocks), dim3(64 * kNW), 0, s, g, code, b, c, omega, scale, out, bcoarse, \
                     zchunk, tiles_x, tiles_y, nzc, partials, ctrl, lineoff, LW)
  if (mode == 1) { if (lineoff) OI_MGF(1, true); else OI_MGF(1, false); }
  else if (mode == 3) { if (lineoff) OI_MGF(3, true); else OI_MGF(3, false); }
  else { if (lineoff) OI_MGF(2, true); else OI_MGF(2, false); }
#undef OI_MGF
  return blocks;
}

}  // namespace oi
