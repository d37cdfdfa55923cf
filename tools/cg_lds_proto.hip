// cg_lds_proto.hip -- round-2 experiment: does the single-kernel CG iteration stop being latency-bound when the operands of
// the next plane are requested by direct-to-LDS loads (global_load_lds_dwordx4: no VGPRs held while in flight) one
// iteration ahead?  A reduced model of k_cg1 on DENSE arrays with a per-line live mask (49 % of the 8-voxel lines live, in
// correlated blobs): r = r_in - a q_in, p = r + b p_in on the tile and its halo ring, q = A p from a double-buffered LDS
// plane, r / p / q stored, five sums -- the same loop, barrier and halo duties as k_cg1, without the x update and the slot
// table.  Variant 0 loads through registers as k_cg1 does; variant 1 prefetches through LDS.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tools/cg_lds_proto tools/cg_lds_proto.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)
typedef double v2d __attribute__((ext_vector_type(2)));
constexpr int kBX = 128, kLP = kBX + 4;

struct G { int nx, ny, nzl; long long plane; int LW; };

__global__ void k_make_mask(G g, uint8_t* __restrict__ live) {
  const long long n = (long long)(g.nzl + 2) * g.ny * g.LW;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x) {
    const int li = (int)(t % g.LW), j = (int)((t / g.LW) % g.ny), gz = (int)(t / ((long long)g.LW * g.ny));
    const float x = 8.f * li + 4.f, y = (float)j, z = (float)gz;
    const float f = __sinf(6.2831853f * x / 53.f) + __sinf(6.2831853f * y / 47.f) + __sinf(6.2831853f * z / 59.f) +
                    0.35f * __sinf(6.2831853f * (x + 2.f * y + 3.f * z) / 31.f);
    live[t] = (gz >= 1 && gz <= g.nzl && f > 0.02f) ? 1 : 0;
  }
}
// 16-bit live mask per 128-voxel row segment (what the library's segment records would hold): scalar-loadable
__global__ void k_make_segmask(G g, const uint8_t* __restrict__ live, unsigned* __restrict__ M, int SW) {
  const long long n = (long long)(g.nzl + 2) * g.ny * SW;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x) {
    const int sx = (int)(t % SW);
    const long long row = t / SW;
    unsigned m = 0;
    for (int l = 0; l < 16 && 16 * sx + l < g.LW; ++l) m |= (unsigned)live[row * g.LW + 16 * sx + l] << l;
    M[t] = m;
  }
}
__global__ void k_fill(double* __restrict__ a, long long n, unsigned seed) {
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)t * 2654435761u ^ seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    a[t] = (double)(h & 0xffff) / 65536.0 - 0.5;
  }
}

template <int R, int NW, int GLDS>
__global__ void __launch_bounds__(64 * NW, 4) k_cgp(G g, const uint8_t* __restrict__ live, const unsigned* __restrict__ M, int SW,
                                                     const double* __restrict__ r_in,
                                                     const double* __restrict__ q_in, const double* __restrict__ p_in,
                                                     double* __restrict__ r_out, double* __restrict__ q_out,
                                                     double* __restrict__ p_out, double alpha, double beta, int zchunk,
                                                     int tiles_x, int tiles_y, int nzc, double* __restrict__ partials) {
  constexpr int BY = NW * R;
  static_assert(2 * BY <= 64, "x-halo lanes");
  __shared__ __attribute__((aligned(16))) double tile[2][(BY + 2) * kLP];
  __shared__ __attribute__((aligned(16))) double raw[GLDS ? 3 : 1][GLDS ? (BY + 2) * kBX : 2];  // rows 0 / BY + 1 = y halo
  __shared__ __attribute__((aligned(16))) double rawx[GLDS ? 3 : 1][GLDS ? NW * 8 : 2];          // x-halo pairs: 64 B per wave
  __shared__ double red[NW][4];
  const int ntiles = tiles_x * tiles_y * nzc;
  int L = blockIdx.x;
  if ((ntiles & 7) == 0) L = (blockIdx.x & 7) * (ntiles >> 3) + (blockIdx.x >> 3);
  const int tx = L % tiles_x, ty = (L / tiles_x) % tiles_y, zc = L / (tiles_x * tiles_y);
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = tx * kBX + 2 * lane, j0 = ty * BY;
  const int gs = 1 + zc * zchunk, ge = min(g.nzl + 1, gs + zchunk);
  const int hrow = (w == 0) ? j0 - 1 : j0 + BY;
  const bool yhalo = (w < 2) && hrow >= 0 && hrow < g.ny;
  const int xh_row = j0 + (lane >> 1);
  const int xh_i = (lane & 1) ? tx * kBX + kBX : tx * kBX - 1;
  const bool xhalo = (w == 2) && lane < 2 * BY && xh_row < g.ny && xh_i >= 0 && xh_i < g.nx;

  auto live_at = [&](int gz, int j, int ii) -> unsigned {
    return (gz <= g.nzl + 1 && j >= 0 && j < g.ny) ? live[((long long)gz * g.ny + j) * g.LW + (ii >> 3)] : 0u;
  };
  auto load_live = [&](int gz, unsigned (&lv)[R]) {
#pragma unroll
    for (int rr = 0; rr < R; ++rr) lv[rr] = live_at(gz, j0 + w + rr * NW, i);
  };
  unsigned hlive = 0;
  auto load_halo_live = [&](int gz) {
    hlive = 0;
    if (yhalo) hlive = live_at(gz, hrow, i);
    else if (xhalo) hlive = live_at(gz, xh_row, xh_i);
  };
  // ---- variant 1: live masks of whole row segments through the scalar cache (no vector load in the loop)
  const unsigned line = lane >> 2;
  auto seg_mask = [&](int gz, int j, int sx) -> unsigned {  // wave-uniform arguments -> s_load_dword
    return (gz <= g.nzl + 1 && j >= 0 && j < g.ny && sx >= 0 && sx < SW) ? M[((long long)gz * g.ny + j) * SW + sx] : 0u;
  };
  struct Masks { unsigned own[R], xl[R], xr[R], yh; };  // own rows, their left / right neighbour segments, this wave's y-halo row
  auto load_masks = [&](Masks& K, int gz) {
#pragma unroll
    for (int rr = 0; rr < R; ++rr) {
      const int j = j0 + w + rr * NW;
      K.own[rr] = seg_mask(gz, j, tx); K.xl[rr] = seg_mask(gz, j, tx - 1); K.xr[rr] = seg_mask(gz, j, tx + 1);
    }
    K.yh = (w < 2) ? seg_mask(gz, hrow, tx) : 0u;
  };
  // ---- register path (variant 0)
  auto load_plane = [&](int gz, const unsigned (&lv)[R], v2d (&pn)[R], v2d (&rn)[R], bool store) {
#pragma unroll
    for (int rr = 0; rr < R; ++rr) {
      const int j = j0 + w + rr * NW;
      const v2d zero = {0.0, 0.0};
      pn[rr] = zero; rn[rr] = zero;
      if (lv[rr]) {
        const long long e = (long long)gz * g.plane + (long long)j * g.nx + i;
        const v2d rv = *reinterpret_cast<const v2d*>(r_in + e), qv = *reinterpret_cast<const v2d*>(q_in + e);
        const v2d pin = *reinterpret_cast<const v2d*>(p_in + e);
        const v2d r = {rv.x - alpha * qv.x, rv.y - alpha * qv.y};
        const v2d p = {r.x + beta * pin.x, r.y + beta * pin.y};
        rn[rr] = r; pn[rr] = p;
        if (store) { *reinterpret_cast<v2d*>(r_out + e) = r; *reinterpret_cast<v2d*>(p_out + e) = p; }
      }
    }
  };
  // ---- LDS-DMA path (variant 1): requests, and one iteration later the arithmetic
  auto glds16 = [&](const double* src, double* dst_wave_base) {
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
                                     (void __attribute__((address_space(3)))*)dst_wave_base, 16, 0, 0);
  };
  auto issue_glds = [&](int gz, const Masks& K) {
#pragma unroll
    for (int rr = 0; rr < R; ++rr) {
      const int j = j0 + w + rr * NW, row = w + rr * NW + 1;
      if (K.own[rr]) {  // wave-uniform
        if ((K.own[rr] >> line) & 1u) {
          const long long e = (long long)gz * g.plane + (long long)j * g.nx + i;
          glds16(r_in + e, &raw[0][row * kBX]); glds16(q_in + e, &raw[GLDS ? 1 : 0][row * kBX]); glds16(p_in + e, &raw[GLDS ? 2 : 0][row * kBX]);
        }
      }
    }
    if (K.yh) {
      if ((K.yh >> line) & 1u) {
        const long long e = (long long)gz * g.plane + (long long)hrow * g.nx + i;
        const int row = (w == 0) ? 0 : BY + 1;
        glds16(r_in + e, &raw[0][row * kBX]); glds16(q_in + e, &raw[GLDS ? 1 : 0][row * kBX]); glds16(p_in + e, &raw[GLDS ? 2 : 0][row * kBX]);
      }
    }
    // x halo of this wave's own rows: lane 2 rr = left (last line of the segment to the left), 2 rr + 1 = right
    bool any = false;
#pragma unroll
    for (int rr = 0; rr < R; ++rr) any = any || ((K.xl[rr] >> 15) & 1u) || (K.xr[rr] & 1u);
    if (any) {
      unsigned want = 0;
      int rsel = 0;
#pragma unroll
      for (int rr = 0; rr < R; ++rr) {
        if (lane == 2 * rr) { want = (K.xl[rr] >> 15) & 1u; rsel = rr; }
        if (lane == 2 * rr + 1) { want = K.xr[rr] & 1u; rsel = rr; }
      }
      if (lane < 2 * R && want) {
        const int j = j0 + w + rsel * NW;
        const int xi = (lane & 1) ? tx * kBX + kBX : tx * kBX - 2;  // the aligned pair that holds the halo voxel
        const long long e = (long long)gz * g.plane + (long long)j * g.nx + xi;
        glds16(r_in + e, &rawx[0][w * 8]); glds16(q_in + e, &rawx[GLDS ? 1 : 0][w * 8]); glds16(p_in + e, &rawx[GLDS ? 2 : 0][w * 8]);
      }
    }
  };
  auto combine_lds = [&](int gz, const Masks& K, v2d (&pn)[R], v2d (&rn)[R], bool store) {
#pragma unroll
    for (int rr = 0; rr < R; ++rr) {
      const int j = j0 + w + rr * NW, row = w + rr * NW + 1;
      const v2d zero = {0.0, 0.0};
      pn[rr] = zero; rn[rr] = zero;
      if (K.own[rr]) {
        if ((K.own[rr] >> line) & 1u) {
          const v2d rv = *reinterpret_cast<const v2d*>(&raw[0][row * kBX + 2 * lane]);
          const v2d qv = *reinterpret_cast<const v2d*>(&raw[GLDS ? 1 : 0][row * kBX + 2 * lane]);
          const v2d pin = *reinterpret_cast<const v2d*>(&raw[GLDS ? 2 : 0][row * kBX + 2 * lane]);
          const v2d r = {rv.x - alpha * qv.x, rv.y - alpha * qv.y};
          const v2d p = {r.x + beta * pin.x, r.y + beta * pin.y};
          rn[rr] = r; pn[rr] = p;
          if (store) {
            const long long e = (long long)gz * g.plane + (long long)j * g.nx + i;
            *reinterpret_cast<v2d*>(r_out + e) = r; *reinterpret_cast<v2d*>(p_out + e) = p;
          }
        }
      }
    }
  };
  // p_k of the halo ring of plane gz from the raw LDS data, into the LDS plane
  auto stage_lds = [&](int bsel, const Masks& K, const v2d (&pn)[R]) {
    double* T = tile[bsel];
#pragma unroll
    for (int rr = 0; rr < R; ++rr) *reinterpret_cast<v2d*>(&T[(w + rr * NW + 1) * kLP + 2 + 2 * lane]) = pn[rr];
    if (w < 2) {
      v2d h = {0.0, 0.0};
      if ((K.yh >> line) & 1u) {
        const int row = (w == 0) ? 0 : BY + 1;
        const v2d rv = *reinterpret_cast<const v2d*>(&raw[0][row * kBX + 2 * lane]);
        const v2d qv = *reinterpret_cast<const v2d*>(&raw[GLDS ? 1 : 0][row * kBX + 2 * lane]);
        const v2d pv = *reinterpret_cast<const v2d*>(&raw[GLDS ? 2 : 0][row * kBX + 2 * lane]);
        h.x = (rv.x - alpha * qv.x) + beta * pv.x; h.y = (rv.y - alpha * qv.y) + beta * pv.y;
      }
      *reinterpret_cast<v2d*>(&T[((w == 0) ? 0 : BY + 1) * kLP + 2 + 2 * lane]) = h;
    }
    if (lane < 2 * R) {
      unsigned want = 0;
      int rsel = 0;
#pragma unroll
      for (int rr = 0; rr < R; ++rr) {
        if (lane == 2 * rr) { want = (K.xl[rr] >> 15) & 1u; rsel = rr; }
        if (lane == 2 * rr + 1) { want = K.xr[rr] & 1u; rsel = rr; }
      }
      double h = 0.0;
      if (want) {
        const int k = w * 8 + 2 * lane + ((lane & 1) ? 0 : 1);  // left halo = second voxel of its pair, right halo = first
        h = (rawx[0][k] - alpha * rawx[GLDS ? 1 : 0][k]) + beta * rawx[GLDS ? 2 : 0][k];
      }
      T[(w + rsel * NW + 1) * kLP + ((lane & 1) ? 2 + kBX : 1)] = h;
    }
  };
  auto stage = [&](int bsel, int gz, const v2d (&pn)[R], unsigned hl) {
    double* T = tile[bsel];
#pragma unroll
    for (int rr = 0; rr < R; ++rr) *reinterpret_cast<v2d*>(&T[(w + rr * NW + 1) * kLP + 2 + 2 * lane]) = pn[rr];
    if (w < 2) {
      v2d h = {0.0, 0.0};
      if (hl) {
        v2d rv, qv, pv;
        if (GLDS) {
          const int row = (w == 0) ? 0 : BY + 1;
          rv = *reinterpret_cast<const v2d*>(&raw[0][row * kBX + 2 * lane]);
          qv = *reinterpret_cast<const v2d*>(&raw[GLDS ? 1 : 0][row * kBX + 2 * lane]);
          pv = *reinterpret_cast<const v2d*>(&raw[GLDS ? 2 : 0][row * kBX + 2 * lane]);
        } else {
          const long long e = (long long)gz * g.plane + (long long)hrow * g.nx + i;
          rv = *reinterpret_cast<const v2d*>(r_in + e); qv = *reinterpret_cast<const v2d*>(q_in + e);
          pv = *reinterpret_cast<const v2d*>(p_in + e);
        }
        h.x = (rv.x - alpha * qv.x) + beta * pv.x; h.y = (rv.y - alpha * qv.y) + beta * pv.y;
      }
      *reinterpret_cast<v2d*>(&T[((w == 0) ? 0 : BY + 1) * kLP + 2 + 2 * lane]) = h;
    } else if (w == 2 && lane < 2 * BY) {
      double h = 0.0;
      if (hl) {
        double rv, qv, pv;
        if (GLDS) {
          const int k = 2 * lane + (xh_i & 1);
          rv = rawx[0][k]; qv = rawx[GLDS ? 1 : 0][k]; pv = rawx[GLDS ? 2 : 0][k];
        } else {
          const long long e = (long long)gz * g.plane + (long long)xh_row * g.nx + xh_i;
          rv = r_in[e]; qv = q_in[e]; pv = p_in[e];
        }
        h = (rv - alpha * qv) + beta * pv;
      }
      T[((lane >> 1) + 1) * kLP + ((lane & 1) ? 2 + kBX : 1)] = h;
    }
  };
  auto barrier = [&]() {
    if (GLDS) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
    else __syncthreads();
  };

  v2d pn_prev[R], pn_cur[R], pn_nxt[R], rn_cur[R], rn_nxt[R];
  unsigned lv_cur[R], lv_nxt[R], lv_nn[R], lv_tmp[R];
  (void)lv_nn;
  double s_rr = 0, s_pq = 0, s_zq = 0, s_qq = 0;
  auto compute = [&](int gz) {
    const double* T = tile[gz & 1];
#pragma unroll
    for (int rr = 0; rr < R; ++rr) {
      if (!lv_cur[rr]) continue;
      const int j = j0 + w + rr * NW;
      const int base = (w + rr * NW + 1) * kLP + 2 + 2 * lane;
      const v2d lf = *reinterpret_cast<const v2d*>(&T[base - 2]), rt = *reinterpret_cast<const v2d*>(&T[base + 2]);
      const v2d ym = *reinterpret_cast<const v2d*>(&T[base - kLP]), yp = *reinterpret_cast<const v2d*>(&T[base + kLP]);
      const v2d pc = pn_cur[rr], rc = rn_cur[rr];
      const double s0 = ((((lf.y + pc.y) + ym.x) + yp.x) + pn_prev[rr].x) + pn_nxt[rr].x;
      const double s1 = ((((pc.x + rt.x) + ym.y) + yp.y) + pn_prev[rr].y) + pn_nxt[rr].y;
      const v2d q = {3.0 * pc.x - s0, 3.0 * pc.y - s1};
      *reinterpret_cast<v2d*>(q_out + (long long)gz * g.plane + (long long)j * g.nx + i) = q;
      s_rr += rc.x * rc.x; s_rr += rc.y * rc.y;
      s_pq += pc.x * q.x; s_pq += pc.y * q.y;
      s_zq += rc.x * q.x; s_zq += rc.y * q.y;
      s_qq += q.x * q.x; s_qq += q.y * q.y;
    }
  };

  // prologue (register path for both variants: planes gs - 1 and gs)
  load_halo_live(gs);
  load_live(gs - 1, lv_tmp); load_live(gs, lv_cur); load_live(gs + 1, lv_nxt);
  load_plane(gs - 1, lv_tmp, pn_prev, rn_nxt, false);
  load_plane(gs, lv_cur, pn_cur, rn_cur, true);
  {
    // (variant 1: the halo of plane gs through the register path as well)
    double* T = tile[gs & 1];
#pragma unroll
    for (int rr = 0; rr < R; ++rr) *reinterpret_cast<v2d*>(&T[(w + rr * NW + 1) * kLP + 2 + 2 * lane]) = pn_cur[rr];
    if (w < 2) {
      v2d h = {0.0, 0.0};
      if (hlive) {
        const long long e = (long long)gs * g.plane + (long long)hrow * g.nx + i;
        const v2d rv = *reinterpret_cast<const v2d*>(r_in + e), qv = *reinterpret_cast<const v2d*>(q_in + e), pv = *reinterpret_cast<const v2d*>(p_in + e);
        h.x = (rv.x - alpha * qv.x) + beta * pv.x; h.y = (rv.y - alpha * qv.y) + beta * pv.y;
      }
      *reinterpret_cast<v2d*>(&T[((w == 0) ? 0 : BY + 1) * kLP + 2 + 2 * lane]) = h;
    } else if (w == 2 && lane < 2 * BY) {
      double h = 0.0;
      if (hlive) {
        const long long e = (long long)gs * g.plane + (long long)xh_row * g.nx + xh_i;
        h = (r_in[e] - alpha * q_in[e]) + beta * p_in[e];
      }
      T[((lane >> 1) + 1) * kLP + ((lane & 1) ? 2 + kBX : 1)] = h;
    }
  }
  load_halo_live(gs + 1);
  if (GLDS) {
    Masks K1, K2, K3;  // planes gz + 1, gz + 2, gz + 3
    load_masks(K1, gs + 1); load_masks(K2, gs + 2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (prologue loads done before the first DMA is counted)
    issue_glds(gs + 1, K1);
    barrier();
    for (int gz = gs; gz < ge; ++gz) {
      load_masks(K3, gz + 3);                              // scalar loads: they do not touch the vector-memory counter
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // raw holds plane gz + 1 once this wave's DMAs have landed
      combine_lds(gz + 1, K1, pn_nxt, rn_nxt, gz + 1 < ge);
      if (gz + 1 < ge) stage_lds((gz + 1) & 1, K1, pn_nxt);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // raw has been read: it may be overwritten
      if (gz + 2 <= ge) issue_glds(gz + 2, K2);            // in flight across the stencil, the barrier and the next step's start
#pragma unroll
      for (int rr = 0; rr < R; ++rr) lv_cur[rr] = lv_cur[rr];
      compute(gz);
#pragma unroll
      for (int rr = 0; rr < R; ++rr) {
        pn_prev[rr] = pn_cur[rr]; pn_cur[rr] = pn_nxt[rr]; rn_cur[rr] = rn_nxt[rr];
        lv_cur[rr] = (K1.own[rr] >> line) & 1u;
      }
      K1 = K2; K2 = K3;
      barrier();
    }
  } else {
    barrier();
    for (int gz = gs; gz < ge; ++gz) {
      load_live(gz + 2, lv_nn);
      load_plane(gz + 1, lv_nxt, pn_nxt, rn_nxt, gz + 1 < ge);
      if (gz + 1 < ge) { stage((gz + 1) & 1, gz + 1, pn_nxt, hlive); load_halo_live(gz + 2); }
      compute(gz);
#pragma unroll
      for (int rr = 0; rr < R; ++rr) {
        pn_prev[rr] = pn_cur[rr]; pn_cur[rr] = pn_nxt[rr]; rn_cur[rr] = rn_nxt[rr];
        lv_cur[rr] = lv_nxt[rr]; lv_nxt[rr] = lv_nn[rr];
      }
      barrier();
    }
  }
  if (GLDS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  double sums[4] = {s_rr, s_pq, s_zq, s_qq};
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    double v = sums[c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if (lane == 0) red[w][c] = v;
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    double t = 0.0;
    for (int k = 0; k < NW; ++k) t += red[k][threadIdx.x];
    partials[4 * (long long)blockIdx.x + threadIdx.x] = t;
  }
}

template <typename F>
static double timeit(F f, int reps) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); f();
  CK(hipEventRecord(a));
  for (int i = 0; i < reps; ++i) f();
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms / reps;
}

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 1024;
  G g; g.nx = N; g.ny = N; g.nzl = N; g.plane = (long long)N * N; g.LW = N / 8;
  const long long n = (long long)(N + 2) * g.plane, nl = (long long)(N + 2) * N * g.LW;
  const long long pad = argc > 2 ? atoll(argv[2]) : -1;  // >= 0: all vectors from ONE arena, `pad` bytes between them
  double *r0, *q0, *p0, *r1, *q1, *p1, *r2, *q2, *p2, *part;
  uint8_t* live;
  unsigned* M;
  const int SW = N / 128;
  if (pad < 0) {
    CK(hipMalloc(&r0, n * 8)); CK(hipMalloc(&q0, n * 8)); CK(hipMalloc(&p0, n * 8));
    CK(hipMalloc(&r1, n * 8)); CK(hipMalloc(&q1, n * 8)); CK(hipMalloc(&p1, n * 8));
    CK(hipMalloc(&r2, n * 8)); CK(hipMalloc(&q2, n * 8)); CK(hipMalloc(&p2, n * 8));
  } else {
    char* arena;
    const long long stride = n * 8 + pad;
    CK(hipMalloc(&arena, 9 * stride + 4096));
    double** v[9] = {&r0, &q0, &p0, &r1, &q1, &p1, &r2, &q2, &p2};
    for (int k = 0; k < 9; ++k) *v[k] = (double*)(arena + k * stride);
  }
  printf("bases mod 2 MiB:");
  for (double* a : {r0, q0, p0, r1, q1, p1, r2, q2, p2}) printf(" %llx", (unsigned long long)a & 0x1fffff);
  printf("  (r0 = %p)\n", (void*)r0);
  CK(hipMalloc(&live, nl)); CK(hipMalloc(&part, (1 << 20) * 8));
  CK(hipMalloc(&M, (size_t)(N + 2) * N * SW * sizeof(unsigned)));
  hipLaunchKernelGGL(k_make_mask, dim3(4096), dim3(256), 0, 0, g, live);
  hipLaunchKernelGGL(k_make_segmask, dim3(4096), dim3(256), 0, 0, g, live, M, SW);
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, r0, n, 1u);
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, q0, n, 2u);
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, p0, n, 3u);
  for (double* a : {r1, q1, p1, r2, q2, p2}) CK(hipMemset(a, 0, n * 8));
  CK(hipDeviceSynchronize());
  {
    std::vector<uint8_t> hl(nl);
    CK(hipMemcpy(hl.data(), live, nl, hipMemcpyDeviceToHost));
    long long c = 0;
    for (long long t = (long long)N * g.LW; t < nl - (long long)N * g.LW; ++t) c += hl[t];
    printf("live 8-voxel lines: %.3f of the owned planes\n", (double)c / ((double)N * N * g.LW));
  }
#define RUN(R, NW, GL, ZC, RO, QO, PO)                                                                                     \
  {                                                                                                                        \
    const int by = R * NW, tx = N / 128, ty = (N + by - 1) / by, nzc = (N + ZC - 1) / ZC, blocks = tx * ty * nzc;          \
    const double ms = timeit([&] { hipLaunchKernelGGL((k_cgp<R, NW, GL>), dim3(blocks), dim3(64 * NW), 0, 0, g, live, M, SW, r0, q0, p0, RO, QO, PO, 0.3, 0.2, ZC, tx, ty, nzc, part); }, 10); \
    CK(hipGetLastError());                                                                                                 \
    printf("R %d NW %2d %s zchunk %3d blocks %6d  %7.3f ms\n", R, NW, GL ? "LDS-DMA prefetch" : "register loads  ", ZC, blocks, ms); \
    fflush(stdout);                                                                                                        \
  }
  RUN(2, 16, 0, 64, r1, q1, p1)
  RUN(1, 16, 1, 64, r2, q2, p2)
  {  // same results?
    std::vector<double> a(g.plane), b(g.plane);
    long long bad = 0;
    double* pa[3] = {r1, q1, p1};
    double* pb[3] = {r2, q2, p2};
    for (int v = 0; v < 3; ++v)
      for (int gz : {1, 2, 63, 64, 65, N / 2, N}) {
        CK(hipMemcpy(a.data(), pa[v] + (long long)gz * g.plane, g.plane * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(b.data(), pb[v] + (long long)gz * g.plane, g.plane * 8, hipMemcpyDeviceToHost));
        long long bv = 0;
        for (long long e = 0; e < g.plane; ++e) bv += memcmp(&a[e], &b[e], 8) != 0;
        if (bv) printf("  array %d plane %d: %lld differing values\n", v, gz, bv);
        bad += bv;
      }
    printf("register path vs LDS-DMA path: %lld differing values in the sampled planes\n", bad);
  }
  RUN(2, 16, 0, 64, r1, q1, p1)
  RUN(1, 16, 1, 64, r2, q2, p2) RUN(2, 8, 1, 64, r2, q2, p2)
  RUN(2, 16, 0, 64, r1, q1, p1) RUN(2, 8, 1, 64, r2, q2, p2)
  if (argc > 3 && !strcmp(argv[3], "slots")) {
    // NS vector slots in one arena; an iteration reads the triple A and writes the triple B (any free slots): how does
    // the time of the 2-cycle A->B, B->A spread over the choices?
    const int NS = argc > 4 ? atoi(argv[4]) : 9;
    const long long st0 = (n * 8 + (2 << 20) - 1) / (2 << 20) * (2 << 20);
    char* big;
    if (hipMalloc(&big, NS * st0) != hipSuccess) { printf("no room\n"); return 0; }
    CK(hipMemset(big, 0, NS * st0));
    printf("%d slots of %.3f GiB at %p\n", NS, (double)st0 / (1 << 30), (void*)big);
    auto slot = [&](int k) { return (double*)(big + k * st0); };
    auto t1 = [&](const int* A, const int* B, int lds) {
      double *ri = slot(A[0]), *qi = slot(A[1]), *pi = slot(A[2]), *ro = slot(B[0]), *qo = slot(B[1]), *po = slot(B[2]);
      const int ZC = 64, tx = N / 128, nzc = (N + ZC - 1) / ZC;
      if (lds) { const int ty = (N + 15) / 16; return timeit([&] { hipLaunchKernelGGL((k_cgp<2, 8, 1>), dim3(tx * ty * nzc), dim3(512), 0, 0, g, live, M, SW, ri, qi, pi, ro, qo, po, 0.3, 0.2, ZC, tx, ty, nzc, part); }, 4); }
      const int ty = (N + 31) / 32;
      return timeit([&] { hipLaunchKernelGGL((k_cgp<2, 16, 0>), dim3(tx * ty * nzc), dim3(1024), 0, 0, g, live, M, SW, ri, qi, pi, ro, qo, po, 0.3, 0.2, ZC, tx, ty, nzc, part); }, 4);
    };
    unsigned rng = 12345;
    auto rnd = [&](int m) { rng = rng * 1664525u + 1013904223u; return (int)((rng >> 8) % m); };
    double best[2] = {1e9, 1e9}, worst[2] = {0, 0}, sum[2] = {0, 0};
    const int trials = 48;
    for (int t = 0; t < trials; ++t) {
      int perm[16];
      for (int k = 0; k < NS; ++k) perm[k] = k;
      if (t > 0) for (int k = NS - 1; k > 0; --k) { const int j = rnd(k + 1), x = perm[k]; perm[k] = perm[j]; perm[j] = x; }
      const int* A = perm; const int* B = perm + 3;
      printf("A %d %d %d  B %d %d %d :", A[0], A[1], A[2], B[0], B[1], B[2]);
      for (int lds = 0; lds < 2; ++lds) {
        const double ab = t1(A, B, lds), ba = t1(B, A, lds);
        printf("  %s %.3f + %.3f = %.3f", lds ? "lds" : "reg", ab, ba, ab + ba);
        const double c = 0.5 * (ab + ba);
        best[lds] = c < best[lds] ? c : best[lds]; worst[lds] = c > worst[lds] ? c : worst[lds]; sum[lds] += c;
      }
      printf("\n"); fflush(stdout);
    }
    for (int lds = 0; lds < 2; ++lds) printf("%s: best %.3f  mean %.3f  worst %.3f ms per iteration\n", lds ? "lds" : "reg", best[lds], sum[lds] / trials, worst[lds]);
    return 0;
  }
  if (argc > 3 && !strcmp(argv[3], "map")) {
    // one 190 GiB arena; the six vectors of a launch at base + k * stride: time against base and against stride
    char* big;
    const long long GiB = 1ll << 30, total = 190 * GiB;
    if (hipMalloc(&big, total) != hipSuccess) { printf("no room for the map arena\n"); return 0; }
    printf("map arena at %p\n", (void*)big);
    auto place = [&](long long base, long long stride, const int* perm) {
      double* w[6];
      for (int k = 0; k < 6; ++k) w[k] = (double*)(big + base + perm[k] * stride);
      CK(hipMemcpy(w[0], r0, n * 8, hipMemcpyDeviceToDevice)); CK(hipMemcpy(w[1], q0, n * 8, hipMemcpyDeviceToDevice));
      CK(hipMemcpy(w[2], p0, n * 8, hipMemcpyDeviceToDevice));
      double *sr0 = r0, *sq0 = q0, *sp0 = p0;
      r0 = w[0]; q0 = w[1]; p0 = w[2];
      RUN(2, 16, 0, 64, w[3], w[4], w[5])
      RUN(2, 8, 1, 64, w[3], w[4], w[5])
      r0 = sr0; q0 = sq0; p0 = sp0;
    };
    const int id[6] = {0, 1, 2, 3, 4, 5}, il[6] = {0, 2, 4, 1, 3, 5}, rv[6] = {5, 4, 3, 2, 1, 0};
    const long long st0 = (n * 8 + (2 << 20) - 1) / (2 << 20) * (2 << 20);
    for (long long b = 0; b + 6 * st0 <= total; b += 12 * GiB) { printf("base %3lld GiB stride %.3f GiB\n", b / GiB, (double)st0 / GiB); place(b, st0, id); }
    for (long long extra : {256ll << 20, 1 * GiB, 3 * GiB, 8 * GiB, 20 * GiB}) {
      if (6 * (st0 + extra) > total) break;
      printf("base   0 GiB stride %.3f GiB\n", (double)(st0 + extra) / GiB); place(0, st0 + extra, id);
    }
    printf("base   0 GiB interleaved in/out order\n"); place(0, st0, il);
    printf("base   0 GiB reversed order\n"); place(0, st0, rv);
    return 0;
  }
  if (pad >= 0) {  // are the timing modes a property of WHERE an allocation lands?  more arenas in the same process
    for (int extra = 0; extra < 2; ++extra) {
      char* arena2;
      const long long stride = n * 8 + pad;
      if (hipMalloc(&arena2, 9 * stride + 4096) != hipSuccess) { printf("no room for another arena\n"); break; }
      double* w[9];
      for (int k = 0; k < 9; ++k) w[k] = (double*)(arena2 + k * stride);
      CK(hipMemcpy(w[0], r0, n * 8, hipMemcpyDeviceToDevice)); CK(hipMemcpy(w[1], q0, n * 8, hipMemcpyDeviceToDevice));
      CK(hipMemcpy(w[2], p0, n * 8, hipMemcpyDeviceToDevice));
      for (int k = 3; k < 9; ++k) CK(hipMemset(w[k], 0, n * 8));
      printf("arena %d at %p\n", extra + 2, (void*)arena2);
      double *sr0 = r0, *sq0 = q0, *sp0 = p0;
      r0 = w[0]; q0 = w[1]; p0 = w[2];
      RUN(2, 16, 0, 64, w[3], w[4], w[5])
      RUN(2, 8, 1, 64, w[6], w[7], w[8])
      r0 = sr0; q0 = sq0; p0 = sp0;
      printf("first arena again:\n");
      RUN(2, 16, 0, 64, r1, q1, p1)
      RUN(2, 8, 1, 64, r2, q2, p2)
    }
  }
  return 0;
}
