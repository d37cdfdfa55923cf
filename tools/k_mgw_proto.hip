// DISCARDED PROTOTYPE (round 5, profiles/r05/mg_forms_round5_experiments_discarded.txt): compiled as openimpala_amd/csrc/oi_k_mgw.hip behind an option
// "mg_wave"; bit-identical solutions, slower than k_cg1b FORM 2.  Kept like tools/apply_proto.hip / tools/cg_lds_proto.hip: not built, not shipped.
// oi_k_mgw.hip -- the fine-level kernels of the multigrid-preconditioned iteration on COMPACTED vectors in a WAVE-AUTONOMOUS form.
//
// k_cg1b's forms 1-3 (oi_k_cg1b.hip) are instruction-bound, not bandwidth-bound: a block of eight waves shares an LDS tile, meets at
// a barrier once per plane, and every wave spends ~570 instructions (250 of them scalar: roles, ring positions, descriptors) on the
// four voxels a lane owns per plane -- 2.2 instructions per voxel, 0.55-0.66 of the HBM roofline on the bytes they move.  The
// stand-alone apply k_apply7 (oi_k_dense.hip) needs 0.63 per voxel and reaches 0.75 with no LDS tile and no barrier: a wave owns RY rows
// of a 128-voxel tile column, keeps the planes it needs in registers, gets its x neighbours from the lanes beside it (DPP) and re-reads
// the row above and below its strip (the neighbouring wave's rows: cache hits).  This file builds the multigrid forms on that skeleton,
// for the compacted layout: a row segment's slots come from its segment record (one scalar load per row and plane), every vector access
// goes through a per-plane buffer descriptor with an out-of-range offset where a lane has nothing to move (as in k_cg1b).
//
// Same arithmetic, operation for operation, as the k_cg1b forms they stand in for: the vectors and the coarse right-hand side they leave
// are bit-identical.  The per-block partial sums (r.r, z.b, p.q) are taken over other tiles, so the sums differ in the last bits.
#include "oi_kern.h"

namespace oi {

#ifndef OI_MGW_NT
#define OI_MGW_NT 1  // r_k written nontemporally (as k_cg1b FORM 2, kNtSt2)
#endif

// FORM 2: r_k = r_{k-1} - alpha_{k-1} q_{k-1} (stored), partial r_k.r_k, v = omega r_k / c, res = r_k - A v, the 2x2x2 block sums of res as
// the fp32 right-hand side of the first coarse level.  a_in = r_{k-1}, b_in = q_{k-1}, o1 = r_k.  Chunks hold whole z pairs.
template <int FORM, int RY, int NW>
__global__ void __launch_bounds__(64 * NW) k_mgw(Geom g, const uint8_t* __restrict__ code, const double* __restrict__ a_in,
                                                  const double* __restrict__ b_in, double* __restrict__ o1, const double* __restrict__ hist, int k,
                                                  int zchunk, int tiles_x, int tiles_y, int nzc, double* __restrict__ partials,
                                                  const Ctrl* __restrict__ ctrl, const int* __restrict__ lineoff,
                                                  const long long* __restrict__ plane_first, const uint2* __restrict__ seg, int LW,
                                                  float* __restrict__ bcoarse, MgCoarse C, double omega) {
  static_assert(FORM == 2, "forms built so far");
  static_assert((RY & 1) == 0 && 2 * RY <= 64, "whole y pairs per wave; one lane per x-halo value");
  constexpr int NR = RY + 2;  // rows j0 - 1 .. j0 + RY
  if (ctrl && ctrl->done) return;
  __shared__ double wtab[8];  // omega / c
  __shared__ double red[NW];
  if (threadIdx.x < 8) wtab[threadIdx.x] = threadIdx.x ? omega / (double)threadIdx.x : 0.0;
  __syncthreads();

  const int L = tile_of_block(blockIdx.x, tiles_x * tiles_y, nzc, g.tune.tile_map);
  const int tx = L % tiles_x, ty = (L / tiles_x) % tiles_y, zc = L / (tiles_x * tiles_y);
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int i = tx * kBX + 2 * lane, j0 = (ty * NW + w) * RY;
  const bool in_x = i < g.pitch;
  const int gs = 1 + zc * zchunk, ge = min(g.nzl + 1, gs + zchunk);  // owned planes [gs, ge)
  const double alpha = k > 0 ? safe_div(hist[H_STRIDE * (k - 1) + H_RZ], hist[H_STRIDE * (k - 1) + H_PQ]) : 0.0;

  // per-lane parts of the offsets
  const int ivox = in_x ? i : kOob;         // code bytes inside a row
  const int sub = (i & (kLV - 1)) * 8;      // this lane's pair inside its line (bytes)
  const unsigned lbit = 1u << (lane >> (kLVs - 1)), lbelow = lbit - 1u;  // its line inside the 128-voxel segment
  // the x halo: lane 2 r + side holds the voxel left (side 0) / right (side 1) of the tile column in row j0 + r
  const int jh = j0 + (lane >> 1);
  const int ih = (lane & 1) ? tx * kBX + kBX : tx * kBX - 1;
  const bool hvalid = lane < 2 * RY && jh < g.ny && ih >= 0 && ih < g.pitch;
  const int hvox = hvalid ? jh * g.pitch + ih : kOob;
  const int hlin = hvalid ? (jh * LW + (ih >> kLVs)) * 4 : kOob;
  const int hsub = (ih & (kLV - 1)) * 8;
  // rows (scalars): offset of the row inside a plane (out of range for a row outside the volume), the row whose segment record is read,
  // and the mask that empties the record of a row outside
  int joff[NR], jrec[NR];
  unsigned rmask[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const int j = j0 - 1 + r;
    const bool ok = j >= 0 && j < g.ny;
    joff[r] = ok ? j * g.pitch : kOob;
    jrec[r] = min(max(j, 0), g.ny - 1);
    rmask[r] = ok ? 0xffffu : 0u;
  }
  const long long seg_plane = (long long)g.ny * tiles_x;
  const long long plane_lines4 = 4ll * g.ny * LW;

  struct Meta { unsigned cc[NR]; uint2 rec[NR]; unsigned hc; int hslot; };
  struct Raw { double2 a[NR], b[NR]; double ah, bh; int off[RY]; unsigned cc[NR]; unsigned hc; long long pfa, nb; };
  struct Plane { double2 v[NR]; double xh; double2 rc[RY]; unsigned cc[RY]; };

  // code bytes, segment records and (x halo) the table entry of plane p; halo = false: a plane that only serves as z neighbour
  auto meta_issue = [&](int p, bool halo, Meta& m) {
    const __amdgpu_buffer_rsrc_t cb = buf_of(code + (long long)p * g.plane, g.plane);
    const uint2* sp = seg + (long long)p * seg_plane + tx;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      m.cc[r] = 0; m.rec[r] = make_uint2(0u, 0u);
      if (halo || (r >= 1 && r <= RY)) {
        m.cc[r] = __builtin_amdgcn_raw_buffer_load_b16(cb, ivox + joff[r], 0, 0);
        m.rec[r] = sp[(long long)jrec[r] * tiles_x];
      }
    }
    m.hc = 0; m.hslot = 0;
    if (halo) {
      m.hc = __builtin_amdgcn_raw_buffer_load_b8(cb, hvox, 0, 0);
      m.hslot = (int)__builtin_amdgcn_raw_buffer_load_b32(buf_of(reinterpret_cast<const char*>(lineoff) + (long long)p * plane_lines4, plane_lines4), hlin, 0, 0);
    }
  };
  // the vectors of plane p
  auto vec_issue = [&](int p, bool halo, const Meta& m, Raw& raw) {
    const long long pfa = plane_first[p], pfb = plane_first[p + 1];
    raw.pfa = pfa; raw.nb = (long long)(8 * kLV) * (pfb - pfa);
    const __amdgpu_buffer_rsrc_t ab = buf_of(a_in + kLV * pfa, raw.nb), bb = buf_of(b_in + kLV * pfa, raw.nb);
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      raw.a[r] = raw.b[r] = make_double2(0.0, 0.0);
      raw.cc[r] = m.cc[r];
      if (halo || (r >= 1 && r <= RY)) {
        const unsigned mask = m.rec[r].y & rmask[r];
        const int slot = (int)m.rec[r].x + __builtin_popcount(mask & lbelow);
        const int off = (in_x && (mask & lbit)) ? (slot - (int)pfa) * (8 * kLV) + sub : kOob;
        raw.a[r] = bld2(ab, off);
        raw.b[r] = bld2(bb, off);
        if (r >= 1 && r <= RY) raw.off[r - 1] = off;
      }
    }
    raw.hc = m.hc;
    raw.ah = raw.bh = 0.0;
    if (halo) {
      const int hoff = (hvalid && m.hc) ? (m.hslot - (int)pfa) * (8 * kLV) + hsub : kOob;
      raw.ah = bld1(ab, hoff);
      raw.bh = bld1(bb, hoff);
    }
  };
  // r_k, its store, and the staged v = omega r_k / c of plane p from the loaded values
  auto convert = [&](bool store, const Raw& raw, Plane& P) {
    const __amdgpu_buffer_rsrc_t ob = buf_of(o1 + kLV * raw.pfa, raw.nb);
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const double r0 = raw.a[r].x - alpha * raw.b[r].x, r1 = raw.a[r].y - alpha * raw.b[r].y;
      const unsigned c = raw.cc[r];
      P.v[r] = make_double2(r0 * wtab[c & 0xffu], r1 * wtab[c >> 8]);
      if (r >= 1 && r <= RY) {
        P.rc[r - 1] = make_double2(r0, r1);
        P.cc[r - 1] = c;
        if (store) bst2<OI_MGW_NT ? 2 : 0>(ob, raw.off[r - 1], r0, r1);
      }
    }
    P.xh = (raw.ah - alpha * raw.bh) * wtab[raw.hc & 7u];
  };

  double s_rr = 0.0;
  double acc[RY / 2];
#pragma unroll
  for (int m = 0; m < RY / 2; ++m) acc[m] = 0.0;
  auto stencil = [&](int gz, const Plane& pv, const Plane& cu, const Plane& nx) {
    double xs[RY];
#pragma unroll
    for (int r = 0; r < RY; ++r) {
      const double2 pc = cu.v[r + 1], ym = cu.v[r], yp = cu.v[r + 2], zm = pv.v[r + 1], zp = nx.v[r + 1], rc = cu.rc[r];
      const double lf = dpp_shr1(pc.y, lane_bcast(cu.xh, 2 * r)), rt = dpp_shl1(pc.x, lane_bcast(cu.xh, 2 * r + 1));
      const unsigned c0 = cu.cc[r] & 0xffu, c1 = cu.cc[r] >> 8;
      const double s0 = ((((lf + pc.y) + ym.x) + yp.x) + zm.x) + zp.x;
      const double s1 = ((((pc.x + rt) + ym.y) + yp.y) + zm.y) + zp.y;
      const double q0 = c0 ? (double)c0 * pc.x - s0 : 0.0;
      const double q1 = c1 ? (double)c1 * pc.y - s1 : 0.0;
      const double o0 = c0 ? rc.x - q0 : 0.0, o1_ = c1 ? rc.y - q1 : 0.0;
      xs[r] = o0 + o1_;
      s_rr += rc.x * rc.x;
      s_rr += rc.y * rc.y;
    }
    // the 2 x 2 x 2 block sums: x pairs in the lane, y pairs in the wave's strip, z pairs over two steps
    const int kg = g.z0 + gz - 1;  // global fine plane
    const bool store = (kg & 1) != 0 || gz == ge - 1;
#pragma unroll
    for (int m = 0; m < RY / 2; ++m) {
      acc[m] += xs[2 * m] + xs[2 * m + 1];
      if (store) {
        const int I = tx * (kBX / 2) + lane, J = (j0 >> 1) + m;
        if (I < C.cnx && J < C.cny) bcoarse[(long long)I + (long long)C.cnx * J + C.csz * ((kg >> 1) - C.z0c + 1)] = (float)acc[m];
        acc[m] = 0.0;
      }
    }
  };

  // Planes gz - 1, gz, gz + 1 live converted in a ring of four (statically indexed: the loop is unrolled by four); the vectors of planes
  // gz + 2 and gz + 3 are in flight (two requests deep: a step is shorter than a memory round trip), the code bytes and records of
  // plane gz + 4 behind them.
  Plane P[4];
  Meta mr;      // the plane whose vectors are requested next
  Raw raw[2];   // in flight: raw[gz & 1 ...] alternate
  {
    Meta m0, m1, m2, m3;
    meta_issue(gs - 1, false, m0);
    meta_issue(gs, true, m1);
    meta_issue(gs + 1, gs + 1 < ge, m2);
    if (gs + 2 <= ge) meta_issue(gs + 2, gs + 2 < ge, m3);
    if (gs + 3 <= ge) meta_issue(gs + 3, gs + 3 < ge, mr);
    Raw r0, r1;
    vec_issue(gs - 1, false, m0, r0);
    vec_issue(gs, true, m1, r1);
    vec_issue(gs + 1, gs + 1 < ge, m2, raw[0]);
    if (gs + 2 <= ge) vec_issue(gs + 2, gs + 2 < ge, m3, raw[1]);
    convert(gs == 1, r0, P[0]);  // (plane 0 of the slab: a ghost plane kept current by the chunk that marches over it)
    convert(true, r1, P[1]);
  }
  for (int gz0 = gs; gz0 < ge; gz0 += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int gz = gz0 + u;
      if (gz < ge) {
        // plane gz + 1 arrives (requested two steps ago): owned, or the slab's upper ghost plane, or another chunk's first plane
        convert(gz + 1 < ge || gz + 1 == g.nzl + 1, raw[u & 1], P[(u + 2) & 3]);
        __builtin_amdgcn_sched_barrier(0);
        if (gz + 3 <= ge) vec_issue(gz + 3, gz + 3 < ge, mr, raw[u & 1]);
        if (gz + 4 <= ge) meta_issue(gz + 4, gz + 4 < ge, mr);
        __builtin_amdgcn_sched_barrier(0);
        stencil(gz, P[u & 3], P[(u + 1) & 3], P[(u + 2) & 3]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  const double t = block_sum(s_rr, red);
  if (threadIdx.x == 0) { partials[2 * (long long)blockIdx.x] = t; partials[2 * (long long)blockIdx.x + 1] = t; }
}

// can the wave-autonomous forms run on this slab?  (open problems on compacted vectors with segment records; the periodic cell problems
// of the homogenization path keep k_cg1b's periodic instantiations)
bool mgw_ok(const Geom& g, const int* lineoff, const long long* plane_first, const void* seg) {
  return g.tune.mg_wave && !g.periodic && lineoff && plane_first && seg && 8ll * g.plane < kBufMax && (g.z0 & 1) == 0;
}

// residual update + fine pre-smoother + restriction (what launch_mg_pre_update / k_cg1b FORM 2 computes): returns the number of blocks
// (partials {r.r, r.r} per block), -1 when the grid does not fit the partial-sum array
int launch_mgw_pre_update(const Geom& g, const uint8_t* code, const double* r_in, double* r_out, const double* q, const double* hist, int k,
                          MgCoarse c, float* bcoarse, double omega, double* partials, const Ctrl* ctrl, const int* lineoff,
                          const long long* plane_first, const void* seg, hipStream_t s) {
  constexpr int RY = 2, NW = 8;
  const int tiles_x = (g.pitch + kBX - 1) / kBX, tiles_y = (g.ny + RY * NW - 1) / (RY * NW);
  const long long cols = (long long)tiles_x * tiles_y;
  int zchunk, nzc;
  fused_chunks(g, cols, true, &zchunk, &nzc);
  if (!grid_fits(g, cols, nzc)) return -1;
  const int blocks = (int)(cols * nzc);
  hipLaunchKernelGGL((k_mgw<2, RY, NW>), dim3(blocks), dim3(64 * NW), 0, s, g, code, r_in, q, r_out, hist, k, zchunk, tiles_x, tiles_y, nzc, partials,
                     ctrl, lineoff, plane_first, (const uint2*)seg, (g.pitch + kLV - 1) / kLV, bcoarse, c, omega);
  return blocks;
}

}  // namespace oi
