// apply_proto.hip -- round-2 experiments behind the stand-alone stencil apply:
//   (1) what a copy-like stream (1 read : 1 write, 16 B per lane) reaches on this box, by launch shape,
//       loads in flight per lane and cache policy -- the ceiling the 17 B/voxel apply is judged against;
//   (2) a barrier-free, LDS-free form of the masked 7-point apply: one wave owns 128 x RY voxels and
//       marches in z; x neighbours by DPP wave shifts, y neighbours from the wave's own rows plus two halo
//       rows (cache hits: the neighbouring wave reads them as its centre rows), z neighbours in registers.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tools/apply_proto tools/apply_proto.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

typedef double v2d __attribute__((ext_vector_type(2)));

// ------------------------------------------------------------------ copy variants
template <int U, int LDP, int STP>  // policies: 0 plain, 1 nontemporal
__global__ void __launch_bounds__(256) k_copy_gs(double* __restrict__ d, const double* __restrict__ s, long long n2) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  for (; t + (U - 1) * stride < n2; t += U * stride) {
    v2d v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const v2d* p = reinterpret_cast<const v2d*>(s + 2 * (t + u * stride));
      v[u] = LDP ? __builtin_nontemporal_load(p) : *p;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      v2d* p = reinterpret_cast<v2d*>(d + 2 * (t + u * stride));
      if (STP) __builtin_nontemporal_store(v[u], p); else *p = v[u];
    }
  }
  for (; t < n2; t += stride) *reinterpret_cast<v2d*>(d + 2 * t) = *reinterpret_cast<const v2d*>(s + 2 * t);
}

// every wave owns contiguous chunks of U KiB; chunks are dealt round-robin over all waves of the grid
template <int U, int LDP, int STP, int T>
__global__ void __launch_bounds__(T) k_copy_chunk(double* __restrict__ d, const double* __restrict__ s, long long n2) {
  const int lane = threadIdx.x & 63;
  const long long wave = (long long)blockIdx.x * (T / 64) + (threadIdx.x >> 6), nwaves = (long long)gridDim.x * (T / 64);
  const long long nchunks = n2 / (64 * U);
  for (long long c = wave; c < nchunks; c += nwaves) {
    const long long t0 = c * 64 * U + lane;
    v2d v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const v2d* p = reinterpret_cast<const v2d*>(s + 2 * (t0 + u * 64));
      v[u] = LDP ? __builtin_nontemporal_load(p) : *p;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      v2d* p = reinterpret_cast<v2d*>(d + 2 * (t0 + u * 64));
      if (STP) __builtin_nontemporal_store(v[u], p); else *p = v[u];
    }
  }
}

__global__ void __launch_bounds__(256) k_read(const double* __restrict__ s, double* out, long long n2) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  double acc = 0;
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  for (; t + 3 * stride < n2; t += 4 * stride) {
    v2d a = *reinterpret_cast<const v2d*>(s + 2 * t), b = *reinterpret_cast<const v2d*>(s + 2 * (t + stride));
    v2d c = *reinterpret_cast<const v2d*>(s + 2 * (t + 2 * stride)), e = *reinterpret_cast<const v2d*>(s + 2 * (t + 3 * stride));
    acc += (a.x + a.y) + (b.x + b.y) + (c.x + c.y) + (e.x + e.y);
  }
  if (acc == 123.456) out[0] = acc;
}

template <typename F>
static double timeit(F f, int reps) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); f();
  CK(hipEventRecord(a));
  for (int i = 0; i < reps; ++i) f();
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
  return ms / reps;
}

// ------------------------------------------------------------------ DPP wave shifts
// shr1: lane l gets the value of lane l - 1, lane 0 gets `fill`;  shl1: lane l gets lane l + 1, lane 63 gets `fill`
__device__ __forceinline__ double shr1(double v, double fill) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(__double2loint(fill), lo, 0x138, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(__double2hiint(fill), hi, 0x138, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double shl1(double v, double fill) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(__double2loint(fill), lo, 0x130, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(__double2hiint(fill), hi, 0x130, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_bcast(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__global__ void k_dpp_check(double* out) {
  const double v = 100.0 + threadIdx.x;
  out[threadIdx.x] = shr1(v, -1.0);
  out[64 + threadIdx.x] = shl1(v, -2.0);
  out[128 + threadIdx.x] = lane_bcast(v, 5);
}

// ------------------------------------------------------------------ the apply
struct G {
  int nx, ny, nzl, pitch;
  long long plane;
};

template <int RY, int XW>
struct PlaneRegs {
  v2d c[XW][RY + 2];           // rows j0 - 1 .. j0 + RY of this lane's voxel pairs (XW segments of 128 voxels)
  double xh;                   // lanes < 2 RY: x halo of row (lane >> 1), side (lane & 1)
  unsigned short cc[XW][RY];   // code bytes of the centre rows
};

// block -> logical tile.  map 0: every XCD (blocks b, b + 8, ...) gets one contiguous range of tiles (x fastest, then y,
// then z chunk); 1: identity; 2: every XCD gets a contiguous y band of EVERY z chunk (the chip works on one chunk)
__device__ __forceinline__ int tile_of_block(int b, int tiles_xy, int nzc, int map) {
  const int ntiles = tiles_xy * nzc;
  if (map == 0 && (ntiles & 7) == 0) return (b & 7) * (ntiles >> 3) + (b >> 3);
  if (map == 2 && (tiles_xy & 7) == 0) {
    const int per = tiles_xy >> 3, idx = b >> 3;
    return (idx / per) * tiles_xy + (b & 7) * per + idx % per;
  }
  return b;
}

// one wave: (128 XW) (x) x RY (y) voxels, marching over `zchunk` planes.  D = planes loaded ahead of the one computed.
// MODE (timing only, wrong results unless 0): 1 = no x-halo load, 2 = no x/y halo loads, 3 = also no code bytes
template <int RY, int NW, int D, int NT, int MODE = 0, int XW = 1>
__global__ void __launch_bounds__(64 * NW) k_apply7(G g, const uint8_t* __restrict__ code, const double* __restrict__ v,
                                                    double* __restrict__ q, double* __restrict__ partials, int zchunk,
                                                    int tiles_x, int tiles_y, int nzc, int map) {
  constexpr int RING = D + 2;
  constexpr int BX = 128 * XW;
  __shared__ double red[NW];
  const int L = tile_of_block(blockIdx.x, tiles_x * tiles_y, nzc, map);
  const int tx = L % tiles_x, ty = (L / tiles_x) % tiles_y, zc = L / (tiles_x * tiles_y);
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i0 = tx * BX, j0 = (ty * NW + w) * RY;
  bool in_x[XW];
#pragma unroll
  for (int s = 0; s < XW; ++s) in_x[s] = i0 + 128 * s + 2 * lane < g.pitch;
  const int gs = 1 + zc * zchunk, ge = min(g.nzl + 1, gs + zchunk);

  // x-halo duty of lanes < 2 RY
  const int hrow = j0 + (lane >> 1);
  const int hi_ = (lane & 1) ? i0 + BX : i0 - 1;
  const bool hvalid = lane < 2 * RY && hrow < g.ny && hi_ >= 0 && hi_ < g.pitch;

  PlaneRegs<RY, XW> P[RING];

  // scalar (wave-uniform) row bases + one 32-bit lane offset: global_load ... v_off, s[base:base+1]
  const unsigned lane_b16 = 16u * lane, lane_b2 = 2u * lane;
  const long long urow = (long long)j0 * g.pitch + i0;  // uniform: first voxel of the wave's row j0
  const unsigned hoff_b = hvalid ? (unsigned)(((lane >> 1) * g.pitch + (hi_ - i0 + 1)) * 8) : 0u;  // from (row j0, i0 - 1)
  auto load_plane = [&](PlaneRegs<RY, XW>& p, int gz, bool with_halo) {
    const char* vp = reinterpret_cast<const char*>(v + (long long)gz * g.plane + urow);
#pragma unroll
    for (int r = 0; r < RY + 2; ++r) {
      const int j = j0 + r - 1;
      const bool edge = (r == 0 || r == RY + 1);
#pragma unroll
      for (int s = 0; s < XW; ++s) {
        v2d val = {0.0, 0.0};
        if ((!edge || (with_halo && MODE != 2 && MODE != 3)) && j >= 0 && j < g.ny && in_x[s])
          val = *reinterpret_cast<const v2d*>(vp + ((long long)(r - 1) * g.pitch + 128 * s) * 8 + lane_b16);
        p.c[s][r] = val;
      }
    }
    double h = 0.0;
    if (MODE == 0) if (with_halo && hvalid) h = *reinterpret_cast<const double*>(vp - 8 + hoff_b);
    p.xh = h;
    if (with_halo) {
      const char* cp = reinterpret_cast<const char*>(code + (long long)gz * g.plane + urow);
#pragma unroll
      for (int r = 0; r < RY; ++r)
#pragma unroll
        for (int s = 0; s < XW; ++s) {
          unsigned short c = 0;
          if (MODE == 3) c = 0x0203; else
          if (j0 + r < g.ny && in_x[s]) c = *reinterpret_cast<const unsigned short*>(cp + (long long)r * g.pitch + 128 * s + lane_b2);
          p.cc[s][r] = c;
        }
    }
  };

  double pq = 0.0;
  auto compute = [&](int gz, const PlaneRegs<RY, XW>& pv, const PlaneRegs<RY, XW>& cu, const PlaneRegs<RY, XW>& nx) {
    char* qp = reinterpret_cast<char*>(q + (long long)gz * g.plane + urow);
#pragma unroll
    for (int r = 0; r < RY; ++r) {
#pragma unroll
      for (int s = 0; s < XW; ++s) {
        const v2d pc = cu.c[s][r + 1], ym = cu.c[s][r], yp = cu.c[s][r + 2];
        const double fl = s == 0 ? lane_bcast(cu.xh, 2 * r) : lane_bcast(cu.c[s > 0 ? s - 1 : 0][r + 1].y, 63);
        const double fr = s == XW - 1 ? lane_bcast(cu.xh, 2 * r + 1) : lane_bcast(cu.c[s < XW - 1 ? s + 1 : 0][r + 1].x, 0);
        const double lf = shr1(pc.y, fl), rt = shl1(pc.x, fr);
        const unsigned c0 = cu.cc[s][r] & 0xffu, c1 = cu.cc[s][r] >> 8;
        const double s0 = ((((lf + pc.y) + ym.x) + yp.x) + pv.c[s][r + 1].x) + nx.c[s][r + 1].x;
        const double s1 = ((((pc.x + rt) + ym.y) + yp.y) + pv.c[s][r + 1].y) + nx.c[s][r + 1].y;
        const double q0 = c0 ? (double)c0 * pc.x - s0 : 0.0;
        const double q1 = c1 ? (double)c1 * pc.y - s1 : 0.0;
        if (j0 + r < g.ny && in_x[s]) {
          v2d o = {q0, q1};
          v2d* a = reinterpret_cast<v2d*>(qp + ((long long)r * g.pitch + 128 * s) * 8 + lane_b16);
          if (NT) __builtin_nontemporal_store(o, a); else *a = o;
        }
        pq += pc.x * q0;
        pq += pc.y * q1;
      }
    }
  };

  // prologue: planes gs - 1 (z neighbour only), gs .. gs + D - 1
  load_plane(P[0], gs - 1, false);
#pragma unroll
  for (int d = 0; d < D; ++d) {
    if (gs + d <= ge) load_plane(P[1 + d], gs + d, gs + d < ge);
  }
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
  // block partial (fixed order)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) pq += __shfl_down(pq, o, 64);
  if (lane == 0) red[w] = pq;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int k = 0; k < NW; ++k) t += red[k];
    partials[blockIdx.x] = t;
  }
}

// ------------------------------------------------------------------ dense single-kernel CG iteration (structure test)
// r = r_in - alpha q_in; p = r + beta p_in; q = A p; stores r_out, p_out, q_out; 49 B/voxel; five dot products
template <int RY>
struct CgPlane {
  v2d p[RY + 2];
  v2d r[RY];
  double xh;
  unsigned short cc[RY];
};
template <int RY, int NW, int D>
__global__ void __launch_bounds__(64 * NW) k_cg7d(G g, const uint8_t* __restrict__ code, const double* __restrict__ r_in,
                                                   const double* __restrict__ q_in, const double* __restrict__ p_in,
                                                   double* __restrict__ r_out, double* __restrict__ q_out,
                                                   double* __restrict__ p_out, double alpha, double beta,
                                                   double* __restrict__ partials, int zchunk, int tiles_x, int tiles_y, int nzc,
                                                   int map) {
  constexpr int RING = D + 2;
  __shared__ double red[NW][5];
  const int L = tile_of_block(blockIdx.x, tiles_x * tiles_y, nzc, map);
  const int tx = L % tiles_x, ty = (L / tiles_x) % tiles_y, zc = L / (tiles_x * tiles_y);
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i0 = tx * 128, j0 = (ty * NW + w) * RY;
  const bool in_x = i0 + 2 * lane < g.pitch;
  const int gs = 1 + zc * zchunk, ge = min(g.nzl + 1, gs + zchunk);
  const int hrow = j0 + (lane >> 1);
  const int hi_ = (lane & 1) ? i0 + 128 : i0 - 1;
  const bool hvalid = lane < 2 * RY && hrow < g.ny && hi_ >= 0 && hi_ < g.pitch;
  const unsigned lane_b16 = 16u * lane, lane_b2 = 2u * lane;
  const long long urow = (long long)j0 * g.pitch + i0;
  const unsigned hoff_b = hvalid ? (unsigned)(((lane >> 1) * g.pitch + (hi_ - i0 + 1)) * 8) : 0u;
  CgPlane<RY> P[RING];
  // kind 0: z neighbour only (centre rows, nothing stored); 1: output plane
  auto load_plane = [&](CgPlane<RY>& pl, int gz, bool out) {
    const long long po = (long long)gz * g.plane + urow;
    const char* rp = reinterpret_cast<const char*>(r_in + po);
    const char* qp = reinterpret_cast<const char*>(q_in + po);
    const char* pp = reinterpret_cast<const char*>(p_in + po);
    char* ro = reinterpret_cast<char*>(r_out + po);
    char* po_ = reinterpret_cast<char*>(p_out + po);
#pragma unroll
    for (int r = 0; r < RY + 2; ++r) {
      const int j = j0 + r - 1;
      const bool edge = (r == 0 || r == RY + 1);
      v2d pv = {0.0, 0.0}, rv = {0.0, 0.0};
      if ((!edge || out) && j >= 0 && j < g.ny && in_x) {
        const long long off = (long long)(r - 1) * g.pitch * 8 + lane_b16;
        const v2d a = *reinterpret_cast<const v2d*>(rp + off), b = *reinterpret_cast<const v2d*>(qp + off);
        const v2d c = *reinterpret_cast<const v2d*>(pp + off);
        rv.x = a.x - alpha * b.x; rv.y = a.y - alpha * b.y;
        pv.x = rv.x + beta * c.x; pv.y = rv.y + beta * c.y;
        if (out && !edge) {
          __builtin_nontemporal_store(rv, reinterpret_cast<v2d*>(ro + off));
          __builtin_nontemporal_store(pv, reinterpret_cast<v2d*>(po_ + off));
        }
      }
      pl.p[r] = pv;
      if (!edge) pl.r[r - 1] = rv;
    }
    double h = 0.0;
    if (out && hvalid) {
      const double a = *reinterpret_cast<const double*>(rp - 8 + hoff_b), b = *reinterpret_cast<const double*>(qp - 8 + hoff_b);
      const double c = *reinterpret_cast<const double*>(pp - 8 + hoff_b);
      h = (a - alpha * b) + beta * c;
    }
    pl.xh = h;
    if (out) {
      const char* cp = reinterpret_cast<const char*>(code + po);
#pragma unroll
      for (int r = 0; r < RY; ++r) {
        unsigned short c = 0;
        if (j0 + r < g.ny && in_x) c = *reinterpret_cast<const unsigned short*>(cp + (long long)r * g.pitch + lane_b2);
        pl.cc[r] = c;
      }
    }
  };
  double s_rr = 0, s_pq = 0, s_zq = 0, s_qdq = 0;
  auto compute = [&](int gz, const CgPlane<RY>& pv, const CgPlane<RY>& cu, const CgPlane<RY>& nx) {
    char* qo = reinterpret_cast<char*>(q_out + (long long)gz * g.plane + urow);
#pragma unroll
    for (int r = 0; r < RY; ++r) {
      const v2d pc = cu.p[r + 1], ym = cu.p[r], yp = cu.p[r + 2], rc = cu.r[r];
      const double lf = shr1(pc.y, lane_bcast(cu.xh, 2 * r)), rt = shl1(pc.x, lane_bcast(cu.xh, 2 * r + 1));
      const unsigned c0 = cu.cc[r] & 0xffu, c1 = cu.cc[r] >> 8;
      const double s0 = ((((lf + pc.y) + ym.x) + yp.x) + pv.p[r + 1].x) + nx.p[r + 1].x;
      const double s1 = ((((pc.x + rt) + ym.y) + yp.y) + pv.p[r + 1].y) + nx.p[r + 1].y;
      const double q0 = c0 ? (double)c0 * pc.x - s0 : 0.0;
      const double q1 = c1 ? (double)c1 * pc.y - s1 : 0.0;
      if (j0 + r < g.ny && in_x) {
        v2d o = {q0, q1};
        __builtin_nontemporal_store(o, reinterpret_cast<v2d*>(qo + (long long)r * g.pitch * 8 + lane_b16));
      }
      s_rr += rc.x * rc.x; s_rr += rc.y * rc.y;
      s_pq += pc.x * q0; s_pq += pc.y * q1;
      s_zq += rc.x * q0; s_zq += rc.y * q1;
      s_qdq += q0 * q0; s_qdq += q1 * q1;
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
  double sums[4] = {s_rr, s_pq, s_zq, s_qdq};
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
    partials[5 * blockIdx.x + threadIdx.x] = t;
  }
}

static void cpu_apply(const G& g, const std::vector<uint8_t>& code, const std::vector<double>& v, std::vector<double>& q) {
  for (int gz = 1; gz <= g.nzl; ++gz)
    for (int j = 0; j < g.ny; ++j)
      for (int i = 0; i < g.pitch; ++i) {
        const long long e = (long long)gz * g.plane + (long long)j * g.pitch + i;
        const unsigned c = code[e];
        const double xm = i > 0 ? v[e - 1] : 0.0, xp = i + 1 < g.pitch ? v[e + 1] : 0.0;
        const double ym = j > 0 ? v[e - g.pitch] : 0.0, yp = j < g.ny - 1 ? v[e + g.pitch] : 0.0;
        const double s = ((((xm + xp) + ym) + yp) + v[e - g.plane]) + v[e + g.plane];
        q[e] = c ? (double)c * v[e] - s : 0.0;
      }
}

struct Variant {
  const char* name;
  void (*launch)(int blocks, G, const uint8_t*, const double*, double*, double*, int, int, int, int, int);
  int ry, nw, xw;
};
template <int RY, int NW, int D, int NT, int MODE = 0, int XW = 1>
static void launch_v(int blocks, G g, const uint8_t* code, const double* v, double* q, double* partials, int zchunk, int tx,
                     int ty, int nzc, int map) {
  hipLaunchKernelGGL((k_apply7<RY, NW, D, NT, MODE, XW>), dim3(blocks), dim3(64 * NW), 0, 0, g, code, v, q, partials, zchunk, tx, ty, nzc, map);
}

static uint64_t rng_state = 88172645463325252ull;
static inline uint64_t xorshift() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }

int main(int argc, char** argv) {
  const bool skip_copy = argc > 1 && !strcmp(argv[1], "nocopy");
  // ---- DPP semantics
  {
    double* d; CK(hipMalloc(&d, 192 * 8));
    hipLaunchKernelGGL(k_dpp_check, dim3(1), dim3(64), 0, 0, d);
    double h[192]; CK(hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
    bool ok = h[0] == -1.0 && h[64 + 63] == -2.0;
    for (int l = 1; l < 64; ++l) ok = ok && h[l] == 100.0 + l - 1;
    for (int l = 0; l < 63; ++l) ok = ok && h[64 + l] == 100.0 + l + 1;
    for (int l = 0; l < 64; ++l) ok = ok && h[128 + l] == 105.0;
    printf("dpp wave_shr/wave_shl/readlane semantics: %s\n", ok ? "OK" : "WRONG");
    if (!ok) { for (int l = 0; l < 192; ++l) printf("%g ", h[l]); printf("\n"); return 1; }
    CK(hipFree(d));
  }
  static const Variant variants[] = {
      {"RY4 NW4 D2 nt", launch_v<4, 4, 2, 1>, 4, 4, 1}, {"RY4 NW8 D2 nt", launch_v<4, 8, 2, 1>, 4, 8, 1},
      {"RY4 NW8 D2 st", launch_v<4, 8, 2, 0>, 4, 8, 1}, {"RY4 NW8 D1 nt", launch_v<4, 8, 1, 1>, 4, 8, 1},
      {"RY4 NW16 D1 nt", launch_v<4, 16, 1, 1>, 4, 16, 1}, {"RY4 NW16 D2 nt", launch_v<4, 16, 2, 1>, 4, 16, 1},
      {"RY8 NW4 D1 nt", launch_v<8, 4, 1, 1>, 8, 4, 1},
      {"RY2 XW2 NW8 D2 nt", launch_v<2, 8, 2, 1, 0, 2>, 2, 8, 2}, {"RY4 XW2 NW4 D1 nt", launch_v<4, 4, 1, 1, 0, 2>, 4, 4, 2},
      {"RY4 XW2 NW4 D1 st", launch_v<4, 4, 1, 0, 0, 2>, 4, 4, 2}, {"RY4 XW2 NW8 D1 nt", launch_v<4, 8, 1, 1, 0, 2>, 4, 8, 2},
      {"RY4 XW2 NW2 D1 nt", launch_v<4, 2, 1, 1, 0, 2>, 4, 2, 2}, {"RY3 XW2 NW4 D1 nt", launch_v<3, 4, 1, 1, 0, 2>, 3, 4, 2},
      // timing only from here
      {"RY4 NW8 D2 nt nohalo", launch_v<4, 8, 2, 1, 2>, 4, 8, 1}, {"RY4 NW8 D2 nt nohalo nocode", launch_v<4, 8, 2, 1, 3>, 4, 8, 1},
      {"RY4 XW2 NW4 D1 nt nohalo nocode", launch_v<4, 4, 1, 1, 3, 2>, 4, 4, 2},
  };
  const int nvar = sizeof variants / sizeof variants[0], nchecked = nvar - 3;

  // ---- correctness on ragged shapes
  for (int shape = 0; shape < 3; ++shape) {
    G g;
    g.nx = shape == 0 ? 301 : shape == 1 ? 128 : 517; g.ny = shape == 0 ? 70 : shape == 1 ? 33 : 19; g.nzl = shape == 0 ? 21 : shape == 1 ? 9 : 13;
    g.pitch = (g.nx + 1) & ~1; g.plane = (long long)g.pitch * g.ny;
    const long long n = (g.nzl + 2) * g.plane;
    std::vector<double> v(n, 0.0), qref(n, 0.0), qh(n);
    std::vector<uint8_t> code(n, 0);
    for (int gz = 0; gz <= g.nzl + 1; ++gz)
      for (int j = 0; j < g.ny; ++j)
        for (int i = 0; i < g.nx; ++i) {
          const long long e = (long long)gz * g.plane + (long long)j * g.pitch + i;
          v[e] = (double)(xorshift() >> 11) / 9007199254740992.0 - 0.5;
          if (gz >= 1 && gz <= g.nzl) code[e] = (uint8_t)(xorshift() % 7);
        }
    cpu_apply(g, code, v, qref);
    double *dv, *dq, *dp; uint8_t* dc;
    CK(hipMalloc(&dv, n * 8)); CK(hipMalloc(&dq, n * 8)); CK(hipMalloc(&dc, n)); CK(hipMalloc(&dp, 65536 * 8));
    CK(hipMemcpy(dv, v.data(), n * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dc, code.data(), n, hipMemcpyHostToDevice));
    for (int k = 0; k < nchecked; ++k)
      for (int zchunk : {4, 64}) for (int map : {0, 2}) {
        const Variant& V = variants[k];
        const int by = V.ry * V.nw, bx = 128 * V.xw, tx = (g.pitch + bx - 1) / bx, ty = (g.ny + by - 1) / by, nzc = (g.nzl + zchunk - 1) / zchunk;
        CK(hipMemset(dq, 0xff, n * 8));
        V.launch(tx * ty * nzc, g, dc, dv, dq, dp, zchunk, tx, ty, nzc, map);
        CK(hipMemcpy(qh.data(), dq, n * 8, hipMemcpyDeviceToHost));
        long long bad = 0;
        for (int gz = 1; gz <= g.nzl; ++gz)
          for (int j = 0; j < g.ny; ++j)
            for (int i = 0; i < g.pitch; ++i) {
              const long long e = (long long)gz * g.plane + (long long)j * g.pitch + i;
              if (memcmp(&qh[e], &qref[e], 8) != 0) { if (bad < 3) printf("  mismatch at (%d,%d,%d): %.17g vs %.17g\n", i, j, gz, qh[e], qref[e]); ++bad; }
            }
        std::vector<double> ph(tx * ty * nzc);
        CK(hipMemcpy(ph.data(), dp, ph.size() * 8, hipMemcpyDeviceToHost));
        double dot = 0, dref = 0;
        for (double t : ph) dot += t;
        for (long long e = 0; e < n; ++e) dref += v[e] * qref[e];
        if (bad || fabs(dot - dref) > 1e-9 * fabs(dref) + 1e-9)
          printf("shape %d %-16s zchunk %3d map %d: %lld mismatches, dot %.12g vs %.12g\n", shape, V.name, zchunk, map, bad, dot, dref);
      }
    printf("shape %d (%d x %d x %d): all variants checked\n", shape, g.nx, g.ny, g.nzl);
    CK(hipFree(dv)); CK(hipFree(dq)); CK(hipFree(dc)); CK(hipFree(dp));
  }

  // ---- big arrays
  const int N = 1024;
  G g; g.nx = N; g.ny = N; g.nzl = N; g.pitch = N; g.plane = (long long)N * N;
  const long long n = (long long)(N + 2) * g.plane;
  double *dv, *dq, *dp; uint8_t* dc;
  CK(hipMalloc(&dv, n * 8)); CK(hipMalloc(&dq, n * 8)); CK(hipMalloc(&dc, n)); CK(hipMalloc(&dp, (1 << 20) * 8));
  {  // fill v with something nonzero, code with 0..6
    std::vector<double> hv(g.plane); std::vector<uint8_t> hc(g.plane);
    for (long long e = 0; e < g.plane; ++e) { hv[e] = (double)(xorshift() >> 11) / 9007199254740992.0; hc[e] = (uint8_t)(xorshift() % 7); }
    for (int gz = 0; gz < N + 2; ++gz) {
      CK(hipMemcpy(dv + gz * g.plane, hv.data(), g.plane * 8, hipMemcpyHostToDevice));
      CK(hipMemcpy(dc + gz * g.plane, hc.data(), g.plane, hipMemcpyHostToDevice));
    }
    CK(hipMemset(dq, 0, n * 8));
  }
  const long long nn = (long long)N * g.plane, n2 = nn / 2;
  const int reps = 10;
  if (!skip_copy) {
    double ms;
    ms = timeit([&] { hipLaunchKernelGGL(k_read, dim3(2048), dim3(256), 0, 0, dv, dp, n2); }, reps);
    printf("read  U4 grid 2048              %7.3f ms %7.1f GB/s\n", ms, 8.0 * nn / ms / 1e6);
#define COPY_GS(U, LP, SP, B)                                                                                              \
  ms = timeit([&] { hipLaunchKernelGGL((k_copy_gs<U, LP, SP>), dim3(B), dim3(256), 0, 0, dq, dv, n2); }, reps);           \
  printf("copy grid-stride U%d ld%d st%d grid %6d  %7.3f ms %7.1f GB/s\n", U, LP, SP, B, ms, 16.0 * nn / ms / 1e6);
#define COPY_CH(U, LP, SP, T, B)                                                                                           \
  ms = timeit([&] { hipLaunchKernelGGL((k_copy_chunk<U, LP, SP, T>), dim3(B), dim3(T), 0, 0, dq, dv, n2); }, reps);       \
  printf("copy chunk U%d ld%d st%d T%4d grid %6d  %7.3f ms %7.1f GB/s\n", U, LP, SP, T, B, ms, 16.0 * nn / ms / 1e6);
    for (int B : {1024, 16384, 65536}) {
      COPY_GS(1, 0, 0, B) COPY_GS(4, 0, 0, B) COPY_GS(4, 1, 1, B)
    }
    for (int rep = 0; rep < 2; ++rep)
    for (int B : {256, 512, 1024, 4096, 8192, 32768}) {
      COPY_CH(8, 0, 0, 256, B) COPY_CH(8, 1, 1, 256, B) COPY_CH(16, 1, 1, 256, B) COPY_CH(16, 0, 0, 256, B)
      COPY_CH(8, 1, 1, 512, B) COPY_CH(8, 1, 1, 1024, B) COPY_CH(2, 1, 1, 256, B)
    }
    // destination shifted against the source by odd multiples of 4 KiB / 64 KiB (channel interleave)
    for (long long sh : {0ll, 512ll, 8192ll + 512ll, (1ll << 20) + 512ll}) {
      double* dq2 = dq + sh;
      ms = timeit([&] { hipLaunchKernelGGL((k_copy_gs<4, 0, 0>), dim3(2048), dim3(256), 0, 0, dq2, dv, n2); }, reps);
      printf("copy grid-stride U4 dst shift %8lld B  %7.3f ms %7.1f GB/s\n", sh * 8, ms, 16.0 * nn / ms / 1e6);
    }
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipMemcpy(dq, dv, nn * 8, hipMemcpyDeviceToDevice));
    CK(hipEventRecord(a));
    for (int r = 0; r < 5; ++r) CK(hipMemcpyAsync(dq, dv, nn * 8, hipMemcpyDeviceToDevice, 0));
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float msf; CK(hipEventElapsedTime(&msf, a, b));
    printf("hipMemcpyDtoD                   %7.3f ms %7.1f GB/s\n", msf / 5, 16.0 * nn / (msf / 5) / 1e6);
  }
  // ---- apply variants at 1024^3
  if (!(argc > 2 && !strcmp(argv[2], "cgonly")))
  for (int k = 0; k < nvar; ++k)
    for (int zchunk : {8, 12, 16, 24, 32, 64}) for (int map : {2}) {
      const Variant& V = variants[k];
      const int by = V.ry * V.nw, bx = 128 * V.xw, tx = (g.pitch + bx - 1) / bx, ty = (g.ny + by - 1) / by, nzc = (g.nzl + zchunk - 1) / zchunk;
      const int blocks = tx * ty * nzc;
      if (blocks > (1 << 20)) continue;
      const double ms = timeit([&] { V.launch(blocks, g, dc, dv, dq, dp, zchunk, tx, ty, nzc, map); }, reps);
      printf("apply7 %-32s zchunk %3d map %d blocks %6d  %7.3f ms  %7.1f GB/s (17 B/voxel)  frac of 8 TB/s %.3f\n", V.name, zchunk, map,
             blocks, ms, 17.0 * nn / ms / 1e6, 17.0 * nn / ms / 1e6 / 8000.0);
      fflush(stdout);
    }
  {  // dense CG-iteration structure test: three more vectors
    double *r2, *q2, *p2, *p1;
    CK(hipMalloc(&r2, n * 8)); CK(hipMalloc(&q2, n * 8)); CK(hipMalloc(&p2, n * 8)); CK(hipMalloc(&p1, n * 8));
    CK(hipMemset(r2, 0, n * 8)); CK(hipMemset(q2, 0, n * 8)); CK(hipMemset(p2, 0, n * 8)); CK(hipMemcpy(p1, dv, n * 8, hipMemcpyDeviceToDevice));
#define CG7(RY, NW, D, ZC, MAP)                                                                                              \
    {                                                                                                                        \
      const int by = RY * NW, tx = (g.pitch + 127) / 128, ty = (g.ny + by - 1) / by, nzc = (g.nzl + ZC - 1) / ZC;            \
      const int blocks = tx * ty * nzc;                                                                                      \
      const double ms = timeit([&] { hipLaunchKernelGGL((k_cg7d<RY, NW, D>), dim3(blocks), dim3(64 * NW), 0, 0, g, dc, dv, dq, p1, r2, q2, p2, 0.3, 0.2, dp, ZC, tx, ty, nzc, MAP); }, reps); \
      printf("cg7d RY%d NW%d D%d zchunk %3d map %d blocks %6d  %7.3f ms  %7.1f GB/s (49 B/voxel)  frac %.3f\n", RY, NW, D, ZC, MAP, blocks, ms, 49.0 * nn / ms / 1e6, 49.0 * nn / ms / 1e6 / 8000.0); \
      fflush(stdout);                                                                                                        \
    }
    CG7(4, 4, 1, 8, 2) CG7(4, 4, 1, 16, 2) CG7(4, 4, 1, 32, 2) CG7(4, 4, 1, 64, 0)
    CG7(4, 8, 1, 8, 2) CG7(4, 8, 1, 16, 2) CG7(4, 8, 1, 32, 2)
    CG7(2, 8, 1, 8, 2) CG7(2, 8, 1, 16, 2) CG7(2, 8, 2, 8, 2) CG7(2, 8, 2, 16, 2) CG7(2, 4, 2, 8, 2)
    CG7(2, 16, 1, 8, 2) CG7(2, 16, 2, 8, 2) CG7(4, 4, 2, 8, 2) CG7(4, 4, 2, 16, 2)
    CG7(3, 4, 1, 8, 2) CG7(3, 8, 1, 8, 2)
  }
  return 0;
}
