// oi_k_setup.hip -- process defaults of the launch shapes; phase bytes -> open bits -> flood fill -> active mask -> code bytes;
// the compacted line lists; boundary fluxes; seeded synthetic structures; pitch copies; remspot.  See oi_kern.h.
#include "oi_kern.h"

namespace oi {

static std::mutex g_tune_mutex;
static Tune g_tune_defaults;
static bool g_tune_env_read = false;
Tune default_tune() {
  std::lock_guard<std::mutex> lock(g_tune_mutex);
  if (!g_tune_env_read) {  // environment overrides, read ONCE per process (experiments and cross-checks)
    g_tune_env_read = true;
    if (const char* e = getenv("OI_TILE_MAP")) g_tune_defaults.tile_map = atoi(e);
    if (const char* e = getenv("OI_ZCHUNK")) g_tune_defaults.zchunk = atoi(e);
    if (const char* e = getenv("OI_APPLY_KERNEL")) g_tune_defaults.apply_kernel = atoi(e);
    if (getenv("OI_MG_PLAIN")) g_tune_defaults.mg_plain = 1;
    if (getenv("OI_MG_NOVEC4")) g_tune_defaults.mg_novec4 = 1;
    if (getenv("OI_MG_DEBUG")) g_tune_defaults.mg_debug = 1;
    if (const char* e = getenv("OI_MG_REPLICATE_CELLS")) g_tune_defaults.mg_replicate_cells = atoll(e);
    if (const char* e = getenv("OI_MG_WMASK")) g_tune_defaults.mg_wmask = atoi(e);
    if (const char* e = getenv("OI_MG_SCALE")) g_tune_defaults.mg_scale = (float)atof(e);
    if (getenv("OI_FORCE_COMM")) g_tune_defaults.force_comm = 1;
    if (const char* e = getenv("OI_MG_NU_COARSE")) g_tune_defaults.mg_nu_coarse = atoi(e) >= 2 ? 2 : (atoi(e) == 1 ? 1 : 0);
    if (const char* e = getenv("OI_PLACE_TRIALS")) g_tune_defaults.place_trials = atoi(e);
    if (const char* e = getenv("OI_PLACE_SLOTS")) g_tune_defaults.place_slots = atoi(e);
    if (const char* e = getenv("OI_PLACE_SPACER_GB")) g_tune_defaults.place_spacer_gb = (float)atof(e);
    if (const char* e = getenv("OI_PLACE_EXTENT_GB")) g_tune_defaults.place_extent_gb = (float)atof(e);
    if (const char* e = getenv("OI_PLACE_MIN_MB")) g_tune_defaults.place_min_mb = atoi(e);
    if (const char* e = getenv("OI_PLACE_RETRY_GBPS")) g_tune_defaults.place_retry_gbps = (float)atof(e);
    if (const char* e = getenv("OI_PLACE_CACHE")) g_tune_defaults.place_cache = atoi(e) != 0;
  }
  return g_tune_defaults;
}
void set_default_tune(const Tune& t) {
  (void)default_tune();  // (environment first, so that an explicit call wins)
  std::lock_guard<std::mutex> lock(g_tune_mutex);
  g_tune_defaults = t;
}


// phase bytes -> "open" bit planes (open = phase == phase_id), and the phase count of
// VolumeFraction::value (reference src/props/VolumeFraction.cpp:22-66).
// One wavefront builds one 64-voxel word with a ballot: 64 coalesced bytes in.
// --------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_pack_open(const uint8_t* __restrict__ phase, Geom g, int phase_id,
                                                    uint64_t* __restrict__ open,
                                                    unsigned long long* __restrict__ n_phase) {
  const int lane = threadIdx.x & 63;
  const long long nw_owned = (long long)g.nzl * g.wplane;
  const long long wave0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
  unsigned long long cnt = 0;
  for (long long w = wave0; w < nw_owned; w += nwaves) {
    const int ww = (int)(w % g.W);
    const long long row = w / g.W;  // k*ny + j over owned planes
    const int i = ww * 64 + lane;
    bool o = false;
    if (i < g.nx) o = phase[row * g.nx + i] == (uint8_t)phase_id;
    const unsigned long long word = __ballot(o);
    if (lane == 0) {
      open[g.wplane + w] = word;  // owned planes start after the lo ghost plane
      cnt += __popcll(word);
    }
  }
  __shared__ double red[4];
  // integer count through the fp64 block sum is exact below 2^53
  const double tot = block_sum((double)cnt, red);
  if (threadIdx.x == 0 && tot != 0.0) atomicAdd(n_phase, (unsigned long long)tot);
}

void launch_pack_open(const uint8_t* phase, const Geom& g, int phase_id, uint64_t* open,
                      unsigned long long* n_phase, hipStream_t s) {
  const long long nw = (long long)g.nzl * g.wplane;
  const int blocks = cap_blocks(nw, 4, 8192);
  hipLaunchKernelGGL(k_pack_open, dim3(blocks), dim3(256), 0, s, phase, g, phase_id, open, n_phase);
}

// --------------------------------------------------------------------------------------
// Flood-fill seeds: cells of the phase on the lo / hi face of the flow axis
// (reference TortuosityHypre.cpp:424-449).  reach arrays must be zeroed before.
// --------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_seed(Geom g, const uint64_t* __restrict__ open,
                                               uint64_t* __restrict__ rlo, uint64_t* __restrict__ rhi) {
  const long long nw_owned = (long long)g.nzl * g.wplane;
  for (long long w = (long long)blockIdx.x * blockDim.x + threadIdx.x; w < nw_owned;
       w += (long long)gridDim.x * blockDim.x) {
    const int ww = (int)(w % g.W);
    const int j = (int)((w / g.W) % g.ny);
    const int kz = g.z0 + (int)(w / g.wplane);
    const uint64_t o = open[g.wplane + w];
    uint64_t lo = 0, hi = 0;
    if (g.dir == 0) {
      if (ww == 0) lo = o & 1ULL;
      if (ww == ((g.nx - 1) >> 6)) hi = o & (1ULL << ((g.nx - 1) & 63));
    } else if (g.dir == 1) {
      if (j == 0) lo = o;
      if (j == g.ny - 1) hi = o;
    } else {
      if (kz == 0) lo = o;
      if (kz == g.nz - 1) hi = o;
    }
    rlo[g.wplane + w] = lo;
    rhi[g.wplane + w] = hi;
  }
}

void launch_seed(const Geom& g, const uint64_t* open, uint64_t* rlo, uint64_t* rhi, hipStream_t s) {
  const int blocks = cap_blocks((long long)g.nzl * g.wplane, 256, 4096);
  hipLaunchKernelGGL(k_seed, dim3(blocks), dim3(256), 0, s, g, open, rlo, rhi);
}

// --------------------------------------------------------------------------------------
// One flood-fill sweep on bit planes (reference parallelFloodFill,
// TortuosityHypre.cpp:336-380: "a phase cell becomes reached if a 6-neighbour is
// reached", iterated to the fixed point).  Per 64-voxel word: OR in the y/z neighbour
// words and the x neighbours across the word boundary, then flood every run of open
// bits that holds a reached bit with the carry trick  ((o + s) ^ o) & o | s  in both bit
// directions.  Monotone, so in-place racy reads only delay, never corrupt.
// Each thread walks TJ rows up and down (Gauss-Seidel in y inside the strip).
// --------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t fill_runs(uint64_t o, uint64_t s) {
  const uint64_t up = (((o + s) ^ o) & o) | s;
  const uint64_t ro = __brevll(o), rs = __brevll(s);
  const uint64_t dn = __brevll((((ro + rs) ^ ro) & ro) | rs);
  return up | dn;
}

constexpr int kFloodTJ = 8;

// reverse: the items are walked from the LAST plane down.  Blocks start in index order and every item reads its z neighbours'
// current words, so a forward sweep carries a front upwards through many planes and downwards through one; alternating the
// direction (round 4) serves the fill from the hi face as well -- Z-direction masks at 1024^3: 0.146 -> see profiles/r04.
// The fixed point does not depend on the order (a monotone closure): the mask is the same bit for bit.
__global__ void __launch_bounds__(256) k_flood_sweep(Geom g, const uint64_t* __restrict__ open,
                                                      uint64_t* rlo, uint64_t* rhi, int* __restrict__ changed, int reverse) {
  const int nstrips = (g.ny + kFloodTJ - 1) / kFloodTJ;
  const long long nitems = (long long)g.nzl * nstrips * g.W;
  bool any = false;
  for (long long tt = (long long)blockIdx.x * blockDim.x + threadIdx.x; tt < nitems;
       tt += (long long)gridDim.x * blockDim.x) {
    const long long t = reverse ? nitems - 1 - tt : tt;
    const int ww = (int)(t % g.W);
    const int strip = (int)((t / g.W) % nstrips);
    const int gz = 1 + (int)(t / ((long long)g.W * nstrips));
    const int j0 = strip * kFloodTJ, j1 = min(g.ny, j0 + kFloodTJ);
    for (int pass = 0; pass < 2; ++pass) {
      for (int jj = 0; jj < j1 - j0; ++jj) {
        const int j = pass == 0 ? j0 + jj : j1 - 1 - jj;
        const long long c = (long long)gz * g.wplane + (long long)j * g.W + ww;
        const uint64_t o = open[c];
        if (o == 0) continue;
        const uint64_t a0 = rlo[c], b0 = rhi[c];
        uint64_t a = a0 | rlo[c - g.wplane] | rlo[c + g.wplane];
        uint64_t b = b0 | rhi[c - g.wplane] | rhi[c + g.wplane];
        if (j > 0) { a |= rlo[c - g.W]; b |= rhi[c - g.W]; }
        if (j < g.ny - 1) { a |= rlo[c + g.W]; b |= rhi[c + g.W]; }
        if (ww > 0) { a |= rlo[c - 1] >> 63; b |= rhi[c - 1] >> 63; }
        if (ww < g.W - 1) { a |= rlo[c + 1] << 63; b |= rhi[c + 1] << 63; }
        a = fill_runs(o, a & o);
        b = fill_runs(o, b & o);
        if (a != a0) { rlo[c] = a; any = true; }
        if (b != b0) { rhi[c] = b; any = true; }
      }
    }
  }
  if (any) *changed = 1;
}

// The same step with every thread walking its word COLUMN through all planes of the slab, up and then down (Gauss-Seidel in z):
// one launch carries a front along z as far as the pore stays open above / below, where the plane-parallel sweep above moves
// it a few planes per launch (the Z-direction mask of a 1024^3 pack took ~450 of those sweeps, 0.135-0.146 s; round 4).
// ny * W threads only, each with nzl dependent steps: dearer per launch (about 2 ms at 1024^3), so one per round of sweeps.
__global__ void __launch_bounds__(256) k_flood_zwalk(Geom g, const uint64_t* __restrict__ open,
                                                      uint64_t* rlo, uint64_t* rhi, int* __restrict__ changed) {
  const long long ncols = (long long)g.ny * g.W;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= ncols) return;
  const int ww = (int)(t % g.W), j = (int)(t / g.W);
  bool any = false;
  for (int pass = 0; pass < 2; ++pass) {
    for (int kk = 0; kk < g.nzl; ++kk) {
      const int gz = pass == 0 ? 1 + kk : g.nzl - kk;
      const long long c = (long long)gz * g.wplane + (long long)j * g.W + ww;
      const uint64_t o = open[c];
      if (o == 0) continue;
      const uint64_t a0 = rlo[c], b0 = rhi[c];
      uint64_t a = a0 | rlo[c - g.wplane] | rlo[c + g.wplane];
      uint64_t b = b0 | rhi[c - g.wplane] | rhi[c + g.wplane];
      if (j > 0) { a |= rlo[c - g.W]; b |= rhi[c - g.W]; }
      if (j < g.ny - 1) { a |= rlo[c + g.W]; b |= rhi[c + g.W]; }
      if (ww > 0) { a |= rlo[c - 1] >> 63; b |= rhi[c - 1] >> 63; }
      if (ww < g.W - 1) { a |= rlo[c + 1] << 63; b |= rhi[c + 1] << 63; }
      a = fill_runs(o, a & o);
      b = fill_runs(o, b & o);
      if (a != a0) { rlo[c] = a; any = true; }
      if (b != b0) { rhi[c] = b; any = true; }
    }
  }
  if (any) *changed = 1;
}
void launch_flood_zwalk(const Geom& g, const uint64_t* open, uint64_t* rlo, uint64_t* rhi, int* changed, hipStream_t s) {
  const long long ncols = (long long)g.ny * g.W;
  hipLaunchKernelGGL(k_flood_zwalk, dim3((unsigned)((ncols + 255) / 256)), dim3(256), 0, s, g, open, rlo, rhi, changed);
}

void launch_flood_sweep(const Geom& g, const uint64_t* open, uint64_t* rlo, uint64_t* rhi, int* changed,
                        hipStream_t s, int reverse) {
  const int nstrips = (g.ny + kFloodTJ - 1) / kFloodTJ;
  const int blocks = cap_blocks((long long)g.nzl * nstrips * g.W, 256, 8192);
  hipLaunchKernelGGL(k_flood_sweep, dim3(blocks), dim3(256), 0, s, g, open, rlo, rhi, changed, reverse);
}

// active = reached from lo AND reached from hi (reference TortuosityHypre.cpp:526-538),
// written over rlo (ghost planes included); counts active cells (owned planes) and the
// active cells on the two Dirichlet faces (they give ||b||).
__global__ void __launch_bounds__(256) k_mask_and(Geom g, uint64_t* rlo, const uint64_t* __restrict__ rhi,
                                                   unsigned long long* __restrict__ counts) {
  unsigned long long na = 0, nlo = 0, nhi = 0;
  for (long long w = (long long)blockIdx.x * blockDim.x + threadIdx.x; w < g.nwords;
       w += (long long)gridDim.x * blockDim.x) {
    const uint64_t a = rlo[w] & rhi[w];
    rlo[w] = a;
    const int gz = (int)(w / g.wplane);
    if (gz >= 1 && gz <= g.nzl && a) {
      na += __popcll(a);
      const int ww = (int)(w % g.W);
      const int j = (int)((w / g.W) % g.ny);
      const int kz = g.z0 + gz - 1;
      if (g.dir == 0) {
        if (ww == 0) nlo += a & 1ULL;
        if (ww == ((g.nx - 1) >> 6)) nhi += (a >> ((g.nx - 1) & 63)) & 1ULL;
      } else if (g.dir == 1) {
        if (j == 0) nlo += __popcll(a);
        if (j == g.ny - 1) nhi += __popcll(a);
      } else {
        if (kz == 0) nlo += __popcll(a);
        if (kz == g.nz - 1) nhi += __popcll(a);
      }
    }
  }
  __shared__ double red[4];
  const double ta = block_sum((double)na, red);
  const double tl = block_sum((double)nlo, red);
  const double th = block_sum((double)nhi, red);
  if (threadIdx.x == 0) {
    if (ta != 0.0) atomicAdd(&counts[0], (unsigned long long)ta);
    if (tl != 0.0) atomicAdd(&counts[1], (unsigned long long)tl);
    if (th != 0.0) atomicAdd(&counts[2], (unsigned long long)th);
  }
}

void launch_mask_and(const Geom& g, uint64_t* rlo_to_active, const uint64_t* rhi, unsigned long long* counts,
                     hipStream_t s) {
  const int blocks = cap_blocks(g.nwords, 256, 4096);
  hipLaunchKernelGGL(k_mask_and, dim3(blocks), dim3(256), 0, s, g, rlo_to_active, rhi, counts);
}

// --------------------------------------------------------------------------------------
// Matrix codes.  code = number of active 6-neighbours for a FREE voxel (active and not
// on a Dirichlet face), else 0: the diagonal of the reference's row
// (TortuosityHypreFill.F90:126-169, dx = 1); the off-diagonals are -1 exactly where the
// neighbour is active, which the apply gets for free because every vector is 0 there.
// --------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_codes(Geom g, const uint64_t* __restrict__ active,
                                                uint8_t* __restrict__ code, unsigned long long* __restrict__ n_free) {
  const long long nown = (long long)g.nzl * g.plane;
  unsigned long long nf = 0;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < nown;
       e += (long long)gridDim.x * blockDim.x) {
    const int i = (int)(e % g.pitch);
    const int j = (int)((e / g.pitch) % g.ny);
    const int gz = 1 + (int)(e / g.plane);
    const int kz = g.z0 + gz - 1;
    int c = 0;
    if (i < g.nx && bit_at(active, g, gz, j, i)) {
      const int cd = g.dir == 0 ? i : (g.dir == 1 ? j : kz);
      const int nd = g.dir == 0 ? g.nx : (g.dir == 1 ? g.ny : g.nz);
      if (cd != 0 && cd != nd - 1) {
        c += (i > 0 && bit_at(active, g, gz, j, i - 1));
        c += (i < g.nx - 1 && bit_at(active, g, gz, j, i + 1));
        c += (j > 0 && bit_at(active, g, gz, j - 1, i));
        c += (j < g.ny - 1 && bit_at(active, g, gz, j + 1, i));
        c += bit_at(active, g, gz - 1, j, i);  // ghost planes are 0 at a domain face
        c += bit_at(active, g, gz + 1, j, i);
      }
    }
    code[g.plane + e] = (uint8_t)c;
    nf += (c != 0);
  }
  __shared__ double red[4];
  const double t = block_sum((double)nf, red);
  if (threadIdx.x == 0 && t != 0.0) atomicAdd(n_free, (unsigned long long)t);
}

void launch_codes(const Geom& g, const uint64_t* active, uint8_t* code, unsigned long long* n_free,
                  hipStream_t s) {
  const int blocks = cap_blocks((long long)g.nzl * g.plane, 256, 8192);
  hipLaunchKernelGGL(k_codes, dim3(blocks), dim3(256), 0, s, g, active, code, n_free);
}

// Initial field: the reference's linear ramp on active cells (TortuosityHypreFill.F90:
// 233-262) with the exact Dirichlet values on the two faces (F90:191-228); ghost planes
// are filled from the ghost bits so no exchange is needed before the first apply.
// (one block per row: no 64-bit division per element, 16-byte stores)
__global__ void __launch_bounds__(256) k_init_field(Geom g, const uint64_t* __restrict__ active,
                                                     double* __restrict__ x) {
  const int nd = g.dir == 0 ? g.nx : (g.dir == 1 ? g.ny : g.nz);
  const double extent = (double)(nd - 1);
  const double factor = (extent < 1e-15) ? 0.0 : 1.0 / extent;
  const long long nrows = (long long)(g.nzl + 2) * g.ny;
  for (long long row = blockIdx.x; row < nrows; row += gridDim.x) {
    const int j = (int)(row % g.ny);
    const int gz = (int)(row / g.ny);
    const int kz = g.z0 + gz - 1;
    const uint64_t* bits = active + (long long)gz * g.wplane + (long long)j * g.W;
    double* xr = x + row * g.pitch;
    for (int i = 2 * (int)threadIdx.x; i < g.pitch; i += 2 * (int)blockDim.x) {  // pitch is even: whole pairs
      double v[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int ii = i + t;
        v[t] = 0.0;
        if (ii < g.nx && ((bits[ii >> 6] >> (ii & 63)) & 1ull)) {
          const int cd = g.dir == 0 ? ii : (g.dir == 1 ? j : kz);
          v[t] = g.vlo + (g.vhi - g.vlo) * (double)cd * factor;
          if (cd == 0) v[t] = g.vlo;
          else if (cd == nd - 1) v[t] = g.vhi;
        }
      }
      st2(xr + i, v[0], v[1]);
    }
  }
}

void launch_init_field(const Geom& g, const uint64_t* active, double* x, hipStream_t s) {
  const long long nrows = (long long)(g.nzl + 2) * g.ny;
  const int blocks = (int)std::min<long long>(nrows, 1 << 20);
  hipLaunchKernelGGL(k_init_field, dim3(blocks), dim3(256), 0, s, g, active, x);
}

__global__ void __launch_bounds__(256) k_unpack_mask(Geom g, const uint64_t* __restrict__ active,
                                                      uint8_t* __restrict__ out) {
  const long long n = (long long)g.nzl * g.ny * g.nx;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n;
       e += (long long)gridDim.x * blockDim.x) {
    const int i = (int)(e % g.nx);
    const int j = (int)((e / g.nx) % g.ny);
    const int gz = 1 + (int)(e / ((long long)g.nx * g.ny));
    out[e] = bit_at(active, g, gz, j, i) ? 1 : 0;
  }
}

void launch_unpack_mask(const Geom& g, const uint64_t* active, uint8_t* mask_out, hipStream_t s) {
  const int blocks = cap_blocks((long long)g.nzl * g.ny * g.nx, 256, 8192);
  hipLaunchKernelGGL(k_unpack_mask, dim3(blocks), dim3(256), 0, s, g, active, mask_out);
}

// r = -A x on free cells (x holds the Dirichlet values), p = z, partials {r.r, r.z}
// lineoff != null: r is stored in the compacted line layout (see k_cg1), only live pairs are written
__global__ void __launch_bounds__(256) k_residual0(Geom g, const uint8_t* __restrict__ code,
                                                    const double* __restrict__ x, double* __restrict__ r,
                                                    double* __restrict__ p, int jacobi, double* __restrict__ partials,
                                                    const int* __restrict__ lineoff, int LW) {
  double rr = 0.0, rz = 0.0;
  OI_PAIR_LOOP_BEGIN
  if (!cc) {  // a pair without a free voxel: zeros (not stored in the compacted layout), none of the seven loads, nothing for the sums
    if (!lineoff) st2(r + e, 0.0, 0.0);
    if (p) st2(p + e, 0.0, 0.0);
    continue;
  }
  const Pair q = apply_pair(g, x, e, i, j, c0, c1);
  const double r0 = c0 ? -q.a : 0.0, r1 = c1 ? -q.b : 0.0;
  const double z0 = (jacobi && c0) ? r0 / (double)c0 : r0;
  const double z1 = (jacobi && c1) ? r1 / (double)c1 : r1;
  if (lineoff) {
    if (cc) {
      const long long gz = e / g.plane;
      st2(r + (long long)kLV * lineoff[(gz * g.ny + j) * LW + (i >> kLVs)] + (i & (kLV - 1)), r0, r1);
    }
  } else {
    st2(r + e, r0, r1);
  }
  if (p) st2(p + e, z0, z1);  // the fused variant forms p_0 = z_0 itself
  rr += r0 * r0; rr += r1 * r1;
  rz += r0 * z0; rz += r1 * z1;
  OI_PAIR_LOOP_END
  __shared__ double red[4];
  const double a = block_sum(rr, red);
  const double b = block_sum(rz, red);
  if (threadIdx.x == 0) { partials[2 * blockIdx.x] = a; partials[2 * blockIdx.x + 1] = b; }
}

// The same residual when x IS the initial field of k_init_field: its values are a function of the position and the active bit, so the
// seven-point stencil is evaluated from the bit planes (1/8 byte per voxel) instead of from x (8 bytes per voxel through seven strided
// loads): 1.2 GB read instead of 9.7 at 1024^3.  Same values, same order of operations as k_residual0 on that field, hence the same
// bits (tests: ..._initial_residual_from_the_mask_bits_is_bit_identical).  Along the flow axis a pair's stencil meets four values of
// the ramp, across it one: they are formed once per pair, the neighbours contribute a bit each.  (Time: 5.4 ms against 5.7 -- the
// pair loop is bound by its chain of dependent round trips, code byte -> bits -> slot -> store, not by the bytes.)
__global__ void __launch_bounds__(256) k_residual0_ramp(Geom g, const uint8_t* __restrict__ code, const uint64_t* __restrict__ active,
                                                         double* __restrict__ r, double* __restrict__ p, int jacobi,
                                                         double* __restrict__ partials, const int* __restrict__ lineoff, int LW) {
  const int nd = g.dir == 0 ? g.nx : (g.dir == 1 ? g.ny : g.nz);
  const double extent = (double)(nd - 1);
  const double factor = (extent < 1e-15) ? 0.0 : 1.0 / extent;
  // what k_init_field stores on an active cell whose coordinate along the flow axis is cd: exactly its expressions
  auto ramp = [&](int cd) -> double {
    double v = g.vlo + (g.vhi - g.vlo) * (double)cd * factor;
    if (cd == 0) v = g.vlo;
    else if (cd == nd - 1) v = g.vhi;
    return v;
  };
  double rr = 0.0, rz = 0.0;
  OI_PAIR_LOOP_BEGIN
  if (!cc) {
    if (!lineoff) st2(r + e, 0.0, 0.0);
    if (p) st2(p + e, 0.0, 0.0);
    continue;
  }
  const int gz = (int)(e / g.plane);
  const uint64_t* rc = active + (long long)gz * g.wplane + (long long)j * g.W;
  // seven words, requested back to back with clamped (always valid) addresses -- a guarded load per neighbour would put seven
  // memory round trips in a row; what lies outside the row / the plane is masked afterwards
  const int w0 = i >> 6, wm = max(i - 1, 0) >> 6, wp = min(i + 2, g.nx - 1) >> 6;
  const bool jm = j > 0, jp = j < g.ny - 1;
  const uint64_t Wc = rc[w0], Wm = rc[wm], Wp = rc[wp];
  const uint64_t Wym = (jm ? rc - g.W : rc)[w0], Wyp = (jp ? rc + g.W : rc)[w0];
  const uint64_t Wzm = (rc - g.wplane)[w0], Wzp = (rc + g.wplane)[w0];
  const int b0 = i & 63;                 // i is even: voxel i + 1 is bit b0 + 1 of the same word
  const bool in1 = i + 1 < g.nx;
  auto at = [](uint64_t w, int b) -> bool { return (w >> b) & 1ull; };
  // the field's values this pair's stencil can meet: along the flow axis the coordinate moves by one, across it it does not
  const int cd0 = g.dir == 0 ? i : (g.dir == 1 ? j : g.z0 + gz - 1);
  const double v0 = ramp(cd0), v1 = g.dir == 0 ? ramp(cd0 + 1) : v0;            // voxels i and i + 1
  const double vm = ramp(cd0 - 1), vp = ramp(g.dir == 0 ? cd0 + 2 : cd0 + 1);    // one step down / up the axis (used where that cell is active)
  const double cx = at(Wc, b0) ? v0 : 0.0, cy = (in1 && at(Wc, b0 + 1)) ? v1 : 0.0;
  const double xm = (i > 0 && at(Wm, (i - 1) & 63)) ? (g.dir == 0 ? vm : v0) : 0.0;
  const double xp = (i + 2 < g.nx && at(Wp, (i + 2) & 63)) ? (g.dir == 0 ? vp : v1) : 0.0;
  const double ym0 = (jm && at(Wym, b0)) ? (g.dir == 1 ? vm : v0) : 0.0, ym1 = (jm && in1 && at(Wym, b0 + 1)) ? (g.dir == 1 ? vm : v1) : 0.0;
  const double yp0 = (jp && at(Wyp, b0)) ? (g.dir == 1 ? vp : v0) : 0.0, yp1 = (jp && in1 && at(Wyp, b0 + 1)) ? (g.dir == 1 ? vp : v1) : 0.0;
  const double zm0 = at(Wzm, b0) ? (g.dir == 2 ? vm : v0) : 0.0, zm1 = (in1 && at(Wzm, b0 + 1)) ? (g.dir == 2 ? vm : v1) : 0.0;
  const double zp0 = at(Wzp, b0) ? (g.dir == 2 ? vp : v0) : 0.0, zp1 = (in1 && at(Wzp, b0 + 1)) ? (g.dir == 2 ? vp : v1) : 0.0;
  // apply_pair on the field: ((((xm + xp) + ym) + yp) + zm) + zp per voxel of the pair
  const double s0 = ((((xm + cy) + ym0) + yp0) + zm0) + zp0;
  const double s1 = ((((cx + xp) + ym1) + yp1) + zm1) + zp1;
  const double qa = c0 ? (double)c0 * cx - s0 : 0.0, qb = c1 ? (double)c1 * cy - s1 : 0.0;
  const double r0 = c0 ? -qa : 0.0, r1 = c1 ? -qb : 0.0;
  const double z0 = (jacobi && c0) ? r0 / (double)c0 : r0;
  const double z1 = (jacobi && c1) ? r1 / (double)c1 : r1;
  if (lineoff) st2(r + (long long)kLV * lineoff[((long long)gz * g.ny + j) * LW + (i >> kLVs)] + (i & (kLV - 1)), r0, r1);
  else st2(r + e, r0, r1);
  if (p) st2(p + e, z0, z1);
  rr += r0 * r0; rr += r1 * r1;
  rz += r0 * z0; rz += r1 * z1;
  OI_PAIR_LOOP_END
  __shared__ double red[4];
  const double a = block_sum(rr, red);
  const double b = block_sum(rz, red);
  if (threadIdx.x == 0) { partials[2 * blockIdx.x] = a; partials[2 * blockIdx.x + 1] = b; }
}

// active != null: x is the initial field of launch_init_field over these bit planes (cg_begin) -- the residual comes from the bits
int launch_residual0(const Geom& g, const uint8_t* code, const double* x, double* r, double* p, int jacobi,
                     double* partials, const int* lineoff, hipStream_t s, const uint64_t* active) {
  const int blocks = cap_blocks((long long)g.nzl * g.ny * (g.pitch / 2), 256, 2048);
  if (active)
    hipLaunchKernelGGL(k_residual0_ramp, dim3(blocks), dim3(256), 0, s, g, code, active, r, p, jacobi, partials, lineoff, (g.pitch + kLV - 1) / kLV);
  else
    hipLaunchKernelGGL(k_residual0, dim3(blocks), dim3(256), 0, s, g, code, x, r, p, jacobi, partials, lineoff,
                       (g.pitch + kLV - 1) / kLV);
  return blocks;
}


// ---- compacted line layout: which aligned kLV-voxel lines (owned AND ghost planes) hold a free voxel ----
__global__ void __launch_bounds__(256) k_line_flags(Geom g, const uint8_t* __restrict__ code, int LW, int* __restrict__ flags) {
  const long long n = (long long)(g.nzl + 2) * g.ny * LW;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x) {
    const int li = (int)(t % LW);
    const long long row = t / LW;  // gz * ny + j
    const long long base = (row / g.ny) * g.plane + (row % g.ny) * g.pitch + (long long)kLV * li;
    const int len = min(kLV, g.pitch - kLV * li);
    int any = 0;
    for (int v = 0; v < len; ++v) any |= code[base + v] != 0;
    flags[t] = any;
  }
}
// slot (exclusive scan of the flags) or -1; slot of the first line of every plane into plane_first[0 .. nzl+2].
// seg (optional): one SEGMENT RECORD per row and 128-voxel tile column -- { slot of the segment's first stored line, bit
// mask of its stored lines } -- from which a wave derives the slots of its whole row segment with one SCALAR load
// (slot of line l = base + popcount(mask below l)) instead of a 4-byte table entry fetched by every lane.
__global__ void __launch_bounds__(256) k_line_finalize(long long n, long long per_plane, int LW, int TX, const int* __restrict__ flags,
                                                        int* __restrict__ slots, long long* __restrict__ plane_first,
                                                        uint2* __restrict__ seg) {
  constexpr int LPT = kBX_lines;  // lines per tile column
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x) {
    const int sl = slots[t];
    if (t % per_plane == 0) plane_first[t / per_plane] = sl;
    if (t == n - 1) plane_first[n / per_plane] = sl + flags[t];
    const int li = (int)(t % LW);
    if (seg && (li % LPT) == 0) {
      unsigned m = 0;
      for (int b = 0; b < LPT && li + b < LW; ++b) m |= flags[t + b] ? (1u << b) : 0u;
      seg[(t / LW) * TX + li / LPT] = make_uint2((unsigned)sl, m);
    }
  }
  // (second pass over the same elements: the records read the flags of up to LPT - 1 later lines, which the slot
  // rewrite does not touch -- flags stay as they are, only `slots` changes, and every thread rewrites its own entries)
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x)
    if (!flags[t]) slots[t] = -1;
}
// ---- exclusive prefix sum of the line flags (set-up, once per handle): three passes, hand-written ----
// blocks of 256 threads x 16 consecutive items: (1) block totals, (2) one block scans the totals in place, (3) every block
// scans its items again and adds its offset.  Integer sums: the order does not matter, the result is exact.
constexpr int kScanItems = 16, kScanBlock = 256 * kScanItems;
__device__ __forceinline__ int scan_block_exclusive(int v, int* total) {  // exclusive scan of one value per thread over 256 threads
  __shared__ int wsum[4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  __syncthreads();  // (wsum may still be read by a previous call)
  if (lane == 63) wsum[w] = inc;
  __syncthreads();
  int base = 0;
  for (int k = 0; k < w; ++k) base += wsum[k];
  if (total) *total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  return base + inc - v;
}
__global__ void __launch_bounds__(256) k_scan_totals(long long n, const int* __restrict__ flags, int* __restrict__ totals) {
  const long long first = (long long)blockIdx.x * kScanBlock + (long long)threadIdx.x * kScanItems;
  int sum = 0;
  for (int k = 0; k < kScanItems; ++k)
    if (first + k < n) sum += flags[first + k];
  int tot = 0;
  (void)scan_block_exclusive(sum, &tot);
  if (threadIdx.x == 0) totals[blockIdx.x] = tot;
}
__global__ void __launch_bounds__(256) k_scan_offsets(int nblocks, int* __restrict__ totals) {  // one block: totals -> their exclusive scan
  int carry = 0;
  for (int b0 = 0; b0 < nblocks; b0 += 256) {
    const int i = b0 + (int)threadIdx.x;
    const int v = i < nblocks ? totals[i] : 0;
    int tot = 0;
    const int ex = scan_block_exclusive(v, &tot);
    if (i < nblocks) totals[i] = carry + ex;
    carry += tot;
  }
}
__global__ void __launch_bounds__(256) k_scan_apply(long long n, const int* __restrict__ flags, const int* __restrict__ offsets,
                                                     int* __restrict__ out) {
  const long long first = (long long)blockIdx.x * kScanBlock + (long long)threadIdx.x * kScanItems;
  int f[kScanItems], sum = 0;
  for (int k = 0; k < kScanItems; ++k) { f[k] = first + k < n ? flags[first + k] : 0; sum += f[k]; }
  int run = offsets[blockIdx.x] + scan_block_exclusive(sum, nullptr);
  for (int k = 0; k < kScanItems; ++k) {
    if (first + k < n) out[first + k] = run;
    run += f[k];
  }
}

// lineoff: (nzl + 2) * ny * LW ints; plane_first_dev: nzl + 3 long longs.  Returns hipSuccess or the failing call's error.
hipError_t launch_build_lines(const Geom& g, const uint8_t* code, int* lineoff, long long* plane_first_dev, void* seg, hipStream_t s) {
  const int LW = (g.pitch + kLV - 1) / kLV;
  const int TX = (g.pitch + 127) / 128;
  const long long n = (long long)(g.nzl + 2) * g.ny * LW;
  if (n >= (1ll << 31)) return hipErrorInvalidValue;
  int* flags = nullptr;
  int* tmp = nullptr;
  hipError_t e = oi::dev_malloc((void**)&flags, sizeof(int) * n);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_line_flags, dim3(cap_blocks(n, 256, 8192)), dim3(256), 0, s, g, code, LW, flags);
  const int sblocks = (int)((n + kScanBlock - 1) / kScanBlock);
  e = oi::dev_malloc((void**)&tmp, sizeof(int) * (size_t)(sblocks > 0 ? sblocks : 1));
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_scan_totals, dim3(sblocks), dim3(256), 0, s, n, flags, tmp);
    hipLaunchKernelGGL(k_scan_offsets, dim3(1), dim3(256), 0, s, sblocks, tmp);
    hipLaunchKernelGGL(k_scan_apply, dim3(sblocks), dim3(256), 0, s, n, flags, tmp, lineoff);
    e = hipGetLastError();
  }
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_line_finalize, dim3(cap_blocks(n, 256, 8192)), dim3(256), 0, s, n, (long long)g.ny * LW, LW, TX, flags, lineoff,
                       plane_first_dev, (uint2*)seg);
    e = hipStreamSynchronize(s);
  }
  if (tmp) (void)oi::dev_free(tmp);
  (void)oi::dev_free(flags);
  return e;
}


// Boundary fluxes (reference global_fluxes, TortuosityHypre.cpp:1066-1104, dx = 1):
// one-sided differences on the lo / hi face where face cell and inner cell are active.
// --------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_flux(Geom g, const uint64_t* __restrict__ active,
                                               const double* __restrict__ x, double* __restrict__ partials) {
  // face cells are indexed by the two non-flow coordinates (a fastest)
  const int na = g.dir == 0 ? g.ny : g.nx;
  const int nb = g.dir == 2 ? g.ny : g.nzl;  // for dir 0/1 the second coordinate is the local plane
  const int nd = g.dir == 0 ? g.nx : (g.dir == 1 ? g.ny : g.nz);
  double fin = 0.0, fout = 0.0;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < (long long)na * nb;
       t += (long long)gridDim.x * blockDim.x) {
    const int a = (int)(t % na), b = (int)(t / na);
    for (int side = 0; side < 2; ++side) {
      int i, j, gz, ii, jj, gg;  // face cell and inner cell
      if (g.dir == 0) {
        j = jj = a; gz = gg = b + 1; i = side ? g.nx - 1 : 0; ii = side ? g.nx - 2 : 1;
      } else if (g.dir == 1) {
        i = ii = a; gz = gg = b + 1; j = side ? g.ny - 1 : 0; jj = side ? g.ny - 2 : 1;
      } else {
        i = ii = a; j = jj = b;
        const int kz = side ? g.nz - 1 : 0;
        gz = kz - g.z0 + 1; gg = side ? gz - 1 : gz + 1;
        if (gz < 1 || gz > g.nzl) continue;  // the face plane belongs to another slab
      }
      if (nd < 2) continue;
      if (!bit_at(active, g, gz, j, i) || !bit_at(active, g, gg, jj, ii)) continue;
      const double vb = x[(long long)gz * g.plane + (long long)j * g.pitch + i];
      const double vi = x[(long long)gg * g.plane + (long long)jj * g.pitch + ii];
      if (side == 0) fin += -((vi - vb) / 1.0);
      else fout += -((vb - vi) / 1.0);
    }
  }
  __shared__ double red[4];
  const double ti = block_sum(fin, red);
  const double to = block_sum(fout, red);
  if (threadIdx.x == 0) { partials[2 * blockIdx.x] = ti; partials[2 * blockIdx.x + 1] = to; }
}

void launch_flux(const Geom& g, const uint64_t* active, const double* x, double* partials, double* out2,
                 hipStream_t s) {
  const int na = g.dir == 0 ? g.ny : g.nx;
  const int nb = g.dir == 2 ? g.ny : g.nzl;
  const int blocks = cap_blocks((long long)na * nb, 256, 1024);
  hipLaunchKernelGGL(k_flux, dim3(blocks), dim3(256), 0, s, g, active, x, partials);
  launch_finalize(partials, blocks, 2, out2, nullptr, 0, 0, s);
}

// --------------------------------------------------------------------------------------
// Seeded synthetic microstructures (pure integer arithmetic, identical to the oracle).
// --------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_gen_bernoulli(uint8_t* __restrict__ phase, int nx, int ny, int z0,
                                                        int nzl, uint64_t seed, uint64_t thr53) {
  const long long n = (long long)nx * ny * nzl;
  const long long off = (long long)nx * ny * z0;
  for (long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x; v < n;
       v += (long long)gridDim.x * blockDim.x) {
    const uint64_t h = mix64(seed * 0xD1342543DE82EF95ULL + (uint64_t)(v + off));
    phase[v] = ((h >> 11) < thr53) ? 1 : 0;
  }
}

void launch_gen_bernoulli(uint8_t* phase, int nx, int ny, int nz, int z0, int nzl, uint64_t seed,
                          uint64_t thr53, hipStream_t s) {
  (void)nz;
  const int blocks = cap_blocks((long long)nx * ny * nzl, 256, 8192);
  hipLaunchKernelGGL(k_gen_bernoulli, dim3(blocks), dim3(256), 0, s, phase, nx, ny, z0, nzl, seed, thr53);
}

// Correlated "blob" structure (BASELINE.json config 5): integer trilinear interpolation of
// 16-bit lattice noise with spacing S; pore where the value is below `level`.
__host__ __device__ inline uint64_t blob_value(uint64_t seed, int S, int i, int j, int k) {
  const int a = i / S, b = j / S, c = k / S;
  const uint64_t w1[3] = {(uint64_t)(2 * (i % S) + 1), (uint64_t)(2 * (j % S) + 1), (uint64_t)(2 * (k % S) + 1)};
  uint64_t num = 0;
  for (int dc = 0; dc < 2; ++dc)
    for (int db = 0; db < 2; ++db)
      for (int da = 0; da < 2; ++da) {
        const uint64_t key = ((uint64_t)(uint32_t)(a + da)) | ((uint64_t)(uint32_t)(b + db) << 21) |
                             ((uint64_t)(uint32_t)(c + dc) << 42);
        uint64_t z = seed * 0xD1342543DE82EF95ULL + key;
        z += 0x9E3779B97F4A7C15ULL;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
        const uint64_t v = (z ^ (z >> 31)) >> 48;
        const uint64_t wx = da ? w1[0] : 2 * S - w1[0], wy = db ? w1[1] : 2 * S - w1[1],
                       wz = dc ? w1[2] : 2 * S - w1[2];
        num += wx * wy * wz * v;
      }
  return num;  // = value * (2S)^3
}

uint64_t blob_value_host(uint64_t seed, int S, int i, int j, int k) { return blob_value(seed, S, i, j, k); }

__global__ void __launch_bounds__(256) k_gen_blobs(uint8_t* __restrict__ phase, int nx, int ny, int z0, int nzl,
                                                    uint64_t seed, int S, uint64_t thr) {
  const long long n = (long long)nx * ny * nzl;
  for (long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x; v < n;
       v += (long long)gridDim.x * blockDim.x) {
    const int i = (int)(v % nx), j = (int)((v / nx) % ny), kz = z0 + (int)(v / ((long long)nx * ny));
    phase[v] = blob_value(seed, S, i, j, kz) < thr ? 1 : 0;
  }
}

void launch_gen_blobs(uint8_t* phase, int nx, int ny, int z0, int nzl, uint64_t seed, int S, uint64_t level,
                      hipStream_t s) {
  const uint64_t thr = level * (uint64_t)(2 * S) * (2 * S) * (2 * S);
  const int blocks = cap_blocks((long long)nx * ny * nzl, 256, 8192);
  hipLaunchKernelGGL(k_gen_blobs, dim3(blocks), dim3(256), 0, s, phase, nx, ny, z0, nzl, seed, S, thr);
}

// spheres binned on the host into cells of edge `cell` half-voxels (>= radius); a voxel
// tests the 27 cells around its own.
__global__ void __launch_bounds__(256) k_gen_spheres(uint8_t* __restrict__ phase, int nx, int ny, int z0, int nzl,
                                                      const int* __restrict__ centres,
                                                      const int* __restrict__ cell_start, int ncx, int ncy, int ncz,
                                                      int cell, int r2h) {
  const long long n = (long long)nx * ny * nzl;
  const long long rr = (long long)r2h * r2h;
  for (long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x; v < n;
       v += (long long)gridDim.x * blockDim.x) {
    const int i = (int)(v % nx), j = (int)((v / nx) % ny), kz = z0 + (int)(v / ((long long)nx * ny));
    const int px = 2 * i + 1, py = 2 * j + 1, pz = 2 * kz + 1;
    const int cx = px / cell, cy = py / cell, cz = pz / cell;
    uint8_t pore = 1;
    for (int dz = -1; dz <= 1 && pore; ++dz)
      for (int dy = -1; dy <= 1 && pore; ++dy)
        for (int dx = -1; dx <= 1 && pore; ++dx) {
          const int ax = cx + dx, ay = cy + dy, az = cz + dz;
          if (ax < 0 || ay < 0 || az < 0 || ax >= ncx || ay >= ncy || az >= ncz) continue;
          const int c = ax + ncx * (ay + ncy * az);
          for (int sidx = cell_start[c]; sidx < cell_start[c + 1]; ++sidx) {
            const long long ddx = px - centres[3 * sidx], ddy = py - centres[3 * sidx + 1],
                            ddz = pz - centres[3 * sidx + 2];
            if (ddx * ddx + ddy * ddy + ddz * ddz < rr) { pore = 0; break; }
          }
        }
    phase[v] = pore;
  }
}

void launch_gen_spheres(uint8_t* phase, int nx, int ny, int nz, int z0, int nzl, const int* centres_dev,
                        const int* cell_start_dev, int ncx, int ncy, int ncz, int cell, int r2h, hipStream_t s) {
  (void)nz;
  const int blocks = cap_blocks((long long)nx * ny * nzl, 256, 8192);
  hipLaunchKernelGGL(k_gen_spheres, dim3(blocks), dim3(256), 0, s, phase, nx, ny, z0, nzl, centres_dev,
                     cell_start_dev, ncx, ncy, ncz, cell, r2h);
}

// --------------------------------------------------------------------------------------
// dense (host order, row pitch nx, owned planes only) <-> pitched / ghosted device layout
// --------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_pitch_in(Geom g, const double* __restrict__ dense, double* __restrict__ vec) {
  const long long nown = (long long)g.nzl * g.plane;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < nown;
       e += (long long)gridDim.x * blockDim.x) {
    const int i = (int)(e % g.pitch);
    const long long row = e / g.pitch;
    vec[g.plane + e] = (i < g.nx) ? dense[row * g.nx + i] : 0.0;
  }
}
__global__ void __launch_bounds__(256) k_pitch_out(Geom g, const double* __restrict__ vec, double* __restrict__ dense) {
  const long long n = (long long)g.nzl * g.ny * g.nx;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n;
       e += (long long)gridDim.x * blockDim.x) {
    const int i = (int)(e % g.nx);
    const long long row = e / g.nx;
    dense[e] = vec[g.plane + row * g.pitch + i];
  }
}
__global__ void __launch_bounds__(256) k_codes_out(Geom g, const uint8_t* __restrict__ code, uint8_t* __restrict__ dense) {
  const long long n = (long long)g.nzl * g.ny * g.nx;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n;
       e += (long long)gridDim.x * blockDim.x) {
    const int i = (int)(e % g.nx);
    const long long row = e / g.nx;
    dense[e] = code[g.plane + row * g.pitch + i];
  }
}

void launch_pitch_copy_in(const Geom& g, const double* dense, double* vec, hipStream_t s) {
  hipLaunchKernelGGL(k_pitch_in, dim3(cap_blocks((long long)g.nzl * g.plane, 256, 8192)), dim3(256), 0, s, g, dense,
                     vec);
}
void launch_pitch_copy_out(const Geom& g, const double* vec, double* dense, hipStream_t s) {
  hipLaunchKernelGGL(k_pitch_out, dim3(cap_blocks((long long)g.nzl * g.ny * g.nx, 256, 8192)), dim3(256), 0, s, g,
                     vec, dense);
}
// tortuosity_remspot (Tortuosity_filcc.F90:88-177) as preconditionPhaseFab runs it (TortuosityHypre.cpp:264-289): every
// box of the BoxArray is filtered IN PLACE in its own fab, x fastest, tile by tile; neighbours in other boxes are read from
// ghost cells filled before the pass.  A voxel therefore depends on the NEW values of those of its -x/-y/-z neighbours
// that lie in the SAME box (any tile order visits them first) and on the OLD values of everything else: a triangular
// system.  Sweeping
//     cur[m] = f(old[m]; cur or old [m-1], [m-sy], [m-sz]; old[m+1], old[m+sy], old[m+sz])
// in any order converges to its unique solution (= the sequential result) in as many sweeps as the longest chain of
// flipped voxels, a handful in practice; `changed` counts the updates of a sweep.  cutx/cuty/cutz[c] != 0 marks the first
// cell of a box along that axis (global index); arrays carry one ghost plane below and above the nzl owned planes
// (global planes [z0, z0 + nzl)), so a z-slab relaxes with its neighbours: `old` ghosts travel once per pass, `cur`
// ghosts once per sweep.
__global__ void __launch_bounds__(256) k_remspot_sweep(const uint8_t* __restrict__ old, uint8_t* __restrict__ cur, int nx,
                                                        int ny, int nzl, int z0, int nz, const uint8_t* __restrict__ cutx,
                                                        const uint8_t* __restrict__ cuty, const uint8_t* __restrict__ cutz,
                                                        int* __restrict__ changed) {
  const long long sy = nx, sz = (long long)nx * ny, n = sz * nzl;
  int ch = 0;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x) {
    const int i = (int)(t % nx), j = (int)((t / nx) % ny), kl = (int)(t / sz), k = z0 + kl;
    const long long m = t + sz;  // (one ghost plane first)
    const uint8_t c = old[m];
    const bool connected = (i > 0 && (cutx[i] ? old[m - 1] : cur[m - 1]) == c) || (i < nx - 1 && old[m + 1] == c) ||
                           (j > 0 && (cuty[j] ? old[m - sy] : cur[m - sy]) == c) || (j < ny - 1 && old[m + sy] == c) ||
                           (k > 0 && (cutz[k] ? old[m - sz] : cur[m - sz]) == c) || (k < nz - 1 && old[m + sz] == c);
    const uint8_t v = connected ? c : (uint8_t)(c == 0 ? 1 : 0);
    if (cur[m] != v) { cur[m] = v; ch = 1; }
  }
  if (ch) atomicOr(changed, 1);
}
void launch_remspot_sweep(const uint8_t* old, uint8_t* cur, int nx, int ny, int nzl, int z0, int nz, const uint8_t* cutx,
                          const uint8_t* cuty, const uint8_t* cutz, int* changed, hipStream_t s) {
  hipLaunchKernelGGL(k_remspot_sweep, dim3(cap_blocks((long long)nx * ny * nzl, 256, 8192)), dim3(256), 0, s, old, cur, nx,
                     ny, nzl, z0, nz, cutx, cuty, cutz, changed);
}

// counts the aligned groups of `group` voxels (along x, within a row) of the owned planes that hold at
// least one free voxel: the cache lines of r/q/p/x the solver kernels cannot skip
__global__ void __launch_bounds__(256) k_count_live_groups(Geom g, const uint8_t* __restrict__ code, int group,
                                                            unsigned long long* __restrict__ out) {
  const int gpr = (g.nx + group - 1) / group;  // groups per row
  const long long n = (long long)g.nzl * g.ny * gpr;
  unsigned long long live = 0;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x) {
    const int gi = (int)(t % gpr);
    const long long row = t / gpr;  // (plane - 1) * ny + j
    const long long base = g.plane + (row / g.ny) * g.plane + (row % g.ny) * g.pitch + (long long)gi * group;
    const int len = min(group, g.nx - gi * group);
    bool any = false;
    for (int v = 0; v < len; ++v) any |= code[base + v] != 0;
    live += any ? 1ull : 0ull;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) live += __shfl_down(live, o, 64);
  if ((threadIdx.x & 63) == 0 && live) atomicAdd(out, live);
}
void launch_count_live_groups(const Geom& g, const uint8_t* code, int group, unsigned long long* out, hipStream_t s) {
  const int gpr = (g.nx + group - 1) / group;
  hipLaunchKernelGGL(k_count_live_groups, dim3(cap_blocks((long long)g.nzl * g.ny * gpr, 256, 8192)), dim3(256), 0, s, g,
                     code, group, out);
}

void launch_codes_out(const Geom& g, const uint8_t* code, uint8_t* dense, hipStream_t s) {
  hipLaunchKernelGGL(k_codes_out, dim3(cap_blocks((long long)g.nzl * g.ny * g.nx, 256, 8192)), dim3(256), 0, s, g,
                     code, dense);
}

}  // namespace oi
