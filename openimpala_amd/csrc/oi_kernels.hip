// oi_kernels.hip -- hand-written CDNA4 (gfx950) kernels of the flow-through tortuosity path.
//
// Everything here is HBM-bandwidth-bound integer/fp64 streaming work: 64-wide
// wavefronts read x-contiguous rows with 16-byte lanes, the stencil's x/y reuse goes
// through LDS tiles, z reuse through registers, dot products through wavefront
// shuffles + one LDS slot per wave + per-block partials summed in a fixed order (no
// float atomics: bit-reproducible).  No MFMA: 0.26 flop/B.
//
// Compiled with -ffp-contract=off: every product/sum is rounded exactly as in the
// CPU oracle, so single-kernel outputs are bit-identical to it.
#include "oi_device.h"
#include "oi_pool.h"


#include <algorithm>
#include <cstdlib>
#include <mutex>

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

// process defaults of the launch shapes; handles copy them at creation (see oi_device.h)
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
    if (const char* e = getenv("OI_PLACE_CACHE")) g_tune_defaults.place_cache = atoi(e) != 0;
  }
  return g_tune_defaults;
}
void set_default_tune(const Tune& t) {
  (void)default_tune();  // (environment first, so that an explicit call wins)
  std::lock_guard<std::mutex> lock(g_tune_mutex);
  g_tune_defaults = t;
}

static inline int cap_blocks(long long work_items, int per_block, int cap) {
  long long b = (work_items + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (int)b;
}

// --------------------------------------------------------------------------------------
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
__global__ void __launch_bounds__(256) k_init_field(Geom g, const uint64_t* __restrict__ active,
                                                     double* __restrict__ x) {
  const int nd = g.dir == 0 ? g.nx : (g.dir == 1 ? g.ny : g.nz);
  const double extent = (double)(nd - 1);
  const double factor = (extent < 1e-15) ? 0.0 : 1.0 / extent;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < g.nvec;
       e += (long long)gridDim.x * blockDim.x) {
    const int i = (int)(e % g.pitch);
    const int j = (int)((e / g.pitch) % g.ny);
    const int gz = (int)(e / g.plane);
    const int kz = g.z0 + gz - 1;
    double v = 0.0;
    if (i < g.nx && bit_at(active, g, gz, j, i)) {
      const int cd = g.dir == 0 ? i : (g.dir == 1 ? j : kz);
      v = g.vlo + (g.vhi - g.vlo) * (double)cd * factor;
      if (cd == 0) v = g.vlo;
      else if (cd == nd - 1) v = g.vhi;
    }
    x[e] = v;
  }
}

void launch_init_field(const Geom& g, const uint64_t* active, double* x, hipStream_t s) {
  const int blocks = cap_blocks(g.nvec, 256, 8192);
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

int launch_residual0(const Geom& g, const uint8_t* code, const double* x, double* r, double* p, int jacobi,
                     double* partials, const int* lineoff, hipStream_t s) {
  const int blocks = cap_blocks((long long)g.nzl * g.ny * (g.pitch / 2), 256, 2048);
  hipLaunchKernelGGL(k_residual0, dim3(blocks), dim3(256), 0, s, g, code, x, r, p, jacobi, partials, lineoff,
                     (g.pitch + kLV - 1) / kLV);
  return blocks;
}

constexpr int kBX_lines = 128 / kLV;  // stored lines per row of a 128-voxel tile column (segment records)
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
constexpr int kBX = 128;            // tile width in voxels = 64 lanes x 2
constexpr int kLP = kBX + 4;        // LDS row pitch in doubles
constexpr int kNW = 8;              // waves per block

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

// Defaults from the 1024^3 sweep in profiles/r01/tile_sweep.txt: 16-row tiles for the CG
// kernel (32 rows need 190 VGPRs -> one block per CU), 32-row tiles for the stand-alone apply.
// (tile shapes and chunk lengths are per-handle state: Geom::tune)

// Planes per block of the z-marching kernels.  Large volumes: >= ~4096 blocks, but chunks of >= 64 planes so the
// two redundant halo planes per chunk stay near 3 %.  Small volumes (the reference's 100^3 sample gives 7 tile
// columns) would leave most of the 256 CUs idle behind a few long marches: there the chunks shrink, down to 8
// planes, until ~1024 blocks exist -- the march is latency-bound, so halo planes are cheap next to idle CUs.
static int auto_zchunk(const Geom& g, long long cols) {
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
static void pick_chunks(const Geom& g, long long cols, bool even, int* zchunk_out, int* nzc_out) {
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
static bool grid_fits(const Geom& g, long long cols, int nzc) {
  return cols * nzc <= (g.tune.max_blocks > 0 ? g.tune.max_blocks : kMaxPartials);
}

template <int R, bool PURE, typename PT, int LEAN>
static void launch_tiled_r(int blocks, const Geom& g, const uint8_t* code, double* x, const double* r, const void* p_in,
                           void* p_out, double* q, int jacobi, const double* hist, int k, int zchunk, int tiles_x,
                           int tiles_y, int nzc, double* partials, const Ctrl* ctrl, const int* lineoff, hipStream_t s) {
  if (lineoff && !PURE)
    hipLaunchKernelGGL((k_fused_a<R, PURE, PT, LEAN, !PURE>), dim3(blocks), dim3(64 * kNW), 0, s, g, code, x, r, (const PT*)p_in,
                       (PT*)p_out, q, jacobi, hist, k, zchunk, tiles_x, tiles_y, nzc, partials, ctrl, lineoff,
                       (g.pitch + kLV - 1) / kLV);
  else
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

static void fused_chunks(const Geom& g, long long cols, bool even, int* zchunk, int* nzc);
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

template <int R, int NW, typename PT, bool JAC, bool COMPACT>
static void launch_cg1_k(int blocks, const Geom& g, const uint8_t* code, double* x, const double* r_in, double* r_out,
                         const double* q_in, double* q_out, const void* p_in, void* p_out, double* rec, int k,
                         int zchunk, int tiles_x, int tiles_y, int nzc, int part, int partial_base, double* partials,
                         Ctrl* ctrl, const int* lineoff, hipStream_t s) {
  hipLaunchKernelGGL((k_cg1<R, NW, PT, JAC, COMPACT>), dim3(blocks), dim3(64 * NW), 0, s, g, code, x, r_in, r_out, q_in,
                     q_out, (const PT*)p_in, (PT*)p_out, rec, k, g.tune.cg1_xskip, zchunk, tiles_x, tiles_y, nzc, part,
                     partial_base, partials, ctrl, lineoff, (g.pitch + kLV - 1) / kLV, g.tune.cg1_fullline);
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
#define OI_CG1K(JAC, CMP)                                                                                              \
  launch_cg1_k<R, NW, PT, JAC, CMP>(blocks, g, code, x, r_in, r_out, q_in, q_out, p_in, p_out, rec, k, zchunk, tiles_x, \
                                    tiles_y, nzc, part, partial_base, partials, ctrl, lineoff, s)
  if (jacobi) { if (lineoff) OI_CG1K(true, true); else OI_CG1K(true, false); }
  else { if (lineoff) OI_CG1K(false, true); else OI_CG1K(false, false); }
#undef OI_CG1K
}

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

// The first kernel of the two-kernel iteration.  Compacted vectors with their segment records at hand (the multigrid-
// preconditioned solve): the descriptor-addressed form, k_cg1b<.., FORM 1>; everything else: k_fused_a.  BOTH take the
// chunking of fused_chunks, so that the per-block partial sums -- and with them the iterates -- are the same bit for bit
// whichever kernel and layout runs (tests: ..._layouts_compacted_and_dense, ..._two_solver_paths_agree).
static void fused_chunks(const Geom& g, long long cols, bool even, int* zchunk, int* nzc) {
  Geom gg = g;
  // (a periodic problem on dense vectors keeps the chunks it always had; on compacted vectors -- live_lines is set -- it takes these)
  if (g.tune.zchunk <= 0 && g.tune.fused_zchunk > 0 && (!g.periodic || g.live_lines > 0.f) &&
      cols * ((g.nzl + g.tune.fused_zchunk - 1) / g.tune.fused_zchunk) >= 2048)
    gg.tune.zchunk = g.tune.fused_zchunk;
  pick_chunks(gg, cols, even, zchunk, nzc);
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
    if (p_f32)
      launch_tiled_r<2, false, float, 0>(blocks, g, code, x, r, p_in, p_out, q, jacobi, hist, k, zchunk, tiles_x, tiles_y, nzc, partials, ctrl, lineoff, s);
    else
      launch_tiled_r<2, false, double, 0>(blocks, g, code, x, r, p_in, p_out, q, jacobi, hist, k, zchunk, tiles_x, tiles_y, nzc, partials, ctrl, lineoff, s);
    return blocks;
  }
  if (p_f32) return launch_tiled<false, float>(g, code, x, r, p_in, p_out, q, jacobi, hist, k, partials, ctrl, s, lineoff);
  return launch_tiled<false, double>(g, code, x, r, p_in, p_out, q, jacobi, hist, k, partials, ctrl, s, lineoff);
}

// after the 5 sums of kernel k are in rec[k] (all-reduced in multi-rank runs): convergence
// test on the exact ||r_k||, alpha_k, and beta_{k+1} from the one-step extrapolation
__device__ __forceinline__ void cg1_scalars(double* rk, Ctrl* ctrl, int k) {
  const Cg1Scalars sc = cg1_scalars_of(rk, ctrl->tol2);
  if (!ctrl->done && sc.converged) { ctrl->done = k + 1; }
  rk[R1_ALPHA] = sc.alpha;
  rk[R1_BETA] = sc.beta;
}

// three waves per component (15 waves): lane L = 0 .. 191 of a component adds partials L, L + 192, ... with four
// independent running sums (the loads of a round are in flight together: the descriptor-addressed CG kernel launches
// 32768 blocks at 1024^3 and one wave per component took 0.2 ms over them), then a fixed tree -- the order of
// summation depends on the number of blocks only, never on timing
__global__ void __launch_bounds__(960) k_finalize1(const double* __restrict__ partials, int nblocks, double* rk,
                                                    Ctrl* ctrl, int k, int with_scalars) {
  if (ctrl->done) return;
  __shared__ double part[15];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int c = wv % 5, L = (wv / 5) * 64 + lane;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  int b = L;
  for (; b + 576 < nblocks; b += 768) {
    a0 += partials[(long long)b * 5 + c];
    a1 += partials[(long long)(b + 192) * 5 + c];
    a2 += partials[(long long)(b + 384) * 5 + c];
    a3 += partials[(long long)(b + 576) * 5 + c];
  }
  for (; b < nblocks; b += 192) a0 += partials[(long long)b * 5 + c];
  const double v = wave_sum((a0 + a1) + (a2 + a3));
  if (lane == 0) part[wv] = v;
  __syncthreads();
  if (threadIdx.x < 5) rk[threadIdx.x] = (part[threadIdx.x] + part[threadIdx.x + 5]) + part[threadIdx.x + 10];
  __syncthreads();
  if (threadIdx.x == 0 && with_scalars) cg1_scalars(rk, ctrl, k);
}

// The same sums by kFinSlices blocks (large grids): block j adds the records of slice j -- wave c component c, four running sums
// per lane -- and leaves five slice sums in `scratch`; the block that draws the last ticket adds the slices in slice order.  Which
// block that is depends on timing, what it computes does not: the order of summation is fixed by (nblocks, kFinSlices).
// scratch: 5 * kFinSlices doubles + the ticket (zero between launches: the last block resets it).
__global__ void __launch_bounds__(320) k_finalize1_sliced(const double* __restrict__ partials, int nblocks, double* scratch, double* rk,
                                                           Ctrl* ctrl, int k, int with_scalars) {
  if (ctrl->done) return;  // (set by the CG kernel before this launch, or by the last block below after every block has passed here)
  __shared__ int last;
  const int c = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int S = (nblocks + kFinSlices - 1) / kFinSlices;
  const int b0 = blockIdx.x * S, b1 = min(nblocks, b0 + S);
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  int b = b0 + lane;
  for (; b + 192 < b1; b += 256) {
    a0 += partials[(long long)b * 5 + c];
    a1 += partials[(long long)(b + 64) * 5 + c];
    a2 += partials[(long long)(b + 128) * 5 + c];
    a3 += partials[(long long)(b + 192) * 5 + c];
  }
  for (; b < b1; b += 64) a0 += partials[(long long)b * 5 + c];
  const double v = wave_sum((a0 + a1) + (a2 + a3));
  unsigned* ticket = reinterpret_cast<unsigned*>(scratch + 5 * kFinSlices);
  if (lane == 0) __hip_atomic_store(&scratch[blockIdx.x * 5 + c], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) last = atomicAdd(ticket, 1u) == (unsigned)kFinSlices - 1u;
  __syncthreads();
  if (!last) return;
  __threadfence();
  // (agent-scope loads: the slice sums come from other CUs)
  const double sl = lane < kFinSlices ? __hip_atomic_load(&scratch[lane * 5 + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
  static_assert(kFinSlices <= 64, "one slice per lane");
  const double t = wave_sum(sl);
  if (lane == 0) rk[c] = t;
  __syncthreads();
  if (threadIdx.x == 0) {
    *ticket = 0u;
    if (with_scalars) cg1_scalars(rk, ctrl, k);
  }
}

__global__ void k_scalars1(double* rk, Ctrl* ctrl, int k) {
  if (!ctrl->done) cg1_scalars(rk, ctrl, k);
}

void launch_finalize1(const double* partials, int nblocks, double* rec_k, Ctrl* ctrl, int k, int with_scalars,
                      hipStream_t s, double* scratch) {
  if (scratch && nblocks >= 4096) {
    hipLaunchKernelGGL(k_finalize1_sliced, dim3(kFinSlices), dim3(320), 0, s, partials, nblocks, scratch, rec_k, ctrl, k, with_scalars);
    return;
  }
  hipLaunchKernelGGL(k_finalize1, dim3(1), dim3(960), 0, s, partials, nblocks, rec_k, ctrl, k, with_scalars);
}
void launch_scalars1(double* rec_k, Ctrl* ctrl, int k, hipStream_t s) {
  hipLaunchKernelGGL(k_scalars1, dim3(1), dim3(1), 0, s, rec_k, ctrl, k);
}

// --------------------------------------------------------------------------------------
// Deterministic final sums: one block adds the per-block partials in a fixed order.
// --------------------------------------------------------------------------------------
// (1024 threads, four independent running sums per thread: the descriptor-addressed kernels of the two-kernel iteration launch
// 32768 blocks at 1024^3, and 256 threads with one running sum took 57 us over them, three times per multigrid-PCG iteration;
// the order of summation depends on the number of blocks only)
__global__ void __launch_bounds__(1024) k_finalize(const double* __restrict__ partials, int nblocks, int ncomp,
                                                    double* __restrict__ dst, Ctrl* ctrl, int check, int iter_done) {
  if (ctrl && ctrl->done) return;
  __shared__ double red[16];
  const int T = blockDim.x;
  for (int c = 0; c < ncomp; ++c) {
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int b = threadIdx.x;
    for (; b + 3 * T < nblocks; b += 4 * T) {
      a0 += partials[(long long)b * ncomp + c];
      a1 += partials[(long long)(b + T) * ncomp + c];
      a2 += partials[(long long)(b + 2 * T) * ncomp + c];
      a3 += partials[(long long)(b + 3 * T) * ncomp + c];
    }
    for (; b < nblocks; b += T) a0 += partials[(long long)b * ncomp + c];
    const double t = block_sum((a0 + a1) + (a2 + a3), red);
    if (threadIdx.x == 0) {
      dst[c] = t;
      if (check && c == 0 && t <= ctrl->tol2) ctrl->done = iter_done;
    }
  }
}

void launch_finalize(const double* partials, int nblocks, int ncomp, double* dst, Ctrl* ctrl, int check,
                     int iter_done, hipStream_t s) {
  // (small grids: a block of 256 has the partials in flight just as well)
  hipLaunchKernelGGL(k_finalize, dim3(1), dim3(nblocks > 4096 ? 1024 : 256), 0, s, partials, nblocks, ncomp, dst, ctrl, check, iter_done);
}

__global__ void k_check_done(const double* __restrict__ rr, Ctrl* ctrl, int iter_done) {
  if (!ctrl->done && *rr <= ctrl->tol2) ctrl->done = iter_done;
}

void launch_check_done(const double* rr, Ctrl* ctrl, int iter_done, hipStream_t s) {
  hipLaunchKernelGGL(k_check_done, dim3(1), dim3(1), 0, s, rr, ctrl, iter_done);
}

// dst[i] = sum over sources of srcs[s][i]  (in-process "all-reduce" of the local transport)
__global__ void k_sum_scalars(double* dst, const double* const* srcs, int nsrc, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    double v = 0.0;
    for (int s = 0; s < nsrc; ++s) v += srcs[s][i];
    dst[i] = v;
  }
}

void launch_sum_scalars(double* dst, const double* const* srcs, int nsrc, int n, hipStream_t s) {
  hipLaunchKernelGGL(k_sum_scalars, dim3((n + 63) / 64), dim3(64), 0, s, dst, srcs, nsrc, n);
}

struct SumSrcs { const float* p[kMaxSumRanks]; };
__global__ void __launch_bounds__(256) k_sum_f32_ranks(float* __restrict__ dst, SumSrcs srcs, int nsrc, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float v = srcs.p[0][i];
    for (int s = 1; s < nsrc; ++s) v += srcs.p[s][i];
    dst[i] = v;
  }
}

void launch_sum_f32_ranks(float* dst, const float* const* srcs_host, int nsrc, size_t n, hipStream_t s) {
  SumSrcs a = {};
  for (int r = 0; r < nsrc && r < kMaxSumRanks; ++r) a.p[r] = srcs_host[r];
  const size_t blocks = std::min<size_t>((n + 255) / 256, 4096);
  hipLaunchKernelGGL(k_sum_f32_ranks, dim3((unsigned)std::max<size_t>(blocks, 1)), dim3(256), 0, s, dst, a, nsrc, n);
}

// --------------------------------------------------------------------------------------
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
