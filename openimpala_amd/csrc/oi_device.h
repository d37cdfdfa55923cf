// oi_device.h -- device-side layout and kernel launchers shared by the host driver.
//
// HBM layout of one z-slab (global planes [z0, z0+nzl)) -- see DESIGN.md section 3:
//   * fp64 vectors x, r, q, p0, p1: (nzl+2) planes of ny rows of `pitch` doubles
//     (pitch = nx rounded up to even so every row is 16-byte aligned); plane 0 and
//     plane nzl+1 are ghost planes (zeros at a domain face, the neighbour slab's
//     boundary plane otherwise), so z-neighbours never need a bounds test.
//   * code bytes (same shape, 1 B/voxel): number of active 6-neighbours of a FREE
//     voxel, 0 for every other voxel -- the whole matrix.
//   * bit planes (open / reach_lo / reach_hi -> active): 64 voxels per uint64 word
//     along x, W = ceil(nx/64) words per row, same ghost-plane convention.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace oi {

// Launch shapes of the z-marching kernels.  A handle takes a snapshot of the process defaults when it is created
// (default_tune(), changed through the oi_tort_set_* entry points under a mutex) and every launcher reads the
// handle's own copy (Geom::tune): no tuning state is shared between live handles.
struct Tune {
  int tile_map = 2;      // block -> tile: 0 every XCD one contiguous range of ALL tiles; 1 identity; 2 every XCD a
                         // contiguous y band of EVERY z chunk, z chunks in launch order (the chip sweeps the volume in z)
  int zchunk = 0;        // planes per block of the z-marching kernels, 0 = automatic
  int fused_rows = 2;    // rows per thread of k_fused_a (tile height 8 x rows)
  int fused_kernel = 0;  // first kernel of the two-kernel iteration on compacted vectors: 0 = k_cg1b<FORM 1> (descriptor-addressed), 1 = k_fused_a
  int mg_norot = 0;      // 1: keep k_update_r_lines + the pre-smoother as two kernels (cross-check of the rotated iteration)
  int fused_zchunk = 32; // its planes per block on large volumes (both kernels and both layouts: same partial sums); 16 until the
                         // nontemporal policy of round 4: 32 takes 2.2 % off the multigrid-PCG iteration (profiles/r04/mg_forms_chunk_length.txt)
  int pure_rows = 4;     // the same for the LDS-tiled stand-alone apply
  int cg1_rows = 2, cg1_waves = 8, cg1_auto = 1;  // k_cg1 tile: rows per thread, waves per block; auto = by plane size
  int cg1_xskip = 1;     // k_cg1: x touched every second iteration
  int cg1_fullline = 1;  // compacted layout: stored lines read and written whole
  int cg1_compact = 1;   // single-kernel variant: r, q, p stored as compacted live lines
  int cg1_seg = 1;       // k_cg1b: line slots from per-row segment records (scalar loads) instead of the per-line table
  int cg1_nt = 1;        // k_cg1b (all forms): 1 = the streamed vectors take nontemporal accesses (which ones: kNtSt* / kNtLd* in
                         // oi_k_cg1b.hip, chosen by in-process A/B, profiles/r04/k_cg1b_cache_policies_*.txt); 0 = default policy
  int dense_nt = 31;     // LDS-tiled / flat kernels of the dense layouts (homogenization, two-kernel and unfused variants): nontemporal stores of
                         // p_out (1), x (2), q (4), r (8), the smoother's output (16).  All on: -2 % on the homogenization iteration at 1024^3
                         // (19.08 -> 18.7 ms, profiles/r04/homogenization_dense_kernels_nontemporal.txt; single bits within the noise); 0: A/B
  int cg1_kernel = 0;    // compacted layout: 0 = k_cg1b (descriptor-addressed, all loads of a plane in one burst), 1 = k_cg1
  int apply_kernel = 0;  // stand-alone apply: 0 = wave-autonomous k_apply7, 1 = LDS-tiled k_fused_a<PURE>
  int mg_plain = 0, mg_novec4 = 0, mg_debug = 0;  // multigrid cross-check switches (env OI_MG_PLAIN / _NOVEC4 / _DEBUG)
  long long mg_replicate_cells = 0;  // > 0: levels with at most this many cells are replicated on every slab (tests)
  int mg_wmask = -1;     // >= 0: levels visited twice per parent visit (cycle-shape experiments)
  float mg_scale = 0.f;  // > 0: scale of the coarse correction
  float mg_omega = 0.f;  // in (0, 1]: Jacobi damping of the cycle (default 0.9)
  float mg_omega_coarse = 0.f;  // in (0, 1]: damping on the coarse levels, if it is to differ from the fine level's (experiments)
  int cg1_fin_sliced = 1;     // 0: the final sums of the single-kernel iteration by one block (cross-check of k_finalize1_sliced)
  int mg_xfirst = 0;          // 1: x += alpha p stays in the first kernel of the rotated iteration (cross-check; default: in the post-smoother)
  int mg_blocked = 1;         // coarse levels: 4 x 2 x 2 cells per thread with the restriction in registers where a level allows it (round 4); 0: cross-check
  int mg_xpair = 1;           // x update inside the post-smoother: 1 = every second iteration, two terms (round 4); 0 = every iteration (cross-check)
  int mg_post_kernel = 0;     // 1: the fine post-smoother as k_mg_fine<2, 2> also where k_cg1b FORM 3 could run (cross-check)
  int mg_coarsest_plain = 0;  // 1: the coarsest level's sweeps as separate launches (cross-check of the one-launch kernel)
  int mg_fused_images = 1;    // periodic coarse levels that live whole on one rank: the cycle's kernels write the z images of what they store (no
                              // image-plane copy launch after them); 0: the copies run as well (cross-check)
  int mg_coarse_tiled = 1;    // coarse levels of >= 2^21 cells: correction + post-smoothing through an LDS tile (k_mg_post4t); 0: k_mg_post4 (cross-check); 2: every four-cell level (tests)
  int mg_nu_coarse = 0;  // damped-Jacobi sweeps before / after the coarse correction on the coarse levels (the fine level: 1): 1, 2, or
                         // 0 = by the structure -- 2 where fewer than 35 % of the 8-voxel lines hold a free voxel (low porosity: fewer
                         // iterations pay for the dearer cycle, profiles/r03/mg_nu_blobs768.txt, c5_share_multigrid_pcg.txt), else 1
  int place_trials = 48; // single-kernel CG: vector placements (six of place_slots allocations made back to back) timed when
                         // the handle is built, the fastest kept; 0 = six slots in order, no search (env OI_PLACE_TRIALS)
  int place_slots = 12;  // slots allocated for the search (6 are kept, the others freed after it), memory permitting
                         // (env OI_PLACE_SLOTS)
  float place_spacer_gb = 48.f;  // LARGEST spacer allocated between the two halves of the slots and freed again (memory
                         // permitting; 0 = none; env OI_PLACE_SPACER_GB): it fills the set up to place_extent_gb
  float place_extent_gb = 60.f;  // extent (x + slots + spacer) the set of allocations should span in HBM: a 1024^3 set
                         // spans it by itself, the set of a 256-plane slab gets a spacer (env OI_PLACE_EXTENT_GB)
  int place_min_mb = 256;  // vectors below this size (they sit in the 256 MB Infinity Cache) are not searched for
  float place_alloc_budget_s = 0.5f;  // spare slots (and the spacer) are allocated only while the set's allocations have
                         // taken less than this (the driver clears memory other processes have used: seconds for tens of GB)
  float place_retry_gbps = 5150.f;  // fp64 vectors: a kept placement whose kernel moves fewer ALGORITHMIC GB/s than this (one byte per voxel + 60 per
                         // voxel of a live line) is in the slow plateau some sets of allocations have wherever the vectors lie in them
                         // (one process in thirteen: 4980-5020 against 5240-5450 GB/s at 1024^3, profiles/r05/placement_slow_plateau.txt)
                         // -> a second set is allocated and searched, the better kept; 0 = never (env OI_PLACE_RETRY_GBPS)
  int place_cache = 1;   // reuse the placement an earlier handle of this process found for the same set of allocations
  int ring_lines_rccl = 0;  // homogenization on a ring of GPUs over RCCL: 1 = compacted line lists travel (oi_comm::exchange_ring_v), 0 = the
                         // dense vectors and whole ghost planes (exchange_ring).  0 until exchange_ring_v has run on more than one GPU: no
                         // multi-GPU node has been available to any round, and the variable-length ring exchange is the code with the
                         // posting-order argument and the skipped empty messages.  The in-process transport always compacts.
  int resid0_bits = 1;   // cg_begin: the initial residual from the active bits (k_residual0_ramp); 0: from x through the stencil (cross-check, same bits)
  int force_comm = 0;    // keep a one-rank communicator (single-GPU rehearsal of the RCCL path; env OI_FORCE_COMM)
  int max_blocks = 4096; // capacity (blocks) of the handle's per-block partial-sum array
};
Tune default_tune();                 // snapshot of the process defaults (+ the OI_* environment overrides)
void set_default_tune(const Tune&);  // new handles only
void tune_set_cg1_tile(Tune& t, int tile_rows);  // 8 / 16 / 32 rows (33: the 8-wave x 4-row form), <= 0 automatic

struct Geom {
  int nx, ny, nz;      // global voxel counts
  int z0, nzl;         // this slab: global planes [z0, z0+nzl)
  int dir;             // flow axis 0/1/2
  int pitch;           // doubles (and code bytes) per row
  int W;               // uint64 words per row of a bit plane
  long long plane;     // pitch * ny      elements per plane
  long long wplane;    // W * ny          words per bit plane
  long long nvec;      // (nzl+2) * plane elements per vector incl. ghost planes
  long long nwords;    // (nzl+2) * wplane
  double vlo, vhi;
  float live_lines = 0.f;  // compacted layout: stored lines / all lines of this slab (0: not known) -- launch shapes may depend on it
  int periodic;        // 1: x and y wrap by index in the tiled kernels (homogenization cell problem; z wraps through the ghost planes)
  Tune tune;
};

// slots of the device-resident CG history: hist[3*k + {0,1,2}] = {p.q, r.r, r.z} of iteration k
enum { H_PQ = 0, H_RR = 1, H_RZ = 2, H_STRIDE = 3 };

struct Ctrl {
  int done;        // 0 while iterating; else the number of iterations after which ||r|| <= tol
  int pad;
  double tol2;     // (eps * ||b||)^2
};

enum KernelId {
  K_PACK = 0, K_SEED, K_FLOOD, K_MASK, K_CODES, K_INIT, K_RESID0, K_APPLY, K_UPDATE_P, K_UPDATE_XR,
  K_FUSED_A, K_UPDATE_R, K_AXPY, K_FINALIZE, K_FLUX, K_GEN, K_CG1, K_MG, K_MG_POST /* the fine post-smoother alone: inside K_MG's span */, K_COUNT
};

// grid cap of the flat (non-tiled) reducing kernels; the z-marching kernels are bounded by Tune::max_blocks
constexpr int kMaxPartials = 4096;
// shortest z chunk the automatic choice makes, and the grid bound that follows for a slab: 128 x 8 voxel tile
// columns x chunks of kMinZChunk planes.  A handle sizes its partial-sum array for it (8 doubles per block;
// 1024^3: 131072 blocks = 8 MB), so no launch shape can write past the array.
constexpr int kMinZChunk = 8;
inline int partial_blocks_bound(int pitch, int ny, int nzl) {
  long long b = (long long)((pitch + 127) / 128) * ((ny + 7) / 8) * ((nzl + kMinZChunk - 1) / kMinZChunk);
  if (b < kMaxPartials) b = kMaxPartials;
  if (b > (1ll << 22)) b = 1ll << 22;
  return (int)b;
}
// compacted line layout: voxels per stored line (aligned along x).  8 fp64 = one 64-byte HBM burst; stored lines are
// contiguous, so finer lines do not mean partial cache-line accesses.  Measured on 1024^3 (profiles/r01/ab_linevox.txt):
// 16 -> 8 voxels stores 15 % fewer lines (spheres; 32 % at porosity 0.2) and runs 3-7 % faster; 4 gains nothing more.
// kLG = lanes of a wave (one voxel pair each) that share a stored line.  DENSE vectors are predicated on whole
// 128-byte cache lines (kDenseLG lanes): a 64-byte predicate there means partial-line stores and measures 3 % slower.
#ifndef OI_LINE_VOX
#define OI_LINE_VOX 8
#endif
constexpr int kLV = OI_LINE_VOX;
constexpr int kLVs = kLV == 16 ? 4 : kLV == 8 ? 3 : kLV == 4 ? 2 : 5;
constexpr int kLG = kLV / 2;
constexpr int kDenseLG = 8;
static_assert((1 << kLVs) == kLV, "line length: 4, 8, 16 or 32 voxels");

// ---- launchers (all asynchronous on `s`) ----
void launch_pack_open(const uint8_t* phase /*nzl planes, no ghosts, row pitch nx*/, const Geom& g,
                      int phase_id, uint64_t* open, unsigned long long* n_phase, hipStream_t s);
void launch_seed(const Geom& g, const uint64_t* open, uint64_t* rlo, uint64_t* rhi, hipStream_t s);
void launch_flood_sweep(const Geom& g, const uint64_t* open, uint64_t* rlo, uint64_t* rhi, int* changed,
                        hipStream_t s, int reverse = 0 /* 1: from the last plane down */);
void launch_flood_zwalk(const Geom& g, const uint64_t* open, uint64_t* rlo, uint64_t* rhi, int* changed, hipStream_t s);
void launch_mask_and(const Geom& g, uint64_t* rlo_to_active, const uint64_t* rhi,
                     unsigned long long* counts /*[0]=active,[1]=lo face,[2]=hi face*/, hipStream_t s);
void launch_codes(const Geom& g, const uint64_t* active, uint8_t* code, unsigned long long* n_free,
                  hipStream_t s);
void launch_init_field(const Geom& g, const uint64_t* active, double* x, hipStream_t s);
void launch_unpack_mask(const Geom& g, const uint64_t* active, uint8_t* mask_out /*nzl planes, pitch nx*/,
                        hipStream_t s);

// r = -A x on free cells, p = z (z = r or r/code), partials {rr, rz}
// (lineoff != null: r goes to the compacted line layout, see launch_build_lines)
// (active != null: x is the initial field launch_init_field makes of these bit planes -- the residual is formed from the bits, same bits)
int launch_residual0(const Geom& g, const uint8_t* code, const double* x, double* r, double* p, int jacobi,
                     double* partials, const int* lineoff, hipStream_t s, const uint64_t* active = nullptr);
// Compacted line layout of the single-kernel iteration: only the aligned kLV-voxel lines (owned and ghost
// planes) that hold a free voxel are stored, in (plane, row, line) order.  lineoff[(gz * ny + j) * LW + line],
// LW = ceil(pitch / 16), is a line's slot or -1; plane_first_dev[gz] (gz = 0 .. nzl + 2) the first slot of a plane.
// seg (optional): (nzl + 2) * ny * ceil(pitch / 128) segment records of 8 bytes (see k_line_finalize)
hipError_t launch_build_lines(const Geom& g, const uint8_t* code, int* lineoff, long long* plane_first_dev, void* seg, hipStream_t s);
// q = A v, partials {v.q}; returns number of partial blocks
int launch_apply(const Geom& g, const uint8_t* code, const double* v, double* q, double* partials,
                 const Ctrl* ctrl, hipStream_t s);
// p = z + beta p with beta = rz_k / rz_{k-1} read from hist
void launch_update_p(const Geom& g, const uint8_t* code, const double* r, double* p, int jacobi,
                     const double* hist, int k, const Ctrl* ctrl, hipStream_t s);
// x += alpha p; r -= alpha q; partials {rr, rz};  alpha = rz_k / pq_k
int launch_update_xr(const Geom& g, const uint8_t* code, double* x, double* r, const double* p, const double* q,
                     int jacobi, const double* hist, int k, double* partials, const Ctrl* ctrl, hipStream_t s);
// fused: x += alpha_{k-1} p_in ; p_out = z + beta_k p_in ; q = A p_out ; partials {p_out.q}
// p_in / p_out are double* or, with p_f32 != 0, float* (fp32 STORAGE of the search direction)
int launch_fused_a(const Geom& g, const uint8_t* code, double* x, const double* r, const void* p_in,
                   void* p_out, int p_f32, double* q, int jacobi, const double* hist, int k, double* partials,
                   const Ctrl* ctrl, const int* lineoff /* != null: r, p, q in the compacted line layout */,
                   const long long* plane_first /* + seg: the descriptor-addressed form (k_cg1b, FORM 1) may run */, const void* seg,
                   hipStream_t s, int x_elsewhere = 0 /* 1: x already took alpha_{k-1} p_{k-1} (launch_mg_post); descriptor form only */);
// stand-alone tiled stencil apply q = A v (17 B/voxel), partials {v.q}
int launch_apply_tiled(const Geom& g, const uint8_t* code, const double* v, double* q, double* partials,
                       const Ctrl* ctrl, hipStream_t s);
// the same for a CG vector (vanishing on non-free voxels): lines without a free voxel are skipped
int launch_apply_cgvec(const Geom& g, const uint8_t* code, const double* v, double* q, double* partials, const Ctrl* ctrl,
                       hipStream_t s);
// dst = src (ndoubles even): the box's reference 1 read : 1 write stream, see k_stream_copy
void launch_stream_copy(double* dst, const double* src, long long ndoubles, hipStream_t s);
void launch_stream_3r3w(double* d0, double* d1, double* d2, const double* s0, const double* s1, const double* s2,
                        long long ndoubles, hipStream_t s);
// q-free CG iteration of the capacity mode (x, r, p only): p.Ap without storing q, then
// x += alpha p, r -= alpha A p with A p recomputed on the fly; partials {p.q} / {r.r, r.z}
int launch_lean_dot(const Geom& g, const uint8_t* code, const double* p, double* partials, const Ctrl* ctrl,
                    hipStream_t s);
int launch_lean_update(const Geom& g, const uint8_t* code, const double* p, double* x, double* r, int jacobi,
                       const double* hist, int k, double* partials, const Ctrl* ctrl, hipStream_t s);
// coarse correction handed to the fine-level post-smoother of the multigrid cycle: fp32 arrays in
// the first coarse level's layout (cnx * cny cells per plane, planes csz apart, one ghost plane
// first; z0c = global index of the first owned coarse plane); the correction is xc + xc2
struct MgCoarse {
  const float* xc;
  const float* xc2;  // second W-cycle visit, may be null
  int cnx, cny, z0c;
  long long csz;
  int cnzl = 0, periodic = 0;  // owned coarse planes; 1 = periodic cell problem (plain fine kernels only)
};
// the two-kernel iteration on compacted vectors, descriptor-addressed forms (k_cg1b FORM 1 / 2): may they run on this slab?
bool fused_descr_ok(const Geom& g, const int* lineoff, const long long* plane_first, const void* seg, int jacobi);
// a periodic problem (Geom::periodic) may take those forms (their periodic instantiations) on a slab with an even first plane
bool periodic_tiles_ok(const Geom& g);
// residual update + fine pre-smoother + restriction in one pass (k_cg1b FORM 2); see oi_k_cg1b.hip
int launch_mg_pre_update(const Geom& g, const uint8_t* code, const double* r_in, double* r_out, const double* q, const double* hist, int k,
                         MgCoarse c, float* bcoarse, double omega, double* partials, const Ctrl* ctrl, const int* lineoff,
                         const long long* plane_first, const void* seg, hipStream_t s);
// the fine post-smoother on compacted vectors, descriptor-addressed (k_cg1b FORM 3): what launch_mg_fine mode 2 computes
// x != null: also x += alpha_{k-1} p_{k-1} (alpha from hist[k - 1]; p_prev in fp32 if p_f32), the update the first kernel of the
// iteration then leaves out (launch_fused_a, x_elsewhere)
int launch_mg_post(const Geom& g, const uint8_t* code, const double* b, MgCoarse c, double omega, double scale, double* z,
                   double* partials, const Ctrl* ctrl, const int* lineoff, const long long* plane_first, const void* seg, hipStream_t s,
                   double* x = nullptr, const void* p_prev = nullptr, int p_f32 = 0, const double* hist = nullptr, int k = 0,
                   const void* p_prev2 = nullptr /* != null: the x update is pairwise -- even k >= 2 only, alpha_{k-2} p_{k-2} then alpha_{k-1} p_{k-1} */);
// fine-level halves of the multigrid cycle on the LDS tile (see oi_k_dense.hip)
// mode 1 pre (out = residual), 2 post (out = z, partials z.b), 3 pre with the restriction fused in
// (bcoarse = fp32 right-hand side of the first coarse level, nothing else stored; needs an even g.z0)
int launch_mg_fine(int mode, const Geom& g, const uint8_t* code, const double* b, MgCoarse c, double omega, double scale,
                   double* out, float* bcoarse, double* partials, const Ctrl* ctrl, const int* lineoff /* b, out compacted */,
                   hipStream_t s);
// single-kernel CG iteration (see oi_k_cg1b.hip): rec = per-iteration scalar records of 8 doubles
struct Cg1Grid { int R, NW, tiles_x, tiles_y, zchunk, nzc, blocks; };
Cg1Grid cg1_grid(const Geom& g, bool descr, int split, int part);
// true in the TEST library (make xcheck, -DOI_XCHECK=1): the superseded kernels kept as bit-identity references are present --
// k_cg1 (the branching first form of the single-kernel iteration: dense vectors, option cg1_kernel = 1) and k_fused_a on compacted
// vectors (option fused_kernel = 1, other tile heights).  The product library refuses configurations that need them.
bool xcheck_built();
int launch_cg1(const Geom& g, const uint8_t* code, double* x, const double* r_in, double* r_out, const double* q_in,
               double* q_out, const void* p_in, void* p_out, int p_f32, int jacobi, double* rec, int k,
               double* partials, Ctrl* ctrl, int split, int part, int partial_base, const int* lineoff, const long long* plane_first,
               const void* seg, hipStream_t s);
constexpr int kFinSlices = 64, kFinScratch = 5 * kFinSlices + 2;
void launch_finalize1(const double* partials, int nblocks, double* rec_k, Ctrl* ctrl, int k, int with_scalars,
                      hipStream_t s,
                      double* scratch = nullptr /* kFinScratch doubles, zeroed once: large grids are summed by kFinSlices blocks */);
void launch_scalars1(double* rec_k, Ctrl* ctrl, int k, hipStream_t s);
void launch_axpy_ptr(const Geom& g, double* x, const void* p, int p_f32, const double* alpha_ptr, const int* lineoff,
                     hipStream_t s);
constexpr int kRec1Alpha = 5;
constexpr int kRec1Stride = 8;  // doubles per iteration record; [0] = r.r
// r -= alpha q; partials {rr, rz}
// (nslots >= 0: r, q are compacted line lists; the owned planes are slots [first_slot, first_slot + nslots))
int launch_update_r(const Geom& g, const uint8_t* code, double* r, const double* q, int jacobi,
                    const double* hist, int k, double* partials, const Ctrl* ctrl, long long first_slot, long long nslots,
                    hipStream_t s);
// x += alpha_k p  (deferred update flush of the fused variant)
// p_next != null: x = (x + alpha_k p) + alpha_{k+1} p_next in one pass over x (the same two dependent additions as two calls)
void launch_axpy_hist(const Geom& g, double* x, const void* p, int p_f32, const double* hist, int k, const int* lineoff,
                      hipStream_t s, const void* p_next = nullptr);
// dst[slot..slot+ncomp) = sum of partials (deterministic order); optional convergence test on
// dst[check_rr_index]: sets ctrl->done = iter_done when <= tol2
void launch_finalize(const double* partials, int nblocks, int ncomp, double* dst, Ctrl* ctrl, int check,
                     int iter_done, hipStream_t s);
void launch_check_done(const double* rr, Ctrl* ctrl, int iter_done, hipStream_t s);
// boundary fluxes: out[0] = flux_in sum, out[1] = flux_out sum (local part)
void launch_flux(const Geom& g, const uint64_t* active, const double* x, double* partials, double* out2,
                 hipStream_t s);
void launch_gen_bernoulli(uint8_t* phase, int nx, int ny, int nz, int z0, int nzl, uint64_t seed,
                          uint64_t thr53, hipStream_t s);
void launch_gen_spheres(uint8_t* phase, int nx, int ny, int nz, int z0, int nzl, const int* centres_dev,
                        const int* cell_start_dev, int ncx, int ncy, int ncz, int cell, int r2h, hipStream_t s);
void launch_gen_blobs(uint8_t* phase, int nx, int ny, int z0, int nzl, uint64_t seed, int S, uint64_t level,
                      hipStream_t s);
uint64_t blob_value_host(uint64_t seed, int S, int i, int j, int k);
void launch_sum_scalars(double* dst, const double* const* srcs, int nsrc, int n, hipStream_t s);
// dst[i] = srcs[0][i] + srcs[1][i] + ... in rank order (large-array sum of the in-process transport)
constexpr int kMaxSumRanks = 16;
void launch_sum_f32_ranks(float* dst, const float* const* srcs_host, int nsrc, size_t n, hipStream_t s);
// dense host-order (nx per row, nzl planes) <-> pitched/ghosted device vector
void launch_pitch_copy_in(const Geom& g, const double* dense, double* vec, hipStream_t s);
void launch_pitch_copy_out(const Geom& g, const double* vec, double* dense, hipStream_t s);
void launch_codes_out(const Geom& g, const uint8_t* code, uint8_t* dense, hipStream_t s);
// one relaxation sweep of the (sequentially defined) remspot pass, see oi_k_setup.hip
// (arrays with one ghost plane below and above the nzl owned planes; cut*[c] != 0: cell c starts a box along that axis)
void launch_remspot_sweep(const uint8_t* old, uint8_t* cur, int nx, int ny, int nzl, int z0, int nz, const uint8_t* cutx,
                          const uint8_t* cuty, const uint8_t* cutz, int* changed, hipStream_t s);
// *out += number of aligned x-groups of `group` voxels (owned planes) with at least one free voxel
void launch_count_live_groups(const Geom& g, const uint8_t* code, int group, unsigned long long* out, hipStream_t s);

}  // namespace oi
