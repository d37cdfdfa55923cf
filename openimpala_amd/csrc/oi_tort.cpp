// oi_tort.cpp -- host driver behind the C ABI (include/openimpala_amd.h): the MI355X
// replacement for OpenImpala::TortuosityHypre (reference src/props/TortuosityHypre.cpp).
//
//   create : phase bytes -> open bits -> flood fill from both faces -> active mask ->
//            code bytes (the matrix) ; TortuosityHypre ctor :100-191
//   value  : CG on the Dirichlet-eliminated system entirely on the device (scalars stay
//            in HBM, the host only polls a convergence flag) -> boundary fluxes -> tau ;
//            value() :761-891, solve() :654-756, global_fluxes() :1000-1134
//
// There is NO CPU fallback: without a usable HIP device every entry point fails with
// OI_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <memory>
#include <mutex>
#include <tuple>
#include <vector>

#include "oi_comm.h"
#include "oi_device.h"
#include "oi_mg.h"
#include "oi_pool.h"

namespace oi {
const char* last_error();
}
using namespace oi;

#define HIPC(call)                                                                       \
  do {                                                                                   \
    hipError_t e_ = (call);                                                              \
    if (e_ != hipSuccess) {                                                              \
      oi::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
      (void)hipGetLastError(); /* reported: do not leave it for the next launch check to find */ \
      return OI_HIP_ERROR;                                                               \
    }                                                                                    \
  } while (0)
#define OIC(call)                \
  do {                           \
    oi_status s_ = (call);       \
    if (s_ != OI_OK) return s_;  \
  } while (0)

static double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static const char* kKernelNames[K_COUNT] = {
    "pack_open", "seed", "flood_sweep", "mask_and", "codes", "init_field", "residual0", "apply",
    "update_p", "update_xr", "fused_a", "update_r", "axpy", "finalize", "flux", "generate", "cg1", "mg_vcycle", "mg_post"};

struct ProfRec {
  int kid;
  hipEvent_t a, b;
};

static constexpr int kMaxSlots = 16;  // array bound (Tune::place_slots)
// x and the vector slots of the single-kernel iteration: separate allocations made back to back (no search: one block)
struct VecSet {
  char* x = nullptr;
  char* slot[kMaxSlots] = {nullptr};
  char* block = nullptr;  // no search: the six slots are one allocation
  int nslots = 0, nfront = 0;  // nfront slots lie in front of the (freed) spacer
  size_t bytes = 0;            // device memory held
};

struct oi_tort {
  Geom g{};
  oi_tort_opts opts{};
  oi_comm* comm = nullptr;  // not owned
  int phase_id = 0;
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t comm_stream = nullptr;         // ghost-plane exchange of the single-kernel variant
  int comm_stream_priority = 0;
  hipStream_t aux_stream = nullptr;          // interior chunks of a slab with neighbours, beside the boundary chunks
  hipEvent_t ev_boundary = nullptr, ev_exchanged = nullptr, ev_start = nullptr, ev_interior = nullptr;

  // device memory
  uint64_t *open = nullptr, *active = nullptr /* reach_lo, then the mask */, *rhi = nullptr;
  uint8_t* code = nullptr;
  double *x = nullptr, *r = nullptr, *q = nullptr;
  double *r2 = nullptr, *q2 = nullptr;  // second buffers of the single-kernel variant
  double* rec = nullptr;                // its per-iteration scalar records
  // compacted line layout of r, q, p in the single-kernel variant (see launch_build_lines)
  int* lineoff = nullptr;
  unsigned long long* seg = nullptr;  // one record per row and 128-voxel tile column: first slot + mask of the stored lines
  long long* plane_first_dev = nullptr;
  std::vector<long long> plane_first;   // first slot of every plane, gz = 0 .. nzl + 2
  long long nvec_cg = 0;                // elements of r / q / p (= g.nvec in the dense layout)
  // single-kernel variant: x and r, q, p[0], r2, q2, p[1] are allocations of one SET made back to back; which six slots
  // of the set hold the vectors is chosen by timing (place_vectors) -- the HBM address mapping makes the iteration time
  // depend on where the three vectors read and the three written lie relative to each other -- and the rest is freed
  VecSet vs;
  size_t x_bytes = 0;  // > 0: x belongs to the set
  size_t slot_bytes = 0;
  size_t place_spacer_bytes = 0;
  double place_alloc_s = 0.0, place_search_s = 0.0, place_peak_bytes = 0.0;
  int place_cache_hit = 0;
  int placement[7] = {0, 1, 2, 3, 4, 5, 0};
  int placement_tried = 0;
  double placement_ms_first = 0.0, placement_ms_best = 0.0;
  double* z = nullptr;                  // preconditioned residual of the multigrid variant
  bool lean = false;  // OI_KERNEL_LEAN: no q vector, A p recomputed in the update pass
  oi::MgHierarchy* mg = nullptr;
  int mg_tiled = 1;  // 0 (env OI_MG_PLAIN): plain fine-level kernels, kept as a cross-check
  void* p[2] = {nullptr, nullptr};  // search direction: fp64, or fp32 storage when p_f32
  bool p_f32 = false;
  size_t p_elem = sizeof(double);
  double* partials = nullptr;
  double* fin_scratch = nullptr;  // kFinScratch doubles: slice sums + ticket of the sliced final reduction (launch_finalize1)
  double* hist = nullptr;
  Ctrl* ctrl = nullptr;
  double* scal = nullptr;                // 8 doubles of scratch (flux sums)
  unsigned long long* counters = nullptr;  // 8 u64
  int* changed = nullptr;
  // pinned host mirrors
  Ctrl* h_ctrl = nullptr;
  unsigned long long* h_counters = nullptr;
  int* h_changed = nullptr;
  double* h_scal = nullptr;
  size_t bytes_device = 0;

  // state
  bool rot = false;    // multigrid two-kernel iteration, rotated: kernel A of iteration k forms r_k = r_{k-1} - alpha q, pre-smooths and
                       // restricts it in one pass (launch_mg_pre_update; r double-buffered in r / r2), so iteration k is complete
                       // when launch k + 1 has run -- the bookkeeping of the single-kernel variant
  bool fused = true;   // fused two-kernel iteration or single-kernel iteration (double-buffered p)
  bool cg1 = false;    // single-kernel iteration
  int xskip = 0;       // cg1: x touched every second iteration (frozen at cg_begin)
  int xpair = 0;       // rotated multigrid iteration with the x update in the post-smoother: that update is pairwise (frozen at cg_begin)
  bool jacobi = false;
  bool percolates = false;
  bool vectors_ready = false;
  bool cg_active = false;
  bool cg_ever = false;    // a CG run has been begun at least once (the device-side history is initialised)
  bool solved = false;
  int maxiter = 0;
  int iters_issued = 0;
  int iters_done = 0;
  int hist_cap = 0;
  double bnorm = 0.0;
  long long n_lo = 0, n_hi = 0;
  oi_tort_stats st{};
  oi_status value_status = OI_OK;

  // profiling
  std::vector<ProfRec> prof_pending;
  std::vector<hipEvent_t> ev_pool;
  double k_ms[K_COUNT] = {0};
  long long k_n[K_COUNT] = {0};
};

// ---- profiling helpers -----------------------------------------------------------------
static hipEvent_t prof_event(oi_tort* h) {
  if (!h->ev_pool.empty()) {
    hipEvent_t e = h->ev_pool.back();
    h->ev_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  (void)hipEventCreate(&e);
  return e;
}
static void prof_harvest(oi_tort* h) {
  if (h->prof_pending.empty()) return;
  (void)hipStreamSynchronize(h->stream);
  for (auto& p : h->prof_pending) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
      h->k_ms[p.kid] += ms;
      h->k_n[p.kid] += 1;
    }
    h->ev_pool.push_back(p.a);
    h->ev_pool.push_back(p.b);
  }
  h->prof_pending.clear();
}
struct ProfScope {
  oi_tort* h;
  ProfRec rec;
  bool on;
  ProfScope(oi_tort* h_, int kid) : h(h_), on(h_->opts.profile != 0) {
    if (on) {
      rec.kid = kid;
      rec.a = prof_event(h);
      rec.b = prof_event(h);
      (void)hipEventRecord(rec.a, h->stream);
    }
  }
  ~ProfScope() {
    if (on) {
      (void)hipEventRecord(rec.b, h->stream);
      h->prof_pending.push_back(rec);
      if (h->prof_pending.size() > 8192) prof_harvest(h);
    }
  }
};
#define PROF(kid) ProfScope prof_scope_(h, kid)

// ---- small helpers ---------------------------------------------------------------------
template <typename T>
static oi_status dmalloc(oi_tort* h, T** p, size_t count) {
  HIPC(oi::dev_malloc((void**)p, count * sizeof(T)));  // (hipMalloc, or a block of the process's pool: oi_pool.h)
  h->bytes_device += count * sizeof(T);
  return OI_OK;
}

// scratch of one call: released on every way out
template <typename T>
struct DevTmp {
  T* p = nullptr;
  DevTmp() = default;
  DevTmp(const DevTmp&) = delete;
  DevTmp& operator=(const DevTmp&) = delete;
  ~DevTmp() { if (p) (void)oi::dev_free(p); }
};

static oi_status ensure_device(int device, int* out_device) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
    oi::set_error("no HIP device visible: this library has no CPU fallback");
    return OI_NO_DEVICE;
  }
  if (device >= 0) HIPC(hipSetDevice(device));
  HIPC(hipGetDevice(out_device));
  return OI_OK;
}

static void free_setup_buffers(oi_tort* h) {
  if (h->open) { (void)oi::dev_free(h->open); h->bytes_device -= sizeof(uint64_t) * h->g.nwords; h->open = nullptr; }
  if (h->rhi) { (void)oi::dev_free(h->rhi); h->bytes_device -= sizeof(uint64_t) * h->g.nwords; h->rhi = nullptr; }
}

static void vecset_free(VecSet& v);
extern "C" void oi_tort_destroy(oi_tort* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  for (auto& p : h->prof_pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
  for (auto e : h->ev_pool) (void)hipEventDestroy(e);
  if (h->vs.nslots) {
    if (h->vs.x) h->x = nullptr;
    h->r = h->q = h->r2 = h->q2 = nullptr; h->p[0] = h->p[1] = nullptr;
    vecset_free(h->vs);
  }
  void* dev[] = {h->open, h->active, h->rhi, h->code, h->x, h->r, h->q, h->p[0], h->p[1], h->r2, h->q2, h->rec, h->lineoff, h->seg, h->plane_first_dev,
                 h->partials, h->fin_scratch, h->hist, h->ctrl, h->scal, h->counters, h->changed};
  for (void* d : dev)
    if (d) (void)oi::dev_free(d);  // (arrays of dmalloc; the vectors of a placement set were handed back above, to the driver)
  void* pin[] = {h->h_ctrl, h->h_counters, h->h_changed, h->h_scal};
  for (void* d : pin)
    if (d) (void)hipHostFree(d);
  if (h->comm_stream) { (void)hipStreamSynchronize(h->comm_stream); (void)hipStreamDestroy(h->comm_stream); }
  if (h->aux_stream) { (void)hipStreamSynchronize(h->aux_stream); (void)hipStreamDestroy(h->aux_stream); }
  if (h->ev_boundary) (void)hipEventDestroy(h->ev_boundary);
  if (h->ev_exchanged) (void)hipEventDestroy(h->ev_exchanged);
  if (h->ev_start) (void)hipEventDestroy(h->ev_start);
  if (h->ev_interior) (void)hipEventDestroy(h->ev_interior);
  if (h->mg) oi::mg_destroy(h->mg);
  if (h->z) (void)oi::dev_free(h->z);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

extern "C" void oi_tort_default_opts(oi_tort_opts* o) {
  if (!o) return;
  o->eps = 1e-9;  // reference default, TortuosityHypre.cpp:142
  o->maxiter = 0;
  o->precond = OI_PRECOND_DEFAULT;  // the reference's solve is multigrid-preconditioned too
  o->remspot_passes = 0;
  o->remspot_box_size = 32;  // the reference's default box_size (Diffusion.cpp:129, ba.maxSize)
  o->verbose = 0;
  o->check_every = 0;
  o->kernel = OI_KERNEL_AUTO;
  o->profile = 0;
  o->device = -1;
  o->p_storage = OI_STORE_F64;
}

// exchange one ghost plane pair of an array with `elem` bytes per voxel-slot
static oi_status exchange_planes(oi_tort* h, void* base, size_t plane_bytes) {
  if (!h->comm) return OI_OK;
  char* b = (char*)base;
  const Geom& g = h->g;
  return h->comm->exchange(b + plane_bytes * 1, b + plane_bytes * 0, b + plane_bytes * (size_t)g.nzl,
                           b + plane_bytes * (size_t)(g.nzl + 1), plane_bytes, h->stream);
}

// ghost planes of a CG vector of the single-kernel variant (dense or compacted), on stream `s`
static oi_status exchange_cg_vec(oi_tort* h, void* base, size_t elem, hipStream_t s) {
  if (!h->comm) return OI_OK;
  char* b = (char*)base;
  const Geom& g = h->g;
  if (!h->lineoff) {
    const size_t pb = elem * (size_t)g.plane;
    return h->comm->exchange(b + pb, b, b + pb * (size_t)g.nzl, b + pb * (size_t)(g.nzl + 1), pb, s);
  }
  const std::vector<long long>& pf = h->plane_first;
  const size_t le = (size_t)oi::kLV * elem;  // bytes per stored line
  return h->comm->exchange_v(b + le * pf[1], le * (pf[2] - pf[1]), b + le * pf[0], le * (pf[1] - pf[0]),
                             b + le * pf[g.nzl], le * (pf[g.nzl + 1] - pf[g.nzl]), b + le * pf[g.nzl + 1],
                             le * (pf[g.nzl + 2] - pf[g.nzl + 1]), s);
}

// Compacted planes travel as line lists whose length every rank derives from its OWN copy of the code bytes (ghost planes
// included).  exchange_v skips zero-length messages on each side independently, so neighbours that disagreed on a length
// would dead-lock the send/recv group: prove the agreement once, at set-up.  Every rank sends each neighbour {lines it
// will send, lines it expects} and compares; the verdict is max-reduced so that all ranks fail together.
static oi_status check_plane_message_lengths(oi_tort* h, hipStream_t s) {
  if (!h->comm) return OI_OK;
  const Geom& g = h->g;
  const std::vector<long long>& pf = h->plane_first;
  unsigned long long* d = h->counters;  // 8 device words: [0,1] to below, [2,3] from below, [4,5] to above, [6,7] from above
  unsigned long long host[8] = {(unsigned long long)(pf[2] - pf[1]), (unsigned long long)(pf[1] - pf[0]), 0, 0,
                                (unsigned long long)(pf[g.nzl + 1] - pf[g.nzl]), (unsigned long long)(pf[g.nzl + 2] - pf[g.nzl + 1]), 0, 0};
  HIPC(hipMemcpyAsync(d, host, sizeof host, hipMemcpyHostToDevice, s));
  OIC(h->comm->exchange(d + 0, d + 2, d + 4, d + 6, 2 * sizeof(unsigned long long), s));
  unsigned long long back[8];
  HIPC(hipMemcpyAsync(back, d, sizeof back, hipMemcpyDeviceToHost, s));
  HIPC(hipStreamSynchronize(s));
  HIPC(hipMemsetAsync(d, 0, sizeof host, s));  // (the counters are reused by the set-up passes that follow)
  int bad = 0;
  if (h->comm->rank > 0 && (back[2] != host[1] || back[3] != host[0])) bad = 1;                   // below: its send = my expect, ...
  if (h->comm->rank < h->comm->size - 1 && (back[6] != host[5] || back[7] != host[4])) bad = 1;  // above
  OIC(h->comm->allreduce_max_i32_host(&bad, 1, s));
  if (bad) {
    oi::set_error("slab %d of %d: neighbours disagree on the stored-line count of a ghost plane (send %llu / expect %llu below, "
                  "send %llu / expect %llu above; they report %llu / %llu and %llu / %llu)", h->comm->rank, h->comm->size, host[0], host[1],
                  host[4], host[5], back[2], back[3], back[6], back[7]);
    return OI_BAD_ARG;
  }
  return OI_OK;
}

extern "C" int oi_tort_line_voxels(void) { return oi::kLV; }
// Process DEFAULTS of the launch shapes and layouts: a handle copies them when it is created (oi::Tune in its
// Geom) and is not touched by later calls; oi_tort_retune re-reads them for one handle.
extern "C" void oi_tort_set_cg1_layout(int compact_lines) {
  // bit 0: compacted lines (single-kernel iteration and the multigrid-preconditioned two-kernel iteration);
  // bit 1 set: pair-granular (partial-line) accesses inside stored lines (A/B switch)
  oi::Tune t = oi::default_tune();
  t.cg1_compact = (compact_lines & 1) ? 1 : 0;
  t.cg1_fullline = (compact_lines & 2) ? 0 : 1;
  oi::set_default_tune(t);
}

// ---- construction ----------------------------------------------------------------------
static oi_status cg_one_iteration(oi_tort* h, int k);

// ---- placement of the vectors of the single-kernel iteration --------------------------------------------------------------
// MI355X: the time of a kernel that streams several arrays at the same offsets depends on the DISTANCES between the
// arrays it reads and the arrays it writes (HBM address mapping: tools/hbm_window.hip, profiles/r02/hbm_window_*.txt --
// a plain copy runs at 4.6 to 6.0 TB/s depending on which address bit separates source and destination).  The six
// double-buffered vectors laid out one after the other are in the slowest class at 1024^3.  So, when a handle whose
// vectors do not fit the Infinity Cache is built, x and `place_slots` vector-sized SLOTS are allocated back to back (each
// its own allocation: consecutive allocations are consecutive in HBM -- that is what the distance experiments rest on),
// which six slots hold r, q, p[0], r2, q2, p[1] is decided by timing the real kernel on a few placements, and the slots
// the search did not choose are FREED again: the handle ends up holding its seven arrays and nothing else.  The
// arithmetic does not see the choice.  (Round 2 held all the slots for the handle's lifetime -- 60.7 GB at 1024^3 for
// 35.5 GB of arrays -- and always put a 48 GB spacer between the two halves of the slots.)
static constexpr size_t kSlotAlign = (size_t)2 << 20;

// what the search of an earlier handle of this process found for the same set of allocations (device, sizes, counts):
// the next handle of that shape allocates the same set (the driver hands the memory just freed out again), takes the
// cached placement, times ONE cycle and searches only if that is slower than it was -- a REV study or `direction = All`
// pays for one search, not for one per handle.
namespace {
struct PlaceKey {
  int device, nslots, nfront;
  size_t slot_bytes, x_bytes, spacer_bytes;
  // everything else that sets the time of the kernel being placed: element size of p, Jacobi, tile shape, chunk length,
  // launch order, pairwise x update, kernel form.  One entry per configuration -- a faster configuration of the same
  // shape must not pass another one's `<= 1.04 * ms_best` check unsearched, a slower one must not overwrite its entry.
  unsigned long long cfg;
  bool operator<(const PlaceKey& o) const {
    return std::tie(device, nslots, nfront, slot_bytes, x_bytes, spacer_bytes, cfg) <
           std::tie(o.device, o.nslots, o.nfront, o.slot_bytes, o.x_bytes, o.spacer_bytes, o.cfg);
  }
};
struct PlaceEntry { int pl[6]; double ms_first, ms_best; };
std::mutex g_place_mutex;
std::map<PlaceKey, PlaceEntry> g_place_cache;
}  // namespace

static void vecset_free(VecSet& v) {
  for (int k = 0; k < v.nslots; ++k)
    if (v.slot[k] && !v.block) (void)oi::dev_free(v.slot[k]);
  if (v.block) (void)oi::dev_free(v.block);
  if (v.x) (void)oi::dev_free(v.x);
  v = VecSet();
}

// pl[0..5] = slots of r, q, p[0], r2, q2, p[1]
static void assign_slots(oi_tort* h, const int* pl) {
  const VecSet& v = h->vs;
  h->r = (double*)v.slot[pl[0]]; h->q = (double*)v.slot[pl[1]]; h->p[0] = v.slot[pl[2]];
  h->r2 = (double*)v.slot[pl[3]]; h->q2 = (double*)v.slot[pl[4]]; h->p[1] = v.slot[pl[5]];
  if (v.x) h->x = (double*)v.x;
  for (int k = 0; k < 6; ++k) h->placement[k] = pl[k];
  h->placement[6] = 0;  // x is the first allocation of the set
}

// the slots the placement does not use go back to the driver
static void release_unchosen(oi_tort* h) {
  VecSet& v = h->vs;
  if (v.block) return;
  for (int k = 0; k < v.nslots; ++k) {
    bool used = false;
    for (int c = 0; c < 6; ++c) used = used || h->placement[c] == k;
    if (!used && v.slot[k]) { (void)oi::dev_free(v.slot[k]); v.slot[k] = nullptr; v.bytes -= h->slot_bytes; }
  }
}

// x (search only) and the slots.  With `search`: x, the first half of the slots, a spacer, the second half -- the spacer
// is freed again at once.  It exists to give SMALL sets the distances a large one has by itself: a read stream and a
// written stream fall into different bank groups from some 40 GiB of distance on (a copy runs at 6.05 instead of 5.4-5.7
// TB/s: profiles/r02/placement/hbm_window_dist256.txt); the 59 GB set of a 1024^3 volume spans that without help, the
// 15-30 GB set of a 256- or 512-plane slab does not (those slabs stayed in the slow class without it,
// free_offset_search_discarded.txt).  So the spacer only fills the set up to `place_extent_gb`; allocating memory another
// process has used costs the driver's clearing of it (1.3-4.8 s for 48 GB on this pool: profiles/r03/vmm_probe_*.txt,
// GPUTEST_r02's 3.07 s of handle creation), which is why it is no larger than it has to be.
static oi_status vecset_alloc(oi_tort* h, bool search) {
  VecSet& v = h->vs;
  v = VecSet();
  const size_t vec = (size_t)h->nvec_cg * sizeof(double);
  h->slot_bytes = (vec + kSlotAlign - 1) / kSlotAlign * kSlotAlign;
  const double t0 = now_s();
  if (!search) {
    h->x_bytes = 0;
    if (oi::raw_malloc((void**)&v.block, 6 * h->slot_bytes) != hipSuccess) {
      (void)hipGetLastError();
      oi::set_error("out of device memory for the CG vectors (%zu bytes)", 6 * h->slot_bytes);
      return OI_HIP_ERROR;
    }
    v.nslots = 6; v.nfront = 6; v.bytes = 6 * h->slot_bytes;
    for (int k = 0; k < 6; ++k) v.slot[k] = v.block + (size_t)k * h->slot_bytes;
    h->place_alloc_s += now_s() - t0;
    return OI_OK;
  }
  // x joins the set (where it lies relative to the vectors matters as much as where they lie)
  h->x_bytes = ((size_t)h->g.nvec * sizeof(double) + kSlotAlign - 1) / kSlotAlign * kSlotAlign;
  size_t free_b = 0, total_b = 0;
  HIPC(hipMemGetInfo(&free_b, &total_b));
  // (hipMemGetInfo counts the blocks the process's pool holds as used; an allocation that needs them gets them back -- raw_malloc
  // trims the pool when the driver refuses -- so they count as free when the set is sized)
  free_b += oi::dev_pool_held(h->device);
  // spare slots only while a quarter of the card stays free for the caller's own arrays
  int want = std::min(kMaxSlots, std::max(6, h->g.tune.place_slots));
  for (; want > 6 && (size_t)want * h->slot_bytes + h->x_bytes + total_b / 4 > free_b; --want) {}
  const size_t set_bytes = (size_t)want * h->slot_bytes + h->x_bytes;
  double spacer_gb = std::min((double)h->g.tune.place_spacer_gb, (double)h->g.tune.place_extent_gb - set_bytes / 1e9);
  if (spacer_gb < 4.0 || set_bytes + (size_t)(spacer_gb * 1e9) + total_b / 8 > free_b) spacer_gb = 0.0;
  // attempts: with the spacer; without; the six slots that are needed
  for (int attempt = 0;; ++attempt) {
    if (attempt == 1) spacer_gb = 0.0;
    if (attempt == 2) want = 6;
    const int na = spacer_gb > 0.0 ? std::max(6, (want + 1) / 2) : want;  // (the six needed slots lie in front of the spacer)
    char* spacer = nullptr;
    double t_spare = now_s();
    bool ok = oi::raw_malloc((void**)&v.x, h->x_bytes) == hipSuccess;
    for (int k = 0; ok && k < want; ++k) {
      // SPARE slots (and the spacer) only while the driver hands memory out quickly: memory another process has used is
      // cleared in one lump when an allocation runs past the clean memory (one hipMalloc of 4.2 GB: 0.2 ms or 5.8 s,
      // profiles/r03/alloc_times_clean_dirty_clean.txt).  x and six slots are needed whatever they cost; what the search adds is bounded
      // (the budget counts the SPARE allocations only: a lump that fell on x or the six slots a handle needs anyway is paid and has
      // left the memory clean -- giving up the spare slots then would cost the placement, 6.93 instead of 6.4 ms per iteration in
      // profiles/r03/bench_r03d_after_the_test_suite.json, and save nothing)
      if (k == 6) t_spare = now_s();
      if (k >= 6 && now_s() - t_spare > (double)h->g.tune.place_alloc_budget_s) { want = k; break; }
      if (k == na && k >= 6) ok = oi::raw_malloc((void**)&spacer, (size_t)(spacer_gb * 1e9)) == hipSuccess;
      const double ta = now_s();
      ok = ok && oi::raw_malloc((void**)&v.slot[k], h->slot_bytes) == hipSuccess;
      if (ok) v.nslots = k + 1;
      if (h->opts.verbose > 2) fprintf(stderr, "[openimpala_amd]   slot %d allocated in %.1f ms (%.2f s since the set was begun)\n", k, 1e3 * (now_s() - ta), now_s() - t0);
    }
    if (spacer) (void)oi::dev_free(spacer);
    if (ok) {
      const size_t set_bytes = (size_t)v.nslots * h->slot_bytes + h->x_bytes;
      v.nfront = std::min(na, v.nslots); v.bytes = set_bytes;
      h->place_spacer_bytes = (size_t)(spacer_gb * 1e9);
      break;
    }
    (void)hipGetLastError();
    vecset_free(v);
    if (attempt >= 2) { oi::set_error("out of device memory for the CG vectors (%zu bytes)", 6 * h->slot_bytes + h->x_bytes); return OI_HIP_ERROR; }
  }
  h->place_peak_bytes = std::max(h->place_peak_bytes, (double)(h->bytes_device + v.bytes));
  h->place_alloc_s += now_s() - t0;
  return OI_OK;
}

// one timed 2-cycle of the real kernel at the current assignment (kernel 2: reads r2,q2,p[1], writes r,q,p[0] and x;
// kernel 3: the other way), ms per iteration; `warm` untimed cycles first
static oi_status time_cycle(oi_tort* h, hipEvent_t a, hipEvent_t b, int warm, double* per_iter) {
  hipStream_t s = h->stream;
  for (int rep = 0; rep < warm + 1; ++rep) {
    if (rep == warm) HIPC(hipEventRecord(a, s));
    OIC(cg_one_iteration(h, 2));
    OIC(cg_one_iteration(h, 3));
  }
  float ms = 0.f;
  HIPC(hipEventRecord(b, s));
  HIPC(hipEventSynchronize(b));
  HIPC(hipEventElapsedTime(&ms, a, b));
  *per_iter = 0.5 * ms;
  return OI_OK;
}

// time `place_trials` placements and keep the fastest.  Zeroed vectors and records: the launches are ordinary iterations
// of the zero system (tol2 < 0 so that they do not stop); cg_begin initialises everything afterwards.  Every rank runs
// the same number of launches (the slab path has collectives inside), the choice itself is local.  A cached placement
// of the same set of allocations is verified with one cycle first.
static oi_status place_search(oi_tort* h) {
  int trials = h->g.tune.place_trials;
  hipStream_t s = h->stream;
  VecSet& v = h->vs;
  const int ns = v.nslots;
  HIPC(hipMemsetAsync(v.x, 0, h->x_bytes, s));
  for (int k = 0; k < ns; ++k) HIPC(hipMemsetAsync(v.slot[k], 0, h->slot_bytes, s));
  HIPC(hipMemsetAsync(h->rec, 0, sizeof(double) * kRec1Stride * h->hist_cap, s));
  Ctrl c;
  c.done = 0; c.pad = 0; c.tol2 = -1.0;
  *h->h_ctrl = c;
  HIPC(hipMemcpyAsync(h->ctrl, h->h_ctrl, sizeof(Ctrl), hipMemcpyHostToDevice, s));
  const int profile = h->opts.profile;
  h->opts.profile = 0;
  hipEvent_t a = nullptr, b = nullptr;
  HIPC(hipEventCreate(&a)); HIPC(hipEventCreate(&b));
  oi_status rc = OI_OK;
  const oi::Tune& tu = h->g.tune;
  const unsigned long long cfg = (unsigned long long)(h->p_f32 ? 1 : 0) | (unsigned long long)(h->jacobi ? 2 : 0) |
                                 (unsigned long long)(tu.cg1_xskip ? 4 : 0) | (unsigned long long)(tu.cg1_fullline ? 8 : 0) |
                                 (unsigned long long)(tu.cg1_seg ? 16 : 0) | ((unsigned long long)(tu.cg1_kernel & 3) << 5) |
                                 ((unsigned long long)(tu.tile_map & 7) << 7) | ((unsigned long long)(tu.cg1_rows & 15) << 10) |
                                 ((unsigned long long)(tu.cg1_waves & 63) << 14) | ((unsigned long long)(tu.zchunk & 0xffff) << 20);
  const PlaceKey key{h->device, ns, v.nfront, h->slot_bytes, h->x_bytes, h->place_spacer_bytes, cfg};
  int best[6] = {0, 1, 2, 3, 4, 5};
  double best_ms = 0.0, per_iter = 0.0;
  bool done = false;
  h->placement_tried = 0;
  // ---- the cache of this process ----
  PlaceEntry hit{};
  int have = 0;
  if (h->g.tune.place_cache) {
    std::lock_guard<std::mutex> lk(g_place_mutex);
    auto it = g_place_cache.find(key);
    if (it != g_place_cache.end()) { hit = it->second; have = 1; }
  }
  int all_have = have ? 0 : 1;  // (1 = somebody lacks an entry)
  if (h->comm && h->comm->allreduce_max_i32_host(&all_have, 1, s) != OI_OK) rc = OI_RCCL_ERROR;
  if (rc == OI_OK && !all_have) {
    assign_slots(h, hit.pl);
    rc = time_cycle(h, a, b, 1, &per_iter);
    int miss = (rc == OI_OK && per_iter <= 1.04 * hit.ms_best) ? 0 : 1;
    if (rc == OI_OK && h->comm && h->comm->allreduce_max_i32_host(&miss, 1, s) != OI_OK) rc = OI_RCCL_ERROR;
    if (rc == OI_OK && !miss) {
      for (int k = 0; k < 6; ++k) best[k] = hit.pl[k];
      best_ms = per_iter;
      h->placement_ms_first = hit.ms_first;
      h->placement_tried = 1;
      h->place_cache_hit = 1;
      done = true;
    }
  }
  // ---- the search ----
  unsigned rng = 20260001u;
  auto rnd = [&](int m) { rng = rng * 1664525u + 1013904223u; return (int)((rng >> 8) % (unsigned)m); };
  // a tenth of all placements are in the fastest class (profiles/r02/k_cg1b_sweep.txt): 40 random ones miss it once in 70
  // handles, 21 missed it once in 9; the last eight trials are small changes of the best
  int nrandom = trials > 12 ? trials - 8 : trials - trials / 3;
  for (int t = 0; !done && t < trials && rc == OI_OK; ++t) {
    int pl[6];
    if (t < nrandom) {  // the in-order layout, then random ones
      int perm[kMaxSlots];
      for (int k = 0; k < ns; ++k) perm[k] = k;
      if (t > 0)
        for (int k = ns - 1; k > 0; --k) std::swap(perm[k], perm[rnd(k + 1)]);
      for (int k = 0; k < 6; ++k) pl[k] = perm[k];
    } else {  // then small changes of the best one: a vector to a free slot, two vectors swapped
      for (int k = 0; k < 6; ++k) pl[k] = best[k];
      if (ns > 6 && rnd(2)) {
        bool used[kMaxSlots] = {false};
        for (int k = 0; k < 6; ++k) used[pl[k]] = true;
        int nth = rnd(ns - 6);
        for (int q = 0; q < ns; ++q)
          if (!used[q] && nth-- == 0) { pl[rnd(6)] = q; break; }
      } else {
        const int i = rnd(6), j = (i + 1 + rnd(5)) % 6;
        std::swap(pl[i], pl[j]);
      }
    }
    assign_slots(h, pl);
    rc = time_cycle(h, a, b, t == 0 ? 1 : 0, &per_iter);  // one untimed cycle at the very start
    if (rc != OI_OK) break;
    if (h->opts.verbose > 2)
      fprintf(stderr, "[openimpala_amd]   placement %2d: %d %d %d | %d %d %d  %.3f ms\n", t, pl[0], pl[1], pl[2], pl[3], pl[4], pl[5], per_iter);
    if (t == 0) {
      h->placement_ms_first = per_iter;
      // the search may take about one second: very large slabs (config 5's share: 49 ms per iteration) get fewer trials.
      // Every rank must run the same number of launches, so the slowest rank's time decides
      int tenth_ms = (int)std::min(1e8, per_iter * 10.0) + 1;
      if (h->comm && h->comm->allreduce_max_i32_host(&tenth_ms, 1, s) != OI_OK) { rc = OI_RCCL_ERROR; break; }
      const int afford = std::max(8, (int)(1000.0 / (2.0 * 0.1 * tenth_ms)));
      if (afford < trials) { trials = afford; nrandom = trials > 12 ? trials - 8 : trials - trials / 3; }
    }
    if (t == 0 || per_iter < best_ms) { best_ms = per_iter; for (int k = 0; k < 6; ++k) best[k] = pl[k]; }
    h->placement_tried = t + 1;
  }
  (void)hipEventDestroy(a); (void)hipEventDestroy(b);
  h->opts.profile = profile;
  OIC(rc);
  assign_slots(h, best);
  h->placement_ms_best = best_ms;
  if (!done && h->g.tune.place_cache) {
    PlaceEntry e{};
    for (int k = 0; k < 6; ++k) e.pl[k] = best[k];
    e.ms_first = h->placement_ms_first; e.ms_best = best_ms;
    std::lock_guard<std::mutex> lk(g_place_mutex);
    g_place_cache[key] = e;
  }
  if (h->opts.verbose > 1)
    fprintf(stderr, "[openimpala_amd] vector placement: %d slots, %d tried%s, %.3f -> %.3f ms per iteration (%d %d %d | %d %d %d)\n", ns,
            h->placement_tried, done ? " (cached placement verified)" : "", h->placement_ms_first, best_ms, best[0], best[1], best[2], best[3],
            best[4], best[5]);
  return OI_OK;
}

// About one process in ten to fifteen gets allocations in which everything is slow wherever the vectors lie -- a 3-read / 3-write
// stream of the vectors moves 5.3-5.4 TB/s there against 6.05-6.4 elsewhere (profiles/r02/placement/final_12_runs.txt,
// bench_r02f_flat_arena.json).  That is a property of where the driver put those allocations, so the remedy is other allocations:
// when the stream of the kept vectors is slow, a second set is allocated while the chosen seven arrays of the first are still held
// (so that it lands elsewhere), searched, and the better set kept.  Costs a second search in that one case only.
static oi_status place_vectors(oi_tort* h) {
  if (!h->vs.x || h->g.tune.place_trials <= 0 || h->hist_cap < 4) {
    release_unchosen(h);  // (a set allocated for a search that is not going to run keeps its six slots in order, nothing else)
    return OI_OK;
  }
  const double t0 = now_s();
  OIC(place_search(h));
  release_unchosen(h);
  hipStream_t s = h->stream;
  int again = 0;
  // (round 4: no longer only when the search came back flat -- with the nontemporal policy EVERY landscape is nearly flat, a good set
  // of allocations goes from 6.28 to 6.09 ms and a slow one from 7.03 to 6.79, profiles/r04/r04e_rocprof_summary_slow_placement_class.txt;
  // what tells them apart is the stream of the kept vectors, 6.05-6.3 TB/s against 5.4)
  if (!h->p_f32 && !h->place_cache_hit) {
    const long long nd = h->nvec_cg / 512 * 512;
    hipEvent_t a = nullptr, b = nullptr;
    HIPC(hipEventCreate(&a)); HIPC(hipEventCreate(&b));
    float ms = 0.f;
    for (int rep = 0; rep < 3; ++rep) {
      if (rep == 1) (void)hipEventRecord(a, s);
      launch_stream_3r3w(h->r, h->q, (double*)h->p[0], h->r2, h->q2, (const double*)h->p[1], nd, s);
    }
    const bool ok = hipEventRecord(b, s) == hipSuccess && hipEventSynchronize(b) == hipSuccess && hipEventElapsedTime(&ms, a, b) == hipSuccess;
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    if (ok && ms > 0.f && 2.0 * 48.0 * (double)nd / (ms * 1e-3) < 5.8e12) again = 1;  // bytes per second of the stream
    if (again && h->opts.verbose > 1) fprintf(stderr, "[openimpala_amd] vector placement: the stream of the kept vectors is slow (%.0f GB/s)\n", 2.0 * 48.0 * (double)nd / (ms * 1e-3) / 1e9);
    // (round 5) and by the kernel itself.  The stream does not see the other slow class: sets in which every placement of the
    // vectors sits on a plateau 8 % up (6.46-6.53 ms at 1024^3 where thirty other processes kept 5.92-6.14, the stream of the kept
    // vectors at 6.0-6.1 TB/s like theirs: profiles/r05/placement_slow_plateau.txt; x and the code bytes are not moved by the search).
    // Known since round 2 ("one process in ten"), and one bench line in thirteen carried it.  In algorithmic bytes per second the
    // two groups are 4 % apart at their closest.
    const double alg_bytes = (double)h->g.nx * h->g.ny * h->g.nzl + 60.0 * (double)h->nvec_cg;
    // (single-domain handles only: the threshold is calibrated on whole volumes -- a slab's iteration carries its plane exchange, 4.6 TB/s
    // on the one-rank rehearsal of a 128-plane slab, and would ask for a second search on every rank every time)
    if (!h->comm && h->g.tune.place_retry_gbps > 0.f && h->placement_ms_best > 0.0 &&
        alg_bytes / (h->placement_ms_best * 1e-3) < (double)h->g.tune.place_retry_gbps * 1e9) {
      // only where a second x + six slots leave a quarter of the card free: a slab that fills the card (configuration 5's share: 135 GB
      // of vectors, 4070 GB/s for other reasons) is not worth tens of GB of the driver's clearing
      size_t free_b = 0, total_b = 0;
      if (hipMemGetInfo(&free_b, &total_b) == hipSuccess &&
          6 * h->slot_bytes + h->x_bytes + total_b / 4 <= free_b + oi::dev_pool_held(h->device)) {
        again = 1;
        if (h->opts.verbose > 1) fprintf(stderr, "[openimpala_amd] vector placement: the kept kernel is slow (%.0f algorithmic GB/s)\n", alg_bytes / (h->placement_ms_best * 1e-3) / 1e9);
      }
      else
        (void)hipGetLastError();
    }
  }
  if (h->comm) OIC(h->comm->allreduce_max_i32_host(&again, 1, s));  // (the search has collectives inside on slabs)
  if (again) {
    struct Kept { VecSet v; int pl[7], tried; double first, best; size_t spacer; } old;
    old.v = h->vs;
    for (int k = 0; k < 7; ++k) old.pl[k] = h->placement[k];
    old.tried = h->placement_tried; old.first = h->placement_ms_first; old.best = h->placement_ms_best; old.spacer = h->place_spacer_bytes;
    h->vs = VecSet();
    h->bytes_device += old.v.bytes;  // (held during the second search: the peak figure counts it)
    oi_status rc = vecset_alloc(h, true);
    h->bytes_device -= old.v.bytes;
    // (from here to the end of this block nothing returns early: until one of the two sets has been freed and the other is
    // h->vs again, `old.v` is owned by this frame only -- an error is carried in `rc` to the branch that puts the first set back)
    int failed = rc == OI_OK ? 0 : 1;
    oi_status rc_comm = OI_OK;
    if (h->comm) rc_comm = h->comm->allreduce_max_i32_host(&failed, 1, s);
    if (rc_comm != OI_OK) failed = 1;
    const int cache = h->g.tune.place_cache;
    h->g.tune.place_cache = 0;  // (the entry of this shape describes the FIRST set)
    if (!failed) rc = place_search(h);
    h->g.tune.place_cache = cache;
    if (hipStreamSynchronize(s) != hipSuccess && rc == OI_OK && !failed) {
      oi::set_error("HIP error after the second placement search: %s", hipGetErrorString(hipGetLastError()));
      rc = OI_HIP_ERROR;
    }
    if (!failed && rc == OI_OK && h->placement_ms_best < old.best) {
      release_unchosen(h);
      vecset_free(old.v);
      h->placement_tried += old.tried;
      if (h->opts.verbose > 1)
        fprintf(stderr, "[openimpala_amd] vector placement: second set of allocations kept (%.3f ms; the first: %.3f -> %.3f)\n",
                h->placement_ms_best, old.first, old.best);
    } else {  // the first set stays
      const int tried2 = failed ? 0 : h->placement_tried;
      if (h->opts.verbose > 1 && !failed && rc == OI_OK)
        fprintf(stderr, "[openimpala_amd] vector placement: the first set of allocations stays (%.3f ms; the second: %.3f)\n", old.best, h->placement_ms_best);
      vecset_free(h->vs);
      h->vs = old.v;
      h->place_spacer_bytes = old.spacer;
      assign_slots(h, old.pl);
      h->placement_tried = old.tried + tried2;
      h->placement_ms_first = old.first; h->placement_ms_best = old.best;
      if (failed) (void)hipGetLastError();
      if (rc_comm != OI_OK) return rc_comm;
      if (!(rc == OI_OK || failed)) return rc;
    }
  }
  h->place_search_s += now_s() - t0;
  return OI_OK;
}

static oi_status build(oi_tort* h, const uint8_t* phase, bool on_device) {
  Geom& g = h->g;
  const double t0 = now_s();
  HIPC(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  hipStream_t s = h->stream;
  if (h->comm) {
    // The exchange stream is an ordinary stream.  Round 4's review asked for the highest stream priority, so that RCCL's send/recv
    // kernel cannot queue behind the interior march's resident blocks; measured on the one-rank rehearsal of the split-launch path
    // (1024 x 1024 x 128 slab, profiles/r05/slab_exchange_stream_priority_ab.txt): ANY stream made by hipStreamCreateWithPriority --
    // greatest or least -- makes the two concurrent launches of an iteration 7-8 % slower (0.95-0.97 against 0.88-0.90 ms), three
    // times out of three.  Whether a priority pays once a real exchange kernel runs can only be measured on a multi-GPU node:
    // OI_COMM_PRIO=greatest (or least) in the environment selects it for such a run.
    const char* ab = getenv("OI_COMM_PRIO");
    if (ab && (!strcmp(ab, "greatest") || !strcmp(ab, "least"))) {
      int prio_least = 0, prio_greatest = 0;
      HIPC(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));  // (numerically lower = higher priority)
      const int prio = !strcmp(ab, "least") ? prio_least : prio_greatest;
      HIPC(hipStreamCreateWithPriority(&h->comm_stream, hipStreamNonBlocking, prio));
      h->comm_stream_priority = prio;
    } else {
      HIPC(hipStreamCreateWithFlags(&h->comm_stream, hipStreamNonBlocking));
    }
    HIPC(hipStreamCreateWithFlags(&h->aux_stream, hipStreamNonBlocking));
    HIPC(hipEventCreateWithFlags(&h->ev_boundary, hipEventDisableTiming));
    HIPC(hipEventCreateWithFlags(&h->ev_exchanged, hipEventDisableTiming));
    HIPC(hipEventCreateWithFlags(&h->ev_start, hipEventDisableTiming));
    HIPC(hipEventCreateWithFlags(&h->ev_interior, hipEventDisableTiming));
  }

  // Slabs build in lock-step: every stretch of rank-local work (allocations, set-up kernels) between two collectives ends in
  // oi::agree_status, so a slab that cannot get its memory takes its neighbours out with an error instead of leaving them waiting.
  const size_t nphase = (size_t)g.nx * g.ny * g.nzl;
  auto local_bits = [&]() -> oi_status {
    OIC(dmalloc(h, &h->open, (size_t)g.nwords));
    OIC(dmalloc(h, &h->active, (size_t)g.nwords));
    OIC(dmalloc(h, &h->rhi, (size_t)g.nwords));
    OIC(dmalloc(h, &h->counters, 8));
    OIC(dmalloc(h, &h->changed, 1));
    OIC(dmalloc(h, &h->scal, 8));
    OIC(dmalloc(h, &h->partials, (size_t)8 * g.tune.max_blocks));
    // slice sums + ticket of k_finalize1_sliced: their own array (the ticket must stay zero between launches, so nothing else may
    // ever write there -- round 3 carved it out of the tail of `partials`, safe only while no kernel used the full stride)
    OIC(dmalloc(h, &h->fin_scratch, (size_t)kFinScratch));
    HIPC(hipMemsetAsync(h->fin_scratch, 0, sizeof(double) * kFinScratch, h->stream));
    OIC(dmalloc(h, &h->ctrl, 1));
    HIPC(hipHostMalloc((void**)&h->h_ctrl, sizeof(Ctrl)));
    HIPC(hipHostMalloc((void**)&h->h_counters, 8 * sizeof(unsigned long long)));
    HIPC(hipHostMalloc((void**)&h->h_changed, sizeof(int)));
    HIPC(hipHostMalloc((void**)&h->h_scal, 8 * sizeof(double)));
    HIPC(hipMemsetAsync(h->open, 0, sizeof(uint64_t) * g.nwords, s));
    HIPC(hipMemsetAsync(h->active, 0, sizeof(uint64_t) * g.nwords, s));
    HIPC(hipMemsetAsync(h->rhi, 0, sizeof(uint64_t) * g.nwords, s));
    HIPC(hipMemsetAsync(h->counters, 0, 8 * sizeof(unsigned long long), s));
    // phase bytes -> open bits (+ phase voxel count)
    const uint8_t* dphase = phase;
    DevTmp<uint8_t> tmp;
    if (!on_device) {
      HIPC(oi::dev_malloc((void**)&tmp.p, nphase));
      HIPC(hipMemcpyAsync(tmp.p, phase, nphase, hipMemcpyHostToDevice, s));
      dphase = tmp.p;
    }
    { PROF(K_PACK); launch_pack_open(dphase, g, h->phase_id, h->open, &h->counters[3], s); }
    HIPC(hipStreamSynchronize(s));
    return OI_OK;
  };
  OIC(oi::agree_status(h->comm, local_bits(), s, "the allocation of the mask bits"));
  h->st.t_upload_s = now_s() - t0;

  // flood fill from both faces (generateActivityMask, TortuosityHypre.cpp:394-558)
  const double t1 = now_s();
  { PROF(K_SEED); launch_seed(g, h->open, h->active, h->rhi, s); }
  const int sweeps_per_round = 8;
  int sweeps = 0;
  // Run to the fixed point, always: the fill is monotone (a round either claims a voxel or is the last), so it ends after at
  // most one round per voxel.  The reference caps its sweeps at nx + ny + nz + 2 and only warns (:328,382); a cap here
  // would silently hand an incomplete mask -- wrong active_vf and tau with status OK -- to serpentine paths that cross
  // slab boundaries once per round.  The bound below is a guard against a broken transport, not a budget: it is an error.
  const long long max_rounds = (long long)g.nx * g.ny * g.nz + 8;
  bool fixed_point = false;
  for (long long round = 0; round < max_rounds; ++round) {
    HIPC(hipMemsetAsync(h->changed, 0, sizeof(int), s));
    OIC(exchange_planes(h, h->active, sizeof(uint64_t) * g.wplane));
    OIC(exchange_planes(h, h->rhi, sizeof(uint64_t) * g.wplane));
    for (int i = 0; i < sweeps_per_round; ++i) {
      PROF(K_FLOOD);
      launch_flood_sweep(g, h->open, h->active, h->rhi, h->changed, s, i & 1);  // (up, down, up, ...: fronts move both ways in z)
    }
    // ... and, for flow along z, one walk of every word column through the slab's planes: what the plane-parallel sweeps are slow at
    // (1024^3 spheres: Z mask 0.135 -> 0.033 s; the walk costs ~8 ms a launch and bought nothing for X / Y, whose fronts travel in-plane:
    // 0.018 -> 0.033 and 0.043 -> 0.116 s with it, so they go without)
    if (g.dir == 2 && g.nzl >= 16) { PROF(K_FLOOD); launch_flood_zwalk(g, h->open, h->active, h->rhi, h->changed, s); }
    sweeps += sweeps_per_round;
    HIPC(hipMemcpyAsync(h->h_changed, h->changed, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPC(hipStreamSynchronize(s));
    int ch = *h->h_changed;
    if (h->comm) OIC(h->comm->allreduce_max_i32_host(&ch, 1, s));
    if (!ch) { fixed_point = true; break; }
  }
  if (!fixed_point) {
    oi::set_error("flood fill did not reach its fixed point within %lld rounds (one per voxel): transport failure?", max_rounds);
    return OI_HIP_ERROR;
  }
  h->st.flood_sweeps = sweeps;
  OIC(exchange_planes(h, h->active, sizeof(uint64_t) * g.wplane));
  OIC(exchange_planes(h, h->rhi, sizeof(uint64_t) * g.wplane));
  { PROF(K_MASK); launch_mask_and(g, h->active, h->rhi, h->counters, s); }
  HIPC(hipMemcpyAsync(h->h_counters, h->counters, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
  HIPC(hipStreamSynchronize(s));
  unsigned long long cnt[4] = {h->h_counters[0], h->h_counters[1], h->h_counters[2], h->h_counters[3]};
  if (h->comm) OIC(h->comm->allreduce_sum_u64_host(cnt, 4, s));
  const double total = (double)g.nx * g.ny * g.nz;
  h->st.n_active = (int64_t)cnt[0];
  h->n_lo = (long long)cnt[1];
  h->n_hi = (long long)cnt[2];
  h->st.n_phase = (int64_t)cnt[3];
  h->st.vf = (double)cnt[3] / total;
  h->st.active_vf = (double)cnt[0] / total;  // TortuosityHypre.cpp:549-553
  h->st.t_mask_s = now_s() - t1;
  free_setup_buffers(h);
  h->percolates = h->st.active_vf > std::numeric_limits<double>::epsilon();  // :170
  h->st.tau = std::numeric_limits<double>::quiet_NaN();
  if (!h->percolates) {
    h->solved = true;  // value() returns NaN without solving (:170-178)
    h->value_status = OI_NO_PERCOLATION;
    return OI_OK;
  }

  // the matrix: one code byte per voxel
  const double t2 = now_s();
  auto local_codes = [&]() -> oi_status {
    OIC(dmalloc(h, &h->code, (size_t)g.nvec));
    HIPC(hipMemsetAsync(h->code, 0, (size_t)g.nvec, s));
    HIPC(hipMemsetAsync(h->counters, 0, 8 * sizeof(unsigned long long), s));
    { PROF(K_CODES); launch_codes(g, h->active, h->code, &h->counters[0], s); }
    return OI_OK;
  };
  OIC(oi::agree_status(h->comm, local_codes(), s, "the allocation of the code bytes"));
  OIC(exchange_planes(h, h->code, (size_t)g.plane));
  HIPC(hipMemcpyAsync(h->h_counters, h->counters, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
  HIPC(hipStreamSynchronize(s));
  unsigned long long nf = h->h_counters[0];
  if (h->comm) OIC(h->comm->allreduce_sum_u64_host(&nf, 1, s));
  h->st.n_free = (int64_t)nf;
  // ||b|| of the reference's full system: vlo / vhi on the active Dirichlet-face cells
  h->bnorm = std::sqrt((double)h->n_lo * g.vlo * g.vlo + (double)h->n_hi * g.vhi * g.vhi);

  h->maxiter = h->opts.maxiter > 0 ? h->opts.maxiter : 40 * (g.nx + g.ny + g.nz);
  h->hist_cap = h->maxiter + 2;
  OIC(oi::agree_status(h->comm, dmalloc(h, &h->hist, (size_t)H_STRIDE * h->hist_cap), s, "the allocation of the iteration history"));
  // (x of the single-kernel variant may join the vector arena, see alloc_vector_slots)
  h->nvec_cg = g.nvec;
  const bool want_mg = h->opts.precond == OI_PRECOND_MG;
  h->mg_tiled = g.tune.mg_plain ? 0 : 1;
  // the multigrid-preconditioned two-kernel iteration takes the compacted layout when its fine-level
  // kernels are the tiled ones with the restriction fused in (even first plane, see mg_vcycle)
  // (... and when the descriptor-addressed first kernel can serve them: k_fused_a on compacted vectors is a cross-check kernel of
  // the test library -- other tile heights, option fused_kernel -- the product keeps such handles on dense vectors)
  const bool mg_descr = g.tune.fused_kernel == 0 && g.tune.fused_rows == 2 && 8ll * g.plane < 0x7fff0000ll;
  const bool mg_compact = want_mg && h->fused && h->mg_tiled && (g.z0 & 1) == 0 && (mg_descr || oi::xcheck_built());
  if ((h->cg1 || (want_mg && h->fused)) && g.tune.cg1_compact) {
    // r, q, p (and z): only the kLV-voxel lines that hold a free voxel
    const long long nlines = (long long)(g.nzl + 2) * g.ny * ((g.pitch + oi::kLV - 1) / oi::kLV);
    // neighbours exchange ghost planes as line lists: every slab takes the same layout
    int dense = ((h->cg1 || mg_compact) && nlines < (1ll << 31)) ? 0 : 1;
    if (h->comm) OIC(h->comm->allreduce_max_i32_host(&dense, 1, s));
    if (dense && h->cg1 && !oi::xcheck_built()) {
      oi::set_error("the single-kernel CG iteration needs the compacted line layout (%lld lines of %d voxels do not fit 31 bits, or a slab could "
                    "not take it): choose kernel = fused, unfused or lean for this volume", nlines, oi::kLV);
      return OI_BAD_ARG;
    }
    if (!dense) {
      auto local_lines = [&]() -> oi_status {
        OIC(dmalloc(h, &h->lineoff, (size_t)nlines));
        OIC(dmalloc(h, &h->plane_first_dev, (size_t)g.nzl + 3));
        OIC(dmalloc(h, &h->seg, (size_t)(g.nzl + 2) * g.ny * ((g.pitch + 127) / 128)));  // segment records (k_cg1b, both forms)
        HIPC(launch_build_lines(g, h->code, h->lineoff, h->plane_first_dev, h->seg, s));
        h->plane_first.resize((size_t)g.nzl + 3);
        HIPC(hipMemcpy(h->plane_first.data(), h->plane_first_dev, sizeof(long long) * (g.nzl + 3), hipMemcpyDeviceToHost));
        h->nvec_cg = std::max<long long>(16, (long long)oi::kLV * h->plane_first[g.nzl + 2]);
        g.live_lines = (float)((double)h->plane_first[g.nzl + 2] / (double)nlines);
        return OI_OK;
      };
      OIC(oi::agree_status(h->comm, local_lines(), s, "the set-up of the compacted line lists"));
      OIC(check_plane_message_lengths(h, s));
    }
  }
  if (h->cg1) {
    // (on slabs every rank must take the same decision: the search runs iterations, and those have collectives inside)
    int no_search = (g.tune.place_trials > 0 && h->hist_cap >= 4 && (size_t)h->nvec_cg * sizeof(double) >= ((size_t)g.tune.place_min_mb << 20)) ? 0 : 1;
    if (h->comm) OIC(h->comm->allreduce_max_i32_host(&no_search, 1, s));
    auto local_vectors = [&]() -> oi_status {
      OIC(dmalloc(h, &h->rec, (size_t)kRec1Stride * h->hist_cap));
      OIC(vecset_alloc(h, !no_search));
      const int first[6] = {0, 1, 2, 3, 4, 5};
      assign_slots(h, first);
      if (!h->vs.x) OIC(dmalloc(h, &h->x, (size_t)g.nvec));
      return OI_OK;
    };
    OIC(oi::agree_status(h->comm, local_vectors(), s, "the allocation of the CG vectors"));
  } else {
    auto local_vectors = [&]() -> oi_status {
      OIC(dmalloc(h, &h->x, (size_t)g.nvec));
      OIC(dmalloc(h, &h->r, (size_t)h->nvec_cg));
      if (!h->lean) OIC(dmalloc(h, &h->q, (size_t)h->nvec_cg));
      for (int b = 0; b < (h->fused ? 2 : 1); ++b) {
        HIPC(oi::dev_malloc(&h->p[b], (size_t)h->nvec_cg * h->p_elem));
        h->bytes_device += (size_t)h->nvec_cg * h->p_elem;
      }
      if (want_mg) {
        OIC(dmalloc(h, &h->z, (size_t)h->nvec_cg));
        HIPC(hipMemsetAsync(h->z, 0, sizeof(double) * h->nvec_cg, s));
      }
      return OI_OK;
    };
    OIC(oi::agree_status(h->comm, local_vectors(), s, "the allocation of the CG vectors"));
  }
  if (want_mg && h->fused && h->lineoff) {
    // the rotated iteration (residual update fused into the pre-smoother): every slab must be able to run it
    int cannot = (fused_descr_ok(g, h->lineoff, h->plane_first_dev, h->seg, h->jacobi ? 1 : 0) && h->mg_tiled && (g.z0 & 1) == 0 && !g.tune.mg_norot) ? 0 : 1;
    if (h->comm) OIC(h->comm->allreduce_max_i32_host(&cannot, 1, s));
    h->rot = !cannot;
    if (h->rot) OIC(oi::agree_status(h->comm, dmalloc(h, &h->r2, (size_t)h->nvec_cg), s, "the allocation of the second residual buffer"));
  }
  if (want_mg && g.tune.mg_nu_coarse == 0) {
    // sweeps per coarse-level visit by the structure (every slab the same: the cycle has collectives inside)
    int one = (h->lineoff && g.live_lines > 0.f && g.live_lines < 0.35f) ? 0 : 1;
    if (h->comm) OIC(h->comm->allreduce_max_i32_host(&one, 1, s));
    g.tune.mg_nu_coarse = one ? 1 : 2;
  }
  if (want_mg) {
    size_t mg_bytes = 0;
    h->mg = mg_setup(g, h->code, h->comm, 0, s, &mg_bytes);  // (agrees on its own allocations)
    if (!h->mg) return OI_HIP_ERROR;
    h->bytes_device += mg_bytes;
    if (h->opts.verbose > 1)
      fprintf(stderr, "[openimpala_amd] multigrid hierarchy: %d levels (replicated from level %d), %.1f MB\n",
              mg_levels(h->mg), mg_replicated_from(h->mg), mg_bytes / 1e6);
  }
  if (h->cg1) {
    OIC(place_vectors(h));
    h->bytes_device += h->vs.bytes;  // (after the unchosen slots have gone back)
    h->place_peak_bytes = std::max(h->place_peak_bytes, (double)h->bytes_device);
  }
  h->vectors_ready = true;
  h->st.t_setup_s = now_s() - t2;
  h->st.bytes_device = (double)h->bytes_device;
  return OI_OK;
}

static oi_status create_common(oi_tort** out, oi_comm* comm, const uint8_t* phase, bool on_device, int nx,
                               int ny, int nz, int z0, int nzl, int phase_id, int dir, double vlo, double vhi,
                               const oi_tort_opts* opts) {
  if (!out) { oi::set_error("%s: null handle or argument out of range", __func__); return OI_BAD_ARG; }
  *out = nullptr;
  if (!phase || nx < 1 || ny < 1 || nz < 1 || nzl < 1 || z0 < 0 || z0 + nzl > nz || dir < 0 || dir > 2) {
    oi::set_error("bad geometry/direction");
    return OI_BAD_ARG;
  }
  oi_tort_opts o;
  oi_tort_default_opts(&o);
  if (opts) o = *opts;
  if (!(o.eps > 0.0)) {  // AMREX_ALWAYS_ASSERT(m_eps > 0), TortuosityHypre.cpp:159
    oi::set_error("eps must be positive");
    return OI_BAD_ARG;
  }
  if (o.precond == OI_PRECOND_DEFAULT)
    o.precond = (o.kernel == OI_KERNEL_FUSED1 || o.kernel == OI_KERNEL_LEAN) ? OI_PRECOND_NONE : OI_PRECOND_MG;
  if (o.precond < OI_PRECOND_NONE || o.precond > OI_PRECOND_MG) {
    oi::set_error("unknown preconditioner %d", o.precond);
    return OI_BAD_ARG;
  }
  int device = 0;
  OIC(ensure_device(comm ? comm->device : o.device, &device));
  std::unique_ptr<oi_tort, void (*)(oi_tort*)> h(new oi_tort(), oi_tort_destroy);
  h->opts = o;
  // a one-rank communicator is dropped unless OI_FORCE_COMM is set (single-GPU rehearsal of
  // the collective code path)
  const oi::Tune tune0 = oi::default_tune();
  h->comm = (comm && (comm->size > 1 || tune0.force_comm)) ? comm : nullptr;
  h->device = device;
  h->phase_id = phase_id;
  h->jacobi = o.precond == OI_PRECOND_JACOBI;
  h->lean = o.kernel == OI_KERNEL_LEAN;
  h->fused = o.kernel != OI_KERNEL_UNFUSED && !h->lean;
  h->cg1 = o.kernel == OI_KERNEL_FUSED1 || o.kernel == OI_KERNEL_AUTO;  // fastest variant is the default
  if (h->cg1 && o.precond != OI_PRECOND_MG && !oi::xcheck_built() && (!tune0.cg1_compact || tune0.cg1_kernel != 0)) {
    // the branching first form of that iteration (k_cg1: dense vectors, cg1_kernel = 1) is a superseded kernel kept as the
    // bit-identity reference of the shipped one, in the test library only
    oi::set_error("the single-kernel CG iteration on dense vectors / with cg1_kernel = 1 runs k_cg1, a cross-check kernel of the test "
                  "library (make xcheck): not in this build");
    return OI_BAD_ARG;
  }
  if (o.kernel < OI_KERNEL_AUTO || o.kernel > OI_KERNEL_LEAN) {
    oi::set_error("unknown kernel variant %d", o.kernel);
    return OI_BAD_ARG;
  }
  if (h->lean && o.precond == OI_PRECOND_MG) {
    oi::set_error("the capacity mode (kernel = lean) runs plain or Jacobi-preconditioned CG");
    return OI_BAD_ARG;
  }
  if (o.precond == OI_PRECOND_MG) {
    // with neighbours the hierarchy keeps the slab partition (see oi_mg.hip); r's and z's ghost planes travel
    if (o.kernel == OI_KERNEL_FUSED1) {
      oi::set_error("the multigrid preconditioner runs on the two-kernel or unfused iteration (kernel = auto | fused | unfused)");
      return OI_BAD_ARG;
    }
    h->fused = o.kernel != OI_KERNEL_UNFUSED;  // z = M^-1 r enters the fused kernel in place of r
    h->cg1 = false;
  }
  h->p_f32 = o.p_storage == OI_STORE_F32;
  h->p_elem = h->p_f32 ? sizeof(float) : sizeof(double);
  if (h->p_f32 && !h->fused) {
    oi::set_error("fp32 storage of the search direction needs the fused kernel");
    return OI_BAD_ARG;
  }
  Geom& g = h->g;
  g.nx = nx; g.ny = ny; g.nz = nz; g.z0 = z0; g.nzl = nzl; g.dir = dir;
  g.pitch = (nx + 1) & ~1;
  g.W = (nx + 63) / 64;
  g.plane = (long long)g.pitch * ny;
  g.wplane = (long long)g.W * ny;
  g.nvec = (long long)(nzl + 2) * g.plane;
  g.nwords = (long long)(nzl + 2) * g.wplane;
  g.vlo = vlo; g.vhi = vhi;
  g.tune = tune0;  // this handle's launch shapes: a snapshot of the process defaults
  g.tune.max_blocks = oi::partial_blocks_bound(g.pitch, ny, nzl);
  const oi_status st = build(h.get(), phase, on_device);
  if (st != OI_OK) return st;
  *out = h.release();
  return OI_OK;
}

// preconditionPhaseFab (TortuosityHypre.cpp:248-292): `passes` remspot passes on a private copy of the
// phase field, on the device (dense nx*ny*nz bytes).  *out_dev is hipMalloc'ed.
// first cells of the boxes ba.maxSize(box_size) cuts an axis of n cells into (Diffusion.cpp:266-268).  AMReX 25.03 is
// not in this image; this restates BoxList::maxSize as published (Src/Base/AMReX_BoxList.cpp): chunk and length are
// halved while both are even, numblk = ceil(len / chunk), sz = len / numblk, extra = len % numblk, the first `extra`
// blocks get sz + 1 (older releases chopped the larger blocks off the HIGH end; the two agree when box_size divides n).
static void box_cut_flags(int n, int chunk, std::vector<uint8_t>& flag) {
  flag.assign((size_t)n, 0);
  if (chunk <= 0 || n <= chunk) return;
  int ratio = 1, bs = chunk, nlen = n;
  while ((bs % 2 == 0) && (nlen % 2 == 0)) { ratio *= 2; bs /= 2; nlen /= 2; }
  const int numblk = (nlen + bs - 1) / bs, sz = nlen / numblk, extra = nlen % numblk;
  for (int b = 1; b < numblk; ++b) flag[(size_t)(((b < extra) ? b * (sz + 1) : b * sz + extra) * ratio)] = 1;
}

// preconditionPhaseFab (TortuosityHypre.cpp:248-292): `passes` remspot passes on a private copy of the phase field, on
// the device.  `phase` holds this rank's planes [z0, z0 + nzl) of nz (dense nx*ny*nzl bytes); with `comm` the slabs
// relax together (ghost planes of the pre-pass field once per pass, of the relaxing field once per sweep).
// *out_dev is hipMalloc'ed; *out_phase points at its first owned plane.
static oi_status remspot_private_copy(oi_comm* comm, const uint8_t* phase, bool on_device, int nx, int ny, int nz, int z0,
                                      int nzl, int passes, int box_size, int device_opt, uint8_t** out_dev,
                                      uint8_t** out_phase) {
  int device = 0;
  OIC(ensure_device(comm ? comm->device : device_opt, &device));
  if (comm && comm->size <= 1) comm = nullptr;
  const size_t plane = (size_t)nx * ny, n = plane * (size_t)nzl, ntot = n + 2 * plane;
  uint8_t *old = nullptr, *cur = nullptr, *cuts = nullptr;
  int *changed = nullptr, *h_changed = nullptr;
  hipStream_t s = nullptr;
  oi_status rc = OI_OK;
  std::vector<uint8_t> fx, fy, fz;
  box_cut_flags(nx, box_size, fx);
  box_cut_flags(ny, box_size, fy);
  box_cut_flags(nz, box_size, fz);
  do {
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess ||
        oi::raw_malloc((void**)&old, ntot) != hipSuccess || oi::raw_malloc((void**)&cur, ntot) != hipSuccess ||
        oi::raw_malloc((void**)&cuts, (size_t)nx + ny + nz) != hipSuccess ||
        oi::raw_malloc((void**)&changed, sizeof(int)) != hipSuccess || hipHostMalloc((void**)&h_changed, sizeof(int)) != hipSuccess ||
        hipMemsetAsync(old, 0, ntot, s) != hipSuccess ||
        hipMemcpyAsync(old + plane, phase, n, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s) != hipSuccess ||
        hipMemcpyAsync(cuts, fx.data(), (size_t)nx, hipMemcpyHostToDevice, s) != hipSuccess ||
        hipMemcpyAsync(cuts + nx, fy.data(), (size_t)ny, hipMemcpyHostToDevice, s) != hipSuccess ||
        hipMemcpyAsync(cuts + nx + ny, fz.data(), (size_t)nz, hipMemcpyHostToDevice, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess) {
      rc = OI_HIP_ERROR;
      break;
    }
    for (int pass = 0; pass < passes && rc == OI_OK; ++pass) {
      // FillBoundary: the neighbours' pre-pass boundary planes
      if (comm && (rc = comm->exchange(old + plane, old, old + plane * (size_t)nzl, old + plane * (size_t)(nzl + 1), plane, s)) != OI_OK) break;
      if (hipMemcpyAsync(cur, old, ntot, hipMemcpyDeviceToDevice, s) != hipSuccess) { rc = OI_HIP_ERROR; break; }
      for (long long sweep = 0;; ++sweep) {
        if (hipMemsetAsync(changed, 0, sizeof(int), s) != hipSuccess) { rc = OI_HIP_ERROR; break; }
        if (comm && (rc = comm->exchange(cur + plane, cur, cur + plane * (size_t)nzl, cur + plane * (size_t)(nzl + 1), plane, s)) != OI_OK) break;
        launch_remspot_sweep(old, cur, nx, ny, nzl, z0, nz, cuts, cuts + nx, cuts + nx + ny, changed, s);
        if (hipMemcpyAsync(h_changed, changed, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess ||
            hipStreamSynchronize(s) != hipSuccess) { rc = OI_HIP_ERROR; break; }
        int ch = *h_changed;
        if (comm && (rc = comm->allreduce_max_i32_host(&ch, 1, s)) != OI_OK) break;
        if (!ch) break;
        if (sweep > (long long)nx + ny + nz + 8) { oi::set_error("remspot relaxation did not settle"); rc = OI_HIP_ERROR; break; }
      }
      std::swap(old, cur);  // the settled field is the input of the next pass
    }
  } while (0);
  if (rc == OI_HIP_ERROR && !*oi::last_error()) oi::set_error("remspot: HIP failure: %s", hipGetErrorString(hipGetLastError()));
  if (cur) (void)oi::dev_free(cur);
  if (cuts) (void)oi::dev_free(cuts);
  if (changed) (void)oi::dev_free(changed);
  if (h_changed) (void)hipHostFree(h_changed);
  if (s) (void)hipStreamDestroy(s);
  if (rc != OI_OK) { if (old) (void)oi::dev_free(old); return rc; }
  *out_dev = old;
  *out_phase = old + plane;
  return OI_OK;
}

static oi_status create_whole(oi_tort** out, const uint8_t* phase, bool on_device, int nx, int ny, int nz, int phase_id,
                              int dir, double vlo, double vhi, const oi_tort_opts* opts) {
  if (opts && opts->remspot_passes > 0 && phase && nx > 0 && ny > 0 && nz > 0) {
    uint8_t *priv = nullptr, *priv_phase = nullptr;
    OIC(remspot_private_copy(nullptr, phase, on_device, nx, ny, nz, 0, nz, opts->remspot_passes, opts->remspot_box_size,
                             opts->device, &priv, &priv_phase));
    const oi_status st = create_common(out, nullptr, priv_phase, true, nx, ny, nz, 0, nz, phase_id, dir, vlo, vhi, opts);
    (void)oi::dev_free(priv);
    return st;
  }
  return create_common(out, nullptr, phase, on_device, nx, ny, nz, 0, nz, phase_id, dir, vlo, vhi, opts);
}

extern "C" oi_status oi_tort_create(oi_tort** out, const uint8_t* phase, int nx, int ny, int nz, int phase_id,
                                    int dir, double vlo, double vhi, const oi_tort_opts* opts) {
  return create_whole(out, phase, false, nx, ny, nz, phase_id, dir, vlo, vhi, opts);
}

extern "C" oi_status oi_tort_create_device(oi_tort** out, const uint8_t* phase_dev, int nx, int ny, int nz,
                                           int phase_id, int dir, double vlo, double vhi,
                                           const oi_tort_opts* opts) {
  return create_whole(out, phase_dev, true, nx, ny, nz, phase_id, dir, vlo, vhi, opts);
}

extern "C" oi_status oi_tort_create_slab(oi_tort** out, oi_comm* comm, const uint8_t* phase, int on_device,
                                         int nx, int ny, int nz_global, int z0, int nz_local, int phase_id,
                                         int dir, double vlo, double vhi, const oi_tort_opts* opts) {
  if (opts && opts->remspot_passes > 0 && phase && nx > 0 && ny > 0 && nz_local > 0 && z0 >= 0 && z0 + nz_local <= nz_global) {
    // every rank filters its slab; boxes that straddle a slab boundary relax across it (collective)
    uint8_t *priv = nullptr, *priv_phase = nullptr;
    OIC(remspot_private_copy(comm, phase, on_device != 0, nx, ny, nz_global, z0, nz_local, opts->remspot_passes,
                             opts->remspot_box_size, opts->device, &priv, &priv_phase));
    const oi_status st = create_common(out, comm, priv_phase, true, nx, ny, nz_global, z0, nz_local, phase_id, dir, vlo, vhi, opts);
    (void)oi::dev_free(priv);
    return st;
  }
  return create_common(out, comm, phase, on_device != 0, nx, ny, nz_global, z0, nz_local, phase_id, dir, vlo, vhi,
                       opts);
}

// ---- CG --------------------------------------------------------------------------------
static oi_status reduce_to(oi_tort* h, int nblocks, int ncomp, double* dst, bool check, int iter_done) {
  if (nblocks < 0) { oi::set_error("launch grid exceeds the handle's partial-sum capacity (%d blocks)", h->g.tune.max_blocks); return OI_BAD_ARG; }
  const bool multi = h->comm != nullptr;
  {
    PROF(K_FINALIZE);
    launch_finalize(h->partials, nblocks, ncomp, dst, h->ctrl, (check && !multi) ? 1 : 0, iter_done, h->stream);
  }
  if (multi) {
    OIC(h->comm->allreduce_sum_f64(dst, ncomp, h->stream));
    if (check) launch_check_done(dst, h->ctrl, iter_done, h->stream);
  }
  return OI_OK;
}

extern "C" oi_status oi_tort_cg_begin(oi_tort* h) {
  if (!h) { oi::set_error("%s: null handle or argument out of range", __func__); return OI_BAD_ARG; }
  if (!h->percolates || !h->vectors_ready) return OI_NO_PERCOLATION;
  HIPC(hipSetDevice(h->device));
  const Geom& g = h->g;
  hipStream_t s = h->stream;
  Ctrl c;
  c.done = 0; c.pad = 0;
  c.tol2 = (h->opts.eps * h->bnorm) * (h->opts.eps * h->bnorm);
  *h->h_ctrl = c;
  HIPC(hipMemcpyAsync(h->ctrl, h->h_ctrl, sizeof(Ctrl), hipMemcpyHostToDevice, s));
  HIPC(hipMemsetAsync(h->hist, 0, sizeof(double) * H_STRIDE * h->hist_cap, s));
  HIPC(hipMemsetAsync(h->r, 0, sizeof(double) * h->nvec_cg, s));
  if (h->q) HIPC(hipMemsetAsync(h->q, 0, sizeof(double) * h->nvec_cg, s));
  HIPC(hipMemsetAsync(h->p[0], 0, h->p_elem * h->nvec_cg, s));
  if (h->p[1]) HIPC(hipMemsetAsync(h->p[1], 0, h->p_elem * h->nvec_cg, s));
  { PROF(K_INIT); launch_init_field(g, h->active, h->x, s); }
  int nb;
  if (h->rot) HIPC(hipMemsetAsync(h->r2, 0, sizeof(double) * h->nvec_cg, s));
  h->xskip = h->cg1 ? g.tune.cg1_xskip : 0;
  h->xpair = (h->rot && g.tune.mg_xpair && !g.tune.mg_xfirst && oi::mg_post_descr(g, h->lineoff, h->plane_first_dev, h->seg)) ? 1 : 0;
  if (h->cg1) {
    // kernel 0 reads buffer 1: r_{-1} := r_0 with alpha_{-1} = 0, q_{-1} = p_{-1} = 0
    HIPC(hipMemsetAsync(h->r2, 0, sizeof(double) * h->nvec_cg, s));
    HIPC(hipMemsetAsync(h->q2, 0, sizeof(double) * h->nvec_cg, s));
    HIPC(hipMemsetAsync(h->rec, 0, sizeof(double) * kRec1Stride * h->hist_cap, s));
    { PROF(K_RESID0); nb = launch_residual0(g, h->code, h->x, h->r2, nullptr, h->jacobi, h->partials, h->lineoff, s, g.tune.resid0_bits ? h->active : nullptr); }
    OIC(exchange_cg_vec(h, h->r2, sizeof(double), s));
  } else {
    { PROF(K_RESID0); nb = launch_residual0(g, h->code, h->x, h->r, (h->fused || h->mg) ? nullptr : (double*)h->p[0], h->jacobi, h->partials, h->lineoff, s, g.tune.resid0_bits ? h->active : nullptr); }
    // hist[k=0] = {-, rr_0, rz_0}; converged before the first iteration -> done = 1
    OIC(reduce_to(h, nb, 2, h->hist + H_RR, true, 1));
    if (h->fused || h->mg) OIC(exchange_cg_vec(h, h->r, sizeof(double), s));
  }
  h->iters_issued = 0;
  h->iters_done = 0;
  h->cg_active = true;
  h->cg_ever = true;
  return OI_OK;
}

static oi_status cg_one_iteration(oi_tort* h, int k) {
  const Geom& g = h->g;
  hipStream_t s = h->stream;
  int nb;
  if (h->cg1) {
    const int in = (k + 1) & 1, out = k & 1;
    double* rb[2] = {h->r, h->r2};
    double* qb[2] = {h->q, h->q2};
    double* rk = h->rec + (size_t)kRec1Stride * k;
    const int pf = h->p_f32 ? 1 : 0, jac = h->jacobi ? 1 : 0;
    if (!h->comm) {
      {
        PROF(K_CG1);
        nb = launch_cg1(g, h->code, h->x, rb[in], rb[out], qb[in], qb[out], h->p[in], h->p[out], pf, jac, h->rec, k,
                        h->partials, h->ctrl, 0, 0, 0, h->lineoff, h->plane_first_dev, h->seg, s);
      }
      if (nb < 0) { oi::set_error("launch grid exceeds the handle's partial-sum capacity (%d blocks)", g.tune.max_blocks); return OI_BAD_ARG; }
      // (alpha_k, beta_{k+1} and the stop test are formed in the prologue of kernel k + 1; cg_iterate closes the last record)
      { PROF(K_FINALIZE); launch_finalize1(h->partials, nb, rk, h->ctrl, k, 0, s, g.tune.cg1_fin_sliced ? h->fin_scratch : nullptr); }
    } else {
      // slab with neighbours: the two boundary chunks and the interior chunks are two launches on two streams that run
      // SIDE BY SIDE (neither depends on the other; back to back on one stream each pays its own ramp-up and tail: 1.18
      // against 1.03 ms at 128 planes).  The boundary chunks' q planes travel on the comm stream (RCCL point-to-point over
      // xGMI) as soon as THAT launch is done, while the interior chunks still march.
      int nb_b, nb_i;
      {
        PROF(K_CG1);  // one scope over both launches: the time until both have finished
        HIPC(hipEventRecord(h->ev_start, s));
        HIPC(hipStreamWaitEvent(h->aux_stream, h->ev_start, 0));
        nb_b = launch_cg1(g, h->code, h->x, rb[in], rb[out], qb[in], qb[out], h->p[in], h->p[out], pf, jac, h->rec, k,
                          h->partials, h->ctrl, 1, 1, 0, h->lineoff, h->plane_first_dev, h->seg, s);
        HIPC(hipEventRecord(h->ev_boundary, s));
        nb_i = launch_cg1(g, h->code, h->x, rb[in], rb[out], qb[in], qb[out], h->p[in], h->p[out], pf, jac, h->rec, k,
                          h->partials, h->ctrl, 1, 2, nb_b < 0 ? 0 : nb_b, h->lineoff, h->plane_first_dev, h->seg, h->aux_stream);
        HIPC(hipEventRecord(h->ev_interior, h->aux_stream));
        // (a transport whose exchange may not overlap its own collectives keeps everything on the solver stream)
        hipStream_t xs = h->comm->exchange_overlaps ? h->comm_stream : s;
        HIPC(hipStreamWaitEvent(xs, h->ev_boundary, 0));
        OIC(exchange_cg_vec(h, qb[out], sizeof(double), xs));  // the ONE plane exchange per iteration
        HIPC(hipEventRecord(h->ev_exchanged, xs));
        HIPC(hipStreamWaitEvent(s, h->ev_interior, 0));
      }
      if (nb_b < 0 || nb_i < 0) { oi::set_error("launch grid exceeds the handle's partial-sum capacity (%d blocks)", g.tune.max_blocks); return OI_BAD_ARG; }
      { PROF(K_FINALIZE); launch_finalize1(h->partials, nb_b + nb_i, rk, h->ctrl, k, 0, s, g.tune.cg1_fin_sliced ? h->fin_scratch : nullptr); }
      OIC(h->comm->allreduce_sum_f64(rk, 5, s));
      HIPC(hipStreamWaitEvent(s, h->ev_exchanged, 0));  // the next kernel reads the ghost planes of q
    }
  } else if (h->rot) {
    // rotated multigrid two-kernel iteration.  A: r_k = r_{k-1} - alpha_{k-1} q_{k-1} (k = 0: a copy of r_0), ||r_k||^2 and the
    // stop test, pre-smoother + restriction; B: the rest of the cycle, z_k and r_k.z_k; C: p_k, q_k, p_k.q_k, x += alpha p
    void* p_in = h->p[(k + 1) & 1];
    void* p_out = h->p[k & 1];
    const double* r_in = (k & 1) ? h->r2 : h->r;
    double* r_out = (k & 1) ? h->r : h->r2;
    MgCoarse cref{};
    float* bcoarse = nullptr;
    double omega = 0.0;
    if (!mg_coarse_target(h->mg, &cref, &bcoarse, &omega, s)) return OI_HIP_ERROR;
    {
      PROF(K_UPDATE_R);
      nb = launch_mg_pre_update(g, h->code, r_in, r_out, h->q, h->hist, k, cref, bcoarse, omega, h->partials, h->ctrl, h->lineoff,
                                h->plane_first_dev, h->seg, s);
    }
    OIC(reduce_to(h, nb, 2, h->hist + H_STRIDE * k + H_RR, true, k + 1));
    // x += alpha_{k-1} p_{k-1}: in the post-smoother (instruction-bound, it carries the bytes for less) when that runs as k_cg1b FORM 3,
    // else in the first kernel as in the unrotated iteration
    // (round 4: pairwise there -- launch k even adds alpha_{k-2} p_{k-2}, still in p_out, and alpha_{k-1} p_{k-1}; odd launches leave x alone)
    const bool xpost = h->xpair || (!g.tune.mg_xfirst && mg_post_descr(g, h->lineoff, h->plane_first_dev, h->seg));
    const MgXUpdate xu{h->x, p_in, h->p_f32 ? 1 : 0, h->hist, k, h->xpair ? p_out : nullptr};
    {
      PROF(K_MG);
      // (profiling: the fine post-smoother also on its own -- K_MG's span minus K_MG_POST's is the coarse levels)
      ProfRec post{K_MG_POST, nullptr, nullptr};
      if (h->opts.profile) { post.a = prof_event(h); post.b = prof_event(h); }
      nb = mg_vcycle_after_restriction(h->mg, g, h->code, r_out, h->z, h->partials, h->ctrl, h->lineoff, h->plane_first_dev, h->seg, s, xpost ? &xu : nullptr, post.a);
      if (post.b) { (void)hipEventRecord(post.b, s); h->prof_pending.push_back(post); }
    }
    if (nb < 0) return OI_RCCL_ERROR;
    OIC(reduce_to(h, nb, 1, h->hist + H_STRIDE * k + H_RZ, false, 0));
    OIC(exchange_cg_vec(h, h->z, sizeof(double), s));
    {
      PROF(K_FUSED_A);
      nb = launch_fused_a(g, h->code, h->x, h->z, p_in, p_out, h->p_f32 ? 1 : 0, h->q, 0, h->hist, k, h->partials, h->ctrl, h->lineoff,
                          h->plane_first_dev, h->seg, s, xpost ? 1 : 0);
    }
    OIC(reduce_to(h, nb, 1, h->hist + H_STRIDE * k + H_PQ, false, 0));
    OIC(exchange_cg_vec(h, h->q, sizeof(double), s));  // (kernel A of the next iteration forms r on the ghost planes from r and q there)
  } else if (h->fused) {
    void* p_in = h->p[(k + 1) & 1];
    void* p_out = h->p[k & 1];
    const double* zsrc = h->r;
    if (h->mg) {  // preconditioned residual from the multigrid V-cycle, and r.z into hist[k]
      { PROF(K_MG); nb = mg_vcycle(h->mg, g, h->code, h->r, h->q, h->z, h->partials, h->ctrl, h->mg_tiled, h->lineoff, s, h->plane_first_dev, h->seg); }
      if (nb < 0) return OI_RCCL_ERROR;
      OIC(reduce_to(h, nb, 1, h->hist + H_STRIDE * k + H_RZ, false, 0));
      OIC(exchange_cg_vec(h, h->z, sizeof(double), s));
      zsrc = h->z;
    }
    {
      PROF(K_FUSED_A);
      nb = launch_fused_a(g, h->code, h->x, zsrc, p_in, p_out, h->p_f32 ? 1 : 0, h->q, h->jacobi, h->hist, k, h->partials,
                          h->ctrl, h->lineoff, h->plane_first_dev, h->seg, s);
    }
    OIC(reduce_to(h, nb, 1, h->hist + H_STRIDE * k + H_PQ, false, 0));
    {
      PROF(K_UPDATE_R);
      // compacted: one flat stream over the stored lines of the owned planes
      const long long first = h->lineoff ? h->plane_first[1] : 0;
      const long long nslots = h->lineoff ? h->plane_first[g.nzl + 1] - h->plane_first[1] : -1;
      nb = launch_update_r(g, h->code, h->r, h->q, h->jacobi, h->hist, k, h->partials, h->ctrl, first, nslots, s);
    }
    OIC(reduce_to(h, nb, 2, h->hist + H_STRIDE * (k + 1) + H_RR, true, k + 2));
    OIC(exchange_cg_vec(h, h->r, sizeof(double), s));
  } else if (h->mg) {
    // PCG with the multigrid V-cycle: z = M^-1 r, r.z, p = z + beta p, q = A p, x/r update
    double* p = (double*)h->p[0];
    { PROF(K_MG); nb = mg_vcycle(h->mg, g, h->code, h->r, h->q, h->z, h->partials, h->ctrl, h->mg_tiled, nullptr, s); }
    if (nb < 0) return OI_RCCL_ERROR;
    OIC(reduce_to(h, nb, 1, h->hist + H_STRIDE * k + H_RZ, false, 0));
    if (k > 0) {
      PROF(K_UPDATE_P);
      launch_update_p(g, h->code, h->z, p, 0, h->hist, k, h->ctrl, s);
    } else {
      HIPC(hipMemcpyAsync(p, h->z, sizeof(double) * g.nvec, hipMemcpyDeviceToDevice, s));
    }
    OIC(exchange_planes(h, p, sizeof(double) * g.plane));
    { PROF(K_APPLY); nb = launch_apply_tiled(g, h->code, p, h->q, h->partials, h->ctrl, s); }
    OIC(reduce_to(h, nb, 1, h->hist + H_STRIDE * k + H_PQ, false, 0));
    {
      PROF(K_UPDATE_XR);
      nb = launch_update_xr(g, h->code, h->x, h->r, p, h->q, 0, h->hist, k, h->partials, h->ctrl, s);
    }
    OIC(reduce_to(h, nb, 2, h->hist + H_STRIDE * (k + 1) + H_RR, true, k + 2));
    OIC(exchange_planes(h, h->r, sizeof(double) * g.plane));
  } else if (h->lean) {
    // q-free iteration: p = z + beta p; p.Ap (nothing stored); x += alpha p, r -= alpha A p (A p recomputed)
    double* p = (double*)h->p[0];
    if (k > 0) {
      PROF(K_UPDATE_P);
      launch_update_p(g, h->code, h->r, p, h->jacobi, h->hist, k, h->ctrl, s);
    }
    OIC(exchange_planes(h, p, sizeof(double) * g.plane));
    { PROF(K_APPLY); nb = launch_lean_dot(g, h->code, p, h->partials, h->ctrl, s); }
    OIC(reduce_to(h, nb, 1, h->hist + H_STRIDE * k + H_PQ, false, 0));
    { PROF(K_UPDATE_XR); nb = launch_lean_update(g, h->code, p, h->x, h->r, h->jacobi, h->hist, k, h->partials, h->ctrl, s); }
    OIC(reduce_to(h, nb, 2, h->hist + H_STRIDE * (k + 1) + H_RR, true, k + 2));
  } else {
    if (k > 0) {
      PROF(K_UPDATE_P);
      launch_update_p(g, h->code, h->r, (double*)h->p[0], h->jacobi, h->hist, k, h->ctrl, s);
    }
    OIC(exchange_planes(h, h->p[0], sizeof(double) * g.plane));
    { PROF(K_APPLY); nb = launch_apply(g, h->code, (double*)h->p[0], h->q, h->partials, h->ctrl, s); }
    OIC(reduce_to(h, nb, 1, h->hist + H_STRIDE * k + H_PQ, false, 0));
    {
      PROF(K_UPDATE_XR);
      nb = launch_update_xr(g, h->code, h->x, h->r, (double*)h->p[0], h->q, h->jacobi, h->hist, k, h->partials, h->ctrl,
                            s);
    }
    OIC(reduce_to(h, nb, 2, h->hist + H_STRIDE * (k + 1) + H_RR, true, k + 2));
  }
  return OI_OK;
}

extern "C" oi_status oi_tort_cg_iterate(oi_tort* h, int n) {
  if (!h || n < 0) { oi::set_error("%s: null handle or argument out of range", __func__); return OI_BAD_ARG; }
  if (!h->cg_active) { oi::set_error("%s: no CG run is open (oi_tort_cg_begin first)", __func__); return OI_BAD_ARG; }
  HIPC(hipSetDevice(h->device));
  const int limit = h->maxiter + ((h->cg1 || h->rot) ? 1 : 0);  // launch k of the single-kernel / rotated variants completes iteration k - 1
  int issued = 0;
  for (int i = 0; i < n; ++i) {
    if (h->iters_issued >= limit) break;
    OIC(cg_one_iteration(h, h->iters_issued));
    ++h->iters_issued;
    ++issued;
  }
  // single-kernel iteration: records are closed (alpha, beta, stop test) by the NEXT kernel's prologue; close the last one
  // of this batch here so that the host sees the flag and cg_finish finds alpha
  if (h->cg1 && issued > 0)
    launch_scalars1(h->rec + (size_t)kRec1Stride * (h->iters_issued - 1), h->ctrl, h->iters_issued - 1, h->stream);
  return OI_OK;
}

// reads the device control block; updates iters_done / converged
static oi_status poll_ctrl(oi_tort* h) {
  HIPC(hipMemcpyAsync(h->h_ctrl, h->ctrl, sizeof(Ctrl), hipMemcpyDeviceToHost, h->stream));
  HIPC(hipStreamSynchronize(h->stream));
  return OI_OK;
}

extern "C" oi_status oi_tort_cg_finish(oi_tort* h) {
  if (!h || !h->cg_active) { oi::set_error("%s: no CG run is open (oi_tort_cg_begin first)", __func__); return OI_BAD_ARG; }
  HIPC(hipSetDevice(h->device));
  OIC(poll_ctrl(h));
  const int done = h->h_ctrl->done;
  // iterations whose x/r updates are in effect
  const int K = done ? done - 1 : ((h->cg1 || h->rot) ? std::max(0, h->iters_issued - 1) : h->iters_issued);
  if (h->cg1 && h->xskip && (K & 1) && h->iters_issued > 0) {
    // pairwise x updates: an odd final state still lacks alpha_{K-1} p_{K-1}
    PROF(K_AXPY);
    launch_axpy_ptr(h->g, h->x, h->p[(K - 1) & 1], h->p_f32 ? 1 : 0, h->rec + (size_t)kRec1Stride * (K - 1) + kRec1Alpha,
                    h->lineoff, h->stream);
  }
  // (rotated variant: stopped at convergence -- kernel A of launch K raised the flag, kernels B and C of that launch returned at
  // once -- x lacks alpha_{K-1} p_{K-1} like the plain fused variant; stopped for another reason, K = launches - 1 and kernel C
  // of the last launch has already added alpha_{K-1} p_{K-1}: x = x_K as it is)
  if (h->rot && h->xpair) {
    // pairwise x update in the post-smoother: after launch m, x = x_e with e the largest even number <= m.  Stopped at convergence
    // (kernels B and C of launch K returned at once): m = K - 1; stopped otherwise: m = K.  What x still lacks, oldest term first
    // (the same two dependent additions the kernel makes); p_j lives in p[j & 1] and neither buffer has been overwritten since
    const int m = done ? K - 1 : K;
    const int e = m < 0 ? 0 : (m & ~1);
    // (two terms pending: one pass over x)
    if (K - e == 2) {
      PROF(K_AXPY);
      launch_axpy_hist(h->g, h->x, h->p[e & 1], h->p_f32 ? 1 : 0, h->hist, e, h->lineoff, h->stream, h->p[(e + 1) & 1]);
    } else {
      for (int j = e; j < K; ++j) {
        PROF(K_AXPY);
        launch_axpy_hist(h->g, h->x, h->p[j & 1], h->p_f32 ? 1 : 0, h->hist, j, h->lineoff, h->stream);
      }
    }
  } else if (h->fused && !h->cg1 && K > 0 && !(h->rot && !done)) {
    // the fused variant applies x += alpha_{k} p_{k} inside iteration k+1: flush the last one
    PROF(K_AXPY);
    launch_axpy_hist(h->g, h->x, h->p[(K - 1) & 1], h->p_f32 ? 1 : 0, h->hist, K - 1, h->lineoff, h->stream);
  }
  double rr = 0.0;
  const double* rr_src = h->cg1 ? h->rec + (size_t)kRec1Stride * K : h->hist + H_STRIDE * K + H_RR;
  HIPC(hipMemcpyAsync(h->h_scal, rr_src, sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPC(hipStreamSynchronize(h->stream));
  rr = (h->cg1 && h->iters_issued == 0) ? std::numeric_limits<double>::quiet_NaN() : h->h_scal[0];
  h->iters_done = K;
  h->st.iterations = K;
  h->st.final_relres = h->bnorm > 0.0 ? std::sqrt(rr) / h->bnorm : std::sqrt(rr);
  // converged <=> finite and 0 <= relres <= eps  (TortuosityHypre.cpp:685-688)
  h->st.converged = done != 0 && std::isfinite(h->st.final_relres) && h->st.final_relres <= h->opts.eps;
  h->cg_active = false;
  if (h->opts.profile) prof_harvest(h);
  return OI_OK;
}

extern "C" oi_status oi_tort_set_eps(oi_tort* h, double eps) {
  if (!h || !(eps > 0.0)) { oi::set_error("%s: null handle or argument out of range", __func__); return OI_BAD_ARG; }
  if (h->cg_active) { oi::set_error("%s: a CG run is open (oi_tort_cg_finish first)", __func__); return OI_BAD_ARG; }
  h->opts.eps = eps;
  return OI_OK;
}

extern "C" oi_status oi_tort_sync(oi_tort* h) {
  if (!h) { oi::set_error("%s: null handle or argument out of range", __func__); return OI_BAD_ARG; }
  HIPC(hipSetDevice(h->device));
  HIPC(hipStreamSynchronize(h->stream));
  if (h->opts.profile) prof_harvest(h);
  return OI_OK;
}

extern "C" oi_status oi_tort_cg_state(oi_tort* h, int* iterations_done, double* relres, int* converged) {
  if (!h || !h->vectors_ready) { oi::set_error("%s: null handle or argument out of range", __func__); return OI_BAD_ARG; }
  if (!h->cg_ever) {  // no run yet: the history and the flag on the device are not initialised
    if (iterations_done) *iterations_done = 0;
    if (relres) *relres = 0.0;
    if (converged) *converged = 0;
    return OI_OK;
  }
  HIPC(hipSetDevice(h->device));
  OIC(poll_ctrl(h));
  const int done = h->h_ctrl->done;
  const int K = done ? done - 1 : ((h->cg1 || h->rot) ? std::max(0, h->iters_issued - 1) : h->iters_issued);
  const double* rr_src = h->cg1 ? h->rec + (size_t)kRec1Stride * K : h->hist + H_STRIDE * K + H_RR;
  HIPC(hipMemcpyAsync(h->h_scal, rr_src, sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPC(hipStreamSynchronize(h->stream));
  if (iterations_done) *iterations_done = K;
  if (relres) *relres = h->bnorm > 0.0 ? std::sqrt(h->h_scal[0]) / h->bnorm : std::sqrt(h->h_scal[0]);
  if (converged) *converged = done != 0;
  return OI_OK;
}

// Launches iterations of the open CG run until the device's stop test fires (or maxiter / `max_launches` launches): the loop
// value() runs between cg_begin and cg_finish, exposed so that a host (bench.py) times exactly what the library does.
//   * plain / Jacobi CG (thousands of cheap iterations whose kernels return at once after the flag is up): batches of 32 with the
//     flag of the PREVIOUS batch looked at while the next one runs -- the device never waits for the host;
//   * multigrid-preconditioned CG (tens of iterations of ~12 ms at 1024^3; the coarse levels of a launch issued after convergence
//     still run, 1.7 ms each): the host reads the flag and ||r||^2 after every batch (one 50 us bubble per batch), estimates the
//     contraction per iteration from the last two readings and issues only as many launches as the residual still needs -- the
//     solve ends with one launch that finds the flag up instead of up to eight.
extern "C" oi_status oi_tort_cg_run(oi_tort* h, int max_launches, int* launches_issued) {
  if (!h) { oi::set_error("%s: null handle", __func__); return OI_BAD_ARG; }
  if (!h->cg_active) { oi::set_error("%s: no CG run is open (oi_tort_cg_begin first)", __func__); return OI_BAD_ARG; }
  HIPC(hipSetDevice(h->device));
  const int limit = h->maxiter + ((h->cg1 || h->rot) ? 1 : 0);
  const int first_issued = h->iters_issued;
  auto budget_left = [&] { return max_launches <= 0 || h->iters_issued - first_issued < max_launches; };
  const int batch = h->opts.check_every > 0 ? h->opts.check_every : (h->mg ? 4 : 32);
  oi_status rc = poll_ctrl(h);  // converged before iterating?
  bool stop = rc == OI_OK && h->h_ctrl->done;
  if (h->mg) {
    const double tol2 = (h->opts.eps * h->bnorm) * (h->opts.eps * h->bnorm);
    int next = batch;
    constexpr int NH = 8;  // the last residuals seen: (iteration, log ||r||^2)
    int hk[NH]; double hl[NH]; int nh = 0;
    while (rc == OI_OK && !stop && h->iters_issued < limit && budget_left()) {
      int n = std::max(1, std::min(next, batch));
      if (max_launches > 0) n = std::min(n, max_launches - (h->iters_issued - first_issued));
      rc = oi_tort_cg_iterate(h, n);
      if (rc != OI_OK) break;
      rc = poll_ctrl(h);
      if (rc != OI_OK) break;
      if (h->h_ctrl->done) { stop = true; break; }
      // ||r_K||^2 of the last iteration whose residual exists, and how many more the contraction so far asks for
      const int K = h->rot ? std::max(0, h->iters_issued - 1) : h->iters_issued;
      if (hipMemcpyAsync(h->h_scal, h->hist + H_STRIDE * K + H_RR, sizeof(double), hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
          hipStreamSynchronize(h->stream) != hipSuccess) { rc = OI_HIP_ERROR; break; }
      const double rr = h->h_scal[0];
      next = batch;
      if (rr > 0.0 && tol2 > 0.0 && rr > tol2) {
        const double lr = std::log(rr);
        // two estimates of the iterations still needed: from the contraction of the last step and from that of the last four to five
        // iterations.  CG's contraction is not uniform here (0.65 to 0.9 per iteration, slowing down as the solve goes on:
        // profiles/r05/mg_residual_history.txt): the window runs ahead of the truth while the contraction decays, one slow step runs behind
        // it (a step of 0.92 after five of 0.75 once sent a whole batch out one iteration before the end: 24 launches for 21 iterations
        // along Y).  A poll costs ~0.2 ms of idle device, a launch issued after convergence ~0.5 ms (its hundred kernels return at their
        // first instruction): one by one as soon as EITHER estimate says the end is two iterations away or BOTH say it is within a batch
        // and a half.  (1024^3 spheres: X 23 launches for 22 iterations, Y 22 for 21, Z 24 for 23; 0.310 / 0.299 / 0.322 s.)
        int ref = -1, last = -1;
        for (int q = 0; q < nh; ++q) {
          if (K - hk[q] >= 1 && K - hk[q] <= 5 && (ref < 0 || hk[q] < hk[ref])) ref = q;
          if (K - hk[q] >= 1 && (last < 0 || hk[q] > hk[last])) last = q;
        }
        auto need_from = [&](int q) {
          if (q < 0 || !(hl[q] > lr)) return 1e30;  // (no contraction seen: no estimate)
          return (std::log(tol2) - lr) / ((lr - hl[q]) / (double)(K - hk[q]));
        };
        const double need_w = need_from(ref), need_1 = need_from(last);
        if (need_w < 1e29 || need_1 < 1e29) {
          const double lo = std::min(need_w, need_1), hi = std::max(need_w < 1e29 ? need_w : need_1, need_1 < 1e29 ? need_1 : need_w);
          next = (lo <= 2.0 || hi <= (double)batch + 2.0) ? 1 : batch;
        }
        if (nh < NH) { hk[nh] = K; hl[nh] = lr; ++nh; }
        else { for (int q = 1; q < NH; ++q) { hk[q - 1] = hk[q]; hl[q - 1] = hl[q]; } hk[NH - 1] = K; hl[NH - 1] = lr; }
      }
      if (h->opts.verbose > 1) fprintf(stderr, "[openimpala_amd] %d launches issued, ||r||^2 = %.3e, next batch %d\n", h->iters_issued, rr, std::min(next, batch));
    }
  } else {
    hipEvent_t ev[2] = {nullptr, nullptr};
    Ctrl* land = nullptr;  // pinned landing pads for the two in-flight polls
    if (hipEventCreateWithFlags(&ev[0], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&ev[1], hipEventDisableTiming) != hipSuccess ||
        hipHostMalloc((void**)&land, 2 * sizeof(Ctrl)) != hipSuccess) {
      rc = OI_HIP_ERROR;
    } else {
      land[0].done = land[1].done = 0;
      int nb = 0;
      while (rc == OI_OK && !stop && h->iters_issued < limit && budget_left()) {
        int n = batch;
        if (max_launches > 0) n = std::min(n, max_launches - (h->iters_issued - first_issued));
        rc = oi_tort_cg_iterate(h, n);
        if (rc != OI_OK) break;
        const int slot = nb & 1;
        if (hipMemcpyAsync(&land[slot], h->ctrl, sizeof(Ctrl), hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
            hipEventRecord(ev[slot], h->stream) != hipSuccess) { rc = OI_HIP_ERROR; break; }
        if (nb > 0) {  // look at the PREVIOUS batch while this one runs
          if (hipEventSynchronize(ev[slot ^ 1]) != hipSuccess) { rc = OI_HIP_ERROR; break; }
          if (land[slot ^ 1].done) stop = true;
        }
        ++nb;
        if (h->opts.verbose > 1 && (nb % 16) == 0) fprintf(stderr, "[openimpala_amd] issued %d CG iterations\n", h->iters_issued);
      }
    }
    if (ev[0]) (void)hipEventDestroy(ev[0]);
    if (ev[1]) (void)hipEventDestroy(ev[1]);
    if (land) (void)hipHostFree(land);
    if (rc == OI_HIP_ERROR && !*oi::last_error()) oi::set_error("%s: HIP failure: %s", __func__, hipGetErrorString(hipGetLastError()));
  }
  if (launches_issued) *launches_issued = h->iters_issued - first_issued;
  return rc;
}

static oi_status solve(oi_tort* h) {
  const double t0 = now_s();
  OIC(oi_tort_cg_begin(h));
  OIC(oi_tort_cg_run(h, 0, nullptr));
  OIC(oi_tort_cg_finish(h));
  h->st.t_solve_s = now_s() - t0;
  return OI_OK;
}

static oi_status fluxes(oi_tort* h) {
  const double t0 = now_s();
  OIC(exchange_planes(h, h->x, sizeof(double) * h->g.plane));
  { PROF(K_FLUX); launch_flux(h->g, h->active, h->x, h->partials, h->scal, h->stream); }
  if (h->comm) OIC(h->comm->allreduce_sum_f64(h->scal, 2, h->stream));
  HIPC(hipMemcpyAsync(h->h_scal, h->scal, 2 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPC(hipStreamSynchronize(h->stream));
  h->st.flux_in = h->h_scal[0] * 1.0;  // face_area_element = dx*dx = 1 (TortuosityHypre.cpp:1123-1133)
  h->st.flux_out = h->h_scal[1] * 1.0;
  h->st.t_flux_s = now_s() - t0;
  return OI_OK;
}

// TortuosityHypre::value (TortuosityHypre.cpp:761-891)
extern "C" oi_status oi_tort_value(oi_tort* h, int refresh, double* tau) {
  if (!h || !tau) { oi::set_error("%s: null handle or argument out of range", __func__); return OI_BAD_ARG; }
  const double nan = std::numeric_limits<double>::quiet_NaN();
  const double inf = std::numeric_limits<double>::infinity();
  *tau = nan;
  if (!h->percolates) {  // :764-777
    h->st.tau = nan;
    return OI_NO_PERCOLATION;
  }
  if (h->solved && !refresh) {
    *tau = h->st.tau;
    return h->value_status;
  }
  HIPC(hipSetDevice(h->device));
  OIC(solve(h));
  h->solved = true;
  const Geom& g = h->g;
  if (!h->st.converged) {  // :782-787
    if (h->opts.verbose >= 1)
      fprintf(stderr, "[openimpala_amd] WARNING: solver did not converge (%d iterations, relres %.3e)\n",
              h->st.iterations, h->st.final_relres);
    h->st.tau = nan;
    h->value_status = OI_NOT_CONVERGED;
    return h->value_status;
  }
  OIC(fluxes(h));
  constexpr double tiny = 1.e-15;    // tiny_flux_threshold, :64
  constexpr double flux_tol = 1.e-6;  // :794
  const double fin = std::abs(h->st.flux_in), fout = std::abs(h->st.flux_out);
  const double avg = 0.5 * (fin + fout);
  if (avg > tiny && std::abs(fin - fout) / avg > flux_tol) {  // :800-823
    if (h->opts.verbose >= 1) fprintf(stderr, "[openimpala_amd] WARNING: flux not conserved\n");
    h->st.tau = nan;
    h->value_status = OI_FLUX_NOT_CONSERVED;
    return h->value_status;
  }
  const double nd[3] = {(double)g.nx, (double)g.ny, (double)g.nz};
  const double L = nd[g.dir];  // ProbLength: the RealBox is [0,N] per axis (Diffusion.cpp:302-305)
  const double A = g.dir == 0 ? nd[1] * nd[2] : (g.dir == 1 ? nd[0] * nd[2] : nd[0] * nd[1]);
  const double grad = (g.vhi - g.vlo) / L;  // :841
  double value;
  h->st.deff = 0.0;
  if (avg < tiny) value = inf;                 // :846-852
  else if (std::abs(grad) < tiny) value = inf;  // :860-865
  else {
    const double deff = (avg / A) / std::abs(grad);  // :868
    h->st.deff = deff;
    value = std::abs(deff) < tiny ? inf : h->st.active_vf / deff;  // :869-876
  }
  h->st.tau = value;
  h->value_status = OI_OK;
  *tau = value;
  return OI_OK;
}

extern "C" oi_status oi_tort_get_stats(const oi_tort* h, oi_tort_stats* st) {
  if (!h || !st) { oi::set_error("%s: null handle or argument out of range", __func__); return OI_BAD_ARG; }
  *st = h->st;
  st->bytes_device = (double)h->bytes_device;
  return OI_OK;
}

extern "C" oi_status oi_tort_get_solution(oi_tort* h, double* phi) {
  if (!h || !phi) { oi::set_error("%s: null handle or argument out of range", __func__); return OI_BAD_ARG; }
  if (!h->vectors_ready) return OI_NO_PERCOLATION;
  HIPC(hipSetDevice(h->device));
  const Geom& g = h->g;
  const size_t n = (size_t)g.nx * g.ny * g.nzl;
  double* dense = nullptr;
  HIPC(oi::raw_malloc((void**)&dense, n * sizeof(double)));
  launch_pitch_copy_out(g, h->x, dense, h->stream);
  hipError_t e = hipMemcpyAsync(phi, dense, n * sizeof(double), hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  (void)oi::dev_free(dense);
  HIPC(e);
  return OI_OK;
}

extern "C" oi_status oi_tort_get_mask(oi_tort* h, uint8_t* mask) {
  if (!h || !mask) { oi::set_error("%s: null handle or argument out of range", __func__); return OI_BAD_ARG; }
  HIPC(hipSetDevice(h->device));
  const Geom& g = h->g;
  const size_t n = (size_t)g.nx * g.ny * g.nzl;
  uint8_t* dense = nullptr;
  HIPC(oi::raw_malloc((void**)&dense, n));
  launch_unpack_mask(g, h->active, dense, h->stream);
  hipError_t e = hipMemcpyAsync(mask, dense, n, hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  (void)oi::dev_free(dense);
  HIPC(e);
  return OI_OK;
}

extern "C" oi_status oi_tort_debug_get_codes(oi_tort* h, uint8_t* codes) {
  if (!h || !codes) { oi::set_error("%s: null handle or argument out of range", __func__); return OI_BAD_ARG; }
  if (!h->vectors_ready) return OI_NO_PERCOLATION;
  HIPC(hipSetDevice(h->device));
  const Geom& g = h->g;
  const size_t n = (size_t)g.nx * g.ny * g.nzl;
  uint8_t* dense = nullptr;
  HIPC(oi::raw_malloc((void**)&dense, n));
  launch_codes_out(g, h->code, dense, h->stream);
  hipError_t e = hipMemcpyAsync(codes, dense, n, hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  (void)oi::dev_free(dense);
  HIPC(e);
  return OI_OK;
}

// q = A v with either stencil kernel; uses r / q as scratch, so only outside a solve
extern "C" oi_status oi_tort_debug_apply(oi_tort* h, const double* v, double* q, int variant) {
  if (!h || !v || !q) { oi::set_error("%s: null handle or argument out of range", __func__); return OI_BAD_ARG; }
  if (!h->vectors_ready) return OI_NO_PERCOLATION;
  if (h->cg_active) { oi::set_error("%s: a CG run is open (oi_tort_cg_finish first)", __func__); return OI_BAD_ARG; }
  if (!h->q) { oi::set_error("the capacity mode keeps no A p vector: use another kernel variant for the stand-alone apply"); return OI_BAD_ARG; }
  HIPC(hipSetDevice(h->device));
  const Geom& g = h->g;
  // the compacted single-kernel layout keeps r / q as line lists: the stand-alone apply gets dense scratch
  struct Scratch { double *a = nullptr, *b = nullptr; ~Scratch() { if (a) (void)oi::dev_free(a); if (b) (void)oi::dev_free(b); } } scratch;
  double *vr = h->r, *vq = h->q;
  if (h->lineoff) {
    HIPC(oi::raw_malloc((void**)&scratch.a, sizeof(double) * g.nvec));
    HIPC(oi::raw_malloc((void**)&scratch.b, sizeof(double) * g.nvec));
    HIPC(hipMemsetAsync(scratch.a, 0, sizeof(double) * g.nvec, h->stream));
    HIPC(hipMemsetAsync(scratch.b, 0, sizeof(double) * g.nvec, h->stream));
    vr = scratch.a; vq = scratch.b;
  }
  const size_t n = (size_t)g.nx * g.ny * g.nzl;
  double* dense = nullptr;
  HIPC(oi::raw_malloc((void**)&dense, n * sizeof(double)));
  oi_status rc = OI_OK;
  do {
    if (hipMemcpyAsync(dense, v, n * sizeof(double), hipMemcpyHostToDevice, h->stream) != hipSuccess ||
        hipMemsetAsync(vr, 0, sizeof(double) * g.nvec, h->stream) != hipSuccess) { rc = OI_HIP_ERROR; break; }
    launch_pitch_copy_in(g, dense, vr, h->stream);
    rc = exchange_planes(h, vr, sizeof(double) * g.plane);
    if (rc != OI_OK) break;
    if (variant == OI_KERNEL_UNFUSED) launch_apply(g, h->code, vr, vq, h->partials, nullptr, h->stream);
    else launch_apply_tiled(g, h->code, vr, vq, h->partials, nullptr, h->stream);
    launch_pitch_copy_out(g, vq, dense, h->stream);
    if (hipMemcpyAsync(q, dense, n * sizeof(double), hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
        hipStreamSynchronize(h->stream) != hipSuccess) rc = OI_HIP_ERROR;
  } while (0);
  (void)oi::dev_free(dense);
  if (rc == OI_HIP_ERROR) oi::set_error("debug_apply: HIP failure: %s", hipGetErrorString(hipGetLastError()));
  return rc;
}

extern "C" oi_status oi_tort_bench_apply(oi_tort* h, int reps, int variant, double* ms_per_launch) {
  if (!h || reps < 1 || !ms_per_launch) { oi::set_error("%s: null handle or argument out of range", __func__); return OI_BAD_ARG; }
  if (!h->vectors_ready) return OI_NO_PERCOLATION;
  if (h->cg_active) { oi::set_error("%s: a CG run is open (oi_tort_cg_finish first)", __func__); return OI_BAD_ARG; }
  if (!h->q) { oi::set_error("the capacity mode keeps no A p vector: use another kernel variant for the stand-alone apply"); return OI_BAD_ARG; }
  HIPC(hipSetDevice(h->device));
  const Geom& g = h->g;
  // the compacted single-kernel layout keeps r / q as line lists: the stand-alone apply gets dense scratch
  struct Scratch { double *a = nullptr, *b = nullptr; ~Scratch() { if (a) (void)oi::dev_free(a); if (b) (void)oi::dev_free(b); } } scratch;
  double *vr = h->r, *vq = h->q;
  if (h->lineoff) {
    HIPC(oi::raw_malloc((void**)&scratch.a, sizeof(double) * g.nvec));
    HIPC(oi::raw_malloc((void**)&scratch.b, sizeof(double) * g.nvec));
    HIPC(hipMemsetAsync(scratch.a, 0, sizeof(double) * g.nvec, h->stream));
    HIPC(hipMemsetAsync(scratch.b, 0, sizeof(double) * g.nvec, h->stream));
    vr = scratch.a; vq = scratch.b;
  }
  // a smooth non-trivial field: the ramp (p = x0); one untimed launch warms the code path.  variant
  // OI_KERNEL_LEAN times the apply as the CG uses it: on a vector that vanishes off the free set (the initial
  // residual -A x0), with the live-line predication of the CG kernels
  if (variant == OI_KERNEL_LEAN) {
    launch_init_field(g, h->active, vq, h->stream);
    (void)launch_residual0(g, h->code, vq, vr, nullptr, 0, h->partials, nullptr, h->stream);
    HIPC(hipMemsetAsync(vq, 0, sizeof(double) * g.nvec, h->stream));
  } else {
    launch_init_field(g, h->active, vr, h->stream);
  }
  hipEvent_t a, b;
  HIPC(hipEventCreate(&a));
  HIPC(hipEventCreate(&b));
  for (int i = -1; i < reps; ++i) {
    if (i == 0) HIPC(hipEventRecord(a, h->stream));
    if (variant == OI_KERNEL_UNFUSED) launch_apply(g, h->code, vr, vq, h->partials, nullptr, h->stream);
    else if (variant == OI_KERNEL_LEAN) launch_apply_cgvec(g, h->code, vr, vq, h->partials, nullptr, h->stream);
    else launch_apply_tiled(g, h->code, vr, vq, h->partials, nullptr, h->stream);
  }
  HIPC(hipEventRecord(b, h->stream));
  HIPC(hipEventSynchronize(b));
  float ms = 0.f;
  HIPC(hipEventElapsedTime(&ms, a, b));
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  *ms_per_launch = (double)ms / reps;
  return OI_OK;
}

// ms per pass of a kernel that only moves the handle's six double-buffered CG vectors (r2, q2, p[1] -> r, q, p[0]) where
// they lie: the in-situ ceiling of a 3-read / 3-write stream, reported by bench.py beside the CG kernel's own rate.
// Only between CG runs (the vectors are scratch then; cg_begin initialises them).
extern "C" oi_status oi_tort_bench_stream(oi_tort* h, int reps, double* ms_per_pass, double* bytes_per_pass) {
  if (!h || !ms_per_pass || reps < 1) { oi::set_error("%s: null handle or argument out of range", __func__); return OI_BAD_ARG; }
  if (!h->cg1 || !h->vectors_ready || h->cg_active || h->p_f32) {
    oi::set_error("%s: needs the single-kernel variant (fp64 vectors) with no CG run in flight", __func__);
    return OI_BAD_ARG;
  }
  HIPC(hipSetDevice(h->device));
  const long long nd = h->nvec_cg / 512 * 512;
  hipEvent_t a = nullptr, b = nullptr;
  HIPC(hipEventCreate(&a)); HIPC(hipEventCreate(&b));
  float ms = 0.f;
  auto go = [&] { launch_stream_3r3w(h->r, h->q, (double*)h->p[0], h->r2, h->q2, (const double*)h->p[1], nd, h->stream); };
  go();
  oi_status st = OI_OK;
  if (hipEventRecord(a, h->stream) != hipSuccess) st = OI_HIP_ERROR;
  for (int r = 0; r < reps; ++r) go();
  if (hipEventRecord(b, h->stream) != hipSuccess || hipEventSynchronize(b) != hipSuccess || hipEventElapsedTime(&ms, a, b) != hipSuccess)
    st = OI_HIP_ERROR;
  (void)hipEventDestroy(a); (void)hipEventDestroy(b);
  OIC(st);
  *ms_per_pass = ms / reps;
  if (bytes_per_pass) *bytes_per_pass = 6.0 * 8.0 * (double)nd;
  return OI_OK;
}

// Launch shape the single-kernel iteration would take for a slab of nx x ny x nz_local voxels under the CURRENT process
// defaults: host arithmetic only, no device needed (the CPU tests check wide and thin planes against the capacity of the
// handle's partial-sum array).  descr = 1: the descriptor-addressed kernel of the compacted layout.
extern "C" oi_status oi_debug_cg1_grid(int nx, int ny, int nz_local, int descr, int split, int part, int* tile_rows, int* zchunk,
                                       int* blocks, int* capacity) {
  if (nx < 1 || ny < 1 || nz_local < 1) { oi::set_error("%s: argument out of range", __func__); return OI_BAD_ARG; }
  Geom g{};
  g.nx = nx; g.ny = ny; g.nz = nz_local; g.z0 = 0; g.nzl = nz_local;
  g.pitch = (nx + 1) & ~1;
  g.W = (nx + 63) / 64;
  g.plane = (long long)g.pitch * ny;
  g.tune = oi::default_tune();
  g.tune.max_blocks = oi::partial_blocks_bound(g.pitch, ny, nz_local);
  const oi::Cg1Grid gr = oi::cg1_grid(g, descr != 0, split, part);
  if (tile_rows) *tile_rows = gr.R * gr.NW;
  if (zchunk) *zchunk = gr.zchunk;
  if (blocks) *blocks = gr.blocks;
  if (capacity) *capacity = g.tune.max_blocks;
  return OI_OK;
}

extern "C" oi_status oi_tort_placement(oi_tort* h, int* nslots, int* slots7, int* tried, double* ms_first, double* ms_best) {
  if (!h) { oi::set_error("%s: null handle", __func__); return OI_BAD_ARG; }
  if (nslots) *nslots = h->vs.x ? h->vs.nslots : 0;
  if (slots7) for (int k = 0; k < 7; ++k) slots7[k] = h->placement[k];
  if (tried) *tried = h->placement_tried;
  if (ms_first) *ms_first = h->placement_ms_first;
  if (ms_best) *ms_best = h->placement_ms_best;
  return OI_OK;
}

extern "C" oi_status oi_tort_placement_cost(oi_tort* h, double* alloc_s, double* search_s, double* peak_bytes, double* kept_bytes, int* cache_hit) {
  if (!h) { oi::set_error("%s: null handle", __func__); return OI_BAD_ARG; }
  if (alloc_s) *alloc_s = h->place_alloc_s;
  if (search_s) *search_s = h->place_search_s;
  if (peak_bytes) *peak_bytes = h->place_peak_bytes;
  if (kept_bytes) *kept_bytes = (double)h->vs.bytes;
  if (cache_hit) *cache_hit = h->place_cache_hit;
  return OI_OK;
}

// ||b - A x|| / ||b|| of the CURRENT solution, recomputed from scratch: A x by the stand-alone apply (k_apply7 -- a kernel no
// solver variant uses inside its iteration), squared norm over the owned planes by a fixed-order reduction.  x carries
// the Dirichlet values, so on the free rows b - A x = -(A x); identity rows have no residual.  The convergence test and
// stats.final_relres use the RECURRENCE residual r_k = r_{k-1} - alpha q (as HYPRE's Krylov solvers do); this entry point
// is the independent check that the two agree.  Needs one dense scratch vector (8 bytes per voxel).
extern "C" oi_status oi_tort_true_residual(oi_tort* h, double* relres) {
  if (!h || !relres) { oi::set_error("%s: null handle or argument out of range", __func__); return OI_BAD_ARG; }
  *relres = std::numeric_limits<double>::quiet_NaN();
  if (!h->vectors_ready) return OI_NO_PERCOLATION;
  if (h->cg_active) { oi::set_error("%s: a CG run is open (oi_tort_cg_finish first)", __func__); return OI_BAD_ARG; }
  HIPC(hipSetDevice(h->device));
  const Geom& g = h->g;
  double* q = nullptr;
  if (oi::raw_malloc((void**)&q, sizeof(double) * g.nvec) != hipSuccess) {
    (void)hipGetLastError();
    oi::set_error("%s: no room for a scratch vector of %lld doubles", __func__, g.nvec);
    return OI_HIP_ERROR;
  }
  oi_status rc = OI_OK;
  do {
    if (hipMemsetAsync(q, 0, sizeof(double) * g.nvec, h->stream) != hipSuccess) { rc = OI_HIP_ERROR; break; }
    if ((rc = exchange_planes(h, h->x, sizeof(double) * g.plane)) != OI_OK) break;
    Geom ga = g;
    ga.tune.apply_kernel = 0;  // always the wave-autonomous apply: independent of the LDS-tiled solver kernels
    if (launch_apply_tiled(ga, h->code, h->x, q, h->partials, nullptr, h->stream) < 0) { rc = OI_BAD_ARG; break; }
    const int nb = launch_dot(g, q, q, h->partials, h->stream);
    launch_finalize(h->partials, nb, 1, h->scal, nullptr, 0, 0, h->stream);
    if (h->comm && (rc = h->comm->allreduce_sum_f64(h->scal, 1, h->stream)) != OI_OK) break;
    if (hipMemcpyAsync(h->h_scal, h->scal, sizeof(double), hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
        hipStreamSynchronize(h->stream) != hipSuccess) { rc = OI_HIP_ERROR; break; }
    *relres = h->bnorm > 0.0 ? std::sqrt(h->h_scal[0]) / h->bnorm : std::sqrt(h->h_scal[0]);
  } while (0);
  (void)oi::dev_free(q);
  if (rc == OI_HIP_ERROR) oi::set_error("%s: HIP failure: %s", __func__, hipGetErrorString(hipGetLastError()));
  return rc;
}

// checkMatrixProperties (TortuosityHypre.cpp:896-982) for the matrix-free operator: the
// stored code of every voxel must equal the row the reference would assemble:
//   inactive or Dirichlet cell -> identity row (code 0); the Dirichlet right-hand side
//   vlo/vhi is the value init_field stores; active interior cell -> diagonal = #active
//   neighbours > 0 and off-diagonals -1 on exactly those neighbours (row sum 0).
// Implemented by regenerating the codes from the mask into scratch and comparing.
extern "C" oi_status oi_tort_check_matrix_properties(oi_tort* h, int64_t* n_violations) {
  if (!h || !n_violations) { oi::set_error("%s: null handle or argument out of range", __func__); return OI_BAD_ARG; }
  *n_violations = 0;
  if (!h->vectors_ready) return OI_NO_PERCOLATION;
  if (h->cg_active) { oi::set_error("%s: a CG run is open (oi_tort_cg_finish first)", __func__); return OI_BAD_ARG; }
  HIPC(hipSetDevice(h->device));
  const Geom& g = h->g;
  const size_t n = (size_t)g.nx * g.ny * g.nzl;
  std::vector<uint8_t> code(n), mask(n);
  OIC(oi_tort_debug_get_codes(h, code.data()));
  OIC(oi_tort_get_mask(h, mask.data()));
  // ghost-aware neighbour test needs the neighbouring slabs' boundary planes
  std::vector<uint8_t> lo((size_t)g.nx * g.ny, 0), hi((size_t)g.nx * g.ny, 0);
  if (h->comm) {
    // ghost planes of the active bits were exchanged during construction: unpack them
    Geom gg = g;
    DevTmp<uint8_t> dense;
    HIPC(oi::raw_malloc((void**)&dense.p, (size_t)g.nx * g.ny));
    gg.nzl = 1;
    // plane 0 (lo ghost): unpack by shifting the base pointer one plane down
    launch_unpack_mask(gg, h->active - g.wplane, dense.p, h->stream);
    HIPC(hipMemcpyAsync(lo.data(), dense.p, lo.size(), hipMemcpyDeviceToHost, h->stream));
    launch_unpack_mask(gg, h->active + (long long)g.nzl * g.wplane, dense.p, h->stream);
    HIPC(hipMemcpyAsync(hi.data(), dense.p, hi.size(), hipMemcpyDeviceToHost, h->stream));
    HIPC(hipStreamSynchronize(h->stream));
  }
  const int nd[3] = {g.nx, g.ny, g.nz};
  long long bad = 0;
  const long long sx = 1, sy = g.nx, sz = (long long)g.nx * g.ny;
  for (int k = 0; k < g.nzl; ++k)
    for (int j = 0; j < g.ny; ++j)
      for (int i = 0; i < g.nx; ++i) {
        const long long m = i + sy * j + sz * k;
        const int c[3] = {i, j, g.z0 + k};
        const bool act = mask[m] != 0;
        const bool dirichlet = act && (c[g.dir] == 0 || c[g.dir] == nd[g.dir] - 1);
        if (!act || dirichlet) { bad += code[m] != 0; continue; }
        int cnt = 0;
        cnt += (i > 0 && mask[m - sx]) + (i < g.nx - 1 && mask[m + sx]);
        cnt += (j > 0 && mask[m - sy]) + (j < g.ny - 1 && mask[m + sy]);
        cnt += (k > 0 ? mask[m - sz] : lo[i + sy * j]) != 0;
        cnt += (k < g.nzl - 1 ? mask[m + sz] : hi[i + sy * j]) != 0;
        if (cnt <= 0 || code[m] != cnt) ++bad;  // Aii > 0 and zero row sum
      }
  unsigned long long b = (unsigned long long)bad;
  if (h->comm) OIC(h->comm->allreduce_sum_u64_host(&b, 1, h->stream));
  *n_violations = (int64_t)b;
  return OI_OK;
}

// fraction of the aligned `group_voxels`-long x-runs (this slab) that contain a free voxel
extern "C" oi_status oi_tort_live_fraction(oi_tort* h, int group_voxels, double* fraction) {
  if (!h || !fraction || group_voxels < 1) { oi::set_error("%s: null handle or argument out of range", __func__); return OI_BAD_ARG; }
  *fraction = 0.0;
  if (!h->percolates || !h->code) return OI_NO_PERCOLATION;
  HIPC(hipSetDevice(h->device));
  const Geom& g = h->g;
  HIPC(hipMemsetAsync(h->counters, 0, sizeof(unsigned long long), h->stream));
  launch_count_live_groups(g, h->code, group_voxels, h->counters, h->stream);
  HIPC(hipMemcpyAsync(h->h_counters, h->counters, sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
  HIPC(hipStreamSynchronize(h->stream));
  const double groups = (double)g.nzl * g.ny * ((g.nx + group_voxels - 1) / group_voxels);
  *fraction = (double)h->h_counters[0] / groups;
  return OI_OK;
}

extern "C" int oi_tort_kernel_times(oi_tort* h, int cap, int* kernel_ids, double* total_ms, int64_t* launches) {
  if (!h) return 0;
  prof_harvest(h);
  int n = 0;
  for (int k = 0; k < K_COUNT && n < cap; ++k) {
    if (h->k_n[k] == 0) continue;
    kernel_ids[n] = k;
    total_ms[n] = h->k_ms[k];
    launches[n] = h->k_n[k];
    ++n;
  }
  return n;
}
extern "C" const char* oi_kernel_name(int kid) { return (kid >= 0 && kid < K_COUNT) ? kKernelNames[kid] : "?"; }
extern "C" void oi_tort_reset_kernel_times(oi_tort* h) {
  if (!h) return;
  prof_harvest(h);
  for (int k = 0; k < K_COUNT; ++k) { h->k_ms[k] = 0; h->k_n[k] = 0; }
}

// ---- stand-alone helpers ---------------------------------------------------------------
extern "C" oi_status oi_volume_fraction(const uint8_t* phase, size_t n, int phase_id, int64_t* phase_count,
                                        int64_t* total_count) {
  if (!phase || !phase_count || !total_count) { oi::set_error("%s: null handle or argument out of range", __func__); return OI_BAD_ARG; }
  int device = 0;
  OIC(ensure_device(-1, &device));
  *total_count = (int64_t)n;
  *phase_count = 0;
  if (n == 0) return OI_OK;
  // treat the array as rows of up to 2^20 voxels
  Geom g{};
  g.nx = (int)std::min<size_t>(n, (size_t)1 << 20);
  const size_t rows = n / g.nx;
  g.ny = 1; g.nz = (int)rows; g.nzl = (int)rows; g.z0 = 0;
  g.W = (g.nx + 63) / 64; g.wplane = g.W; g.nwords = (long long)(g.nzl + 2) * g.wplane;
  DevTmp<uint8_t> d;
  DevTmp<uint64_t> bits;
  DevTmp<unsigned long long> cnt;
  HIPC(oi::raw_malloc((void**)&d.p, n));
  HIPC(oi::raw_malloc((void**)&bits.p, sizeof(uint64_t) * g.nwords));
  HIPC(oi::raw_malloc((void**)&cnt.p, sizeof(unsigned long long)));
  HIPC(hipMemset(cnt.p, 0, sizeof(unsigned long long)));
  HIPC(hipMemcpy(d.p, phase, n, hipMemcpyHostToDevice));
  launch_pack_open(d.p, g, phase_id, bits.p, cnt.p, nullptr);
  unsigned long long c = 0;
  HIPC(hipMemcpy(&c, cnt.p, sizeof(c), hipMemcpyDeviceToHost));
  // tail that does not fill a row
  for (size_t v = rows * g.nx; v < n; ++v) c += phase[v] == (uint8_t)phase_id;
  *phase_count = (int64_t)c;
  return OI_OK;
}

// tortuosity_remspot (Tortuosity_filcc.F90:88-177) on a host array: in place, x-fastest order, one box.
// Runs on the device like everything else (no CPU path).
extern "C" oi_status oi_remspot(uint8_t* q, int nx, int ny, int nz, int passes, int box_size) {
  if (!q || nx < 1 || ny < 1 || nz < 1 || passes < 0) { oi::set_error("%s: null handle or argument out of range", __func__); return OI_BAD_ARG; }
  if (passes == 0) return OI_OK;
  uint8_t* dev = nullptr;
  uint8_t* dev_phase = nullptr;
  OIC(remspot_private_copy(nullptr, q, false, nx, ny, nz, 0, nz, passes, box_size, -1, &dev, &dev_phase));
  const hipError_t e = hipMemcpy(q, dev_phase, (size_t)nx * ny * nz, hipMemcpyDeviceToHost);
  (void)oi::dev_free(dev);
  HIPC(e);
  return OI_OK;
}

extern "C" oi_status oi_gen_bernoulli_device(uint8_t* phase_dev, int nx, int ny, int nz_global, int z0,
                                             int nz_local, uint64_t seed, uint64_t thr53) {
  if (!phase_dev || nx < 1 || ny < 1 || nz_local < 1) { oi::set_error("%s: null handle or argument out of range", __func__); return OI_BAD_ARG; }
  launch_gen_bernoulli(phase_dev, nx, ny, nz_global, z0, nz_local, seed, thr53, nullptr);
  HIPC(hipGetLastError());
  HIPC(hipDeviceSynchronize());
  return OI_OK;
}

static inline uint64_t mix64_host(uint64_t z) {
  z += 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

extern "C" oi_status oi_gen_spheres_device(uint8_t* phase_dev, int nx, int ny, int nz_global, int z0,
                                           int nz_local, uint64_t seed, int nspheres, int r2h) {
  if (!phase_dev || nx < 1 || ny < 1 || nz_local < 1 || nspheres < 0 || r2h < 1) { oi::set_error("%s: null handle or argument out of range", __func__); return OI_BAD_ARG; }
  // centres in half-voxel units (same draw as the oracle), binned into cubic cells >= radius
  const int cell = std::max(r2h, 8);
  const int ncx = (2 * nx) / cell + 1, ncy = (2 * ny) / cell + 1, ncz = (2 * nz_global) / cell + 1;
  std::vector<int> c(3 * (size_t)std::max(nspheres, 1));
  std::vector<int> cell_of(nspheres);
  std::vector<int> start((size_t)ncx * ncy * ncz + 1, 0);
  for (int s = 0; s < nspheres; ++s) {
    c[3 * s + 0] = (int)(mix64_host(seed * 0xD1342543DE82EF95ULL + 3ULL * s + 0) % (uint64_t)(2 * nx));
    c[3 * s + 1] = (int)(mix64_host(seed * 0xD1342543DE82EF95ULL + 3ULL * s + 1) % (uint64_t)(2 * ny));
    c[3 * s + 2] = (int)(mix64_host(seed * 0xD1342543DE82EF95ULL + 3ULL * s + 2) % (uint64_t)(2 * nz_global));
    cell_of[s] = c[3 * s] / cell + ncx * (c[3 * s + 1] / cell + ncy * (c[3 * s + 2] / cell));
    ++start[cell_of[s] + 1];
  }
  for (size_t i = 1; i < start.size(); ++i) start[i] += start[i - 1];
  std::vector<int> fill(start.begin(), start.end() - 1), sorted(3 * (size_t)std::max(nspheres, 1));
  for (int s = 0; s < nspheres; ++s) {
    const int d = fill[cell_of[s]]++;
    sorted[3 * d] = c[3 * s]; sorted[3 * d + 1] = c[3 * s + 1]; sorted[3 * d + 2] = c[3 * s + 2];
  }
  DevTmp<int> dc, ds;
  HIPC(oi::raw_malloc((void**)&dc.p, sizeof(int) * sorted.size()));
  HIPC(oi::raw_malloc((void**)&ds.p, sizeof(int) * start.size()));
  HIPC(hipMemcpy(dc.p, sorted.data(), sizeof(int) * sorted.size(), hipMemcpyHostToDevice));
  HIPC(hipMemcpy(ds.p, start.data(), sizeof(int) * start.size(), hipMemcpyHostToDevice));
  launch_gen_spheres(phase_dev, nx, ny, nz_global, z0, nz_local, dc.p, ds.p, ncx, ncy, ncz, cell, r2h, nullptr);
  HIPC(hipDeviceSynchronize());
  return OI_OK;
}

extern "C" oi_status oi_gen_blobs_device(uint8_t* phase_dev, int nx, int ny, int nz_global, int z0, int nz_local,
                                         uint64_t seed, int spacing, uint64_t level) {
  if (!phase_dev || nx < 1 || ny < 1 || nz_local < 1 || spacing < 2 || spacing > 512 || level > 65535) { oi::set_error("%s: null handle or argument out of range", __func__); return OI_BAD_ARG; }
  (void)nz_global;
  launch_gen_blobs(phase_dev, nx, ny, z0, nz_local, seed, spacing, level, nullptr);
  HIPC(hipGetLastError());
  HIPC(hipDeviceSynchronize());
  return OI_OK;
}

// level (0..65535) whose blob structure has the wanted porosity: the quantile of the
// interpolated noise over a fixed sample of voxels (host arithmetic, no device needed)
extern "C" uint64_t oi_blobs_level(uint64_t seed, int spacing, double porosity) {
  if (spacing < 2 || !(porosity > 0.0) || !(porosity < 1.0)) return 0;
  const int n = 96;  // 96^3 sample stepping through several lattice cells
  const int step = std::max(1, (8 * spacing) / n + 1);
  std::vector<uint64_t> vals;
  vals.reserve((size_t)n * n * n);
  for (int k = 0; k < n; ++k)
    for (int j = 0; j < n; ++j)
      for (int i = 0; i < n; ++i) vals.push_back(blob_value_host(seed, spacing, i * step, j * step + 1, k * step + 2));
  const size_t q = (size_t)(porosity * vals.size());
  std::nth_element(vals.begin(), vals.begin() + q, vals.end());
  const uint64_t scale = (uint64_t)(2 * spacing) * (2 * spacing) * (2 * spacing);
  return std::min<uint64_t>(65535, vals[q] / scale + 1);
}

extern "C" oi_status oi_dev_malloc(void** ptr, size_t bytes) {
  if (!ptr) { oi::set_error("%s: null handle or argument out of range", __func__); return OI_BAD_ARG; }
  int device = 0;
  OIC(ensure_device(-1, &device));
  HIPC(oi::raw_malloc(ptr, bytes));
  return OI_OK;
}
extern "C" oi_status oi_dev_free(void* ptr) { HIPC(oi::dev_free(ptr)); return OI_OK; }

// The process's pool of released handle arrays (oi_pool.h): OFF by default.
extern "C" oi_status oi_dev_pool(int enable, double cap_gb) {
  oi::dev_pool_set(enable != 0, cap_gb > 0.0 ? (size_t)(cap_gb * 1e9) : 0);
  return OI_OK;
}
extern "C" double oi_dev_pool_trim(void) { return (double)oi::dev_pool_trim(); }
extern "C" oi_status oi_dev_pool_stats(double* held_bytes, long long* held_blocks, long long* hits, long long* misses, int* enabled) {
  const oi::DevPoolStats st = oi::dev_pool_stats();
  if (held_bytes) *held_bytes = (double)st.held_bytes;
  if (held_blocks) *held_blocks = (long long)st.held_blocks;
  if (hits) *hits = (long long)st.hits;
  if (misses) *misses = (long long)st.misses;
  if (enabled) *enabled = st.on;
  return OI_OK;
}
// the ledger of oi_pool.h: requested bytes / blocks of every device allocation of this library that is live now (blocks the pool
// holds are not live), and the high-water mark since the last reset.  Exact: the leak tests compare these, not hipMemGetInfo.
extern "C" long long oi_dev_live_bytes(void) { return (long long)oi::dev_ledger().live_bytes; }
extern "C" long long oi_dev_live_blocks(void) { return (long long)oi::dev_ledger().live_blocks; }
extern "C" long long oi_dev_peak_bytes(int reset) {
  const long long p = (long long)oi::dev_ledger().peak_bytes;
  if (reset) oi::dev_ledger_reset_peak();
  return p;
}
// test hook: the n-th device allocation from now fails once with an out-of-memory error (0 disarms)
extern "C" void oi_debug_fail_alloc_after(long long n) { oi::dev_fail_after(n); }
extern "C" oi_status oi_dev_mem_info(size_t* free_bytes, size_t* total_bytes) {
  if (!free_bytes || !total_bytes) { oi::set_error("%s: null handle or argument out of range", __func__); return OI_BAD_ARG; }
  int device = 0;
  OIC(ensure_device(-1, &device));
  HIPC(hipMemGetInfo(free_bytes, total_bytes));
  return OI_OK;
}
// time `reps` launches of the library's reference copy stream (two scratch arrays of `bytes` each, allocated here)
// ---- page-locked output buffers for the readers (oi_io.cpp's allocation hooks) ---------------------------------------------
namespace oi {
extern void* (*g_host_alloc)(size_t);
extern void (*g_host_free)(void*);
}
namespace {
std::mutex g_pinned_mutex;
std::vector<void*> g_pinned;  // page-locked buffers handed out (few and large: a linear search is fine)
void* pinned_alloc(size_t n) {
  void* p = nullptr;
  if (hipHostMalloc(&p, n ? n : 1, hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError();
    return malloc(n);  // (no device, or the pinned pool is exhausted: pageable memory is still a correct buffer)
  }
  std::lock_guard<std::mutex> lock(g_pinned_mutex);
  g_pinned.push_back(p);
  return p;
}
void pinned_free(void* p) {
  if (!p) return;
  {
    std::lock_guard<std::mutex> lock(g_pinned_mutex);
    for (size_t k = 0; k < g_pinned.size(); ++k)
      if (g_pinned[k] == p) {
        g_pinned[k] = g_pinned.back();
        g_pinned.pop_back();
        (void)hipHostFree(p);
        return;
      }
  }
  free(p);
}
}  // namespace
// on: the readers (oi_read_*) return page-locked buffers from now on (oi_free_host releases either kind)
extern "C" void oi_read_set_pinned(int on) {
  // (the free hook stays installed once set: buffers handed out earlier must still find their way back)
  oi::g_host_free = pinned_free;
  oi::g_host_alloc = on ? pinned_alloc : nullptr;
}

extern "C" oi_status oi_dev_stream_copy_bench(size_t bytes, int reps, double* ms_per_copy) {
  if (!ms_per_copy || reps < 1 || bytes < 4096) { oi::set_error("%s: null handle or argument out of range", __func__); return OI_BAD_ARG; }
  int device = 0;
  OIC(ensure_device(-1, &device));
  const long long nd = (long long)(bytes / 16) * 2;
  double *a = nullptr, *b = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  oi_status st = OI_OK;
  float ms = 0.f;
  if (oi::raw_malloc((void**)&a, nd * 8) != hipSuccess || oi::raw_malloc((void**)&b, nd * 8) != hipSuccess) {
    oi::set_error("%s: not enough device memory for 2 x %zu bytes", __func__, bytes);
    (void)hipGetLastError();
    st = OI_HIP_ERROR;
  } else if (hipMemset(a, 0, nd * 8) != hipSuccess || hipMemset(b, 0, nd * 8) != hipSuccess ||
             hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
    st = OI_HIP_ERROR;
  } else {
    launch_stream_copy(b, a, nd, nullptr);
    (void)hipEventRecord(e0, nullptr);
    for (int r = 0; r < reps; ++r) launch_stream_copy(b, a, nd, nullptr);
    (void)hipEventRecord(e1, nullptr);
    if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) st = OI_HIP_ERROR;
  }
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  if (a) (void)oi::dev_free(a);
  if (b) (void)oi::dev_free(b);
  if (st == OI_OK) *ms_per_copy = (double)ms / reps;
  return st;
}
extern "C" oi_status oi_dev_upload(void* dst, const void* src, size_t bytes) {
  HIPC(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
  return OI_OK;
}
extern "C" oi_status oi_dev_download(void* dst, const void* src, size_t bytes) {
  HIPC(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
  return OI_OK;
}
extern "C" int oi_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}
extern "C" const char* oi_last_error(void) { return oi::last_error(); }
extern "C" const char* oi_status_string(oi_status s) {
  switch (s) {
    case OI_OK: return "ok";
    case OI_BAD_ARG: return "bad argument";
    case OI_NO_PERCOLATION: return "no percolating path (active volume fraction is zero)";
    case OI_NOT_CONVERGED: return "solver did not converge";
    case OI_FLUX_NOT_CONSERVED: return "flux not conserved";
    case OI_HIP_ERROR: return "HIP error";
    case OI_RCCL_ERROR: return "RCCL error";
    case OI_NO_DEVICE: return "no HIP device";
    case OI_IO_ERROR: return "I/O error";
  }
  return "?";
}
extern "C" int oi_abi_version(void) { return OI_ABI_VERSION; }
extern "C" int oi_build_has_xcheck(void) { return oi::xcheck_built() ? 1 : 0; }
extern "C" void oi_tort_set_tuning(int fused_tile_rows, int fused_zchunk) {
  oi::Tune t = oi::default_tune();
  if (fused_tile_rows == 8 || fused_tile_rows == 16 || fused_tile_rows == 32) t.fused_rows = fused_tile_rows / 8;
  t.zchunk = fused_zchunk > 0 ? fused_zchunk : 0;
  oi::set_default_tune(t);
}
extern "C" void oi_tort_set_apply_tuning(int tile_rows) {
  oi::Tune t = oi::default_tune();
  if (tile_rows == 8 || tile_rows == 16 || tile_rows == 32) t.pure_rows = tile_rows / 8;
  oi::set_default_tune(t);
}
extern "C" void oi_tort_set_cg1_tuning(int tile_rows, int x_every_second) {
  oi::Tune t = oi::default_tune();
  oi::tune_set_cg1_tile(t, tile_rows);
  t.cg1_xskip = x_every_second ? 1 : 0;
  oi::set_default_tune(t);
}
// named switches of the process defaults (experiments, cross-checks; the keys are listed in include/openimpala_amd.h)
extern "C" oi_status oi_tort_set_option(const char* key, double value) {
  if (!key) { oi::set_error("%s: null key", __func__); return OI_BAD_ARG; }
  oi::Tune t = oi::default_tune();
  const std::string k(key);
  const int iv = (int)value;
  if (k == "tile_map") t.tile_map = iv;
  else if (k == "cg1_nt") t.cg1_nt = iv;
  else if (k == "dense_nt") t.dense_nt = iv & 31;
  else if (k == "zchunk") t.zchunk = iv > 0 ? iv : 0;
  else if (k == "apply_kernel") t.apply_kernel = iv;
  else if (k == "mg_plain") t.mg_plain = iv != 0;
  else if (k == "mg_novec4") t.mg_novec4 = iv != 0;
  else if (k == "mg_debug") t.mg_debug = iv != 0;
  else if (k == "mg_replicate_cells") t.mg_replicate_cells = (long long)value;
  else if (k == "mg_wmask") t.mg_wmask = iv;
  else if (k == "mg_scale") t.mg_scale = (float)value;
  else if (k == "mg_omega") t.mg_omega = (float)value;
  else if (k == "mg_omega_coarse") t.mg_omega_coarse = (float)value;
  else if (k == "mg_coarsest_plain") t.mg_coarsest_plain = iv != 0;
  else if (k == "mg_nu_coarse") t.mg_nu_coarse = iv >= 2 ? 2 : (iv == 1 ? 1 : 0);
  else if (k == "force_comm") t.force_comm = iv != 0;
  else if (k == "resid0_bits") t.resid0_bits = iv != 0;
  else if (k == "ring_lines_rccl") t.ring_lines_rccl = iv != 0;
  else if (k == "place_trials") t.place_trials = iv > 0 ? iv : 0;
  else if (k == "place_slots") t.place_slots = iv;
  else if (k == "place_spacer_gb") t.place_spacer_gb = (float)value;
  else if (k == "cg1_kernel" || k == "fused_kernel") {
    // 1 selects a superseded kernel (k_cg1 / k_fused_a on compacted vectors) kept as the bit-identity reference: test library only
    if (iv != 0 && !oi::xcheck_built()) { oi::set_error("%s: '%s' = %d selects a cross-check kernel of the test library (make xcheck), not in this build", __func__, key, iv); return OI_BAD_ARG; }
    if (k == "cg1_kernel") t.cg1_kernel = iv; else t.fused_kernel = iv;
  }
  else if (k == "cg1_seg") t.cg1_seg = iv != 0;
  else if (k == "mg_norot") t.mg_norot = iv != 0;
  else if (k == "mg_xpair") t.mg_xpair = iv != 0;
  else if (k == "mg_blocked") t.mg_blocked = iv != 0;
  else if (k == "mg_coarse_tiled") t.mg_coarse_tiled = iv < 0 ? 0 : iv;
  else if (k == "mg_fused_images") t.mg_fused_images = iv != 0;
  else if (k == "mg_post_kernel") t.mg_post_kernel = iv != 0;
  else if (k == "mg_xfirst") t.mg_xfirst = iv != 0;
  else if (k == "cg1_fin_sliced") t.cg1_fin_sliced = iv != 0;
  else if (k == "fused_zchunk") t.fused_zchunk = iv > 0 ? iv : 0;
  else if (k == "place_min_mb") t.place_min_mb = iv > 0 ? iv : 0;
  else if (k == "place_extent_gb") t.place_extent_gb = (float)value;
  else if (k == "place_cache") t.place_cache = iv != 0;
  else if (k == "place_alloc_budget_s") t.place_alloc_budget_s = (float)value;
  else if (k == "place_retry_gbps") t.place_retry_gbps = value > 0.0 ? (float)value : 0.f;
  else { oi::set_error("%s: unknown key '%s'", __func__, key); return OI_BAD_ARG; }
  oi::set_default_tune(t);
  return OI_OK;
}
// take the CURRENT process defaults into a live handle (launch shapes only: the layout, the capacity of the
// partial-sum array and the multigrid hierarchy were fixed when the handle was built)
extern "C" oi_status oi_tort_retune(oi_tort* h) {
  if (!h) { oi::set_error("%s: null handle", __func__); return OI_BAD_ARG; }
  const oi::Tune old = h->g.tune;
  oi::Tune t = oi::default_tune();
  t.max_blocks = old.max_blocks;
  t.cg1_compact = old.cg1_compact;
  t.cg1_xskip = h->cg_active ? old.cg1_xskip : t.cg1_xskip;  // frozen while a CG run is in flight
  t.mg_plain = old.mg_plain; t.mg_novec4 = old.mg_novec4;
  t.mg_replicate_cells = old.mg_replicate_cells; t.mg_wmask = old.mg_wmask; t.mg_scale = old.mg_scale; t.mg_nu_coarse = old.mg_nu_coarse; t.mg_omega = old.mg_omega; t.mg_omega_coarse = old.mg_omega_coarse; t.mg_coarsest_plain = old.mg_coarsest_plain; t.mg_post_kernel = old.mg_post_kernel; t.mg_xfirst = old.mg_xfirst; t.mg_xpair = old.mg_xpair; t.mg_blocked = old.mg_blocked; t.mg_coarse_tiled = old.mg_coarse_tiled; t.mg_fused_images = old.mg_fused_images; t.cg1_fin_sliced = old.cg1_fin_sliced;
  t.force_comm = old.force_comm; t.ring_lines_rccl = old.ring_lines_rccl;
  t.place_trials = old.place_trials; t.place_slots = old.place_slots; t.place_spacer_gb = old.place_spacer_gb;
  t.place_min_mb = old.place_min_mb; t.place_extent_gb = old.place_extent_gb; t.place_cache = old.place_cache;
  t.place_alloc_budget_s = old.place_alloc_budget_s; t.place_retry_gbps = old.place_retry_gbps;
  h->g.tune = t;
  return OI_OK;
}
