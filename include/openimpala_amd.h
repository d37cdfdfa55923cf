/*
 * openimpala_amd.h -- C ABI of the MI355X-native flow-through tortuosity solver.
 *
 * Drop-in boundary for ONE path of kramergroup/openImpala: the `src/props`
 * TortuosityHypre solve (percolation mask -> masked 7-point Laplacian ->
 * Krylov solve -> boundary fluxes -> tau).  Each entry point names the reference
 * interface it replaces (paths relative to the reference tree).  Plain pointers and
 * sizes only; no AMReX, HYPRE or torch types cross this boundary.
 *
 * Data convention (identical to the reference's Fortran-ordered boxes and to the
 * TIFF/RAW/HDF5 payload order): voxel (i,j,k) = (X,Y,Z) lives at
 * phase[i + nx*(j + ny*k)]; X = TIFF width (unit stride), Z = TIFF page (slowest).
 * Direction codes follow OpenImpala::Direction (src/props/Tortuosity.H:9-13):
 * X = 0, Y = 1, Z = 2.
 *
 * Threading: a handle is not thread-safe (as the reference class); distinct handles
 * may be driven from distinct host threads.  The library never aborts the process:
 * hard failures come back as oi_status, soft failures additionally as NaN / +Inf in
 * *tau exactly as TortuosityHypre::value() would return them.
 */
#ifndef OPENIMPALA_AMD_H
#define OPENIMPALA_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OI_ABI_VERSION 2

typedef enum oi_status {
  OI_OK = 0,
  OI_BAD_ARG = 1,
  OI_NO_PERCOLATION = 2,     /* active volume fraction 0: tau = NaN  (TortuosityHypre.cpp:170-178,764-777) */
  OI_NOT_CONVERGED = 3,      /* tau = NaN                            (TortuosityHypre.cpp:782-787) */
  OI_FLUX_NOT_CONSERVED = 4, /* |in|-|out| > 1e-6 relative: tau = NaN (TortuosityHypre.cpp:794-823) */
  OI_HIP_ERROR = 5,
  OI_RCCL_ERROR = 6,
  OI_NO_DEVICE = 7,          /* no HIP device / extension unusable: never a silent CPU fallback */
  OI_IO_ERROR = 8
} oi_status;

typedef struct oi_tort oi_tort; /* replaces an OpenImpala::TortuosityHypre object */
typedef struct oi_comm oi_comm; /* replaces MPI_COMM_WORLD / amrex::ParallelDescriptor for this path */

/* MG: one 2x2x2 aggregation-multigrid cycle per PCG iteration (what SMG is to the reference's
 * FlexGMRES): tens of iterations instead of O(10 N); works on z-slabs too */
/* NONE / JACOBI: (Jacobi-preconditioned) CG; MG: one aggregation-multigrid cycle per PCG iteration, the role of SMG in
 * the reference's FlexGMRES (TortuosityHypre.cpp:664-693).  DEFAULT (what oi_tort_default_opts selects): MG where the
 * kernel variant runs it (auto / fused / unfused), plain CG for the explicitly chosen fused1 / lean variants. */
enum { OI_PRECOND_NONE = 0, OI_PRECOND_JACOBI = 1, OI_PRECOND_MG = 2, OI_PRECOND_DEFAULT = 3 };
/* UNFUSED: apply / p-update / x,r-update kernels (89 B/voxel-iter); FUSED: two kernels (73 B);
 * FUSED1: ONE kernel per iteration with a single reduction (61 B), see DESIGN.md section 4;
 * LEAN: capacity mode -- only x, r, p are stored (25 B/voxel of HBM instead of 57): A p is never
 * written, the x/r update recomputes it (74 B/voxel-iter, still exact fp64 CG); for volumes that
 * do not fit otherwise (BASELINE config 5: 4096^3 on 8 GPUs = 215 GB per GPU) */
enum { OI_KERNEL_AUTO = 0, OI_KERNEL_UNFUSED = 1, OI_KERNEL_FUSED = 2, OI_KERNEL_FUSED1 = 3, OI_KERNEL_LEAN = 4 };
enum { OI_STORE_F64 = 0, OI_STORE_F32 = 1 };

/* Options.  `eps` / `maxiter` are the reference's hypre.eps / hypre.maxiter
 * (TortuosityHypre.cpp:142-149: 1e-9 / 200).  The reference's 200 counts
 * multigrid-preconditioned FlexGMRES iterations; CG needs O(10 N) iterations, so
 * maxiter <= 0 selects 40*(nx+ny+nz). */
typedef struct oi_tort_opts {
  double eps;
  int maxiter;
  int precond;        /* OI_PRECOND_* */
  int remspot_passes; /* tortuosity.remspot_passes (TortuosityHypre.cpp:254-256), default 0 */
  int remspot_box_size; /* the inputs file's box_size (default 32, the reference's: Diffusion.cpp:185, tTortuosity.cpp:125): the filter runs
                         * per box of ba.maxSize(box_size), see oi_remspot; <= 0 = one box spanning the whole domain; works on z-slabs */
  int verbose;        /* 0 silent .. 3 */
  int check_every;    /* host looks at the device convergence flag every this many iterations (<=0: auto) */
  int kernel;         /* OI_KERNEL_* */
  int profile;        /* 1: bracket every kernel launch with HIP events (oi_tort_kernel_times) */
  int device;         /* HIP device ordinal; -1 = the calling thread's current device */
  int p_storage;      /* OI_STORE_F64 (default) | OI_STORE_F32: keep the CG search direction p in fp32
                         in HBM (8 B/voxel-iter less traffic).  x, r, q and ALL arithmetic stay fp64; r
                         remains the exact recurrence residual of x, only the search direction is
                         perturbed at the 1e-7 level (fused kernel only). */
} oi_tort_opts;

void oi_tort_default_opts(oi_tort_opts* opts);

/* What TortuosityHypre exposes through its getters (TortuosityHypre.H:114-120)
 * plus timings of the phases the reference wraps in BL_PROFILE. */
typedef struct oi_tort_stats {
  int converged;          /* getSolverConverged() */
  int iterations;         /* getSolverIterations() */
  double final_relres;    /* getFinalRelativeResidualNorm(): ||r|| / ||b|| of the full system, r the solver's RECURRENCE residual
                           * (what the stop test sees, as in HYPRE); oi_tort_true_residual recomputes b - A x independently */
  double flux_in;         /* getFluxIn() */
  double flux_out;        /* getFluxOut() */
  double active_vf;       /* getActiveVolumeFraction() */
  double vf;              /* VolumeFraction::value for phase_id over the whole domain */
  double deff;            /* D_eff / D_0 (TortuosityHypre.cpp:868) */
  double tau;             /* cached value() */
  int64_t n_active;       /* voxels in the percolating mask (global) */
  int64_t n_free;         /* unknowns: active voxels not on a Dirichlet face (global) */
  int64_t n_phase;        /* voxels with phase == phase_id (global) */
  int flood_sweeps;       /* device sweeps the percolation flood fill took */
  double t_upload_s, t_mask_s, t_setup_s, t_solve_s, t_flux_s;
  double bytes_device;    /* device memory held by the handle */
} oi_tort_stats;

/* ---- construction: TortuosityHypre::TortuosityHypre (TortuosityHypre.cpp:100-191) ----
 * Copies `phase` (host memory, nx*ny*nz bytes; the caller may free it at once, like the
 * reference's deep copy at :132), optionally runs the remspot filter, builds the
 * percolation mask on the device and prepares the linear system.  vf of the reference
 * signature is recomputed on the device (VolumeFraction.cpp:22-66). */
oi_status oi_tort_create(oi_tort** out, const uint8_t* phase, int nx, int ny, int nz, int phase_id,
                         int dir, double vlo, double vhi, const oi_tort_opts* opts);

/* Same, `phase_dev` already resident in HBM (device pointer, same layout). */
oi_status oi_tort_create_device(oi_tort** out, const uint8_t* phase_dev, int nx, int ny, int nz,
                                int phase_id, int dir, double vlo, double vhi,
                                const oi_tort_opts* opts);

/* One z-slab of a volume decomposed over `comm` (replaces the BoxArray +
 * DistributionMapping + FillBoundary machinery, Diffusion.cpp:266-268).  This rank
 * owns global planes [z0, z0 + nz_local); `phase` holds only those planes.
 * `on_device` != 0: `phase` is a device pointer.  Collective over comm. */
oi_status oi_tort_create_slab(oi_tort** out, oi_comm* comm, const uint8_t* phase, int on_device,
                              int nx, int ny, int nz_global, int z0, int nz_local, int phase_id,
                              int dir, double vlo, double vhi, const oi_tort_opts* opts);

/* TortuosityHypre::~TortuosityHypre (TortuosityHypre.cpp:195-203) */
void oi_tort_destroy(oi_tort* h);

/* Tortuosity::value(refresh) (Tortuosity.H:47-69, TortuosityHypre.cpp:761-891).
 * Solves on first call or when refresh != 0; *tau is NaN / +Inf exactly where the
 * reference returns them.  Collective over the handle's comm. */
oi_status oi_tort_value(oi_tort* h, int refresh, double* tau);

/* TortuosityHypre::checkMatrixProperties (TortuosityHypre.cpp:896-982) evaluated on the
 * matrix-free operator's coefficients: *n_violations = 0 means pass. */
oi_status oi_tort_check_matrix_properties(oi_tort* h, int64_t* n_violations);

/* getters (TortuosityHypre.H:114-120) */
oi_status oi_tort_get_stats(const oi_tort* h, oi_tort_stats* stats);

/* The potential field (what the reference writes as plotfile component
 * "solution_potential", TortuosityHypre.cpp:710-749): nx*ny*nz_local doubles. */
oi_status oi_tort_get_solution(oi_tort* h, double* phi);
/* "active_mask" plotfile component: nx*ny*nz_local bytes (0/1). */
oi_status oi_tort_get_mask(oi_tort* h, uint8_t* mask);

/* VolumeFraction::value (VolumeFraction.cpp:22-66) on a host array, computed on the device. */
oi_status oi_volume_fraction(const uint8_t* phase, size_t n, int phase_id, int64_t* phase_count,
                             int64_t* total_count);

/* preconditionPhaseFab (TortuosityHypre.cpp:248-292) on a host array, in place, computed on the device: `passes` passes
 * of tortuosity_remspot (Tortuosity_filcc.F90:88-177) over the boxes ba.maxSize(box_size) makes of the domain
 * (Diffusion.cpp:266-268; box_size 32 is the reference's default).  The reference filters every box in place in its own
 * fab, so a cell sees the new value of an earlier neighbour only inside its own box -- the result depends on box_size,
 * and this entry point reproduces that dependence for ONE OpenMP thread (with several the reference's tiles race).
 * box_size <= 0: the whole domain as one box.  Box cuts follow AMReX's BoxList::maxSize as published (AMReX itself is
 * not available here); releases before the 2019 rewrite put the larger blocks at the high end instead, which differs
 * only when box_size does not divide an extent. */
oi_status oi_remspot(uint8_t* phase, int nx, int ny, int nz, int passes, int box_size);

/* change hypre.eps of an existing handle (takes effect at the next solve / cg_begin) */
oi_status oi_tort_set_eps(oi_tort* h, double eps);

/* ---- homogenization path (the reference's default calculation_method) ----
 * One object serves the three cell problems chi_x, chi_y, chi_z of a periodic volume and the
 * D_eff tensor built from them.  Replaces OpenImpala::EffectiveDiffusivityHypre
 * (EffectiveDiffusivityHypre.cpp: ctor + generateActiveMask :213-323, solve :543-687,
 * getChiSolution :689-; assembly EffDiffFillMtx.F90:42-264) and
 * calculate_Deff_tensor_homogenization (Diffusion.cpp:60-167).
 * opts: eps / maxiter = hypre.eps / hypre.maxiter (defaults 1e-9 / 1000, :120,152-154). */
typedef struct oi_effdiff oi_effdiff;
oi_status oi_effdiff_create(oi_effdiff** out, const uint8_t* phase, int nx, int ny, int nz, int phase_id,
                            const oi_tort_opts* opts);
/* One z-slab [z0, z0 + nz_local) per rank of `comm` (declared below); the ranks form a RING (periodic z:
 * rank 0 and the last rank exchange planes too).  Collective; `phase` holds the slab's planes only, on the
 * host or (on_device != 0) in HBM.  get_chi returns the slab, tensor / info the global values on every rank. */
oi_status oi_effdiff_create_slab(oi_effdiff** out, oi_comm* comm, const uint8_t* phase, int on_device, int nx, int ny,
                                 int nz_global, int z0, int nz_local, int phase_id, const oi_tort_opts* opts);
/* solve(): *converged = 0 leaves chi_k = 0 exactly as the reference does (:641-645) */
oi_status oi_effdiff_solve(oi_effdiff* h, int dir_k, int* converged, int* iterations, double* relres);
oi_status oi_effdiff_get_chi(oi_effdiff* h, int dir_k, double* chi);
oi_status oi_effdiff_tensor(oi_effdiff* h, double* deff9 /* row-major 3x3, D_eff / D_material */);
oi_status oi_effdiff_info(const oi_effdiff* h, int64_t* n_active, int* iterations3, double* relres3);
/* *compacted = 1: the CG vectors hold only the 8-voxel lines with an active cell and the solves run the flow-through path's
 * rotated multigrid iteration (multigrid preconditioner; one GPU, or a ring whose slabs all start on an even plane); *vector_elements = doubles per vector */
oi_status oi_effdiff_layout(const oi_effdiff* h, int* compacted, int64_t* vector_elements);
void oi_effdiff_destroy(oi_effdiff* h);

/* ---- stepping interface (bench / tests): the pieces of solve() individually ---- */
oi_status oi_tort_cg_begin(oi_tort* h);              /* x0 = ramp, r0, p0; resets the iteration count */
oi_status oi_tort_cg_iterate(oi_tort* h, int n);     /* enqueue n CG iterations (asynchronous) */
oi_status oi_tort_cg_finish(oi_tort* h);             /* flush the deferred x update, synchronise */
/* The library's own convergence loop between cg_begin and cg_finish -- what oi_tort_value runs: launches iterations until the device's
 * stop test fires, opts.maxiter is reached, or (max_launches > 0) that many launches have been issued by this call.  Plain CG: batches
 * of 32 launches with the flag of the previous batch read while the next one runs; multigrid-preconditioned CG: the batch shrinks to
 * what the residual still needs (estimated from its contraction so far), so the solve does not end on launches whose coarse levels run
 * for nothing.  *launches_issued (optional) = launches this call enqueued. */
oi_status oi_tort_cg_run(oi_tort* h, int max_launches, int* launches_issued);
oi_status oi_tort_sync(oi_tort* h);
oi_status oi_tort_cg_state(oi_tort* h, int* iterations_done, double* relres, int* converged);
/* element-wise checks of single kernels against the oracle: v, q are host arrays of
 * nx*ny*nz_local doubles; variant = OI_KERNEL_UNFUSED | OI_KERNEL_FUSED */
oi_status oi_tort_debug_apply(oi_tort* h, const double* v, double* q, int variant);
oi_status oi_tort_debug_get_codes(oi_tort* h, uint8_t* codes);
/* ||b - A x|| / ||b|| of the current solution recomputed from scratch on the device (A x by the stand-alone apply kernel,
 * which no solver variant uses inside its iteration; one dense scratch vector, 8 bytes per voxel; collective on slabs).
 * Same normalisation as stats.final_relres. */
oi_status oi_tort_true_residual(oi_tort* h, double* relres);
/* Launch shape the single-kernel CG iteration would take for a slab of nx x ny x nz_local voxels under the current process
 * defaults -- tile rows, planes per block, blocks (-1: would not fit) -- and the capacity (blocks) of a handle's
 * partial-sum array for that slab.  Host arithmetic only: no device is touched. */
oi_status oi_debug_cg1_grid(int nx, int ny, int nz_local, int descr, int split, int part, int* tile_rows, int* zchunk,
                            int* blocks, int* capacity);
/* Where the vectors of the single-kernel iteration were put.  When a vector is larger than the 256 MB Infinity Cache
 * ("place_min_mb"), x and "place_slots" vector-sized slots are allocated back to back when the handle is built (with a
 * spacer between the two halves where the set would otherwise span less than "place_extent_gb"), `place_trials`
 * placements of the six double-buffered vectors in those slots (oi_tort_set_option, default 48) are timed with the real
 * kernel, the fastest kept and the slots it does not use FREED: on MI355X the iteration time depends on the distances
 * between the arrays read and the arrays written (HBM address mapping).  nslots = 0: no search.
 * slots7[0..5] = slots of r, q, p[0], r2, q2, p[1]; slots7[6] = 0 (x is the first allocation of the set).  ms_first /
 * ms_best = ms per iteration of the in-order layout and of the layout kept (0 when nothing was timed).  The results of
 * the solve do not depend on the placement. */
oi_status oi_tort_placement(oi_tort* h, int* nslots, int* slots7, int* tried, double* ms_first, double* ms_best);
/* What the placement cost when the handle was built: seconds spent allocating the set (the driver clears memory another
 * process has used: seconds for tens of GB) and searching, device bytes held by the handle at the peak of the search,
 * bytes of x and the six vectors kept afterwards, and whether the process's placement cache answered (a handle of the
 * same shape built earlier: one verification cycle instead of a search; "place_cache"). */
oi_status oi_tort_placement_cost(oi_tort* h, double* alloc_s, double* search_s, double* peak_bytes, double* kept_bytes, int* cache_hit);
/* ms per pass of a kernel that only MOVES the six double-buffered vectors of the single-kernel iteration where they lie
 * (three read, three written, 16-byte nontemporal accesses): the in-situ ceiling of that traffic mix on this card at
 * this placement.  Only between CG runs (the vectors are scratch then). */
oi_status oi_tort_bench_stream(oi_tort* h, int reps, double* ms_per_pass, double* bytes_per_pass);
/* run the stand-alone stencil apply `reps` times on device-resident vectors and
 * return the HIP-event time of the launches (ms per launch); variant OI_KERNEL_LEAN = the apply as the CG
 * uses it (a vector that vanishes off the free set, live-line predication) */
oi_status oi_tort_bench_apply(oi_tort* h, int reps, int variant, double* ms_per_launch);

/* Launch shapes and layouts.  The oi_tort_set_* calls below change the process DEFAULTS (under a mutex); a handle
 * copies the defaults when it is created and is never touched by a later call -- no tuning state is shared between
 * live handles (oi_tort_retune re-reads the defaults for ONE handle, from the thread that drives it).
 * Fused CG kernel: rows per LDS tile (8, 16 or 32) and planes per block of every z-marching kernel (0 = automatic). */
void oi_tort_set_tuning(int fused_tile_rows, int fused_planes_per_block);
/* rows per LDS tile (8, 16 or 32) of the stand-alone stencil apply */
void oi_tort_set_apply_tuning(int tile_rows);
/* single-kernel CG iteration: rows per LDS tile (8, 16, 32; <= 0 automatic) and whether x is
 * updated every second iteration with both pending terms (1, default) or every iteration (0) */
void oi_tort_set_cg1_tuning(int tile_rows, int x_every_second);
/* layout of r, q, p in the single-kernel iteration: 1 (default) = COMPACTED -- only the aligned lines of
 * oi_tort_line_voxels() voxels (8 = one 64-byte HBM burst) that hold a free voxel are stored (contiguous in
 * HBM, half the memory at porosity 0.4), 0 = dense with
 * predicated accesses.  Affects handles created after the call.  Bit 1 (value 2) switches the kernel from
 * whole-line to pair-granular accesses inside live lines (an A/B switch: whole lines avoid partial-line
 * stores and measure 8-15 % faster).  The multigrid-preconditioned two-kernel iteration (r, q, z, p) takes the
 * same layout (a third less memory and 4 % faster than dense vectors at 1024^3) when every slab starts on an
 * even plane. */
void oi_tort_set_cg1_layout(int compact_lines);
/* voxels per stored line of the compacted layout (a build constant) */
int oi_tort_line_voxels(void);
/* named switches of the process defaults (experiments and cross-checks; none is needed in normal use):
 *   "tile_map"      0 / 1 / 2   block -> tile order of the z-marching kernels (2 = z sweep, default)
 *   "zchunk"        planes per block (0 = automatic)      "apply_kernel"  0 = wave-autonomous apply, 1 = LDS-tiled
 *   "mg_plain" "mg_novec4" "mg_debug"   multigrid cross-check kernels / per-level norms
 *   "mg_replicate_cells" "mg_wmask" "mg_scale"   multigrid level distribution and cycle shape
 *   "force_comm"    keep a one-rank communicator (single-GPU rehearsal of the RCCL path)
 *   "resid0_bits"   1 (default): cg_begin forms the initial residual from the active bits (the initial field is a function of position
 *                   and mask); 0: from x through the stencil -- the same bits (cross-check)
 *   "ring_lines_rccl" homogenization on a ring of GPUs over RCCL: 1 = compacted line lists travel between the slabs (variable-length
 *                   ring exchange), 0 (default) = dense vectors and whole ghost planes.  UNVALIDATED: neither ring exchange has run on
 *                   more than one GPU (no multi-GPU node in any round); the default is the simpler of the two
 *   "place_trials"  single-kernel CG: vector placements timed when a handle is built (default 48; 0 = six slots in
 *                   order, no search; see oi_tort_placement)
 *   "cg1_kernel"    single-kernel CG, compacted layout: 0 = k_cg1b (descriptor-addressed accesses, default), 1 = k_cg1
 *                   (branches around plain global accesses; same iterates bit for bit, kept as the cross-check)
 *   "place_slots"   slots allocated for the search (default 12, 6 to 16: six are kept, the others freed after the search;
 *                   fewer are taken when less than a quarter of the card would stay free)
 *   "place_extent_gb" extent the set of allocations should span in HBM (default 60): smaller sets get a spacer between
 *                   the halves of the slots, allocated and freed again at set-up, of at most "place_spacer_gb" (default 48;
 *                   0 = never a spacer)
 *   "place_alloc_budget_s" spare slots and the spacer are allocated only while the allocations of the set have taken less
 *                   than this (default 0.5 s): the driver clears memory another process has used when it is allocated, and
 *                   the x + six slots a handle needs anyway may already have cost seconds on a card that was full before
 *   "place_retry_gbps" fp64 vectors: when the kept placement's kernel moves fewer algorithmic GB/s than this (default 5150; one byte per
 *                   voxel + 60 per voxel of a live line), or a plain stream of the kept vectors fewer than 5800, the set of allocations
 *                   is taken to be one of the slow ones (where the driver put it, not which slots were chosen): a second set is
 *                   allocated while the first is held, searched, and the better of the two kept.  0 = the kernel's rate never asks; slabs with
 *                   neighbours are judged by the stream alone
 *   "place_min_mb"  vectors below this size are not searched for (default 256: they sit in the Infinity Cache)
 *   "place_cache"   1 (default): a handle reuses the placement an earlier handle of this process found for the same set of
 *                   allocations after verifying it with one timed cycle
 *   "mg_nu_coarse"  damped-Jacobi sweeps before / after the correction on the coarse levels: 1, 2, or 0 (default) = by the structure
 *                   (two where fewer than 35 % of the stored lines hold a free voxel, i.e. at low porosity)
 *   "mg_omega"      Jacobi damping of the cycle's smoothers (default 0.9)
 *   cross-checks of the multigrid-preconditioned iteration, same iterates bit for bit: "mg_norot" (1: residual update and
 *                   pre-smoother as two kernels), "fused_kernel" (1: k_fused_a as first kernel), "mg_post_kernel" (1: the fine
 *                   post-smoother as k_mg_fine), "mg_xfirst" (1: x += alpha p in the first kernel, not in the post-smoother),
 *                   "mg_coarsest_plain" (1: the coarsest level's sweeps as separate launches), "mg_xpair" (0: the x update inside the
 *                   post-smoother every iteration instead of two terms every second one), "mg_blocked" (0: pre-smoothing residual and
 *                   restriction of the coarse levels as two kernels instead of one), "mg_coarse_tiled" (0: correction + post-smoothing of the
 *                   big coarse levels without the LDS tile; 2: the tile on every four-cell level), "mg_fused_images" (0: periodic levels keep
 *                   their image-plane copies although the kernels write the images); of the single-kernel iteration:
 *                   "cg1_seg" (0: per-line slot table instead of segment records), "cg1_fin_sliced" (0: final sums by one block)
 *   "dense_nt"      bit mask (default 31): nontemporal stores of p (1), x (2), q (4), r (8) and the smoother's output (16) in the
 *                   kernels of the dense layouts (homogenization, two-kernel and unfused variants); 0 = the default policy (A/B)
 *   "cg1_nt"        1 (default): the streamed vectors of the descriptor-addressed kernels take nontemporal stores / loads (same
 *                   arithmetic, other cache policy); 0 = the default policy everywhere (A/B)
 * The environment variables OI_TILE_MAP, OI_ZCHUNK, OI_APPLY_KERNEL, OI_MG_PLAIN, OI_MG_NOVEC4, OI_MG_DEBUG,
 * OI_MG_REPLICATE_CELLS, OI_MG_WMASK, OI_MG_SCALE, OI_FORCE_COMM, OI_PLACE_TRIALS, OI_PLACE_SLOTS, OI_PLACE_SPACER_GB, OI_PLACE_EXTENT_GB, OI_PLACE_MIN_MB, OI_PLACE_CACHE seed the same defaults, read ONCE per process.
 * OI_COMM_PRIO=greatest|least (read when a slab handle is built): the ghost-plane exchange stream is made with that stream priority
 * instead of as an ordinary stream (measured 7-8 % slower on a one-rank rehearsal; for experiments on a multi-GPU node).
 * OI_BAD_ARG for an unknown key. */
oi_status oi_tort_set_option(const char* key, double value);
/* copy the current process defaults of the launch shapes into a live handle (layout, multigrid hierarchy and the
 * capacity of its partial-sum array stay as built) */
oi_status oi_tort_retune(oi_tort* h);

/* The solver kernels predicate their vector loads/stores on the code byte, so a cache line without a
 * free voxel is never moved.  Fraction of the aligned runs of `group_voxels` voxels along x (this slab)
 * that hold at least one free voxel -- 16 doubles = one 128-byte cache line (what dense, predicated vectors
 * move), oi_tort_line_voxels() = what the compacted vectors store; used by bench.py to state the
 * bytes the kernel cannot avoid. */
oi_status oi_tort_live_fraction(oi_tort* h, int group_voxels, double* fraction);

/* per-kernel HIP-event timings gathered while opts.profile = 1.  names: fixed list,
 * see oi_kernel_name().  Returns the number of kernels written (<= cap). */
int oi_tort_kernel_times(oi_tort* h, int cap, int* kernel_ids, double* total_ms, int64_t* launches);
const char* oi_kernel_name(int kernel_id);
void oi_tort_reset_kernel_times(oi_tort* h);

/* ---- communicator: one process per GPU, RCCL over xGMI ----
 * Bootstrap as with MPI_Init in the reference (Diffusion.cpp:173-174) but host-agnostic:
 * rank 0 obtains 128 opaque bytes, the host moves them to every rank by any channel
 * (torch.distributed, MPI, a file), every rank calls oi_comm_create. */
#define OI_COMM_ID_BYTES 128
oi_status oi_comm_unique_id(void* id_out);
oi_status oi_comm_create(oi_comm** out, const void* id, int nranks, int rank, int device);
/* `nranks` linked in-process communicators that run as `nranks` host threads on ONE
 * device (device-to-device copies instead of RCCL): exercises every slab code path
 * on a single-GPU box. */
oi_status oi_comm_create_local(oi_comm** out_array, int nranks, int device);
void oi_comm_destroy(oi_comm* c);
int oi_comm_rank(const oi_comm* c);
int oi_comm_size(const oi_comm* c);
/* What the transport really is (a multi-GPU bench line reports it, so that a run is self-describing): kind 1 = RCCL, 2 = in-process;
 * ranks_collective / ranks_p2p = ncclCommCount of the all-reduce communicator and of the plane-exchange communicator (RCCL; the rank
 * count of the group otherwise); exchange_overlaps = 1 when the plane exchange runs on its own stream beside the
 * scalar all-reduce (it has a communicator of its own), 0 when one communicator serves both and everything stays on one stream. */
oi_status oi_comm_info(const oi_comm* c, int* kind, int* ranks_collective, int* ranks_p2p, int* exchange_overlaps);

/* ---- seeded synthetic microstructures, generated in HBM (BASELINE.json configs) ----
 * phase_dev: device pointer to nx*ny*nz_local bytes holding global planes
 * [z0, z0+nz_local).  Bit-identical to the oracle's generators. */
oi_status oi_gen_bernoulli_device(uint8_t* phase_dev, int nx, int ny, int nz_global, int z0,
                                  int nz_local, uint64_t seed, uint64_t thr53);
oi_status oi_gen_spheres_device(uint8_t* phase_dev, int nx, int ny, int nz_global, int z0,
                                int nz_local, uint64_t seed, int nspheres, int radius_half_voxels);

/* correlated low-porosity "blob" structure (config 5): integer trilinear interpolation of
 * lattice noise with `spacing` voxels between lattice points; pore (1) where the value is
 * below `level`.  oi_blobs_level returns the level that gives the wanted porosity. */
oi_status oi_gen_blobs_device(uint8_t* phase_dev, int nx, int ny, int nz_global, int z0, int nz_local,
                              uint64_t seed, int spacing, uint64_t level);
uint64_t oi_blobs_level(uint64_t seed, int spacing, double porosity);

/* ---- device memory helpers for hosts without a GPU array library ---- */
oi_status oi_dev_malloc(void** ptr, size_t bytes);
/* free / total memory of the current device (a volume needs 25-60 bytes per voxel) */
oi_status oi_dev_mem_info(size_t* free_bytes, size_t* total_bytes);
oi_status oi_dev_free(void* ptr);
/* The process's POOL of released handle arrays (off by default).  A process that builds solver object after solver object of the
 * same shape -- the reference's direction loop, Diffusion.cpp:655-706: a new TortuosityHypre per direction; its REV study, :317-502 --
 * pays the driver's clearing of recycled memory again and again on this hardware (one hipMalloc of a few GB: 0.2 ms or seconds).
 * With the pool on, the device arrays of a destroyed handle are kept and the next request of that size on that device takes one
 * back, zeroed; the vectors of the placement search never come from it.  cap_gb > 0 bounds what is kept (0: everything, given back
 * when an allocation fails).  oi_dev_pool(0, 0) and oi_dev_pool_trim return the pooled blocks to the driver (trim: bytes released). */
oi_status oi_dev_pool(int enable, double cap_gb);
double oi_dev_pool_trim(void);
oi_status oi_dev_pool_stats(double* held_bytes, long long* held_blocks, long long* hits, long long* misses, int* enabled);
/* The library's allocation LEDGER: every device allocation it makes (handles, transports, scratch, oi_dev_malloc) is entered when made
 * and struck when freed.  live_bytes / live_blocks = requested bytes / number of the allocations alive now (blocks the pool holds are
 * not alive); peak_bytes = high-water mark of live_bytes since the last reset (reset != 0 restarts it at the current value).  After
 * oi_*_destroy of everything a process built these are back at what they were before: the leak tests assert equality. */
long long oi_dev_live_bytes(void);
long long oi_dev_live_blocks(void);
long long oi_dev_peak_bytes(int reset);
/* test hook: the n-th device allocation from now (1 = the next) fails once, as if the device were full; 0 disarms */
void oi_debug_fail_alloc_after(long long n);
/* measurement aid: ms per launch of the library's reference copy stream (linear, 16-byte nontemporal accesses) between
 * two scratch arrays of `bytes` each -- what a 1 read : 1 write stream reaches on this box; bench.py reports it beside
 * the stencil apply */
oi_status oi_dev_stream_copy_bench(size_t bytes, int reps, double* ms_per_copy);
oi_status oi_dev_upload(void* dst_dev, const void* src_host, size_t bytes);
oi_status oi_dev_download(void* dst_host, const void* src_dev, size_t bytes);
int oi_device_count(void);
const char* oi_last_error(void); /* thread-local text of the last failure */
const char* oi_status_string(oi_status s);
int oi_abi_version(void);
/* 1 in the TEST library (libopenimpala_amd_xcheck.so, `make xcheck`): the superseded kernels kept as bit-identity references of the shipped
 * ones are present (k_cg1: the single-kernel iteration on dense vectors / option cg1_kernel = 1; k_fused_a on compacted vectors: option
 * fused_kernel = 1).  0 in the product library, which answers OI_BAD_ARG to configurations that need them. */
int oi_build_has_xcheck(void);

/* ---- readers (src/io): file -> thresholded 0/1 phase bytes, voxel > thr -> 1 ----
 * oi_read_*: *phase is malloc'ed by the library (release with oi_free_host).
 * TiffReader::threshold (TiffReader.cpp:289-444,449-455), RawReader::threshold
 * (RawReader.cpp:379-491), HDF5Reader::threshold (HDF5Reader.cpp:333-402). */
/* on != 0: the readers return PAGE-LOCKED buffers from now on (hipHostMalloc), so that the upload of a volume or of a
 * z-slab is one DMA transfer; oi_free_host releases either kind.  Without a device the readers keep returning malloc'ed
 * memory. */
void oi_read_set_pinned(int on);
oi_status oi_read_tiff(const char* path, double threshold, uint8_t** phase, int* nx, int* ny, int* nz,
                       int* bits_per_sample);
/* planes [z0, z0 + nz_local) only (nz_local < 0: all from z0); *nz = depth of the FILE.  The file is
 * memory-mapped, so a rank loading its z-slab of a large stack touches only those planes
 * (slab-wise load instead of a whole-volume host staging buffer). */
oi_status oi_read_tiff_planes(const char* path, double threshold, int z0, int nz_local, uint8_t** phase, int* nx, int* ny,
                              int* nz, int* bits_per_sample);
/* numbered single-slice files <base><index zero-padded to digits><suffix>, index = start_index ...
 * start_index + num_files - 1: TiffReader::readFileSequence (TiffReader.H:117-123, TiffReader.cpp:85-88,209-285) */
oi_status oi_read_tiff_sequence(const char* base_pattern, int num_files, int start_index, int digits, const char* suffix,
                                double threshold, uint8_t** phase, int* nx, int* ny, int* nz, int* bits_per_sample);
/* dtype: OI_RAW_* below, same numbering as the reference's RawDataType (RawReader.H:30-46) */
enum {
  OI_RAW_UNKNOWN = 0, OI_RAW_UINT8 = 1, OI_RAW_INT8, OI_RAW_INT16_LE, OI_RAW_INT16_BE, OI_RAW_UINT16_LE, OI_RAW_UINT16_BE,
  OI_RAW_INT32_LE, OI_RAW_INT32_BE, OI_RAW_UINT32_LE, OI_RAW_UINT32_BE, OI_RAW_FLOAT32_LE,
  OI_RAW_FLOAT32_BE, OI_RAW_FLOAT64_LE, OI_RAW_FLOAT64_BE
};
oi_status oi_read_raw(const char* path, int nx, int ny, int nz, int dtype, double threshold,
                      uint8_t** phase);
oi_status oi_read_raw_planes(const char* path, int nx, int ny, int nz, int dtype, double threshold, int z0, int nz_local,
                             uint8_t** phase);
/* HDF5Reader::threshold (HDF5Reader.cpp:111-187,333-402): rank-3 dataset `dataset` (e.g. "image" or "g1/g2/image"),
 * dims (Z,Y,X) in C order.  Read straight from the file format (no libhdf5): the default ("earliest") format AND the
 * "latest" format of HDF5 >= 1.10 / h5py libver='latest' -- superblock v0-v3, v1 / v2 object headers, symbol-table and
 * compact-link groups, contiguous / compact / chunked layouts with v1 B-tree, single-chunk, implicit, fixed-array and
 * extensible-array chunk indexes (filtered, paged), deflate + shuffle, (u)int8-64 / float32 / float64 of either byte
 * order.  Refused with a message that names the feature: dense (fractal-heap) groups, v2-B-tree chunk indexes (two or
 * more unlimited dimensions), virtual datasets, other filters. */
oi_status oi_read_hdf5(const char* path, const char* dataset, double threshold, uint8_t** phase, int* nx, int* ny,
                       int* nz);
/* planes [z0, z0 + nz_local) only (nz_local < 0: all from z0): the hyperslab a rank reads (HDF5Reader.cpp:255-331) */
oi_status oi_read_hdf5_planes(const char* path, const char* dataset, double threshold, int z0, int nz_local,
                              uint8_t** phase, int* nx, int* ny, int* nz);
/* DatReader::threshold (DatReader.cpp:74-160,204-241): 3 x int32 header + uint16 voxels */
oi_status oi_read_dat(const char* path, double threshold, uint8_t** phase, int* nx, int* ny, int* nz);
void oi_free_host(void* p);

#ifdef __cplusplus
}
#endif
#endif /* OPENIMPALA_AMD_H */
