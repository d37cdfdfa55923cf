/*
 * oi_oracle.c -- CPU restatement (TEST INFRASTRUCTURE, not product code) of the
 * flow-through tortuosity path of kramergroup/openImpala.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this file's library.  The product path (openimpala_amd/csrc) never links,
 * imports or calls anything in oracle/.
 *
 * Pinning status: PARITY UNPINNED (numerically).  The reference's own tests hold no
 * number for this path (tests/inputs/tTortuosity.inputs:28-32 and
 * tVolumeFraction.inputs:53-54 carry none) and the reference cannot be built in this
 * image (AMReX 25.03, HYPRE v2.32.0 and AMReX's Fortran modules are absent; no
 * stand-ins are written), so there is no oracle/_ref and no reference-held value to
 * check this restatement against.  What it IS checked against (tests/test_oracle.py,
 * tests/test_reference_fixtures.py):
 *   (1) the reference's own structural tests, restated and run on its own data files:
 *       checkMatrixProperties on the materialised rows (TortuosityHypre.cpp:896-982), the
 *       1e-6 flux-conservation gate (:794-804), "tau is finite" (tTortuosity.cpp:304),
 *       counts == direct recount and VF0 + VF1 = 1 (tVolumeFraction.cpp:413,441,449),
 *       reader dimensions / {0, 1} after thresholding (tRawReader.cpp, tHDF5Reader.cpp,
 *       tTiffReader.cpp), tensor symmetry and sign (tEffectiveDiffusivity.cpp:423-445);
 *   (2) the analytic answers the reference's discretisation implies (all-pore cube
 *       tau = (N-1)/N, straight channel, blocked plane -> NaN);
 *   (3) tests/golden/sample_goldens.json -- the table of BASELINE.md section 2.  It was
 *       produced by the survey stage, which assembled the system with the reference's
 *       Fortran kernel compiled behind hand-written stand-ins for AMReX's modules; that
 *       is NOT the reference built here, no script in this repository regenerates it,
 *       and it therefore counts as a cross-check between two restatements, not as a pin.
 *       This file reproduces every entry of it to <= 2e-9.
 *
 * Every function cites the reference file:line it restates (paths relative to
 * the reference tree).  Layout everywhere: x fastest, idx = i + nx*(j + ny*k),
 * identical to the AMReX/Fortran box order and to the TIFF/RAW payload order.
 *
 * Build: see oracle/Makefile (gcc -O2 -fopenmp -ffp-contract=off -shared).
 */
#include <math.h>
#include <float.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef long long i64;

#define IDX(i, j, k) ((i64)(i) + (i64)nx * ((i64)(j) + (i64)ny * (i64)(k)))

int oi_oracle_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

void oi_oracle_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

/* ------------------------------------------------------------------------- */
/* Seeded synthetic microstructures (spec shared with the HIP generators; the   */
/* reference's data/create_sample_structure.py needs porespy and is unseeded).  */
/* Pure integer arithmetic so CPU and GPU agree bit for bit.                    */
/* ------------------------------------------------------------------------- */
static inline uint64_t oi_mix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

static inline uint64_t oi_hash(uint64_t seed, uint64_t ctr) {
  return oi_mix64(seed * 0xD1342543DE82EF95ULL + ctr);
}

/* voxel = 1 with probability thr53 / 2^53, iid per linear voxel index */
void oi_oracle_gen_bernoulli(uint8_t* phase, int nx, int ny, int nz, uint64_t seed,
                             uint64_t thr53) {
  const i64 n = (i64)nx * ny * nz;
#pragma omp parallel for schedule(static)
  for (i64 v = 0; v < n; ++v) phase[v] = ((oi_hash(seed, (uint64_t)v) >> 11) < thr53) ? 1 : 0;
}

/* Overlapping spheres: centre s at half-voxel coordinates (h % 2n) per axis,
 * radius r2h half-voxels; voxel (i,j,k) centre = (2i+1, 2j+1, 2k+1).  Solid = 0,
 * pore = 1.  Brute force over spheres: the oracle only runs small cases. */
void oi_oracle_gen_spheres(uint8_t* phase, int nx, int ny, int nz, uint64_t seed,
                           int nspheres, int r2h) {
  int* c = (int*)malloc(sizeof(int) * 3 * (size_t)(nspheres > 0 ? nspheres : 1));
  for (int s = 0; s < nspheres; ++s) {
    c[3 * s + 0] = (int)(oi_hash(seed, 3ULL * s + 0) % (uint64_t)(2 * nx));
    c[3 * s + 1] = (int)(oi_hash(seed, 3ULL * s + 1) % (uint64_t)(2 * ny));
    c[3 * s + 2] = (int)(oi_hash(seed, 3ULL * s + 2) % (uint64_t)(2 * nz));
  }
  const i64 rr = (i64)r2h * r2h;
#pragma omp parallel for collapse(2) schedule(static)
  for (int k = 0; k < nz; ++k)
    for (int j = 0; j < ny; ++j)
      for (int i = 0; i < nx; ++i) {
        uint8_t pore = 1;
        for (int s = 0; s < nspheres; ++s) {
          i64 dx = 2 * i + 1 - c[3 * s], dy = 2 * j + 1 - c[3 * s + 1], dz = 2 * k + 1 - c[3 * s + 2];
          if (dx * dx + dy * dy + dz * dz < rr) { pore = 0; break; }
        }
        phase[IDX(i, j, k)] = pore;
      }
  free(c);
}

/* Correlated "blob" structure for the low-porosity configuration (BASELINE.json config 5):
 * trilinear interpolation, in pure integer arithmetic, of 16-bit lattice noise with
 * spacing S voxels; voxel = 1 (pore) where the interpolated value is below `level`
 * (0..65535).  iid noise at porosity 0.2 would not percolate (site threshold 0.3116). */
static inline uint64_t blob_value(uint64_t seed, int S, int i, int j, int k) {
  const int a = i / S, b = j / S, c = k / S;
  const uint64_t w1[3] = {(uint64_t)(2 * (i % S) + 1), (uint64_t)(2 * (j % S) + 1), (uint64_t)(2 * (k % S) + 1)};
  uint64_t num = 0;
  for (int dc = 0; dc < 2; ++dc)
    for (int db = 0; db < 2; ++db)
      for (int da = 0; da < 2; ++da) {
        const uint64_t key = ((uint64_t)(uint32_t)(a + da)) | ((uint64_t)(uint32_t)(b + db) << 21) |
                             ((uint64_t)(uint32_t)(c + dc) << 42);
        const uint64_t v = oi_hash(seed, key) >> 48;
        const uint64_t wx = da ? w1[0] : 2 * S - w1[0], wy = db ? w1[1] : 2 * S - w1[1],
                       wz = dc ? w1[2] : 2 * S - w1[2];
        num += wx * wy * wz * v;
      }
  return num; /* = value * (2S)^3 */
}

void oi_oracle_gen_blobs(uint8_t* phase, int nx, int ny, int nz, uint64_t seed, int S, uint64_t level) {
  const uint64_t thr = level * (uint64_t)(2 * S) * (2 * S) * (2 * S);
#pragma omp parallel for collapse(2) schedule(static)
  for (int k = 0; k < nz; ++k)
    for (int j = 0; j < ny; ++j)
      for (int i = 0; i < nx; ++i) phase[IDX(i, j, k)] = blob_value(seed, S, i, j, k) < thr ? 1 : 0;
}

/* ------------------------------------------------------------------------- */
/* VolumeFraction::value   (src/props/VolumeFraction.cpp:22-66)                */
/* ------------------------------------------------------------------------- */
void oi_oracle_volume_fraction(const uint8_t* phase, i64 n, int phase_id, i64* phase_count,
                               i64* total_count) {
  i64 pc = 0;
#pragma omp parallel for reduction(+ : pc) schedule(static)
  for (i64 v = 0; v < n; ++v) pc += (phase[v] == phase_id);
  *phase_count = pc;
  *total_count = n;
}

/* ------------------------------------------------------------------------- */
/* tortuosity_remspot   (src/props/Tortuosity_filcc.F90:88-177), one pass over  */
/* the whole domain as ONE box: in place, x-fastest (Gauss-Seidel) order, a     */
/* neighbour outside the domain never matches.  (The reference applies it per   */
/* 32^3 tile, so its result depends on box_size/OpenMP tiling; dormant by       */
/* default: tortuosity.remspot_passes = 0, TortuosityHypre.cpp:254-263.)        */
/* ------------------------------------------------------------------------- */
void oi_oracle_remspot(uint8_t* q, int nx, int ny, int nz) {
  for (int k = 0; k < nz; ++k)
    for (int j = 0; j < ny; ++j)
      for (int i = 0; i < nx; ++i) {
        const uint8_t c = q[IDX(i, j, k)];
        int connected = 0;
        if (i > 0 && q[IDX(i - 1, j, k)] == c) connected = 1;
        if (i < nx - 1 && q[IDX(i + 1, j, k)] == c) connected = 1;
        if (j > 0 && q[IDX(i, j - 1, k)] == c) connected = 1;
        if (j < ny - 1 && q[IDX(i, j + 1, k)] == c) connected = 1;
        if (k > 0 && q[IDX(i, j, k - 1)] == c) connected = 1;
        if (k < nz - 1 && q[IDX(i, j, k + 1)] == c) connected = 1;
        if (!connected) q[IDX(i, j, k)] = (c == 0) ? 1 : 0;
      }
}

/* ------------------------------------------------------------------------- */
/* The same filter AS THE REFERENCE RUNS IT (preconditionPhaseFab,                */
/* src/props/TortuosityHypre.cpp:264-289): per pass, FillBoundary, then every     */
/* box of the BoxArray -- Diffusion.cpp:266-268: ba.maxSize(box_size), box_size   */
/* 32 by default -- is filtered IN PLACE in its own fab, tile by tile             */
/* (MFIter with TilingIfNotGPU; AMReX's default tile is 1024000 x 8 x 8 cells).   */
/* A cell therefore sees the NEW value of a neighbour that lies in the SAME box   */
/* and was visited earlier, and the value from BEFORE the pass (the fab's ghost   */
/* cells) of a neighbour in another box.  One OpenMP thread: with several, tiles  */
/* of a box run concurrently and the reference's own result is a race.            */
/* Box cuts: AMReX 25.03's BoxList::maxSize (third party, not in this image) as   */
/* published in Src/Base/AMReX_BoxList.cpp: per axis, halve chunk and length     */
/* while both are even (ratio), numblk = ceil(len / chunk), sz = len / numblk,    */
/* extra = len % numblk; block b spans [(b < extra ? b (sz+1) : b sz + extra),    */
/* ...) x ratio -- the larger blocks first.  (Older AMReX releases chopped from   */
/* the high end, i.e. larger blocks last; the two agree whenever box_size divides */
/* the extent.)                                                                   */
/* ------------------------------------------------------------------------- */
int oi_oracle_box_cuts(int n, int chunk, int* first /* >= n + 1 ints: first cell of each block, then n */) {
  if (chunk <= 0 || n <= chunk) { first[0] = 0; first[1] = n; return 1; }
  int ratio = 1, bs = chunk, nlen = n;
  while ((bs % 2 == 0) && (nlen % 2 == 0)) { ratio *= 2; bs /= 2; nlen /= 2; }
  const int numblk = (nlen + bs - 1) / bs, sz = nlen / numblk, extra = nlen % numblk;
  for (int b = 0; b <= numblk; ++b) first[b] = ((b < extra) ? b * (sz + 1) : b * sz + extra) * ratio;
  return numblk;
}

void oi_oracle_remspot_boxes(uint8_t* q, int nx, int ny, int nz, int box_size) {
  const i64 n = (i64)nx * ny * nz;
  uint8_t* old = (uint8_t*)malloc((size_t)n); /* what FillBoundary put into every fab's ghost cells */
  memcpy(old, q, (size_t)n);
  int* cx = (int*)malloc(sizeof(int) * (size_t)(nx + 2));
  int* cy = (int*)malloc(sizeof(int) * (size_t)(ny + 2));
  int* cz = (int*)malloc(sizeof(int) * (size_t)(nz + 2));
  const int nbx = oi_oracle_box_cuts(nx, box_size, cx), nby = oi_oracle_box_cuts(ny, box_size, cy),
            nbz = oi_oracle_box_cuts(nz, box_size, cz);
  const int tile = 8; /* FabArrayBase::mfiter_tile_size = (1024000, 8, 8) */
  for (int bz = 0; bz < nbz; ++bz)
    for (int by = 0; by < nby; ++by)
      for (int bx = 0; bx < nbx; ++bx) {
        const int x0 = cx[bx], x1 = cx[bx + 1], y0 = cy[by], y1 = cy[by + 1], z0 = cz[bz], z1 = cz[bz + 1];
        /* value of a neighbour as this box's fab holds it: in place inside the box, ghost (pre-pass) outside */
#define FABVAL(i, j, k) (((i) >= x0 && (i) < x1 && (j) >= y0 && (j) < y1 && (k) >= z0 && (k) < z1) ? q[IDX(i, j, k)] : old[IDX(i, j, k)])
        for (int tz = z0; tz < z1; tz += tile)
          for (int ty = y0; ty < y1; ty += tile)
            for (int k = tz; k < tz + tile && k < z1; ++k)
              for (int j = ty; j < ty + tile && j < y1; ++j)
                for (int i = x0; i < x1; ++i) {
                  const uint8_t c = q[IDX(i, j, k)];
                  int connected = 0;
                  if (i > 0 && FABVAL(i - 1, j, k) == c) connected = 1;
                  if (i < nx - 1 && FABVAL(i + 1, j, k) == c) connected = 1;
                  if (j > 0 && FABVAL(i, j - 1, k) == c) connected = 1;
                  if (j < ny - 1 && FABVAL(i, j + 1, k) == c) connected = 1;
                  if (k > 0 && FABVAL(i, j, k - 1) == c) connected = 1;
                  if (k < nz - 1 && FABVAL(i, j, k + 1) == c) connected = 1;
                  if (!connected) q[IDX(i, j, k)] = (c == 0) ? 1 : 0;
                }
#undef FABVAL
      }
  free(old); free(cx); free(cy); free(cz);
}

/* ------------------------------------------------------------------------- */
/* parallelFloodFill   (src/props/TortuosityHypre.cpp:297-389): 6-connected      */
/* reachability from the seed cells, restricted to phase == phase_id.  The      */
/* reference sweeps to a fixed point (<= nx+ny+nz+2 sweeps); the fixed point is */
/* the connected set, computed here by a queue.                                 */
/* ------------------------------------------------------------------------- */
static void flood_from_face(const uint8_t* phase, int nx, int ny, int nz, int phase_id, int dir,
                            int hi_face, uint8_t* reach, i64* queue, i64* nseeds) {
  const i64 n = (i64)nx * ny * nz;
  memset(reach, 0, (size_t)n);
  i64 head = 0, tail = 0;
  const int nd[3] = {nx, ny, nz};
  const int fixed = hi_face ? nd[dir] - 1 : 0;
  /* seeds: cells of the phase on the face (TortuosityHypre.cpp:424-449) */
  for (int k = 0; k < nz; ++k)
    for (int j = 0; j < ny; ++j)
      for (int i = 0; i < nx; ++i) {
        const int c[3] = {i, j, k};
        if (c[dir] != fixed) continue;
        const i64 v = IDX(i, j, k);
        if (phase[v] == phase_id) { reach[v] = 1; queue[tail++] = v; }
      }
  *nseeds = tail;
  const i64 sx = 1, sy = nx, sz = (i64)nx * ny;
  while (head < tail) {
    const i64 v = queue[head++];
    const int i = (int)(v % nx), j = (int)((v / nx) % ny), k = (int)(v / sz);
#define TRY(cond, w)                                                        \
  if ((cond) && !reach[(w)] && phase[(w)] == phase_id) { reach[(w)] = 1; queue[tail++] = (w); }
    TRY(i > 0, v - sx) TRY(i < nx - 1, v + sx) TRY(j > 0, v - sy) TRY(j < ny - 1, v + sy)
    TRY(k > 0, v - sz) TRY(k < nz - 1, v + sz)
#undef TRY
  }
}

/* generateActivityMask   (src/props/TortuosityHypre.cpp:394-558): mask = cells
 * reached from the lo face AND from the hi face of the flow axis; empty when
 * either face holds no cell of the phase (:508-514).  Returns #active. */
i64 oi_oracle_active_mask(const uint8_t* phase, int nx, int ny, int nz, int phase_id, int dir,
                          uint8_t* mask) {
  const i64 n = (i64)nx * ny * nz;
  uint8_t* rlo = (uint8_t*)malloc((size_t)n);
  uint8_t* rhi = (uint8_t*)malloc((size_t)n);
  i64* queue = (i64*)malloc(sizeof(i64) * (size_t)n);
  i64 nlo = 0, nhi = 0, nact = 0;
  flood_from_face(phase, nx, ny, nz, phase_id, dir, 0, rlo, queue, &nlo);
  flood_from_face(phase, nx, ny, nz, phase_id, dir, 1, rhi, queue, &nhi);
  if (nlo == 0 || nhi == 0) {
    memset(mask, 0, (size_t)n);
  } else {
    for (i64 v = 0; v < n; ++v) { mask[v] = (uint8_t)(rlo[v] & rhi[v]); nact += mask[v]; }
  }
  free(rlo); free(rhi); free(queue);
  return nact;
}

/* ------------------------------------------------------------------------- */
/* tortuosity_fillmtx   (src/props/TortuosityHypreFill.F90:44-314) for the      */
/* whole domain as one box.  a[7*m + s], s = C,-x,+x,-y,+y,-z,+z (:20-26),       */
/* m = i + nx*(j + ny*k) (:101).  Phase/mask outside the domain count as         */
/* inactive: the mask's ghost cells are cell_inactive because the tortuosity    */
/* geometry is non-periodic (TortuosityHypre.cpp:309,522; Diffusion.cpp:671-677).*/
/* ------------------------------------------------------------------------- */
void oi_oracle_fillmtx(const uint8_t* phase, const uint8_t* mask, int nx, int ny, int nz,
                       int phase_id, int dir, const double* dxinv /* [3] = 1/dx^2 */, double vlo,
                       double vhi, double* a, double* rhs, double* xinit) {
  const double small_real = 1.0e-15; /* F90:34 */
  const int nd[3] = {nx, ny, nz};
#pragma omp parallel for collapse(2) schedule(static)
  for (int k = 0; k < nz; ++k)
    for (int j = 0; j < ny; ++j)
      for (int i = 0; i < nx; ++i) {
        const i64 m = IDX(i, j, k);
        double* row = a + 7 * m;
        for (int s = 0; s < 7; ++s) row[s] = 0.0;
        if (phase[m] != phase_id || mask[m] == 0) { /* F90:111-118 */
          row[0] = 1.0; rhs[m] = 0.0; xinit[m] = 0.0;
          continue;
        }
        double diag = 0.0; /* F90:121-169 */
#define NB(cond, w, s, ax)                                                   \
  if ((cond) && phase[(w)] == phase_id && mask[(w)] == 1) { row[s] = -dxinv[ax]; diag += dxinv[ax]; }
        NB(i > 0, m - 1, 1, 0) NB(i < nx - 1, m + 1, 2, 0)
        NB(j > 0, m - nx, 3, 1) NB(j < ny - 1, m + nx, 4, 1)
        NB(k > 0, m - (i64)nx * ny, 5, 2) NB(k < nz - 1, m + (i64)nx * ny, 6, 2)
#undef NB
        row[0] = diag;
        if (fabs(diag) < small_real) { /* F90:172-181: isolated active cell decoupled */
          for (int s = 0; s < 7; ++s) row[s] = 0.0;
          row[0] = 1.0; rhs[m] = 0.0; xinit[m] = 0.0;
          continue;
        }
        rhs[m] = 0.0; /* F90:184 */
        const int c[3] = {i, j, k};
        int on_dirichlet = 0; /* F90:191-228 */
        if (c[dir] == 0) {
          for (int s = 0; s < 7; ++s) row[s] = 0.0;
          row[0] = 1.0; rhs[m] = vlo; on_dirichlet = 1;
        } else if (c[dir] == nd[dir] - 1) {
          for (int s = 0; s < 7; ++s) row[s] = 0.0;
          row[0] = 1.0; rhs[m] = vhi; on_dirichlet = 1;
        }
        /* F90:233-262: linear ramp initial guess */
        if (fabs(row[0] - 1.0) > small_real || on_dirichlet) {
          const double extent = (double)(nd[dir] - 1);
          const double factor = (fabs(extent) < small_real) ? 0.0 : 1.0 / extent;
          xinit[m] = vlo + (vhi - vlo) * (double)c[dir] * factor;
        } else {
          /* An active non-Dirichlet cell whose diagonal is exactly 1 (one active
           * neighbour) is skipped by the test at F90:233, so the reference leaves
           * xinit(m_idx) unassigned: it keeps what the caller's scratch vector held
           * (value-initialised 0 on a thread's first tile, TortuosityHypre.cpp:606).
           * Restated as 0; it is only a starting guess and does not change the
           * solution of the system. */
          xinit[m] = 0.0;
        }
      }
}

/* TortuosityHypre::checkMatrixProperties   (src/props/TortuosityHypre.cpp:896-982)
 * on materialised rows.  Returns the number of violated checks (0 = pass). */
i64 oi_oracle_check_matrix(const double* a, const double* rhs, const uint8_t* mask, int nx,
                           int ny, int nz, int dir, double vlo, double vhi) {
  const double tol = 1.e-14; /* :903 */
  const int nd[3] = {nx, ny, nz};
  i64 bad = 0;
  for (int k = 0; k < nz; ++k)
    for (int j = 0; j < ny; ++j)
      for (int i = 0; i < nx; ++i) {
        const i64 m = IDX(i, j, k);
        const double* row = a + 7 * m;
        const double b = rhs[m];
        int naninf = isnan(b) || isinf(b);
        for (int s = 0; s < 7; ++s) naninf |= isnan(row[s]) || isinf(row[s]);
        if (naninf) ++bad; /* :945-947 */
        const int c[3] = {i, j, k};
        const int active = mask[m] == 1;
        const int dirichlet = active && (c[dir] == 0 || c[dir] == nd[dir] - 1); /* :950-956 */
        if (!active) { /* :957-960 */
          if (fabs(row[0] - 1.0) > tol || fabs(b) > tol) ++bad;
          for (int s = 1; s < 7; ++s) if (fabs(row[s]) > tol) ++bad;
        } else if (dirichlet) { /* :961-965 */
          const double eb = (c[dir] == 0) ? vlo : vhi;
          if (fabs(row[0] - 1.0) > tol || fabs(b - eb) > tol) ++bad;
          for (int s = 1; s < 7; ++s) if (fabs(row[s]) > tol) ++bad;
        } else { /* :966-972 */
          if (row[0] <= tol) ++bad;
          if (fabs(b) > tol) ++bad;
          double sum = 0.0;
          for (int s = 0; s < 7; ++s) sum += row[s];
          if (fabs(sum) > tol) ++bad;
        }
      }
  return bad;
}

/* ------------------------------------------------------------------------- */
/* Solve of the ASSEMBLED system (what HYPRE is handed,                        */
/* TortuosityHypre.cpp:635-647,681-683).  Identity rows (Aii = 1, no            */
/* off-diagonals) pin x = rhs; their columns are folded into the right-hand     */
/* side of the remaining rows, which leaves an SPD block solved by (Jacobi-)CG. */
/* Convergence as the reference reads it from HYPRE (:685-688):                  */
/* ||b - A x||_2 / ||b||_2 <= eps over the FULL system (identity rows have zero  */
/* residual, so the numerator equals the reduced residual norm).                */
/* Returns 1 if converged.                                                      */
/* ------------------------------------------------------------------------- */
static inline int is_identity_row(const double* row) {
  return row[0] == 1.0 && row[1] == 0.0 && row[2] == 0.0 && row[3] == 0.0 && row[4] == 0.0 &&
         row[5] == 0.0 && row[6] == 0.0;
}

int oi_oracle_solve_assembled(const double* a, const double* rhs, const double* xinit, int nx,
                              int ny, int nz, double eps, int maxiter, int jacobi, double* x,
                              int* iters_out, double* relres_out) {
  const i64 n = (i64)nx * ny * nz;
  const i64 off[7] = {0, -1, 1, -nx, nx, -(i64)nx * ny, (i64)nx * ny};
  uint8_t* fr = (uint8_t*)malloc((size_t)n);
  double* r = (double*)calloc((size_t)n, sizeof(double));
  double* p = (double*)calloc((size_t)n, sizeof(double));
  double* q = (double*)calloc((size_t)n, sizeof(double));
  double bb = 0.0;
#pragma omp parallel for reduction(+ : bb) schedule(static)
  for (i64 m = 0; m < n; ++m) {
    fr[m] = !is_identity_row(a + 7 * m);
    x[m] = fr[m] ? xinit[m] : rhs[m];
    bb += rhs[m] * rhs[m];
  }
  const double bnorm = sqrt(bb);
  /* r = b - A x on free rows (x holds the pinned values on identity rows) */
  double rr = 0.0, rz = 0.0;
#pragma omp parallel for reduction(+ : rr, rz) schedule(static)
  for (i64 m = 0; m < n; ++m) {
    if (!fr[m]) continue;
    const double* row = a + 7 * m;
    double s = row[0] * x[m];
    for (int t = 1; t < 7; ++t) if (row[t] != 0.0) s += row[t] * x[m + off[t]];
    r[m] = rhs[m] - s;
    const double z = jacobi ? r[m] / row[0] : r[m];
    p[m] = z;
    rr += r[m] * r[m];
    rz += r[m] * z;
  }
  int it = 0, conv = 0;
  double relres = (bnorm > 0.0) ? sqrt(rr) / bnorm : sqrt(rr);
  if (relres <= eps) conv = 1;
  while (!conv && it < maxiter) {
    double pq = 0.0;
#pragma omp parallel for reduction(+ : pq) schedule(static)
    for (i64 m = 0; m < n; ++m) {
      if (!fr[m]) continue;
      const double* row = a + 7 * m;
      double s = row[0] * p[m];
      for (int t = 1; t < 7; ++t) if (row[t] != 0.0) s += row[t] * p[m + off[t]]; /* p = 0 on pinned rows */
      q[m] = s;
      pq += p[m] * s;
    }
    const double alpha = rz / pq;
    double rr_new = 0.0, rz_new = 0.0;
#pragma omp parallel for reduction(+ : rr_new, rz_new) schedule(static)
    for (i64 m = 0; m < n; ++m) {
      if (!fr[m]) continue;
      x[m] += alpha * p[m];
      r[m] -= alpha * q[m];
      const double z = jacobi ? r[m] / a[7 * m] : r[m];
      rr_new += r[m] * r[m];
      rz_new += r[m] * z;
    }
    ++it;
    relres = (bnorm > 0.0) ? sqrt(rr_new) / bnorm : sqrt(rr_new);
    if (relres <= eps) { conv = 1; break; }
    const double beta = rz_new / rz;
#pragma omp parallel for schedule(static)
    for (i64 m = 0; m < n; ++m) {
      if (!fr[m]) continue;
      const double z = jacobi ? r[m] / a[7 * m] : r[m];
      p[m] = z + beta * p[m];
    }
    rr = rr_new; rz = rz_new;
  }
  *iters_out = it;
  *relres_out = relres;
  free(fr); free(r); free(p); free(q);
  return conv && isfinite(relres);
}

/* ------------------------------------------------------------------------- */
/* Matrix-free form of the same system.  code[m] = number of       */
/* active 6-neighbours if cell m is FREE (active and not on a Dirichlet face),   */
/* else 0.  With v = 0 on every inactive cell:                                   */
/*     (A v)[m] = code[m]*v[m] - (v[-x]+v[+x]+v[-y]+v[+y]+v[-z]+v[+z])   (free m)  */
/* which is row m of the reference matrix (F90:126-169, dx = 1) applied to v.     */
/* ------------------------------------------------------------------------- */
void oi_oracle_codes(const uint8_t* mask, int nx, int ny, int nz, int dir, uint8_t* code) {
  const int nd[3] = {nx, ny, nz};
#pragma omp parallel for collapse(2) schedule(static)
  for (int k = 0; k < nz; ++k)
    for (int j = 0; j < ny; ++j)
      for (int i = 0; i < nx; ++i) {
        const i64 m = IDX(i, j, k);
        const int c[3] = {i, j, k};
        int cnt = 0;
        if (mask[m] && c[dir] != 0 && c[dir] != nd[dir] - 1) {
          cnt += (i > 0 && mask[m - 1]) + (i < nx - 1 && mask[m + 1]);
          cnt += (j > 0 && mask[m - nx]) + (j < ny - 1 && mask[m + nx]);
          cnt += (k > 0 && mask[m - (i64)nx * ny]) + (k < nz - 1 && mask[m + (i64)nx * ny]);
        }
        code[m] = (uint8_t)cnt;
      }
}

/* q = A v for the matrix-free operator; the summation order is the contract the
 * HIP kernels follow: ((((xm + xp) + ym) + yp) + zm) + zp, then c*v - sum. */
static inline double apply_at(const uint8_t* code, const double* v, int nx, int ny, int nz, int i,
                              int j, int k) {
  const i64 m = IDX(i, j, k);
  const double xm = (i > 0) ? v[m - 1] : 0.0, xp = (i < nx - 1) ? v[m + 1] : 0.0;
  const double ym = (j > 0) ? v[m - nx] : 0.0, yp = (j < ny - 1) ? v[m + nx] : 0.0;
  const double zm = (k > 0) ? v[m - (i64)nx * ny] : 0.0;
  const double zp = (k < nz - 1) ? v[m + (i64)nx * ny] : 0.0;
  const double s = ((((xm + xp) + ym) + yp) + zm) + zp;
  return (double)code[m] * v[m] - s;
}

void oi_oracle_apply(const uint8_t* code, const double* v, int nx, int ny, int nz, double* q) {
#pragma omp parallel for collapse(2) schedule(static)
  for (int k = 0; k < nz; ++k)
    for (int j = 0; j < ny; ++j)
      for (int i = 0; i < nx; ++i) {
        const i64 m = IDX(i, j, k);
        q[m] = code[m] ? apply_at(code, v, nx, ny, nz, i, j, k) : 0.0;
      }
}

/* Initial field: the reference's ramp on active cells (F90:233-262), which is
 * exactly vlo / vhi on the Dirichlet faces up to one rounding; the faces are
 * then set to the exact Dirichlet values (the identity rows of F90:191-228). */
void oi_oracle_initial_field(const uint8_t* mask, int nx, int ny, int nz, int dir, double vlo,
                             double vhi, double* x) {
  const int nd[3] = {nx, ny, nz};
  const double extent = (double)(nd[dir] - 1);
  const double factor = (fabs(extent) < 1e-15) ? 0.0 : 1.0 / extent;
#pragma omp parallel for collapse(2) schedule(static)
  for (int k = 0; k < nz; ++k)
    for (int j = 0; j < ny; ++j)
      for (int i = 0; i < nx; ++i) {
        const i64 m = IDX(i, j, k);
        const int c[3] = {i, j, k};
        double v = 0.0;
        if (mask[m]) {
          v = vlo + (vhi - vlo) * (double)c[dir] * factor;
          if (c[dir] == 0) v = vlo;
          else if (c[dir] == nd[dir] - 1) v = vhi;
        }
        x[m] = v;
      }
}

/* ||b||_2 of the full reference system: b = vlo / vhi on active Dirichlet-face
 * cells, 0 elsewhere (F90:184,196,201). */
double oi_oracle_bnorm(const uint8_t* mask, int nx, int ny, int nz, int dir, double vlo,
                       double vhi) {
  const int nd[3] = {nx, ny, nz};
  i64 nlo = 0, nhi = 0;
  for (int k = 0; k < nz; ++k)
    for (int j = 0; j < ny; ++j)
      for (int i = 0; i < nx; ++i) {
        const int c[3] = {i, j, k};
        if (!mask[IDX(i, j, k)]) continue;
        if (c[dir] == 0) ++nlo;
        else if (c[dir] == nd[dir] - 1) ++nhi;
      }
  return sqrt((double)nlo * vlo * vlo + (double)nhi * vhi * vhi);
}

/* Matrix-free (Jacobi-)CG: the algorithm the HIP path runs.  x enters as the
 * initial field (Dirichlet values included) and leaves as the solution.
 * Returns 1 if ||r||/||b|| <= eps within maxiter. */
int oi_oracle_cg(const uint8_t* code, int nx, int ny, int nz, double bnorm, double eps,
                 int maxiter, int jacobi, double* x, int* iters_out, double* relres_out) {
  const i64 n = (i64)nx * ny * nz;
  double* r = (double*)calloc((size_t)n, sizeof(double));
  double* p = (double*)calloc((size_t)n, sizeof(double));
  double* q = (double*)calloc((size_t)n, sizeof(double));
  double rr = 0.0, rz = 0.0;
#pragma omp parallel for collapse(2) reduction(+ : rr, rz) schedule(static)
  for (int k = 0; k < nz; ++k)
    for (int j = 0; j < ny; ++j)
      for (int i = 0; i < nx; ++i) {
        const i64 m = IDX(i, j, k);
        if (!code[m]) continue;
        r[m] = -apply_at(code, x, nx, ny, nz, i, j, k);
        const double z = jacobi ? r[m] / (double)code[m] : r[m];
        p[m] = z;
        rr += r[m] * r[m];
        rz += r[m] * z;
      }
  int it = 0, conv = 0;
  double relres = (bnorm > 0.0) ? sqrt(rr) / bnorm : sqrt(rr);
  if (relres <= eps) conv = 1;
  while (!conv && it < maxiter) {
    double pq = 0.0;
#pragma omp parallel for collapse(2) reduction(+ : pq) schedule(static)
    for (int k = 0; k < nz; ++k)
      for (int j = 0; j < ny; ++j)
        for (int i = 0; i < nx; ++i) {
          const i64 m = IDX(i, j, k);
          if (!code[m]) { q[m] = 0.0; continue; }
          q[m] = apply_at(code, p, nx, ny, nz, i, j, k);
          pq += p[m] * q[m];
        }
    const double alpha = rz / pq;
    double rr_new = 0.0, rz_new = 0.0;
#pragma omp parallel for reduction(+ : rr_new, rz_new) schedule(static)
    for (i64 m = 0; m < n; ++m) {
      if (!code[m]) continue;
      x[m] += alpha * p[m];
      r[m] -= alpha * q[m];
      const double z = jacobi ? r[m] / (double)code[m] : r[m];
      rr_new += r[m] * r[m];
      rz_new += r[m] * z;
    }
    ++it;
    relres = (bnorm > 0.0) ? sqrt(rr_new) / bnorm : sqrt(rr_new);
    if (relres <= eps) { conv = 1; break; }
    const double beta = rz_new / rz;
#pragma omp parallel for schedule(static)
    for (i64 m = 0; m < n; ++m) {
      if (!code[m]) continue;
      const double z = jacobi ? r[m] / (double)code[m] : r[m];
      p[m] = z + beta * p[m];
    }
    rr = rr_new; rz = rz_new;
  }
  *iters_out = it;
  *relres_out = relres;
  free(r); free(p); free(q);
  return conv && isfinite(relres);
}

/* Fixed number of CG iterations, no convergence test: the cpu_baseline leg of
 * bench.py times this (same work per iteration as oi_oracle_cg). */
void oi_oracle_cg_fixed(const uint8_t* code, int nx, int ny, int nz, int niter, double* x) {
  int it; double rel;
  oi_oracle_cg(code, nx, ny, nz, 1.0, -1.0, niter, 0, x, &it, &rel);
}

/* ------------------------------------------------------------------------- */
/* global_fluxes   (src/props/TortuosityHypre.cpp:1000-1134), dx = 1.            */
/* ------------------------------------------------------------------------- */
void oi_oracle_fluxes(const double* x, const uint8_t* mask, int nx, int ny, int nz, int dir,
                      double* flux_in, double* flux_out, i64* n_in, i64* n_out) {
  const int nd[3] = {nx, ny, nz};
  const i64 stride[3] = {1, nx, (i64)nx * ny};
  const i64 sh = stride[dir];
  double fin = 0.0, fout = 0.0;
  i64 cin = 0, cout = 0;
  for (int k = 0; k < nz; ++k)
    for (int j = 0; j < ny; ++j)
      for (int i = 0; i < nx; ++i) {
        const int c[3] = {i, j, k};
        const i64 m = IDX(i, j, k);
        if (c[dir] == 0 && mask[m]) { /* :1066-1084 */
          ++cin;
          if (nd[dir] > 1 && mask[m + sh]) fin += -((x[m + sh] - x[m]) / 1.0);
        }
        if (c[dir] == nd[dir] - 1 && mask[m]) { /* :1085-1104 */
          ++cout;
          if (nd[dir] > 1 && mask[m - sh]) fout += -((x[m] - x[m - sh]) / 1.0);
        }
      }
  *flux_in = fin * 1.0; /* face_area_element = 1, :1123-1133 */
  *flux_out = fout * 1.0;
  *n_in = cin;
  *n_out = cout;
}

/* TortuosityHypre::value   (src/props/TortuosityHypre.cpp:761-891): the scalar
 * post-processing.  status: 0 ok, 2 no percolation, 3 not converged, 4 flux not
 * conserved (tau = NaN for 2..4), 0 with tau = +Inf for vanishing flux/gradient. */
double oi_oracle_tau(double active_vf, int converged, double flux_in, double flux_out, int nx,
                     int ny, int nz, int dir, double vlo, double vhi, double* deff_out,
                     int* status) {
  const double tiny = 1.e-15; /* :64 */
  const double nd[3] = {(double)nx, (double)ny, (double)nz};
  *deff_out = 0.0;
  if (active_vf <= DBL_EPSILON) { *status = 2; return NAN; } /* :764-777 */
  if (!converged) { *status = 3; return NAN; }               /* :782-787 */
  const double fin = fabs(flux_in), fout = fabs(flux_out);
  const double avg = 0.5 * (fin + fout);
  if (avg > tiny) { /* :794-808 */
    const double rel = fabs(fin - fout) / avg;
    if (rel > 1.0e-6) { *status = 4; return NAN; } /* :819-823 */
  }
  *status = 0;
  const double L = nd[dir];
  const double A = (dir == 0) ? nd[1] * nd[2] : (dir == 1) ? nd[0] * nd[2] : nd[0] * nd[1];
  const double grad = (vhi - vlo) / L; /* :841 */
  if (avg < tiny) return INFINITY;       /* :846-852 */
  if (fabs(grad) < tiny) return INFINITY; /* :860-865 */
  const double deff = (avg / A) / fabs(grad); /* :868 */
  *deff_out = deff;
  if (fabs(deff) < tiny) return INFINITY; /* :869-873 */
  return active_vf / deff;               /* :876 */
}

/* ========================================================================= */
/* Homogenization path (the reference's default calculation_method):          */
/* periodic cell problem chi_k, one solve per direction, then the D_eff tensor. */
/* "Next" row of SURVEY 8f-3; the reference pins no number for it, so this      */
/* restatement is checked only against its own invariants (parity unpinned).   */
/* ========================================================================= */

static inline i64 wrap_idx(int i, int j, int k, int nx, int ny, int nz) {
  i = (i + nx) % nx; j = (j + ny) % ny; k = (k + nz) % nz;
  return IDX(i, j, k);
}

/* effdiff_fillmtx   (src/props/EffDiffFillMtx.F90:42-264) on the whole periodic domain,
 * dx = 1.  Active = phase == phase_id (EffectiveDiffusivityHypre.cpp:213-323, no percolation
 * filter; ghost cells are periodic images).  Row of an active cell: off-diagonal -1 towards
 * every ACTIVE neighbour, diagonal = 6 whatever the neighbours are (F90: both branches add
 * inv_dx2), rhs = -(D(+)-D(-))/2 + [D(-) == 0] - [D(+) == 0] along dir_k, which simplifies to
 * (D(+) - D(-))/2.  Inactive cell: identity row, rhs 0.  xinit = 0. */
void oi_oracle_effdiff_fillmtx(const uint8_t* phase, int nx, int ny, int nz, int phase_id, int dir_k,
                               double* a, double* rhs) {
  for (int k = 0; k < nz; ++k)
    for (int j = 0; j < ny; ++j)
      for (int i = 0; i < nx; ++i) {
        const i64 m = IDX(i, j, k);
        double* row = a + 7 * m;
        for (int s = 0; s < 7; ++s) row[s] = 0.0;
        rhs[m] = 0.0;
        if (phase[m] != phase_id) { row[0] = 1.0; continue; }
        const i64 nb[6] = {wrap_idx(i - 1, j, k, nx, ny, nz), wrap_idx(i + 1, j, k, nx, ny, nz),
                           wrap_idx(i, j - 1, k, nx, ny, nz), wrap_idx(i, j + 1, k, nx, ny, nz),
                           wrap_idx(i, j, k - 1, nx, ny, nz), wrap_idx(i, j, k + 1, nx, ny, nz)};
        double diag = 0.0, flux_bc = 0.0;
        for (int s = 0; s < 6; ++s) {
          const int act = phase[nb[s]] == phase_id;
          diag += 1.0; /* F90:126-205: added in both branches */
          if (act) row[1 + s] = -1.0;
          else if (s / 2 == dir_k) flux_bc += (s % 2 == 0) ? 1.0 : -1.0; /* (ONE/dx) signs, F90:131-133,141-143 */
        }
        row[0] = diag;
        const double dm = phase[nb[2 * dir_k]] == phase_id ? 1.0 : 0.0, dp = phase[nb[2 * dir_k + 1]] == phase_id ? 1.0 : 0.0;
        rhs[m] = -(dp - dm) * 0.5 + flux_bc; /* F90:208-217 */
      }
}

/* Matrix-free CG on that system (SPD: symmetric, diagonal 6 >= number of -1 entries, strict
 * at every interface).  chi enters as 0.  Convergence ||b - A chi|| / ||b|| <= eps as the
 * reference reads it from HYPRE (EffectiveDiffusivityHypre.cpp:608-612).  Returns converged. */
int oi_oracle_effdiff_solve(const uint8_t* phase, int nx, int ny, int nz, int phase_id, int dir_k, double eps,
                            int maxiter, double* chi, int* iters_out, double* relres_out) {
  const i64 n = (i64)nx * ny * nz;
  double* r = (double*)calloc((size_t)n, sizeof(double));
  double* p = (double*)calloc((size_t)n, sizeof(double));
  double* q = (double*)calloc((size_t)n, sizeof(double));
  double rr = 0.0;
#pragma omp parallel for collapse(2) reduction(+ : rr) schedule(static)
  for (int k = 0; k < nz; ++k)
    for (int j = 0; j < ny; ++j)
      for (int i = 0; i < nx; ++i) {
        const i64 m = IDX(i, j, k);
        chi[m] = 0.0;
        if (phase[m] != phase_id) continue;
        const int c[3] = {i, j, k};
        int cm[3] = {i, j, k}, cp[3] = {i, j, k};
        cm[dir_k] = c[dir_k] - 1; cp[dir_k] = c[dir_k] + 1;
        const double dm = phase[wrap_idx(cm[0], cm[1], cm[2], nx, ny, nz)] == phase_id ? 1.0 : 0.0;
        const double dp = phase[wrap_idx(cp[0], cp[1], cp[2], nx, ny, nz)] == phase_id ? 1.0 : 0.0;
        r[m] = (dp - dm) * 0.5;
        p[m] = r[m];
        rr += r[m] * r[m];
      }
  const double bnorm = sqrt(rr);
  int it = 0, conv = 0;
  double relres = 0.0;
  if (bnorm == 0.0) { conv = 1; }
  while (!conv && it < maxiter) {
    double pq = 0.0;
#pragma omp parallel for collapse(2) reduction(+ : pq) schedule(static)
    for (int k = 0; k < nz; ++k)
      for (int j = 0; j < ny; ++j)
        for (int i = 0; i < nx; ++i) {
          const i64 m = IDX(i, j, k);
          if (phase[m] != phase_id) { q[m] = 0.0; continue; }
          /* p is 0 on inactive cells, so summing all six neighbours = summing the active ones;
           * same order as the HIP kernel: ((((xm + xp) + ym) + yp) + zm) + zp */
          const double s = ((((p[wrap_idx(i - 1, j, k, nx, ny, nz)] + p[wrap_idx(i + 1, j, k, nx, ny, nz)]) +
                              p[wrap_idx(i, j - 1, k, nx, ny, nz)]) + p[wrap_idx(i, j + 1, k, nx, ny, nz)]) +
                            p[wrap_idx(i, j, k - 1, nx, ny, nz)]) + p[wrap_idx(i, j, k + 1, nx, ny, nz)];
          q[m] = 6.0 * p[m] - s;
          pq += p[m] * q[m];
        }
    const double alpha = rr / pq;
    double rr_new = 0.0;
#pragma omp parallel for reduction(+ : rr_new) schedule(static)
    for (i64 m = 0; m < n; ++m) {
      chi[m] += alpha * p[m];
      r[m] -= alpha * q[m];
      rr_new += r[m] * r[m];
    }
    ++it;
    relres = sqrt(rr_new) / bnorm;
    if (relres <= eps) { conv = 1; break; }
    const double beta = rr_new / rr;
#pragma omp parallel for schedule(static)
    for (i64 m = 0; m < n; ++m) p[m] = r[m] + beta * p[m];
    rr = rr_new;
  }
  *iters_out = it;
  *relres_out = relres;
  free(r); free(p); free(q);
  return conv;
}

/* calculate_Deff_tensor_homogenization   (src/props/Diffusion.cpp:60-167), dx = 1:
 * D[i][j] = (1/N_total) sum over ACTIVE cells of (delta_ij - d chi_j / d x_i), central
 * differences over periodic neighbours (chi is 0 on inactive cells). */
void oi_oracle_effdiff_tensor(const uint8_t* phase, int nx, int ny, int nz, int phase_id, const double* chi_x,
                              const double* chi_y, const double* chi_z, double* D /* [9] row-major */) {
  const double* chi[3] = {chi_x, chi_y, chi_z};
  double sum[9] = {0};
  for (int k = 0; k < nz; ++k)
    for (int j = 0; j < ny; ++j)
      for (int i = 0; i < nx; ++i) {
        if (phase[IDX(i, j, k)] != phase_id) continue;
        const i64 nb[6] = {wrap_idx(i - 1, j, k, nx, ny, nz), wrap_idx(i + 1, j, k, nx, ny, nz),
                           wrap_idx(i, j - 1, k, nx, ny, nz), wrap_idx(i, j + 1, k, nx, ny, nz),
                           wrap_idx(i, j, k - 1, nx, ny, nz), wrap_idx(i, j, k + 1, nx, ny, nz)};
        for (int row = 0; row < 3; ++row)     /* derivative axis */
          for (int col = 0; col < 3; ++col) { /* which chi */
            const double grad = (chi[col][nb[2 * row + 1]] - chi[col][nb[2 * row]]) * 0.5;
            sum[3 * row + col] += (row == col ? 1.0 : 0.0) - grad;
          }
      }
  const double ntot = (double)nx * ny * nz;
  for (int t = 0; t < 9; ++t) D[t] = sum[t] / ntot;
}
