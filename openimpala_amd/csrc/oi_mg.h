// oi_mg.h -- aggregation-multigrid V-cycle preconditioner (see oi_mg.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "oi_comm.h"
#include "oi_device.h"

namespace oi {
struct MgHierarchy;
// builds every coarse operator from the fine code bytes of this slab (ghost code planes valid);
// collective over `comm` when it has more than one rank
// periodic != 0: the homogenization cell problem (code = 6 on active cells, x / y / z wrap; the ghost planes of `code`
// and of every fine vector handed to the cycle hold the periodic images / ring neighbours' planes)
MgHierarchy* mg_setup(const Geom& g, const uint8_t* code, oi_comm* comm, int periodic, hipStream_t s, size_t* bytes_out);
void mg_destroy(MgHierarchy* mg);
// z = M^-1 r; r, scratch, z are fine vectors (pitched, ghost planes); scratch is overwritten and
// r's ghost planes must hold the neighbours' values (z's are NOT filled).
// Also leaves the per-block partial sums of z.r in `partials`; returns their number (-1: transport failure).
// tiled = 0 selects the plain (cache-served) fine-level kernels kept as a cross-check.
// lineoff != null: r and z are in the compacted line layout (launch_build_lines); needs tiled != 0 and an even g.z0
int mg_vcycle(MgHierarchy* mg, const Geom& g, const uint8_t* code, const double* r, double* scratch, double* z,
              double* partials, const Ctrl* ctrl, int tiled, const int* lineoff, hipStream_t s,
              const long long* plane_first = nullptr, const void* seg = nullptr /* with both: the post-smoother as k_cg1b FORM 3 */);
// The cycle in two halves, for a caller that forms the first coarse right-hand side itself (launch_mg_pre_update: residual
// update + pre-smoother + restriction in one kernel): where the block sums go (`prepare`s the target), then the rest of
// the cycle -- coarse levels and the fine post-smoother, partial sums of z.r as mg_vcycle leaves them
bool mg_coarse_target(MgHierarchy* mg, MgCoarse* c, float** bcoarse, double* omega, hipStream_t s);
// xu != null (and mg_post_descr(...) true): the post-smoother also applies x += alpha_{k-1} p_{k-1}; the caller then launches the
// first kernel of the iteration with x_elsewhere = 1
struct MgXUpdate { double* x; const void* p_prev; int p_f32; const double* hist; int k; const void* p_prev2; /* != null: pairwise (launch_mg_post) */ };
bool mg_post_descr(const Geom& g, const int* lineoff, const long long* plane_first, const void* seg);  // does the post-smoother run as k_cg1b FORM 3?
int mg_vcycle_after_restriction(MgHierarchy* mg, const Geom& g, const uint8_t* code, const double* r, double* z, double* partials,
                                const Ctrl* ctrl, const int* lineoff, const long long* plane_first, const void* seg, hipStream_t s,
                                const MgXUpdate* xu = nullptr, hipEvent_t before_post = nullptr /* recorded between the coarse levels and the post-smoother */);
int mg_levels(const MgHierarchy* mg);
int mg_replicated_from(const MgHierarchy* mg);  // first level (1 = first coarse) solved redundantly on every rank, -1 = none
// partial sums of a.b over the owned planes; returns the number of partials
int launch_dot(const Geom& g, const double* a, const double* b, double* partials, hipStream_t s);
}  // namespace oi
