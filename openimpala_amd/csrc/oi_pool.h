// oi_pool.h -- device allocations of the handles (tortuosity, homogenization, multigrid hierarchy) behind one pair of calls, so that a
// process which builds handle after handle of the same shape -- `direction = All`, a REV study -- can keep the arrays of a destroyed
// handle for the next one instead of returning them to the driver.
//
// Why (round 4, profiles/r04/app_diffusion_end_to_end_1024_raw_u16.txt; round 3, alloc_times_clean_dirty_clean.txt): on MI355X the
// driver clears memory before it hands it out again, in one lump when an allocation runs past the memory that is still clean -- one
// hipMalloc of 4 GB returns after 0.2 ms or after 5.8 s.  The second of three otherwise identical solver objects of one `Diffusion`
// run took 1.6 s to build instead of 0.07 s.  With the pool ON a released block goes into a per-process list (after a device
// synchronisation: what hipFree does implicitly) and the next request of that size on that device takes it back, zeroed (fresh memory
// from this driver is zero; nothing here may start to depend on what a previous handle left).  OFF (the default) both calls are
// hipMalloc / hipFree and a destroyed handle's memory is the driver's again at once (tests/test_zz_robustness_gpu.py).
// The vectors of the placement search (oi_tort.cpp: vecset_alloc) never come from the pool: the search rests on fresh, consecutive
// allocations, and the slots it does not keep go back to the driver.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>

namespace oi {
hipError_t dev_malloc(void** p, size_t bytes);
hipError_t dev_free(void* p);
template <class T>
inline hipError_t dev_malloc(T** p, size_t bytes) { return dev_malloc(reinterpret_cast<void**>(p), bytes); }
void dev_pool_set(bool on, size_t cap_bytes /* 0 = no cap: the pool gives everything back when an allocation fails */);
size_t dev_pool_trim();  // every pooled block back to the driver; returns the bytes released
struct DevPoolStats { size_t held_bytes, held_blocks, hits, misses; int on; };
DevPoolStats dev_pool_stats();
}  // namespace oi
