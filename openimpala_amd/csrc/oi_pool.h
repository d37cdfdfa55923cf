// oi_pool.h -- EVERY device allocation of the library goes through this pair of calls: one ledger of what is live (so that the
// leak tests compare exact byte counts instead of reading hipMemGetInfo), and, for the arrays of the handles (tortuosity,
// homogenization, multigrid hierarchy), an optional per-process pool, so that a process which builds handle after handle of the same
// shape -- `direction = All`, a REV study -- can keep the arrays of a destroyed handle for the next one instead of returning them to
// the driver.
//
// Why a pool (round 4, profiles/r04/app_diffusion_end_to_end_1024_raw_u16.txt; round 3, alloc_times_clean_dirty_clean.txt): on MI355X
// the driver clears memory before it hands it out again, in one lump when an allocation runs past the memory that is still clean --
// one hipMalloc of 4 GB returns after 0.2 ms or after 5.8 s.  The second of three otherwise identical solver objects of one
// `Diffusion` run took 1.6 s to build instead of 0.07 s.  With the pool ON a released block goes into a per-process list (after a
// synchronisation of ITS device: what hipFree does implicitly) and the next request of that size on that device takes it back, zeroed
// (fresh memory from this driver is zero; nothing here may start to depend on what a previous handle left).  OFF (the default)
// dev_malloc / dev_free are hipMalloc / hipFree plus the ledger entry.
//
//   dev_malloc  -- poolable block (handle arrays)
//   raw_malloc  -- never from the pool: the vectors of the placement search (oi_tort.cpp: vecset_alloc -- the search rests on fresh,
//                  consecutive allocations), scratch of the debug / bench entry points, the transports' staging buffers, oi_dev_malloc
//   dev_free    -- either kind (the ledger knows which); a pointer the ledger has never seen goes to hipFree as it is
// Both allocators give the pool back to the driver and try once more when the driver refuses a request.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>

namespace oi {
hipError_t dev_malloc(void** p, size_t bytes);
hipError_t raw_malloc(void** p, size_t bytes);
hipError_t dev_free(void* p);
template <class T>
inline hipError_t dev_malloc(T** p, size_t bytes) { return dev_malloc(reinterpret_cast<void**>(p), bytes); }
template <class T>
inline hipError_t raw_malloc(T** p, size_t bytes) { return raw_malloc(reinterpret_cast<void**>(p), bytes); }
void dev_pool_set(bool on, size_t cap_bytes /* 0 = no cap: the pool gives everything back when an allocation fails */);
size_t dev_pool_trim();  // every pooled block back to the driver; returns the bytes released
size_t dev_pool_held(int device);  // bytes the pool holds on `device` (hipMemGetInfo counts them as used)
struct DevPoolStats { size_t held_bytes, held_blocks, hits, misses; int on; };
DevPoolStats dev_pool_stats();
// the ledger: requested bytes / number of the blocks handed out and not yet freed (pooled blocks are NOT live), and the high-water mark
struct DevLedger { size_t live_bytes, live_blocks, peak_bytes; };
DevLedger dev_ledger();
void dev_ledger_reset_peak();
// test hook: the driver refuses the n-th allocation made of it from now (1 = the next one; requests served by the pool do not count), once;
// 0 disarms.  Like a real refusal it makes the allocator give the pool back and ask again.
void dev_fail_after(long long n);
}  // namespace oi
