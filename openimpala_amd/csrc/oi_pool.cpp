// oi_pool.cpp -- see oi_pool.h
#include "oi_pool.h"

#include <atomic>
#include <map>
#include <mutex>
#include <unordered_map>
#include <utility>

namespace oi {
namespace {
struct Block {
  int dev;
  size_t block;      // bytes of the driver allocation
  size_t requested;  // bytes the caller asked for (what the ledger sums)
  bool poolable;
};
struct Pool {
  std::mutex m;
  bool on = false;
  size_t cap = 0, held = 0, hits = 0, misses = 0;
  std::multimap<std::pair<int, size_t>, void*> free_;  // (device, block bytes) -> block
  std::unordered_map<void*, Block> live;               // every block handed out and not yet freed
  size_t live_bytes = 0, peak_bytes = 0;
};
Pool& pool() {
  static Pool* p = new Pool();  // (never destroyed: the HIP runtime may be gone before static destructors run)
  return *p;
}
std::atomic<long long> g_fail_after{0};

size_t trim_locked(Pool& P) {
  size_t n = 0;
  for (auto& kv : P.free_) { (void)hipFree(kv.second); n += kv.first.second; }
  P.free_.clear();
  P.held = 0;
  return n;
}
void record_locked(Pool& P, void* p, const Block& b) {
  P.live[p] = b;
  P.live_bytes += b.requested;
  if (P.live_bytes > P.peak_bytes) P.peak_bytes = P.live_bytes;
}
bool injected_failure() {
  long long v = g_fail_after.load(std::memory_order_relaxed);
  while (v > 0) {
    if (g_fail_after.compare_exchange_weak(v, v - 1, std::memory_order_relaxed)) return v == 1;
  }
  return false;
}
// hipMalloc; when the driver refuses, the pool goes back to it and the request is made once more
// (an injected failure stands for the driver's refusal: the pool is given back and the request repeated all the same)
hipError_t driver_malloc(Pool& P, void** p, size_t bytes) {
  hipError_t e = injected_failure() ? hipErrorOutOfMemory : hipMalloc(p, bytes);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    size_t gave = 0;
    { std::lock_guard<std::mutex> lk(P.m); gave = trim_locked(P); }
    if (gave) e = hipMalloc(p, bytes);
  }
  return e;
}
}  // namespace

hipError_t raw_malloc(void** p, size_t bytes) {
  Pool& P = pool();
  int dev = 0;
  (void)hipGetDevice(&dev);
  const hipError_t e = driver_malloc(P, p, bytes);
  if (e == hipSuccess) {
    std::lock_guard<std::mutex> lk(P.m);
    record_locked(P, *p, Block{dev, bytes, bytes, false});
  }
  return e;
}

hipError_t dev_malloc(void** p, size_t bytes) {
  Pool& P = pool();
  int dev = 0;
  (void)hipGetDevice(&dev);
  void* got = nullptr;
  size_t block = 0;
  bool on;
  {
    std::lock_guard<std::mutex> lk(P.m);
    on = P.on;
    if (on) {
      // the smallest pooled block of this device that holds the request and wastes at most an eighth of it
      auto it = P.free_.lower_bound({dev, bytes});
      if (it != P.free_.end() && it->first.first == dev && it->first.second <= bytes + bytes / 8 + 4096) {
        got = it->second; block = it->first.second;
        P.free_.erase(it);
        P.held -= block;
        ++P.hits;
      }
    }
  }
  if (got) {
    // zeroed, like the fresh memory it stands in for; on the legacy stream and waited for: the caller's own streams do not order with it
    hipError_t e = hipMemsetAsync(got, 0, bytes, nullptr);
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    if (e != hipSuccess) { (void)hipFree(got); return e; }
    std::lock_guard<std::mutex> lk(P.m);
    record_locked(P, got, Block{dev, block, bytes, true});
    *p = got;
    return hipSuccess;
  }
  const hipError_t e = driver_malloc(P, p, bytes);
  if (e == hipSuccess) {
    std::lock_guard<std::mutex> lk(P.m);
    record_locked(P, *p, Block{dev, bytes, bytes, true});
    if (on) ++P.misses;
  }
  return e;
}

hipError_t dev_free(void* p) {
  if (!p) return hipSuccess;
  Pool& P = pool();
  Block what{0, 0, 0, false};
  bool keep = false;
  {
    std::lock_guard<std::mutex> lk(P.m);
    auto it = P.live.find(p);
    if (it != P.live.end()) {
      what = it->second;
      P.live.erase(it);
      P.live_bytes -= what.requested;
      keep = what.poolable && P.on && (P.cap == 0 || P.held + what.block <= P.cap);
    }
  }
  if (!keep) return hipFree(p);
  // hipFree waits for the device before it lets go of memory; a block that goes into the pool must not be in use either --
  // on the device the BLOCK lives on, which need not be the caller's current one
  int cur = 0;
  (void)hipGetDevice(&cur);
  if (cur != what.dev) (void)hipSetDevice(what.dev);
  const hipError_t e = hipDeviceSynchronize();
  if (cur != what.dev) (void)hipSetDevice(cur);
  if (e != hipSuccess) { (void)hipFree(p); return e; }
  std::lock_guard<std::mutex> lk(P.m);
  P.free_.insert({{what.dev, what.block}, p});
  P.held += what.block;
  return hipSuccess;
}

void dev_pool_set(bool on, size_t cap_bytes) {
  Pool& P = pool();
  std::lock_guard<std::mutex> lk(P.m);
  P.on = on;
  P.cap = cap_bytes;
  if (!on) (void)trim_locked(P);
}

size_t dev_pool_trim() {
  Pool& P = pool();
  std::lock_guard<std::mutex> lk(P.m);
  return trim_locked(P);
}

size_t dev_pool_held(int device) {
  Pool& P = pool();
  std::lock_guard<std::mutex> lk(P.m);
  size_t n = 0;
  for (auto it = P.free_.lower_bound({device, 0}); it != P.free_.end() && it->first.first == device; ++it) n += it->first.second;
  return n;
}

DevPoolStats dev_pool_stats() {
  Pool& P = pool();
  std::lock_guard<std::mutex> lk(P.m);
  return DevPoolStats{P.held, P.free_.size(), P.hits, P.misses, P.on ? 1 : 0};
}

DevLedger dev_ledger() {
  Pool& P = pool();
  std::lock_guard<std::mutex> lk(P.m);
  return DevLedger{P.live_bytes, P.live.size(), P.peak_bytes};
}

void dev_ledger_reset_peak() {
  Pool& P = pool();
  std::lock_guard<std::mutex> lk(P.m);
  P.peak_bytes = P.live_bytes;
}

void dev_fail_after(long long n) { g_fail_after.store(n > 0 ? n : 0, std::memory_order_relaxed); }
}  // namespace oi
