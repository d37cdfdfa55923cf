// oi_pool.cpp -- see oi_pool.h
#include "oi_pool.h"

#include <map>
#include <mutex>
#include <unordered_map>
#include <utility>

namespace oi {
namespace {
struct Pool {
  std::mutex m;
  bool on = false;
  size_t cap = 0, held = 0, hits = 0, misses = 0;
  std::multimap<std::pair<int, size_t>, void*> free_;        // (device, block bytes) -> block
  std::unordered_map<void*, std::pair<int, size_t>> live;    // blocks handed out by dev_malloc while the pool was on
};
Pool& pool() {
  static Pool* p = new Pool();  // (never destroyed: the HIP runtime may be gone before static destructors run)
  return *p;
}
size_t trim_locked(Pool& P) {
  size_t n = 0;
  for (auto& kv : P.free_) { (void)hipFree(kv.second); n += kv.first.second; }
  P.free_.clear();
  P.held = 0;
  return n;
}
}  // namespace

hipError_t dev_malloc(void** p, size_t bytes) {
  Pool& P = pool();
  {
    std::lock_guard<std::mutex> lk(P.m);
    if (!P.on) return hipMalloc(p, bytes);
  }
  int dev = 0;
  (void)hipGetDevice(&dev);
  void* got = nullptr;
  size_t block = 0;
  {
    std::lock_guard<std::mutex> lk(P.m);
    // the smallest pooled block of this device that holds the request and wastes at most an eighth of it
    auto it = P.free_.lower_bound({dev, bytes});
    if (it != P.free_.end() && it->first.first == dev && it->first.second <= bytes + bytes / 8 + 4096) {
      got = it->second; block = it->first.second;
      P.free_.erase(it);
      P.held -= block;
      P.live[got] = {dev, block};
      ++P.hits;
    }
  }
  if (got) {
    // zeroed, like the fresh memory it stands in for; on the legacy stream and waited for: the caller's own streams do not order with it
    hipError_t e = hipMemsetAsync(got, 0, bytes, nullptr);
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    if (e != hipSuccess) return e;
    *p = got;
    return hipSuccess;
  }
  hipError_t e = hipMalloc(p, bytes);
  if (e != hipSuccess) {  // give the pool back and try once more
    (void)hipGetLastError();
    { std::lock_guard<std::mutex> lk(P.m); (void)trim_locked(P); }
    e = hipMalloc(p, bytes);
  }
  if (e == hipSuccess) {
    std::lock_guard<std::mutex> lk(P.m);
    P.live[*p] = {dev, bytes};
    ++P.misses;
  }
  return e;
}

hipError_t dev_free(void* p) {
  if (!p) return hipSuccess;
  Pool& P = pool();
  std::pair<int, size_t> what{0, 0};
  bool keep = false;
  {
    std::lock_guard<std::mutex> lk(P.m);
    auto it = P.live.find(p);
    if (it != P.live.end()) {
      what = it->second;
      P.live.erase(it);
      keep = P.on && (P.cap == 0 || P.held + what.second <= P.cap);
    }
  }
  if (!keep) return hipFree(p);
  // hipFree waits for the device before it lets go of memory; a block that goes into the pool must not be in use either
  const hipError_t e = hipDeviceSynchronize();
  if (e != hipSuccess) { (void)hipFree(p); return e; }
  std::lock_guard<std::mutex> lk(P.m);
  P.free_.insert({what, p});
  P.held += what.second;
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

DevPoolStats dev_pool_stats() {
  Pool& P = pool();
  std::lock_guard<std::mutex> lk(P.m);
  return DevPoolStats{P.held, P.free_.size(), P.hits, P.misses, P.on ? 1 : 0};
}
}  // namespace oi
