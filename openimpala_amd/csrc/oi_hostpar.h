// oi_hostpar.h -- host-side helpers shared by the file readers (oi_io.cpp, oi_hdf5.cpp): a thread pool for the independent
// planes / pages / chunks of a volume and the typed "sample > threshold" pass.  No device API (the readers are also built
// alone, by g++, for the sanitizer drivers).
//
// What the reference does here: every reader hands the raw samples to one sequential loop `value > threshold ? 1 : 0` in
// double precision (TiffReader.cpp:385-434, RawReader.H:214, HDF5Reader.cpp:333-402).  A 1024^3 16-bit volume is 2 GB of
// samples; at the 0.4 Gvoxel/s of a per-sample switch that pass alone outlasts the three solves that follow it.
#pragma once
#include <sched.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <exception>
#include <limits>
#include <mutex>
#include <thread>
#include <type_traits>
#include <vector>

namespace oi {

// host cores this process may use (affinity / cgroup aware through sched_getaffinity)
inline unsigned host_threads() {
  cpu_set_t set;
  CPU_ZERO(&set);
  if (sched_getaffinity(0, sizeof(set), &set) == 0) {
    const int n = CPU_COUNT(&set);
    if (n > 0) return (unsigned)n;
  }
  const unsigned h = std::thread::hardware_concurrency();
  return h ? h : 1;
}

// threads parallel_for uses for n items: at most 16, at most n, at most the host cores this process may use
inline unsigned parallel_workers(size_t n) {
  return (unsigned)std::max<size_t>(1, std::min<size_t>({(size_t)host_threads(), (size_t)16, n}));
}

// f(i, w) for every i in [0, n), items handed out one at a time (pages and chunks differ in cost); w < parallel_workers(n) names
// the thread that runs the item (scratch buffers per worker, owned by the caller); f must be safe to run concurrently for
// different i.  An exception thrown by f (bad_alloc of a scratch buffer ...) stops the hand-out and is rethrown here, on the
// calling thread, after every worker has been joined -- as the sequential loops this replaces let the caller's handler see it.
template <typename F>
void parallel_for_w(size_t n, F&& f) {
  if (n == 0) return;
  const unsigned nthreads = parallel_workers(n);
  if (nthreads == 1) {
    for (size_t i = 0; i < n; ++i) f(i, 0u);
    return;
  }
  std::atomic<size_t> next{0};
  std::exception_ptr first;
  std::mutex first_mutex;
  auto worker = [&](unsigned w) {
    for (;;) {
      const size_t i = next.fetch_add(1);
      if (i >= n) return;
      try {
        f(i, w);
      } catch (...) {
        std::lock_guard<std::mutex> lk(first_mutex);
        if (!first) first = std::current_exception();
        next.store(n);  // nothing more is handed out
        return;
      }
    }
  };
  std::vector<std::thread> pool;
  pool.reserve(nthreads - 1);
  try {
    for (unsigned t = 1; t < nthreads; ++t) pool.emplace_back(worker, t);
  } catch (...) {  // (a thread could not be started: the ones that run finish the work)
  }
  worker(0);
  for (auto& t : pool) t.join();
  if (first) std::rethrow_exception(first);
}
template <typename F>
void parallel_for(size_t n, F&& f) {
  parallel_for_w(n, [&](size_t i, unsigned) { f(i); });
}

// ---- dst[i] = (double)sample_i > threshold ? 1 : 0 for n samples of one type, bit for bit what the per-sample double
// compare gives.  Integers of up to 32 bits: (double)v > t  <=>  v > floor(t) (v is an integer), so the compare runs in the
// sample's own type and the compiler vectorises it; 64-bit integers and floats keep the double compare.
enum SampleClass { kSampleUnsigned = 0, kSampleSigned = 1, kSampleFloat = 2 };

template <typename T>
inline T load_sample(const uint8_t* p, bool swap) {
  T v;
  memcpy(&v, p, sizeof(T));
  if (swap) {
    if (sizeof(T) == 2) { uint16_t u; memcpy(&u, &v, 2); u = __builtin_bswap16(u); memcpy(&v, &u, 2); }
    else if (sizeof(T) == 4) { uint32_t u; memcpy(&u, &v, 4); u = __builtin_bswap32(u); memcpy(&v, &u, 4); }
    else if (sizeof(T) == 8) { uint64_t u; memcpy(&u, &v, 8); u = __builtin_bswap64(u); memcpy(&v, &u, 8); }
  }
  return v;
}

template <typename T>
inline void threshold_int(const uint8_t* src, size_t n, bool swap, double thr, uint8_t* dst) {
  static_assert(std::is_integral<T>::value && sizeof(T) <= 4, "exact in double");
  if (std::isnan(thr)) { memset(dst, 0, n); return; }                                                  // v > NaN is false
  const double fl = std::floor(thr);
  if (fl < (double)std::numeric_limits<T>::min()) { memset(dst, 1, n); return; }                       // every sample is greater
  if (fl >= (double)std::numeric_limits<T>::max()) { memset(dst, 0, n); return; }                      // none is
  const T t = (T)fl;
  if (sizeof(T) == 1 || !swap) {
    for (size_t i = 0; i < n; ++i) { T v; memcpy(&v, src + i * sizeof(T), sizeof(T)); dst[i] = v > t ? 1 : 0; }
  } else {
    for (size_t i = 0; i < n; ++i) dst[i] = load_sample<T>(src + i * sizeof(T), true) > t ? 1 : 0;
  }
}
template <typename T>
inline void threshold_dbl(const uint8_t* src, size_t n, bool swap, double thr, uint8_t* dst) {
  for (size_t i = 0; i < n; ++i) dst[i] = (double)load_sample<T>(src + i * sizeof(T), swap) > thr ? 1 : 0;
}

// bytes: 1, 2, 4 or 8; cls: SampleClass; swap: the file's byte order differs from the host's.  false for an unknown type.
inline bool threshold_run(const uint8_t* src, size_t n, int bytes, int cls, bool swap, double thr, uint8_t* dst) {
  if (cls == kSampleFloat) {
    if (bytes == 4) threshold_dbl<float>(src, n, swap, thr, dst);
    else if (bytes == 8) threshold_dbl<double>(src, n, swap, thr, dst);
    else return false;
    return true;
  }
  const bool sg = cls == kSampleSigned;
  switch (bytes) {
    case 1: if (sg) threshold_int<int8_t>(src, n, false, thr, dst); else threshold_int<uint8_t>(src, n, false, thr, dst); return true;
    case 2: if (sg) threshold_int<int16_t>(src, n, swap, thr, dst); else threshold_int<uint16_t>(src, n, swap, thr, dst); return true;
    case 4: if (sg) threshold_int<int32_t>(src, n, swap, thr, dst); else threshold_int<uint32_t>(src, n, swap, thr, dst); return true;
    case 8: if (sg) threshold_dbl<int64_t>(src, n, swap, thr, dst); else threshold_dbl<uint64_t>(src, n, swap, thr, dst); return true;
    default: return false;
  }
}

}  // namespace oi
