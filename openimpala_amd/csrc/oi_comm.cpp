// oi_comm.cpp -- RCCL (one process per GPU) and in-process (host threads on one GPU)
// transports behind oi_comm.  See oi_comm.h.
#include "oi_comm.h"

#include <rccl/rccl.h>

#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
#include <vector>

#include "oi_device.h"
#include "oi_pool.h"

namespace oi {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
const char* last_error() { return g_err; }
}  // namespace oi

#define HIPC(call)                                                                       \
  do {                                                                                   \
    hipError_t e_ = (call);                                                              \
    if (e_ != hipSuccess) {                                                              \
      oi::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
      (void)hipGetLastError(); /* reported: do not leave it for the next launch check to find */ \
      return OI_HIP_ERROR;                                                               \
    }                                                                                    \
  } while (0)
#define NCCLC(call)                                                                      \
  do {                                                                                   \
    ncclResult_t r_ = (call);                                                            \
    if (r_ != ncclSuccess) {                                                             \
      oi::set_error("%s failed: %s (%s:%d)", #call, ncclGetErrorString(r_), __FILE__, __LINE__); \
      return OI_RCCL_ERROR;                                                              \
    }                                                                                    \
  } while (0)

// --------------------------------------------------------------------------------------
// RCCL transport: one rank = one process = one GPU.  Both neighbour messages of an
// exchange go into ONE group so the two xGMI links are driven concurrently.
// --------------------------------------------------------------------------------------
struct RcclComm : oi_comm {
  ncclComm_t nccl = nullptr;      // collectives (all-reduce), on the solver's stream
  ncclComm_t nccl_p2p = nullptr;  // ghost-plane send/recv: a communicator of its own (ncclCommSplit), because the
                                  // single-kernel iteration runs the exchange on a second stream while the scalar
                                  // all-reduce is in flight, and operations of ONE communicator must not overlap
  void* scratch = nullptr;  // 64 bytes of device memory for the host-value collectives
  ~RcclComm() override {
    if (scratch) (void)oi::dev_free(scratch);
    if (nccl_p2p && nccl_p2p != nccl) ncclCommDestroy(nccl_p2p);
    if (nccl) ncclCommDestroy(nccl);
  }
  oi_status exchange(const void* send_lo, void* recv_lo, const void* send_hi, void* recv_hi, size_t bytes,
                     hipStream_t s) override {
    if (size == 1) return OI_OK;  // no neighbours
    NCCLC(ncclGroupStart());
    if (rank > 0) {
      NCCLC(ncclSend(send_lo, bytes, ncclInt8, rank - 1, nccl_p2p, s));
      NCCLC(ncclRecv(recv_lo, bytes, ncclInt8, rank - 1, nccl_p2p, s));
    }
    if (rank < size - 1) {
      NCCLC(ncclSend(send_hi, bytes, ncclInt8, rank + 1, nccl_p2p, s));
      NCCLC(ncclRecv(recv_hi, bytes, ncclInt8, rank + 1, nccl_p2p, s));
    }
    NCCLC(ncclGroupEnd());
    return OI_OK;
  }
  oi_status exchange_v(const void* send_lo, size_t n_send_lo, void* recv_lo, size_t n_recv_lo, const void* send_hi,
                       size_t n_send_hi, void* recv_hi, size_t n_recv_hi, hipStream_t s) override {
    if (size == 1) return OI_OK;
    NCCLC(ncclGroupStart());
    if (rank > 0) {
      if (n_send_lo) NCCLC(ncclSend(send_lo, n_send_lo, ncclInt8, rank - 1, nccl_p2p, s));
      if (n_recv_lo) NCCLC(ncclRecv(recv_lo, n_recv_lo, ncclInt8, rank - 1, nccl_p2p, s));
    }
    if (rank < size - 1) {
      if (n_send_hi) NCCLC(ncclSend(send_hi, n_send_hi, ncclInt8, rank + 1, nccl_p2p, s));
      if (n_recv_hi) NCCLC(ncclRecv(recv_hi, n_recv_hi, ncclInt8, rank + 1, nccl_p2p, s));
    }
    NCCLC(ncclGroupEnd());
    return OI_OK;
  }
  oi_status exchange_ring(const void* send_lo, void* recv_lo, const void* send_hi, void* recv_hi, size_t bytes,
                          hipStream_t s) override {
    if (size == 1) {  // my own planes wrap around
      HIPC(hipMemcpyAsync(recv_lo, send_hi, bytes, hipMemcpyDeviceToDevice, s));
      HIPC(hipMemcpyAsync(recv_hi, send_lo, bytes, hipMemcpyDeviceToDevice, s));
      return OI_OK;
    }
    const int prev = (rank + size - 1) % size, next = (rank + 1) % size;
    // with two ranks both neighbours are the same peer: messages between a pair match in posting
    // order, so sends go (lo, hi) and receives (hi, lo) -- the peer's first plane is my upper ghost
    NCCLC(ncclGroupStart());
    NCCLC(ncclSend(send_lo, bytes, ncclInt8, prev, nccl_p2p, s));
    NCCLC(ncclSend(send_hi, bytes, ncclInt8, next, nccl_p2p, s));
    NCCLC(ncclRecv(recv_hi, bytes, ncclInt8, next, nccl_p2p, s));
    NCCLC(ncclRecv(recv_lo, bytes, ncclInt8, prev, nccl_p2p, s));
    NCCLC(ncclGroupEnd());
    return OI_OK;
  }
  oi_status exchange_ring_v(const void* send_lo, size_t n_send_lo, void* recv_lo, size_t n_recv_lo, const void* send_hi,
                            size_t n_send_hi, void* recv_hi, size_t n_recv_hi, hipStream_t s) override {
    if (size == 1) {
      if (n_send_hi) HIPC(hipMemcpyAsync(recv_lo, send_hi, n_send_hi, hipMemcpyDeviceToDevice, s));
      if (n_send_lo) HIPC(hipMemcpyAsync(recv_hi, send_lo, n_send_lo, hipMemcpyDeviceToDevice, s));
      return OI_OK;
    }
    const int prev = (rank + size - 1) % size, next = (rank + 1) % size;
    // (two ranks: both neighbours are the same peer, messages between a pair match in posting order -- sends (lo, hi),
    // receives (hi, lo), as in exchange_ring; the lengths were agreed at set-up, so a skipped send meets a skipped receive)
    NCCLC(ncclGroupStart());
    if (n_send_lo) NCCLC(ncclSend(send_lo, n_send_lo, ncclInt8, prev, nccl_p2p, s));
    if (n_send_hi) NCCLC(ncclSend(send_hi, n_send_hi, ncclInt8, next, nccl_p2p, s));
    if (n_recv_hi) NCCLC(ncclRecv(recv_hi, n_recv_hi, ncclInt8, next, nccl_p2p, s));
    if (n_recv_lo) NCCLC(ncclRecv(recv_lo, n_recv_lo, ncclInt8, prev, nccl_p2p, s));
    NCCLC(ncclGroupEnd());
    return OI_OK;
  }
  // a one-rank communicator still goes through RCCL (identity): lets a single-GPU box
  // exercise the library's RCCL calls next to torch.distributed's own communicator
  oi_status allreduce_sum_f64(double* dev, int n, hipStream_t s) override {
    NCCLC(ncclAllReduce(dev, dev, n, ncclDouble, ncclSum, nccl, s));
    return OI_OK;
  }
  oi_status allreduce_sum_f32(float* dev, size_t n, hipStream_t s) override {
    NCCLC(ncclAllReduce(dev, dev, n, ncclFloat, ncclSum, nccl, s));
    return OI_OK;
  }
  oi_status allreduce_sum_u64_host(unsigned long long* vals, int n, hipStream_t s) override {
    if (n > 8) return OI_BAD_ARG;
    HIPC(hipMemcpyAsync(scratch, vals, sizeof(unsigned long long) * n, hipMemcpyHostToDevice, s));
    NCCLC(ncclAllReduce(scratch, scratch, n, ncclUint64, ncclSum, nccl, s));
    HIPC(hipMemcpyAsync(vals, scratch, sizeof(unsigned long long) * n, hipMemcpyDeviceToHost, s));
    HIPC(hipStreamSynchronize(s));
    return OI_OK;
  }
  oi_status allreduce_max_i32_host(int* vals, int n, hipStream_t s) override {
    if (n > 16) return OI_BAD_ARG;
    HIPC(hipMemcpyAsync(scratch, vals, sizeof(int) * n, hipMemcpyHostToDevice, s));
    NCCLC(ncclAllReduce(scratch, scratch, n, ncclInt32, ncclMax, nccl, s));
    HIPC(hipMemcpyAsync(vals, scratch, sizeof(int) * n, hipMemcpyDeviceToHost, s));
    HIPC(hipStreamSynchronize(s));
    return OI_OK;
  }
};

// --------------------------------------------------------------------------------------
// In-process transport: `size` host threads, each driving one slab on the SAME device.
// Messages are device-to-device copies ordered with HIP events; threads rendezvous on a
// reusable barrier.  It exists so every slab code path (ghost planes, split reductions,
// flood-fill rounds) runs on a single-GPU box; it is not a performance path.
// --------------------------------------------------------------------------------------
namespace {
struct LocalShared {
  int n = 0;
  std::mutex m;
  std::condition_variable cv;
  int waiting = 0;
  unsigned long long generation = 0;
  std::vector<const void*> send_lo, send_hi;
  std::vector<hipEvent_t> ev_ready, ev_done;
  std::vector<double*> stage;          // per-rank device staging for the f64 all-reduce
  const double** stage_ptrs_dev = nullptr;  // device array of the n staging pointers
  std::vector<unsigned long long> hu64;
  std::vector<int> hi32;
  std::vector<float*> big;             // per-rank scratch of the large f32 all-reduce (grown on demand)
  std::vector<size_t> big_cap;
  void barrier() {
    std::unique_lock<std::mutex> lk(m);
    const unsigned long long gen = generation;
    if (++waiting == n) {
      waiting = 0;
      ++generation;
      cv.notify_all();
    } else {
      cv.wait(lk, [&] { return generation != gen; });
    }
  }
  ~LocalShared() {
    for (auto e : ev_ready) (void)hipEventDestroy(e);
    for (auto e : ev_done) (void)hipEventDestroy(e);
    for (auto p : stage) (void)oi::dev_free(p);
    for (auto p : big)
      if (p) (void)oi::dev_free(p);
    if (stage_ptrs_dev) (void)oi::dev_free((void*)stage_ptrs_dev);
  }
};
constexpr int kStageDoubles = 16;
}  // namespace

struct LocalComm : oi_comm {
  std::shared_ptr<LocalShared> sh;
  oi_status exchange(const void* send_lo, void* recv_lo, const void* send_hi, void* recv_hi, size_t bytes,
                     hipStream_t s) override {
    if (size == 1) return OI_OK;
    sh->send_lo[rank] = send_lo;
    sh->send_hi[rank] = send_hi;
    HIPC(hipEventRecord(sh->ev_ready[rank], s));
    sh->barrier();
    if (rank > 0) {
      HIPC(hipStreamWaitEvent(s, sh->ev_ready[rank - 1], 0));
      HIPC(hipMemcpyAsync(recv_lo, sh->send_hi[rank - 1], bytes, hipMemcpyDeviceToDevice, s));
    }
    if (rank < size - 1) {
      HIPC(hipStreamWaitEvent(s, sh->ev_ready[rank + 1], 0));
      HIPC(hipMemcpyAsync(recv_hi, sh->send_lo[rank + 1], bytes, hipMemcpyDeviceToDevice, s));
    }
    HIPC(hipEventRecord(sh->ev_done[rank], s));
    sh->barrier();
    // nobody may overwrite a send plane before its neighbours have copied it
    if (rank > 0) HIPC(hipStreamWaitEvent(s, sh->ev_done[rank - 1], 0));
    if (rank < size - 1) HIPC(hipStreamWaitEvent(s, sh->ev_done[rank + 1], 0));
    sh->barrier();  // events are re-recorded by the next collective only after everyone has waited
    return OI_OK;
  }
  oi_status exchange_v(const void* send_lo, size_t n_send_lo, void* recv_lo, size_t n_recv_lo, const void* send_hi,
                       size_t n_send_hi, void* recv_hi, size_t n_recv_hi, hipStream_t s) override {
    if (size == 1) return OI_OK;
    (void)n_send_lo; (void)n_send_hi;  // the receiver copies: its sizes are the sender's
    sh->send_lo[rank] = send_lo;
    sh->send_hi[rank] = send_hi;
    HIPC(hipEventRecord(sh->ev_ready[rank], s));
    sh->barrier();
    if (rank > 0) {
      HIPC(hipStreamWaitEvent(s, sh->ev_ready[rank - 1], 0));
      if (n_recv_lo) HIPC(hipMemcpyAsync(recv_lo, sh->send_hi[rank - 1], n_recv_lo, hipMemcpyDeviceToDevice, s));
    }
    if (rank < size - 1) {
      HIPC(hipStreamWaitEvent(s, sh->ev_ready[rank + 1], 0));
      if (n_recv_hi) HIPC(hipMemcpyAsync(recv_hi, sh->send_lo[rank + 1], n_recv_hi, hipMemcpyDeviceToDevice, s));
    }
    HIPC(hipEventRecord(sh->ev_done[rank], s));
    sh->barrier();
    if (rank > 0) HIPC(hipStreamWaitEvent(s, sh->ev_done[rank - 1], 0));
    if (rank < size - 1) HIPC(hipStreamWaitEvent(s, sh->ev_done[rank + 1], 0));
    sh->barrier();
    return OI_OK;
  }
  oi_status exchange_ring(const void* send_lo, void* recv_lo, const void* send_hi, void* recv_hi, size_t bytes,
                          hipStream_t s) override {
    if (size == 1) {
      HIPC(hipMemcpyAsync(recv_lo, send_hi, bytes, hipMemcpyDeviceToDevice, s));
      HIPC(hipMemcpyAsync(recv_hi, send_lo, bytes, hipMemcpyDeviceToDevice, s));
      return OI_OK;
    }
    const int prev = (rank + size - 1) % size, next = (rank + 1) % size;
    sh->send_lo[rank] = send_lo;
    sh->send_hi[rank] = send_hi;
    HIPC(hipEventRecord(sh->ev_ready[rank], s));
    sh->barrier();
    HIPC(hipStreamWaitEvent(s, sh->ev_ready[prev], 0));
    HIPC(hipMemcpyAsync(recv_lo, sh->send_hi[prev], bytes, hipMemcpyDeviceToDevice, s));
    HIPC(hipStreamWaitEvent(s, sh->ev_ready[next], 0));
    HIPC(hipMemcpyAsync(recv_hi, sh->send_lo[next], bytes, hipMemcpyDeviceToDevice, s));
    HIPC(hipEventRecord(sh->ev_done[rank], s));
    sh->barrier();
    HIPC(hipStreamWaitEvent(s, sh->ev_done[prev], 0));
    HIPC(hipStreamWaitEvent(s, sh->ev_done[next], 0));
    sh->barrier();
    return OI_OK;
  }
  oi_status exchange_ring_v(const void* send_lo, size_t n_send_lo, void* recv_lo, size_t n_recv_lo, const void* send_hi,
                            size_t n_send_hi, void* recv_hi, size_t n_recv_hi, hipStream_t s) override {
    if (size == 1) {
      if (n_send_hi) HIPC(hipMemcpyAsync(recv_lo, send_hi, n_send_hi, hipMemcpyDeviceToDevice, s));
      if (n_send_lo) HIPC(hipMemcpyAsync(recv_hi, send_lo, n_send_lo, hipMemcpyDeviceToDevice, s));
      return OI_OK;
    }
    (void)n_send_lo; (void)n_send_hi;  // the receiver copies: its sizes are the sender's
    const int prev = (rank + size - 1) % size, next = (rank + 1) % size;
    sh->send_lo[rank] = send_lo;
    sh->send_hi[rank] = send_hi;
    HIPC(hipEventRecord(sh->ev_ready[rank], s));
    sh->barrier();
    HIPC(hipStreamWaitEvent(s, sh->ev_ready[prev], 0));
    if (n_recv_lo) HIPC(hipMemcpyAsync(recv_lo, sh->send_hi[prev], n_recv_lo, hipMemcpyDeviceToDevice, s));
    HIPC(hipStreamWaitEvent(s, sh->ev_ready[next], 0));
    if (n_recv_hi) HIPC(hipMemcpyAsync(recv_hi, sh->send_lo[next], n_recv_hi, hipMemcpyDeviceToDevice, s));
    HIPC(hipEventRecord(sh->ev_done[rank], s));
    sh->barrier();
    HIPC(hipStreamWaitEvent(s, sh->ev_done[prev], 0));
    HIPC(hipStreamWaitEvent(s, sh->ev_done[next], 0));
    sh->barrier();
    return OI_OK;
  }
  oi_status allreduce_sum_f64(double* dev, int n, hipStream_t s) override {
    if (size == 1) return OI_OK;
    if (n > kStageDoubles) return OI_BAD_ARG;
    HIPC(hipMemcpyAsync(sh->stage[rank], dev, sizeof(double) * n, hipMemcpyDeviceToDevice, s));
    HIPC(hipEventRecord(sh->ev_ready[rank], s));
    sh->barrier();
    for (int r = 0; r < size; ++r)
      if (r != rank) HIPC(hipStreamWaitEvent(s, sh->ev_ready[r], 0));
    oi::launch_sum_scalars(dev, sh->stage_ptrs_dev, size, n, s);  // fixed order: identical on all ranks
    HIPC(hipEventRecord(sh->ev_done[rank], s));
    sh->barrier();
    for (int r = 0; r < size; ++r)
      if (r != rank) HIPC(hipStreamWaitEvent(s, sh->ev_done[r], 0));
    sh->barrier();
    return OI_OK;
  }
  oi_status allreduce_sum_f32(float* dev, size_t n, hipStream_t s) override {
    if (size == 1) return OI_OK;
    if (size > oi::kMaxSumRanks) return OI_BAD_ARG;
    // (the scratch grows on demand: a rank that cannot get it must not leave the others at the barrier -- everybody learns of it
    // at the first rendezvous and everybody returns the error)
    int failed = 0;
    if (sh->big_cap[rank] < n) {
      if (hipStreamSynchronize(s) != hipSuccess) failed = 1;
      if (sh->big[rank]) (void)oi::dev_free(sh->big[rank]);
      sh->big[rank] = nullptr;
      sh->big_cap[rank] = 0;
      if (!failed && oi::raw_malloc((void**)&sh->big[rank], sizeof(float) * n) != hipSuccess) {
        (void)hipGetLastError();
        sh->big[rank] = nullptr;
        failed = 1;
      }
      if (!failed) sh->big_cap[rank] = n;
    }
    sh->hi32[rank * 16] = failed;
    sh->barrier();
    int any = 0;
    for (int r = 0; r < size; ++r) any |= sh->hi32[r * 16];
    sh->barrier();  // (everyone has read the flags before anyone's next collective overwrites them)
    if (any) {
      oi::set_error(failed ? "in-process transport: no device memory for the %zu-byte scratch of a large all-reduce (slab %d)"
                           : "in-process transport: another slab could not allocate the scratch of a large all-reduce (%zu bytes; this is slab %d)",
                    sizeof(float) * n, rank);
      return OI_HIP_ERROR;
    }
    sh->send_lo[rank] = dev;
    HIPC(hipEventRecord(sh->ev_ready[rank], s));
    sh->barrier();
    const float* src[oi::kMaxSumRanks] = {};
    for (int r = 0; r < size; ++r) {
      src[r] = (const float*)sh->send_lo[r];
      if (r != rank) HIPC(hipStreamWaitEvent(s, sh->ev_ready[r], 0));
    }
    oi::launch_sum_f32_ranks(sh->big[rank], src, size, n, s);  // rank order: identical on all ranks
    HIPC(hipEventRecord(sh->ev_done[rank], s));
    sh->barrier();
    for (int r = 0; r < size; ++r)
      if (r != rank) HIPC(hipStreamWaitEvent(s, sh->ev_done[r], 0));  // everyone has read my input
    HIPC(hipMemcpyAsync(dev, sh->big[rank], sizeof(float) * n, hipMemcpyDeviceToDevice, s));
    sh->barrier();
    return OI_OK;
  }
  oi_status allreduce_sum_u64_host(unsigned long long* vals, int n, hipStream_t s) override {
    if (size == 1) return OI_OK;
    if (n > 8) return OI_BAD_ARG;
    HIPC(hipStreamSynchronize(s));
    for (int i = 0; i < n; ++i) sh->hu64[rank * 8 + i] = vals[i];
    sh->barrier();
    for (int i = 0; i < n; ++i) {
      unsigned long long t = 0;
      for (int r = 0; r < size; ++r) t += sh->hu64[r * 8 + i];
      vals[i] = t;
    }
    sh->barrier();
    return OI_OK;
  }
  oi_status allreduce_max_i32_host(int* vals, int n, hipStream_t s) override {
    if (size == 1) return OI_OK;
    if (n > 16) return OI_BAD_ARG;
    HIPC(hipStreamSynchronize(s));
    for (int i = 0; i < n; ++i) sh->hi32[rank * 16 + i] = vals[i];
    sh->barrier();
    for (int i = 0; i < n; ++i) {
      int t = sh->hi32[i];
      for (int r = 1; r < size; ++r) t = sh->hi32[r * 16 + i] > t ? sh->hi32[r * 16 + i] : t;
      vals[i] = t;
    }
    sh->barrier();
    return OI_OK;
  }
};

extern "C" {

oi_status oi_comm_unique_id(void* id_out) {
  static_assert(OI_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
  if (!id_out) return OI_BAD_ARG;
  ncclUniqueId id;
  NCCLC(ncclGetUniqueId(&id));
  memcpy(id_out, &id, sizeof(id));
  return OI_OK;
}

oi_status oi_comm_create(oi_comm** out, const void* id, int nranks, int rank, int device) {
  if (!out || !id || nranks < 1 || rank < 0 || rank >= nranks) return OI_BAD_ARG;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
    oi::set_error("no HIP device");
    return OI_NO_DEVICE;
  }
  if (device >= 0) HIPC(hipSetDevice(device));
  HIPC(hipGetDevice(&device));
  auto c = std::make_unique<RcclComm>();
  c->rank = rank;
  c->size = nranks;
  c->device = device;
  c->kind = 1;
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof(uid));
  NCCLC(ncclCommInitRank(&c->nccl, nranks, uid, rank));
  // second communicator over the same ranks for the plane exchanges (also with one rank: the rehearsal on a
  // single GPU then covers the call)
  if (ncclCommSplit(c->nccl, 0, rank, &c->nccl_p2p, nullptr) != ncclSuccess || !c->nccl_p2p) {
    c->nccl_p2p = c->nccl;        // one communicator: the solver then keeps exchange and all-reduce on one stream
    c->exchange_overlaps = false;
  }
  HIPC(oi::raw_malloc(&c->scratch, 128));
  *out = c.release();
  return OI_OK;
}

oi_status oi_comm_create_local(oi_comm** out_array, int nranks, int device) {
  if (!out_array || nranks < 1 || nranks > 64) return OI_BAD_ARG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
    oi::set_error("no HIP device");
    return OI_NO_DEVICE;
  }
  if (device >= 0) HIPC(hipSetDevice(device));
  HIPC(hipGetDevice(&device));
  auto sh = std::make_shared<LocalShared>();
  sh->n = nranks;
  sh->send_lo.assign(nranks, nullptr);
  sh->send_hi.assign(nranks, nullptr);
  sh->hu64.assign(nranks * 8, 0);
  sh->hi32.assign(nranks * 16, 0);
  sh->ev_ready.resize(nranks);
  sh->ev_done.resize(nranks);
  sh->stage.resize(nranks);
  sh->big.assign(nranks, nullptr);
  sh->big_cap.assign(nranks, 0);
  for (int r = 0; r < nranks; ++r) {
    HIPC(hipEventCreateWithFlags(&sh->ev_ready[r], hipEventDisableTiming));
    HIPC(hipEventCreateWithFlags(&sh->ev_done[r], hipEventDisableTiming));
    HIPC(oi::raw_malloc((void**)&sh->stage[r], sizeof(double) * kStageDoubles));
  }
  HIPC(oi::raw_malloc((void**)&sh->stage_ptrs_dev, sizeof(double*) * nranks));
  HIPC(hipMemcpy((void*)sh->stage_ptrs_dev, sh->stage.data(), sizeof(double*) * nranks, hipMemcpyHostToDevice));
  for (int r = 0; r < nranks; ++r) {
    auto* c = new LocalComm();
    c->rank = r;
    c->size = nranks;
    c->device = device;
    c->kind = 2;
    c->sh = sh;
    out_array[r] = c;
  }
  return OI_OK;
}

void oi_comm_destroy(oi_comm* c) { delete c; }
int oi_comm_rank(const oi_comm* c) { return c ? c->rank : 0; }
int oi_comm_size(const oi_comm* c) { return c ? c->size : 1; }

oi_status oi_comm_info(const oi_comm* c, int* kind, int* ranks_collective, int* ranks_p2p, int* exchange_overlaps) {
  if (!c) { oi::set_error("%s: null communicator", __func__); return OI_BAD_ARG; }
  int k = c->kind, nc = c->size, np = c->size;
  if (const RcclComm* r = dynamic_cast<const RcclComm*>(c)) {
    NCCLC(ncclCommCount(r->nccl, &nc));
    NCCLC(ncclCommCount(r->nccl_p2p, &np));
  }
  if (kind) *kind = k;
  if (ranks_collective) *ranks_collective = nc;
  if (ranks_p2p) *ranks_p2p = np;
  if (exchange_overlaps) *exchange_overlaps = c->exchange_overlaps ? 1 : 0;
  return OI_OK;
}

}  // extern "C"
