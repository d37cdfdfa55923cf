// fake_rccl.cpp -- TEST DOUBLE of the RCCL entry points oi_comm.cpp's RcclComm calls, for a box with ONE GPU.
//
// Why: RcclComm (one rank per GPU: ghost-plane send/recv in groups, scalar all-reduces, a split communicator) has never run with
// more than one rank -- no multi-GPU node has been available to any round -- and RCCL refuses two ranks on one device.  This file
// implements the same C entry points IN-PROCESS: ranks are host threads of one process that all drive the same GPU.  Loaded with
// RTLD_GLOBAL BEFORE libopenimpala_amd.so (tests/test_fake_rccl_gpu.py does that in a child process), it takes the place of
// librccl's symbols, and the UNCHANGED RcclComm code runs its real call sequence with 2, 3, 4 ranks.
//
// What it models (NCCL's documented point-to-point semantics): operations between ncclGroupStart and ncclGroupEnd are issued
// together; between one pair of ranks the i-th send of the sender matches the i-th receive of the receiver, in the order each
// side issued them; a send and its receive must carry the same byte count (a mismatch is reported as an error here, where the real
// library would hang or corrupt); a collective returns on every rank only when every rank has called it.  A rank that waits longer
// than 30 s for its peer fails with ncclInternalError instead of hanging (what a dead-lock of the call pattern would look like).
// Two modes.  Default: every call synchronises its stream, moves the bytes with waited-for device-to-device copies and returns when
// its operations are complete -- stricter ordering than the real library.  FAKE_RCCL_ASYNC=1: send / receive are enqueued on the
// caller's streams and ordered by events only, as in the real library (run_group_async); =2 also holds every receive back by 2 ms.
// How much that proves is limited: a build of the solver WITHOUT its final "solver stream waits for the exchange" still gave the
// right answers under mode 2 on this box (round 5, gpurun_out only) -- HIP multiplexes the streams of all rank threads onto a few
// hardware queues, which orders more than the events do -- so the asynchronous mode checks the event bookkeeping for dead-locks
// and use-after-destroy, not for missing waits.  Not modelled in any mode: the collectives' asynchrony, xGMI, RCCL's own kernels.
// Test infrastructure only: nothing in the product links or loads it.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <type_traits>
#include <vector>

namespace {
struct Msg {
  const void* ptr;
  size_t bytes;
  bool taken = false;
  // FAKE_RCCL_ASYNC=1: `ready` is recorded on the sender's stream when the send is issued, `done` on the receiver's stream behind its
  // copy; each side's STREAM waits for the other's event, no host thread waits for the device (see run_group)
  hipEvent_t ready = nullptr, done = nullptr;
  ~Msg() {
    if (ready) (void)hipEventDestroy(ready);
    if (done) (void)hipEventDestroy(done);
  }
};
struct World {
  int n = 0;
  std::mutex m;
  std::condition_variable cv;
  int joined = 0, left = 0;
  std::map<std::pair<int, int>, std::deque<std::shared_ptr<Msg>>> box;  // (src, dst) -> messages in the order they were sent
  // collectives: every rank deposits, the last one to arrive opens the generation
  unsigned long long coll_gen = 0;
  int coll_arrived = 0;
  std::vector<std::vector<unsigned char>> stage;
  int split_seq = 0;
  std::map<int, std::shared_ptr<World>> splits;  // split number -> child world
};
std::mutex g_reg_mutex;
std::map<std::string, std::shared_ptr<World>> g_registry;
unsigned long long g_next_id = 1;
constexpr auto kPatience = std::chrono::seconds(30);

struct Op {
  bool send;
  void* buf;
  size_t bytes;
  int peer;
  struct ncclComm* comm;
  hipStream_t stream;
};
thread_local int t_depth = 0;
thread_local std::vector<Op> t_ops;

size_t dtype_bytes(ncclDataType_t t) {
  switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
  }
}
}  // namespace

struct ncclComm {
  std::shared_ptr<World> w;
  int rank = 0;
  int my_splits = 0;
};

namespace {
template <class Pred>
bool wait_for(World& w, std::unique_lock<std::mutex>& lk, Pred p, const char* what, int rank) {
  if (w.cv.wait_for(lk, kPatience, p)) return true;
  fprintf(stderr, "[fake_rccl] rank %d of %d gave up waiting for %s (a dead-lock of the call pattern)\n", rank, w.n, what);
  return false;
}

// every rank calls it with the same `n` bytes: on return every rank's `out` holds all ranks' contributions in rank order
ncclResult_t gather_all(ncclComm* c, const void* mine, size_t n, std::vector<std::vector<unsigned char>>* out) {
  World& w = *c->w;
  std::unique_lock<std::mutex> lk(w.m);
  const unsigned long long gen = w.coll_gen;
  if (w.stage.size() != (size_t)w.n) w.stage.assign(w.n, {});
  w.stage[c->rank].assign((const unsigned char*)mine, (const unsigned char*)mine + n);
  if (++w.coll_arrived == w.n) {
    w.coll_arrived = 0;
    ++w.coll_gen;
    w.cv.notify_all();
  } else if (!wait_for(w, lk, [&] { return w.coll_gen != gen; }, "a collective (not every rank called it)", c->rank)) {
    return ncclInternalError;
  }
  *out = w.stage;  // (copied under the lock; the next collective overwrites the staging only after everyone has left this one:
  // a second rendezvous below)
  const unsigned long long gen2 = w.coll_gen;
  if (++w.coll_arrived == w.n) {
    w.coll_arrived = 0;
    ++w.coll_gen;
    w.cv.notify_all();
  } else if (!wait_for(w, lk, [&] { return w.coll_gen != gen2; }, "the end of a collective", c->rank)) {
    return ncclInternalError;
  }
  return ncclSuccess;
}

int async_level() {
  static const int v = getenv("FAKE_RCCL_ASYNC") ? atoi(getenv("FAKE_RCCL_ASYNC")) : 0;
  return v;
}
bool async_mode() { return async_level() != 0; }
// FAKE_RCCL_ASYNC=2: every receive's copy is held back by ~2 ms of spinning on its stream -- a slow link.  A consumer that does not
// wait for the exchange's stream then reads the OLD plane for certain, instead of winning a race it would lose on a loaded node.
__global__ void k_fake_delay(long long ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {}
}
bool delay_ok(hipStream_t s) {
  if (async_level() < 2) return true;
  hipLaunchKernelGGL(k_fake_delay, dim3(1), dim3(1), 0, s, 200000ll);  // wall_clock64 counts at 100 MHz: 2 ms
  return hipGetLastError() == hipSuccess;
}

// ASYNCHRONOUS mode (FAKE_RCCL_ASYNC=1): like the real library, a send / receive is work enqueued on the caller's stream.  The sender
// records an event where the send was issued; the receiver's stream waits for it, copies, and records a second event the sender's
// stream waits for before anything later on that stream may overwrite the buffer.  No stream is synchronised with the host, so a
// consumer on ANOTHER stream that does not wait for the exchange's stream (a missing event in the solver) reads stale planes here
// as it would on a real node.  (The host threads still meet to hand the events over; collectives stay host-synchronous.)
ncclResult_t run_group_async(std::vector<Op>& ops) {
  std::vector<std::shared_ptr<Msg>> mine;
  for (const Op& o : ops) {
    if (!o.send) continue;
    World& w = *o.comm->w;
    auto m = std::make_shared<Msg>();
    m->ptr = o.buf;
    m->bytes = o.bytes;
    if (hipEventCreateWithFlags(&m->ready, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&m->done, hipEventDisableTiming) != hipSuccess ||
        hipEventRecord(m->ready, o.stream) != hipSuccess)
      return ncclUnhandledCudaError;
    {
      std::lock_guard<std::mutex> lk(w.m);
      w.box[{o.comm->rank, o.peer}].push_back(m);
    }
    w.cv.notify_all();
    mine.push_back(m);
  }
  ncclResult_t rc = ncclSuccess;
  for (const Op& o : ops) {
    if (o.send) continue;
    World& w = *o.comm->w;
    std::shared_ptr<Msg> m;
    {
      std::unique_lock<std::mutex> lk(w.m);
      auto& q = w.box[{o.peer, o.comm->rank}];
      if (!wait_for(w, lk, [&] { return !q.empty(); }, "a send that matches its receive", o.comm->rank)) return ncclInternalError;
      m = q.front();
      q.pop_front();
    }
    if (m->bytes != o.bytes) {
      fprintf(stderr, "[fake_rccl] rank %d: receive of %zu bytes from rank %d met a send of %zu bytes\n", o.comm->rank, o.bytes, o.peer, m->bytes);
      rc = ncclInvalidArgument;
    } else if (hipStreamWaitEvent(o.stream, m->ready, 0) != hipSuccess ||
               !delay_ok(o.stream) ||
               (o.bytes && hipMemcpyAsync(o.buf, m->ptr, o.bytes, hipMemcpyDeviceToDevice, o.stream) != hipSuccess) ||
               hipEventRecord(m->done, o.stream) != hipSuccess) {
      rc = ncclUnhandledCudaError;
    }
    {
      std::lock_guard<std::mutex> lk(w.m);
      m->taken = true;  // (= the copy is ENQUEUED and `done` recorded behind it)
    }
    w.cv.notify_all();
  }
  for (size_t k = 0, s = 0; k < ops.size(); ++k) {
    if (!ops[k].send) continue;
    World& w = *ops[k].comm->w;
    auto m = mine[s++];
    {
      std::unique_lock<std::mutex> lk(w.m);
      if (!wait_for(w, lk, [&] { return m->taken; }, "the receive that matches a send", ops[k].comm->rank)) return ncclInternalError;
    }
    if (hipStreamWaitEvent(ops[k].stream, m->done, 0) != hipSuccess) rc = ncclUnhandledCudaError;  // the buffer is the sender's again behind this point
  }
  return rc;
}

ncclResult_t run_group(std::vector<Op>& ops) {
  if (async_mode()) return run_group_async(ops);
  // data the operations read must be complete: the stream(s) they were issued on
  for (const Op& o : ops)
    if (hipStreamSynchronize(o.stream) != hipSuccess) return ncclUnhandledCudaError;
  std::vector<std::shared_ptr<Msg>> mine;
  for (const Op& o : ops) {  // sends first: nobody blocks before its own messages are out
    if (!o.send) continue;
    World& w = *o.comm->w;
    auto m = std::make_shared<Msg>();
    m->ptr = o.buf;
    m->bytes = o.bytes;
    {
      std::lock_guard<std::mutex> lk(w.m);
      w.box[{o.comm->rank, o.peer}].push_back(m);
    }
    w.cv.notify_all();
    mine.push_back(m);
  }
  ncclResult_t rc = ncclSuccess;
  for (const Op& o : ops) {  // receives in the order they were issued
    if (o.send) continue;
    World& w = *o.comm->w;
    std::shared_ptr<Msg> m;
    {
      std::unique_lock<std::mutex> lk(w.m);
      auto& q = w.box[{o.peer, o.comm->rank}];
      if (!wait_for(w, lk, [&] { return !q.empty(); }, "a send that matches its receive", o.comm->rank)) return ncclInternalError;
      m = q.front();
      q.pop_front();
    }
    if (m->bytes != o.bytes) {
      fprintf(stderr, "[fake_rccl] rank %d: receive of %zu bytes from rank %d met a send of %zu bytes\n", o.comm->rank, o.bytes, o.peer, m->bytes);
      rc = ncclInvalidArgument;
    } else if (o.bytes && (hipMemcpyAsync(o.buf, m->ptr, o.bytes, hipMemcpyDeviceToDevice, o.stream) != hipSuccess ||
                           hipStreamSynchronize(o.stream) != hipSuccess)) {
      // (on the receive's own stream and waited for: a blocking hipMemcpy between device buffers may return before the bytes have
      // landed, and the null stream does not order with the solver's non-blocking streams)
      rc = ncclUnhandledCudaError;
    }
    {
      std::lock_guard<std::mutex> lk(w.m);
      m->taken = true;
    }
    w.cv.notify_all();
  }
  for (size_t k = 0, s = 0; k < ops.size(); ++k) {  // my send buffers are free again once their receivers have copied them
    if (!ops[k].send) continue;
    World& w = *ops[k].comm->w;
    std::unique_lock<std::mutex> lk(w.m);
    auto m = mine[s++];
    if (!wait_for(w, lk, [&] { return m->taken; }, "the receive that matches a send", ops[k].comm->rank)) return ncclInternalError;
  }
  return rc;
}
}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  std::lock_guard<std::mutex> lk(g_reg_mutex);
  memset(id, 0, sizeof(*id));
  snprintf(id->internal, sizeof(id->internal), "fake-rccl-%llu", g_next_id++);
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
  if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  std::shared_ptr<World> w;
  {
    std::lock_guard<std::mutex> lk(g_reg_mutex);
    auto& slot = g_registry[std::string(id.internal, strnlen(id.internal, sizeof(id.internal)))];
    if (!slot) { slot = std::make_shared<World>(); slot->n = nranks; }
    w = slot;
  }
  if (w->n != nranks) return ncclInvalidArgument;
  auto* c = new ncclComm();
  c->w = w;
  c->rank = rank;
  std::unique_lock<std::mutex> lk(w->m);
  ++w->joined;
  w->cv.notify_all();
  if (!wait_for(*w, lk, [&] { return w->joined >= w->n; }, "the other ranks' ncclCommInitRank", rank)) { delete c; return ncclInternalError; }
  *comm = c;
  return ncclSuccess;
}

ncclResult_t ncclCommSplit(ncclComm_t comm, int color, int key, ncclComm_t* newcomm, ncclConfig_t*) {
  if (!comm || !newcomm) return ncclInvalidArgument;
  (void)color; (void)key;  // (oi_comm.cpp splits every rank into ONE colour, keyed by its rank: the same membership and order)
  World& w = *comm->w;
  std::shared_ptr<World> child;
  {
    std::lock_guard<std::mutex> lk(w.m);
    auto& slot = w.splits[comm->my_splits];
    if (!slot) { slot = std::make_shared<World>(); slot->n = w.n; }
    child = slot;
  }
  ++comm->my_splits;
  auto* c = new ncclComm();
  c->w = child;
  c->rank = comm->rank;
  std::unique_lock<std::mutex> lk(child->m);
  ++child->joined;
  child->cv.notify_all();
  if (!wait_for(*child, lk, [&] { return child->joined >= child->n; }, "the other ranks' ncclCommSplit", comm->rank)) { delete c; return ncclInternalError; }
  *newcomm = c;
  return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t comm, int* count) {
  if (!comm || !count) return ncclInvalidArgument;
  *count = comm->w->n;
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
  delete comm;
  return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r) {
  switch (r) {
    case ncclSuccess: return "no error (fake_rccl)";
    case ncclUnhandledCudaError: return "unhandled HIP error (fake_rccl)";
    case ncclInternalError: return "internal error: a rank waited for its peer in vain (fake_rccl)";
    case ncclInvalidArgument: return "invalid argument / mismatched message sizes (fake_rccl)";
    default: return "error (fake_rccl)";
  }
}

ncclResult_t ncclGroupStart() {
  ++t_depth;
  return ncclSuccess;
}

ncclResult_t ncclGroupEnd() {
  if (t_depth <= 0) return ncclInvalidUsage;
  if (--t_depth > 0) return ncclSuccess;
  std::vector<Op> ops;
  ops.swap(t_ops);
  return run_group(ops);
}

static ncclResult_t p2p(bool send, void* buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t s) {
  if (!comm || peer < 0 || peer >= comm->w->n || peer == comm->rank || !dtype_bytes(dt)) return ncclInvalidArgument;
  t_ops.push_back(Op{send, buf, count * dtype_bytes(dt), peer, comm, s});
  if (t_depth > 0) return ncclSuccess;
  std::vector<Op> ops;
  ops.swap(t_ops);
  return run_group(ops);
}
ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t s) {
  return p2p(true, const_cast<void*>(buf), count, dt, peer, comm, s);
}
ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t s) {
  return p2p(false, buf, count, dt, peer, comm, s);
}

ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t comm, hipStream_t s) {
  if (!comm || !dtype_bytes(dt) || t_depth > 0) return ncclInvalidArgument;
  const size_t nb = count * dtype_bytes(dt);
  if (hipStreamSynchronize(s) != hipSuccess) return ncclUnhandledCudaError;
  std::vector<unsigned char> mine(nb);
  if (nb && hipMemcpy(mine.data(), sendbuff, nb, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
  std::vector<std::vector<unsigned char>> all;
  const ncclResult_t rc = gather_all(comm, mine.data(), nb, &all);
  if (rc != ncclSuccess) return rc;
  for (const auto& a : all)
    if (a.size() != nb) { fprintf(stderr, "[fake_rccl] all-reduce: ranks disagree on the count\n"); return ncclInvalidArgument; }
  // rank order, like a fixed-order tree would: the same result on every rank
  auto reduce = [&](auto* out) {
    using T = std::remove_pointer_t<decltype(out)>;
    for (size_t i = 0; i < count; ++i) {
      T acc;
      memcpy(&acc, all[0].data() + i * sizeof(T), sizeof(T));
      for (size_t r = 1; r < all.size(); ++r) {
        T v;
        memcpy(&v, all[r].data() + i * sizeof(T), sizeof(T));
        if (op == ncclSum) acc = (T)(acc + v);
        else if (op == ncclMax) acc = v > acc ? v : acc;
        else if (op == ncclMin) acc = v < acc ? v : acc;
      }
      out[i] = acc;
    }
  };
  std::vector<unsigned char> res(nb);
  if (op != ncclSum && op != ncclMax && op != ncclMin) return ncclInvalidArgument;
  switch (dt) {
    case ncclFloat64: reduce((double*)res.data()); break;
    case ncclFloat32: reduce((float*)res.data()); break;
    case ncclUint64: reduce((unsigned long long*)res.data()); break;
    case ncclInt64: reduce((long long*)res.data()); break;
    case ncclInt32: reduce((int*)res.data()); break;
    case ncclUint32: reduce((unsigned*)res.data()); break;
    default: return ncclInvalidArgument;
  }
  if (nb && hipMemcpy(recvbuff, res.data(), nb, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
  return ncclSuccess;
}

// tells the test that THIS library answered (and not librccl)
int fake_rccl_present(void) { return 1; }

}  // extern "C"
