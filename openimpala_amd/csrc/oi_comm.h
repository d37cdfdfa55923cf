// oi_comm.h -- slab-neighbour transport of the tortuosity path.
//
// The reference moves ghost cells with amrex::FillBoundary and reduces scalars with
// amrex::ParallelDescriptor / HYPRE-internal MPI (call sites: SURVEY.md 2.2).  Here a
// rank is one process owning one MI355X; z-slabs form an open chain, so the only
// messages are one contiguous ghost plane to each z-neighbour (RCCL point-to-point
// over xGMI) and a few scalars (RCCL all-reduce).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/openimpala_amd.h"

struct oi_comm {
  int rank = 0, size = 1, device = 0;
  int kind = 0;  // 1 = RCCL (one process or host thread per GPU), 2 = in-process (host threads on one GPU)
  // true when the plane exchanges may run on a second stream WHILE a collective of this communicator is in flight
  // on another (RCCL: they use a communicator of their own; two operations of one RCCL communicator must not overlap)
  bool exchange_overlaps = true;
  virtual ~oi_comm() {}
  // Sends `send_lo` to rank-1 (lands in its recv_hi) and `send_hi` to rank+1 (lands in its
  // recv_lo); chain ends skip the missing neighbour.  Asynchronous on `s`.
  virtual oi_status exchange(const void* send_lo, void* recv_lo, const void* send_hi, void* recv_hi,
                             size_t bytes, hipStream_t s) = 0;
  // the same with a size per message (compacted planes differ in length): my lower neighbour's n_send_lo is
  // its upper neighbour's n_recv_hi, and so on; a zero-length message is skipped on both sides
  virtual oi_status exchange_v(const void* send_lo, size_t n_send_lo, void* recv_lo, size_t n_recv_lo, const void* send_hi,
                               size_t n_send_hi, void* recv_hi, size_t n_recv_hi, hipStream_t s) = 0;
  // the same on a RING (rank 0 and rank size-1 are neighbours too): periodic z of the homogenization
  // cell problems.  With one rank the planes wrap onto the rank's own ghosts.
  virtual oi_status exchange_ring(const void* send_lo, void* recv_lo, const void* send_hi, void* recv_hi,
                                  size_t bytes, hipStream_t s) = 0;
  // the ring exchange with a size per message (compacted planes of a periodic problem): what I send down is my lower neighbour's
  // n_recv_hi, what I send up my upper neighbour's n_recv_lo; a zero-length message is skipped on both sides.  One rank: the
  // planes wrap onto the rank's own ghosts (n_send_hi bytes into recv_lo, n_send_lo bytes into recv_hi)
  virtual oi_status exchange_ring_v(const void* send_lo, size_t n_send_lo, void* recv_lo, size_t n_recv_lo, const void* send_hi,
                                    size_t n_send_hi, void* recv_hi, size_t n_recv_hi, hipStream_t s) = 0;
  // in-place sum of n doubles resident on the device, asynchronous on `s`
  virtual oi_status allreduce_sum_f64(double* dev, int n, hipStream_t s) = 0;
  // in-place sum of a (large) float array resident on the device, asynchronous on `s`: gathers the
  // slab-partitioned right-hand side of the first replicated multigrid level
  virtual oi_status allreduce_sum_f32(float* dev, size_t n, hipStream_t s) = 0;
  // host-value collectives for the (rare) setup steps; they synchronise `s`
  virtual oi_status allreduce_sum_u64_host(unsigned long long* vals, int n, hipStream_t s) = 0;
  virtual oi_status allreduce_max_i32_host(int* vals, int n, hipStream_t s) = 0;
};

namespace oi {
void set_error(const char* fmt, ...);
// A rank-local step (allocations, a set-up kernel) sits between two collectives of a set-up: if it fails on ONE rank and that rank
// simply returns, its neighbours wait in the next collective for good.  So the outcome of such a step is max-reduced before anyone
// goes on: every rank gets OI_OK, or every rank returns -- the failing one with its own status and message, the others with
// OI_HIP_ERROR and "another slab failed during <what>".  Without a communicator it is the identity.
inline oi_status agree_status(oi_comm* comm, oi_status local, hipStream_t s, const char* what) {
  if (!comm || comm->size <= 1) return local;
  int bad = local != OI_OK ? 1 : 0;
  const oi_status rc = comm->allreduce_max_i32_host(&bad, 1, s);
  if (local != OI_OK) return local;
  if (rc != OI_OK) return rc;
  if (bad) {
    set_error("slab %d of %d: another slab failed during %s", comm->rank, comm->size, what);
    return OI_HIP_ERROR;
  }
  return OI_OK;
}
}
