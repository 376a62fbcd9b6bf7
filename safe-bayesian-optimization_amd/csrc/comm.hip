// comm.hip -- RCCL plumbing: one process per GPU, ranks joined over xGMI with a broadcast unique id.
// The reference has no communication layer at all; these collectives exist only
// because the candidate set is sharded across the GPUs of a node.
#include <chrono>
#include <cstring>
#include <vector>
#include <rccl/rccl.h>
#include "internal.hpp"

namespace sbo {
static int nccl_fail(ncclResult_t r, const char* what) {
  return fail(SBO_E_COMM, std::string("RCCL: ") + ncclGetErrorString(r) + " in " + what);
}
#define SBO_NCCL(x)                                        \
  do {                                                     \
    ncclResult_t r__ = (x);                                \
    if (r__ != ncclSuccess) return sbo::nccl_fail(r__, #x); \
  } while (0)

// accounting of one collective (sbo_profile.comm_*): bytes on the send side, and with option comm_events an event pair
struct CommScope {
  sbo_ctx* c;
  int slot = -1;
  std::chrono::steady_clock::time_point t0;
  CommScope(sbo_ctx* c_, size_t bytes) : c(c_) {
    c->comm_bytes += (long long)bytes;
    ++c->comm_calls;
    t0 = std::chrono::steady_clock::now();
    if (c->comm_events && c->comm && c->comm_nev + 2 <= 16) {
      slot = c->comm_nev;
      c->comm_nev += 2;
      for (int k = 0; k < 2; ++k)
        if (!c->comm_ev[slot + k] && hipEventCreate(&c->comm_ev[slot + k]) != hipSuccess) slot = -1;
      if (slot >= 0) (void)hipEventRecord(c->comm_ev[slot], c->stream);
    }
  }
  ~CommScope() {
    if (slot >= 0) (void)hipEventRecord(c->comm_ev[slot + 1], c->stream);
    if (!c->comm) c->comm_host_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  }
};

// host-staged collective for the rehearsal transport
static int relay_reduce(sbo_ctx* c, void* dev, size_t count, int elem, int op) {
  if (!c->relay_allreduce) return fail(SBO_E_COMM, "no transport: neither an RCCL communicator nor relay callbacks are installed");
  std::vector<unsigned char> h(count * 8);
  SBO_HIP(hipMemcpyAsync(h.data(), dev, h.size(), hipMemcpyDeviceToHost, c->stream));
  SBO_HIP(hipStreamSynchronize(c->stream));
  if (c->relay_allreduce(c->relay_user, h.data(), (int64_t)count, elem, op) != 0)
    return fail(SBO_E_COMM, "relay all-reduce callback failed");
  SBO_HIP(hipMemcpyAsync(dev, h.data(), h.size(), hipMemcpyHostToDevice, c->stream));
  SBO_HIP(hipStreamSynchronize(c->stream));
  return SBO_OK;
}

// in-place all-reduce helpers used by the sweeps (no-ops for a single rank)
int comm_allreduce_max_u64(sbo_ctx* c, unsigned long long* dev, int count) {
  if (!multi_rank(c)) return SBO_OK;
  CommScope cs_(c, sizeof(unsigned long long) * (size_t)count);
  if (!c->comm) return relay_reduce(c, dev, count, 0, 1);
  SBO_NCCL(ncclAllReduce(dev, dev, count, ncclUint64, ncclMax, (ncclComm_t)c->comm, c->stream));
  return SBO_OK;
}
int comm_allreduce_min_u64(sbo_ctx* c, unsigned long long* dev, int count) {
  if (!multi_rank(c)) return SBO_OK;
  CommScope cs_(c, sizeof(unsigned long long) * (size_t)count);
  if (!c->comm) return relay_reduce(c, dev, count, 0, 2);
  SBO_NCCL(ncclAllReduce(dev, dev, count, ncclUint64, ncclMin, (ncclComm_t)c->comm, c->stream));
  return SBO_OK;
}
int comm_allreduce_sum_f64(sbo_ctx* c, double* dev, int count) {
  if (!multi_rank(c)) return SBO_OK;
  CommScope cs_(c, sizeof(double) * (size_t)count);
  if (!c->comm) return relay_reduce(c, dev, count, 1, 0);
  SBO_NCCL(ncclAllReduce(dev, dev, count, ncclDouble, ncclSum, (ncclComm_t)c->comm, c->stream));
  return SBO_OK;
}
int comm_allgather_bytes(sbo_ctx* c, const void* send, void* recv, size_t bytes_per_rank) {
  if (!multi_rank(c)) return SBO_OK;
  CommScope cs_(c, bytes_per_rank);
  if (!c->comm) {
    if (!c->relay_allgather) return fail(SBO_E_COMM, "no transport: neither an RCCL communicator nor relay callbacks are installed");
    std::vector<unsigned char> hs(bytes_per_rank), hr(bytes_per_rank * c->world);
    SBO_HIP(hipMemcpyAsync(hs.data(), send, hs.size(), hipMemcpyDeviceToHost, c->stream));
    SBO_HIP(hipStreamSynchronize(c->stream));
    if (c->relay_allgather(c->relay_user, hs.data(), hr.data(), (int64_t)bytes_per_rank) != 0)
      return fail(SBO_E_COMM, "relay all-gather callback failed");
    SBO_HIP(hipMemcpyAsync(recv, hr.data(), hr.size(), hipMemcpyHostToDevice, c->stream));
    SBO_HIP(hipStreamSynchronize(c->stream));
    return SBO_OK;
  }
  SBO_NCCL(ncclAllGather(send, recv, bytes_per_rank, ncclUint8, (ncclComm_t)c->comm, c->stream));
  return SBO_OK;
}
}  // namespace sbo

using namespace sbo;

extern "C" {

int sbo_comm_unique_id(void* id_out) {
  if (!id_out) return fail(SBO_E_INVALID, "id_out is NULL");
  static_assert(sizeof(ncclUniqueId) <= SBO_COMM_ID_BYTES, "unique id does not fit");
  ncclUniqueId id;
  SBO_NCCL(ncclGetUniqueId(&id));
  memset(id_out, 0, SBO_COMM_ID_BYTES);
  memcpy(id_out, &id, sizeof(id));
  return SBO_OK;
}

int sbo_comm_init(sbo_ctx* c, int world_size, int rank, const void* id) {
  if (!c) return fail(SBO_E_INVALID, "ctx is NULL");
  if (world_size < 1 || rank < 0 || rank >= world_size) return fail(SBO_E_INVALID, "bad world_size / rank");
  if (c->comm) return fail(SBO_E_INVALID, "communicator already initialised");
  // A one-rank world needs no communicator: the collectives are identities and are skipped.  With an id it still gets a
  // real one (option "comm_selftest" then sends C1 / C2 / C3 through RCCL anyway: the one-GPU test of the RCCL calls).
  if (world_size == 1 && !id) {
    c->world = 1;
    c->rank = 0;
    return SBO_OK;
  }
  if (!id) return fail(SBO_E_INVALID, "id is NULL");
  SBO_HIP(hipSetDevice(c->device));
  // the library is compiled against /opt/rocm's <rccl/rccl.h>; the process may have mapped another librccl first (a
  // launcher that imported torch): refuse a major-version mismatch instead of calling through a different ABI
  int ver = 0;
  SBO_NCCL(ncclGetVersion(&ver));
  if (ver / 10000 != NCCL_MAJOR)
    return fail(SBO_E_COMM, "RCCL version mismatch: loaded " + std::to_string(ver) + ", compiled against " + std::to_string(NCCL_VERSION_CODE));
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof(uid));
  ncclComm_t comm;
  SBO_NCCL(ncclCommInitRank(&comm, world_size, uid, rank));
  // world / rank change only once the communicator exists: a failed init leaves the context single-rank and usable
  c->comm = comm;
  c->world = world_size;
  c->rank = rank;
  return SBO_OK;
}

int sbo_comm_init_relay(sbo_ctx* c, int world_size, int rank, sbo_relay_allreduce_fn allreduce,
                        sbo_relay_allgather_fn allgather, void* user) {
  if (!c) return fail(SBO_E_INVALID, "ctx is NULL");
  if (world_size < 1 || rank < 0 || rank >= world_size) return fail(SBO_E_INVALID, "bad world_size / rank");
  if (world_size > 1 && (!allreduce || !allgather)) return fail(SBO_E_INVALID, "relay callbacks are NULL");
  if (c->comm) {   // switching transports: drop the RCCL communicator
    ncclCommDestroy((ncclComm_t)c->comm);
    c->comm = nullptr;
  }
  c->world = world_size;
  c->rank = rank;
  c->relay_allreduce = allreduce;
  c->relay_allgather = allgather;
  c->relay_user = user;
  return SBO_OK;
}

int sbo_comm_barrier(sbo_ctx* c) {
  if (!c) return fail(SBO_E_INVALID, "ctx is NULL");
  if (multi_rank(c)) {
    int rc = comm_allreduce_sum_f64(c, (double*)c->scal.p + 500, 1);
    if (rc) return rc;
  }
  SBO_HIP(hipStreamSynchronize(c->stream));
  return SBO_OK;
}

int sbo_comm_destroy_internal(sbo_ctx* c) {
  if (c)
    for (auto& ev : c->comm_ev)
      if (ev) { (void)hipEventDestroy(ev); ev = nullptr; }
  if (c && c->comm) {
    ncclCommDestroy((ncclComm_t)c->comm);
    c->comm = nullptr;
  }
  return SBO_OK;
}

}  // extern "C"
