// Multi-GPU exchange for the diabat-sharded EVB step: one process per GPU, RCCL over xGMI.
// Diabats d >= 2 live on rank (d-2) mod R; force slots are laid out [rank][slot] so a single in-place ncclAllGather per
// store completes it on every rank.  The Hamiltonian entries travel the same way (2 doubles per slot).
#include <rccl/rccl.h>

#include "ctx.hpp"

static inline int slots_per_rank(const msevb_ctx* c, int D) { return D <= 1 ? 0 : (D - 1 + c->n_ranks - 1) / c->n_ranks; }

extern "C" {

int msevb_comm_unique_id(void* id128) {
  ncclUniqueId id;
  if (ncclGetUniqueId(&id) != ncclSuccess) return MSEVB_ERR_COMM;
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId size");
  memcpy(id128, &id, 128);
  return MSEVB_OK;
}

int msevb_comm_init(msevb_ctx* c, const void* id128, int rank, int n_ranks) {
  if (!c || n_ranks < 1 || rank < 0 || rank >= n_ranks) return MSEVB_ERR_ARG;
  (void)hipSetDevice(c->device);
  ncclUniqueId id;
  memcpy(&id, id128, 128);
  ncclComm_t comm;
  if (ncclCommInitRank(&comm, n_ranks, id, rank) != ncclSuccess) FAIL(MSEVB_ERR_COMM, "ncclCommInitRank failed");
  c->nccl_comm = comm; c->rank = rank; c->n_ranks = n_ranks;
  return MSEVB_OK;
}

int msevb_comm_allgather_evb(msevb_ctx* c) {
  const int D = c->D, S = slots_per_rank(c, D), R = c->n_ranks;
  if (S == 0) return MSEVB_OK;
  const size_t n3 = 3 * (size_t)c->N;
  ncclComm_t comm = (ncclComm_t)c->nccl_comm;
  static thread_local DBuf<double> d_h;
  if (d_h.alloc((size_t)R * S * 2)) FAIL(MSEVB_ERR_HIP, "hipMalloc failed (comm)");
  std::vector<double> h((size_t)R * S * 2, 0.0);
  for (int d = 1; d < D; ++d) if ((d - 1) % R == c->rank) { const int s = (d - 1) / R; h[((size_t)c->rank * S + s) * 2] = c->H[d][d]; h[((size_t)c->rank * S + s) * 2 + 1] = c->H[c->parent[d]][d]; }
  HIPCHK(hipMemcpyAsync(d_h.p, h.data(), h.size() * 8, hipMemcpyHostToDevice, c->stream));
  if (ncclGroupStart() != ncclSuccess) FAIL(MSEVB_ERR_COMM, "ncclGroupStart");
  ncclResult_t r1 = ncclAllGather(d_h.p + (size_t)c->rank * S * 2, d_h.p, (size_t)S * 2, ncclDouble, comm, c->stream);
  ncclResult_t r2 = ncclAllGather(c->d_Fdiag.p + n3 + (size_t)c->rank * S * n3, c->d_Fdiag.p + n3, (size_t)S * n3, ncclDouble, comm, c->stream);
  ncclResult_t r3 = ncclAllGather(c->d_Foff.p + (size_t)c->rank * S * n3, c->d_Foff.p, (size_t)S * n3, ncclDouble, comm, c->stream);
  if (ncclGroupEnd() != ncclSuccess || r1 != ncclSuccess || r2 != ncclSuccess || r3 != ncclSuccess) FAIL(MSEVB_ERR_COMM, "ncclAllGather failed");
  HIPCHK(hipMemcpyAsync(h.data(), d_h.p, h.size() * 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  for (int d = 1; d < D; ++d) { const int r = (d - 1) % R, s = (d - 1) / R; c->H[d][d] = h[((size_t)r * S + s) * 2]; c->H[c->parent[d]][d] = h[((size_t)r * S + s) * 2 + 1]; }
  return MSEVB_OK;
}

// Keeps the replicas bit-identical: every rank integrates all atoms, and its forces carry the rounding of its own atomic-add order;
// without this, trajectories could drift apart in the last bits until two ranks enumerate different diabat sets (different
// all-gather sizes).  One 24-byte-per-atom broadcast per step over xGMI.
int msevb_comm_broadcast_forces(msevb_ctx* c) {
  if (!c || !c->nccl_comm || c->n_ranks < 2) return MSEVB_OK;
  if (ncclBroadcast(c->d_force.p, c->d_force.p, 3 * (size_t)c->N, ncclDouble, 0, (ncclComm_t)c->nccl_comm, c->stream) != ncclSuccess)
    FAIL(MSEVB_ERR_COMM, "ncclBroadcast failed");
  return MSEVB_OK;
}

}  // extern "C"
