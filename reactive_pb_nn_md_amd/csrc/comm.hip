// Multi-GPU exchange for the diabat-sharded EVB step: one process per GPU, RCCL over xGMI.
// Diabats d >= 2 live on rank (d-2) mod R.  Per step: one tiny all-gather (Hamiltonian entries + principal energy slots, aux2 stream)
// and, strictly after it, one all-reduce of the partial Hellmann-Feynman forces (24 B per atom, main stream): the two collectives of a
// communicator are never in flight together.  Force stores never travel.
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
  c->shard_principal = n_ranks > 1;      // the partial principal forces meet in the all-reduce of the blend
  return MSEVB_OK;
}

// ONE small all-gather per step.  Per rank: the Hamiltonian entries of its diabats (2 doubles per slot: H_dd, H_pd), the principal
// diabat's H'_11 as rank 0 has it, and the rank's eight principal energy slots (S_ELEC_PAIR .. S_RECIP; partial sums when the principal
// diabat is sharded, replicated slots zeroed on ranks != 0 by the caller).  Every rank then holds the same bytes, assembles the same
// Hamiltonian and sums the energy slots in rank order: identical eigenvectors and identical topology decisions on all ranks.
// The force stores stay where they were computed: every rank blends ITS diabats and the partial forces are summed (msevb_comm_allreduce_forces).
int msevb_comm_allgather_evb(msevb_ctx* c, int with_energies) {
  const int D = c->D, S = slots_per_rank(c, D), R = c->n_ranks, P = 2 * S + 1 + 8;
  ncclComm_t comm = (ncclComm_t)c->nccl_comm;
  DBuf<double>& d_h = c->d_comm_h;
  if (d_h.alloc((size_t)R * P)) FAIL(MSEVB_ERR_HIP, "hipMalloc failed (comm)");
  std::vector<double> h((size_t)R * P, 0.0);
  double* mine = h.data() + (size_t)c->rank * P;
  for (int d = 1; d < D; ++d) if ((d - 1) % R == c->rank) { const int s = (d - 1) / R; mine[2 * s] = c->H[d][d]; mine[2 * s + 1] = c->H[c->parent[d]][d]; }
  mine[2 * S] = c->H[0][0];
  if (with_energies) for (int k = 0; k < 8; ++k) mine[2 * S + 1 + k] = c->h_energies[S_ELEC_PAIR + k];      // S_ELEC_PAIR .. S_RECIP are contiguous
  HIPCHK(hipMemcpyAsync(d_h.p + (size_t)c->rank * P, mine, (size_t)P * 8, hipMemcpyHostToDevice, c->aux2));
  if (ncclAllGather(d_h.p + (size_t)c->rank * P, d_h.p, (size_t)P, ncclDouble, comm, c->aux2) != ncclSuccess) FAIL(MSEVB_ERR_COMM, "ncclAllGather failed");
  HIPCHK(hipMemcpyAsync(h.data(), d_h.p, h.size() * 8, hipMemcpyDeviceToHost, c->aux2));
  HIPCHK(hipStreamSynchronize(c->aux2));      // aux2 already holds the energy downloads: the main stream keeps running
  for (int d = 1; d < D; ++d) { const int r = (d - 1) % R, s = (d - 1) / R; c->H[d][d] = h[(size_t)r * P + 2 * s]; c->H[c->parent[d]][d] = h[(size_t)r * P + 2 * s + 1]; }
  c->H[0][0] = h[2 * S];
  if (with_energies) for (int k = 0; k < 8; ++k) { double e = 0.0; for (int r = 0; r < R; ++r) e += h[(size_t)r * P + 2 * S + 1 + k]; c->h_energies[S_ELEC_PAIR + k] = e; }
  return MSEVB_OK;
}

// Stand-alone sum of the eight principal energy slots over the ranks (the EVB force call itself carries them in its all-gather).
int msevb_comm_allreduce_energies(msevb_ctx* c) {
  if (!c || !c->nccl_comm) return MSEVB_OK;
  DBuf<double>& d_e = c->d_comm_e;
  if (d_e.alloc(8)) FAIL(MSEVB_ERR_HIP, "hipMalloc failed (comm)");
  HIPCHK(hipMemcpyAsync(d_e.p, c->h_energies + S_ELEC_PAIR, 8 * sizeof(double), hipMemcpyHostToDevice, c->aux2));
  if (ncclAllReduce(d_e.p, d_e.p, 8, ncclDouble, ncclSum, (ncclComm_t)c->nccl_comm, c->aux2) != ncclSuccess) FAIL(MSEVB_ERR_COMM, "ncclAllReduce failed");
  HIPCHK(hipMemcpyAsync(c->h_energies + S_ELEC_PAIR, d_e.p, 8 * sizeof(double), hipMemcpyDeviceToHost, c->aux2));
  HIPCHK(hipStreamSynchronize(c->aux2));
  return MSEVB_OK;
}

// Sum of the ranks' partial Hellmann-Feynman forces, in place; the result is bit-identical on every rank, which keeps the replicated
// trajectories identical (every rank integrates all atoms).  24 bytes per atom over xGMI instead of all-gathering 2 (D-1) force stores.
int msevb_comm_allreduce_forces(msevb_ctx* c) {
  if (!c || !c->nccl_comm) return MSEVB_OK;
  if (ncclAllReduce(c->d_force.p, c->d_force.p, 3 * (size_t)c->N, ncclDouble, ncclSum, (ncclComm_t)c->nccl_comm, c->stream) != ncclSuccess)
    FAIL(MSEVB_ERR_COMM, "ncclAllReduce failed");
  return MSEVB_OK;
}

// called by msevb_destroy
void msevb_comm_release(msevb_ctx* c) {
  if (!c || !c->nccl_comm) return;
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->aux2) (void)hipStreamSynchronize(c->aux2);
  (void)ncclCommDestroy((ncclComm_t)c->nccl_comm);
  c->nccl_comm = nullptr;
}

}  // extern "C"
