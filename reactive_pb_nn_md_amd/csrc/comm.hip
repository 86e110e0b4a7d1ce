// Multi-GPU exchange for the diabat-sharded EVB step: one process per GPU, RCCL over xGMI.
// Diabats d >= 2 live on rank (d-2) mod R.  Per step, since round 5: one tiny all-gather of Hamiltonian entries, in place and device to device on the grid stream
// (the rank's slot is written by its energy-fetch kernel, the gathered matrix is solved on the device by every rank), one 8-double all-reduce of the principal energy slots on
// the descriptor stream, one all-reduce of the partial Hellmann-Feynman forces (24 B per atom) on the main stream behind the pre-enqueued blend - issued in this order on every
// rank and ordered by the data they carry.  Force stores never travel.  Option evb_device_solve_ranks = 0: round 4's step - the all-gather's payload staged through pinned host
// memory (msevb_comm_allgather_evb, below), the host's Jacobi between it and the all-reduce.
#include <rccl/rccl.h>

#include <chrono>

#include "ctx.hpp"


// The payload of the per-step all-gather starts and ends on the host (the Hamiltonian entries are assembled there from the published
// energies, the gathered ones feed the host's Jacobi).  No copy commands and no stream synchronisation around the collective: a kernel
// reads the rank's payload straight from pinned host memory, a second one writes the gathered payloads back to pinned memory followed by
// a sequence number the host spins on (a hipStreamSynchronize sleeps on an interrupt: 50-100 us to wake up, on the step's critical path).
namespace {
__global__ void k_comm_stage(const double* __restrict__ host_src, double* __restrict__ dst, int n) {
  for (int k = threadIdx.x; k < n; k += blockDim.x) dst[k] = host_src[k];
}
__global__ void k_comm_publish(const double* __restrict__ src, double* __restrict__ host_dst, int n, volatile long long* flag, long long seq) {
  for (int k = threadIdx.x; k < n; k += blockDim.x) host_dst[k] = src[k];
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) *flag = seq;
}
// (diagnostic / test) holds the stream for `ticks` of the 100 MHz wall clock, then ends: what a collective with a missing peer looks like to
// the host, for a bounded time (test_comm_wait_gives_up_on_a_missing_peer)
__global__ void k_comm_stall(long long ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}
// The host's wait for the gathered payload.  It ENDS: when the sequence number arrives, when the stream reports an error, when the stream
// is idle without the number - or after comm_timeout_ms (option, default 30 s) of wall clock, because a peer that died or never entered
// the collective leaves hipStreamQuery at hipErrorNotReady for ever (the round-3 loop spun on that without a limit).
int comm_wait(msevb_ctx* c, long long seq) {
  volatile long long* flag = c->h_comm_flag;
  const auto t0 = std::chrono::steady_clock::now();
  for (long spins = 1;; ++spins) {
    if (*flag >= seq) return MSEVB_OK;
    if ((spins & 0x3fff) == 0) {      // a collective that failed never publishes: ask the stream now and then
      const hipError_t e = hipStreamQuery(c->aux2);
      if (e == hipSuccess) { if (*flag >= seq) return MSEVB_OK; FAIL(MSEVB_ERR_COMM, "all-gather finished without publishing its result"); }
      if (e != hipErrorNotReady) FAIL(MSEVB_ERR_HIP, std::string("HIP error behind the EVB all-gather: ") + hipGetErrorString(e));
      if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(c->comm_timeout_ms)) {
        char b[160]; snprintf(b, sizeof b, "EVB all-gather: no result after %d ms (a peer rank is missing or has stopped); rank %d of %d", c->comm_timeout_ms, c->rank, c->n_ranks);
        FAIL(MSEVB_ERR_COMM, b);
      }
    }
  }
}
}  // namespace

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
// The payload of that all-gather, as plain functions (no context, no GPU): what a rank contributes and how the gathered bytes turn back
// into the Hamiltonian.  msevb_comm_allgather_evb uses them; tests/test_sharding_gloo.py drives them over a gloo all-gather on the CPU.
// Layout per rank, P = 2 S + 1 + 8 doubles with S = ceil((D - 1) / R): {H_dd, H_pd} of its diabats d = 1 + rank + R s (0-based d, slot s),
// the principal H'_11 as the rank has it, its eight principal energy slots.  hamiltonian: D x D row-major, upper triangle.
int msevb_evb_payload_size(int n_diabat, int n_ranks) { return n_diabat <= 1 ? 9 : 2 * ((n_diabat - 1 + n_ranks - 1) / n_ranks) + 9; }
int msevb_evb_payload_pack(int n_diabat, int n_ranks, int rank, const double* hamiltonian, const int* parent, const double* energies8, double* out) {
  if (n_diabat < 1 || n_ranks < 1 || rank < 0 || rank >= n_ranks || !hamiltonian || !parent || !out) return MSEVB_ERR_ARG;
  const int D = n_diabat, R = n_ranks, P = msevb_evb_payload_size(D, R), S = (P - 9) / 2;
  for (int k = 0; k < P; ++k) out[k] = 0.0;
  for (int d = 1; d < D; ++d) if ((d - 1) % R == rank) { const int s = (d - 1) / R; out[2 * s] = hamiltonian[(size_t)d * D + d]; out[2 * s + 1] = hamiltonian[(size_t)parent[d] * D + d]; }
  out[2 * S] = hamiltonian[0];
  if (energies8) for (int k = 0; k < 8; ++k) out[2 * S + 1 + k] = energies8[k];
  return MSEVB_OK;
}
// all: the R payloads in rank order.  Fills the entries of every diabat, H'_11 from rank 0, and (energies8 != NULL) the rank-ordered sums
// of the energy slots - the same bytes, the same order of additions on every rank: identical eigenvectors and topology decisions everywhere.
int msevb_evb_payload_unpack(int n_diabat, int n_ranks, const double* all, const int* parent, double* hamiltonian, double* energies8) {
  if (n_diabat < 1 || n_ranks < 1 || !all || !parent || !hamiltonian) return MSEVB_ERR_ARG;
  const int D = n_diabat, R = n_ranks, P = msevb_evb_payload_size(D, R), S = (P - 9) / 2;
  for (int d = 1; d < D; ++d) { const int r = (d - 1) % R, s = (d - 1) / R; hamiltonian[(size_t)d * D + d] = all[(size_t)r * P + 2 * s]; hamiltonian[(size_t)parent[d] * D + d] = all[(size_t)r * P + 2 * s + 1]; }
  hamiltonian[0] = all[2 * S];
  if (energies8) for (int k = 0; k < 8; ++k) { double e = 0.0; for (int r = 0; r < R; ++r) e += all[(size_t)r * P + 2 * S + 1 + k]; energies8[k] = e; }
  return MSEVB_OK;
}

int msevb_comm_allgather_evb(msevb_ctx* c, int with_energies) {
  const int D = c->D, R = c->n_ranks, P = msevb_evb_payload_size(D, R);
  ncclComm_t comm = (ncclComm_t)c->nccl_comm;
  DBuf<double>& d_h = c->d_comm_h;
  if (d_h.alloc((size_t)R * P)) FAIL(MSEVB_ERR_HIP, "hipMalloc failed (comm)");
  const size_t need = (size_t)(R + 1) * P;      // pinned: [R][P] gathered payloads, then this rank's own
  if (need > c->h_comm_cap) {
    HIPCHK(hipStreamSynchronize(c->aux2));
    if (c->h_comm) HIPCHK(hipHostFree(c->h_comm));
    c->h_comm_cap = 2 * need + 64;
    HIPCHK(hipHostMalloc((void**)&c->h_comm, c->h_comm_cap * sizeof(double), hipHostMallocDefault));
  }
  if (!c->h_comm_flag) { HIPCHK(hipHostMalloc((void**)&c->h_comm_flag, 64, hipHostMallocDefault)); *c->h_comm_flag = 0; }
  std::vector<double> Hm((size_t)D * D, 0.0);
  for (int i = 0; i < D; ++i) for (int j = 0; j < D; ++j) Hm[(size_t)i * D + j] = c->H[i][j];
  double* all = c->h_comm;
  double* mine = c->h_comm + (size_t)R * P;
  if (msevb_evb_payload_pack(D, R, c->rank, Hm.data(), c->parent, with_energies ? c->h_energies + S_ELEC_PAIR : nullptr, mine) != MSEVB_OK) FAIL(MSEVB_ERR_ARG, "EVB payload");      // S_ELEC_PAIR .. S_RECIP are contiguous
  if (c->comm_stall_ms > 0) { hipLaunchKernelGGL(k_comm_stall, dim3(1), dim3(1), 0, c->aux2, 100000LL * c->comm_stall_ms); c->comm_stall_ms = 0; }      // (one-shot test hook)
  hipLaunchKernelGGL(k_comm_stage, dim3(1), dim3(128), 0, c->aux2, mine, d_h.p + (size_t)c->rank * P, P);
  if (ncclAllGather(d_h.p + (size_t)c->rank * P, d_h.p, (size_t)P, ncclDouble, comm, c->aux2) != ncclSuccess) FAIL(MSEVB_ERR_COMM, "ncclAllGather failed");
  hipLaunchKernelGGL(k_comm_publish, dim3(1), dim3(256), 0, c->aux2, d_h.p, all, R * P, (volatile long long*)c->h_comm_flag, ++c->comm_seq);
  HIPCHK(hipGetLastError());
  CHK(comm_wait(c, c->comm_seq));      // aux2 already holds the energy downloads: the main stream keeps running
  if (msevb_evb_payload_unpack(D, R, all, c->parent, Hm.data(), with_energies ? c->h_energies + S_ELEC_PAIR : nullptr) != MSEVB_OK) FAIL(MSEVB_ERR_ARG, "EVB payload");
  for (int d = 1; d < D; ++d) { c->H[d][d] = Hm[(size_t)d * D + d]; c->H[c->parent[d]][d] = Hm[(size_t)c->parent[d] * D + d]; }
  c->H[0][0] = Hm[0];
  return MSEVB_OK;
}

// Device-to-device variants for the path on which the Hamiltonian is solved on the device (evb.hip, dev_solve_mr): nothing is staged through the host and the
// host does not wait - the collectives are enqueued with the step.  Order of the call's collectives on every rank: all-gather (grid stream, behind the energy-fetch
// kernel that wrote this rank's slot), all-reduce of the energy slots (descriptor stream, behind their reductions), all-reduce of the forces (main stream, behind the blend).
}  // extern "C"
namespace {
__global__ void k_comm_e8_pack(const double* __restrict__ slots, double* __restrict__ out, int zero_recip) {      // S_ELEC_PAIR .. S_RECIP are contiguous; the replicated reciprocal term counts on rank 0
  const int k = threadIdx.x;
  if (k < 8) out[k] = (zero_recip && k == S_RECIP - S_ELEC_PAIR) ? 0.0 : slots[S_ELEC_PAIR + k];
}
}  // namespace
// the host's wait for a number a kernel BEHIND a collective writes to pinned memory (eigenvector behind the all-gather, energies behind their all-reduce): like comm_wait it
// ends - with the number, with an error of the stream, or after comm_timeout_ms, because a peer that died or never entered the collective leaves the stream busy for ever
int msevb_comm_wait_flag(msevb_ctx* c, volatile long long* flag, long long seq, hipStream_t st, const char* what) {
  const auto t0 = std::chrono::steady_clock::now();
  for (long spins = 1;; ++spins) {
    if (*flag == seq) return MSEVB_OK;
    if ((spins & 0x3fff) == 0) {
      const hipError_t e = hipStreamQuery(st);
      if (e == hipSuccess) { if (*flag == seq) return MSEVB_OK; FAIL(MSEVB_ERR_COMM, std::string(what) + ": the stream finished without publishing its result"); }
      if (e != hipErrorNotReady) FAIL(MSEVB_ERR_HIP, std::string("HIP error behind ") + what + ": " + hipGetErrorString(e));
      if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(c->comm_timeout_ms)) {
        char b[200]; snprintf(b, sizeof b, "%s: no result after %d ms (a peer rank is missing or has stopped); rank %d of %d", what, c->comm_timeout_ms, c->rank, c->n_ranks);
        FAIL(MSEVB_ERR_COMM, b);
      }
    }
  }
}
int msevb_comm_allgather_device(msevb_ctx* c, int P, hipStream_t st) {
  if (!c || !c->nccl_comm) return MSEVB_OK;
  if (c->comm_stall_ms > 0) { hipLaunchKernelGGL(k_comm_stall, dim3(1), dim3(1), 0, st, 100000LL * c->comm_stall_ms); c->comm_stall_ms = 0; }      // (one-shot test hook: what a missing peer looks like)
  if (ncclAllGather(c->d_comm_h.p + (size_t)c->rank * P, c->d_comm_h.p, (size_t)P, ncclDouble, (ncclComm_t)c->nccl_comm, st) != ncclSuccess) FAIL(MSEVB_ERR_COMM, "ncclAllGather failed");
  return MSEVB_OK;
}
int msevb_comm_allreduce_e8_device(msevb_ctx* c, hipStream_t st) {
  if (!c || !c->nccl_comm) return MSEVB_OK;
  if (c->d_comm_e.alloc(8)) FAIL(MSEVB_ERR_HIP, "hipMalloc failed (comm)");
  hipLaunchKernelGGL(k_comm_e8_pack, dim3(1), dim3(64), 0, st, (const double*)c->d_energies.p, c->d_comm_e.p, c->rank != 0 ? 1 : 0);
  if (ncclAllReduce(c->d_comm_e.p, c->d_comm_e.p, 8, ncclDouble, ncclSum, (ncclComm_t)c->nccl_comm, st) != ncclSuccess) FAIL(MSEVB_ERR_COMM, "ncclAllReduce failed");
  return MSEVB_OK;
}
extern "C" {

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
  if (c->h_comm) { (void)hipHostFree(c->h_comm); c->h_comm = nullptr; c->h_comm_cap = 0; }
  if (c->h_comm_flag) { (void)hipHostFree(c->h_comm_flag); c->h_comm_flag = nullptr; }
  if (c->h_comm_e8) { (void)hipHostFree(c->h_comm_e8); c->h_comm_e8 = nullptr; }
}

}  // extern "C"
