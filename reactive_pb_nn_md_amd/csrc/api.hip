// C-ABI entry points of libmsevb_hip.so (include/msevb_hip.h): static-data setters, msevb_finalize, the force-call
// orchestration of calculate_total_force_energy (src/total_energy_forces.f90:19-99) and read-back helpers.
#include <algorithm>
#include <cstdlib>

#include <execinfo.h>
#include <signal.h>
#include <unistd.h>

#include <atomic>
#include "ctx.hpp"

thread_local StepRec* t_step_rec = nullptr;      // capture / replay of a force call (ctx.hpp)
long g_alloc_epoch = 0;
ApiProf g_api_prof;
std::atomic<int> g_live_contexts{0};      // contexts alive in this process (the linear step graphs need their four streams on hardware queues of their own: one context at a time)
// Room for every stream's own hardware queue (the runtime's default is four per process, shared round-robin beyond that): the library's four streams order themselves by
// kernels that wait on the device for kernels of the other three - two of them on one queue could wait for each other.  Read by the runtime when it initialises, i.e. at the
// first HIP call after this library was loaded; a value the user has set is left alone.
__attribute__((constructor)) static void msevb_queue_room() { setenv("GPU_MAX_HW_QUEUES", "8", 0); }
thread_local LinCap* t_lin = nullptr;      // learn / capture pass of a call as four linear graphs (ctx.hpp)


struct SlotCounts { int n[N_SLOTS]; };
__global__ __launch_bounds__(1024) void k_reduce_slots(SlotCounts cnt, const double* __restrict__ part, double* __restrict__ out) {
  const int slot = blockIdx.x, n = cnt.n[slot];
  if (n <= 0) return;
  __shared__ double red[1024];
  const double* p = part + (size_t)slot * MAX_BLOCKS;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;      // independent chains: the loads of a trip are all in flight together
  int k = threadIdx.x;
  for (; k + 3072 < n; k += 4096) { s0 += p[k]; s1 += p[k + 1024]; s2 += p[k + 2048]; s3 += p[k + 3072]; }
  for (; k < n; k += 1024) s0 += p[k];
  red[threadIdx.x] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  for (int st = 512; st > 0; st >>= 1) { if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st]; __syncthreads(); }
  if (threadIdx.x == 0) out[slot] = red[0];
}

// partial sums are reduced lazily: one launch for all pending slots
int reduce_partials(msevb_ctx* c, int slot, int nblocks) { c->slot_count[slot] = nblocks; return MSEVB_OK; }
int flush_reductions(msevb_ctx* c) {
  SlotCounts sc; bool any = false;
  for (int s = 0; s < N_SLOTS; ++s) { sc.n[s] = c->slot_count[s]; any |= sc.n[s] > 0; c->slot_count[s] = 0; }
  if (!any) return MSEVB_OK;
  hipLaunchKernelGGL(k_reduce_slots, dim3(N_SLOTS), dim3(1024), 0, c->stream, sc, c->d_partials.p, c->d_energies.p);
  HIPCHK(hipGetLastError());
  return MSEVB_OK;
}

// reduces ONE slot now, on the given stream (its partial sums must be complete there)
int flush_one(msevb_ctx* c, int slot, hipStream_t st) {
  if (c->slot_count[slot] <= 0) return MSEVB_OK;
  SlotCounts sc;
  for (int s = 0; s < N_SLOTS; ++s) sc.n[s] = 0;
  sc.n[slot] = c->slot_count[slot];
  c->slot_count[slot] = 0;
  hipLaunchKernelGGL(k_reduce_slots, dim3(N_SLOTS), dim3(1024), 0, st, sc, c->d_partials.p, c->d_energies.p);
  HIPCHK(hipGetLastError());
  return MSEVB_OK;
}

// ... and the pending slots s0..s1 in one launch on the given stream
int flush_range(msevb_ctx* c, int s0, int s1, hipStream_t st) {
  SlotCounts sc; bool any = false;
  for (int s = 0; s < N_SLOTS; ++s) sc.n[s] = 0;
  for (int s = s0; s <= s1; ++s) { sc.n[s] = c->slot_count[s]; any |= sc.n[s] > 0; c->slot_count[s] = 0; }
  if (!any) return MSEVB_OK;
  hipLaunchKernelGGL(k_reduce_slots, dim3(N_SLOTS), dim3(1024), 0, st, sc, c->d_partials.p, c->d_energies.p);
  HIPCHK(hipGetLastError());
  return MSEVB_OK;
}

int fetch_energies(msevb_ctx* c) {
  CHK(flush_reductions(c));
  HIPCHK(hipMemcpyAsync(c->h_energies, c->d_energies.p, N_SLOTS * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return MSEVB_OK;
}

// caller order <-> library order of a (3,N) array (msevb_set_host_order): scatter = 1: lib[perm[i]] = host[i]; 0: host[i] = lib[perm[i]]
__global__ void k_permute3(int N, const int* __restrict__ perm, const double* __restrict__ src, double* __restrict__ dst, int scatter) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int j = perm[i];
  const int s = scatter ? i : j, d = scatter ? j : i;
  dst[3 * (size_t)d] = src[3 * (size_t)s]; dst[3 * (size_t)d + 1] = src[3 * (size_t)s + 1]; dst[3 * (size_t)d + 2] = src[3 * (size_t)s + 2];
}

extern "C" {

// MSEVB_SEGV_TRACE=1 (diagnostic): a fault inside the runtime (no debugger on the GPU boxes) prints the native call stack before the process dies
static void segv_trace(int sig) {
  void* frames[64];
  const int n = backtrace(frames, 48);
  const char msg[] = "[msevb] fatal signal, native stack:\n";
  (void)!write(2, msg, sizeof msg - 1);
  backtrace_symbols_fd(frames, n, 2);
  signal(sig, SIG_DFL);
  raise(sig);
}
int msevb_create(msevb_ctx** out, int device_id) {
  { static const bool tr = getenv("MSEVB_SEGV_TRACE") != nullptr; static bool installed = false; if (tr && !installed) {
      installed = true;
      static char alt[1 << 16];      // (its own stack: a stack overflow inside the runtime must still get its trace out)
      stack_t ss; ss.ss_sp = alt; ss.ss_size = sizeof alt; ss.ss_flags = 0; (void)sigaltstack(&ss, nullptr);
      struct sigaction sa; memset(&sa, 0, sizeof sa); sa.sa_handler = segv_trace; sa.sa_flags = SA_ONSTACK; sigemptyset(&sa.sa_mask);
      (void)sigaction(SIGSEGV, &sa, nullptr); (void)sigaction(SIGABRT, &sa, nullptr); (void)sigaction(SIGBUS, &sa, nullptr);
    } }
  if (!out) return MSEVB_ERR_ARG;
  msevb_ctx* c = new msevb_ctx();
  *out = c;
  g_live_contexts++;
  c->device = device_id;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { c->err = "no HIP device visible: libmsevb_hip has no CPU fallback"; return MSEVB_ERR_HIP; }
  HIPCHK(hipSetDevice(device_id));
  HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));      // equal priorities: a favoured main stream starves the side chains
  c->own_stream = true;
  {
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    // the side chains feed the host's turn (Hamiltonian, blend coefficients), the pair loop only has to be done when the blend starts:
    // side streams above the main one (+1 % measured; the opposite order lost 10 %)
    const char* sp = getenv("MSEVB_SIDE_PRIORITY");      // (diagnostic) "normal" / "low": side streams at the main stream's priority / below it
    const int pr = !sp ? hi : (sp[0] == 'l' ? lo : 0);
    HIPCHK(hipStreamCreateWithPriority(&c->aux, hipStreamNonBlocking, pr));
    HIPCHK(hipStreamCreateWithPriority(&c->aux2, hipStreamNonBlocking, pr));
    HIPCHK(hipStreamCreateWithPriority(&c->aux3, hipStreamNonBlocking, pr));
  }
  HIPCHK(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&c->ev_fork0, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&c->ev_join2, hipEventDisableTiming));
  HIPCHK(hipHostMalloc((void**)&c->h_energies, N_SLOTS * sizeof(double), hipHostMallocDefault));
  memset(c->h_energies, 0, N_SLOTS * sizeof(double));
  const char* names[PH_COUNT] = {"verlet list construction", "sorted atom refresh", "real space (pair loop)", "intra-molecular (non-bonded + bonded)",
                                 "Q_grid (spread)", "Fourier transforms + B*C*FQ", "Q grid derivatives (gather)", "ms-evb diabat real space",
                                 "ms-evb reciprocal space", "ms-evb Hellmann-Feynman blend", "md integration"};
  for (int p = 0; p < PH_COUNT; ++p) c->ph[p].name = names[p];
  c->timing = false;
  return MSEVB_OK;
}

int msevb_destroy(msevb_ctx* c) {
  if (!c) return MSEVB_OK;
  g_live_contexts--;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  if (c->aux2) (void)hipStreamSynchronize(c->aux2);
  stamps_report(c);
  for (auto& e : c->pinned_user) if (e.registered) { (void)hipHostUnregister(const_cast<void*>(e.p)); e.registered = false; }
  evb_release(c);
  msevb_comm_release(c);
  if (c->pme_graph_exec) { (void)hipGraphExecDestroy((hipGraphExec_t)c->pme_graph_exec); c->pme_graph_exec = nullptr; }
  if (c->aux) { (void)hipStreamSynchronize(c->aux); (void)hipStreamDestroy(c->aux); c->aux = nullptr; }
  if (c->aux2) { (void)hipStreamSynchronize(c->aux2); (void)hipStreamDestroy(c->aux2); c->aux2 = nullptr; }
  if (c->aux3) { (void)hipStreamSynchronize(c->aux3); (void)hipStreamDestroy(c->aux3); c->aux3 = nullptr; }
  if (c->ev_join2) (void)hipEventDestroy(c->ev_join2);
  if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
  if (c->ev_fork0) (void)hipEventDestroy(c->ev_fork0);
  if (c->ev_join) (void)hipEventDestroy(c->ev_join);
  if (c->ev_refresh) (void)hipEventDestroy(c->ev_refresh);
  if (c->ev_top2) (void)hipEventDestroy(c->ev_top2);
  if (c->ev_weights) (void)hipEventDestroy(c->ev_weights);
  pme_teardown(c);
  DBuf<int>* ib[] = {&c->d_atype, &c->d_mol_type, &c->d_mol_natom, &c->d_mol_atoms, &c->d_mol_of, &c->d_cell_of, &c->d_perm, &c->d_cell_keys_tmp,
                     &c->d_perm_tmp, &c->d_cell_start, &c->d_inv_perm, &c->d_tile_max, &c->d_nbr, &c->d_nbr2, &c->d_prune_mode, &c->d_frozen, &c->d_type_sorted, &c->d_mol_sorted, &c->d_base, &c->d_splindex, &c->d_base_s};
  for (auto b : ib) b->release();
  DBuf<double>* db[] = {&c->d_chg, &c->d_mass, &c->d_B6, &c->d_B5, &c->d_tt, &c->d_dtt, &c->d_CB, &c->d_xyz, &c->d_vel, &c->d_force, &c->d_force_recip,
                        &c->d_xyz_store, &c->d_disp, &c->d_maxdisp, &c->d_Q, &c->d_theta, &c->d_green, &c->d_w, &c->d_dw, &c->d_uscale, &c->d_uscale_s, &c->d_partials, &c->d_energies,
                        &c->d_Fdiag, &c->d_Foff};
  for (auto b : db) b->release();
  c->d_pair_par.release(); c->d_moltypes.release(); c->d_ewald_tab.release(); c->d_ewald_tay.release(); c->d_nbr_count.release(); c->d_nbr_split.release(); c->d_host_perm.release(); c->d_stage3.release(); c->d_nbr2_count.release(); c->d_posq_prune.release(); c->d_flat.release(); c->d_cell_trips.release(); c->d_nbr_start.release();
  c->d_w_s.release(); c->d_rows_s.release(); c->d_brick_flag.release(); c->d_comm_h.release(); c->d_comm_e.release(); c->d_bad.release(); c->d_fsorted.release();
  if (c->h_ke) (void)hipHostFree(c->h_ke);
  c->d_sort_tmp.release(); c->d_sort_hist.release(); c->d_posq.release(); c->d_grid.release(); c->d_fft_work.release(); c->d_fft_work_b.release();
  for (int p = 0; p < PH_COUNT; ++p) for (hipEvent_t e : c->ph[p].ev) (void)hipEventDestroy(e);
  if (c->h_energies) (void)hipHostFree(c->h_energies);
  if (c->h_nbr_stats) (void)hipHostFree(c->h_nbr_stats);
  if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return MSEVB_OK;
}

const char* msevb_last_error(const msevb_ctx* c) { return c ? c->err.c_str() : "null context"; }

int msevb_set_stream(msevb_ctx* c, void* s) {
  if (!c) return MSEVB_ERR_ARG;
  if (c->own_stream && c->stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
  if (s) { c->stream = (hipStream_t)s; c->own_stream = false; }
  else { HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)); c->own_stream = true; }
  if (c->fft_info) rocfft_execution_info_set_stream(c->fft_info, c->stream);
  return MSEVB_OK;
}
int msevb_synchronize(msevb_ctx* c) {      // every stream the library enqueues on (the side streams normally end in an event the main stream waits for)
  HIPCHK(hipStreamSynchronize(c->stream));
  for (hipStream_t st : {c->aux, c->aux2, c->aux3}) if (st) HIPCHK(hipStreamSynchronize(st));
  return MSEVB_OK;
}

int msevb_set_system(msevb_ctx* c, int N, int M, const double* box, double rc, double rv, double alpha, int K, int order, int spline_grid, int erfc_grid) {
  if (!c || N <= 0 || M <= 0 || !box || K <= 0) { if (c) c->err = "msevb_set_system: bad argument"; return MSEVB_ERR_ARG; }
  // box(3,3) column-major; only orthorhombic boxes are supported, as in the reference (src/main_ms_evb.f90:62-68)
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) if (i != j && fabs(box[i + 3 * j]) > 1e-5) FAIL(MSEVB_ERR_ARG, "code assumes an orthorhombic box");
  if (!(rc < rv)) FAIL(MSEVB_ERR_ARG, "verlet_cutoff must be set greater than real_space_cutoff");
  for (int d = 0; d < 3; ++d) {
    c->box[d] = box[d + 3 * d];
    if (rv >= 0.5 * c->box[d]) FAIL(MSEVB_ERR_ARG, "verlet skin width is too big for this box");
  }
  c->N = N; c->M = M; c->rc = rc; c->rv = rv; c->alpha = alpha; c->K = K; c->order = order; c->spline_grid = spline_grid; c->erfc_grid = erfc_grid;
  // the weight-free PME kernels pack floor(f G / 6) and floor(f G / 5) into 16-bit fields of a row seed (pme_row_seed, pme.hip): G / 5 must stay below 65 536
  // (the reference's tables have 100 000 rows, src/glob_v.f90); larger tables would silently read wrong rows
  if (spline_grid < 12 || spline_grid > 5 * 65535) FAIL(MSEVB_ERR_LIMIT, "spline_grid outside 12 ... 327 675 (row seeds of the PME kernels hold 16-bit table offsets)");
  if (erfc_grid < 2) FAIL(MSEVB_ERR_ARG, "erfc_grid must be at least 2");
  if (!c->prune_delta_user) c->prune_delta = std::min(0.75, 0.5 * (rv - rc));      // margin of the rolling pair list: half the Verlet skin, at most 0.75 A (0.45 / 0.6 / 0.75 / 0.9 A at r_v - r_c = 1.5: 0.411 / 0.402 / 0.399 / 0.399 ms per step; option prune_delta overrides)
  c->finalized = false;
  return MSEVB_OK;
}

int msevb_set_atoms(msevb_ctx* c, const int* atype, const double* chg, const double* mass, const int* first, const int* natom, const int* mtype) {
  if (!c || c->N <= 0) return MSEVB_ERR_STATE;
  const int N = c->N, M = c->M;
  c->h_atype.assign(N, 0); c->h_chg.assign(chg, chg + N); c->h_mass.assign(mass, mass + N);
  for (int i = 0; i < N; ++i) c->h_atype[i] = atype[i] - 1;
  c->h_mol_type.assign(M, 0); c->h_mol_natom.assign(natom, natom + M); c->h_mol_atoms.assign((size_t)M * MAX_MOL_ATOMS, -1); c->h_mol_of.assign(N, -1);
  for (int m = 0; m < M; ++m) {
    c->h_mol_type[m] = mtype[m] - 1;
    if (natom[m] > MAX_MOL_ATOMS - 1) FAIL(MSEVB_ERR_LIMIT, "molecule with more than 7 atoms (one slot is reserved for an accepted proton)");
    for (int k = 0; k < natom[m]; ++k) {
      const int a = first[m] - 1 + k;
      if (a < 0 || a >= N) FAIL(MSEVB_ERR_ARG, "molecule atom index out of range");
      c->h_mol_atoms[(size_t)m * MAX_MOL_ATOMS + k] = a; c->h_mol_of[a] = m;
    }
  }
  for (int i = 0; i < N; ++i) if (c->h_mol_of[i] < 0) FAIL(MSEVB_ERR_ARG, "atom not covered by any molecule");
  c->finalized = false;
  return MSEVB_OK;
}

int msevb_set_forcefield(msevb_ctx* c, int nt, int ld, const int* vt, const double* vp, const double* vp14, const int* bt, const double* bp, const double* chg) {
  if (!c || nt <= 0 || nt > MAX_TYPES || ld < nt) return MSEVB_ERR_ARG;
  c->nt = nt;
  c->h_vdw_type.assign(nt * nt, 0); c->h_vdw_par.assign(nt * nt * 6, 0); c->h_vdw_par14.assign(nt * nt * 6, 0); c->h_bond_type.assign(nt * nt, 0);
  c->h_bond_par.assign(nt * nt * 3, 0); c->h_atype_chg.assign(chg, chg + nt);
  for (int i = 0; i < nt; ++i) for (int j = 0; j < nt; ++j) {
    c->h_vdw_type[i * nt + j] = vt[i + ld * j]; c->h_bond_type[i * nt + j] = bt[i + ld * j];
    for (int k = 0; k < 6; ++k) { c->h_vdw_par[(i * nt + j) * 6 + k] = vp[i + ld * (j + ld * k)]; c->h_vdw_par14[(i * nt + j) * 6 + k] = vp14[i + ld * (j + ld * k)]; }
    for (int k = 0; k < 3; ++k) c->h_bond_par[(i * nt + j) * 3 + k] = bp[i + ld * (j + ld * k)];
  }
  c->finalized = false;
  return MSEVB_OK;
}

int msevb_set_angle_types(msevb_ctx* c, int n, const int* idx, const double* par) {
  c->h_ang_idx.assign(idx, idx + 4 * n); c->h_ang_par.assign(par, par + 2 * n);
  for (int a = 0; a < n; ++a) for (int k = 0; k < 3; ++k) c->h_ang_idx[4 * a + k] -= 1;
  c->finalized = false;
  return MSEVB_OK;
}
int msevb_set_dihedral_types(msevb_ctx* c, int n, const int* idx, const double* par) {
  c->h_dih_idx.assign(idx, idx + 5 * n); c->h_dih_par.assign(par, par + 6 * n);
  for (int a = 0; a < n; ++a) for (int k = 0; k < 4; ++k) c->h_dih_idx[5 * a + k] -= 1;
  c->finalized = false;
  return MSEVB_OK;
}

int msevb_set_molecule_type(msevb_ctx* c, int it, int na, const int* atype, int nb, const int* bonds, int nang, const int* angles, int nd, const int* dihs,
                            const int* excl, const int* rp, const int* rb) {
  if (!c || it < 1 || it > MAX_MOLTYPES || na < 1 || na > MAX_MOL_ATOMS) return MSEVB_ERR_ARG;
  if (nb > MAX_BONDED || nang > MAX_BONDED || nd > MAX_BONDED) FAIL(MSEVB_ERR_LIMIT, "too many bonded terms in one molecule type");
  MolTypeHost& m = c->h_mt[it - 1];
  m.natom = na;
  m.atype.assign(na, 0); for (int i = 0; i < na; ++i) m.atype[i] = atype[i] - 1;
  m.bonds.assign(bonds, bonds + 2 * nb); for (auto& v : m.bonds) v -= 1;
  m.angles.assign(angles, angles + 3 * nang); for (auto& v : m.angles) v -= 1;
  m.dihedrals.assign(dihs, dihs + 4 * nd); for (auto& v : m.dihedrals) v -= 1;
  m.excl.assign(na * na, 0);
  for (int i = 0; i < na; ++i) for (int j = 0; j < na; ++j) m.excl[i * na + j] = excl[i + na * j];   // pair_exclusions(i,j) column-major
  m.reactive_protons.assign(na, 0); m.reactive_basic.assign(na, 0);
  if (rp) m.reactive_protons.assign(rp, rp + na);
  if (rb) m.reactive_basic.assign(rb, rb + na);
  if (it > c->nmt) c->nmt = it;
  c->finalized = false;
  return MSEVB_OK;
}

int msevb_set_evb(msevb_ctx* c, int nmt, int ld_m, int ld_i, const int* acid, const int* base, const int* conj, const int* pidx, const int* hidx,
                  const int* conj_at, const double* eref, const double* exa, const double* exp_, const int* da_i, const double* da_p, const int* pa_i,
                  const double* pa_p, const int* cp_i, const double* cp_p, const int* cp_t, int hyd, int max_chain) {
  if (!c || nmt > MAX_MOLTYPES || max_chain > MAX_CHAIN || max_chain < 1) return MSEVB_ERR_ARG;
  if (nmt > c->nmt) c->nmt = nmt;
  for (int i = 0; i < nmt; ++i) {
    c->e_acid[i] = acid[i]; c->e_base[i] = base[i]; c->e_conj_mol[i] = conj[i] - 1; c->e_proton_type[i] = pidx[i] - 1; c->e_heavy_type[i] = hidx[i] - 1;
    c->e_ref[i] = eref[i];
    for (int j = 0; j < nmt; ++j) c->e_exch_proton[i][j] = exp_[i + ld_m * j];
  }
  for (int i = 0; i < c->nt; ++i) { c->e_conj_atype[i] = conj_at[i] - 1; c->e_exch_atomic[i] = exa[i]; }
  c->n_da = c->n_pa = c->n_cp = 0;
  for (int r = 0; r < ld_i && r < MAX_INTERACTION; ++r) { if (da_i[r] == 0) break;
    for (int k = 0; k < 3; ++k) c->da_types[r][k] = da_i[r + ld_i * k] - 1;
    for (int k = 0; k < 6; ++k) c->da_par[r][k] = da_p[r + ld_i * k];
    c->n_da++; }
  for (int r = 0; r < ld_i && r < MAX_INTERACTION; ++r) { if (pa_i[r] == 0) break;
    for (int k = 0; k < 2; ++k) c->pa_types[r][k] = pa_i[r + ld_i * k] - 1;
    for (int k = 0; k < 5; ++k) c->pa_par[r][k] = pa_p[r + ld_i * k];
    c->n_pa++; }
  for (int r = 0; r < ld_i && r < MAX_INTERACTION; ++r) { if (cp_i[r] == 0) break;
    for (int k = 0; k < 3; ++k) c->cp_types[r][k] = cp_i[r + ld_i * k] - 1;
    for (int k = 0; k < 10; ++k) c->cp_par[r][k] = cp_p[r + ld_i * k];
    c->cp_func[r] = cp_t[r];
    c->n_cp++; }
  c->hydronium = hyd - 1; c->max_chain = max_chain; c->evb_set = true;
  return MSEVB_OK;
}

static int resolve_moltypes(msevb_ctx* c, std::vector<MolTypeDev>& out) {
  const int nt = c->nt;
  out.assign(c->nmt, MolTypeDev());
  for (int it = 0; it < c->nmt; ++it) {
    const MolTypeHost& h = c->h_mt[it];
    MolTypeDev& d = out[it];
    memset(&d, 0, sizeof(d));
    if (h.natom == 0) continue;
    d.natom = h.natom; d.nb = (int)h.bonds.size() / 2; d.na = (int)h.angles.size() / 3; d.nd = (int)h.dihedrals.size() / 4;
    for (int i = 0; i < h.natom; ++i) d.atype[i] = h.atype[i];
    for (int b = 0; b < d.nb; ++b) {
      const int i = h.bonds[2 * b], j = h.bonds[2 * b + 1], ti = h.atype[i], tj = h.atype[j];
      d.bond[b][0] = i; d.bond[b][1] = j; d.bond_type[b] = c->h_bond_type[ti * nt + tj];
      if (d.bond_type[b] < 1 || d.bond_type[b] > 3) FAIL(MSEVB_ERR_ARG, "bond type isn't implemented! Please select 1 (harmonic), 2 (GROMOS-96) or 3 (Morse)");
      for (int k = 0; k < 3; ++k) d.bond_par[b][k] = c->h_bond_par[(ti * nt + tj) * 3 + k];
    }
    for (int a = 0; a < d.na; ++a) {
      int t[3];
      for (int k = 0; k < 3; ++k) { d.ang[a][k] = h.angles[3 * a + k]; t[k] = h.atype[d.ang[a][k]]; }
      int found = -1;
      for (size_t r = 0; r < c->h_ang_idx.size() / 4; ++r) if (c->h_ang_idx[4 * r] == t[0] && c->h_ang_idx[4 * r + 1] == t[1] && c->h_ang_idx[4 * r + 2] == t[2]) { found = (int)r; break; }
      if (found < 0) FAIL(MSEVB_ERR_ARG, "requested angle type potential not implemented (no angletype for an angle of a molecule type)");
      d.ang_type[a] = c->h_ang_idx[4 * found + 3]; d.ang_par[a][0] = c->h_ang_par[2 * found]; d.ang_par[a][1] = c->h_ang_par[2 * found + 1];
    }
    for (int q = 0; q < d.nd; ++q) {
      int t[4];
      for (int k = 0; k < 4; ++k) { d.dih[q][k] = h.dihedrals[4 * q + k]; t[k] = h.atype[d.dih[q][k]]; }
      int found = -1;
      for (size_t r = 0; r < c->h_dih_idx.size() / 5; ++r)
        if (c->h_dih_idx[5 * r] == t[0] && c->h_dih_idx[5 * r + 1] == t[1] && c->h_dih_idx[5 * r + 2] == t[2] && c->h_dih_idx[5 * r + 3] == t[3]) { found = (int)r; break; }
      if (found < 0) FAIL(MSEVB_ERR_ARG, "no dihedraltype for a dihedral of a molecule type");
      d.dih_type[q] = c->h_dih_idx[5 * found + 4];
      for (int k = 0; k < 6; ++k) d.dih_par[q][k] = c->h_dih_par[6 * found + k];
    }
    for (int i = 0; i < h.natom; ++i) for (int j = 0; j < h.natom; ++j) {
      d.excl[i][j] = h.excl[i * h.natom + j];
      const int ti = h.atype[i], tj = h.atype[j];
      PairParam& pp = d.nb_par[i][j];
      pp.type = c->h_vdw_type[ti * nt + tj];
      const std::vector<double>& src = (d.excl[i][j] == 2 && pp.type == 0) ? c->h_vdw_par14 : c->h_vdw_par;
      pp.nonzero = 0;
      for (int k = 0; k < 6; ++k) { pp.p[k] = src[(ti * nt + tj) * 6 + k]; if (k != 1 && pp.p[k] != 0.0) pp.nonzero = 1; }
      if (pp.type == 0) pp.nonzero = 1;
    }
  }
  return MSEVB_OK;
}

int upload_topology(msevb_ctx* c) {   // atom types / charges / molecule membership (changes with the principal diabat)
  c->topo_epoch++;
  const int N = c->N, M = c->M;
  HIPCHK(hipMemcpyAsync(c->d_atype.p, c->h_atype.data(), N * sizeof(int), hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(c->d_chg.p, c->h_chg.data(), N * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(c->d_mol_type.p, c->h_mol_type.data(), M * sizeof(int), hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(c->d_mol_natom.p, c->h_mol_natom.data(), M * sizeof(int), hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(c->d_mol_atoms.p, c->h_mol_atoms.data(), (size_t)M * MAX_MOL_ATOMS * sizeof(int), hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(c->d_mol_of.p, c->h_mol_of.data(), N * sizeof(int), hipMemcpyHostToDevice, c->stream));
  // frozen atoms follow the atom types (which follow the topology)
  c->n_free = (double)N; c->inv_mass_sum_free = 0.0; c->any_frozen = false;
  std::vector<int> fr;
  if (!c->freeze_type.empty()) {
    fr.assign(N, 0);
    int nfr = 0;
    for (int a = 0; a < N; ++a) { const int t = c->h_atype[a]; if (t >= 0 && t < (int)c->freeze_type.size() && c->freeze_type[t] == 1) { fr[a] = 1; ++nfr; } }
    c->any_frozen = nfr > 0;
    c->n_free = (double)(N - nfr);
  }
  for (int a = 0; a < N; ++a) if (fr.empty() || !fr[a]) c->inv_mass_sum_free += 1.0 / c->h_mass[a];
  if (c->any_frozen) {
    if (c->d_frozen.alloc(N)) FAIL(MSEVB_ERR_HIP, "hipMalloc failed (frozen flags)");
    HIPCHK(hipMemcpyAsync(c->d_frozen.p, fr.data(), N * sizeof(int), hipMemcpyHostToDevice, c->stream));
  }
  HIPCHK(hipStreamSynchronize(c->stream));
  return MSEVB_OK;
}

// atype_freeze(i_type) == 1 (src/md_integration.f90:473, :512, :89-169): atoms of such types are neither kicked nor moved, do not
// count in the centre-of-mass momentum and do not give any of it up
int msevb_set_frozen_types(msevb_ctx* c, const int* atype_freeze, int n_types) {
  if (!c || !c->finalized) { if (c) c->err = "context not finalized"; return MSEVB_ERR_STATE; }
  (void)hipSetDevice(c->device);
  if (!atype_freeze || n_types < 0) { c->err = "bad argument"; return MSEVB_ERR_ARG; }
  c->freeze_type.assign(atype_freeze, atype_freeze + n_types);
  return upload_topology(c);
}

int msevb_finalize(msevb_ctx* c) {
  if (!c || c->N <= 0 || c->nt <= 0 || c->h_atype.empty()) { if (c) c->err = "msevb_finalize: system, atoms and force field must be set first"; return MSEVB_ERR_STATE; }
  HIPCHK(hipSetDevice(c->device));
  const int N = c->N, M = c->M, nt = c->nt;
  for (int i = 0; i < N; ++i) if (c->h_atype[i] < 0 || c->h_atype[i] >= nt) FAIL(MSEVB_ERR_ARG, "atom_type_index out of range");
  for (int m = 0; m < M; ++m) {
    const int t = c->h_mol_type[m];
    if (t < 0 || t >= c->nmt || c->h_mt[t].natom == 0) FAIL(MSEVB_ERR_ARG, "couldn't find '[ moleculetype ]' data for a molecule type in use");
    if (c->h_mt[t].natom != c->h_mol_natom[m]) FAIL(MSEVB_ERR_ARG, "molecule size does not match its molecule type");
  }
  // B_max * rc must stay inside the Tang-Toennies table (src/initialize_routines.f90:553)
  for (int k = 0; k < nt * nt; ++k) if (c->h_vdw_type[k] == 1 && c->h_vdw_par[k * 6 + 1] * c->rc > TT_MAX) FAIL(MSEVB_ERR_ARG, "Table size for damping function is smaller than B_max*the real space cutoff");
  CHK(build_tables(c));
  if (c->d_atype.alloc(N) || c->d_chg.alloc(N) || c->d_mass.alloc(N) || c->d_mol_type.alloc(M) || c->d_mol_natom.alloc(M) ||
      c->d_mol_atoms.alloc((size_t)M * MAX_MOL_ATOMS) || c->d_mol_of.alloc(N) || c->d_xyz.alloc(3 * (size_t)N) || c->d_vel.alloc(3 * (size_t)N) ||
      c->d_force.alloc(3 * (size_t)N) || c->d_posq.alloc(N) || c->d_type_sorted.alloc(N) || c->d_mol_sorted.alloc(N) ||
      c->d_partials.alloc((size_t)N_SLOTS * MAX_BLOCKS) || c->d_energies.alloc(N_SLOTS) || c->d_pair_par.alloc(nt * nt) || c->d_moltypes.alloc(c->nmt) ||
      c->d_ewald_tab.alloc(c->erfc_grid + 2) || c->d_ewald_tay.alloc(c->erfc_grid + 2) || c->d_B6.alloc(c->spline_grid) || c->d_B5.alloc(c->spline_grid) || c->d_tt.alloc(4 * TT_GRID) ||
      c->d_dtt.alloc(4 * TT_GRID) || c->d_CB.alloc(c->t_CB.size()))
    FAIL(MSEVB_ERR_HIP, "hipMalloc failed");
  CHK(upload_topology(c));
  HIPCHK(hipMemcpy(c->d_mass.p, c->h_mass.data(), N * sizeof(double), hipMemcpyHostToDevice));
  std::vector<PairParam> pp(nt * nt);
  for (int k = 0; k < nt * nt; ++k) {
    pp[k].type = c->h_vdw_type[k];
    pp[k].nonzero = pp[k].type == 0 ? 1 : 0;
    for (int q = 0; q < 6; ++q) { pp[k].p[q] = c->h_vdw_par[k * 6 + q]; if (pp[k].type == 1 && q != 1 && pp[k].p[q] != 0.0) pp[k].nonzero = 1; }
  }
  c->has_sapt = false;
  for (auto& q : pp) if (q.type == 1 && q.nonzero) c->has_sapt = true;
  HIPCHK(hipMemcpy(c->d_pair_par.p, pp.data(), pp.size() * sizeof(PairParam), hipMemcpyHostToDevice));
  std::vector<MolTypeDev> mts;
  CHK(resolve_moltypes(c, mts));
  HIPCHK(hipMemcpy(c->d_moltypes.p, mts.data(), mts.size() * sizeof(MolTypeDev), hipMemcpyHostToDevice));
  std::vector<double2> et(c->erfc_grid + 2);      // one spare row behind r_c: ceil(r_c / dx) may round up by one (pair_flat.hip does not clamp the row from above)
  for (int i = 0; i <= c->erfc_grid; ++i) et[i] = make_double2(c->t_erfc[i], c->t_scale[i]);
  et[c->erfc_grid + 1] = et[c->erfc_grid];
  HIPCHK(hipMemcpy(c->d_ewald_tab.p, et.data(), et.size() * sizeof(double2), hipMemcpyHostToDevice));
  for (int i = 0; i <= c->erfc_grid; ++i) et[i].y = c->t_ewE[i];
  et[c->erfc_grid + 1] = et[c->erfc_grid];
  HIPCHK(hipMemcpy(c->d_ewald_tay.p, et.data(), et.size() * sizeof(double2), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(c->d_B6.p, c->t_B6.data(), c->t_B6.size() * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(c->d_B5.p, c->t_B5.data(), c->t_B5.size() * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(c->d_tt.p, c->t_tt.data(), c->t_tt.size() * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(c->d_dtt.p, c->t_dtt.data(), c->t_dtt.size() * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(c->d_CB.p, c->t_CB.data(), c->t_CB.size() * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemset(c->d_vel.p, 0, 3 * (size_t)N * 8));
  HIPCHK(hipMemset(c->d_force.p, 0, 3 * (size_t)N * 8));
  HIPCHK(hipMemset(c->d_energies.p, 0, N_SLOTS * 8));
  HIPCHK(hipDeviceSynchronize());      // (hipMemset may return early; the library's non-blocking streams do not wait for the null stream)
  CHK(pme_setup(c));
  c->list_valid = false;
  c->flag_verlet_list = 1;
  c->finalized = true;
  // MSEVB_OPTIONS="name=value,name=value": msevb_set_option calls for hosts that have no other way to make them (the unchanged Fortran main
  // program) and for bisecting a schedule; an unknown name is an error, as it is through the entry point
  if (const char* env = getenv("MSEVB_OPTIONS")) {
    std::string all(env);
    size_t p = 0;
    while (p < all.size()) {
      size_t q = all.find(',', p); if (q == std::string::npos) q = all.size();
      const std::string kv = all.substr(p, q - p); p = q + 1;
      const size_t eq = kv.find('=');
      if (kv.empty()) continue;
      if (eq == std::string::npos) FAIL(MSEVB_ERR_ARG, "MSEVB_OPTIONS: expected name=value");
      CHK(msevb_set_option(c, kv.substr(0, eq).c_str(), atof(kv.c_str() + eq + 1)));
    }
  }
  return MSEVB_OK;
}

int msevb_get_table(msevb_ctx* c, const char* name, double* out, int n) {
  const std::vector<double>* t = nullptr;
  const std::string s(name);
  if (s == "B6_spline") t = &c->t_B6; else if (s == "B5_spline") t = &c->t_B5; else if (s == "erfc_table") t = &c->t_erfc;
  else if (s == "ewaldscale_table") t = &c->t_scale; else if (s == "Tang_Toennies_table") t = &c->t_tt; else if (s == "dTang_Toennies_table") t = &c->t_dtt;
  else if (s == "CB") t = &c->t_CB;
  if (!t) return MSEVB_ERR_ARG;
  if (out) memcpy(out, t->data(), sizeof(double) * (size_t)std::min<size_t>(n, t->size()));
  return (int)t->size();
}
int msevb_get_ewald_self(msevb_ctx* c, double* e) { *e = c->ewald_self; return MSEVB_OK; }

#define NEED_FINAL() do { if (!c || !c->finalized) { if (c) c->err = "context not finalized (msevb_finalize)"; return MSEVB_ERR_STATE; } (void)hipSetDevice(c->device); } while (0)

void pin_user_buffer(msevb_ctx* c, const void* p, size_t bytes) {
  if (!c->pin_host_arrays || !p || bytes < (1u << 16)) return;
  int free_slot = -1, coldest = 0;
  for (int k = 0; k < 8; ++k) {
    auto& e = c->pinned_user[k];
    if (e.p == p && e.bytes == bytes) {
      if (!e.registered && ++e.hits >= 2) e.registered = hipHostRegister(const_cast<void*>(p), bytes, hipHostRegisterDefault) == hipSuccess;      // a failure just leaves it pageable
      if (!e.registered) (void)hipGetLastError();
      return;
    }
    if (!e.p && free_slot < 0) free_slot = k;
    if (c->pinned_user[k].hits < c->pinned_user[coldest].hits) coldest = k;
  }
  const int k = free_slot >= 0 ? free_slot : coldest;      // temporaries (a new address every call) only ever pass through here
  if (c->pinned_user[k].registered) (void)hipHostUnregister(const_cast<void*>(c->pinned_user[k].p));
  c->pinned_user[k] = {p, bytes, 1, false};
}

int msevb_set_host_order(msevb_ctx* c, const int* h2l) {
  NEED_FINAL();
  if (!h2l) { c->host_perm_on = false; return MSEVB_OK; }
  std::vector<int> p(c->N); std::vector<char> seen(c->N, 0);
  bool ident = true;
  for (int i = 0; i < c->N; ++i) {
    const int j = h2l[i] - 1;
    if (j < 0 || j >= c->N || seen[j]) FAIL(MSEVB_ERR_ARG, "msevb_set_host_order: not a permutation of 1..N");
    seen[j] = 1; p[i] = j; ident = ident && j == i;
  }
  if (ident) { c->host_perm_on = false; return MSEVB_OK; }
  if (c->d_host_perm.alloc(c->N) || c->d_stage3.alloc(3 * (size_t)c->N)) FAIL(MSEVB_ERR_HIP, "hipMalloc failed (host order)");
  HIPCHK(hipMemcpy(c->d_host_perm.p, p.data(), (size_t)c->N * sizeof(int), hipMemcpyHostToDevice));
  c->host_perm_on = true;
  return MSEVB_OK;
}
// (3,N) arrays across the ABI: straight copies, or through the caller's atom order (msevb_set_host_order); asynchronous on the main stream
int upload_atoms3(msevb_ctx* c, double* dst_dev, const double* src_host) {
  const size_t bytes = 3 * (size_t)c->N * 8;
  if (!c->host_perm_on) { HIPCHK(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, c->stream)); return MSEVB_OK; }
  HIPCHK(hipMemcpyAsync(c->d_stage3.p, src_host, bytes, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_permute3, dim3((c->N + 255) / 256), dim3(256), 0, c->stream, c->N, c->d_host_perm.p, (const double*)c->d_stage3.p, dst_dev, 1);
  HIPCHK(hipGetLastError());
  return MSEVB_OK;
}
int download_atoms3(msevb_ctx* c, double* dst_host, const double* src_dev) {
  const size_t bytes = 3 * (size_t)c->N * 8;
  if (!c->host_perm_on) { HIPCHK(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, c->stream)); return MSEVB_OK; }
  hipLaunchKernelGGL(k_permute3, dim3((c->N + 255) / 256), dim3(256), 0, c->stream, c->N, c->d_host_perm.p, src_dev, c->d_stage3.p, 0);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(dst_host, c->d_stage3.p, bytes, hipMemcpyDeviceToHost, c->stream));
  return MSEVB_OK;
}

int msevb_set_positions(msevb_ctx* c, const double* xyz) {
  NEED_FINAL();
  pin_user_buffer(c, xyz, 3 * (size_t)c->N * 8);
  CHK(upload_atoms3(c, c->d_xyz.p, xyz));
  HIPCHK(hipMemsetAsync(c->d_energies.p + S_FAULT, 0, 8, c->stream));      // new coordinates: the sticky bad-coordinate flag starts over
  HIPCHK(hipMemsetAsync(c->d_energies.p + S_TIMEOUT, 0, 8, c->stream));
  c->h_energies[S_FAULT] = 0.0; c->h_energies[S_TIMEOUT] = 0.0;
  HIPCHK(hipStreamSynchronize(c->stream));
  return MSEVB_OK;
}
// periodic_box_change (src/pme.f90:634-664): new box lengths -> CB array, reciprocal lattice, stale cells / list / captured PME graph
int msevb_set_box(msevb_ctx* c, const double* box3) {
  NEED_FINAL();
  if (!box3) { c->err = "bad argument"; return MSEVB_ERR_ARG; }
  double L = box3[0];
  for (int d = 0; d < 3; ++d) { if (!(box3[d] > 0.0)) FAIL(MSEVB_ERR_ARG, "box lengths must be positive"); L = box3[d] < L ? box3[d] : L; }
  if (!(c->rc < c->rv && c->rv < 0.5 * L)) FAIL(MSEVB_ERR_ARG, "need real_space_cutoff < verlet_cutoff < L/2 (src/general_routines.f90:1115-1130)");
  HIPCHK(hipDeviceSynchronize());
  c->topo_epoch++;
  for (int d = 0; d < 3; ++d) c->box[d] = box3[d];
  CHK(build_box_tables(c));
  HIPCHK(hipMemcpy(c->d_CB.p, c->t_CB.data(), c->t_CB.size() * 8, hipMemcpyHostToDevice));
  CHK(pme_build_green(c));
  if (c->pme_graph_exec) { (void)hipGraphExecDestroy((hipGraphExec_t)c->pme_graph_exec); c->pme_graph_exec = nullptr; }      // box and lattice are baked into its kernel arguments
  memset(c->pme_graph_key, 0, sizeof c->pme_graph_key);
  c->list_valid = false; c->cells_valid = false; c->flag_verlet_list = 1;
  return MSEVB_OK;
}

int msevb_get_positions(msevb_ctx* c, double* xyz) { NEED_FINAL(); CHK(download_atoms3(c, xyz, c->d_xyz.p)); HIPCHK(hipStreamSynchronize(c->stream)); return MSEVB_OK; }
int msevb_set_velocities(msevb_ctx* c, const double* v) { NEED_FINAL(); CHK(upload_atoms3(c, c->d_vel.p, v)); HIPCHK(hipStreamSynchronize(c->stream)); return MSEVB_OK; }
int msevb_get_velocities(msevb_ctx* c, double* v) { NEED_FINAL(); CHK(download_atoms3(c, v, c->d_vel.p)); HIPCHK(hipStreamSynchronize(c->stream)); return MSEVB_OK; }
int msevb_get_forces(msevb_ctx* c, double* f) { NEED_FINAL(); CHK(download_atoms3(c, f, c->d_force.p)); HIPCHK(hipStreamSynchronize(c->stream)); return MSEVB_OK; }
int msevb_set_forces(msevb_ctx* c, const double* f) { NEED_FINAL(); if (!f) return MSEVB_ERR_ARG; CHK(upload_atoms3(c, c->d_force.p, f)); HIPCHK(hipStreamSynchronize(c->stream)); return MSEVB_OK; }
void* msevb_device_forces(msevb_ctx* c) { return c ? c->d_force.p : nullptr; }
void* msevb_device_positions(msevb_ctx* c) { return c ? c->d_xyz.p : nullptr; }

int msevb_construct_verlet_list(msevb_ctx* c) { NEED_FINAL(); CHK(nb_build(c)); c->flag_verlet_list = 0; return MSEVB_OK; }

int msevb_update_verlet_displacements(msevb_ctx* c, int* flag) {
  NEED_FINAL();
  CHK(nb_check_displacements(c, 0, nullptr));
  CHK(fetch_energies(c));
  const double skin = 1.2 * (c->rv - c->rc);   // verlet_thresh = 1.2d0 (src/glob_v.f90:394, src/general_routines.f90:1320)
  c->flag_verlet_list = (c->h_energies[S_MAXD1] + c->h_energies[S_MAXD2]) > skin ? 1 : 0;
  if (flag) *flag = c->flag_verlet_list;
  return MSEVB_OK;
}

int msevb_get_verlet_flag(msevb_ctx* c, int* flag) { if (!c || !flag) return MSEVB_ERR_ARG; *flag = c->flag_verlet_list; return MSEVB_OK; }

int msevb_get_verlet_list(msevb_ctx* c, int* vp, int* nl, long* nn) {
  NEED_FINAL();
  CHK(nb_settle(c));
  if (!c->list_valid) FAIL(MSEVB_ERR_STATE, "no Verlet list");
  const int N = c->N;
  std::vector<unsigned> st(N), cnt(N);
  std::vector<int> nb((size_t)N * c->nbr_cap), perm(N);
  HIPCHK(hipMemcpy(st.data(), c->d_nbr_start.p, N * sizeof(unsigned), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(cnt.data(), c->d_nbr_count.p, N * sizeof(unsigned), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(nb.data(), c->d_nbr.p, nb.size() * sizeof(int), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(perm.data(), c->d_perm.p, N * sizeof(int), hipMemcpyDeviceToHost));
  std::vector<std::vector<int>> half(N);
  std::vector<int> cstart, keys;
  if (c->pair_tiles) {     // half list with tile-local indices: decode with the cell structure
    const int ncell = c->ncell[0] * c->ncell[1] * c->ncell[2];
    cstart.resize(ncell + 1); keys.resize(N);
    HIPCHK(hipMemcpy(cstart.data(), c->d_cell_start.p, (ncell + 1) * sizeof(int), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(keys.data(), c->d_cell_of.p, N * sizeof(int), hipMemcpyDeviceToHost));
  }
  for (int k = 0; k < N; ++k) {
    const int i = perm[k];
    int slots[HALF_SLOTS], off[HALF_SLOTS + 1];
    if (c->pair_tiles) {
      cell_half_neighbourhood(keys[k], c->ncell[0], c->ncell[1], c->ncell[2], slots);
      off[0] = 0;
      for (int s = 0; s < HALF_SLOTS; ++s) off[s + 1] = off[s] + cstart[slots[s] + 1] - cstart[slots[s]];
    }
    if ((size_t)st[k] + cnt[k] > nb.size()) { char b[160]; snprintf(b, sizeof b, "msevb_get_verlet_list: row %d of the device list is out of range (start %u, count %u, stride %u)", k, st[k], cnt[k], c->nbr_cap); FAIL(MSEVB_ERR_STATE, b); }
    for (unsigned q = st[k]; q < st[k] + cnt[k]; ++q) {
      int js = (int)((unsigned)nb[q] & NBR_IDX_MASK);
      if (!c->pair_tiles && js >= N) { char b[160]; snprintf(b, sizeof b, "msevb_get_verlet_list: entry %u of row %d names atom %d of %d", q - st[k], k, js, N); FAIL(MSEVB_ERR_STATE, b); }
      if (c->pair_tiles) {      // every pair is stored once, with the atom that comes first in its cell's half neighbourhood
        int s = 0; while (s + 1 < HALF_SLOTS && js >= off[s + 1]) ++s;
        js = cstart[slots[s]] + (js - off[s]);
        const int j = perm[js];
        half[std::min(i, j)].push_back(std::max(i, j));
      } else {                  // full list: every pair twice
        const int j = perm[js];
        if (j > i) half[i].push_back(j);
      }
    }
  }
  long tot = 0;
  for (int i = 0; i < N; ++i) { std::sort(half[i].begin(), half[i].end()); if (vp) vp[i] = (int)tot + 1; if (nl) for (int j : half[i]) nl[tot++] = j + 1; else tot += (long)half[i].size(); }
  if (vp) vp[N] = (int)tot + 1;
  if (nn) *nn = tot;
  return MSEVB_OK;
}

int msevb_zero_forces(msevb_ctx* c) { NEED_FINAL(); HIPCHK(hipMemsetAsync(c->d_force.p, 0, 3 * (size_t)c->N * 8, c->stream)); return MSEVB_OK; }

int msevb_real_space_energy_force(msevb_ctx* c, double* e) {
  NEED_FINAL();
  CHK(nb_settle(c));
  if (!c->list_valid) CHK(nb_build(c)); else CHK(nb_refresh_sorted(c));
  CHK(pair_forces(c, S_ELEC_PAIR, S_VDW_PAIR));
  CHK(molecule_forces(c, S_ELEC_INTRA, true, false));
  CHK(fetch_energies(c));
  if (e) { e[0] = c->h_energies[S_ELEC_PAIR]; e[1] = c->h_energies[S_VDW_PAIR]; e[2] = c->h_energies[S_ELEC_INTRA]; e[3] = c->h_energies[S_VDW_INTRA]; }
  return MSEVB_OK;
}
int msevb_pme_reciprocal_space_energy_force(msevb_ctx* c, double* e) {
  NEED_FINAL();
  CHK(pme_forces(c, S_RECIP));
  CHK(fetch_energies(c));
  c->E_recip = c->h_energies[S_RECIP];
  if (e) *e = c->E_recip;
  return MSEVB_OK;
}
int msevb_intra_molecular_energy_force(msevb_ctx* c, double* e) {
  NEED_FINAL();
  CHK(molecule_forces(c, S_ELEC_INTRA, false, true));
  CHK(fetch_energies(c));
  if (e) { e[0] = c->h_energies[S_BOND]; e[1] = c->h_energies[S_ANGLE]; e[2] = c->h_energies[S_DIH]; }
  return MSEVB_OK;
}

__global__ void k_stamp(unsigned long long* out) { *out = wall_clock64(); }
// fixed-address set: stamp 0 also accumulates the time since the PREVIOUS call's blend end (slot 9) in slots 13 / 14 - the part of an MD step that lies between two force calls
__global__ void k_stamp0_fixed(unsigned long long* h) { const unsigned long long t = wall_clock64(); if (h[9] && t > h[9] && t - h[9] < 100000ull) { h[13] += t - h[9]; h[14] += 1; } h[0] = t; }
void stamp(msevb_ctx* c, int idx, hipStream_t st) {
  if (c->stamps_on < 0) { const char* e = getenv("MSEVB_STAMPS"); c->stamps_on = e ? (e[0] == '2' ? 2 : 1) : 0; }      // 2: ONE set of stamps at fixed addresses - the launches are the same every step, so they can sit inside the linear step graphs
  if (!c->stamps_on) return;
  if (!c->h_stamps) { if (hipHostMalloc((void**)&c->h_stamps, 4 * 16 * sizeof(unsigned long long), hipHostMallocDefault) != hipSuccess) { c->stamps_on = 0; return; } memset(c->h_stamps, 0, 512); }
  if (c->stamps_on == 2) {
    if (idx == 0 && !(t_step_rec && t_step_rec->mode == 3)) {      // (host side, once per call: the set the previous call left is complete - the host has waited for its energies since)
      unsigned long long* old = c->h_stamps;
      bool ok = old[0] && old[9] > old[2] && old[2] > old[1] && old[1] >= old[0] && old[1] - old[0] < 12000ull;
      for (int k = 1; k < 11 && ok; ++k) if (old[k] && (old[k] < old[0] || old[k] - old[0] > 100000ull)) ok = false;
      if (ok) { for (int k = 1; k < 16; ++k) if (old[k] >= old[0]) c->stamp_sum[k] += (double)(old[k] - old[0]) * 0.01; c->stamp_n++; }
    }
    if (idx == 0) hipLaunchKernelGGL(k_stamp0_fixed, dim3(1), dim3(1), 0, st, c->h_stamps); else hipLaunchKernelGGL(k_stamp, dim3(1), dim3(1), 0, st, c->h_stamps + idx);
    return;
  }
  if (idx == 0) {      // a ring of four stamp sets: the set written two steps ago is complete by now
    c->stamp_step++;
    unsigned long long* old = c->h_stamps + 16 * ((c->stamp_step + 2) & 3);
    if (c->stamp_step > 4 && old[0] && old[9] > old[0] && old[2] > old[1] && old[1] - old[0] < 12000ull) {      // steps that rebuild the list (pair start > 120 us) are left out
      for (int k = 1; k < 16; ++k) if (old[k] >= old[0]) c->stamp_sum[k] += (double)(old[k] - old[0]) * 0.01;
      c->stamp_n++;
    }
    memset(c->h_stamps + 16 * (c->stamp_step & 3), 0, 128);
  }
  hipLaunchKernelGGL(k_stamp, dim3(1), dim3(1), 0, st, c->h_stamps + 16 * (c->stamp_step & 3) + idx);
}
void stamps_report(msevb_ctx* c) {
  if (c->stamps_on < 1 || !c->stamp_n) return;
  const char* nm[16] = {"step start", "pair start", "pair end", "spec_prep start", "env chain end", "grid chain start", "grid chain end", "coupling chain end", "blend phase start", "blend end", "fetch A published", "", "", "", "", ""};
  fprintf(stderr, "[msevb gpu stamps, us after stamp 0, %ld steps]", c->stamp_n);
  for (int k = 1; k < 11; ++k) fprintf(stderr, " %s %.0f |", nm[k], c->stamp_sum[k] / c->stamp_n);
  if (c->stamps_on == 2 && c->h_stamps[14]) fprintf(stderr, " between the blend end of a call and stamp 0 of the next %.1f us (%llu pairs of calls)", (double)c->h_stamps[13] * 0.01 / (double)c->h_stamps[14], c->h_stamps[14]);
  fprintf(stderr, "\n");
}

// the principal-diabat force call, everything enqueued on the stream; energies land in c->h_energies after fetch_energies()
int total_force_enqueue(msevb_ctx* c) {
  c->timing_now = c->timing && (c->timing_tick++ % c->timing_stride) == 0;
  c->weights_side = false;      // (set again by pme_side_weights when the MS-EVB call has the stored weights made for THIS call's positions)
  // Verlet freshness: src/total_energy_forces.f90:30-39 (rebuild when the flag was raised by the previous call)
  CHK(nb_settle(c));
  stamp(c, 0, c->stream);
  const bool rebuild = c->flag_verlet_list == 1 || !c->list_valid;
  int self_nb = 0;
  const int fused_nb = c->refresh_fused_nb; c->refresh_fused_nb = 0;      // one-shot: a repeated force call (MSEVB_RETRY) refreshes the ordinary way (idempotent, and it zeroes the force array again)
  // A step that keeps its list: the PME front end (stencils, spreading, FFTs) needs the NEW POSITIONS only - not the refreshed sorted arrays,
  // not the zeroed force array - and is the longest chain beside the pair loop: it leaves for the side stream before the refresh kernels
  const bool early_pme = c->overlap && !rebuild && c->cells_valid;
  if (early_pme) {
    if (c->ev_pos_ready) HIPCHK(hipStreamWaitEvent(c->aux, c->ev_pos_ready, 0));      // (the MS-EVB call recorded it right behind the integrator's kernel)
    else { HIPCHK(hipEventRecord(c->ev_fork0, c->stream)); HIPCHK(hipStreamWaitEvent(c->aux, c->ev_fork0, 0)); }
    CHK(pme_front(c, S_RECIP, true));
  }
  if (rebuild) {      // cells + sorted arrays now; the rows are filled below, beside what only needs the cells
    c->rebuild_zero_force_now = true;      // (k_gather_sorted zeroes the force array on its way: no fill command)
    const int rc_b = nb_build_begin(c);
    const bool zeroed = c->rebuild_zero_force_now == false;
    c->rebuild_zero_force_now = false;
    CHK(rc_b); c->flag_verlet_list = 0;
    if (!zeroed) HIPCHK(hipMemsetAsync(c->d_force.p, 0, 3 * (size_t)c->N * 8, c->stream));
  } else if (c->refresh_side_now && c->overlap && c->ev_pos_ready) {      // beside the neighbourhood kernel, which holds the main stream right now
    if (!c->ev_refresh) HIPCHK(hipEventCreateWithFlags(&c->ev_refresh, hipEventDisableTiming));
    HIPCHK(hipStreamWaitEvent(c->aux3, c->ev_pos_ready, 0));
    CHK(nb_step_refresh(c, c->aux3));
    HIPCHK(hipEventRecord(c->ev_refresh, c->aux3));
    HIPCHK(hipStreamWaitEvent(c->stream, c->ev_refresh, 0));
  } else {
    // the pair kernel can take the "does my list still cover r_c" decision itself: k_top2_final then only serves the host's rebuild flag and runs beside the pair loop
    self_nb = (c->overlap && c->pair_self_mode && !c->prune_freeze && c->d_prune_mode.p && c->d_posq_prune.p && pair_flat_usable(c)) ? (fused_nb > 0 ? fused_nb : (c->N + 255) / 256) : 0;
    if (self_nb > 2048) self_nb = 0;      // (every pair workgroup reads all the partial maxima: 16 B x self_nb each - at configs[4] size 10 264 blocks x 7 900 workgroups; k_top2_final decides there)
    if (fused_nb > 0) { if (!self_nb) CHK(nb_step_top2(c, fused_nb)); }      // the integrator's first kernel did the refresh
    else CHK(nb_step_refresh(c, nullptr, self_nb == 0));
  }
  if (c->overlap) {     // independent chains, all adding to the force with fp64 atomics
    HIPCHK(aux_fork(c));      // (round 5: forking the side streams from the call's ev_pos_ready instead - one signal packet less between the integrator's kernel and the pair loop - is 1.6 % SLOWER:
                              //  the side kernels then reach the GPU before the pair loop's first workgroups; HISTORY.md F)
    if (self_nb) {
      if (!c->ev_top2) HIPCHK(hipEventCreateWithFlags(&c->ev_top2, hipEventDisableTiming));
      CHK(nb_step_top2(c, self_nb, c->aux3, false));
      HIPCHK(hipEventRecord(c->ev_top2, c->aux3));
      c->pair_self_nb = self_nb;
    }
    if (rebuild) CHK(nb_build_finish(c, true));                      // main: list rows first (the longest chain of a rebuild step: rows, then the pair loop)
    stamp(c, 1, c->stream);
    CHK(pair_forces(c, S_ELEC_PAIR, S_VDW_PAIR));                    // main: the pair loop (longest) goes out first
    c->pair_self_nb = 0;
    c->top2_join_pending = false;
    if (self_nb) { if (c->skip_side_joins_now) c->top2_join_pending = true;      /* (MS-EVB fast path: the stream that publishes the energies waits for it itself - nothing on the main stream between the pair loop and the blend phase) */
                   else HIPCHK(hipStreamWaitEvent(c->stream, c->ev_top2, 0)); }      // (the displacement maxima are part of what the main stream publishes)
    stamp(c, 2, c->stream);
    if (c->evb_pair_first || c->molecules_after_pair_now) { if (!c->ev_pair_done) HIPCHK(hipEventCreateWithFlags(&c->ev_pair_done, hipEventDisableTiming)); HIPCHK(hipEventRecord(c->ev_pair_done, c->stream)); }
    if (!c->defer_molecules_now)                                     // (the MS-EVB call sends them out itself, behind its coupling chain on the same stream)
      CHK(molecule_forces(c, S_ELEC_INTRA, true, true, c->aux3));    // aux3: per-molecule terms (few, long threads); aux2 stays free for the
                                                                     // EVB descriptors + environment chain, which the host enqueues next
    if (!early_pme) CHK(pme_front(c, S_RECIP, true));                // aux: stencils, spreading, FFTs, convolution
    CHK(pme_tail(c, true));                                          // main: wait for aux, gather (skipped by the EVB fast path)
    if (!c->defer_molecules_now) {
      HIPCHK(hipEventRecord(c->ev_join2, c->aux3));
      if (!c->skip_side_joins_now) HIPCHK(hipStreamWaitEvent(c->stream, c->ev_join2, 0));
    }
  } else {
    if (rebuild) CHK(nb_build_finish(c, true));
    CHK(pair_forces(c, S_ELEC_PAIR, S_VDW_PAIR));
    CHK(molecule_forces(c, S_ELEC_INTRA, true, true));
    CHK(pme_forces(c, S_RECIP, false));
  }
  return MSEVB_OK;
}
int total_force_collect(msevb_ctx* c, double* en) {
  const double* h = c->h_energies;
  // the index arithmetic of the reference (and ours) covers atoms at most one box image away; a blown-up trajectory is reported, not run
  if (h[S_TIMEOUT] != 0.0) FAIL(MSEVB_ERR_TIMEOUT, "a kernel of the MS-EVB call gave up waiting for a mark of another stream (2 s): a kernel of the side chains never finished, or the waiting kernel and the one it waits for could not run at the same time (profilers that serialise kernels: option evb_publish_spin=0)");
  if (h[S_FAULT] != 0.0) {      // code + 16 x (atom + 1) of the LAST kernel that met such a coordinate (k_pme_prep: 1; special atoms of the MS-EVB call: 3 / 4)
    const long v = (long)h[S_FAULT]; const int code = (int)(v % 16); const long atom = v / 16 - 1;
    char b[512]; double xv[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (atom >= 0 && atom < c->N) {      // its position, velocity and force as they are now (diagnostic; synchronous copies on an error path)
      (void)hipDeviceSynchronize();
      (void)hipMemcpy(xv, c->d_xyz.p + 3 * atom, 24, hipMemcpyDeviceToHost);
      if (c->d_vel.p) (void)hipMemcpy(xv + 3, c->d_vel.p + 3 * atom, 24, hipMemcpyDeviceToHost);
      (void)hipMemcpy(xv + 6, c->d_force.p + 3 * atom, 24, hipMemcpyDeviceToHost);
    }
    snprintf(b, sizeof b, "atom coordinates are not finite or more than one box length outside the cell (unstable trajectory?): atom %ld (0-based; kernel code %d) x = %.6g %.6g %.6g  v = %.6g %.6g %.6g  F = %.6g %.6g %.6g, EVB force call %ld",
             atom, code, xv[0], xv[1], xv[2], xv[3], xv[4], xv[5], xv[6], xv[7], xv[8], c->evb_calls);
    FAIL(MSEVB_ERR_NUMERIC, b);
  }
  if (c->list_valid) {   // flag for the NEXT call (src/general_routines.f90:1320-1326)
    const double skin = 1.2 * (c->rv - c->rc);
    if ((h[S_MAXD1] + h[S_MAXD2]) > skin) c->flag_verlet_list = 1;
  }
  c->E_recip = h[S_RECIP];
  en[MSEVB_E_ELEC] = h[S_ELEC_PAIR] + h[S_ELEC_INTRA] + h[S_RECIP] + c->ewald_self;
  en[MSEVB_E_VDW] = h[S_VDW_PAIR] + h[S_VDW_INTRA];
  en[MSEVB_E_BOND] = h[S_BOND]; en[MSEVB_E_ANGLE] = h[S_ANGLE]; en[MSEVB_E_DIHEDRAL] = h[S_DIH]; en[MSEVB_E_RECIP] = h[S_RECIP];
  en[MSEVB_E_POTENTIAL] = en[MSEVB_E_ELEC] + en[MSEVB_E_VDW] + en[MSEVB_E_BOND] + en[MSEVB_E_ANGLE] + en[MSEVB_E_DIHEDRAL];
  en[7] = 0.0;
  return MSEVB_OK;
}

int msevb_calculate_total_force_energy(msevb_ctx* c, double* energies, double* force) {
  NEED_FINAL();
  do {      // repeated only when a list row outgrew its stride in a rebuild (the stride has grown by then)
    CHK(total_force_enqueue(c));
    if (force) { pin_user_buffer(c, force, 3 * (size_t)c->N * 8); CHK(download_atoms3(c, force, c->d_force.p)); }
    CHK(fetch_energies(c));
  } while (nb_check_overflow(c));
  double en[MSEVB_N_ENERGY];
  CHK(total_force_collect(c, en));
  if (energies) memcpy(energies, en, sizeof en);
  return MSEVB_OK;
}

int msevb_get_pme_state(msevb_ctx* c, double* Q, double* theta, double* us, double* fr) {
  NEED_FINAL();
  const size_t K3 = (size_t)c->K * c->K * c->K;
  HIPCHK(hipStreamSynchronize(c->stream));
  if (Q) HIPCHK(hipMemcpy(Q, c->d_Q.p, K3 * 8, hipMemcpyDeviceToHost));
  if (theta) HIPCHK(hipMemcpy(theta, c->d_theta.p, K3 * 8, hipMemcpyDeviceToHost));
  if (us) HIPCHK(hipMemcpy(us, c->d_uscale.p, 3 * (size_t)c->N * 8, hipMemcpyDeviceToHost));
  if (fr) HIPCHK(hipMemcpy(fr, c->d_force_recip.p, 3 * (size_t)c->N * 8, hipMemcpyDeviceToHost));
  return MSEVB_OK;
}
int msevb_get_pme_indices(msevb_ctx* c, int* nearpt, int* splindex) {
  NEED_FINAL();
  CHK(pme_prepare_atoms(c, true));      // the B6 table rows are not kept by the force path: recompute them from the current positions
  HIPCHK(hipStreamSynchronize(c->stream));
  if (nearpt) HIPCHK(hipMemcpy(nearpt, c->d_base.p, 3 * (size_t)c->N * 4, hipMemcpyDeviceToHost));
  if (splindex) HIPCHK(hipMemcpy(splindex, c->d_splindex.p, 18 * (size_t)c->N * 4, hipMemcpyDeviceToHost));
  return MSEVB_OK;
}

int msevb_get_topology(msevb_ctx* c, int* at, double* chg, int* mn, int* mt, int* ma, int* hyd) {
  if (!c) return MSEVB_ERR_ARG;
  if (at) for (int i = 0; i < c->N; ++i) at[i] = c->h_atype[i] + 1;
  if (chg) memcpy(chg, c->h_chg.data(), c->N * 8);
  if (mn) memcpy(mn, c->h_mol_natom.data(), c->M * 4);
  if (mt) for (int m = 0; m < c->M; ++m) mt[m] = c->h_mol_type[m] + 1;
  if (ma) for (size_t k = 0; k < (size_t)c->M * MAX_MOL_ATOMS; ++k) ma[k] = c->h_mol_atoms[k] + 1;
  if (hyd) *hyd = c->hydronium + 1;
  return MSEVB_OK;
}

int msevb_get_phase_times(msevb_ctx* c, int* n, const char** names, double* ms, long* calls) {
  if (!c) return MSEVB_ERR_ARG;
  (void)hipStreamSynchronize(c->stream);
  phase_harvest(c);
  for (int p = 0; p < PH_COUNT; ++p) {
    PhaseTimer& t = c->ph[p];
    if (names) names[p] = t.name.c_str();
    if (ms) ms[p] = t.calls ? t.ms / t.calls : 0.0;
    if (calls) calls[p] = t.calls;
  }
  if (n) *n = PH_COUNT;
  return MSEVB_OK;
}
int msevb_reset_phase_times(msevb_ctx* c) { (void)hipStreamSynchronize(c->stream); phase_harvest(c); for (int p = 0; p < PH_COUNT; ++p) { c->ph[p].ms = 0; c->ph[p].calls = 0; } return MSEVB_OK; }
int msevb_set_option(msevb_ctx* c, const char* name, double v) {
  const std::string s(name);
  c->topo_epoch++;
  if (s == "evb_speculate") { c->evb_speculate = v != 0.0; return MSEVB_OK; }
  if (s == "evb_device_plan") { c->evb_device_plan = v != 0.0; return MSEVB_OK; }      // 0: enumeration + descriptor build on the host (evb_plan.hpp, the same source the device runs)
  if (s == "step_graph") { c->evb_step_graph = (int)v; return MSEVB_OK; }      // the enqueue part of a speculative MS-EVB call as four captured linear hipGraphs (evb.hip): 0 off, 1 on except with a real communicator, 2 on
  if (s == "phase_timing") { c->timing = v != 0.0; c->timing_now = false; return MSEVB_OK; }
  if (s == "phase_timing_stride") { c->timing_stride = v < 1.0 ? 1 : (int)v; return MSEVB_OK; }
  if (s == "fft_batch") { c->fft_batch = (int)v; return MSEVB_OK; }
  if (s == "spread_tiles") { c->spread_tiles = v != 0.0; return MSEVB_OK; }
  if (s == "pme_weightless") { c->pme_weightless = v != 0.0; return MSEVB_OK; }
  if (s == "cell_sort_own") { c->cell_sort_own = v != 0.0; return MSEVB_OK; }
  if (s == "pme_side_weights") { c->pme_side_weights = v != 0.0; return MSEVB_OK; }
  if (s == "spread_mfma") { c->spread_mfma = v != 0.0; return MSEVB_OK; }
  if (s == "gather_mfma") { c->gather_mfma = v != 0.0; return MSEVB_OK; }
  if (s == "evb_pair_first") { c->evb_pair_first = v != 0.0; return MSEVB_OK; }
  if (s == "evb_device_solve_ranks") { c->evb_device_solve_ranks = v != 0.0; return MSEVB_OK; }      // more than one rank: 1 = solve on the device behind the all-gather, blend and all-reduce pre-enqueued; 0 = the host's Jacobi between the collectives
  if (s == "evb_pme_split") { c->evb_pme_split = v != 0.0; return MSEVB_OK; }
  if (s == "evb_publish_spin") { c->evb_publish_spin = v != 0.0; return MSEVB_OK; }
  if (s == "evb_region_aside") { c->evb_region_aside = v != 0.0; return MSEVB_OK; }
  if (s == "evb_late_mol_join") { c->evb_late_mol_join = v != 0.0; return MSEVB_OK; }
  if (s == "evb_single_join") { c->evb_single_join = v != 0.0; return MSEVB_OK; }
  if (s == "evb_defer_fold") { c->evb_defer_fold = v != 0.0; return MSEVB_OK; }
  if (s == "evb_coupling_blocks") { c->evb_coupling_blocks = (int)v; return MSEVB_OK; }
  if (s == "evb_region_main") { c->evb_region_main = v != 0.0; return MSEVB_OK; }
  if (s == "evb_defer_molecules") { c->evb_defer_molecules = v != 0.0; c->evb_molecules_after_pair = v == 2.0; return MSEVB_OK; }      // 1: behind the coupling chain; 2: behind the PAIR LOOP (in the blend phase's window)
  if (s == "pme_gather_threads") { const int t = (int)v; if (t < 0 || t > 512 || (t & 63)) return MSEVB_ERR_ARG; c->pme_gather_threads = t; return MSEVB_OK; }
  if (s == "pme_tile_threads") { c->pme_tile_threads = v >= 512 ? 512 : 256; c->pme_graph_key[0] = 0; if (c->pme_graph_exec) { (void)hipGraphExecDestroy((hipGraphExec_t)c->pme_graph_exec); c->pme_graph_exec = nullptr; } return MSEVB_OK; }
  if (s == "fft_own") { c->fft_own = v != 0.0; return MSEVB_OK; }
  if (s == "spread_bricks") { c->spread_bricks = v != 0.0; return MSEVB_OK; }
  if (s == "evb_device_solve") { c->evb_device_solve = v != 0.0; c->evb_device_solve_method = v == 2.0 ? 0 : 1; return MSEVB_OK; }      // 1: squaring (+ Jacobi when needed), 2: Jacobi always, 0: host
  if (s == "pair_threads") { c->pair_threads = v == 512.0 ? 512 : 1024; return MSEVB_OK; }
  if (s == "evb_recip_linear") { c->evb_recip_linear = v != 0.0; return MSEVB_OK; }
  if (s == "ensemble_nvt") { c->nvt = v != 0.0; return MSEVB_OK; }
  if (s == "friction_coeff") { if (!(v > 0.0)) return MSEVB_ERR_ARG; c->friction = v; return MSEVB_OK; }
  if (s == "temperature") { if (!(v >= 0.0)) return MSEVB_ERR_ARG; c->temperature = v; return MSEVB_OK; }
  if (s == "langevin_seed") { c->langevin_seed = (unsigned long long)v; c->langevin_tick = 0; return MSEVB_OK; }
  if (s == "overlap") { c->overlap = v != 0.0; return MSEVB_OK; }
  if (s == "pair_prune") { c->prune = v != 0.0; if (c->prune && c->finalized) (void)nb_prune_reset(c); return MSEVB_OK; }
  if (s == "prune_delta") { if (!(v > 0.0)) return MSEVB_ERR_ARG; c->prune_delta = v; c->prune_delta_user = true;      /* (kept across msevb_set_system / box changes) */ if (c->prune && c->finalized) (void)nb_prune_reset(c); return MSEVB_OK; }
  if (s == "pin_host_arrays") { c->pin_host_arrays = v != 0.0; return MSEVB_OK; }
  if (s == "nbr_cap") { if (v < 64.0) return MSEVB_ERR_ARG; c->nbr_cap = ((unsigned)v + 63u) & ~63u; c->list_valid = false; return MSEVB_OK; }      // row stride (tests: force the overflow path)
  if (s == "nbr_tiles") { c->nbr_tiles = v != 0.0; c->list_valid = false; return MSEVB_OK; }
  if (s == "graphs") { c->graphs = v != 0.0; return MSEVB_OK; }
  if (s == "evb_element_stores") { c->evb_element_stores = v != 0.0; return MSEVB_OK; }
  if (s == "shard_principal") { c->shard_principal = v != 0.0; return MSEVB_OK; }
  if (s == "pair_flat") { c->pair_flat_opt = (int)v; if (c->prune && c->finalized) (void)nb_prune_reset(c); return MSEVB_OK; }
  if (s == "pair_split") {      // the rows' cut positions are counted by the list build only while the option is on: a list built without them is rebuilt now
    const bool was = c->pair_split_opt != 0;
    c->pair_split_opt = (int)v;
    if (!was && c->pair_split_opt != 0 && c->list_valid) CHK(nb_build(c));
    return MSEVB_OK;
  }
  if (s == "pair_self_mode") { c->pair_self_mode = v != 0.0; return MSEVB_OK; }
  if (s == "prune_freeze") { c->prune_freeze = v != 0.0; return MSEVB_OK; }
  if (s == "pair_taylor") { c->pair_taylor = v != 0.0; return MSEVB_OK; }
  if (s == "evb_fft_patch") { c->evb_fft_patch = v != 0.0; return MSEVB_OK; }
  if (s == "evb_blend_aside") { c->evb_blend_aside = v != 0.0; return MSEVB_OK; }
  if (s == "step_fuse_refresh") { c->step_fuse_refresh = v != 0.0; return MSEVB_OK; }
  if (s == "pair_pipe") { c->pair_pipe = v != 0.0; return MSEVB_OK; }
  if (s == "comm_timeout_ms") { if (!(v >= 1.0)) return MSEVB_ERR_ARG; c->comm_timeout_ms = (int)v; return MSEVB_OK; }
  if (s == "comm_stall_ms") { c->comm_stall_ms = (int)v; return MSEVB_OK; }
  if (s == "pair_tiles") { c->pair_tiles_opt = (int)v; c->list_valid = false; return MSEVB_OK; }
  c->err = "unknown option " + s;
  return MSEVB_ERR_ARG;
}
int msevb_get_info(msevb_ctx* c, const char* name, double* value) {
  const std::string s(name);
  if (!value) FAIL(MSEVB_ERR_ARG, "msevb_get_info: value is NULL");
  if (s == "pair_kernel") { *value = !c->list_valid ? -1.0 : c->pair_tiles ? (pair_flat_usable(c) ? 2.0 : 1.0) : 0.0; return MSEVB_OK; }
  if (s == "list_build_tiles") { *value = c->list_valid && c->nbr_tiles_used ? 1.0 : 0.0; return MSEVB_OK; }
  if (s == "pme_tiles") { *value = pme_tiles_ok(c) ? 1.0 : 0.0; return MSEVB_OK; }
  if (s == "pme_weightless") { *value = pme_weightless_now(c) ? 1.0 : 0.0; return MSEVB_OK; }
  if (s == "n_cells") { *value = (double)c->ncell[0] * c->ncell[1] * c->ncell[2]; return MSEVB_OK; }
  if (s == "cells_per_axis_min") { *value = (double)std::min(c->ncell[0], std::min(c->ncell[1], c->ncell[2])); return MSEVB_OK; }
  if (s == "fft_own") { *value = fft_lds_usable(c) ? 1.0 : 0.0; return MSEVB_OK; }
  if (s == "n_diabat") { *value = (double)c->D; return MSEVB_OK; }
  if (s == "spec_steps") { *value = (double)c->evb_spec_steps; return MSEVB_OK; }      // MS-EVB calls that started from the recorded diabat tree
  if (s == "spec_misses") { *value = (double)c->evb_spec_misses; return MSEVB_OK; }    // ... of which the device found the tree changed (call repeated with a fresh enumeration)
  if (s == "host_dfs_calls") { *value = (double)c->evb_host_dfs_calls; return MSEVB_OK; }      // enumerations (DFS + descriptor build) done on the host
  if (s == "device_plans") { *value = (double)c->evb_device_plans; return MSEVB_OK; }      // ... and on the device (k_evb_plan)
  if (s == "evb_calls") { *value = (double)c->evb_calls; return MSEVB_OK; }
  if (s == "graph_launches") { *value = (double)c->evb_graph_launches; return MSEVB_OK; }      // MS-EVB calls that went out as ONE hipGraphLaunch
  if (s == "graph_captures") { *value = (double)c->evb_graph_captures; return MSEVB_OK; }      // ... how often the call was captured (once per diabat tree and refresh mode)
  if (s == "graph_mismatches") { *value = (double)c->evb_graph_mismatches; return MSEVB_OK; }  // replays whose launches would have differed from the captured ones (call repeated the ordinary way): 0 unless the library has a bug
  c->err = "unknown info " + s;
  return MSEVB_ERR_ARG;
}

}  // extern "C"
