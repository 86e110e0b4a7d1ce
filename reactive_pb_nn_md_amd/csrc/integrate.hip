// Device-resident velocity Verlet (replaces md_integrate_atomic, NVE branch, src/md_integration.f90:438-541):
// half kick + drift (:469-486), molecule COM update and wrap into the box (update_r_com src/general_routines.f90:420,
// shift_molecules_into_box :1145), force call, second half kick with the |F| > 1e5 guard (:508-529), removal of the
// centre-of-mass momentum (:125-177) and calculate_kinetic_energy (src/total_energy_forces.f90:106).
// NVT (option ensemble_nvt): both kicks become the reference's Langevin leap-frog update (langevin_integrator, :186-228)
//     v <- a v + (1 - a)/gamma * F/m * 100 + sqrt(2 kB T gamma 1000) / sqrt(m/1000) / 100 * sqrt((1 - a^2)/(2 gamma)) * N(0,1),   a = exp(-gamma dt/2)
// with three fresh normal deviates per atom and kick.  The reference draws them with random_number() seeded from the clock
// (src/general_routines.f90:726), so its stream cannot be reproduced; ours is counter based - Philox4x32-10 keyed by the seed, counter
// (atom, kick number) - so a trajectory depends on the seed only, not on launch geometry, and the host can restate it (langevin.py).
#include "ctx.hpp"


namespace {

struct Langevin { int on; double a, b, c; unsigned k0, k1; unsigned long long kick; };      // b = (1-a)/gamma*100, c = noise amplitude * sqrt(1000)/100: times 1/sqrt(m)

__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1, unsigned out[4]) {
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
// three standard normal deviates for (atom, kick): Box-Muller on 32-bit uniforms in (0,1]
__device__ __forceinline__ void normal3(const Langevin& L, int atom, double z[3]) {
  unsigned r[4];
  philox4x32_10((unsigned)atom, (unsigned)L.kick, (unsigned)(L.kick >> 32), 0u, L.k0, L.k1, r);
  const double u0 = ((double)r[0] + 1.0) * (1.0 / 4294967296.0), u1 = (double)r[1] * (1.0 / 4294967296.0);
  const double u2 = ((double)r[2] + 1.0) * (1.0 / 4294967296.0), u3 = (double)r[3] * (1.0 / 4294967296.0);
  const double ra = sqrt(-2.0 * log(u0)), rb = sqrt(-2.0 * log(u2));
  z[0] = ra * cos(6.283185307179586 * u1); z[1] = ra * sin(6.283185307179586 * u1); z[2] = rb * cos(6.283185307179586 * u3);
}

// per-block partial sums of momentum (3) and kinetic energy
// (frozen atoms - atype_freeze - count in the kinetic energy, not in the momentum: src/md_integration.f90:125-177, total_energy_forces.f90:106)
__global__ __launch_bounds__(256) void k_momentum_ke(int N, const double* __restrict__ mass, const double* __restrict__ v, double* __restrict__ part, const int* __restrict__ frozen) {
  __shared__ double red[4][256];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  double p[4] = {0, 0, 0, 0};
  if (i < N) { const double m = mass[i]; const bool fr = frozen && frozen[i]; for (int d = 0; d < 3; ++d) { p[d] = fr ? 0.0 : m * v[3 * i + d]; p[3] += 0.5 * m * v[3 * i + d] * v[3 * i + d] / 100.0; } }
  for (int s = 0; s < 4; ++s) red[s][threadIdx.x] = p[s];
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) { if (threadIdx.x < st) for (int s = 0; s < 4; ++s) red[s][threadIdx.x] += red[s][threadIdx.x + st]; __syncthreads(); }
  if (threadIdx.x == 0) { part[(size_t)S_PX * MAX_BLOCKS + blockIdx.x] = red[0][0]; part[(size_t)S_PY * MAX_BLOCKS + blockIdx.x] = red[1][0];
    part[(size_t)S_PZ * MAX_BLOCKS + blockIdx.x] = red[2][0]; part[(size_t)S_KE * MAX_BLOCKS + blockIdx.x] = red[3][0]; }
}
__global__ void k_remove_momentum(int N, double n_free, const double* __restrict__ mass, const double* __restrict__ en, double* __restrict__ v, const int* __restrict__ frozen) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 3 * N || (frozen && frozen[i / 3])) return;
  v[i] -= en[S_PX + i % 3] / n_free / mass[i / 3];
}

// Fused step kernels of the device-resident MD loop -----------------------------------------------------------------------------
// first half: EIGHT LANES per molecule (lane k = the molecule's k-th atom; a thread per molecule walked its atoms through three dependent,
// uncoalesced loads each and took 17 us in front of everything else of the step) - removal of the centre-of-mass momentum measured at the
// end of the previous step (:125-177), half kick + drift of the atom (:469-486), then the molecule is shifted back into the box by its
// centre of mass, which the lanes sum in atom order from each other's registers (same sum, same bits as the serial loop).
// remove == 2: the momentum is summed HERE from the per-block partial sums the previous step's second kernel left (every block sums them
// again - 3 x nb numbers - which takes the reduction kernel and the 56-byte download out of the chain in front of the pair loop); block 0
// also finishes the kinetic-energy / velocity sums of that step and posts all seven to `post` (pinned host memory) and to the energy slots.
// REFRESH (option step_fuse_refresh, steps that keep their Verlet list): the kernel also does what k_step_refresh (neighbor.hip) does for the atom it
// has just moved - cell-sorted copy {x,y,z,q} (a scattered 32-byte store where the refresh kernel gathers), accumulated displacement since the list
// build and since the pair list was pruned with the per-block top-2 norms (update_verlet_displacements, src/general_routines.f90:1259), force
// accumulator zeroed - so that the pair loop starts one kernel and one launch gap (~15 us) earlier; k_top2_final follows as before.
struct StepRefresh { int on; const int* inv_perm; const double* chg; double4* posq; double* store; double* disp; double* part1; double* part2; const double4* posq_prune; double* part3; double* part4; };
__global__ __launch_bounds__(256) void k_step_first_half(int M, int N, double dt, int remove, const int* __restrict__ mol_natom, const int* __restrict__ mol_atoms,
                                                         const double* __restrict__ mass, double* __restrict__ F, double* __restrict__ en,
                                                         double* __restrict__ v, double* __restrict__ x, BoxDev box, Langevin L,
                                                         const double* __restrict__ part, int nb, double* __restrict__ post, const int* __restrict__ frozen, double n_free, StepRefresh R) {
  static_assert(MAX_MOL_ATOMS == 8, "eight lanes per molecule");
  __shared__ double red[7][4];
  double P[3] = {0.0, 0.0, 0.0};
  if (remove == 2) {
    const int slot[7] = {S_PX, S_PY, S_PZ, S_KE, S_VX, S_VY, S_VZ};
    const int nq = blockIdx.x == 0 ? 7 : 3;
    for (int q = 0; q < nq; ++q) {
      double s = 0.0;
      for (int k = threadIdx.x; k < nb; k += 256) s += part[(size_t)slot[q] * MAX_BLOCKS + k];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
      if ((threadIdx.x & 63) == 0) red[q][threadIdx.x >> 6] = s;
    }
    __syncthreads();
    for (int d = 0; d < 3; ++d) P[d] = ((red[d][0] + red[d][1]) + red[d][2]) + red[d][3];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      double tot[7];
      for (int q = 0; q < 7; ++q) tot[q] = ((red[q][0] + red[q][1]) + red[q][2]) + red[q][3];
      en[S_PX] = tot[0]; en[S_PY] = tot[1]; en[S_PZ] = tot[2]; en[S_KE] = tot[3]; en[S_VX] = tot[4]; en[S_VY] = tot[5]; en[S_VZ] = tot[6];
      if (post) { post[0] = tot[3]; post[1] = tot[0]; post[2] = tot[1]; post[3] = tot[2]; post[4] = tot[4]; post[5] = tot[5]; post[6] = tot[6]; }      // order of the slots S_KE..S_VZ
    }
  } else if (remove == 1) { P[0] = en[S_PX]; P[1] = en[S_PY]; P[2] = en[S_PZ]; }
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int m = t >> 3, k = t & 7, lane0 = (threadIdx.x & 63) & ~7;
  const int n = m < M ? mol_natom[m] : 0;
  const bool live = k < n;
  const int a = live ? mol_atoms[m * MAX_MOL_ATOMS + k] : 0;
  double ma = 0.0, xn[3] = {0.0, 0.0, 0.0}, vn[3] = {0.0, 0.0, 0.0};
  bool moved = false;
  // what the refresh part needs of this atom is requested NOW (it depends on the atom's index alone): in flight during the kick, the drift and the centre-of-mass
  // shuffles instead of two more dependent rounds behind them (sorted position -> snapshot row)
  int ks = 0; double4 qpr = make_double4(0.0, 0.0, 0.0, 0.0); double sto[3] = {0.0, 0.0, 0.0}, dsp[3] = {0.0, 0.0, 0.0}, chg_a = 0.0;
  if (live && R.on) {
    ks = R.inv_perm[a];
    for (int d = 0; d < 3; ++d) { sto[d] = R.store[3 * a + d]; dsp[d] = R.disp[3 * a + d]; }
    chg_a = R.chg[a];
    if (R.posq_prune) qpr = R.posq_prune[ks];
  }
  if (live) {
    ma = mass[a];
    if (frozen && frozen[a]) { for (int d = 0; d < 3; ++d) xn[d] = x[3 * a + d]; }      // neither kicked nor moved
    else {
      moved = true;
      double z[3] = {0, 0, 0};
      if (L.on) normal3(L, a, z);
      for (int d = 0; d < 3; ++d) {
        double vv = v[3 * a + d];
        if (remove) vv -= P[d] / n_free / ma;
        if (L.on) vv = L.a * vv + L.b * F[3 * a + d] / ma + L.c / sqrt(ma) * z[d];
        else vv = vv + dt / 2.0 / ma * F[3 * a + d] * 100.0;
        vn[d] = vv;
        xn[d] = x[3 * a + d] + vv * dt;
      }
    }
  }
  // centre of mass: the atoms' contributions in atom order (uniform trip count over the wave: shuffles)
  double com[3] = {0, 0, 0}, mt = 0;
  const double c0 = xn[0] * ma, c1 = xn[1] * ma, c2 = xn[2] * ma;
#pragma unroll
  for (int j = 0; j < MAX_MOL_ATOMS; ++j) {
    const double p0 = __shfl(c0, lane0 + j, 64), p1 = __shfl(c1, lane0 + j, 64), p2 = __shfl(c2, lane0 + j, 64), pm = __shfl(ma, lane0 + j, 64);
    if (j < n) { com[0] += p0; com[1] += p1; com[2] += p2; mt += pm; }
  }
  if (!live && !R.on) return;
  double nrm = 0.0, nrm_p = 0.0;
  if (live) {
    double sh[3];
    for (int d = 0; d < 3; ++d) { const double f = com[d] / mt * box.invL[d]; sh[d] = f < 0.0 ? box.L[d] : (f > 1.0 ? -box.L[d] : 0.0); }
    const bool any = sh[0] != 0.0 || sh[1] != 0.0 || sh[2] != 0.0;
    if (moved) for (int d = 0; d < 3; ++d) v[3 * a + d] = vn[d];
    double xf[3];
    for (int d = 0; d < 3; ++d) xf[d] = any ? xn[d] + sh[d] : xn[d];
    if (moved || any) for (int d = 0; d < 3; ++d) x[3 * a + d] = xf[d];
    if (R.on) {      // k_step_refresh's work for this atom, same expressions (neighbor.hip)
      if (R.posq_prune) {
        const double4 q = qpr;
        double e[3] = {xf[0] - q.x, xf[1] - q.y, xf[2] - q.z};
        for (int d = 0; d < 3; ++d) e[d] -= box.L[d] * floor(e[d] * box.invL[d] + 0.5);
        nrm_p = sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
      }
      R.posq[ks] = make_double4(xf[0], xf[1], xf[2], chg_a);
      double ad[3];
      for (int d = 0; d < 3; ++d) {
        const double dc = xf[d] - sto[d];
        const double shd = floor(dc * box.invL[d] + 0.5) * box.L[d];
        ad[d] = dsp[d] + (dc - shd);
        R.disp[3 * a + d] = ad[d];
        R.store[3 * a + d] = xf[d];
        F[3 * a + d] = 0.0;
      }
      nrm = sqrt(ad[0] * ad[0] + ad[1] * ad[1] + ad[2] * ad[2]);
    }
  }
  if (!R.on) return;
  // per-block top-2 of both displacement norms (the order of a maximum does not matter: same bits as k_step_refresh's blocks give k_top2_final): butterfly
  // over the wave, then the four waves' pairs through LDS - one barrier (the 256-wide LDS tree took eight per norm, ~4 us of this kernel)
  __shared__ double s12[4][4];
  double a1 = nrm, a2 = 0.0, p1 = nrm_p, p2 = 0.0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const double b1 = __shfl_xor(a1, o, 64), b2 = __shfl_xor(a2, o, 64), q1 = __shfl_xor(p1, o, 64), q2 = __shfl_xor(p2, o, 64);
    const double n1 = fmax(a1, b1), n2 = fmax(fmin(a1, b1), fmax(a2, b2)), m1 = fmax(p1, q1), m2 = fmax(fmin(p1, q1), fmax(p2, q2));
    a1 = n1; a2 = n2; p1 = m1; p2 = m2;
  }
  if ((threadIdx.x & 63) == 0) { double* w = s12[threadIdx.x >> 6]; w[0] = a1; w[1] = a2; w[2] = p1; w[3] = p2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    a1 = s12[0][0]; a2 = s12[0][1]; p1 = s12[0][2]; p2 = s12[0][3];
    for (int w = 1; w < 4; ++w) {
      const double b1 = s12[w][0], b2 = s12[w][1], q1 = s12[w][2], q2 = s12[w][3];
      const double n1 = fmax(a1, b1), n2 = fmax(fmin(a1, b1), fmax(a2, b2)), m1 = fmax(p1, q1), m2 = fmax(fmin(p1, q1), fmax(p2, q2));
      a1 = n1; a2 = n2; p1 = m1; p2 = m2;
    }
    R.part1[blockIdx.x] = a1; R.part2[blockIdx.x] = a2;
    if (R.posq_prune) { R.part3[blockIdx.x] = p1; R.part4[blockIdx.x] = p2; }
  }
}
// second half: one thread per atom - half kick with the |F| > 1e5 guard (:508-529) and the per-block partial sums of momentum and
// kinetic energy of the kicked velocities (calculate_kinetic_energy, src/total_energy_forces.f90:106)
__global__ __launch_bounds__(256) void k_step_second_half(int N, double dt, const double* __restrict__ mass, const double* __restrict__ F, double* __restrict__ v,
                                                          int* __restrict__ bad, double* __restrict__ part, Langevin L, const int* __restrict__ frozen,
                                                          LinOps tail, long long* lin_cnt, long long* lin_marks, double* lin_fault) {
  // tail: the waits that close the main chain of a force call that went out as linear graphs (ctx.hpp: LinCap) - everything that adds to the force array has finished;
  // performed here, by every workgroup, instead of by a one-thread kernel of their own between the gather and this kernel (~5 us on the step's critical path)
  if (tail.n > 0) { if (threadIdx.x == 0) lin_run_ops(tail, lin_cnt, lin_marks, lin_fault); __syncthreads(); }
  __shared__ double red[7][4];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  double p[7] = {0, 0, 0, 0, 0, 0, 0};
  if (i < N && frozen && frozen[i]) {      // frozen: velocity untouched, kinetic energy counted, no momentum
    const double m = mass[i];
    for (int d = 0; d < 3; ++d) p[3] += 0.5 * m * v[3 * i + d] * v[3 * i + d] / 100.0;
  } else if (i < N) {
    const double m = mass[i];
    double z[3] = {0, 0, 0};
    if (L.on) normal3(L, i, z);
    for (int d = 0; d < 3; ++d) {
      const double f = F[3 * i + d];
      const double vv = L.on ? L.a * v[3 * i + d] + L.b * f / m + L.c / sqrt(m) * z[d] : v[3 * i + d] + dt / 2.0 / m * f * 100.0;
      v[3 * i + d] = vv;
      if (fabs(f) > 10e4) atomicExch(bad, i + 1);
      p[d] = m * vv; p[3] += 0.5 * m * vv * vv / 100.0; p[4 + d] = vv;
    }
  }
  // workgroup sums: butterfly over the wave, the four waves' sums through LDS (one barrier; the 256-wide LDS tree took eight)
#pragma unroll
  for (int s = 0; s < 7; ++s) {
    double v = p[s];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0) red[s][threadIdx.x >> 6] = v;
  }
  __syncthreads();
  if (threadIdx.x < 7) {
    const int slot[7] = {S_PX, S_PY, S_PZ, S_KE, S_VX, S_VY, S_VZ};
    part[(size_t)slot[threadIdx.x] * MAX_BLOCKS + blockIdx.x] = ((red[threadIdx.x][0] + red[threadIdx.x][1]) + red[threadIdx.x][2]) + red[threadIdx.x][3];
  }
}

// The reference removes the centre-of-mass momentum at the end of the step (every atom gives up P/N, src/md_integration.f90:125-177,:528)
// and its kinetic energy is taken after that; we remove it inside the next step's first kernel, so the kinetic energy of the kicked
// velocities is corrected for it: sum m/2 (v - P/(N m))^2 = KE - P.sum(v)/N + P^2/(2 N^2) sum(1/m)
static double ke_after_removal(const msevb_ctx* c, const double* e /* S_KE.. S_VZ */, double inv_mass_sum) {
  const double* P = e + (S_PX - S_KE); const double* V = e + (S_VX - S_KE);
  const double N = c->n_free;      // the atoms that are integrated (all of them unless atom types are frozen)
  return e[0] - (P[0] * V[0] + P[1] * V[1] + P[2] * V[2]) / N / 100.0 + (P[0] * P[0] + P[1] * P[1] + P[2] * P[2]) / (2.0 * N * N) * inv_mass_sum / 100.0;
}

}  // namespace

static int momentum_ke(msevb_ctx* c) {
  const int nb = (c->N + 255) / 256;
  hipLaunchKernelGGL(k_momentum_ke, dim3(nb), dim3(256), 0, c->stream, c->N, c->d_mass.p, c->d_vel.p, c->d_partials.p, c->any_frozen ? c->d_frozen.p : (const int*)nullptr);
  HIPCHK(hipGetLastError());
  CHK(reduce_partials(c, S_PX, nb)); CHK(reduce_partials(c, S_PY, nb)); CHK(reduce_partials(c, S_PZ, nb)); CHK(reduce_partials(c, S_KE, nb));
  return flush_reductions(c);   // the momentum is consumed on the device right away
}

extern "C" int msevb_calculate_kinetic_energy(msevb_ctx* c, double* ke) {
  if (!c || !c->finalized) return MSEVB_ERR_STATE;
  CHK(momentum_ke(c));
  CHK(fetch_energies(c));
  *ke = c->h_energies[S_KE];
  return MSEVB_OK;
}

extern "C" int msevb_md_integrate_atomic(msevb_ctx* c, int n_steps, double dt, int use_evb, double* energies_out) {
  if (!c || !c->finalized) { if (c) c->err = "context not finalized"; return MSEVB_ERR_STATE; }
  (void)hipSetDevice(c->device);
  const int N = c->N, M = c->M, n3 = 3 * N;
  BoxDev box;
  for (int d = 0; d < 3; ++d) { box.L[d] = c->box[d]; box.invL[d] = 1.0 / c->box[d]; }
  DBuf<int>& d_bad = c->d_bad;
  if (d_bad.alloc(1)) FAIL(MSEVB_ERR_HIP, "hipMalloc failed");
  HIPCHK(hipMemsetAsync(d_bad.p, 0, sizeof(int), c->stream));
  // trajectory energies come back asynchronously: the EVB force call synchronises with the host on its own (Hamiltonian), so the loop
  // itself never waits; without EVB the host needs the Verlet flag of every step and fetches the energies at its end
  double*& h_ke = c->h_ke; int& h_ke_cap = c->h_ke_cap;      // 7 doubles per step: S_KE, S_P*, S_V*
  static_assert(S_PX == S_KE + 1 && S_VX == S_KE + 4 && S_VZ == S_KE + 6, "kinetic energy block of the energy slots");
  if (n_steps > h_ke_cap) {
    if (h_ke) HIPCHK(hipHostFree(h_ke));
    h_ke_cap = n_steps + 64;
    HIPCHK(hipHostMalloc((void**)&h_ke, 7 * (size_t)h_ke_cap * sizeof(double), hipHostMallocDefault));
  }
  const double inv_mass_sum = c->inv_mass_sum_free;
  const int* frz = c->any_frozen ? c->d_frozen.p : (const int*)nullptr;
  const int nbm8 = (int)(((size_t)M * MAX_MOL_ATOMS + 255) / 256), nba = (N + 255) / 256;
  Langevin L{};
  if (c->nvt) {      // constants of langevin_integrator (src/md_integration.f90:225): the 1/sqrt(m) stays in the kernel
    const double g = c->friction, a = exp(-g * dt / 2.0);
    L.on = 1; L.a = a; L.b = (1.0 - a) / g * 100.0;
    L.c = sqrt(2.0 * 0.008314462 * c->temperature * g * 1000.0) * sqrt(1000.0) / 100.0 * sqrt((1.0 - exp(-g * dt)) / (2.0 * g));
    L.k0 = (unsigned)c->langevin_seed; L.k1 = (unsigned)(c->langevin_seed >> 32);
  }
  for (int s = 0; s < n_steps; ++s) {
    phase_begin(c, PH_INTEGRATE);
    L.kick = c->langevin_tick++;
    // EVB loop (no host synchronisation of its own): from the second step on the kernel reduces the previous step's partial sums itself
    const bool self_reduce = use_evb && s > 0;
    StepRefresh R{};
    // the refresh of the sorted arrays rides in this kernel on steps that keep their list (decided exactly as total_force_enqueue will: the flag the previous
    // force call raised); a call that finds it has to rebuild after all (nb_settle) rebuilds from the positions, which are complete either way
    const bool fuse = c->step_fuse_refresh && c->list_valid && c->cells_valid && c->flag_verlet_list != 1 && !c->nbr_check_pending && !(use_evb && c->evb_region_main) && c->d_posq.p && c->d_xyz_store.p && c->d_inv_perm.p;
    if (fuse) {
      const bool pr = c->prune && c->d_posq_prune.p && c->d_prune_mode.p;
      R.on = 1; R.inv_perm = c->d_inv_perm.p; R.chg = c->d_chg.p; R.posq = c->d_posq.p; R.store = c->d_xyz_store.p; R.disp = c->d_disp.p;
      R.part1 = c->d_partials.p + (size_t)S_MAXD1 * MAX_BLOCKS; R.part2 = c->d_partials.p + (size_t)S_MAXD2 * MAX_BLOCKS;
      R.posq_prune = pr ? c->d_posq_prune.p : (const double4*)nullptr; R.part3 = c->d_partials.p + (size_t)S_PRD1 * MAX_BLOCKS; R.part4 = c->d_partials.p + (size_t)S_PRD2 * MAX_BLOCKS;
    }
    hipLaunchKernelGGL(k_step_first_half, dim3(nbm8), dim3(256), 0, c->stream, M, N, dt, self_reduce ? 2 : (s > 0 ? 1 : 0), c->d_mol_natom.p, c->d_mol_atoms.p, c->d_mass.p,
                       c->d_force.p, c->d_energies.p, c->d_vel.p, c->d_xyz.p, box, L, c->d_partials.p, nba, self_reduce ? h_ke + 7 * (size_t)(s - 1) : (double*)nullptr, frz, c->n_free, R);
    c->refresh_fused_nb = fuse ? nbm8 : 0;      // consumed (once) by the force call that follows
    phase_end(c, PH_INTEGRATE);
    double epot = 0.0;
    if (use_evb) { c->lin_tail_defer = true; const int rc_f = evb_force_energy(c, &epot); c->lin_tail_defer = false; CHK(rc_f); }
    else {
      CHK(total_force_enqueue(c));
      while (c->nbr_check_pending) {      // rebuild step: the half kick below must not run on forces from a list with a cut row
        HIPCHK(hipStreamSynchronize(c->stream));
        if (!nb_check_overflow(c)) break;
        CHK(total_force_enqueue(c));
      }
    }
    c->refresh_fused_nb = 0;
    L.kick = c->langevin_tick++;
    LinOps tail = c->lin_tail; c->lin_tail.n = 0;      // (left by a force call that went out as linear graphs: its closing waits ride in this kernel)
    hipLaunchKernelGGL(k_step_second_half, dim3(nba), dim3(256), 0, c->stream, N, dt, c->d_mass.p, c->d_force.p, c->d_vel.p, d_bad.p, c->d_partials.p, L, frz, tail, c->lin_cnt, c->lin_marks, c->d_energies.p + S_TIMEOUT);
    HIPCHK(hipGetLastError());
    if (!use_evb || s == n_steps - 1) {      // (inside the EVB loop the next step's first kernel does this)
      CHK(reduce_partials(c, S_PX, nba)); CHK(reduce_partials(c, S_PY, nba)); CHK(reduce_partials(c, S_PZ, nba)); CHK(reduce_partials(c, S_KE, nba));
      CHK(reduce_partials(c, S_VX, nba)); CHK(reduce_partials(c, S_VY, nba)); CHK(reduce_partials(c, S_VZ, nba));
    }
    if (use_evb) {
      if (s == n_steps - 1) {
        CHK(flush_reductions(c));
        HIPCHK(hipMemcpyAsync(h_ke + 7 * (size_t)s, c->d_energies.p + S_KE, 7 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
      }
      if (energies_out) energies_out[2 * s] = epot;
    } else {
      CHK(fetch_energies(c));   // one host synchronisation per step: energies + Verlet flag
      double en[MSEVB_N_ENERGY]; CHK(total_force_collect(c, en));
      if (energies_out) { energies_out[2 * s] = en[MSEVB_E_POTENTIAL]; energies_out[2 * s + 1] = ke_after_removal(c, c->h_energies + S_KE, inv_mass_sum); }
    }
  }
  if (n_steps > 0) hipLaunchKernelGGL(k_remove_momentum, dim3((n3 + 255) / 256), dim3(256), 0, c->stream, N, c->n_free, c->d_mass.p, c->d_energies.p, c->d_vel.p, frz);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(c->stream));
  if (use_evb && energies_out) for (int s = 0; s < n_steps; ++s) energies_out[2 * s + 1] = ke_after_removal(c, h_ke + 7 * (size_t)s, inv_mass_sum);
  int bad = 0;
  HIPCHK(hipMemcpy(&bad, d_bad.p, sizeof(int), hipMemcpyDeviceToHost));
  if (bad) { char b[96]; snprintf(b, sizeof b, "force on atom %d is too big", bad); FAIL(MSEVB_ERR_NUMERIC, b); }
  return MSEVB_OK;
}
