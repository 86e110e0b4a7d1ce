// MS-EVB (replaces ms_evb_calculate_total_force_energy, src/ms_evb.f90:181, and everything under it).
//
// Design (MI355X-first, not the reference's array shifting):
//   * atoms keep the PRINCIPAL numbering for the whole step; a diabat is a sparse delta: the <= (max_chain+1) molecules
//     touched by its proton hops carry overridden types / charges / membership / proton image ("special atoms");
//   * host: tiny work only - diabat enumeration on a device-gathered neighbourhood of the hydronium
//     (evb_conduct_proton_transfer_recursive :498, find_evb_reactive_neighbors :702), hop bookkeeping
//     (evb_change_data_structures_proton_transfer :843), Jacobi (:242);
//   * device, batched over ALL evaluation units of all owned diabats in single launches:
//       k_evb_env      donor/acceptor atoms x all other atoms real space (:1566) + MS-EVB3 repulsion (:2259-2478)
//       k_evb_special  pairs among special atoms, intra-molecular non-bonded and bonded terms of donor/acceptor (:1849, :1900)
//       k_evb_patch    Q_grid patches (modify_Q_grid, src/pme.f90:275) ; rocFFT ; xCB ; energy (:1962-2050)
//       k_evb_gather   reciprocal force of every atom for up to 8 diabats per pass with the principal's stencil weights,
//                      RECOMPUTED weights for special atoms instead of the reference's stored dQ_dr (:2055-2084, :2103)
//       k_evb_coupling exchange-charge Coulomb sum of the off-diagonal element (:1276) + geometric factor (:1180)
//       k_evb_blend    Hellmann-Feynman force (:293-309)
//   * diabats d >= 2 are sharded over ranks ((d-2) mod n_ranks); force slots are laid out [rank][slot] so that one
//     RCCL all-gather completes every rank's store.
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <functional>
#include <map>

#include <atomic>

#include "ctx.hpp"
#include "devmath.hpp"

#include "molecule_dev.hpp"
#include "gather_dev.hpp"

namespace {

constexpr int MAX_SPEC = (MAX_CHAIN + 1) * MAX_MOL_ATOMS;
constexpr int CPL_BINS = 32;      // the workgroups of a coupling add their Zundel-force sums into one of 32 copies (blockIdx.x % 32): up to 1024 workgroups adding to ONE
                                  // set of addresses serialised in the L2 (same-address atomics cost ~30 ns each: 30 us per launch); k_evb_coupling_finish sums the copies
constexpr int REGION_CAP = 8192;
constexpr int REGION_FAST = 2048;     // records fetched speculatively together with the count (one synchronisation)
constexpr int MAX_UNITS = 2 * MAX_CHAIN * MAX_DIABATS + 1;

struct SpecAtom { double x[3]; double q; int idx, type, mol, pad; };

struct UnitDesc {
  double sign;
  int slot;          // force slot (index into the diagonal force store)
  int do_rs;         // 0: repulsion only (principal diabat)
  int do_rep;        // MS-EVB3 repulsion of this topology's hydronium (only the LAST topology of a chain needs it: the
                     // repulsion of hop h's new side and of hop h+1's old side are the same topology and cancel)
  int n_spec, n_mols;
  int mols[MAX_CHAIN + 1];
  int donor, acceptor, hyd;
  int nD, nA, nH;
  int D_at[MAX_MOL_ATOMS], A_at[MAX_MOL_ATOMS], H_at[MAX_MOL_ATOMS];   // indices into spec[], local molecule order
  int mt_D, mt_A;
  int heavy;         // spec index of the hydronium heavy atom
  int type_H;        // atom type of the hydronium's last atom
  SpecAtom spec[MAX_SPEC];
};

struct EvbTablesDev {
  int n_da, n_pa;
  int da_types[MAX_INTERACTION][3]; double da_par[MAX_INTERACTION][6];
  int pa_types[MAX_INTERACTION][2]; double pa_par[MAX_INTERACTION][5];
  double exch_atomic[MAX_TYPES];
  double screen2;     // squared screening radius (heavy atom - any atom) beyond which every repulsion term is zero
  int reactive_p[MAX_MOLTYPES][MAX_MOL_ATOMS], reactive_b[MAX_MOLTYPES][MAX_MOL_ATOMS];      // evb_reactive_protons / evb_reactive_basic_atoms of the molecule types
};

struct CouplingDesc {   // one off-diagonal element, evaluated in the acceptor topology of diabat d
  int slot;             // off-diagonal force slot
  int n_spec, n_mols, mols[MAX_CHAIN + 1];
  int nD, nA, D_at[MAX_MOL_ATOMS], A_at[MAX_MOL_ATOMS];
  int gO1, gO2, gH;     // spec indices: donor heavy, acceptor heavy, transferring proton
  int ftype;
  double par[10];
  double q_transfer;
  double zc[3], shiftd[3], shifta[3];
  double mol_com[MAX_CHAIN + 1][3];   // COM of each special molecule in this topology
  SpecAtom spec[MAX_SPEC];
};

struct RecipAtom { int grid; int slot; int idx; int pad; double q_new; double x_new[3]; double q_old; };

struct RegionRec { int mol, natom; double com[3]; double x[MAX_MOL_ATOMS][3]; };

__device__ __forceinline__ void min_image(double* d, const BoxDev& b) {
  for (int k = 0; k < 3; ++k) d[k] -= b.L[k] * floor(d[k] * b.invL[k] + 0.5);
}
__device__ __forceinline__ void rep_switch(double r, double rs, double rc, double& sw, double& dsw) {   // src/ms_evb.f90:2484
  sw = 0.0; dsw = 0.0;
  if (r < rc) {
    if (r < rs) sw = 1.0;
    else { const double c3 = (rc - rs) * (rc - rs) * (rc - rs), t1 = (r - rs) * (r - rs) / c3, t2 = 3.0 * rc - rs - 2.0 * r; sw = 1.0 - t1 * t2; dsw = -2.0 * (r - rs) * t2 / c3 + 2.0 * t1; }
  }
}

// repulsion between the hydronium of unit u and ONE other atom j (type tj, position xj): three-atom term
// (src/ms_evb.f90:2295-2399) + Born-Mayer terms (:2405-2478).  Adds the force on j to fj and the forces on hydronium
// atoms atomically to the LDS table sF[spec index] (k over u.H_at).  pbc_shift of the reference = nearest image of j relative to the hydronium atom.
__device__ __forceinline__ double evb_repulsion_atom(const UnitDesc& u, const EvbTablesDev& tb, const BoxDev& box, int tj, const double* xj, double* fj, double (*sF)[3]) {
  double E = 0.0;
  int idx = -1;
  const SpecAtom& hv = u.spec[u.heavy];
  for (int q = 0; q < tb.n_da; ++q) if (tb.da_types[q][0] == tj && tb.da_types[q][1] == hv.type && tb.da_types[q][2] == u.type_H) { idx = q; break; }
  if (idx >= 0) {
    const double* p = tb.da_par[idx];
    double sh[3], rO[3];
    for (int d = 0; d < 3; ++d) { const double dc = xj[d] - hv.x[d]; sh[d] = floor(dc * box.invL[d] + 0.5) * box.L[d]; rO[d] = -(dc - sh[d]); }
    const double rOO = sqrt(rO[0] * rO[0] + rO[1] * rO[1] + rO[2] * rO[2]);
    double sw, dsw;
    rep_switch(rOO, p[4], p[5], sw, dsw);
    if (sw != 0.0 || dsw != 0.0) {
      const double fac = p[0] * exp(-p[1] * (rOO - p[2]));
      double sum = 0.0;
      int hk = -1;
      for (int k = 0; k < u.nH; ++k) if (u.H_at[k] == u.heavy) hk = k;
      for (int k = 0; k < u.nH; ++k) {
        const SpecAtom& ha = u.spec[u.H_at[k]];
        if (ha.type != u.type_H) continue;
        double q[3];
        for (int d = 0; d < 3; ++d) { const double rij = -(xj[d] - ha.x[d] - sh[d]); q[d] = (2.0 * xj[d] + rO[d]) / 2.0 - (xj[d] + rij); }
        const double q2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2], eq = exp(-p[3] * q2);
        sum += eq;
        for (int d = 0; d < 3; ++d) {
          const double g = sw * fac * eq * p[3] * q[d];
          atomicAdd(&sF[u.H_at[k]][d], -2.0 * g); atomicAdd(&sF[u.H_at[hk]][d], g); fj[d] += g;
        }
      }
      E += sw * fac * sum;
      for (int d = 0; d < 3; ++d) { const double f = rO[d] / rOO * fac * sum * (sw * p[1] - dsw); atomicAdd(&sF[u.H_at[hk]][d], f); fj[d] -= f; }
    }
  }
  for (int k = 0; k < u.nH; ++k) {
    const SpecAtom& ha = u.spec[u.H_at[k]];
    int ip = -1;
    for (int q = 0; q < tb.n_pa; ++q) if (tb.pa_types[q][0] == tj && tb.pa_types[q][1] == ha.type) { ip = q; break; }
    if (ip < 0) continue;
    const double* p = tb.pa_par[ip];
    double rij[3];
    for (int d = 0; d < 3; ++d) { const double dc = xj[d] - ha.x[d]; rij[d] = -(dc - floor(dc * box.invL[d] + 0.5) * box.L[d]); }
    const double r = sqrt(rij[0] * rij[0] + rij[1] * rij[1] + rij[2] * rij[2]);
    double sw, dsw;
    rep_switch(r, p[3], p[4], sw, dsw);
    if (sw == 0.0 && dsw == 0.0) continue;
    const double fac = p[0] * exp(-p[1] * (r - p[2]));
    E += sw * fac;
    for (int d = 0; d < 3; ++d) { const double f = rij[d] / r * fac * (sw * p[1] - dsw); atomicAdd(&sF[u.H_at[k]][d], f); fj[d] -= f; }
  }
  return E;
}

// ---- candidate atoms: everything within R of the hydronium centre of mass (R = farthest special atom + cutoff), compacted once
// per step so that the environment kernel of every unit walks ~2 % of the box instead of all atoms.  Also re-arms the counter
// of k_evb_region for the next step.
__global__ __launch_bounds__(256) void k_evb_candidates(int N, double cx, double cy, double cz, double R2, const double* __restrict__ xyz, BoxDev box,
                                                        int* __restrict__ counter, int* __restrict__ cand, int* __restrict__ region_counter, const double* __restrict__ sphere = nullptr) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (sphere) { cx = sphere[0]; cy = sphere[1]; cz = sphere[2]; R2 = sphere[3]; }      // speculative step: centre and radius from k_evb_refresh
  if (i == 0 && region_counter) *region_counter = 0;
  bool in = false;
  if (i < N) {
    double dr[3] = {xyz[3 * i] - cx, xyz[3 * i + 1] - cy, xyz[3 * i + 2] - cz};
    min_image(dr, box);
    in = dr[0] * dr[0] + dr[1] * dr[1] + dr[2] * dr[2] < R2;
  }
  const unsigned long long m = __ballot(in);
  if (m == 0ull) return;
  const int lane = threadIdx.x & 63;
  int b = 0;
  if (lane == 0) b = atomicAdd(counter, __popcll(m));
  b = __shfl(b, 0, 64);
  if (in) cand[b + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u))] = i;
}

// ---- environment kernel: special (donor+acceptor / hydronium) atoms of every unit against the candidate NON-special atoms --
// lane = (candidate j, special atom q): 16 q-lanes x 4 candidates per wavefront trip, one pair per lane; the force on j is the butterfly
// sum over its q-lanes (one atomic triple per candidate), the force on special atom q accumulates per lane and meets in LDS at the end.
// The MS-EVB3 repulsion (rare: candidates within the screening radius of the hydronium) is done by the q = 0 lane and adds straight into
// the LDS table.  No private arrays (the thread-per-candidate version kept the hydronium force block in scratch memory).
// (round 5, measured: held to 64 vector registers - the most that fits beside a pair-loop workgroup on its CU - by amdgpu_waves_per_eu(8, 8) this kernel spills 115 registers
//  and the step loses 4-7 %: the spills cost more than the co-residence gives, HISTORY.md F)
__global__ __launch_bounds__(256) void k_evb_env(const int* __restrict__ n_cand, const int* __restrict__ cand, int n_units, const UnitDesc* __restrict__ units, const EvbTablesDev* __restrict__ tabs,
                                                 const double* __restrict__ xyz, const double* __restrict__ chg, const int* __restrict__ atype,
                                                 const int* __restrict__ mol_of, const PairParam* __restrict__ pairpar, int nt,
                                                 const double2* __restrict__ etab, double inv_dx, const double* __restrict__ tt,
                                                 const double* __restrict__ dtt, BoxDev box, double rc2, double* __restrict__ Fstore, size_t fstride,
                                                 double* __restrict__ part, int part_stride) {
  constexpr int QL = 2 * MAX_MOL_ATOMS, CPW = 64 / QL;
  __shared__ UnitDesc u;
  __shared__ EvbTablesDev tb;
  __shared__ double sF[MAX_SPEC][3];
  __shared__ double red[256];
  const int iu = blockIdx.y;
  {
    const int* src = (const int*)&units[iu]; int* dst = (int*)&u;
    for (int k = threadIdx.x; k < (int)(sizeof(UnitDesc) / 4); k += blockDim.x) dst[k] = src[k];
    const int* s2 = (const int*)tabs; int* d2 = (int*)&tb;
    for (int k = threadIdx.x; k < (int)(sizeof(EvbTablesDev) / 4); k += blockDim.x) d2[k] = s2[k];
    for (int k = threadIdx.x; k < MAX_SPEC * 3; k += blockDim.x) (&sF[0][0])[k] = 0.0;
  }
  __syncthreads();
  double* F = Fstore + (size_t)u.slot * fstride;
  const int lane = threadIdx.x & 63, q = lane & (QL - 1), cs = lane / QL;
  const int nz = u.do_rs ? u.nD + u.nA : 0;
  const bool qlive = q < nz;
  const int sp = qlive ? (q < u.nD ? u.D_at[q] : u.A_at[q - u.nD]) : 0;
  const double ax = u.spec[sp].x[0], ay = u.spec[sp].x[1], az = u.spec[sp].x[2], aq = u.spec[sp].q;
  const int atyp = u.spec[sp].type;
  double e = 0.0, fsx = 0.0, fsy = 0.0, fsz = 0.0;
  const int ncand = *n_cand;
  const int wave_g = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), n_waves = gridDim.x * (blockDim.x >> 6);
  for (int c0 = wave_g * CPW; c0 < ncand; c0 += n_waves * CPW) {      // wave-uniform trip count
    const int cj = c0 + cs;
    const bool have = cj < ncand;
    const int j = have ? cand[cj] : 0;
    bool special = !have;
    const int mj = mol_of[j];
    for (int k = 0; k < u.n_mols; ++k) special |= (u.mols[k] == mj);
    const double xj[3] = {xyz[3 * j], xyz[3 * j + 1], xyz[3 * j + 2]};
    const int tj = atype[j];
    double fx = 0.0, fy = 0.0, fz = 0.0;
    if (!special && qlive) {
      double dr[3] = {ax - xj[0], ay - xj[1], az - xj[2]};
      min_image(dr, box);
      const double r2 = dr[0] * dr[0] + dr[1] * dr[1] + dr[2] * dr[2];
      if (r2 < rc2) {
        double fs, ee, ev;
        pair_interaction(r2, aq * chg[j], pairpar[atyp * nt + tj], etab, inv_dx, tt, dtt, fs, ee, ev);
        e += ee + ev;
        fx = -fs * dr[0]; fy = -fs * dr[1]; fz = -fs * dr[2];
        fsx -= fx; fsy -= fy; fsz -= fz;
      }
    }
    if (u.do_rep && !special && q == 0) {   // repulsion: only atoms near the hydronium matter; cheap distance screen first
      const SpecAtom& hv = u.spec[u.heavy];
      double dr[3] = {hv.x[0] - xj[0], hv.x[1] - xj[1], hv.x[2] - xj[2]};
      min_image(dr, box);
      if (dr[0] * dr[0] + dr[1] * dr[1] + dr[2] * dr[2] < tb.screen2) {
        double fr[3] = {0, 0, 0};
        e += evb_repulsion_atom(u, tb, box, tj, xj, fr, sF);
        fx += fr[0]; fy += fr[1]; fz += fr[2];
      }
    }
#pragma unroll
    for (int o = 1; o < QL; o <<= 1) { fx += __shfl_xor(fx, o, 64); fy += __shfl_xor(fy, o, 64); fz += __shfl_xor(fz, o, 64); }
    if (!special && q < 3) { const double v = q == 0 ? fx : q == 1 ? fy : fz; if (v != 0.0) atomicAdd(&F[3 * j + q], u.sign * v); }
  }
#pragma unroll
  for (int o = QL; o < 64; o <<= 1) { fsx += __shfl_xor(fsx, o, 64); fsy += __shfl_xor(fsy, o, 64); fsz += __shfl_xor(fsz, o, 64); }
  if (cs == 0 && qlive) { atomicAdd(&sF[sp][0], fsx); atomicAdd(&sF[sp][1], fsy); atomicAdd(&sF[sp][2], fsz); }
  red[threadIdx.x] = e;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) { if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st]; __syncthreads(); }
  if (threadIdx.x == 0) part[(size_t)iu * part_stride + blockIdx.x] = red[0];
  for (int k = threadIdx.x; k < u.n_spec * 3; k += blockDim.x) {
    const double v = sF[k / 3][k % 3];
    if (v != 0.0) atomicAdd(&F[3 * u.spec[k / 3].idx + k % 3], u.sign * v);
  }
}

// ---- special-special kernel: one wavefront per unit ---------------------------------------------------------------------
// lanes share the special x special pair loop and the repulsion loop; lanes 0 / 1 do the intra-molecular + bonded terms of
// donor / acceptor; forces are collected in LDS (ds_add_f64) and leave with one global atomic per component.
__global__ __launch_bounds__(64) void k_evb_special(int n_units, const UnitDesc* __restrict__ units, const EvbTablesDev* __restrict__ tabs, const MolTypeDev* __restrict__ mts,
                              const PairParam* __restrict__ pairpar, int nt, const double2* __restrict__ etab, double inv_dx,
                              const double* __restrict__ tt, const double* __restrict__ dtt, BoxDev box, double rc2, double alpha, double erf_factor,
                              double conv, double pi_ref, double* __restrict__ Fstore, size_t fstride, double* __restrict__ Eout,
                              const double* __restrict__ env_part, int env_nb, int env_stride, double* __restrict__ Eenv) {
  const int iu = blockIdx.x, lane = threadIdx.x;
  if (iu >= n_units) return;
  if (env_part) {      // the unit's environment energy: sum of k_evb_env's per-block partial sums (was a k_sum_rows launch between the two kernels)
    double e = 0.0;
    for (int b = lane; b < env_nb; b += 64) e += env_part[(size_t)iu * env_stride + b];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) e += __shfl_xor(e, o, 64);
    if (lane == 0) Eenv[iu] = e;
  }
  __shared__ UnitDesc u;
  __shared__ EvbTablesDev tb;
  __shared__ double f[MAX_SPEC][3];
  __shared__ double xs[2][MAX_MOL_ATOMS][3], qs[2][MAX_MOL_ATOMS], fls[2][MAX_MOL_ATOMS][3];
  __shared__ bool inDA[MAX_SPEC];
  {
    const int* src = (const int*)&units[iu]; int* dst = (int*)&u;
    for (int k = lane; k < (int)(sizeof(UnitDesc) / 4); k += 64) dst[k] = src[k];
    const int* s2 = (const int*)tabs; int* d2 = (int*)&tb;
    for (int k = lane; k < (int)(sizeof(EvbTablesDev) / 4); k += 64) d2[k] = s2[k];
    for (int k = lane; k < MAX_SPEC * 3; k += 64) (&f[0][0])[k] = 0.0;
    for (int k = lane; k < MAX_SPEC; k += 64) inDA[k] = false;
  }
  __syncthreads();
  if (lane == 0 && u.do_rs) { for (int k = 0; k < u.nD; ++k) inDA[u.D_at[k]] = true; for (int k = 0; k < u.nA; ++k) inDA[u.A_at[k]] = true; }
  __syncthreads();
  double e = 0.0;
  if (u.do_rs) {
    const int np = u.n_spec * u.n_spec;
    for (int t = lane; t < np; t += 64) {
      const int a = t / u.n_spec, b = t % u.n_spec;
      if (!inDA[a] || u.spec[b].mol == u.spec[a].mol || (inDA[b] && b < a)) continue;
      double dr[3] = {u.spec[a].x[0] - u.spec[b].x[0], u.spec[a].x[1] - u.spec[b].x[1], u.spec[a].x[2] - u.spec[b].x[2]};
      min_image(dr, box);
      const double r2 = dr[0] * dr[0] + dr[1] * dr[1] + dr[2] * dr[2];
      if (r2 < rc2) {
        double fs, ee, ev;
        pair_interaction(r2, u.spec[a].q * u.spec[b].q, pairpar[u.spec[a].type * nt + u.spec[b].type], etab, inv_dx, tt, dtt, fs, ee, ev);
        e += ee + ev;
        for (int d = 0; d < 3; ++d) { atomicAdd(&f[a][d], fs * dr[d]); atomicAdd(&f[b][d], -fs * dr[d]); }
      }
    }
    // intra-molecular non-bonded + bonded terms of donor (lanes 0-31) and acceptor (lanes 32-63), the terms of a molecule spread over
    // its 32 lanes (the 8-atom acid has 28 pairs and a dozen bonded terms: one lane per molecule was the longest path of this kernel)
    {
      const int side = lane >> 5, sl = lane & 31;
      const int n = side == 0 ? u.nD : u.nA;
      const int* lst = side == 0 ? u.D_at : u.A_at;
      const MolTypeDev& mt = mts[side == 0 ? u.mt_D : u.mt_A];
      for (int k = sl; k < n; k += 32) { qs[side][k] = u.spec[lst[k]].q; for (int d = 0; d < 3; ++d) { xs[side][k][d] = u.spec[lst[k]].x[d]; fls[side][k][d] = 0.0; } }
      __syncthreads();
      double e1 = 0, e2 = 0, eb = 0, ea = 0, ed = 0;
      if (n > 1) molecule_nonbonded<true>(mt, n, xs[side], qs[side], fls[side], etab, inv_dx, tt, dtt, rc2, alpha, erf_factor, conv, e1, e2, sl, 32);
      molecule_bonded<true>(mt, xs[side], fls[side], pi_ref, eb, ea, ed, sl, 32);
      e += e1 + e2 + eb + ea + ed;
      __syncthreads();
      for (int k = sl; k < n; k += 32) for (int d = 0; d < 3; ++d) atomicAdd(&f[lst[k]][d], fls[side][k][d]);
    }
  }
  if (u.do_rep) {   // repulsion of the hydronium with special atoms of OTHER molecules
    for (int b = lane; b < u.n_spec; b += 64) {
      if (u.spec[b].mol == u.hyd) continue;
      double fj[3] = {0, 0, 0};
      e += evb_repulsion_atom(u, tb, box, u.spec[b].type, u.spec[b].x, fj, f);      // hydronium forces go straight into the LDS table
      for (int d = 0; d < 3; ++d) atomicAdd(&f[b][d], fj[d]);
    }
  }
  for (int o = 32; o > 0; o >>= 1) e += __shfl_xor(e, o, 64);
  __syncthreads();
  double* F = Fstore + (size_t)u.slot * fstride;
  for (int k = lane; k < u.n_spec * 3; k += 64) { const double v = f[k / 3][k % 3]; if (v != 0.0) atomicAdd(&F[3 * u.spec[k / 3].idx + k % 3], u.sign * v); }
  if (lane == 0) Eout[iu] = e;
}

// ---- reciprocal space -------------------------------------------------------------------------------------------------
__global__ void k_evb_grid_init(size_t K3, int ng, const double* __restrict__ Q, double2* __restrict__ grids) {
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < K3 * ng; k += (size_t)gridDim.x * blockDim.x) grids[k] = make_double2(Q[k % K3], 0.0);
}

// stencil of one special atom in its NEW state: same index arithmetic as k_pme_prep, one thread per (entry, axis, tap) - a thread per entry
// walked 54 dependent table reads (12 us for a hundred atoms).
// xyz != nullptr: entries n .. 2n - 1 are the same atoms as the PRINCIPAL diabat has them (current positions) - the values k_pme_prep writes
// for those atoms, recomputed here so that the linear-response path does not have to wait for the principal front end (outputs obase / ow / odw)
__global__ void k_evb_spec_prep(int n, const RecipAtom* __restrict__ ra, int K, int G, double k00, double k01, double k02, double k10, double k11,
                                double k12, double k20, double k21, double k22, const double* __restrict__ B6, const double* __restrict__ B5,
                                int* __restrict__ sbase, double* __restrict__ sw, double* __restrict__ sdw, double* __restrict__ fault, const double* __restrict__ xyz = nullptr,
                                int* __restrict__ obase = nullptr, double* __restrict__ ow = nullptr, double* __restrict__ odw = nullptr) {
#pragma clang fp contract(off)
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 18 * n * (xyz ? 2 : 1)) return;
  const bool old_state = t >= 18 * n;
  const int tt = old_state ? t - 18 * n : t;
  const int e = tt / 18, r = tt - 18 * e, l = r / 6, k = r - 6 * l;
  const double x = old_state ? xyz[3 * ra[e].idx] : ra[e].x_new[0], y = old_state ? xyz[3 * ra[e].idx + 1] : ra[e].x_new[1], z = old_state ? xyz[3 * ra[e].idx + 2] : ra[e].x_new[2];
  const double ka = l == 0 ? k00 : l == 1 ? k10 : k20, kb = l == 0 ? k01 : l == 1 ? k11 : k21, kc = l == 0 ? k02 : l == 1 ? k12 : k22;
  double u = (double)K * ((ka * x + kb * y) + kc * z);
  if (u < 0.0) u = u + (double)K; else if (u >= (double)K) u = u - (double)K;
  if (!(u >= 0.0 && u < (double)K)) { *fault = 16.0 * (double)(ra[e].idx + 1) + (old_state ? 3.0 : 4.0); u = 0.5; }      // see k_pme_prep (codes 3 / 4: a special atom in its principal / diabat state)
  if (fabs(fmod(u, 1.0)) < 1e-6) u = u + 1e-6;
  const int np = (int)floor(u);
  int* ob = old_state ? obase : sbase; double* w_ = old_state ? ow : sw; double* dw_ = old_state ? odw : sdw;
  if (k == 0) ob[3 * e + l] = np;
  const double a1 = u - (double)(np - k), a2 = a1 - 1.0;
  w_[(3 * e + l) * 6 + k] = B6[(int)ceil(a1 / 6.0 * (double)G) - 1];
  double d = 0.0;
  if (a1 < 5.0) d = B5[(int)ceil(a1 / 5.0 * (double)G) - 1];
  if (0.0 < a2) d = d - B5[(int)ceil(a2 / 5.0 * (double)G) - 1];
  dw_[(3 * e + l) * 6 + k] = d;
}

// one wavefront per entry: remove the atom's principal stencil, add its new stencil (modify_Q_grid, src/pme.f90:275)
__global__ __launch_bounds__(256) void k_evb_patch(int n, const RecipAtom* __restrict__ ra, int K, size_t K3, const int* __restrict__ base,
                                                   const double* __restrict__ w, const int* __restrict__ sbase, const double* __restrict__ sw,
                                                   double2* __restrict__ grids) {
  const int lane = threadIdx.x & 63;
  const int e = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (e >= n) return;
  const RecipAtom r = ra[e];
  double2* g = grids + (size_t)r.grid * K3;
  for (int pass = 0; pass < 2; ++pass) {
    const double q = pass == 0 ? -r.q_old : r.q_new;
    if (fabs(q) <= 1e-6) continue;     // modify_Q_grid skips |chg| <= 1e-6 (src/pme.f90:296)
    const int* b = pass == 0 ? base + 3 * r.idx : sbase + 3 * e;
    const double* ww = pass == 0 ? w + 18 * (size_t)r.idx : sw + 18 * (size_t)e;
    for (int t = lane; t < 216; t += 64) {
      const int k1 = t % 6, k2 = (t / 6) % 6, k3 = t / 36;
      int n1 = b[0] - k1; if (n1 < 0) n1 += K;
      int n2 = b[1] - k2; if (n2 < 0) n2 += K;
      int n3 = b[2] - k3; if (n3 < 0) n3 += K;
      atomicAdd(&g[n1 + (size_t)K * (n2 + (size_t)K * n3)].x, q * ww[k1] * ww[6 + k2] * ww[12 + k3]);
    }
  }
}

// real parts of the ng diabat grids interleaved per grid point: pack[id*ng + g] (one contiguous read per stencil tap)
__global__ void k_evb_pack(size_t K3, int ng, const double2* __restrict__ grids, double* __restrict__ pack) {
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < K3 * ng; k += (size_t)gridDim.x * blockDim.x) {
    const size_t id = k / ng; const int g = (int)(k % ng);
    pack[k] = grids[(size_t)g * K3 + id].x;
  }
}

// special atoms: replace the principal-stencil contribution by the new-state one (update_reciprocal_space_force_dQ_dr :2103)
__global__ __launch_bounds__(256) void k_evb_recip_special(int n, const RecipAtom* __restrict__ ra, int K, size_t K3, const int* __restrict__ base,
                                                           const double* __restrict__ w, const double* __restrict__ dw, const int* __restrict__ sbase,
                                                           const double* __restrict__ sw, const double* __restrict__ sdw, const double2* __restrict__ grids,
                                                           double conv, double k00, double k01, double k02, double k10, double k11, double k12, double k20,
                                                           double k21, double k22, double* __restrict__ Fstore, size_t fstride) {
  const int lane = threadIdx.x & 63;
  const int e = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (e >= n) return;
  const RecipAtom r = ra[e];
  const double2* g = grids + (size_t)r.grid * K3;
  double tot[3] = {0, 0, 0};
  for (int pass = 0; pass < 2; ++pass) {
    const double q = pass == 0 ? -r.q_old : r.q_new;
    const int* b = pass == 0 ? base + 3 * r.idx : sbase + 3 * e;
    const double* ww = pass == 0 ? w + 18 * (size_t)r.idx : sw + 18 * (size_t)e;
    const double* dd = pass == 0 ? dw + 18 * (size_t)r.idx : sdw + 18 * (size_t)e;
    double f1 = 0, f2 = 0, f3 = 0;
    for (int t = lane; t < 216; t += 64) {
      const int k1 = t % 6, k2 = (t / 6) % 6, k3 = t / 36;
      int n1 = b[0] - k1; if (n1 < 0) n1 += K;
      int n2 = b[1] - k2; if (n2 < 0) n2 += K;
      int n3 = b[2] - k3; if (n3 < 0) n3 += K;
      const double th = g[n1 + (size_t)K * (n2 + (size_t)K * n3)].x;
      f1 += dd[k1] * ww[6 + k2] * ww[12 + k3] * th; f2 += dd[6 + k2] * ww[k1] * ww[12 + k3] * th; f3 += dd[12 + k3] * ww[k1] * ww[6 + k2] * th;
    }
    for (int o = 32; o > 0; o >>= 1) { f1 += __shfl_xor(f1, o, 64); f2 += __shfl_xor(f2, o, 64); f3 += __shfl_xor(f3, o, 64); }
    const double s = q * conv, Kd = (double)K;
    tot[0] += -Kd * (k00 * f1 + k10 * f2 + k20 * f3) * s; tot[1] += -Kd * (k01 * f1 + k11 * f2 + k21 * f3) * s; tot[2] += -Kd * (k02 * f1 + k12 * f2 + k22 * f3) * s;
  }
  if (lane == 0) { double* F = Fstore + (size_t)r.slot * fstride; for (int d = 0; d < 3; ++d) atomicAdd(&F[3 * r.idx + d], tot[d]); }      // beside k_evb_env / k_evb_special
}

// centre of mass of every molecule, stored per ATOM (coalesced reads in k_evb_coupling, which takes the minimum image per molecule)
__global__ void k_mol_com(int M, const int* __restrict__ mol_natom, const int* __restrict__ mol_atoms, const double* __restrict__ xyz, const double* __restrict__ mass,
                          double* __restrict__ com_atom) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  const int n = mol_natom[m];
  double com[3] = {0, 0, 0}, mt = 0;
  for (int k = 0; k < n; ++k) { const int a = mol_atoms[m * MAX_MOL_ATOMS + k]; const double ma = mass[a]; for (int d = 0; d < 3; ++d) com[d] += xyz[3 * a + d] * ma; mt += ma; }
  for (int d = 0; d < 3; ++d) com[d] /= mt;
  for (int k = 0; k < n; ++k) { const int a = mol_atoms[m * MAX_MOL_ATOMS + k]; for (int d = 0; d < 3; ++d) com_atom[3 * a + d] = com[d]; }
}

// ---- off-diagonal element -------------------------------------------------------------------------------------------------
// exchange-charge Coulomb sum over all NON-special molecules (evb_diabatic_coupling_electrostatics, src/ms_evb.f90:1276).
// The minimum image is taken per molecule, relative to the Zundel centre of mass, with the molecular centres of mass precomputed per atom by
// k_mol_com.  QL = 8 when the coupling has <= 8 Zundel atoms (water / hydronium), else 16.
// sum over the 16 lanes of each DPP row, left in every lane of the row (the pair kernel's row_sum16: four cross-lane moves inside the VALU, no
// ds_bpermute through the LDS pipe)
template <int CTRL>
__device__ __forceinline__ double evb_dpp_mov(double v) {
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double evb_row_sum16(double v) {
  v += evb_dpp_mov<0xB1>(v);       // quad_perm [1,0,3,2]
  v += evb_dpp_mov<0x4E>(v);       // quad_perm [2,3,0,1]
  v += evb_dpp_mov<0x141>(v);      // row_half_mirror
  v += evb_dpp_mov<0x140>(v);      // row_mirror
  return v;
}

// ONE PASS OVER THE ATOMS FOR ALL COUPLINGS of the step (round 4; rounds 2-3 walked the atoms once per coupling: D - 1 reads of the 60 B an
// atom brings, D - 1 descriptor copies per workgroup).  A thread holds CPA atoms in registers (position, centre of mass of the molecule,
// charge, molecule), then walks the couplings of its workgroup's group (<= CPG, their Zundel atoms staged in LDS once): per coupling the
// minimum image per molecule relative to the Zundel centre, the "special molecule" test, the Zundel atoms with the loop unrolled (the QL x 3
// Zundel-force accumulators are registers; a runtime bound sends them to scratch memory: 400 B per thread, 45 us), DPP row sums and one LDS
// add per 16-lane row and value.  The per-atom sums run over the Zundel atoms in the same order as before: same bits in the element stores.
constexpr int CPA = 4, CPG = 16;
struct CplStage { double zx[16][3], zq[16], zc[3]; int nz, n_mols, slot, mols[MAX_CHAIN + 1], zspec[16]; };
template <int QL>
__device__ __forceinline__ void evb_coupling_one(const CplStage& z, const double (&x)[CPA][3], const double (&cm)[CPA][3], const double (&qc)[CPA], const int (&m)[CPA],
                                                 const int (&ja)[CPA], int N, const BoxDev& box, double* __restrict__ F, double* __restrict__ red) {
  const int nz = z.nz;
  double fz[QL][3];
#pragma unroll
  for (int q = 0; q < QL; ++q) fz[q][0] = fz[q][1] = fz[q][2] = 0.0;
  double vex = 0.0;
#pragma unroll
  for (int a = 0; a < CPA; ++a) {
    const int j = ja[a];
    bool special = j >= N;
    for (int k = 0; k < z.n_mols; ++k) special |= (z.mols[k] == m[a]);
    double fx = 0.0, fy = 0.0, fz_ = 0.0;
    if (!special) {
      double xs[3];
      for (int d = 0; d < 3; ++d) xs[d] = x[a][d] - floor((cm[a][d] - z.zc[d]) * box.invL[d] + 0.5) * box.L[d];
      const double qj = qc[a];
#pragma unroll
      for (int q = 0; q < QL; ++q) {
        if (q < nz) {      // (uniform; no break: the loop must unroll completely)
          const double r0 = -(xs[0] - z.zx[q][0]), r1 = -(xs[1] - z.zx[q][1]), r2c = -(xs[2] - z.zx[q][2]);
          const double r2 = r0 * r0 + r1 * r1 + r2c * r2c;
          double r, inv_r;
          sqrt_rsqrt(r2, r, inv_r);
          const double qq = z.zq[q] * qj * inv_r;
          vex += qq;
          const double g = qq * inv_r * inv_r;      // = -(-qq / r^2)
          const double f0 = g * r0, f1 = g * r1, f2 = g * r2c;
          fx += f0; fy += f1; fz_ += f2;
          fz[q][0] -= f0; fz[q][1] -= f1; fz[q][2] -= f2;
        }
      }
    }
    if (j < N) { F[3 * (size_t)j] = fx; F[3 * (size_t)j + 1] = fy; F[3 * (size_t)j + 2] = fz_; }      // dVex of atom j (0 for special atoms: k_evb_coupling_finish assigns them)
  }
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int q = 0; q < QL; ++q)
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const double v = evb_row_sum16(fz[q][d]);
      if ((lane & 15) == 0 && q < nz) atomicAdd(&red[3 * q + d], v);
    }
  {
    const double v = evb_row_sum16(vex);
    if ((lane & 15) == 0) atomicAdd(&red[48], v);
  }
}
// gridDim.x workgroups share the atoms (256 CPA per trip), gridDim.y = groups of cpg couplings (one group = one pass over the atoms: 96 workgroups of a
// 98 k-atom box with all seven couplings each were 77 us alone - one wave per SIMD walking 12 k dependent instructions; four groups: see DESIGN.md 4.6); part[ic][blockIdx.x] = the workgroup's share of
// Vex, Zundel forces into copy blockIdx.x % CPL_BINS of dVspec (k_evb_coupling_finish sums both)
__global__ __launch_bounds__(256) void k_evb_coupling(int N, int n_cp, const CouplingDesc* __restrict__ cps, const int* __restrict__ mol_of,
                                                      const double* __restrict__ com_atom, const double* __restrict__ xyz, const double* __restrict__ chg,
                                                      const double* __restrict__ exch_atomic,
                                                      BoxDev box, double conv, double* __restrict__ Foff, size_t fstride, double* __restrict__ part,
                                                      int part_stride, double* __restrict__ dVspec, int cpg) {
  __shared__ CplStage zs[CPG];
  __shared__ double red[CPG][49];
  const int g0 = blockIdx.y * cpg, ng = min(cpg, n_cp - g0);      // cpg <= CPG couplings per workgroup (host: as many groups as it takes to fill the GPU with ~500 workgroups)
  {
    const int l = threadIdx.x >> 4, q = threadIdx.x & 15;      // 16 threads per coupling
    if (l < ng) {
      const CouplingDesc& cp = cps[g0 + l];
      CplStage& z = zs[l];
      const int nz = cp.nD + cp.nA;
      if (q < nz) {
        const bool acc = q >= cp.nD;
        const int sp = acc ? cp.A_at[q - cp.nD] : cp.D_at[q];
        const SpecAtom& a = cp.spec[sp];
        const double* shm = acc ? cp.shifta : cp.shiftd;
        for (int d = 0; d < 3; ++d) z.zx[q][d] = cp.zc[d] + (a.x[d] - cp.zc[d] - shm[d]);
        z.zq[q] = (acc && q == nz - 1) ? cp.q_transfer : exch_atomic[a.type];
        z.zspec[q] = sp;
      } else { z.zx[q][0] = z.zx[q][1] = z.zx[q][2] = 0.0; z.zq[q] = 0.0; z.zspec[q] = 0; }      // a zero charge: the padded q contribute exact zeros
      if (q < 3) z.zc[q] = cp.zc[q];
      if (q <= MAX_CHAIN) z.mols[q] = cp.mols[q];
      if (q == 0) { z.nz = nz; z.n_mols = cp.n_mols; z.slot = cp.slot; }
    }
    for (int k = threadIdx.x; k < CPG * 49; k += blockDim.x) (&red[0][0])[k] = 0.0;
  }
  __syncthreads();
  for (int base = blockIdx.x * (256 * CPA); base < N; base += gridDim.x * (256 * CPA)) {
    double x[CPA][3], cm[CPA][3], qc[CPA];
    int m[CPA], ja[CPA];
#pragma unroll
    for (int a = 0; a < CPA; ++a) {
      const int j = base + a * 256 + (int)threadIdx.x, jc = min(j, N - 1);
      ja[a] = j; m[a] = mol_of[jc]; qc[a] = chg[jc] * conv;
      for (int d = 0; d < 3; ++d) { x[a][d] = xyz[3 * (size_t)jc + d]; cm[a][d] = com_atom[3 * (size_t)jc + d]; }
    }
    for (int l = 0; l < ng; ++l) {
      const CplStage& z = zs[l];
      double* F = Foff + (size_t)z.slot * fstride;
      if (z.nz > 8) evb_coupling_one<16>(z, x, cm, qc, m, ja, N, box, F, red[l]);
      else evb_coupling_one<8>(z, x, cm, qc, m, ja, N, box, F, red[l]);
    }
  }
  __syncthreads();
  for (int l = threadIdx.x; l < ng; l += blockDim.x) part[(size_t)(g0 + l) * part_stride + blockIdx.x] = red[l][48];
  for (int k = threadIdx.x; k < ng * 48; k += blockDim.x) {
    const int l = k / 48, r = k - l * 48;
    if (r >= zs[l].nz * 3) continue;
    const double v = red[l][r];
    if (v != 0.0) atomicAdd(&dVspec[((size_t)(g0 + l) * CPL_BINS + (blockIdx.x % CPL_BINS)) * MAX_SPEC * 3 + 3 * zs[l].zspec[r / 3] + r % 3], v);
  }
}

// evb_diabatic_coupling_function (src/ms_evb.f90:1180-1266)
__device__ __forceinline__ void coupling_function(int ftype, const double* p, const double* q, const double* rOO, double& A, double& Vc, double dA[3][3]) {
  const double rm = sqrt(rOO[0] * rOO[0] + rOO[1] * rOO[1] + rOO[2] * rOO[2]), q2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2], qm = sqrt(q2);
  Vc = p[0];
  if (ftype == 1) {
    const double gamma = p[1], P = p[2], k = p[3], D = p[4], beta = p[5], R0 = p[6], Pp = p[7], alpha = p[8], rl0 = p[9];
    const double f1 = exp(-gamma * q2), f2 = 1.0 + P * exp(-k * (rm - D) * (rm - D)), f3 = 0.5 * (1.0 - tanh(beta * (rm - R0))) + Pp * exp(-alpha * (rm - rl0));
    const double ch = cosh(beta * (rm - R0));
    const double d1 = -gamma * 2.0 * qm * f1, d2 = P * -k * 2.0 * (rm - D) * exp(-k * (rm - D) * (rm - D)), d3 = -0.5 * beta / (ch * ch) - Pp * alpha * exp(-alpha * (rm - rl0));
    A = f1 * f2 * f3;
    for (int d = 0; d < 3; ++d) {
      const double tq = d1 * f2 * f3 * q[d] / qm, tr = (f1 * d2 * f3 + f1 * f2 * d3) * rOO[d] / rm;
      dA[0][d] = 0.5 * tq + tr; dA[1][d] = 0.5 * tq - tr; dA[2][d] = -tq;
    }
  } else {
    const double gamma = p[1], k = p[2], D = p[3];
    const double f1 = exp(-gamma * q2), f2 = exp(-k * (rm - D) * (rm - D));
    const double d1 = -gamma * 2.0 * qm * f1, d2 = -k * 2.0 * (rm - D) * f2;
    A = f1 * f2;
    for (int d = 0; d < 3; ++d) { const double tq = d1 * f2 * q[d] / qm, tr = f1 * d2 * rOO[d] / rm; dA[0][d] = 0.5 * tq + tr; dA[1][d] = 0.5 * tq - tr; dA[2][d] = -tq; }
  }
}

// one wavefront per coupling: sum of the coupling kernel's partial sums (its row of `part`: n_big couplings with nb_big entries first, the
// rest with nb_small), special-molecule part of Vex (molecules of the chain that are neither donor nor acceptor: lanes over the
// (special atom, Zundel atom) pairs), geometric factor, element energy; writes {A, Vc+Vex, H_pd} and the forces of the special atoms.
// (The serial version - one thread per coupling - was 33 us of the chain the host's turn waits for.)
__global__ __launch_bounds__(64) void k_evb_coupling_finish(int n_cp, const CouplingDesc* __restrict__ cps, const double* __restrict__ exch_atomic, const double* __restrict__ mass,
                                      BoxDev box, double conv, const double* __restrict__ part, int part_stride, int n_big, int nb_big, int nb_small, double* __restrict__ Vex_env,
                                      double* __restrict__ dVspec, double* __restrict__ Foff, size_t fstride, double* __restrict__ out /*[n_cp][3]*/) {
  const int ic = blockIdx.x;
  if (ic >= n_cp) return;
  __shared__ CouplingDesc cp;      // staged once: the code below walks the descriptor with dependent reads
  {
    const int* src = (const int*)&cps[ic]; int* dst = (int*)&cp;
    for (int k = threadIdx.x; k < (int)(sizeof(CouplingDesc) / 4); k += blockDim.x) dst[k] = src[k];
  }
  __shared__ double dV[MAX_SPEC][3];
  __shared__ bool inDA[MAX_SPEC], other[MAX_SPEC];
  __shared__ double shm_mol[MAX_CHAIN + 1][3];
  __shared__ double zx[2 * MAX_MOL_ATOMS][3], zq[2 * MAX_MOL_ATOMS];
  __shared__ int zs[2 * MAX_MOL_ATOMS];
  double vex = 0.0;
  {      // environment part of Vex: this coupling's row of partial sums
    const int nb = ic < n_big ? nb_big : nb_small;
    for (int k = threadIdx.x; k < nb; k += 64) vex += part[(size_t)ic * part_stride + k];
  }
  __syncthreads();
  for (int k = threadIdx.x; k < cp.n_spec * 3; k += blockDim.x) {      // sum of the CPL_BINS copies the coupling kernels added into, which are left zeroed
    double* p = dVspec + (size_t)ic * CPL_BINS * MAX_SPEC * 3 + k;
    double v[CPL_BINS];
#pragma unroll
    for (int b = 0; b < CPL_BINS; ++b) { v[b] = p[(size_t)b * MAX_SPEC * 3]; p[(size_t)b * MAX_SPEC * 3] = 0.0; }
    double sum = 0.0;
#pragma unroll
    for (int b = 0; b < CPL_BINS; ++b) sum += v[b];
    (&dV[0][0])[k] = sum;
  }
  for (int k = threadIdx.x; k < cp.n_spec; k += 64) inDA[k] = false;
  __syncthreads();
  const int nz = cp.nD + cp.nA;
  if ((int)threadIdx.x < nz) {      // Zundel atoms: image next to the Zundel centre, exchange charge
    const int q = threadIdx.x;
    const bool acc = q >= cp.nD;
    const int sp = acc ? cp.A_at[q - cp.nD] : cp.D_at[q];
    const double* shm = acc ? cp.shifta : cp.shiftd;
    for (int d = 0; d < 3; ++d) zx[q][d] = cp.zc[d] + (cp.spec[sp].x[d] - cp.zc[d] - shm[d]);
    zq[q] = (acc && q == nz - 1) ? cp.q_transfer : exch_atomic[cp.spec[sp].type];
    zs[q] = sp;
    inDA[sp] = true;
  }
  if ((int)threadIdx.x < cp.n_mols) for (int d = 0; d < 3; ++d) shm_mol[threadIdx.x][d] = floor((cp.mol_com[threadIdx.x][d] - cp.zc[d]) * box.invL[d] + 0.5) * box.L[d];
  __syncthreads();
  for (int b = threadIdx.x; b < cp.n_spec; b += 64) {      // atoms of special molecules other than donor / acceptor
    bool isDA = false;
    for (int k = 0; k < cp.n_spec; ++k) if (cp.spec[k].mol == cp.spec[b].mol) isDA |= inDA[k];
    other[b] = !isDA;
  }
  __syncthreads();
  for (int it = threadIdx.x; it < cp.n_spec * nz; it += 64) {
    const int b = it / nz, q = it - b * nz;
    if (!other[b]) continue;
    int im = 0;
    for (int k = 0; k < cp.n_mols; ++k) if (cp.mols[k] == cp.spec[b].mol) im = k;
    const double qi = zq[q], qj = cp.spec[b].q;
    double rij[3];
    for (int d = 0; d < 3; ++d) rij[d] = -(cp.spec[b].x[d] - zx[q][d] - shm_mol[im][d]);
    const double r2 = rij[0] * rij[0] + rij[1] * rij[1] + rij[2] * rij[2], r = sqrt(r2);
    vex += qi * qj / r * conv;
    const double g = -qi * qj / (r2 * r) * conv;
    for (int d = 0; d < 3; ++d) { atomicAdd(&dV[zs[q]][d], g * rij[d]); atomicAdd(&dV[b][d], -g * rij[d]); }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) vex += __shfl_xor(vex, o, 64);
  __syncthreads();
  if (threadIdx.x == 0) {
    Vex_env[ic] = vex;
    // geometric factor (evb_diabatic_coupling_geometric, src/ms_evb.f90:1117-1174)
    const SpecAtom &O1 = cp.spec[cp.gO1], &O2 = cp.spec[cp.gO2], &H = cp.spec[cp.gH];
    double sh[3], rO2[3], rH[3], rOO[3], q[3];
    for (int d = 0; d < 3; ++d) {
      sh[d] = floor((O2.x[d] - O1.x[d]) * box.invL[d] + 0.5) * box.L[d];
      rO2[d] = O1.x[d] + (O2.x[d] - O1.x[d] - sh[d]); rH[d] = O1.x[d] + (H.x[d] - O1.x[d] - sh[d]);
      rOO[d] = O1.x[d] - rO2[d]; q[d] = (O1.x[d] + rO2[d]) / 2.0 - rH[d];
    }
    double A, Vc, dA[3][3];
    coupling_function(cp.ftype, cp.par, q, rOO, A, Vc, dA);
    const double pre = Vc + vex;
    for (int d = 0; d < 3; ++d) { dV[cp.gO1][d] += pre * dA[0][d] / A; dV[cp.gO2][d] += pre * dA[1][d] / A; dV[cp.gH][d] += pre * dA[2][d] / A; }
    out[3 * ic] = A; out[3 * ic + 1] = pre; out[3 * ic + 2] = pre * A;
  }
  __syncthreads();
  double* F = Foff + (size_t)cp.slot * fstride;
  for (int k = threadIdx.x; k < cp.n_spec * 3; k += 64) F[3 * cp.spec[k / 3].idx + k % 3] = dV[k / 3][k % 3];   // still "dV": the factor -A goes into the blend coefficient
}

// Hellmann-Feynman blend: F = sum_d c_d^2 F_dd + 2 sum_(p,d) c_p c_d F_pd (src/ms_evb.f90:293-309).
// principal slot = repulsion delta of the principal diabat (consumes and re-zeroes the delta store); keeps a copy of the principal
// force for msevb_get_evb_element_force (the blend overwrites it in place)
__global__ void k_evb_slot0(size_t n3, const double* __restrict__ F11, double* __restrict__ delta0, double* __restrict__ Fdiag0, double* __restrict__ F11_copy) {
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n3; k += (size_t)gridDim.x * blockDim.x) { Fdiag0[k] = delta0[k]; delta0[k] = 0.0; F11_copy[k] = F11[k]; }
}

// Fast path (no per-element force stores): the stencil gather is LINEAR in the potential, so sum_d c_d^2 (recip force of diabat d) is the
// gather of the BLENDED potential  c_0^2 theta_1 + sum_{d>=1} c_d^2 theta_d  - one gather over one grid instead of one per diabat.
struct GridCoefs { int n; double c0; double c[MAX_DIABATS]; };
// dc != nullptr: the coefficients {c0, c[0..n)} come from device memory (written by k_evb_solve), gc only carries n
__global__ void k_evb_blend_grids(size_t K3, GridCoefs gc, const double* __restrict__ dc, const double* __restrict__ theta1, const double2* __restrict__ grids, double* __restrict__ out) {
  const double c0 = dc ? dc[0] : gc.c0;      // (uniform scalar loads either way; writing into the by-value struct would move it to scratch)
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < K3; k += (size_t)gridDim.x * blockDim.x) {
    double s = c0 * theta1[k];
    for (int g = 0; g < gc.n; ++g) s += (dc ? dc[1 + g] : gc.c[g]) * grids[(size_t)g * K3 + k].x;
    out[k] = s;
  }
}

// ---- reciprocal space of the non-principal diabats WITHOUT a Fourier transform per diabat (default fast path) -----------------------
// The reference transforms every diabat's charge grid (calculate_reciprocal_space_pme, src/ms_evb.f90:1962-2090).  Q_d differs from the
// principal grid Q_1 by the stencils of a few dozen atoms (DQ_d = -q_old S_old + q_new S_new over the atoms of the molecules a hop touched),
// and theta = IFFT(CB FFT(Q)) is LINEAR in Q:  theta_d = theta_1 + G * DQ_d  with the real-space kernel G = IFFT(CB) (K^3 table, built
// with the box).  Hence
//   E_recip,d - E_recip,1 = conv (DQ_d . theta_1 + 1/2 DQ_d . G * DQ_d)        linear term: stencil dots with theta_1 (k_evb_recip_linear)
//                                                                             quadratic term: stencil-pair sums over G (k_evb_recip_pairs)
//   blended potential     = theta_1 + sum_d c_d^2 G * DQ_d = IFFT(CB FFT(Q_1 + sum_d c_d^2 DQ_d))      ONE transform pair, after the Jacobi
//   special atoms' forces in diabat d: their stencils against theta_1 + G * DQ_d               the same two kernels
// instead of D - 1 transform pairs of dense grids (7 x 4 MB x 4 passes per step at 98 k atoms).  A stencil pair (T, S) interacts through
//   sum_{i in T, j in S} wT_i wS_j G(x_T,i - x_S,j) = sum_{o in [-5,5]^3} W(o) G(b_T - b_S - o),  W = separable cross-correlation of the
// two stencils' tap weights (11 values per axis): 1331 table reads per pair.

// two stencils per recip entry: 2 e = the atom as the principal diabat has it (charge -q_old), 2 e + 1 = its new state (charge +q_new).
// Charges modify_Q_grid would skip (|q| <= 1e-6, src/pme.f90:296) and entries whose two states are identical get q = 0.
// one wavefront per special atom (round 5; it was one THREAD per atom - 72 dependent loads and as many stores, 17 us for two workgroups on the chain whose end the energy
// fetch follows): lane k < 18 moves tap k of the four weight arrays, lanes 0-2 the base points, one ballot decides whether the two states differ at all
__global__ __launch_bounds__(64) void k_evb_stencils(int n, const RecipAtom* __restrict__ ra, const int* __restrict__ obase, const double* __restrict__ ow, const double* __restrict__ odw,
                               const int* __restrict__ sbase, const double* __restrict__ sw, const double* __restrict__ sdw, Stencil* __restrict__ st) {
  const int e = blockIdx.x, l = threadIdx.x;
  if (e >= n) return;
  const double q_old = ra[e].q_old, q_new = ra[e].q_new;
  bool differs = false;
  Stencil& a = st[2 * e]; Stencil& b = st[2 * e + 1];
  if (l < 18) {
    const double aw = ow[18 * (size_t)e + l], adw = odw[18 * (size_t)e + l], bw = sw[18 * (size_t)e + l], bdw = sdw[18 * (size_t)e + l];
    a.w[l] = aw; a.dw[l] = adw; b.w[l] = bw; b.dw[l] = bdw;
    differs = aw != bw;
  } else if (l < 21) {
    const int d = l - 18, ab = obase[3 * e + d], bb = sbase[3 * e + d];
    a.base[d] = ab; b.base[d] = bb;
    differs = ab != bb;
  }
  const bool same = q_old == q_new && __ballot(differs) == 0ull;
  if (l == 21) {
    const int grid = ra[e].grid, slot = ra[e].slot, idx = ra[e].idx;
    a.grid = grid; b.grid = grid; a.slot = slot; b.slot = slot; a.idx = idx; b.idx = idx;
    a.q = (same || fabs(q_old) <= 1e-6) ? 0.0 : -q_old;
    b.q = (same || fabs(q_new) <= 1e-6) ? 0.0 : q_new;
  }
}

// one wavefront per stencil: its dot with theta_1 (energy, linear term) and the force theta_1 exerts on it -> the diabat's slot of the delta store
__global__ __launch_bounds__(256) void k_evb_recip_linear(int nst, const Stencil* __restrict__ st, int K, const double* __restrict__ theta, double conv, double k00, double k01,
                                                          double k02, double k10, double k11, double k12, double k20, double k21, double k22, double* __restrict__ Fstore,
                                                          size_t fstride, double* __restrict__ E_lin) {
  const int lane = threadIdx.x & 63;
  const int e = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (e >= nst) return;
  const Stencil& t = st[e];
  const double q = t.q;
  if (q == 0.0) return;
  double f1 = 0, f2 = 0, f3 = 0, en = 0;
  for (int p = lane; p < 216; p += 64) {
    const int k1 = p % 6, k2 = (p / 6) % 6, k3 = p / 36;
    int n1 = t.base[0] - k1; if (n1 < 0) n1 += K;
    int n2 = t.base[1] - k2; if (n2 < 0) n2 += K;
    int n3 = t.base[2] - k3; if (n3 < 0) n3 += K;
    const double th = theta[n1 + (size_t)K * (n2 + (size_t)K * n3)];
    en += t.w[k1] * t.w[6 + k2] * t.w[12 + k3] * th;
    f1 += t.dw[k1] * t.w[6 + k2] * t.w[12 + k3] * th; f2 += t.dw[6 + k2] * t.w[k1] * t.w[12 + k3] * th; f3 += t.dw[12 + k3] * t.w[k1] * t.w[6 + k2] * th;
  }
  for (int o = 32; o > 0; o >>= 1) { f1 += __shfl_xor(f1, o, 64); f2 += __shfl_xor(f2, o, 64); f3 += __shfl_xor(f3, o, 64); en += __shfl_xor(en, o, 64); }
  if (lane == 0) {
    const double s = q * conv, Kd = (double)K;
    double* F = Fstore + (size_t)t.slot * fstride;
    atomicAdd(&F[3 * t.idx], -Kd * (k00 * f1 + k10 * f2 + k20 * f3) * s);
    atomicAdd(&F[3 * t.idx + 1], -Kd * (k01 * f1 + k11 * f2 + k21 * f3) * s);
    atomicAdd(&F[3 * t.idx + 2], -Kd * (k02 * f1 + k12 * f2 + k22 * f3) * s);
    atomicAdd(&E_lin[t.grid], q * en);
  }
}

// One WORKGROUP per target stencil T (blockIdx.x) of a diabat (blockIdx.y); its four waves take the source stencils S of the same diabat in
// turn: the potential S raises at T's taps through the real-space kernel G = IFFT(CB) -> energy q_T q_S sum W G (halved when the grids'
// energies are finished: every unordered pair appears twice, T = S once with its own factor 1/2) and the force on T's atom, summed over S
// in registers and added ONCE per target (one atomic set per pair serialised in the L2 on a handful of addresses: 40 us for 6 300 pairs).
struct GridRanges { int start[MAX_DIABATS + 1]; };
__global__ __launch_bounds__(256) void k_evb_recip_pairs(GridRanges gr, const Stencil* __restrict__ st, int K, const double* __restrict__ green, double conv, double k00,
                                                         double k01, double k02, double k10, double k11, double k12, double k20, double k21, double k22,
                                                         double* __restrict__ Fstore, size_t fstride, double* __restrict__ E_quad) {
  __shared__ double sW[4][3][2][11];      // per wave: [axis][W, dW][o + 5]
  __shared__ double sT[36];               // T.w[18], T.dw[18]
  __shared__ double sS[4][18];            // per wave: S.w[18]
  __shared__ double red[4][4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int g = blockIdx.y, s0 = gr.start[g], ns = gr.start[g + 1] - s0;
  if ((int)blockIdx.x >= ns) return;
  const Stencil& T = st[s0 + blockIdx.x];
  const double qT = T.q;
  if (qT == 0.0) return;      // (block-uniform)
  if (threadIdx.x < 36) sT[threadIdx.x] = threadIdx.x < 18 ? T.w[threadIdx.x] : T.dw[threadIdx.x - 18];
  __syncthreads();
  double en = 0, f1 = 0, f2 = 0, f3 = 0;
  for (int is = wv; is < ns; is += 4) {
    const Stencil& S = st[s0 + is];
    const double qS = S.q;
    if (qS == 0.0) continue;      // (wave-uniform)
    if (lane < 18) sS[wv][lane] = S.w[lane];
    __builtin_amdgcn_wave_barrier();
    if (lane < 33) {      // cross-correlations of the tap weights: W(o) = sum_{i - j = o} wT[i] wS[j], dW with T's derivative weights
      const int ax = lane / 11, o = lane % 11 - 5;
      double a = 0.0, b = 0.0;
      for (int i = 0; i < 6; ++i) { const int j = i - o; if (j >= 0 && j < 6) { const double ws = sS[wv][6 * ax + j]; a += sT[6 * ax + i] * ws; b += sT[18 + 6 * ax + i] * ws; } }
      sW[wv][ax][0][o + 5] = a; sW[wv][ax][1][o + 5] = b;
    }
    __builtin_amdgcn_wave_barrier();
    // tap i of T sits at b_T - i, tap j of S at b_S - j: separation (b_T - b_S) - (i - j)
    const int d1 = T.base[0] - S.base[0], d2 = T.base[1] - S.base[1], d3 = T.base[2] - S.base[2];
    // offsets (o1, o2) on the lanes (121 = 64 + 57: two chunks), o3 in the loop: the lane-dependent index arithmetic is done once per chunk,
    // the plane offset is wave-uniform, and the 11 table reads of a chunk are independent (issued together)
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) {
      const int idx = lane + 64 * ch;
      const bool live = idx < 121;
      const int o1 = live ? idx % 11 : 0, o2 = live ? idx / 11 : 0;
      // |d| <= K - 1 (two stencils on either side of the periodic boundary), |o| <= 5: d - o lies in [-(K + 4), K + 4], which takes TWO wraps to
      // bring into [0, K) (K >= 6).  Round 3 wrapped once: a pair of special atoms straddling the box boundary then read green[-4..-1 + ...] - a wrong
      // table element for most (n2, n3), and for n2 = n3 = 0 the bytes in front of the buffer (the page fault of HISTORY.md D, found with MSEVB_GUARD=1)
      int n1 = d1 - (o1 - 5); n1 += n1 < 0 ? K : 0; n1 += n1 < 0 ? K : 0; n1 -= n1 >= K ? K : 0; n1 -= n1 >= K ? K : 0;
      int n2 = d2 - (o2 - 5); n2 += n2 < 0 ? K : 0; n2 += n2 < 0 ? K : 0; n2 -= n2 >= K ? K : 0; n2 -= n2 >= K ? K : 0;
      const double* row = green + n1 + (size_t)K * n2;
      const double w1 = sW[wv][0][0][o1], w2 = sW[wv][1][0][o2];
      const double cc = live ? qS * w1 * w2 : 0.0, ca = live ? qS * sW[wv][0][1][o1] * w2 : 0.0, cb = live ? qS * w1 * sW[wv][1][1][o2] : 0.0;
      double Gv[11];
#pragma unroll
      for (int o3 = 0; o3 < 11; ++o3) {
        int n3 = d3 - (o3 - 5); n3 += n3 < 0 ? K : 0; n3 += n3 < 0 ? K : 0; n3 -= n3 >= K ? K : 0; n3 -= n3 >= K ? K : 0;
        Gv[o3] = row[(size_t)K * K * n3];
      }
#pragma unroll
      for (int o3 = 0; o3 < 11; ++o3) {
        const double g3 = Gv[o3] * sW[wv][2][0][o3], gd = Gv[o3] * sW[wv][2][1][o3];
        en += cc * g3; f1 += ca * g3; f2 += cb * g3; f3 += cc * gd;
      }
    }
    __builtin_amdgcn_wave_barrier();      // sW / sS are rewritten by the next source stencil
  }
  for (int o = 32; o > 0; o >>= 1) { f1 += __shfl_xor(f1, o, 64); f2 += __shfl_xor(f2, o, 64); f3 += __shfl_xor(f3, o, 64); en += __shfl_xor(en, o, 64); }
  if (lane == 0) { red[wv][0] = en; red[wv][1] = f1; red[wv][2] = f2; red[wv][3] = f3; }
  __syncthreads();
  if (threadIdx.x == 0) {
    en = red[0][0] + red[1][0] + red[2][0] + red[3][0]; f1 = red[0][1] + red[1][1] + red[2][1] + red[3][1];
    f2 = red[0][2] + red[1][2] + red[2][2] + red[3][2]; f3 = red[0][3] + red[1][3] + red[2][3] + red[3][3];
    const double s = qT * conv, Kd = (double)K;
    double* F = Fstore + (size_t)T.slot * fstride;
    atomicAdd(&F[3 * T.idx], -Kd * (k00 * f1 + k10 * f2 + k20 * f3) * s);
    atomicAdd(&F[3 * T.idx + 1], -Kd * (k01 * f1 + k11 * f2 + k21 * f3) * s);
    atomicAdd(&F[3 * T.idx + 2], -Kd * (k02 * f1 + k12 * f2 + k22 * f3) * s);
    atomicAdd(&E_quad[g], qT * en);
  }
}

// blended charge grid: grid += sum over the stencils of c_d^2 q S (the grid holds Q_1 on rank 0, zero on the others); coefficients as kernel
// arguments, or from device memory behind k_evb_solve
// first kernel of a chain that starts on a mark of another stream (2 s limit, then the sticky fault flag, as k_evb_publish)
__device__ __forceinline__ void evb_wait_mark(const long long* mark, long long seq, double* fault, const long long* sb = nullptr) {      // sb: a captured launch - the number is the step block's SB_A
  if (threadIdx.x == 0) {
    if (sb) seq = sb[SB_A];
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(mark, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != seq) {      // (relaxed polls: an acquire per poll is a cache invalidation per poll, from every waiting workgroup)
      if (wall_clock64() - t0 > 200000000LL) { if (fault) *fault = 2.0; break; }
      __builtin_amdgcn_s_sleep(8);
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
  }
  __syncthreads();
}
__global__ void k_evb_wait_mark(const long long* mark, long long seq, double* fault, const long long* sb = nullptr) { evb_wait_mark(mark, seq, fault, sb); }
// wait_mark: the coefficients in dc come from k_evb_solve on ANOTHER stream (the grid stream, right behind the energy fetch): this kernel, first of the
// blend phase on the main stream, starts on that kernel's mark instead of behind a cross-stream event (~10 us) or behind the solve itself
__global__ __launch_bounds__(256) void k_evb_blend_patch(int nst, const Stencil* __restrict__ st, GridCoefs gc, const double* __restrict__ dc, int K, double* __restrict__ grid, int gs,
                                                         const long long* wait_mark = nullptr, long long wait_seq = 0, double* fault = nullptr, const long long* wait_sb = nullptr) {
  if (wait_mark) evb_wait_mark(wait_mark, wait_seq, fault, wait_sb);
  const int lane = threadIdx.x & 63;
  const int e = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (e >= nst) return;
  const Stencil& t = st[e];
  const double q = t.q * (dc ? dc[1 + t.grid] : gc.c[t.grid]);
  if (q == 0.0) return;
  for (int p = lane; p < 216; p += 64) {
    const int k1 = p % 6, k2 = (p / 6) % 6, k3 = p / 36;
    int n1 = t.base[0] - k1; if (n1 < 0) n1 += K;
    int n2 = t.base[1] - k2; if (n2 < 0) n2 += K;
    int n3 = t.base[2] - k3; if (n3 < 0) n3 += K;
    atomicAdd(&grid[gs * (n1 + (size_t)K * (n2 + (size_t)K * n3))], q * t.w[k1] * t.w[6 + k2] * t.w[12 + k3]);      // gs = 2: real parts of a complex grid
  }
}
__global__ void k_grid_real(size_t K3, const double2* __restrict__ grid, double* __restrict__ out) {
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < K3; k += (size_t)gridDim.x * blockDim.x) out[k] = grid[k].x;
}

__global__ void k_copy_grid(size_t n, const double* __restrict__ src, double* __restrict__ dst) {
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (size_t)gridDim.x * blockDim.x) dst[k] = src[k];
}
struct BlendTerms { int n; int slot[2 * MAX_DIABATS]; double coef[2 * MAX_DIABATS]; };   // slot < 0: off-diagonal store, index -slot-1 (kernel argument, no upload)
// The diagonal store holds differences to the principal force F11 (F_dd = F11 + store_d) and sum_d c_d^2 = 1, so
// F = w11 F11 + sum_d c_d^2 store_d + 2 sum c_p c_d F_pd with w11 = 1 (w11 < 1 only for a rank's PARTIAL blend, see evb_partial terms);
// F holds F11 on entry and the blended force on exit.
// zero_diag: Fdiag is the self-cleaning delta store itself (fast path): every diagonal slot read is reset for the next step.
// Fs != nullptr: the pair loop's sorted-order force accumulator (k_pair_half) has not been folded into F yet - the principal force of atom i is
// F[3 i + d] + Fs[d N + inv_perm[i]]; the accumulator is consumed (re-zeroed) here (k_pair_fold, a launch between the pair loop and the blend
// phase on the step's critical path, is skipped by the fast path)
__global__ void k_evb_blend(size_t n3, BlendTerms bt, const double* __restrict__ dc, double w11, double* __restrict__ Fdiag, const double* __restrict__ Foff, size_t fstride, double* __restrict__ F, int zero_diag,
                            double* __restrict__ Fs = nullptr, const int* __restrict__ inv_perm = nullptr, int N = 0) {
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n3; k += (size_t)gridDim.x * blockDim.x) {
    double f11 = F[k];
    if (Fs) { const size_t i = k / 3, a = (k - 3 * i) * (size_t)N + inv_perm[i]; f11 += Fs[a]; Fs[a] = 0.0; }
    double s = w11 * f11;
    for (int t = 0; t < bt.n; ++t) {
      const int sl = bt.slot[t];
      const double cf = dc ? dc[t] : bt.coef[t];      // device-resident coefficients (k_evb_solve) or kernel arguments
      if (sl < 0) s += cf * Foff[(size_t)(-sl - 1) * fstride + k];
      else { double* p = Fdiag + (size_t)sl * fstride + k; s += cf * *p; if (zero_diag) *p = 0.0; }
    }
    F[k] = s;
  }
}

// The same sum WITHOUT the force array's own value: F += sum of coefficient x store, added with atomics, so that it can run on a side stream while the
// main stream transforms and gathers the blended potential (whose kernel also folds the pair loop's accumulator in, pme.hip) - every contribution
// to the force array of an MS-EVB step is then an atomic add, their order does not matter, and no pass over the array is left behind the gather.
__global__ void k_evb_blend_terms(size_t n3, BlendTerms bt, const double* __restrict__ dc, double* __restrict__ Fdiag, const double* __restrict__ Foff, size_t fstride, double* __restrict__ F) {
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n3; k += (size_t)gridDim.x * blockDim.x) {
    double s = 0.0;
    for (int t = 0; t < bt.n; ++t) {
      const int sl = bt.slot[t];
      const double cf = dc ? dc[t] : bt.coef[t];
      if (sl < 0) s += cf * Foff[(size_t)(-sl - 1) * fstride + k];
      else { double* p = Fdiag + (size_t)sl * fstride + k; s += cf * *p; *p = 0.0; }      // (the delta store is consumed)
    }
    if (s != 0.0) atomicAdd(&F[k], s);
  }
}

// ---- Hamiltonian assembly + Jacobi + blend coefficients ON THE DEVICE (single rank, fast path): the blend phase is enqueued behind this
// kernel on the main stream without a host round trip (fetch -> host Jacobi -> launch cost 27-35 us of the step's critical path, and kept
// the host from enqueueing ahead).  diagonalize_evb_hamiltonian + jacobi (src/ms_evb.f90:242-330, src/general_routines.f90:2013-2088): the
// reference's cyclic Jacobi visits the n(n-1)/2 element pairs one after the other (~200 dependent rotations at D = 8: 108 us on one
// wavefront, round 2).  Here the SAME rotations (same angle formula, same "negligible element" rule, same stop criterion: every
// off-diagonal element exactly zero) are applied in the round-robin order of a tournament: n/2 disjoint pairs per round, n - 1 rounds per
// sweep, all pairs of a round at once.  Jacobi converges for either ordering to the same eigenvectors; the results differ from the host's
// row-cyclic sweep in the last bits only (test_device_side_hamiltonian_solve: 1e-13), and the sign of the vector is immaterial (the blend
// uses products c_i c_j).  H' assembly as in evb_force_energy_body (host).
struct SolveDesc {
  int D, nu, ncp, ngrid, nterm, unit00;
  double e_hyd, Hp_norep, once;
  int parent[MAX_DIABATS], grid_of[MAX_DIABATS], cp_of[MAX_DIABATS], unit_off[MAX_DIABATS + 1];
  double eref_delta[MAX_DIABATS];
  int term_i[2 * MAX_DIABATS], term_j[2 * MAX_DIABATS];      // blend terms in blend_terms() order (i == j: diagonal)
  // followed in memory by: int unit_idx[unit_off[D]]; double unit_sign[unit_off[D]] (8-byte aligned)
};
// out_host: {status, lowest eigenvalue of H', eigvec[D], cpl_A[D], H'[D*D] row-major (upper triangle filled)}, then done_flag = done_seq (pinned memory, the host spins on it)
// wait_mark: device-side mark of the kernel that completes the energies (k_evb_publish on the grid stream) - this kernel sits on the MAIN stream
// behind the pair loop and must not start reading E before that one is done; no cross-stream event hop (~10 us each)
struct SolveArgs {
  int on, method, nref /* unit references (= sd->unit_off[D]) */;
  const SolveDesc* sd; const int* unit_idx; const double* unit_sign; const double* E; const double* e_recip1;
  double* dcoef; double* out_host;
  long long mark_seq; long long* done_mark;               // device mark the blend phase (main stream) starts on
  const long long* sb;                                    // a captured launch: the mark's number is sb[SB_A]
  volatile long long* done_flag; long long done_seq;      // pinned flag the host spins on
  double* fault;
  // MORE THAN ONE RANK (diabats sharded): phase 1 (inside the energy-fetch kernel) assembles the Hamiltonian entries of THIS rank's diabats and writes them
  // into the rank's slot of the all-gather payload (layout of msevb_evb_payload_pack, comm.hip: {H_dd, H_pd} per slot, then H'_11); phase 2 (k_evb_solve_gathered, behind the
  // all-gather on the same stream) builds H' from the R gathered payloads - the same bytes on every rank -, finds the ground state and writes the coefficients of the rank's
  // own blend terms.  phase 0: one rank, everything in one go.
  int phase, R, rank, S, P; double* payload;
};
// the descriptor is static for the call: the kernels stage it in LDS BEFORE they wait for the energies (a dozen dependent global loads, ~1 us each, used to sit
// between the last energy and the blend phase's start)
constexpr int SOLVE_PRE = 256;      // unit references staged with the descriptor (the rest, if any, is read from global memory)
struct SolveStage { SolveDesc sd; int ui[SOLVE_PRE]; double us[SOLVE_PRE]; };
__device__ __forceinline__ void evb_solve_stage(const SolveArgs& sa, SolveStage* st) {
  const int* src = (const int*)sa.sd; int* dst = (int*)&st->sd;
  for (int k = threadIdx.x; k < (int)(sizeof(SolveDesc) / 4); k += blockDim.x) dst[k] = src[k];
  for (int k = threadIdx.x; k < min(sa.nref, SOLVE_PRE); k += blockDim.x) { st->ui[k] = sa.unit_idx[k]; st->us[k] = sa.unit_sign[k]; }
}
__device__ __forceinline__ void evb_solve_body(const SolveArgs& sa, const SolveStage* stg) {
#pragma clang fp contract(off)
  const SolveDesc* sd = &stg->sd;
  const int* __restrict__ unit_idx = sa.unit_idx; const double* __restrict__ unit_sign = sa.unit_sign;
  const double* __restrict__ E = sa.E; const double* __restrict__ e_recip1 = sa.e_recip1; double* __restrict__ dcoef = sa.dcoef; double* __restrict__ out_host = sa.out_host;
  const int method = sa.method; long long* done_mark = sa.done_mark; const long long wait_seq = sa.sb ? sa.sb[SB_A] : sa.mark_seq; 
  extern __shared__ double sh[];      // A[n*n], V[n*n], cplA[n], per pair of a round: c, s, d_p - t a_pq, d_q + t a_pq (n/2 each) and p, q (ints)
  const int n = sd->D, tid = threadIdx.x, nth = blockDim.x;
  const int m = (n + 1) & ~1, hp = m / 2;      // tournament over m players (an odd n plays with a dummy that sits out)
  double* A = sh; double* V = A + n * n; double* cA = V + n * n; double* rc = cA + n; double* rs = rc + hp; double* rdp = rs + hp; double* rdq = rdp + hp;
  int* rp = (int*)(rdq + hp); int* rq = rp + hp;
  __shared__ int s_nonzero;
  const int nu = sd->nu, ncp = sd->ncp, ngrid = sd->ngrid;
  for (int k = tid; k < n * n; k += nth) { A[k] = 0.0; V[k] = (k / n == k % n) ? 1.0 : 0.0; }
  for (int k = tid; k < n; k += nth) cA[k] = 0.0;
  __syncthreads();
  const double Er1 = *e_recip1;
  // the units' energies (two dependent global loads each) are fetched by all lanes at once and summed per diabat in their fixed order afterwards
  double* contrib = rdq + hp + 2 * hp;      // scratch behind the round tables: [unit references] now, the squaring's product later
  const int nref = sd->unit_off[n];
  for (int q = tid; q < nref; q += nth) { const int iu = q < SOLVE_PRE ? stg->ui[q] : unit_idx[q]; contrib[q] = (q < SOLVE_PRE ? stg->us[q] : unit_sign[q]) * (E[iu] + E[nu + iu]); }
  __syncthreads();
  const int phase = sa.phase, R = sa.R > 0 ? sa.R : 1;
  for (int d = tid; d < n; d += nth) {
    if (d == 0) { const int iu = sd->unit00; A[0] = sd->Hp_norep + sd->once * (E[iu] + E[nu + iu]); }
    else if (phase != 0 && (d - 1) % R != sa.rank) continue;      // another rank's diabat: its entries come with the all-gather
    else {
      double e = sd->e_hyd + sd->eref_delta[d];
      for (int q = sd->unit_off[d]; q < sd->unit_off[d + 1]; ++q) e += contrib[q];
      e += E[2 * nu + ncp + sd->grid_of[d]] - Er1;
      A[d * n + d] = e;
      const double hpd = E[2 * nu + ncp + ngrid + 3 * sd->cp_of[d] + 2];
      const int p = sd->parent[d];
      A[p * n + d] = hpd; A[d * n + p] = hpd;
      cA[d] = E[2 * nu + ncp + ngrid + 3 * sd->cp_of[d]];
    }
  }
  __syncthreads();
  if (phase == 1) {      // this rank's slot of the payload; the solve follows the all-gather (k_evb_solve_gathered)
    double* mine = sa.payload + (size_t)sa.rank * sa.P;
    for (int k = tid; k < sa.P; k += nth) mine[k] = 0.0;
    __syncthreads();
    for (int d = 1 + tid; d < n; d += nth) if ((d - 1) % R == sa.rank) { const int s = (d - 1) / R; mine[2 * s] = A[d * n + d]; mine[2 * s + 1] = A[sd->parent[d] * n + d]; }
    if (tid == 0) mine[2 * sa.S] = A[0];
    return;
  }
  if (phase == 2) {      // the other ranks' entries (and H'_11 as rank 0 has it: every rank then holds the same matrix, bit for bit)
    for (int d = 1 + tid; d < n; d += nth) {
      const int r = (d - 1) % R, s = (d - 1) / R, p = sd->parent[d];
      const double hdd = sa.payload[(size_t)r * sa.P + 2 * s], hpd = sa.payload[(size_t)r * sa.P + 2 * s + 1];
      A[d * n + d] = hdd; A[p * n + d] = hpd; A[d * n + p] = hpd;
    }
    if (tid == 0) A[0] = sa.payload[2 * sa.S];
    __syncthreads();
  }
  for (int k = tid; k < n * n; k += nth) out_host[2 + 2 * n + k] = (k / n <= k % n) ? A[k] : 0.0;      // H' as the host assembles it (upper triangle)
  int status = -1;
  int gfast = -1; double efast = 0.0;
  // ---- ground state by REPEATED SQUARING (n <= 16): P = sigma I - H' (sigma = Gershgorin's upper bound: P is positive semi-definite, its top
  // eigenvector the ground state of H'), P <- P^2 / tr(P^2) converges to the projector v v^T like rho^(2^k), rho = second / top eigenvalue of P -
  // a dozen 8 x 8 products where the Jacobi sweeps below are ~40 dependent rounds of divisions and square roots (35 us on the main stream behind the
  // pair loop, measured; this: ~3 us).  Accepted only when the residual |H' v - e v| says it converged (a near-degenerate ground state does
  // not within 48 squarings: the Jacobi below then decides, as the reference's would); v agrees with Jacobi's vector to rounding / gap.
  if (n <= 16 && n > 1 && method == 1) {
    double* Pm = V; double* Tm = rdq + hp + 2 * hp;      // (V is rebuilt below if Jacobi has to run; Tm: n*n doubles behind the round tables)
    __shared__ double s_sig, s_tr, s_row[16], s_nrm;
    __syncthreads();      // (contrib, which Tm overlays, has been consumed)
    if (tid < n) { double r = 0.0; for (int j = 0; j < n; ++j) r += j == tid ? A[tid * n + j] : fabs(A[tid * n + j]); s_row[tid] = r; }      // Gershgorin: a_ii + sum_{j != i} |a_ij|
    __syncthreads();
    if (tid == 0) {
      double sg = -1e300, an = 0.0;
      for (int i = 0; i < n; ++i) { sg = fmax(sg, s_row[i]); an = fmax(an, fabs(s_row[i])); }
      double tr = 0.0; for (int i = 0; i < n; ++i) tr += sg - A[i * n + i];
      s_sig = sg; s_tr = tr; s_nrm = an;      // an: a scale of H' for the residual test
    }
    __syncthreads();
    const double sg = s_sig, tr0 = s_tr;
    if (tr0 > 0.0) {
      { const double inv = 1.0 / tr0; for (int k = tid; k < n * n; k += nth) Pm[k] = ((k / n == k % n ? sg : 0.0) - A[k]) * inv; }
      __syncthreads();
      int extra = 0;
      for (int it = 0; it < 48; ++it) {
        for (int k = tid; k < n * n; k += nth) { const int i = k / n, j = k % n; double t = 0.0; for (int l = 0; l < n; ++l) t += Pm[i * n + l] * Pm[l * n + j]; Tm[k] = t; }
        __syncthreads();
        double tr = 0.0; for (int i = 0; i < n; ++i) tr += Tm[i * n + i];      // every lane the same sum: tr(P^2) -> 1 for a projector of trace 1
        const bool done = 1.0 - tr < 1e-15;
        const double inv = 1.0 / tr;
        __syncthreads();
        for (int k = tid; k < n * n; k += nth) Pm[k] = Tm[k] * inv;
        __syncthreads();
        if (done && ++extra == 2) break;
        if (!done) extra = 0;
      }
      if (extra == 2) {      // v = the column of the projector with the largest diagonal element, normalised; e = v^T H' v; residual test
        int jm = 0; for (int j = 1; j < n; ++j) if (Pm[j * n + j] > Pm[jm * n + jm]) jm = j;
        double nrm = 0.0; for (int i = 0; i < n; ++i) nrm += Pm[i * n + jm] * Pm[i * n + jm];
        nrm = 1.0 / sqrt(nrm);
        __syncthreads();
        for (int i = tid; i < n; i += nth) Tm[i] = Pm[i * n + jm] * nrm;      // v
        __syncthreads();
        for (int i = tid; i < n; i += nth) { double t = 0.0; for (int j = 0; j < n; ++j) t += A[i * n + j] * Tm[j]; Tm[n + i] = t; }      // H' v
        __syncthreads();
        double e = 0.0, res = 0.0;
        for (int i = 0; i < n; ++i) e += Tm[i] * Tm[n + i];
        for (int i = 0; i < n; ++i) res = fmax(res, fabs(Tm[n + i] - e * Tm[i]));
        if (res <= 1e-13 * fmax(s_nrm, 1e-300)) { gfast = 0; efast = e; status = 0; }
      }
    }
    __syncthreads();
    if (gfast == 0) {      // hand the vector over in column 0 of V (what the epilogue reads), and stop here
      for (int i = tid; i < n; i += nth) V[i * n] = Tm[i];
      __syncthreads();
    } else {
      for (int k = tid; k < n * n; k += nth) V[k] = (k / n == k % n) ? 1.0 : 0.0;
      __syncthreads();
    }
  }
  for (int it = 1; it <= 50 && gfast < 0; ++it) {
    if (tid == 0) s_nonzero = 0;
    __syncthreads();
    { int nz = 0; for (int k = tid; k < n * n; k += nth) if (k / n < k % n && A[k] != 0.0) nz = 1; if (nz) s_nonzero = 1; }
    __syncthreads();
    if (!s_nonzero) { status = 0; break; }
    __syncthreads();
    for (int r = 0; r < m - 1; ++r) {
      for (int k = tid; k < hp; k += nth) {      // the pairs of round r, each with its rotation (src/general_routines.f90:2043-2062)
        int p = k == 0 ? m - 1 : (r + k) % (m - 1), q = k == 0 ? r : (r - k + (m - 1)) % (m - 1);
        if (p > q) { const int t = p; p = q; q = t; }
        double c = 1.0, s = 0.0, ndp = 0.0, ndq = 0.0;
        if (q < n) {
          const double apq = A[p * n + q], dp = A[p * n + p], dq = A[q * n + q];
          const double g = 100.0 * fabs(apq);
          if (it > 4 && fabs(dp) + g == fabs(dp) && fabs(dq) + g == fabs(dq)) { A[p * n + q] = 0.0; A[q * n + p] = 0.0; }
          else if (apq != 0.0) {
            const double hd = dq - dp;
            double t;
            if (fabs(hd) + g == fabs(hd)) t = apq / hd;
            else { const double theta = 0.5 * hd / apq; t = 1.0 / (fabs(theta) + sqrt(1.0 + theta * theta)); if (theta < 0.0) t = -t; }
            c = 1.0 / sqrt(1 + t * t); s = t * c;
            const double h = t * apq;
            ndp = dp - h; ndq = dq + h;
          }
        } else q = -1;
        rp[k] = p; rq[k] = q; rc[k] = c; rs[k] = s; rdp[k] = ndp; rdq[k] = ndq;
      }
      __syncthreads();
      for (int w = tid; w < n * hp; w += nth) {      // columns p, q of A and of V (A <- A J, V <- V J)
        const int i = w / hp, k = w % hp, p = rp[k], q = rq[k];
        const double s = rs[k];
        if (q < 0 || s == 0.0) continue;
        const double c = rc[k];
        const double ap = A[i * n + p], aq = A[i * n + q]; A[i * n + p] = c * ap - s * aq; A[i * n + q] = s * ap + c * aq;
        const double vp = V[i * n + p], vq = V[i * n + q]; V[i * n + p] = c * vp - s * vq; V[i * n + q] = s * vp + c * vq;
      }
      __syncthreads();
      for (int w = tid; w < n * hp; w += nth) {      // rows p, q (A <- J^T A)
        const int j = w / hp, k = w % hp, p = rp[k], q = rq[k];
        const double s = rs[k];
        if (q < 0 || s == 0.0) continue;
        const double c = rc[k];
        const double ap = A[p * n + j], aq = A[q * n + j]; A[p * n + j] = c * ap - s * aq; A[q * n + j] = s * ap + c * aq;
      }
      __syncthreads();
      for (int k = tid; k < hp; k += nth) {      // the annihilated element is zero, the diagonal moves by t a_pq (the reference's update, :2055-2059: no cancellation)
        const int p = rp[k], q = rq[k];
        if (q < 0 || rs[k] == 0.0) continue;
        A[p * n + q] = 0.0; A[q * n + p] = 0.0; A[p * n + p] = rdp[k]; A[q * n + q] = rdq[k];
      }
      __syncthreads();
    }
  }
  // ground state, blend coefficients
  int gidx = 0; double eg = A[0];
  if (gfast == 0) eg = efast;
  else for (int i = 1; i < n; ++i) if (A[i * n + i] < eg) { eg = A[i * n + i]; gidx = i; }
  if (tid == 0) { out_host[0] = (double)status; out_host[1] = eg; dcoef[0] = V[gidx] * V[gidx]; }
  for (int k = tid; k < n; k += nth) { out_host[2 + k] = V[k * n + gidx]; out_host[2 + n + k] = cA[k]; }
  for (int d = 1 + tid; d < n; d += nth) if (sd->grid_of[d] >= 0) { const double ev = V[d * n + gidx]; dcoef[1 + sd->grid_of[d]] = ev * ev; }
  for (int t = tid; t < sd->nterm; t += nth) {
    const int i = sd->term_i[t], j = sd->term_j[t];
    const double ei = V[i * n + gidx], ej = V[j * n + gidx];
    dcoef[1 + ngrid + t] = i == j ? ei * ej : 2.0 * ei * ej * -cA[j];
  }
  if (done_mark) { __syncthreads(); if (tid == 0) __hip_atomic_store(done_mark, wait_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }      // the blend phase (main stream) starts on this
  // (k_evb_publish raises sa.done_flag for the host together with its own flag, behind the copies)
}


// ---------------------------------------------------------------------------------------------------------------------------------
// SPECULATIVE EVB STEP (f2): the diabat tree (which protons can hop where) changes every few dozen steps, the positions every step.
// On a step that enumerated the tree itself the host leaves a SpecPlan on the device: for every descriptor (unit, coupling, reciprocal
// atom) which diabat and how many hops its topology has behind it, the atoms each hop made whole (evb_change_data_structures_proton_transfer,
// src/ms_evb.f90:876), and - for every molecule the depth-first search expanded - the reactive pairs it found (find_evb_reactive_neighbors,
// :702-764).  The next step then does NOT wait for the neighbourhood record: k_evb_region re-evaluates the search predicates of the expanded
// donors on the new positions (same expressions as the host's search; found pairs must be exactly the recorded ones), k_evb_refresh rewrites the
// positions, centres of mass and shifts inside the descriptors already on the device, and the host enqueues the side chains at once.
// Its verdict travels with the first energy fetch; a changed tree (rare) makes the host repeat the force call the long way (MSEVB_RETRY).
constexpr int SPEC_MAX_PAIRS = 12, SPEC_MAX_DONORS = 32;
struct SpecPlan {
  int n_units, n_cps, n_ras, n_donors, n_pairs_total, hyd;
  int unit_d[MAX_UNITS], unit_k[MAX_UNITS];                                          // diabat of the unit, hops its topology has behind it
  int cp_d[MAX_DIABATS], cp_nhop[MAX_DIABATS], cp_ra0[MAX_DIABATS], cp_ra1[MAX_DIABATS];   // diabat / hops of a coupling; its diabat's reciprocal-space atoms [ra0, ra1)
  int mw_n[MAX_DIABATS][MAX_CHAIN], mw_atoms[MAX_DIABATS][MAX_CHAIN][MAX_MOL_ATOMS];   // acceptor's atoms in the order make_molecule_whole walked them
  int donor_mol[SPEC_MAX_DONORS], donor_np[SPEC_MAX_DONORS], donor_pair[SPEC_MAX_DONORS][SPEC_MAX_PAIRS][3];   // {proton (local), acceptor molecule, basic atom (local)}
};

#include "evb_plan.hpp"      // enumeration + descriptor construction, one source for host and device

// the tree check inside k_evb_region: thread = molecule j, against every expanded donor (their centres of mass and proton positions once per workgroup in LDS)
__device__ __forceinline__ void spec_check_molecule(const SpecPlan* __restrict__ pl, const EvbTablesDev* __restrict__ tb, const int* __restrict__ mol_type, int m, int M,
                                                    const int* __restrict__ mol_natom, const int* __restrict__ mol_atoms, const double* __restrict__ xyz, const double* __restrict__ mass,
                                                    const BoxDev& box, const double* com_m /* of molecule m, valid when m < M */, int* __restrict__ check) {
  __shared__ double s_com[SPEC_MAX_DONORS][3], s_x[SPEC_MAX_DONORS][MAX_MOL_ATOMS][3];
  const int nd = pl->n_donors;
  if ((int)threadIdx.x < nd) {
    const int i = pl->donor_mol[threadIdx.x], n = mol_natom[i];
    double c[3] = {0, 0, 0}, mt = 0;
    for (int k = 0; k < n; ++k) { const int a = mol_atoms[i * MAX_MOL_ATOMS + k]; for (int d = 0; d < 3; ++d) { c[d] += xyz[3 * a + d] * mass[a]; s_x[threadIdx.x][k][d] = xyz[3 * a + d]; } mt += mass[a]; }
    for (int d = 0; d < 3; ++d) s_com[threadIdx.x][d] = c[d] / mt;
  }
  __syncthreads();
  if (m >= M) return;
  const int tj = mol_type[m], nj = mol_natom[m];
  for (int q = 0; q < nd; ++q) {
    const int i = pl->donor_mol[q];
    if (i == m) continue;
    double sh[3], dc[3];
    for (int d = 0; d < 3; ++d) { sh[d] = floor((com_m[d] - s_com[q][d]) * box.invL[d] + 0.5) * box.L[d]; dc[d] = com_m[d] - s_com[q][d] - sh[d]; }
    if (dc[0] * dc[0] + dc[1] * dc[1] + dc[2] * dc[2] >= 25.0) continue;       // evb_first_solvation_cutoff
    const int ti = mol_type[i], ni = mol_natom[i];
    for (int ia = 0; ia < ni; ++ia) {
      if (tb->reactive_p[ti][ia] != 1) continue;
      for (int ja = 0; ja < nj; ++ja) {
        if (tb->reactive_b[tj][ja] != 1) continue;
        const int a = mol_atoms[m * MAX_MOL_ATOMS + ja];
        double r[3];
        for (int d = 0; d < 3; ++d) r[d] = xyz[3 * a + d] - s_x[q][ia][d] - sh[d];
        if (r[0] * r[0] + r[1] * r[1] + r[2] * r[2] >= 6.25) continue;           // evb_reactive_pair_distance
        bool known = false;
        for (int p = 0; p < pl->donor_np[q]; ++p) known = known || (pl->donor_pair[q][p][0] == ia && pl->donor_pair[q][p][1] == m && pl->donor_pair[q][p][2] == ja);
        if (known) atomicAdd(&check[0], 1); else atomicOr(&check[1], 1);
      }
    }
  }
}

// one WAVE per descriptor (lane = special atom): descriptors refreshed in place; the last workgroup to finish writes the candidate sphere and the
// verdict of the tree check (-> device word for later kernels, pinned word for the host).  scratch: {far^2 bits (64), ticket, pad}
__global__ __launch_bounds__(64) void k_evb_refresh(const SpecPlan* __restrict__ pl, UnitDesc* __restrict__ units, CouplingDesc* __restrict__ cps, RecipAtom* __restrict__ ras,
                                                     const double* __restrict__ xyz, const double* __restrict__ mass, const int* __restrict__ mol_natom, const int* __restrict__ mol_atoms,
                                                     BoxDev box, double extra_R, double* __restrict__ sphere, int* __restrict__ check, int* __restrict__ verdict_host,
                                                     unsigned long long* __restrict__ scratch, int* __restrict__ cand_counter = nullptr) {      // verdict_host == nullptr: k_evb_region, on a stream of its own, delivers it; cand_counter: re-armed here instead
  __shared__ double s_pos[MAX_SPEC][3], s_c[3];
  __shared__ int s_mw[MAX_CHAIN][MAX_MOL_ATOMS], s_mwn[MAX_CHAIN];
  const int lane = threadIdx.x, b = blockIdx.x, nu = pl->n_units;
  const bool is_unit = b < nu;
  const int e = b - nu;
  SpecAtom* spec = is_unit ? units[b].spec : cps[e].spec;
  const int n_spec = is_unit ? units[b].n_spec : cps[e].n_spec, d = is_unit ? pl->unit_d[b] : pl->cp_d[e], khops = is_unit ? pl->unit_k[b] : pl->cp_nhop[e];
  if (lane == 0) {      // centre of the candidate sphere: the hydronium's centre of mass (as k_evb_region computes the record's)
    double ch[3] = {0, 0, 0}, mh = 0;
    const int hyd = pl->hyd;
    for (int k = 0; k < mol_natom[hyd]; ++k) { const int a = mol_atoms[hyd * MAX_MOL_ATOMS + k]; for (int q = 0; q < 3; ++q) ch[q] += xyz[3 * a + q] * mass[a]; mh += mass[a]; }
    for (int q = 0; q < 3; ++q) s_c[q] = ch[q] / mh;
  }
  // today's coordinates, then the make-whole passes of the hops this topology has behind it (a chain along the acceptor's atoms: one shuffle per link)
  if (lane < MAX_CHAIN * MAX_MOL_ATOMS) s_mw[lane / MAX_MOL_ATOMS][lane % MAX_MOL_ATOMS] = pl->mw_atoms[d][lane / MAX_MOL_ATOMS][lane % MAX_MOL_ATOMS];      // (one parallel load instead of a chain of dependent ones)
  if (lane < MAX_CHAIN) s_mwn[lane] = pl->mw_n[d][lane];
  const bool act = lane < n_spec;
  const int my = act ? spec[lane].idx : -1;
  double p[3] = {0, 0, 0};
  if (act) for (int q = 0; q < 3; ++q) p[q] = xyz[3 * (size_t)my + q];
  __syncthreads();
  for (int h = 0; h < khops; ++h) {
    const int n = s_mwn[h];
    int mine = -1;
    for (int i = 0; i < n; ++i) if (s_mw[h][i] == my) mine = i;
    for (int i = 1; i < n; ++i) {
      const unsigned long long mp = __ballot(mine == i - 1);
      const int lp = mp ? __ffsll((long long)mp) - 1 : 0;
      const double px = __shfl(p[0], lp, 64), py = __shfl(p[1], lp, 64), pz = __shfl(p[2], lp, 64);
      if (mine == i) {
        const double pr[3] = {px, py, pz};
        double sh[3];
        for (int q = 0; q < 3; ++q) sh[q] = floor((p[q] - pr[q]) * box.invL[q] + 0.5) * box.L[q];
        if (fabs(sh[0]) > 1e-6 || fabs(sh[1]) > 1e-6 || fabs(sh[2]) > 1e-6) for (int q = 0; q < 3; ++q) p[q] = pr[q] + (p[q] - pr[q] - sh[q]);
      }
    }
  }
  __syncthreads();
  double d2 = 0.0;
  if (act) for (int q = 0; q < 3; ++q) { spec[lane].x[q] = p[q]; s_pos[lane][q] = p[q]; double dd = p[q] - s_c[q]; dd -= box.L[q] * floor(dd / box.L[q] + 0.5); d2 += dd * dd; }
  if (!is_unit) d2 = 0.0;      // (the host takes the radius from the units' atoms)
  for (int o = 32; o > 0; o >>= 1) d2 = fmax(d2, __shfl_xor(d2, o, 64));
  __syncthreads();
  if (!is_unit) {
    CouplingDesc& cp = cps[e];
    if (lane < cp.n_mols) {      // centres of mass in the molecules' atom order of this topology (mol_com_h)
      double s[3] = {0, 0, 0}, mt = 0;
      for (int k = 0; k < cp.n_spec; ++k) if (cp.spec[k].mol == cp.mols[lane]) { const double ma = mass[cp.spec[k].idx]; for (int q = 0; q < 3; ++q) s[q] = s[q] + s_pos[k][q] * ma; mt = mt + ma; }
      for (int q = 0; q < 3; ++q) cp.mol_com[lane][q] = s[q] / mt;
    }
    __threadfence_block();
    __syncthreads();
    if (lane == 0) {      // zundel_r_com (src/ms_evb.f90:2946)
      int kD = 0, kA = 0;
      for (int m = 0; m < cp.n_mols; ++m) { if (cp.mols[m] == cp.spec[cp.D_at[0]].mol) kD = m; if (cp.mols[m] == cp.spec[cp.A_at[0]].mol) kA = m; }
      double mdt = 0, mat = 0;
      for (int k = 0; k < cp.nD; ++k) mdt = mdt + mass[cp.spec[cp.D_at[k]].idx];
      for (int k = 0; k < cp.nA; ++k) mat = mat + mass[cp.spec[cp.A_at[k]].idx];
      for (int q = 0; q < 3; ++q) {
        const double cd = cp.mol_com[kD][q], ca = cp.mol_com[kA][q];
        const double sha = floor((ca - cd) * box.invL[q] + 0.5) * box.L[q];
        const double rca = cd + (ca - cd - sha);
        cp.zc[q] = (mdt * cd + mat * rca) / (mdt + mat); cp.shiftd[q] = 0.0; cp.shifta[q] = sha;
      }
    }
    // positions of the reciprocal-space atoms of this coupling's diabat: those of its final topology
    for (int r = pl->cp_ra0[e] + lane; r < pl->cp_ra1[e]; r += 64) {
      int k = 0;
      for (int q = 0; q < cp.n_spec; ++q) if (cp.spec[q].idx == ras[r].idx) k = q;
      for (int q = 0; q < 3; ++q) ras[r].x_new[q] = s_pos[k][q];
    }
  }
  if (lane != 0) return;
  atomicMax(scratch, (unsigned long long)__double_as_longlong(d2));      // (non-negative doubles order like their bit patterns)
  __threadfence();
  if (atomicAdd((unsigned*)(scratch + 1), 1u) != gridDim.x - 1) return;
  __threadfence();
  const double f2 = __longlong_as_double((long long)atomicAdd(scratch, 0ull));
  const double R = sqrt(f2) + extra_R;
  sphere[0] = s_c[0]; sphere[1] = s_c[1]; sphere[2] = s_c[2]; sphere[3] = R * R;
  scratch[0] = 0ull; scratch[1] = 0ull;
  if (cand_counter) *cand_counter = 0;
  if (!verdict_host) return;
  const int ok = (atomicAdd(&check[1], 0) == 0 && atomicAdd(&check[0], 0) == pl->n_pairs_total) ? 1 : 0;
  check[0] = 0; check[1] = 0; check[2] = ok;
  *verdict_host = ok ? 1 : -1;
}

// ---- f2: the enumeration and the descriptors of a step whose diabat tree CHANGED, on the device (one workgroup; evb_plan.hpp holds the logic, the host runs the same source as the checker) ----
// what the host needs back (pinned memory): sizes for its launches, the tree for the Hamiltonian it reports; seq is written last
struct PlanSummary {
  int err, fits, D, nu, ncp, nra, ngrid, ncp_big, n_region, n_donors;
  unsigned long long need_bytes;
  int plog[MAX_DIABATS][MAX_CHAIN][5], parent[MAX_DIABATS], nhop[MAX_DIABATS], grid_of[MAX_DIABATS], cp_of[MAX_DIABATS], unit0[MAX_DIABATS], nras[MAX_DIABATS];
  double eref_delta[MAX_DIABATS];
  long long t[8];      // 100 MHz clock at the kernel's phase boundaries (diagnostic: MSEVB_HOST_TIMING prints them)
  long long seq;
};
struct PlanArgs {
  const PlanTables* tb; const int* mol_type; const int* mol_natom; const int* mol_atoms; const int* atype; const double* chg; const double* mass;
  const RegionRec* region; const int* region_count; int region_cap; int hydronium;
  int* order; int* pair_tab; int* pair_cnt; int* near_tab; int* near_cnt;      // scratch: REGION_CAP, 3 PLAN_MAX_PAIRS REGION_CAP, REGION_CAP, 64 REGION_CAP, REGION_CAP ints
  char* desc; unsigned long long desc_cap;      // units | couplings | reciprocal-space atoms (the layout of the host path: offsets from the counts)
  CouplingDesc* cps_tmp;                        // MAX_DIABATS couplings in diabat order (sorted into desc at the end)
  SpecPlan* plan; PlanSummary* out; long long seq;
};
constexpr int PLAN_LDS_REC = 1024;      // records whose molecule ids / centres of mass are staged in LDS (larger neighbourhoods - chains of four hops - read them from global memory)
constexpr int PLAN_BATCH = 8;           // diabats constructed at a time: one per wave, work spaces in LDS
__global__ __launch_bounds__(512) void k_evb_plan(const PlanArgs A) {
  __shared__ PlanFrame s_frames[MAX_CHAIN + 2];
  __shared__ int s_plog[MAX_DIABATS][MAX_CHAIN][5], s_parent[MAX_DIABATS], s_nhop[MAX_DIABATS], s_unit0[MAX_DIABATS], s_r0[MAX_DIABATS], s_nras[MAX_DIABATS], s_grid[MAX_DIABATS], s_cp[MAX_DIABATS], s_own[MAX_DIABATS];
  __shared__ double s_eref[MAX_DIABATS];
  __shared__ int s_err, s_fits, s_D, s_nu, s_ncp, s_nra, s_n, s_ncp_big, s_pos[MAX_DIABATS], s_list[MAX_DIABATS], s_nlist;
  __shared__ unsigned long long s_off_c, s_off_r;
  __shared__ int s_mol[PLAN_LDS_REC];
  __shared__ PlanTables s_tb;      // (the search reads the box and the reactive-atom flags in its innermost loops: from global memory that was 1 us per iteration)
  // the tabulation phase keeps the records' centres of mass here, the construction phase its work spaces (one phase at a time)
  constexpr size_t UNION_BYTES = sizeof(PlanWork) * PLAN_BATCH > sizeof(double) * 3 * PLAN_LDS_REC ? sizeof(PlanWork) * PLAN_BATCH : sizeof(double) * 3 * PLAN_LDS_REC;
  __shared__ __attribute__((aligned(16))) unsigned char s_union[UNION_BYTES];
  double* s_com = (double*)s_union;
  PlanWork* s_work = (PlanWork*)s_union;
  const int tid = threadIdx.x, nt = blockDim.x;
  if (tid == 0) A.out->t[0] = wall_clock64();
  if (tid == 0) { s_err = PE_OK; s_fits = 1; s_n = min(*A.region_count, A.region_cap); if (*A.region_count > A.region_cap) s_err = PE_REGION_FULL; s_D = 1; s_nu = 1; s_ncp = 0; s_nra = 0; s_ncp_big = 0; s_off_c = 0; s_off_r = 0; s_nlist = 0; }
  for (int k = tid; k < (int)(sizeof(PlanTables) / 4); k += nt) ((int*)&s_tb)[k] = ((const int*)A.tb)[k];
  for (int k = tid; k < MAX_DIABATS * MAX_CHAIN * 5; k += nt) (&s_plog[0][0][0])[k] = -1;
  for (int k = tid; k < MAX_DIABATS; k += nt) { s_parent[k] = -1; s_nhop[k] = 0; s_unit0[k] = 0; s_r0[k] = 0; s_nras[k] = 0; s_grid[k] = -1; s_cp[k] = -1; s_own[k] = 0; s_eref[k] = 0.0; }
  __syncthreads();
  const int n = s_n;
  const bool lds_rec = n <= PLAN_LDS_REC;
  PlanEnv E;
  E.tb = &s_tb; E.mol_type = A.mol_type; E.mol_natom = A.mol_natom; E.mol_atoms = A.mol_atoms; E.atype = A.atype; E.chg = A.chg; E.mass = A.mass;
  E.rec = A.region; E.order = A.order; E.n_region = n; E.hydronium = A.hydronium;
  E.sorted_mol = lds_rec ? s_mol : (const int*)nullptr; E.sorted_com = lds_rec ? s_com : (const double*)nullptr; E.pair_tab = A.pair_tab; E.pair_cnt = A.pair_cnt;
  // A. molecule order of the record (the reference's j_mole loops run in ascending molecule index): rank by counting, the ids are distinct
  for (int k = tid; k < n; k += nt) {
    const int mk = A.region[k].mol; int r = 0;
    for (int j = 0; j < n; ++j) r += A.region[j].mol < mk ? 1 : 0;
    A.order[r] = k;
    if (lds_rec) { s_mol[r] = mk; for (int d = 0; d < 3; ++d) s_com[3 * r + d] = A.region[k].com[d]; }
  }
  __threadfence_block(); __syncthreads();
  if (tid == 0) A.out->t[1] = wall_clock64();
  //    reactive pairs of EVERY record molecule (plan_donor_pairs: the search's own loops), one molecule per thread
  //    in two passes: the shells (centres of mass from LDS, nothing else: every lane of a wave walks the same loop), then the pairs inside them (a dozen molecules each)
  //    Only molecules the search can EXPAND: a donor reached through k hops (k <= max_chain - 1, src/ms_evb.f90:531) lies within 5 k A of the hydronium, every hop
  //    joining two molecules of one first solvation shell (centres of mass within 5 A) - the rest of the record (three quarters of it) can only accept.
  const int a_hyd = plan_find_region_index(E, A.hydronium);
  const double reach = 5.0 * (s_tb.max_chain - 1) + 0.5;
  for (int a = tid; a < n; a += nt) {
    bool cand = a_hyd >= 0;
    if (cand) {
      const double* ch = lds_rec ? s_com + 3 * a_hyd : A.region[A.order[a_hyd]].com; const double* ca = lds_rec ? s_com + 3 * a : A.region[A.order[a]].com;
      double sh[3], d2 = 0.0; plan_pbc_shift(&s_tb, ch, ca, sh);
      for (int d = 0; d < 3; ++d) { const double dc = ca[d] - ch[d] - sh[d]; d2 += dc * dc; }
      cand = d2 <= reach * reach;
    }
    A.near_cnt[a] = cand ? plan_shell(E, a, A.near_tab + 64 * a) : -1;
  }
  __threadfence_block(); __syncthreads();
  for (int a = tid; a < n; a += nt) {
    int cnt = 0;
    const int pe = A.near_cnt[a] < 0 ? (int)PE_DONOR_OUTSIDE : plan_donor_pairs(E, a, A.near_tab + 64 * a, A.near_cnt[a], (int(*)[3])(A.pair_tab + 3 * PLAN_MAX_PAIRS * a), &cnt);
    A.pair_cnt[a] = pe ? -pe : cnt;
  }
  __threadfence_block(); __syncthreads();
  if (tid == 0) A.out->t[2] = wall_clock64();
  // B. the tree (one thread walks the short pair lists), then sizes and offsets
  if (tid == 0 && s_err == PE_OK) {
    int D = 1, fits = 1;
    const int pe = plan_enumerate(E, s_frames, s_plog, s_parent, &D, *A.plan, &fits);
    s_err = pe; s_fits = fits; s_D = D;
    if (pe == PE_OK) {
      int nu = 1, nra = 0, ncp = 0, nl = 0;
      s_list[nl++] = 0;      // (the principal diabat's unit is constructed with the others)
      for (int d = 1; d < D; ++d) {
        if (!plan_owned(&s_tb, d)) continue;
        int nhop = 0;
        const int na = plan_chain_atoms(E, d, s_plog, &nhop);
        s_own[d] = 1; s_nhop[d] = nhop; s_unit0[d] = nu; s_r0[d] = nra; s_nras[d] = na; s_grid[d] = ncp; s_cp[d] = ncp;
        nu += 2 * nhop; nra += na; ++ncp;
        s_list[nl++] = d;
      }
      s_nlist = nl;
      s_nu = nu; s_nra = nra; s_ncp = ncp;
      s_off_c = (unsigned long long)nu * sizeof(UnitDesc); s_off_r = s_off_c + (unsigned long long)ncp * sizeof(CouplingDesc);
      if (nu > MAX_UNITS || ncp > MAX_DIABATS || nra > MAX_DIABATS * MAX_SPEC) s_err = PE_STAGING;
      else if (s_off_r + (unsigned long long)nra * sizeof(RecipAtom) > A.desc_cap) s_err = PE_DESC_ROOM;
    }
  }
  __syncthreads();
  if (tid == 0) A.out->t[3] = wall_clock64();
  E.sorted_com = nullptr;      // (its storage becomes the work spaces now; the molecule ids stay)
  if (s_err == PE_OK) {
    // C. descriptors: one diabat per WAVE (its lane 0 - lanes of one wave on different diabats would take turns), PLAN_BATCH at a time, each on a private copy of the principal topology
    UnitDesc* units = (UnitDesc*)A.desc; RecipAtom* ras = (RecipAtom*)(A.desc + s_off_r);
    const int D = s_D, wave = tid >> 6, lane = tid & 63, nl = s_nlist;
    {      // zero-fill of every descriptor by the whole workgroup (coalesced), not by the one lane that constructs it
      unsigned long long* z = (unsigned long long*)A.desc; const size_t nz = (size_t)s_off_c / 8;
      for (size_t k = tid; k < nz; k += nt) z[k] = 0ull;
      unsigned long long* zc = (unsigned long long*)A.cps_tmp; const size_t nzc = (size_t)s_ncp * sizeof(CouplingDesc) / 8;
      for (size_t k = tid; k < nzc; k += nt) zc[k] = 0ull;
      __threadfence_block(); __syncthreads();
    }
    for (int base = 0; base < nl; base += PLAN_BATCH) {
      const int q = base + wave;
      if (lane == 0 && wave < PLAN_BATCH && q < nl) {
        const int d = s_list[q];
        PlanWork& W = s_work[wave]; PTopo& P = W.P;
        P.n_mols = 0; P.n_atoms = 0; P.hydronium = A.hydronium;
        int perr = PE_OK;
        const long long tc0 = wall_clock64();
        const int em = plan_ensure_mol(E, P, A.hydronium, &perr);
        if (q == nl - 1) A.out->t[7] = wall_clock64() - tc0;
        if (em < 0) atomicMax(&s_err, perr != PE_OK ? perr : (int)PE_HYD_OUTSIDE);
        else if (d == 0) { const int pe = plan_make_unit(E, P, -1, -1, +1.0, 0, 0, 1, units[0], W.where, true); if (pe) atomicMax(&s_err, pe); A.plan->unit_d[0] = 0; A.plan->unit_k[0] = 0; }
        else {
          int nr = 0, nhop = 0; double er = 0.0;
          const int pe = plan_build_diabat(E, W, d, D, s_plog, units, s_unit0[d], ras, s_r0[d], s_grid[d], A.cps_tmp[s_cp[d]], *A.plan, &er, &nr, &nhop, true);
          if (pe) atomicMax(&s_err, pe);
          else if (nr != s_nras[d] || nhop != s_nhop[d]) atomicMax(&s_err, (int)PE_STAGING);
          s_eref[d] = er;
        }
        if (q == nl - 1) A.out->t[6] = wall_clock64() - tc0;
      }
      __syncthreads();
    }
  }
  __threadfence(); __syncthreads();
  if (tid == 0) A.out->t[4] = wall_clock64();
  if (s_err == PE_OK) {
    // D. couplings with more than 8 Zundel atoms first (they take the 16-lane instantiation of k_evb_coupling), otherwise diabat order: a stable partition
    if (tid == 0) {
      int pos = 0, big = 0;
      for (int pass = 0; pass < 2; ++pass) for (int d = 1; d < s_D; ++d) if (s_own[d]) { const CouplingDesc& cp = A.cps_tmp[s_cp[d]]; const bool isbig = cp.nD + cp.nA > 8; if (isbig == (pass == 0)) { s_pos[s_cp[d]] = pos++; big += isbig ? 1 : 0; } }
      s_ncp_big = big;
      SpecPlan& pl = *A.plan;
      for (int d = 1; d < s_D; ++d) if (s_own[d]) { const int k = s_pos[s_cp[d]]; pl.cp_d[k] = d; pl.cp_nhop[k] = s_nhop[d]; pl.cp_ra0[k] = s_r0[d]; pl.cp_ra1[k] = s_r0[d] + s_nras[d]; s_cp[d] = k; }
      pl.n_units = s_nu; pl.n_cps = s_ncp; pl.n_ras = s_nra;
    }
    __syncthreads();
    CouplingDesc* cps = (CouplingDesc*)(A.desc + s_off_c);
    const int words = (int)(sizeof(CouplingDesc) / 8);
    for (int k = 0; k < s_ncp; ++k) { const unsigned long long* src = (const unsigned long long*)&A.cps_tmp[k]; unsigned long long* dst = (unsigned long long*)&cps[s_pos[k]]; for (int q = tid; q < words; q += nt) dst[q] = src[q]; }
  }
  __threadfence_system(); __syncthreads();
  PlanSummary* o = A.out;
  for (int k = tid; k < MAX_DIABATS * MAX_CHAIN * 5; k += nt) (&o->plog[0][0][0])[k] = (&s_plog[0][0][0])[k];
  for (int k = tid; k < MAX_DIABATS; k += nt) { o->parent[k] = s_parent[k]; o->nhop[k] = s_nhop[k]; o->grid_of[k] = s_own[k] ? s_grid[k] : -1; o->cp_of[k] = s_own[k] ? s_cp[k] : -1; o->unit0[k] = s_unit0[k]; o->nras[k] = s_nras[k]; o->eref_delta[k] = s_eref[k]; }
  if (tid == 0) {
    o->err = s_err; o->fits = s_fits; o->D = s_D; o->nu = s_nu; o->ncp = s_ncp; o->nra = s_nra; o->ngrid = s_ncp; o->ncp_big = s_ncp_big; o->n_region = n; o->n_donors = A.plan->n_donors;
    o->need_bytes = s_off_r + (unsigned long long)s_nra * sizeof(RecipAtom);
  }
  __threadfence_system(); __syncthreads();
  if (tid == 0) { o->t[5] = wall_clock64(); __threadfence_system(); o->seq = A.seq; }
}

__global__ void k_evb_region(int M, int hyd, double R2, const int* __restrict__ mol_natom, const int* __restrict__ mol_atoms, const double* __restrict__ xyz,
                             const double* __restrict__ mass, BoxDev box, int* __restrict__ counter, RegionRec* __restrict__ out, int cap,
                             int* __restrict__ cand_counter, RegionRec* __restrict__ out_host, int cap_host,
                             int* __restrict__ ticket, int* __restrict__ host_count, volatile long long* flag, long long seq,
                             const SpecPlan* __restrict__ plan = nullptr, const EvbTablesDev* __restrict__ tabs = nullptr, const int* __restrict__ mol_type = nullptr, int* __restrict__ check = nullptr,
                             int* __restrict__ verdict_host = nullptr, const long long* __restrict__ sb = nullptr) {      // sb: step graph - the sequence number comes from the device-resident step block.  verdict_host: this kernel runs BESIDE the descriptor chain (own stream): it delivers the verdict of the tree check and re-arms its own counter
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  double com_m[3] = {0, 0, 0};
  if (m == 0 && !verdict_host) *cand_counter = 0;      // re-arm k_evb_candidates (which in turn re-arms `counter` for the next step)
  if (m < M) {
    double ch[3] = {0, 0, 0}, mh = 0;
    for (int k = 0; k < mol_natom[hyd]; ++k) { const int a = mol_atoms[hyd * MAX_MOL_ATOMS + k]; for (int d = 0; d < 3; ++d) ch[d] += xyz[3 * a + d] * mass[a]; mh += mass[a]; }
    const int n = mol_natom[m];
    double com[3] = {0, 0, 0}, mt = 0;
    for (int k = 0; k < n; ++k) { const int a = mol_atoms[m * MAX_MOL_ATOMS + k]; for (int d = 0; d < 3; ++d) com[d] += xyz[3 * a + d] * mass[a]; mt += mass[a]; }
    double dr[3];
    for (int d = 0; d < 3; ++d) { com[d] /= mt; dr[d] = com[d] - ch[d] / mh; com_m[d] = com[d]; }
    min_image(dr, box);
    if (dr[0] * dr[0] + dr[1] * dr[1] + dr[2] * dr[2] < R2) {
      const int slot = atomicAdd(counter, 1);
      if (slot < cap) {
        RegionRec r; r.mol = m; r.natom = n;
        for (int d = 0; d < 3; ++d) r.com[d] = com[d];
        for (int k = 0; k < n; ++k) { const int a = mol_atoms[m * MAX_MOL_ATOMS + k]; for (int d = 0; d < 3; ++d) r.x[k][d] = xyz[3 * a + d]; }
        out[slot] = r;
        if (slot < cap_host) out_host[slot] = r;      // pinned host memory, written directly: no copy command behind the kernel
      }
    }
  }
  if (plan) spec_check_molecule(plan, tabs, mol_type, m, M, mol_natom, mol_atoms, xyz, mass, box, com_m, check);      // speculative step: is the diabat tree still the recorded one?
  // the workgroup that finishes last publishes the record count and the sequence number the host spins on (was a launch of its own
  // on the path the host's turn waits for): every block makes its records visible to the host before it draws its ticket
  __shared__ bool last;
  __syncthreads();
  if (threadIdx.x == 0) { __threadfence_system(); last = atomicAdd(ticket, 1) == (int)gridDim.x - 1; }
  __syncthreads();
  if (!last || threadIdx.x != 0) return;
  __threadfence_system();
  *host_count = atomicAdd(counter, 0);
  *ticket = 0;
  if (verdict_host) {      // every workgroup's check counts are in (each drew its ticket behind them)
    const int ok = (atomicAdd(&check[1], 0) == 0 && atomicAdd(&check[0], 0) == plan->n_pairs_total) ? 1 : 0;
    check[0] = 0; check[1] = 0; check[2] = ok;
    *counter = 0;
    *verdict_host = ok ? 1 : -1;
  }
  __threadfence_system();
  *flag = sb ? sb[SB_R] : seq;
}

// copies two small device arrays into pinned host memory (zero-copy stores): the energy downloads, without copy commands
// flag != nullptr: after the data, the sequence number goes to pinned memory - the host spins on it instead of sleeping on an event
// ng > 0: first the diabats' grid energies, E_grid[g] = conv (E_lin[g] + E_quad[g] / 2), accumulators re-zeroed (k_evb_grid_energy - E_grid lies inside a)
__global__ void k_evb_publish(const double* a, int na, double* __restrict__ ha, const double* b, int nb, double* __restrict__ hb,
                              volatile long long* flag = nullptr, long long seq = 0, int ng = 0, double* E_lin = nullptr, double* E_quad = nullptr,
                              double conv = 0.0, double* E_grid = nullptr, const double* __restrict__ slot_part = nullptr, int n_part = 0, double* slot_out = nullptr,
                              const long long* chain_done = nullptr, long long chain_seq = 0, int n_chains = 0, double* fault = nullptr, const SolveArgs sa = SolveArgs(),
                              const long long* __restrict__ sb = nullptr, int sb_flag = 0) {      // sb: step graph - `seq` is sb[sb_flag], the solve's done_seq sb[SB_S] (a and E_grid alias: no __restrict__ on them)
  // n_chains > 0: the kernels on OTHER streams whose energies this kernel publishes end with k_evb_mark(chain_done + k, chain_seq); wait for
  // those marks here instead of through two cross-stream event hops in front of this kernel (~10 us each on the way to the host's turn).
  // The marks are enqueued before this kernel in host order, so streams that share a hardware queue cannot deadlock; after 2 s of wall
  // clock the sticky time-out flag (energy slot S_TIMEOUT) is raised and the kernel goes on (the force call then ends in MSEVB_ERR_TIMEOUT).
  __shared__ SolveStage s_sd;
  if (sa.on) evb_solve_stage(sa, &s_sd);      // (in flight while thread 0 waits for the chains)
  if (n_chains > 0) {
    if (threadIdx.x == 0) {
      const long long t0 = wall_clock64();      // 100 MHz
      for (int k = 0; k < n_chains; ++k)
        while (__hip_atomic_load(chain_done + k, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != (sb ? sb[SB_A] : chain_seq)) {
          if (wall_clock64() - t0 > 200000000LL) { if (fault) *fault = 2.0; break; }      // 2 s (the first launch of a kernel in a process can take milliseconds to load)
          __builtin_amdgcn_s_sleep(2);
        }
    }
    __syncthreads();
  }
  if (n_part > 0) {      // pending partial sums of ONE energy slot (the principal reciprocal energy: was a k_reduce_slots launch in front of this kernel); slot_out may be b
    __shared__ double red[256];
    double s0 = 0.0;
    for (int k = threadIdx.x; k < n_part; k += blockDim.x) s0 += slot_part[k];
    red[threadIdx.x] = s0;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) { if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st]; __syncthreads(); }
    if (threadIdx.x == 0) *slot_out = red[0];
    __threadfence_block();
    __syncthreads();
  }
  if (ng > 0) {
    for (int g = threadIdx.x; g < ng; g += blockDim.x) { E_grid[g] = conv * (E_lin[g] + 0.5 * E_quad[g]); E_lin[g] = 0.0; E_quad[g] = 0.0; }
    __threadfence_block();
    __syncthreads();
  }
  // sa.on: the energies this kernel has just completed are all the Hamiltonian needs - H' assembly, ground state and blend coefficients right here
  // (a launch of its own behind this kernel cost a gap and a minimum kernel duration, ~7 us, on the way to the blend phase's start), and BEFORE the copies
  // for the host: the blend phase waits for the solve's mark, the host only for its bookkeeping
  if (sa.on) { __threadfence_block(); __syncthreads(); evb_solve_body(sa, &s_sd); }
  for (int k = threadIdx.x; k < na; k += blockDim.x) ha[k] = a[k];
  for (int k = threadIdx.x; k < nb; k += blockDim.x) hb[k] = b[k];
  if (flag || (sa.on && sa.done_flag)) {
    __threadfence_system(); __syncthreads();
    if (threadIdx.x == 0) { if (flag) *flag = sb ? sb[sb_flag] : seq; if (sa.on && sa.done_flag && sa.phase != 1) *sa.done_flag = sb ? sb[SB_S] : sa.done_seq; }
  }
}

// more than one rank: Hamiltonian from the gathered payloads, ground state, this rank's blend coefficients (phase 2 of evb_solve_body), on the stream of the all-gather
// right behind it; the blend phase on the main stream starts on its mark, the host reads eigenvector and H' from pinned memory behind its flag
__global__ __launch_bounds__(256) void k_evb_solve_gathered(const SolveArgs sa, const long long* __restrict__ sb) {
  __shared__ SolveStage s_sd;
  evb_solve_stage(sa, &s_sd);
  __syncthreads();
  evb_solve_body(sa, &s_sd);
  if (sa.done_flag) { __threadfence_system(); __syncthreads(); if (threadIdx.x == 0) *sa.done_flag = sb ? sb[SB_S] : sa.done_seq; }
}

// step graph: the numbers of THIS call, set on the main stream right in front of the graph launch (by-value arguments of an ordinary launch)
struct StepBlockVals { long long v[SB_N]; };
__global__ void k_set_step_block(long long* sb, StepBlockVals v, long long* cnt, long long* marks, LinOps head, double* fault) {
  if (threadIdx.x < SB_N) sb[threadIdx.x] = v.v[threadIdx.x];
  __threadfence(); __syncthreads();
  if (threadIdx.x == 0 && cnt) lin_run_ops(head, cnt, marks, fault);      // ... and the head of the main chain: its step counter, the marks the side chains start on
}
// last kernel of a side chain: its stream has finished everything in front of it
__global__ void k_evb_mark(long long* p, long long seq, const long long* sb = nullptr) { __hip_atomic_store(p, sb ? sb[SB_A] : seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }      // sb: a captured launch - the number is the step block's

__global__ void k_sum_rows(int nrow, int ncol, int stride, const double* __restrict__ part, double* __restrict__ out) {
  const int r = blockIdx.x;
  __shared__ double red[256];
  double s = 0;
  for (int k = threadIdx.x; k < ncol; k += 256) s += part[(size_t)r * stride + k];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) { if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st]; __syncthreads(); }
  if (threadIdx.x == 0) out[r] = red[0];
}

// ---------------------------------------------------------------------------------------------------------------- host side
struct EvbWork {   // persistent device scratch of one context
  DBuf<EvbTablesDev> d_tabs; DBuf<double> d_exch;
  DBuf<double> d_part, d_E, d_dVspec; DBuf<int> d_sbase, d_counter; DBuf<double> d_sw, d_sdw;
  DBuf<double2> d_grids; DBuf<RegionRec> d_region; DBuf<int> d_cand;
  DBuf<double> d_Fdelta;      // [n_diag][3N] real-space deltas of the diabats, accumulated on stream2 beside the reciprocal space
  hipEvent_t ev_pairs = nullptr, ev_rs = nullptr, ev_fork2 = nullptr, ev_join2 = nullptr, ev_E = nullptr, ev_fetch = nullptr, ev_cpl = nullptr, ev_flush = nullptr, ev_fetchA = nullptr; bool delta_dirty = true;
  DBuf<double> d_F11, d_pack, d_com_atom;
  DBuf<Stencil> d_stencils; DBuf<double> d_Elq, d_ow, d_odw; DBuf<int> d_obase;      // linear-response reciprocal path: stencil records, {E_lin[MAX_DIABATS], E_quad[MAX_DIABATS], 0}
  int last_nst = 0; bool last_linear = false, elq_zeroed = false;
  hipStream_t stream2 = nullptr; hipEvent_t ev_in = nullptr, ev_region = nullptr;
  RegionRec* h_region = nullptr; int* h_count = nullptr;          // pinned
  long long* h_flag = nullptr; long long seq_A = 0, seq_R = 0, seq_B = 0;      // pinned: sequence number of the last published fetch A
  DBuf<double> d_coef; double* h_solve = nullptr; size_t h_solve_cap = 0; hipEvent_t ev_solve = nullptr; long long seq_S = 0; hipEvent_t ev_bterms = nullptr; bool side_by_mark = false; const long long* blend_mark = nullptr; long long blend_mark_seq = 0;      // device-side Hamiltonian solve (k_evb_solve)
  char* h_desc = nullptr; DBuf<char> d_desc; size_t desc_cap = 0;   // units | couplings | recip atoms, one pinned block -> one upload
  double* h_E = nullptr; size_t hE_cap = 0;                         // pinned
  int grids_cap = 0;
  bool tabs_uploaded = false;
  double screen_r = 0.0;
  // what the last force call left behind, for msevb_evb_partial_blend on the fast path (virtual-rank tests)
  int last_ngrid = 0, last_nra = 0, last_grid_of[MAX_DIABATS]; const RecipAtom* last_d_ra = nullptr; bool last_fast = false;
  bool counters_dirty = true, dv_dirty = true, mol_deferred = false;
  hipEvent_t ev_chains = nullptr;      // environment chain + couplings done (recorded on the descriptor stream)
  bool side_by_A = false;        // every side chain the blend phase needs has ended once fetch A is published (spinning publish kernel, prefill in front of it,
                                 // no kernel behind it): the host, which has seen fetch A, enqueues the blend phase without a GPU-side wait for ev_side
  bool side_wo_mol = false;      // ev_side does not cover the per-molecule kernel (aux3, c->ev_join2): only k_evb_blend reads what that kernel adds, so only it waits
  bool side_valid = false; hipEvent_t ev_side = nullptr;      // recorded on the front-end stream once it has waited for every other side chain: the ONE event the blend phase waits for
  bool pack_prefilled = false; hipEvent_t ev_prefill = nullptr;      // the blended charge grid starts as a copy of Q_1, made while the host waits for the energies
  std::vector<RegionRec> region, region_raw; std::vector<long long> region_key;
  std::vector<UnitDesc> units; std::vector<CouplingDesc> cps, cps_sorted; std::vector<RecipAtom> ras;      // descriptors of the current force call
  // speculative step (SpecPlan above): what the last step that enumerated the tree itself left behind
  SpecPlan plan; bool plan_valid = false, plan_fits = true; long plan_epoch = -1;
  DBuf<SpecPlan> d_plan; DBuf<double> d_sphere; DBuf<int> d_check; DBuf<unsigned long long> d_refresh_scratch; DBuf<long long> d_chain_done;
  int pl_D = 0, pl_plog[MAX_DIABATS][MAX_CHAIN][5], pl_parent[MAX_DIABATS], pl_ngrid = 0, pl_ncp_big = 0;
  std::vector<std::vector<int>> pl_units_of; std::vector<int> pl_grid_of, pl_cp_of; std::vector<double> pl_eref_delta;
  long spec_steps = 0, spec_misses = 0;
  // step graph (ctx.hpp: StepRec): the enqueue part of a speculative call captured once per diabat tree and replayed with ONE launch; two entries: calls whose refresh of the
  // sorted arrays rode in the integrator's first kernel (MD loop) and calls that refresh themselves (force calls through the host-array entry points)
  struct StepGraph {      // four linear graphs, one per stream (ctx.hpp: LinCap); `need` is what the learn pass found (which event records another stream waits for)
    hipGraphExec_t exec[LIN_NST] = {nullptr, nullptr, nullptr, nullptr}; unsigned long long hash = 0; int n = 0; long plan_serial = -1, topo_epoch = -1, alloc_epoch = -1;
    bool learned = false; long learn_serial = -1, learn_topo = -1; unsigned long long need[60]; int nev = 0;
    std::vector<std::pair<const void*, unsigned long long>> trace;      // (MSEVB_GRAPH_DEBUG)
    LinOps tail;      // ... and its closing waits (integrator's second kernel / a kernel of their own)
    LinOps head;      // the main chain's operations in front of its first kernel (step counter, the marks everything else starts on): done by the step-block kernel
    bool have() const { return exec[0] != nullptr; }
    void drop() { for (auto& e : exec) if (e) { (void)hipGraphExecDestroy(e); e = nullptr; } }
  };
  // diabat enumeration / descriptor construction (evb_plan.hpp): static tables, the region's molecule order, scratch frames of the search
  PlanTables ptab; std::vector<int> region_order; std::vector<PlanFrame> plan_frames; PlanWork plan_work;
  // ... on the device (k_evb_plan): tables, scratch, the summary in pinned memory; solve_desc_dirty: the host-built part of the descriptor block (SolveDesc) has to follow the device-built one
  DBuf<PlanTables> d_ptab; PlanTables ptab_dev; bool ptab_dev_valid = false; DBuf<int> d_plan_order, d_plan_near, d_plan_cnt, d_plan_shell; DBuf<CouplingDesc> d_cps_tmp;
  PlanSummary* h_plan_sum = nullptr; long long seq_P = 0; hipEvent_t ev_plan = nullptr; bool solve_desc_dirty = false, solve_desc_on_device = false, device_plan_off_once = false;
  StepGraph sg[2]; hipEvent_t ev_gjoin[3] = {nullptr, nullptr, nullptr}; DBuf<long long> d_sb, d_lin;      // d_lin: the linear graphs' step counters [4] and marks [64]
  LinCap lin; long plan_serial = 0; bool graph_broken = false, graph_skip_once = false; int graph_mismatch_run = 0;
};
// host-side section timing (MSEVB_HOST_TIMING=1): where the CPU thread spends the step
struct HostTimes { double t[12] = {0}; long n = 0; };
HostTimes g_ht;
inline double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }


const RegionRec* find_region(const EvbWork* w, int mol) {
  auto it = std::lower_bound(w->region.begin(), w->region.end(), mol, [](const RegionRec& r, int m) { return r.mol < m; });
  return (it != w->region.end() && it->mol == mol) ? &*it : nullptr;
}

static int plan_err_code(int pe) { return (pe == PE_NO_CONJUGATE || pe == PE_REORDER || pe == PE_NO_HEAVY || pe == PE_NO_COUPLING_INDEX) ? MSEVB_ERR_ARG : MSEVB_ERR_LIMIT; }
// the enumeration's static tables from the context (PlanTables, evb_plan.hpp)
int fill_plan_tables(const msevb_ctx* c, PlanTables& tb) {
  memset(&tb, 0, sizeof tb);
  tb.nmt = c->nmt; tb.max_chain = c->max_chain; tb.rank = c->rank; tb.n_ranks = c->n_ranks;
  if (c->nmt > MAX_MOLTYPES) return -1;
  for (int t = 0; t < c->nmt; ++t) {
    const MolTypeHost& mt = c->h_mt[t];
    tb.mt_natom[t] = mt.natom;
    for (int k = 0; k < MAX_MOL_ATOMS; ++k) {
      tb.mt_atype[t][k] = k < (int)mt.atype.size() ? mt.atype[k] : -1;
      tb.reactive_p[t][k] = k < (int)mt.reactive_protons.size() ? mt.reactive_protons[k] : 0;
      tb.reactive_b[t][k] = k < (int)mt.reactive_basic.size() ? mt.reactive_basic[k] : 0;
    }
    const int nb = (int)mt.bonds.size() / 2;
    if (nb > MAX_BONDED) return -1;
    tb.mt_nbond[t] = nb;
    for (int b = 0; b < nb; ++b) { tb.mt_bond[t][b][0] = mt.bonds[2 * b]; tb.mt_bond[t][b][1] = mt.bonds[2 * b + 1]; }
  }
  memcpy(tb.conj_mol, c->e_conj_mol, sizeof tb.conj_mol); memcpy(tb.proton_type, c->e_proton_type, sizeof tb.proton_type); memcpy(tb.heavy_type, c->e_heavy_type, sizeof tb.heavy_type);
  memcpy(tb.conj_atype, c->e_conj_atype, sizeof tb.conj_atype);
  for (int t = 0; t < MAX_TYPES; ++t) tb.atype_chg[t] = t < (int)c->h_atype_chg.size() ? c->h_atype_chg[t] : 0.0;
  memcpy(tb.e_ref, c->e_ref, sizeof tb.e_ref); memcpy(tb.exch_proton, c->e_exch_proton, sizeof tb.exch_proton);
  tb.n_cp = c->n_cp; memcpy(tb.cp_types, c->cp_types, sizeof tb.cp_types); memcpy(tb.cp_func, c->cp_func, sizeof tb.cp_func); memcpy(tb.cp_par, c->cp_par, sizeof tb.cp_par);
  for (int d = 0; d < 3; ++d) { tb.box[d] = c->box[d]; tb.inv_box[d] = 1.0 / c->box[d]; }
  return 0;
}

// symmetric Jacobi diagonalisation (jacobi, src/general_routines.f90:2013-2088)
int jacobi_h(int n, std::vector<double>& a, std::vector<double>& d, std::vector<double>& v) {
  std::vector<double> b(n), z(n, 0.0);
  auto A = [&](int i, int j) -> double& { return a[(size_t)i * n + j]; };
  auto V = [&](int i, int j) -> double& { return v[(size_t)i * n + j]; };
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) V(i, j) = i == j ? 1.0 : 0.0;
  for (int i = 0; i < n; ++i) { b[i] = A(i, i); d[i] = b[i]; }
  auto rot = [&](double& x, double& y, double s, double tau) { const double g = x, h = y; x = g - s * (h + g * tau); y = h + s * (g - h * tau); };
  for (int it = 1; it <= 50; ++it) {
    double sm = 0.0;
    for (int j = 0; j < n; ++j) for (int i = 0; i < j; ++i) sm += std::fabs(A(i, j));
    if (sm == 0.0) return 0;
    const double tresh = it < 4 ? 0.2 * sm / double(n * n) : 0.0;
    for (int ip = 0; ip < n - 1; ++ip) for (int iq = ip + 1; iq < n; ++iq) {
      const double g = 100.0 * std::fabs(A(ip, iq));
      if (it > 4 && std::fabs(d[ip]) + g == std::fabs(d[ip]) && std::fabs(d[iq]) + g == std::fabs(d[iq])) A(ip, iq) = 0.0;
      else if (std::fabs(A(ip, iq)) > tresh) {
        double h = d[iq] - d[ip], t;
        if (std::fabs(h) + g == std::fabs(h)) t = A(ip, iq) / h;
        else { const double theta = 0.5 * h / A(ip, iq); t = 1.0 / (std::fabs(theta) + std::sqrt(1.0 + theta * theta)); if (theta < 0.0) t = -t; }
        const double cs = 1.0 / std::sqrt(1 + t * t), s = t * cs, tau = s / (1.0 + cs);
        h = t * A(ip, iq);
        z[ip] -= h; z[iq] += h; d[ip] -= h; d[iq] += h; A(ip, iq) = 0.0;
        for (int k = 0; k < ip; ++k) rot(A(k, ip), A(k, iq), s, tau);
        for (int k = ip + 1; k < iq; ++k) rot(A(ip, k), A(k, iq), s, tau);
        for (int k = iq + 1; k < n; ++k) rot(A(ip, k), A(iq, k), s, tau);
        for (int k = 0; k < n; ++k) rot(V(k, ip), V(k, iq), s, tau);
      }
    }
    for (int i = 0; i < n; ++i) { b[i] += z[i]; d[i] = b[i]; z[i] = 0.0; }
  }
  return -1;
}

}  // namespace

static BoxDev mkbox(const msevb_ctx* c) { BoxDev b; for (int d = 0; d < 3; ++d) { b.L[d] = c->box[d]; b.invL[d] = 1.0 / c->box[d]; } return b; }

// slot helpers: diagonal store has 1 + R*S slots (0 = principal), off-diagonal store R*S; diabat d (0-based, d>=1)
static inline int slots_per_rank(const msevb_ctx* c, int D) { return D <= 1 ? 0 : (D - 1 + c->n_ranks - 1) / c->n_ranks; }
static inline int off_slot(const msevb_ctx* c, int D, int d) { return ((d - 1) % c->n_ranks) * slots_per_rank(c, D) + (d - 1) / c->n_ranks; }
static inline int diag_slot(const msevb_ctx* c, int D, int d) { return d == 0 ? 0 : 1 + off_slot(c, D, d); }
static inline bool owned(const msevb_ctx* c, int d);
// terms of this rank's share of the Hellmann-Feynman blend; returns the weight of the principal force F11 in it (rank 0 carries it)
static double blend_terms(const msevb_ctx* c, int D, const double* ev, BlendTerms& bt, int* ti = nullptr, int* tj = nullptr) {
  bt.n = 0;      // ev == nullptr: only the structure (slots, and the (i,j) of every term for k_evb_solve); the coefficients come from the device
  for (int i = 0; i < D; ++i) for (int j = i; j < D; ++j) {
    const bool diag = i == j && (i == 0 ? c->rank == 0 : owned(c, i)), off = i != j && c->parent[j] == i && owned(c, j);
    if (!diag && !off) continue;
    bt.coef[bt.n] = !ev ? 0.0 : diag ? ev[i] * ev[j] : 2.0 * ev[i] * ev[j] * -c->cpl_A[j];
    if (ti) { ti[bt.n] = i; tj[bt.n] = j; }
    bt.slot[bt.n++] = diag ? diag_slot(c, D, i) : -off_slot(c, D, j) - 1;
  }
  return (c->rank == 0 || (c->shard_principal && c->n_ranks > 1)) ? 1.0 : 0.0;      // sharded principal: every rank holds a part of F11
}
// The host waits for small results (neighbourhood record, energies) by spinning on a sequence number the producing kernel writes to
// pinned memory behind its data: hipEventSynchronize sleeps on an interrupt and needs 50-100 us to wake up, a spin sees the store in
// ~2 us.  The event stays the fallback (a kernel that failed never writes; the event reports the error).
static int spin_wait(volatile long long* flag, long long seq, hipEvent_t ev, hipStream_t graph_stream = nullptr) {      // graph_stream != null: the producer ran inside a launched graph - `ev` was never recorded, the device is what can be waited for
  const auto t0 = std::chrono::steady_clock::now();
  long spins = 0;
  while (*flag != seq) {
    __builtin_ia32_pause();
    if ((++spins & 0xfff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200)) break;
  }
  if (*flag != seq && graph_stream) return (hipDeviceSynchronize() == hipSuccess && *flag == seq) ? 0 : -1;
  if (*flag != seq && hipEventSynchronize(ev) != hipSuccess) return -1;
  return 0;
}

// Fast-path blend of THIS rank's share for the ground-state vector ev: one gather of the blended potential of its grids (+ the principal
// one on rank 0), the special atoms' corrections into the delta store, then the blend of delta / off-diagonal stores.  c->d_force holds
// (this rank's part of) F11 on entry and the rank's partial Hellmann-Feynman force on exit.
static int fast_blend_phase(msevb_ctx* c, EvbWork* w, int D, const double* ev, const int* grid_of, int ngrid, int nra, const RecipAtom* d_ra, const double* dcoef = nullptr);

static inline bool owned(const msevb_ctx* c, int d) { return d == 0 || ((d - 1) % c->n_ranks) == c->rank; }

extern "C" int msevb_comm_allgather_evb(msevb_ctx* c, int with_energies);   // comm.hip
extern "C" int msevb_evb_payload_size(int n_diabat, int n_ranks);
int msevb_comm_allgather_device(msevb_ctx* c, int P, hipStream_t st);      // comm.hip: in-place all-gather of the payload slots already in d_comm_h
int msevb_comm_allreduce_e8_device(msevb_ctx* c, hipStream_t st);          // comm.hip: the eight principal energy slots, device to device
int msevb_comm_wait_flag(msevb_ctx* c, volatile long long* flag, long long seq, hipStream_t st, const char* what);      // comm.hip: the host's wait behind a collective, with the communicator's time limit

// dcoef: coefficients in device memory {c0, grid coefficients, blend terms} written by k_evb_solve (then ev is not needed)
static int fast_blend_phase(msevb_ctx* c, EvbWork* w, int D, const double* ev, const int* grid_of, int ngrid, int nra, const RecipAtom* d_ra, const double* dcoef) {
  const int K = c->K;
  const size_t K3 = (size_t)K * K * K, n3 = 3 * (size_t)c->N;
  BlendTerms bt; memset(&bt, 0, sizeof bt);      // (by-value kernel arguments are zero-filled, padding and unused tails included: captured / replayed launches are compared by their bytes)
  double w11 = blend_terms(c, D, ev, bt);
  const bool w11_was_one = w11 != 0.0;
  if (w11 == 0.0) {      // replicated (not sharded) principal diabat on a rank other than 0: F11 counts on rank 0 only, the gather below adds to zero
    if (c->overlap && w->side_valid) { HIPCHK(hipStreamWaitEvent(c->stream, w->ev_side, 0)); if (w->side_wo_mol) HIPCHK(hipStreamWaitEvent(c->stream, c->ev_join2, 0)); }      // (the per-molecule kernel still adds to it)
    else if (c->overlap && w->mol_deferred) HIPCHK(hipStreamWaitEvent(c->stream, c->ev_join2, 0));
    HIPCHK(hipMemsetAsync(c->d_force.p, 0, n3 * 8, c->stream));
    if (c->fsorted_pending) { HIPCHK(hipMemsetAsync(c->d_fsorted.p, 0, n3 * 8, c->stream)); c->fsorted_pending = false; }
    w11 = 1.0;
  }
  // reciprocal space of the blend in ONE gather: blended potential of this rank's diabats (+ the principal one on rank 0) ...
  GridCoefs gc; memset(&gc, 0, sizeof gc); gc.n = ngrid; gc.c0 = 0.0;
  if (ev) { gc.c0 = c->rank == 0 ? ev[0] * ev[0] : 0.0; for (int d = 1; d < D; ++d) if (grid_of[d] >= 0) gc.c[grid_of[d]] = ev[d] * ev[d]; }
  const bool one_wait = c->overlap && w->side_valid;      // every side chain of this force call behind ONE event
  // ASIDE (default): the blend terms are added from the grid stream (k_evb_blend_terms: atomics) while this stream transforms and gathers; the gather folds the
  // pair loop's accumulator in; the main stream joins behind the gather.  Needs: the single-join schedule, a gather call on this rank, and - when the
  // pair force still sits in the sorted-order accumulator - the gather kernel that can fold it.
  const bool gather_here = w->last_linear ? (ngrid > 0 || c->rank == 0) : true;
  const bool aside = c->evb_blend_aside && one_wait && w11_was_one && (c->fsorted_pending ? (gather_here && pme_gather_can_fold(c)) : true);
  bool folded = false;
  if (aside) {
    if (!w->ev_bterms) HIPCHK(hipEventCreateWithFlags(&w->ev_bterms, hipEventDisableTiming));
    const hipStream_t sb = c->aux;      // the grid stream: the energy fetch (and the device-side solve, whose coefficients this kernel reads) precede it there
    if (!w->side_by_A && !w->side_by_mark) HIPCHK(hipStreamWaitEvent(sb, w->ev_side, 0));
    hipLaunchKernelGGL(k_evb_blend_terms, dim3(std::min<size_t>(1024, (n3 + 255) / 256)), dim3(256), 0, sb, n3, bt, dcoef ? dcoef + 1 + ngrid : (const double*)nullptr, w->d_Fdelta.p, c->d_Foff.p, n3, c->d_force.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(w->ev_bterms, sb));
  }
  double* const fold = (aside && c->fsorted_pending) ? c->d_fsorted.p : (double*)nullptr;
  auto join_before_blend = [&]() -> int {      // everything else that adds to the force array has to be in before the kernel that blends (k_evb_blend, or the fused gather)
    if (c->overlap && !one_wait) { HIPCHK(hipStreamWaitEvent(c->stream, w->ev_join2, 0)); if (ngrid) HIPCHK(hipStreamWaitEvent(c->stream, w->ev_rs, 0)); }
    if (c->overlap && !one_wait && w->mol_deferred) HIPCHK(hipStreamWaitEvent(c->stream, c->ev_join2, 0));      // the principal diabat's per-molecule forces (aux3, behind the coupling chain)
    if (one_wait && w->side_wo_mol) HIPCHK(hipStreamWaitEvent(c->stream, c->ev_join2, 0));      // ... which the ONE event left out: patch, solve and gather of the blend phase run beside that kernel
    return MSEVB_OK;
  };
  if (one_wait && !w->side_by_A && !w->side_by_mark) HIPCHK(hipStreamWaitEvent(c->stream, w->ev_side, 0));      // (side_by_A: whoever enqueues this phase has seen fetch A)
  else if (c->overlap && ngrid && !one_wait) HIPCHK(hipStreamWaitEvent(c->stream, w->ev_E, 0));           // convolved diabat grids / grid energies (aux)
  stamp(c, 8, c->stream);
  if (w->d_pack.alloc(K3)) FAIL(MSEVB_ERR_HIP, "hipMalloc failed (blended potential)");
  const bool patch_waits = w->blend_mark && w->last_linear && ngrid > 0 && fft_lds_usable(c);      // k_evb_blend_patch is the phase's first kernel and waits itself
  if (w->blend_mark && !patch_waits) hipLaunchKernelGGL(k_evb_wait_mark, dim3(1), dim3(64), 0, c->stream, w->blend_mark, w->blend_mark_seq, c->d_energies.p + S_TIMEOUT, (const long long*)c->step_sb);
  if (w->last_linear) {
    // blended CHARGE grid Q_1 (rank 0) + sum_d c_d^2 DQ_d -> one transform pair -> blended potential theta_1 + sum_d c_d^2 G * DQ_d
    // (sum_d c_d^2 = 1 over all ranks, so the principal potential enters once, on rank 0)
    if (!ngrid) {
      if (c->rank == 0) CHK(pme_gather(c, nullptr, c->d_chg.p, c->d_force_recip.p, c->d_force.p, true, c->d_theta.p, fold, &folded));
    } else {
      if (w->d_grids.alloc_room(K3)) FAIL(MSEVB_ERR_HIP, "hipMalloc failed (blended charge grid)");
      if (fft_lds_can_patch(c) && c->evb_fft_patch) {      // 64^3: the forward transform applies the stencils to the planes of Q_1 itself - no copy of Q_1, no patch kernel
        FftPatch P; memset(&P, 0, sizeof P); P.nst = w->last_nst; P.st = w->d_stencils.p; P.dc = dcoef;
        for (int g = 0; g < MAX_DIABATS; ++g) P.c[g] = (!dcoef && g < ngrid) ? gc.c[g] : 0.0;
        P.wait_mark = patch_waits ? w->blend_mark : (const long long*)nullptr; P.wait_seq = w->blend_mark_seq; P.wait_sb = patch_waits ? (const long long*)c->step_sb : (const long long*)nullptr; P.fault = c->d_energies.p + S_TIMEOUT;
        CHK(fft_lds_solve(c, c->rank == 0 ? c->d_Q.p : (const double*)nullptr, w->d_grids.p, w->d_pack.p, nullptr, nullptr, c->stream, &P));
        w->pack_prefilled = false;
      } else if (fft_lds_usable(c)) {      // real grid in, real potential out, in place in d_pack (d_grids is the transform's complex work space)
        if (w->pack_prefilled) { if (c->overlap && (!one_wait || w->side_by_A) && !w->side_by_mark) HIPCHK(hipStreamWaitEvent(c->stream, w->ev_prefill, 0)); }      // (went out before the host's turn)
        else if (c->rank == 0) hipLaunchKernelGGL(k_copy_grid, dim3(256), dim3(256), 0, c->stream, K3, (const double*)c->d_Q.p, w->d_pack.p);
        else HIPCHK(hipMemsetAsync(w->d_pack.p, 0, K3 * sizeof(double), c->stream));
        w->pack_prefilled = false;
        hipLaunchKernelGGL(k_evb_blend_patch, dim3((w->last_nst + 3) / 4), dim3(256), 0, c->stream, w->last_nst, w->d_stencils.p, gc, dcoef, K, w->d_pack.p, 1,
                           patch_waits ? w->blend_mark : (const long long*)nullptr, w->blend_mark_seq, c->d_energies.p + S_TIMEOUT, patch_waits ? (const long long*)c->step_sb : (const long long*)nullptr);
        HIPCHK(hipGetLastError());
        CHK(fft_lds_solve(c, w->d_pack.p, w->d_grids.p, w->d_pack.p, nullptr, nullptr, c->stream));
      } else {
        if (c->rank == 0) hipLaunchKernelGGL(k_evb_grid_init, dim3(2048), dim3(256), 0, c->stream, K3, 1, c->d_Q.p, w->d_grids.p);
        else HIPCHK(hipMemsetAsync(w->d_grids.p, 0, K3 * sizeof(double2), c->stream));
        hipLaunchKernelGGL(k_evb_blend_patch, dim3((w->last_nst + 3) / 4), dim3(256), 0, c->stream, w->last_nst, w->d_stencils.p, gc, dcoef, K, (double*)w->d_grids.p, 2, (const long long*)nullptr, 0LL, (double*)nullptr, (const long long*)nullptr);
        HIPCHK(hipGetLastError());
        int nbk = 0;
        hipStream_t keep = c->pme_st; c->pme_st = nullptr;      // on the main stream
        const int rc_conv = pme_convolve(c, w->d_grids.p, 1, w->d_part.p, 1024, &nbk);      // (the energy partials of this transform are not used)
        c->pme_st = keep;
        CHK(rc_conv);
        hipLaunchKernelGGL(k_grid_real, dim3(1024), dim3(256), 0, c->stream, K3, w->d_grids.p, w->d_pack.p);
      }
      CHK(pme_gather(c, nullptr, c->d_chg.p, c->d_force_recip.p, c->d_force.p, true, w->d_pack.p, fold, &folded));
    }
  } else {
    hipLaunchKernelGGL(k_evb_blend_grids, dim3(1024), dim3(256), 0, c->stream, K3, gc, dcoef, c->d_theta.p, w->d_grids.p, w->d_pack.p);
    CHK(pme_gather(c, nullptr, c->d_chg.p, c->d_force_recip.p, c->d_force.p, true, w->d_pack.p, fold, &folded));
  }
  // ... the special atoms' own-stencil corrections per diabat are in the delta store already (k_evb_recip_special ran at the end of the
  // diabat grid chain, before the host's turn); the blend consumes the store and re-zeroes it
  (void)nra; (void)d_ra;
  CHK(join_before_blend());
  if (aside) {
    if (fold && !folded) FAIL(MSEVB_ERR_STATE, "blend phase: the gather that was to fold the pair loop's accumulator in did not run");
    HIPCHK(hipStreamWaitEvent(c->stream, w->ev_bterms, 0));      // the terms went in long ago (15 us of streaming beside 60 us of transform + gather)
  } else {
  hipLaunchKernelGGL(k_evb_blend, dim3(std::min<size_t>(2048, (n3 + 255) / 256)), dim3(256), 0, c->stream, n3, bt, dcoef ? dcoef + 1 + ngrid : (const double*)nullptr, w11, w->d_Fdelta.p, c->d_Foff.p, n3, c->d_force.p, 1,
                     c->fsorted_pending ? c->d_fsorted.p : (double*)nullptr, c->d_inv_perm.p, c->N);
  }
  c->fsorted_pending = false;
  stamp(c, 9, c->stream);
  HIPCHK(hipGetLastError());
  return MSEVB_OK;
}

// the MS-EVB parameter tables of the device kernels (once per context / msevb_set_evb)
static int upload_evb_tables(msevb_ctx* c, EvbWork* w) {
  if (w->tabs_uploaded) return MSEVB_OK;
  if (w->d_tabs.alloc(1) || w->d_exch.alloc(MAX_TYPES)) FAIL(MSEVB_ERR_HIP, "hipMalloc failed (EVB tables)");
  {
    EvbTablesDev tb; memset(&tb, 0, sizeof tb);
    tb.n_da = c->n_da; tb.n_pa = c->n_pa;
    memcpy(tb.da_types, c->da_types, sizeof tb.da_types); memcpy(tb.da_par, c->da_par, sizeof tb.da_par);
    memcpy(tb.pa_types, c->pa_types, sizeof tb.pa_types); memcpy(tb.pa_par, c->pa_par, sizeof tb.pa_par);
    memcpy(tb.exch_atomic, c->e_exch_atomic, sizeof tb.exch_atomic);
    for (int t = 0; t < c->nmt && t < MAX_MOLTYPES; ++t) for (int k = 0; k < MAX_MOL_ATOMS; ++k) {
      tb.reactive_p[t][k] = k < (int)c->h_mt[t].reactive_protons.size() ? c->h_mt[t].reactive_protons[k] : 0;
      tb.reactive_b[t][k] = k < (int)c->h_mt[t].reactive_basic.size() ? c->h_mt[t].reactive_basic[k] : 0;
    }
    double rmax = 0.0;
    for (int k = 0; k < c->n_da; ++k) rmax = std::max(rmax, c->da_par[k][5]);
    for (int k = 0; k < c->n_pa; ++k) rmax = std::max(rmax, c->pa_par[k][4] + 3.0);
    tb.screen2 = (rmax + 0.5) * (rmax + 0.5);
    w->screen_r = rmax + 0.5;
    HIPCHK(hipMemcpy(w->d_tabs.p, &tb, sizeof tb, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(w->d_exch.p, c->e_exch_atomic, sizeof(double) * MAX_TYPES, hipMemcpyHostToDevice));
    w->tabs_uploaded = true;
  }
  return MSEVB_OK;
}

static int evb_force_energy_body(msevb_ctx* c, double* epot_out);
static int evb_force_energy_once(msevb_ctx* c, double* epot_out) {
  try { return evb_force_energy_body(c, epot_out); }
  catch (const std::exception& e) { FAIL(MSEVB_ERR_LIMIT, std::string("EVB host side: ") + e.what()); }      // a flat container ran out of room (never across the C ABI)
}
static int evb_force_energy_body(msevb_ctx* c, double* epot_out) {
  static const bool ht_on = getenv("MSEVB_HOST_TIMING") != nullptr;
  double ht_last = ht_on ? now_us() : 0.0;
  auto HT = [&](int k) { if (ht_on) { const double t = now_us(); g_ht.t[k] += t - ht_last; ht_last = t; } };

  if (!c->evb_set) FAIL(MSEVB_ERR_STATE, "EVB tables not set (msevb_set_evb)");
  if (!c->evb_work) c->evb_work = new EvbWork();
  EvbWork* w = (EvbWork*)c->evb_work;
  const int N = c->N, M = c->M, K = c->K;
  const size_t K3 = (size_t)K * K * K, n3 = 3 * (size_t)N;
  const BoxDev box = mkbox(c);
  HT(0);
  // ---- 1. neighbourhood of the hydronium -> host on a side stream (replaces the all-molecule scans of src/ms_evb.f90:724);
  //         it only needs the positions, so it overlaps with the principal diabat's force kernels
  if (!w->stream2) {
    w->stream2 = c->aux2;
    HIPCHK(hipEventCreateWithFlags(&w->ev_in, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&w->ev_region, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&w->ev_fork2, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&w->ev_join2, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&w->ev_cpl, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&w->ev_flush, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&w->ev_fetchA, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&w->ev_solve, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&w->ev_E, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&w->ev_rs, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&w->ev_fetch, hipEventDisableTiming));
    HIPCHK(hipHostMalloc((void**)&w->h_region, REGION_FAST * sizeof(RegionRec), hipHostMallocDefault));
    HIPCHK(hipHostMalloc((void**)&w->h_count, 64, hipHostMallocDefault));
  }
  if (w->d_region.alloc(REGION_CAP) || w->d_counter.alloc(4) || w->d_cand.alloc(N)) FAIL(MSEVB_ERR_HIP, "hipMalloc failed (EVB region)");
  if (w->counters_dirty) {     // first call, or the previous call failed half-way: afterwards the two kernels re-arm each other's counter
    HIPCHK(hipMemsetAsync(w->d_counter.p, 0, 4 * sizeof(int), c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
  }
  struct Rearm { EvbWork* w; bool ok; ~Rearm() { w->counters_dirty = !ok; if (!ok) { w->dv_dirty = true; w->delta_dirty = true; } } } rearm{w, false};
  // a hop goes to a molecule whose centre of mass is within 5 A of the donor's (evb_first_solvation_cutoff): a chain of max_chain hops
  // stays inside 5 max_chain of the hydronium's
  const double Rreg = 5.0 * c->max_chain + 0.25;
  // The neighbourhood record is the first thing the host's turn needs, and everything the side chains do starts behind that turn: it goes
  // out on the MAIN stream, directly behind the integrator's first kernel (on a side stream it started one cross-queue event hop,
  // ~15 us, later; the pair loop, which it now precedes, has that much slack before the blend needs it)
  hipStream_t s_reg = c->overlap && c->evb_region_main ? c->stream : w->stream2;
  if (!w->h_flag) { HIPCHK(hipHostMalloc((void**)&w->h_flag, 64, hipHostMallocDefault)); memset(w->h_flag, 0, 64); }
  // SPECULATIVE STEP: the tree of the last enumeration is assumed to hold; k_evb_region checks that on the device while the host goes straight on
  const bool spec = c->evb_speculate && c->overlap && w->plan_valid && w->plan_epoch == c->topo_epoch && w->plan.hyd == c->hydronium && w->tabs_uploaded;
  if (!w->d_check.p) {
    if (w->d_plan.alloc(1) || w->d_sphere.alloc(4) || w->d_check.alloc(4) || w->d_refresh_scratch.alloc(2)) FAIL(MSEVB_ERR_HIP, "hipMalloc failed (EVB plan)");
    HIPCHK(hipMemsetAsync(w->d_check.p, 0, 4 * sizeof(int), w->stream2)); HIPCHK(hipMemsetAsync(w->d_refresh_scratch.p, 0, 16, w->stream2));      // (k_evb_refresh leaves the counters zeroed)
  }
  int* h_verdict = (int*)(w->h_flag + 4);
  *h_verdict = 0;
  // STEP GRAPH (ctx.hpp: StepRec, LinCap).  A speculative call that neither rebuilds its list nor samples the phase timers enqueues the same operations with the same arguments
  // as the last such call on this tree: its enqueue part is captured once - every stream as a linear graph of its own, the streams ordered by device-side marks -
  // and replayed with four hipGraphLaunch calls (~3 us each, profiles/r05_graph_probe.txt).
  // gmode 3 = this call learns which event records are waited for across streams (nothing issued, the call is repeated), 1 = this call captures (and launches),
  // 2 = this call replays (the host code below runs for its bookkeeping, the wrappers of ctx.hpp issue nothing and hash what they would have issued).
  if (c->stamps_on < 0) { const char* se = getenv("MSEVB_STAMPS"); c->stamps_on = se ? (se[0] == '2' ? 2 : 1) : 0; }
  const int fused_nb_in = c->refresh_fused_nb;
  const int gslot = c->refresh_fused_nb > 0 ? 1 : 0;
  EvbWork::StepGraph& G = w->sg[gslot];
  int gmode = 0;
  StepRec grec;
  LinCap& LC = w->lin;
  const long alloc_epoch0 = g_alloc_epoch;
  if (c->evb_step_graph && (c->evb_step_graph >= 2 || !c->nccl_comm) && g_live_contexts.load() == 1 && !w->graph_broken && spec && c->stamps_on != 1 && !c->nbr_check_pending && c->flag_verlet_list != 1 && c->list_valid && c->cells_valid &&
      !(c->timing && (c->timing_tick % c->timing_stride) == 0) && !w->solve_desc_dirty && !w->delta_dirty && !w->dv_dirty && !w->counters_dirty && w->d_check.p && c->stream && !c->evb_pair_first && !c->evb_molecules_after_pair &&
      fft_lds_usable(c) /* (rocFFT launches on streams of their own: not inside these captures) */ && w->d_Fdelta.p && w->tabs_uploaded) {
    if (w->graph_skip_once) w->graph_skip_once = false;      // (the call that repeats a mismatched replay)
    else {
      if (w->d_sb.alloc(SB_N)) FAIL(MSEVB_ERR_HIP, "hipMalloc failed (step block)");
      if (!w->d_lin.p) { if (w->d_lin.alloc(8 + 64)) FAIL(MSEVB_ERR_HIP, "hipMalloc failed (marks of the linear graphs)"); HIPCHK((hipMemset)(w->d_lin.p, 0, (8 + 64) * sizeof(long long))); HIPCHK(hipDeviceSynchronize()); }
      const bool have = G.have() && G.plan_serial == w->plan_serial && G.topo_epoch == c->topo_epoch && G.alloc_epoch == g_alloc_epoch;
      gmode = have ? 2 : (G.learned && G.learn_serial == w->plan_serial && G.learn_topo == c->topo_epoch) ? 1 : 3;
      if (gmode != 2) {
        G.drop();
        LC = LinCap();
        LC.st[0] = c->stream; LC.st[1] = c->aux; LC.st[2] = w->stream2; LC.st[3] = c->aux3;
        LC.cnt = w->d_lin.p; LC.marks = w->d_lin.p + 8; LC.fault = c->d_energies.p + S_TIMEOUT;
        if (gmode == 1) { LC.nev = 0; memcpy(LC.need, G.need, sizeof LC.need); LC.verbose = getenv("MSEVB_GRAPH_DEBUG") && c->evb_graph_captures == 0; }
      }
      if (gmode == 1) {
        int nb = 0;
        for (; nb < LIN_NST; ++nb) if (hipStreamBeginCapture(LC.st[nb], hipStreamCaptureModeRelaxed) != hipSuccess) break;
        if (nb < LIN_NST) {      // this runtime cannot capture here: end what has begun, no further attempts
          for (int b = 0; b < nb; ++b) { hipGraph_t g = nullptr; (void)hipStreamEndCapture(LC.st[b], &g); if (g) (void)hipGraphDestroy(g); }
          (void)hipGetLastError(); w->graph_broken = true; gmode = 0;
        } else for (int b = 0; b < LIN_NST; ++b) lin_add_op(LC, b, 0, 0, 0);      // first operation of every chain: the stream's step number advances
      }
    }
  }
  grec.mode = gmode;
  std::vector<std::pair<const void*, unsigned long long>> gtrace;
  if (gmode && getenv("MSEVB_GRAPH_DEBUG")) grec.trace = &gtrace;
  struct StepScope {      // leaves learn / capture / replay mode on every way out of this function
    msevb_ctx* c; EvbWork* w;
    ~StepScope() {
      if (t_step_rec && t_step_rec->mode == 1) {      // an error inside the capture
        for (int b = 0; b < LIN_NST; ++b) { hipGraph_t g = nullptr; (void)hipStreamEndCapture(w->lin.st[b], &g); if (g) (void)hipGraphDestroy(g); }
        (void)hipGetLastError(); w->graph_broken = true;
      }
      t_step_rec = nullptr; t_lin = nullptr; c->step_sb = nullptr;
    }
  } step_scope{c, w};
  if (gmode) { t_step_rec = &grec; c->step_sb = w->d_sb.p; t_lin = (gmode == 1 || gmode == 3) ? &LC : nullptr; }
  const long long* const sb = gmode ? w->d_sb.p : (const long long*)nullptr;
  const hipStream_t gsync = gmode ? (hipStream_t)-1 : (hipStream_t)nullptr;      // (spin_wait's fall-back when the flag does not come: the whole device - the producers sit in graphs of several streams)
  HIPCHK(hipEventRecord(w->ev_in, c->stream));      // (the first operation of a captured call: every side stream joins the capture through this event)
  HIPCHK(hipStreamWaitEvent(w->stream2, w->ev_in, 0));
  // speculative step with couplings: nothing on the device needs the neighbourhood record, only the host's verdict does - the kernel goes to the
  // coupling stream (idle until the descriptors are refreshed; its chain ends with a mark the energy fetch waits for, so the verdict is
  // in pinned memory before the host looks), and the descriptor chain starts with k_evb_refresh instead of behind 24-37 us of scanning
  // more than one rank: the device-side solve runs on the GATHERED Hamiltonian entries (k_evb_solve_gathered behind the all-gather, or - virtual ranks of the tests and of
  // tools/virtual_rank_times.py - behind the payloads msevb_comm_set_virtual_payload left in the all-gather's buffer), the blend phase and the all-reduce are pre-enqueued
  const bool dev_solve_mr_early = !c->evb_element_stores && (c->n_ranks > 1 || c->nccl_comm || !c->virt_payload.empty()) && c->evb_device_solve && c->evb_device_solve_ranks && c->overlap && (c->nccl_comm || !c->virt_payload.empty()) && c->evb_recip_linear && c->d_green.p;
  const bool dev_solve_early = !c->evb_element_stores && ((c->n_ranks == 1 && !c->nccl_comm && c->evb_device_solve) || dev_solve_mr_early), pub_spin_early = c->overlap && c->evb_publish_spin;
  const bool region_aside = spec && !c->evb_region_main && c->evb_region_aside && c->evb_publish_spin && !gmode && w->plan.n_cps > 0;
  if (region_aside) { s_reg = c->aux3; HIPCHK(hipStreamWaitEvent(c->aux3, w->ev_in, 0)); }
  hipLaunchKernelGGL(k_evb_region, dim3((M + 1023) / 1024), dim3(1024), 0, s_reg, M, c->hydronium, Rreg * Rreg, c->d_mol_natom.p, c->d_mol_atoms.p, c->d_xyz.p,
                     c->d_mass.p, box, w->d_counter.p, w->d_region.p, REGION_CAP, w->d_counter.p + 1, w->h_region, REGION_FAST,
                     w->d_counter.p + 2, w->h_count, (volatile long long*)(w->h_flag + 1), (++w->seq_R, gmode ? 0LL : w->seq_R),
                     spec ? (const SpecPlan*)w->d_plan.p : (const SpecPlan*)nullptr, (const EvbTablesDev*)w->d_tabs.p, (const int*)c->d_mol_type.p, w->d_check.p,
                     region_aside ? h_verdict : (int*)nullptr, sb);
  HIPCHK(hipEventRecord(w->ev_region, s_reg));
  // ---- 1b. f2: a call that has to enumerate the tree does so ON THE DEVICE (k_evb_plan: search, descriptors, the plan of the speculative steps) and is then repeated as a
  //          speculative call - the descriptors are on the device already, the host has only read the summary (sizes for its launches, the tree for the Hamiltonian it reports).
  //          evb_device_plan = 0: the host runs the same source (evb_plan.hpp) on the neighbourhood record, as rounds 1-4 did.
  if (!spec && c->evb_device_plan && c->evb_speculate && c->overlap && !w->device_plan_off_once && !c->evb_element_stores) {
    CHK(upload_evb_tables(c, w));
    PlanTables& tbh = w->ptab;
    if (fill_plan_tables(c, tbh)) FAIL(MSEVB_ERR_LIMIT, "EVB: molecule-type tables exceed the fixed sizes of the enumeration");
    if (w->d_ptab.alloc(1) || w->d_plan_order.alloc(REGION_CAP) || w->d_plan_near.alloc((size_t)3 * PLAN_MAX_PAIRS * REGION_CAP) || w->d_plan_cnt.alloc(REGION_CAP) || w->d_cps_tmp.alloc(MAX_DIABATS) || w->d_plan_shell.alloc((size_t)65 * REGION_CAP))
      FAIL(MSEVB_ERR_HIP, "hipMalloc failed (device-side plan)");
    if (!w->h_plan_sum) { HIPCHK(hipHostMalloc((void**)&w->h_plan_sum, sizeof(PlanSummary), hipHostMallocDefault)); memset(w->h_plan_sum, 0, sizeof(PlanSummary)); HIPCHK(hipEventCreateWithFlags(&w->ev_plan, hipEventDisableTiming)); }
    if (!w->ptab_dev_valid || memcmp(&w->ptab_dev, &tbh, sizeof tbh) != 0) {      // (rank layout, box or tables changed)
      HIPCHK(hipMemcpyAsync(w->d_ptab.p, &tbh, sizeof tbh, hipMemcpyHostToDevice, s_reg)); HIPCHK(hipStreamSynchronize(s_reg));
      w->ptab_dev = tbh; w->ptab_dev_valid = true;
    }
    if (w->desc_cap < (size_t)512 * 1024) {      // room for a few dozen descriptors to start with; k_evb_plan says when it needs more
      HIPCHK(hipDeviceSynchronize());
      if (w->h_desc) HIPCHK(hipHostFree(w->h_desc));
      w->desc_cap = (size_t)512 * 1024;
      HIPCHK(hipHostMalloc((void**)&w->h_desc, w->desc_cap, hipHostMallocDefault));
      if (w->d_desc.alloc(w->desc_cap)) FAIL(MSEVB_ERR_HIP, "hipMalloc failed (EVB descriptors)");
    }
    const PlanSummary* ps = w->h_plan_sum;
    for (int attempt = 0; attempt < 3; ++attempt) {
      PlanArgs pa; memset(&pa, 0, sizeof pa);
      pa.tb = w->d_ptab.p; pa.mol_type = c->d_mol_type.p; pa.mol_natom = c->d_mol_natom.p; pa.mol_atoms = c->d_mol_atoms.p; pa.atype = c->d_atype.p; pa.chg = c->d_chg.p; pa.mass = c->d_mass.p;
      pa.region = w->d_region.p; pa.region_count = w->d_counter.p; pa.region_cap = REGION_CAP; pa.hydronium = c->hydronium;
      pa.order = w->d_plan_order.p; pa.pair_tab = w->d_plan_near.p; pa.pair_cnt = w->d_plan_cnt.p; pa.near_tab = w->d_plan_shell.p; pa.near_cnt = w->d_plan_shell.p + (size_t)64 * REGION_CAP;
      const size_t reserve = sizeof(SolveDesc) + 16 * (size_t)MAX_UNITS + 64;      // the host-built tail of the block (SolveDesc + unit references) must fit behind what the kernel writes
      pa.desc = w->d_desc.p; pa.desc_cap = w->desc_cap - reserve; pa.cps_tmp = w->d_cps_tmp.p; pa.plan = w->d_plan.p; pa.out = w->h_plan_sum; pa.seq = ++w->seq_P;
      hipLaunchKernelGGL(k_evb_plan, dim3(1), dim3(512), 0, s_reg, pa);
      HIPCHK(hipGetLastError());
      HIPCHK(hipEventRecord(w->ev_plan, s_reg));
      if (spin_wait((volatile long long*)&w->h_plan_sum->seq, w->seq_P, w->ev_plan)) HIPCHK(hipErrorUnknown);
      if (ps->err != PE_DESC_ROOM) break;
      HIPCHK(hipDeviceSynchronize());      // the descriptor block grows: nothing else of this call is in flight yet
      if (w->h_desc) HIPCHK(hipHostFree(w->h_desc));
      w->desc_cap = 2 * (size_t)ps->need_bytes + 65536 + sizeof(SolveDesc) + 16 * (size_t)MAX_UNITS;
      HIPCHK(hipHostMalloc((void**)&w->h_desc, w->desc_cap, hipHostMallocDefault));
      if (w->d_desc.alloc(w->desc_cap)) FAIL(MSEVB_ERR_HIP, "hipMalloc failed (EVB descriptors)");
    }
    if (ps->err != PE_OK) FAIL(plan_err_code(ps->err), plan_error_text(ps->err));
    c->evb_device_plans++;
    if (ht_on) fprintf(stderr, "[msevb] k_evb_plan phases (us): order %.1f, pairs %.1f, tree %.1f, descriptors %.1f, couplings + summary %.1f; records %d, diabats %d; last diabat alone %.1f (its first ensure_mol %.1f)\n", (ps->t[1] - ps->t[0]) * 0.01, (ps->t[2] - ps->t[1]) * 0.01, (ps->t[3] - ps->t[2]) * 0.01, (ps->t[4] - ps->t[3]) * 0.01, (ps->t[5] - ps->t[4]) * 0.01, ps->n_region, ps->D, ps->t[6] * 0.01, ps->t[7] * 0.01);
    if (!ps->fits) w->device_plan_off_once = true;      // a tree too large for the fixed tables of the speculative step: this call goes the host's way (no plan, no speculation)
    else {
      // what the speculative calls start from (remember_plan's part): the tree, unit / grid / coupling indices, host-side mirrors of the descriptor counts
      const int D = ps->D;
      c->D = D; memcpy(c->plog, ps->plog, sizeof c->plog); memcpy(c->parent, ps->parent, sizeof c->parent);
      w->pl_D = D; memcpy(w->pl_plog, ps->plog, sizeof w->pl_plog); memcpy(w->pl_parent, ps->parent, sizeof w->pl_parent);
      w->pl_units_of.assign(D, std::vector<int>()); w->pl_grid_of.assign(D, -1); w->pl_cp_of.assign(D, -1); w->pl_eref_delta.assign(D, 0.0);
      w->units.assign(ps->nu, UnitDesc()); w->cps.assign(ps->ncp, CouplingDesc()); w->ras.assign(ps->nra, RecipAtom());
      w->pl_units_of[0].push_back(0); w->units[0].sign = +1.0;
      int r0 = 0;
      for (int d = 1; d < D; ++d) {
        if (ps->grid_of[d] < 0) continue;      // (another rank's diabat)
        for (int k = 0; k < 2 * ps->nhop[d]; ++k) { w->pl_units_of[d].push_back(ps->unit0[d] + k); w->units[ps->unit0[d] + k].sign = (k & 1) ? +1.0 : -1.0; }
        w->pl_grid_of[d] = ps->grid_of[d]; w->pl_cp_of[d] = ps->cp_of[d]; w->pl_eref_delta[d] = ps->eref_delta[d];
        for (int k = 0; k < ps->nras[d]; ++k) w->ras[r0 + k].grid = ps->grid_of[d];
        r0 += ps->nras[d];
      }
      w->pl_ngrid = ps->ngrid; w->pl_ncp_big = ps->ncp_big;
      w->plan.hyd = c->hydronium; w->plan.n_units = ps->nu; w->plan.n_cps = ps->ncp; w->plan.n_ras = ps->nra; w->plan.n_donors = ps->n_donors;
      w->plan_valid = true; w->plan_fits = true; w->plan_epoch = c->topo_epoch; w->plan_serial++;
      w->solve_desc_dirty = true;
      // nothing of this attempt is in flight except the neighbourhood kernel, whose record counter nobody has re-armed: zero it on that kernel's stream (the repeated call's
      // kernel follows there) - no "previous call failed half-way" clean-up with its stream synchronisation
      HIPCHK(hipMemsetAsync(w->d_counter.p, 0, sizeof(int), s_reg));
      rearm.ok = true;
      return MSEVB_RETRY;      // the same call again, from the recorded tree (its check passes: the positions are the ones the tree was made from)
    }
  }
  w->device_plan_off_once = false;
  HT(1);
  // ---- 2. principal diabat (calculate_total_force_energy, src/ms_evb.f90:411) - enqueued, not waited for
  const bool fast = !c->evb_element_stores;      // no per-element force stores: one gather of the blended potential (k_evb_blend_grids)
  c->shard_now = c->shard_principal && c->n_ranks > 1;
  c->skip_gather_now = fast;
  const bool defer_mol = c->overlap && c->evb_defer_molecules;
  c->defer_molecules_now = defer_mol;
  c->molecules_after_pair_now = defer_mol && c->evb_molecules_after_pair;
  c->ev_pos_ready = w->ev_in;      // new positions: what the PME front end waits for (not for the neighbourhood kernels behind them)
  c->defer_fold_now = fast && c->evb_defer_fold;      // the blend kernel folds the pair loop's sorted-order accumulator in (k_evb_blend)
  const bool single_join = fast && c->overlap && c->evb_single_join;      // the main stream does not wait for the side streams one by one behind the pair loop
  c->skip_side_joins_now = single_join;
  c->pme_split_now = single_join && c->evb_pme_split && c->evb_recip_linear && c->d_green.p && fft_lds_usable(c);      // spread now; the solve goes out behind the stencil-pair kernels (step 6b)
  c->refresh_side_now = c->overlap && c->evb_region_main;
  c->weights_needed_now = !(fast && c->evb_recip_linear && c->d_green.p);      // per-diabat transforms / element stores read the principal diabat's weights (k_evb_patch, k_stencil_gather, k_evb_recip_special)
  w->side_valid = false; w->side_wo_mol = false; w->side_by_A = false;
  const bool side_w = fast && c->overlap && c->pme_side_weights && pme_weightless_now(c);      // the blend phase's gather reads stored weights made beside the front end
  if (side_w) CHK(pme_side_weights(c, c->aux3, w->ev_in));
  const int rc_tfe = total_force_enqueue(c);
  c->weights_needed_now = false;
  c->shard_now = false; c->skip_gather_now = false; c->defer_molecules_now = false; c->molecules_after_pair_now = false; c->ev_pos_ready = nullptr; c->defer_fold_now = false; c->skip_side_joins_now = false; c->pme_split_now = false; c->refresh_side_now = false;
  CHK(rc_tfe);
  if (side_w) c->weights_side = true;      // (total_force_enqueue cleared it)
  HT(2);
  auto fetch_region = [&]() -> int {
  if (spin_wait((volatile long long*)(w->h_flag + 1), w->seq_R, w->ev_region, gsync)) HIPCHK(hipErrorUnknown);
  const int nreg = *w->h_count;
  if (nreg > REGION_CAP) FAIL(MSEVB_ERR_LIMIT, "EVB neighbourhood buffer too small");
  w->region.resize(nreg);
  {      // records in ascending molecule order (the reference's j_mole loops): sort 8-byte keys, then move each 224-byte record once
    const RegionRec* src = w->h_region;
    if (nreg > REGION_FAST) {
      w->region_raw.resize(nreg);
      HIPCHK(hipMemcpyAsync(w->region_raw.data(), w->d_region.p, (size_t)nreg * sizeof(RegionRec), hipMemcpyDeviceToHost, w->stream2)); HIPCHK(hipStreamSynchronize(w->stream2));
      src = w->region_raw.data();
    }
    w->region_key.resize(nreg);
    for (int k = 0; k < nreg; ++k) w->region_key[k] = ((long long)src[k].mol << 32) | (unsigned)k;
    std::sort(w->region_key.begin(), w->region_key.end());
    for (int k = 0; k < nreg; ++k) w->region[k] = src[(int)(w->region_key[k] & 0xffffffff)];
  }
  return MSEVB_OK;
  };
  if (!spec) CHK(fetch_region());
  HT(3);
  // ---- 3. diabat enumeration on the principal topology (evb_conduct_proton_transfer_recursive, :498)
  for (int d = 0; d < MAX_DIABATS; ++d) { c->parent[d] = -1; for (int h = 0; h < MAX_CHAIN; ++h) for (int q = 0; q < 5; ++q) c->plog[d][h][q] = -1; }
  c->D = 1;
  SpecPlan& plan = w->plan;
  PlanEnv penv; memset(&penv, 0, sizeof penv);
  if (spec) {      // the recorded tree
    c->D = w->pl_D;
    memcpy(c->plog, w->pl_plog, sizeof c->plog); memcpy(c->parent, w->pl_parent, sizeof c->parent);
  } else {      // plan_enumerate (evb_plan.hpp): the reference's depth-first search on the neighbourhood record
    w->plan_valid = false; w->plan_fits = true; c->evb_host_dfs_calls++;
    if (fill_plan_tables(c, w->ptab)) FAIL(MSEVB_ERR_LIMIT, "EVB: molecule-type tables exceed the fixed sizes of the enumeration");
    w->region_order.resize(w->region.size());
    for (size_t k = 0; k < w->region.size(); ++k) w->region_order[k] = (int)k;      // (fetch_region sorted the records by molecule)
    w->plan_frames.resize(MAX_CHAIN + 2);
    penv.tb = &w->ptab; penv.mol_type = c->h_mol_type.data(); penv.mol_natom = c->h_mol_natom.data(); penv.mol_atoms = c->h_mol_atoms.data(); penv.atype = c->h_atype.data();
    penv.chg = c->h_chg.data(); penv.mass = c->h_mass.data(); penv.rec = w->region.data(); penv.order = w->region_order.data(); penv.n_region = (int)w->region.size(); penv.hydronium = c->hydronium;
    int fits = 1;
    const int pe = plan_enumerate(penv, w->plan_frames.data(), c->plog, c->parent, &c->D, plan, &fits);
    if (pe) FAIL(plan_err_code(pe), plan_error_text(pe));
    if (!fits) w->plan_fits = false;
  }
  const int D = c->D, S = slots_per_rank(c, D), n_off = c->n_ranks * S, n_diag = 1 + n_off;
  HT(4);
  // ---- 4. evaluation units / couplings / recip patches of the diabats owned by this rank
  std::vector<UnitDesc>& units = w->units;      // (kept in the work area: the 2.4 kB descriptors are not re-allocated every step)
  std::vector<CouplingDesc>& cps = w->cps;
  std::vector<RecipAtom>& ras = w->ras;
  std::vector<int> grid_of(D, -1), cp_of(D, -1);
  std::vector<double> eref_delta(D, 0.0);
  std::vector<std::vector<int>> units_of(D);
  int ngrid = 0, ncp_big = 0;
  if (spec) { units_of = w->pl_units_of; grid_of = w->pl_grid_of; cp_of = w->pl_cp_of; eref_delta = w->pl_eref_delta; ngrid = w->pl_ngrid; ncp_big = w->pl_ncp_big; }
  else {      // plan_build_diabat (evb_plan.hpp): units, reciprocal-space atoms and coupling of every owned diabat
  int nu_need = 1, ncp_need = 0;
  for (int d = 1; d < D; ++d) if (owned(c, d)) { int nhop = 0; for (int h = 0; h < c->max_chain; ++h) if (c->plog[d][h][0] >= 0) nhop = h + 1; nu_need += 2 * nhop; ++ncp_need; }
  if (nu_need > MAX_UNITS) FAIL(MSEVB_ERR_LIMIT, "EVB descriptor staging too small");
  units.resize(nu_need); cps.resize(ncp_need); ras.resize((size_t)ncp_need * MAX_SPEC);
  PlanWork& PW = w->plan_work; PTopo& Pt = PW.P;
  Pt.n_mols = 0; Pt.n_atoms = 0; Pt.hydronium = c->hydronium;
  { int perr = PE_OK; if (plan_ensure_mol(penv, Pt, c->hydronium, &perr) < 0) FAIL(MSEVB_ERR_LIMIT, "EVB: hydronium outside region"); }
  { const int pe = plan_make_unit(penv, Pt, -1, -1, +1.0, 0, 0, 1, units[0], PW.where); if (pe) FAIL(plan_err_code(pe), plan_error_text(pe)); plan.unit_d[0] = 0; plan.unit_k[0] = 0; units_of[0].push_back(0); }
  std::vector<int> cp_d_unsorted, cp_nhop_unsorted;
  int iu = 1, ir = 0, icp = 0;
  for (int d = 1; d < D; ++d) {
    if (!owned(c, d)) continue;
    int nr = 0, nhop = 0;
    grid_of[d] = ngrid;
    const int pe = plan_build_diabat(penv, PW, d, D, c->plog, units.data(), iu, ras.data(), ir, ngrid, cps[icp], plan, &eref_delta[d], &nr, &nhop);
    if (pe) FAIL(plan_err_code(pe), plan_error_text(pe));
    for (int k = 0; k < 2 * nhop; ++k) units_of[d].push_back(iu + k);
    iu += 2 * nhop; ir += nr; ++ngrid;
    cp_of[d] = icp++;
    cp_d_unsorted.push_back(d); cp_nhop_unsorted.push_back(nhop);
  }
  ras.resize(ir);
  // couplings with more than 8 Zundel atoms first: they take the 16-lane instantiation of k_evb_coupling, the rest the 8-lane one
  {
    std::vector<int> order(cps.size());
    for (size_t k = 0; k < order.size(); ++k) order[k] = (int)k;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return (cps[a].nD + cps[a].nA > 8) > (cps[b].nD + cps[b].nA > 8); });
    std::vector<CouplingDesc>& sorted = w->cps_sorted; sorted.resize(cps.size());
    std::vector<int> newpos(cps.size());
    for (size_t k = 0; k < order.size(); ++k) { sorted[k] = cps[order[k]]; newpos[order[k]] = (int)k; if (sorted[k].nD + sorted[k].nA > 8) ncp_big++; }
    cps.swap(sorted);
    for (int d = 0; d < D; ++d) if (cp_of[d] >= 0) cp_of[d] = newpos[cp_of[d]];
    for (size_t k = 0; k < order.size(); ++k) { plan.cp_d[k] = cp_d_unsorted[order[k]]; plan.cp_nhop[k] = cp_nhop_unsorted[order[k]]; }
    for (size_t k = 0; k < cps.size(); ++k) { plan.cp_ra0[k] = 0; plan.cp_ra1[k] = 0; }
    for (int d = 0; d < D; ++d) if (grid_of[d] >= 0 && cp_of[d] >= 0) {      // (a diabat's reciprocal-space atoms are contiguous)
      int r0 = -1, r1 = -1;
      for (int r = 0; r < (int)ras.size(); ++r) if (ras[r].grid == grid_of[d]) { if (r0 < 0) r0 = r; r1 = r + 1; }
      if (r0 >= 0) { plan.cp_ra0[cp_of[d]] = r0; plan.cp_ra1[cp_of[d]] = r1; }
    }
  }
  }      // (!spec)
  const int nu = (int)units.size(), ncp = (int)cps.size(), nra = (int)ras.size();
  HT(5);
  // ---- 5. device buffers
  const int envb = std::min(16, (N + 255) / 256), cpb = 1024 /* partial-sum row stride; the coupling launches use up to this many blocks per element */;
  const int part_stride = 1024;
  const int n_rows = nu + ncp + ngrid + 8;
  if (c->d_Fdiag.alloc_room((size_t)n_diag * n3) || c->d_Foff.alloc_room((size_t)std::max(n_off, 1) * n3) || 
      w->d_tabs.alloc(1) || w->d_exch.alloc(MAX_TYPES) || w->d_part.alloc_room((size_t)n_rows * part_stride) ||
      w->d_E.alloc_room(2 * n_rows + 3 * std::max(ncp, 1) + 8) || w->d_dVspec.alloc((size_t)MAX_DIABATS * CPL_BINS * MAX_SPEC * 3) ||
      w->d_sbase.alloc_room(3 * std::max(nra, 1)) || w->d_sw.alloc_room(18 * (size_t)std::max(nra, 1)) || w->d_sdw.alloc_room(18 * (size_t)std::max(nra, 1)) ||
      w->d_grids.alloc_room(K3 * std::max(ngrid, 1)) || w->d_pack.alloc_room(fast ? K3 : K3 * std::max(ngrid, 1)) /* (fast path: one blended grid; it may hold the prefilled Q_1 already: no re-allocation) */ || w->d_F11.alloc(n3) || w->d_com_atom.alloc(n3))
    FAIL(MSEVB_ERR_HIP, "hipMalloc failed (EVB work buffers)");
  c->evb_cap = D;
  CHK(upload_evb_tables(c, w));
  if (nu > MAX_UNITS || ncp > MAX_DIABATS || nra > MAX_DIABATS * MAX_SPEC) FAIL(MSEVB_ERR_LIMIT, "EVB descriptor staging too small");
  // single rank, no per-element stores: the Hamiltonian is assembled and diagonalised on the device (k_evb_solve), its descriptor rides along
  const bool dev_solve_mr = dev_solve_mr_early, dev_solve = dev_solve_early;
  int n_unit_refs = 0;
  for (int d = 1; d < D; ++d) n_unit_refs += (int)units_of[d].size();
  const size_t off_u = 0, off_c = off_u + (size_t)nu * sizeof(UnitDesc), off_r = off_c + (size_t)ncp * sizeof(CouplingDesc);
  const size_t off_s = (off_r + (size_t)nra * sizeof(RecipAtom) + 7) & ~(size_t)7, off_si = off_s + sizeof(SolveDesc), off_ss = (off_si + (size_t)n_unit_refs * sizeof(int) + 7) & ~(size_t)7;
  const size_t desc_bytes = dev_solve ? off_ss + (size_t)n_unit_refs * sizeof(double) : off_r + (size_t)nra * sizeof(RecipAtom);
  // "setup" stream: aux2 when overlapping - it already holds this step's neighbourhood fetch and per-molecule kernel; descriptors,
  // zero-fills and everything that needs only positions + descriptors follow there, beside the principal force still running elsewhere
  const hipStream_t ss = c->overlap ? w->stream2 : c->stream;
  if (desc_bytes > w->desc_cap) {
    HIPCHK(hipDeviceSynchronize());
    if (w->h_desc) HIPCHK(hipHostFree(w->h_desc));
    w->desc_cap = desc_bytes + desc_bytes / 2 + 4096;
    HIPCHK(hipHostMalloc((void**)&w->h_desc, w->desc_cap, hipHostMallocDefault));
    if (w->d_desc.alloc(w->desc_cap)) FAIL(MSEVB_ERR_HIP, "hipMalloc failed (EVB descriptors)");
  }
  if (!spec) {
  memcpy(w->h_desc + off_u, units.data(), (size_t)nu * sizeof(UnitDesc));
  if (ncp) memcpy(w->h_desc + off_c, cps.data(), (size_t)ncp * sizeof(CouplingDesc));
  if (nra) memcpy(w->h_desc + off_r, ras.data(), (size_t)nra * sizeof(RecipAtom));
  }
  BlendTerms bt_dev; memset(&bt_dev, 0, sizeof bt_dev);      // slots of the blend terms (the coefficients will come from the device)
  if (dev_solve) {
    SolveDesc* sd = (SolveDesc*)(w->h_desc + off_s);
    int* ui = (int*)(w->h_desc + off_si); double* us = (double*)(w->h_desc + off_ss);
    sd->D = D; sd->nu = nu; sd->ncp = ncp; sd->ngrid = ngrid; sd->unit00 = units_of[0][0];
    sd->once = (c->shard_principal && c->n_ranks > 1 && !c->nccl_comm && c->rank != 0) ? 0.0 : 1.0;      // (virtual ranks: replicated terms count once, on rank 0 - as the host's assembly below)
    sd->e_hyd = c->e_ref[c->h_mol_type[c->hydronium]]; sd->Hp_norep = sd->once * sd->e_hyd;
    int q = 0;
    sd->unit_off[0] = 0; sd->unit_off[1] = 0;
    for (int d = 1; d < D; ++d) {
      sd->parent[d] = c->parent[d]; sd->grid_of[d] = grid_of[d]; sd->cp_of[d] = cp_of[d]; sd->eref_delta[d] = eref_delta[d];
      sd->unit_off[d] = q;
      for (int iu : units_of[d]) { ui[q] = iu; us[q] = units[iu].sign; ++q; }
      sd->unit_off[d + 1] = q;
    }
    sd->parent[0] = -1; sd->grid_of[0] = -1; sd->cp_of[0] = -1; sd->eref_delta[0] = 0.0;
    (void)blend_terms(c, D, nullptr, bt_dev, sd->term_i, sd->term_j);
    sd->nterm = bt_dev.n;
  }
  if (spec && dev_solve && (w->solve_desc_dirty || !w->solve_desc_on_device)) HIPCHK(hipMemcpyAsync(w->d_desc.p + off_s, w->h_desc + off_s, desc_bytes - off_s, hipMemcpyHostToDevice, ss));      // (units / couplings / reciprocal-space atoms were written by k_evb_plan)
  w->solve_desc_dirty = false; w->solve_desc_on_device = dev_solve;      // (a call without the device-side solve leaves no SolveDesc behind: the first call with it - a virtual rank that has just been given its payloads - uploads one)
  if (!spec) {
    HIPCHK(hipMemcpyAsync(w->d_desc.p, w->h_desc, desc_bytes, hipMemcpyHostToDevice, ss));
    // the plan for the steps that follow (they start from these descriptors): only when everything fitted its fixed-size tables
    plan.n_units = nu; plan.n_cps = ncp; plan.n_ras = nra;
    if (w->plan_fits && nu <= MAX_UNITS && ncp <= MAX_DIABATS && c->evb_speculate) HIPCHK(hipMemcpyAsync(w->d_plan.p, &plan, sizeof(SpecPlan), hipMemcpyHostToDevice, ss));
    else w->plan_fits = false;
  } else {      // descriptors of the last enumeration, already on the device: today's geometry written into them by one small kernel
    hipLaunchKernelGGL(k_evb_refresh, dim3(nu + ncp), dim3(64), 0, ss, (const SpecPlan*)w->d_plan.p, (UnitDesc*)(w->d_desc.p + off_u), (CouplingDesc*)(w->d_desc.p + off_c), (RecipAtom*)(w->d_desc.p + off_r),
                       (const double*)c->d_xyz.p, (const double*)c->d_mass.p, (const int*)c->d_mol_natom.p, (const int*)c->d_mol_atoms.p, box, std::max(c->rc, w->screen_r) + 0.5, w->d_sphere.p, w->d_check.p, region_aside ? (int*)nullptr : h_verdict, w->d_refresh_scratch.p,
                       region_aside ? w->d_counter.p + 1 : (int*)nullptr);
    HIPCHK(hipGetLastError());
    w->spec_steps++; c->evb_spec_steps++;
  }
  const UnitDesc* d_units = (const UnitDesc*)(w->d_desc.p + off_u);
  const CouplingDesc* d_cps = (const CouplingDesc*)(w->d_desc.p + off_c);
  const RecipAtom* d_ra = (const RecipAtom*)(w->d_desc.p + off_r);
  double* part_units = w->d_part.p;
  double* part_cp = w->d_part.p + (size_t)nu * part_stride;
  double* part_grid = w->d_part.p + (size_t)(nu + ncp) * part_stride;
  double* E_units_env = w->d_E.p;                 // [nu]
  double* E_units_spec = w->d_E.p + nu;           // [nu]
  double* E_cp_env = w->d_E.p + 2 * nu;           // [ncp]
  double* E_grid = w->d_E.p + 2 * nu + ncp;       // [ngrid]
  double* d_cpout = w->d_E.p + 2 * nu + ncp + ngrid;   // [ncp][3]: one download fetches everything
  {   // delta store of the diagonal real-space terms: zero on first use / after a resize / after a failed call, then self-cleaning
    const size_t need = (size_t)n_diag * n3;
    if (w->d_Fdelta.n < need) { if (w->d_Fdelta.alloc_room(need)) FAIL(MSEVB_ERR_HIP, "hipMalloc failed (EVB delta store)"); w->delta_dirty = true; }
    if (w->delta_dirty) { HIPCHK(hipMemsetAsync(w->d_Fdelta.p, 0, w->d_Fdelta.n * 8, ss)); w->delta_dirty = false; }
  }
  // every element of an owned slot is written by its kernels; slots of other ranks' diabats are never read (partial blend + all-reduce)
  if (w->dv_dirty) { HIPCHK(hipMemsetAsync(w->d_dVspec.p, 0, w->d_dVspec.n * 8, ss)); w->dv_dirty = false; }   // afterwards k_evb_coupling_finish leaves it zeroed
  // B-spline stencils of the special atoms in their new states (needs descriptors only)
  stamp(c, 3, ss);
  const bool linear = !c->evb_element_stores && c->evb_recip_linear && c->d_green.p;      // no transform per diabat: linear response of theta to the charge differences (E_grid holds energy DIFFERENCES)
  if (ngrid) {
    if (linear && (w->d_obase.alloc_room(3 * (size_t)nra) || w->d_ow.alloc_room(18 * (size_t)nra) || w->d_odw.alloc_room(18 * (size_t)nra))) FAIL(MSEVB_ERR_HIP, "hipMalloc failed (EVB stencils)");
    const int nthr = 18 * nra * (linear ? 2 : 1);
    hipLaunchKernelGGL(k_evb_spec_prep, dim3((nthr + 255) / 256), dim3(256), 0, ss, nra, d_ra, K, c->spline_grid, c->kk[0][0], c->kk[0][1], c->kk[0][2],
                       c->kk[1][0], c->kk[1][1], c->kk[1][2], c->kk[2][0], c->kk[2][1], c->kk[2][2], c->d_B6.p, c->d_B5.p, w->d_sbase.p, w->d_sw.p, w->d_sdw.p, c->d_energies.p + S_FAULT,
                       linear ? (const double*)c->d_xyz.p : (const double*)nullptr, w->d_obase.p, w->d_ow.p, w->d_odw.p);
  }
  if (linear) {
    if (w->d_Elq.alloc(2 * MAX_DIABATS + 1) || w->d_stencils.alloc_room((size_t)std::max(2 * nra, 1))) FAIL(MSEVB_ERR_HIP, "hipMalloc failed (EVB stencils)");
    if (!w->elq_zeroed) { HIPCHK(hipMemsetAsync(w->d_Elq.p, 0, (2 * MAX_DIABATS + 1) * 8, ss)); w->elq_zeroed = true; }      // k_evb_grid_energy re-zeroes it every step
    if (!w->ev_pairs) HIPCHK(hipEventCreateWithFlags(&w->ev_pairs, hipEventDisableTiming));
  }
  w->last_linear = linear; w->last_nst = 2 * nra;
  if (c->overlap) {      // descriptors + special stencils are on the device: the main stream and the off-diagonal chain may use them
    if (c->evb_pair_first && c->ev_pair_done) HIPCHK(hipStreamWaitEvent(ss, c->ev_pair_done, 0));      // (schedule experiment: the diabat chains start behind the pair loop instead of beside it)
    HIPCHK(hipEventRecord(w->ev_fork2, ss));
    if (!(gmode && dev_solve_early && pub_spin_early)) HIPCHK(hipStreamWaitEvent(c->stream, w->ev_fork2, 0));      // (linear graphs with the device-side solve: all the main chain takes from the descriptors it takes behind the solve's mark, which follows them)
    HIPCHK(hipStreamWaitEvent(c->aux, w->ev_fork2, 0));
  }
  HT(6);
  const hipStream_t sg = c->overlap ? c->aux : c->stream;      // diabat grids: `aux` is free once the principal convolution is done
  const int nst = 2 * nra;
  // The grid chain of the linear-response path (stencil records, stencil-pair sums, the principal solve, dots with theta_1).  (Round 5: enqueued in FRONT of the
  // environment and coupling chains instead of behind them - 2 536 / 2 474 against 2 512 / 2 501 steps/s, noise: the chains' ends are set by the GPU they share with
  // the pair loop, not by when the host reaches them.)
  auto grid_chain_linear = [&]() -> int {
    phase_begin(c, PH_EVB_RECIP, sg);
    stamp(c, 5, sg);
    // stencil records of the special atoms (old and new state), their dots with theta_1 and their mutual interaction through G = IFFT(CB):
    // energies relative to the principal diabat -> E_grid, forces on the special atoms -> the diabats' slots of the delta store
    // (the stencil records and the stencil-pair sums need no potential: they went out on the descriptor stream right behind the special
    //  atoms' stencils, step 5; here, behind the principal convolution, only the dots with theta_1 are left)
    // The stencil records and the stencil-pair sums need the special atoms' stencils only (descriptor stream, ev_fork2 - long done): they
    // run HERE, on the stream of the principal front end right behind it, with the dots against theta_1 - no event hop in between, and not
    // behind the environment chain of the descriptor stream, where they used to wait ~100 us for kernels they do not depend on
    {
      GridRanges granges; memset(&granges, 0, sizeof granges);
      { int g = 0; granges.start[0] = 0; for (int e = 0; e < nra; ++e) { while (g < ras[e].grid) granges.start[++g] = 2 * e; } while (g < ngrid) granges.start[++g] = nst; }
      int max_ns = 0;
      for (int g = 0; g < ngrid; ++g) max_ns = std::max(max_ns, granges.start[g + 1] - granges.start[g]);
      if (ht_on && g_ht.n == 3) fprintf(stderr, "[msevb] recip entries %d, grids %d, most stencils per grid %d\n", nra, ngrid, max_ns);
      // (round 5: on a FIFTH stream, beside the charge spreading instead of behind it, these two made the step 20 % slower - 0.482 against 0.408 ms, the grid chain ending
      //  at 295 instead of 227 us: this runtime maps the process's streams onto four hardware queues, a fifth shares one - and slowed every later context of the process)
      const hipStream_t sp = sg;
      hipLaunchKernelGGL(k_evb_stencils, dim3(nra), dim3(64), 0, sp, nra, d_ra, w->d_obase.p, w->d_ow.p, w->d_odw.p, w->d_sbase.p, w->d_sw.p, w->d_sdw.p, w->d_stencils.p);
      hipLaunchKernelGGL(k_evb_recip_pairs, dim3(max_ns, ngrid), dim3(256), 0, sp, granges, w->d_stencils.p, K, c->d_green.p, c->conv, c->kk[0][0], c->kk[0][1],
                         c->kk[0][2], c->kk[1][0], c->kk[1][1], c->kk[1][2], c->kk[2][0], c->kk[2][1], c->kk[2][2], w->d_Fdelta.p, n3, w->d_Elq.p + MAX_DIABATS);
    }
    CHK(pme_front_finish(c));      // (split front end: the principal solve goes out here, behind the kernels above that do not need its result)
    hipLaunchKernelGGL(k_evb_recip_linear, dim3((nst + 3) / 4), dim3(256), 0, sg, nst, w->d_stencils.p, K, c->d_theta.p, c->conv, c->kk[0][0], c->kk[0][1], c->kk[0][2], c->kk[1][0],
                       c->kk[1][1], c->kk[1][2], c->kk[2][0], c->kk[2][1], c->kk[2][2], w->d_Fdelta.p, n3, w->d_Elq.p);
    HIPCHK(hipGetLastError());
    if (c->overlap) HIPCHK(hipEventRecord(w->ev_rs, sg));      // the special atoms' forces are in the delta store
    phase_end(c, PH_EVB_RECIP, sg);
    stamp(c, 6, sg);
    return MSEVB_OK;
  };
  // ---- 6a. real-space / bonded / repulsion deltas of all units (ms_evb_diabat_force_energy, :1421)
  const hipStream_t sr = ss;     // on aux2 behind the descriptors: beside the principal force and the reciprocal space of the diabats
  phase_begin(c, PH_EVB_REAL, sr);
  if (spec) {
    hipLaunchKernelGGL(k_evb_candidates, dim3((N + 255) / 256), dim3(256), 0, sr, N, 0.0, 0.0, 0.0, 0.0, c->d_xyz.p, box, w->d_counter.p + 1, w->d_cand.p, region_aside ? (int*)nullptr : w->d_counter.p, (const double*)w->d_sphere.p);
  } else {
    // candidate sphere: centre = hydronium COM, radius = farthest special atom + max(cutoff, repulsion screen) + a little
    const RegionRec* rh = find_region(w, c->hydronium);
    double far2 = 0.0;
    for (const UnitDesc& u : units) for (int k = 0; k < u.n_spec; ++k) {
      double d2 = 0.0;
      for (int q = 0; q < 3; ++q) { double d = u.spec[k].x[q] - rh->com[q]; d -= c->box[q] * std::floor(d / c->box[q] + 0.5); d2 += d * d; }
      far2 = std::max(far2, d2);
    }
    const double Renv = std::sqrt(far2) + std::max(c->rc, w->screen_r) + 0.5;
    hipLaunchKernelGGL(k_evb_candidates, dim3((N + 255) / 256), dim3(256), 0, sr, N, rh->com[0], rh->com[1], rh->com[2], Renv * Renv, c->d_xyz.p, box,
                       w->d_counter.p + 1, w->d_cand.p, w->d_counter.p, (const double*)nullptr);
  }
  hipLaunchKernelGGL(k_evb_env, dim3(envb, nu), dim3(256), 0, sr, w->d_counter.p + 1, w->d_cand.p, nu, d_units, w->d_tabs.p, c->d_xyz.p, c->d_chg.p, c->d_atype.p, c->d_mol_of.p,
                     c->d_pair_par.p, c->nt, c->d_ewald_tab.p, 1.0 / c->erfc_dx, c->d_tt.p, c->d_dtt.p, box, c->rc * c->rc, w->d_Fdelta.p, n3, part_units, part_stride);
  hipLaunchKernelGGL(k_evb_special, dim3(nu), dim3(64), 0, sr, nu, d_units, w->d_tabs.p, c->d_moltypes.p, c->d_pair_par.p, c->nt,
                     c->d_ewald_tab.p, 1.0 / c->erfc_dx, c->d_tt.p, c->d_dtt.p, box, c->rc * c->rc, c->alpha, 2.0 * c->alpha / c->pi_sqrt, c->conv, c->pi,
                     w->d_Fdelta.p, n3, E_units_spec, (const double*)part_units, envb, part_stride, E_units_env);
  HIPCHK(hipGetLastError());
  phase_end(c, PH_EVB_REAL, sr);
  stamp(c, 4, sr);
  const bool pub_spin = c->overlap && c->evb_publish_spin;      // fetch A's publish kernel waits for the chains' marks, not for their events (inside the linear graphs too: the numbers then come from the step block)
  if (pub_spin) {
    if (!w->d_chain_done.p) { if (w->d_chain_done.alloc(4)) FAIL(MSEVB_ERR_HIP, "hipMalloc failed (chain marks)"); HIPCHK(hipMemset(w->d_chain_done.p, 0, 32)); HIPCHK(hipDeviceSynchronize()); }      // (once per context.  hipMemset of device memory may RETURN EARLY and the library's streams do not wait for the null stream: without the synchronisation the fill could land BEHIND the first mark and wipe it - the energy-fetch kernel then sat out its 2 s limit: the intermittent MSEVB_ERR_NUMERIC of round 3, HISTORY.md E; the marks come from different streams)
    hipLaunchKernelGGL(k_evb_mark, dim3(1), dim3(1), 0, sr, w->d_chain_done.p, gmode ? 0LL : w->seq_A + 1, sb);
  }
  if (c->overlap) HIPCHK(hipEventRecord(w->ev_join2, w->stream2));
  // ---- 6a'. off-diagonal elements (evb_diabatic_coupling, :1021): few, latency-bound kernels writing only d_Foff; on aux2 behind the
  //           diagonal deltas (the all-atom gather waits for those, only the blend for these)
  if (ncp) {
    // own stream (aux3): beside the diagonal deltas on aux2 - both chains feed fetch A, which the blend waits for
    hipStream_t so = c->overlap ? c->aux3 : c->stream;
    hipLaunchKernelGGL(k_mol_com, dim3((M + 255) / 256), dim3(256), 0, so, M, c->d_mol_natom.p, c->d_mol_atoms.p, c->d_xyz.p, c->d_mass.p, w->d_com_atom.p);      // positions only
    if (c->overlap) HIPCHK(hipStreamWaitEvent(so, w->ev_fork2, 0));
    // one pass over the atoms for up to CPG couplings: 1024 atoms per workgroup and trip, never more than `cpb` workgroups (the row length of the partial sums)
    int cpb_big = std::min(cpb, std::max(1, (N + 256 * CPA - 1) / (256 * CPA)));
    if (c->evb_coupling_blocks > 0) cpb_big = std::min(cpb, c->evb_coupling_blocks);
    const int cpb_small = cpb_big;
    const int cgroups = std::min(ncp, std::max(1, (512 + cpb_big - 1) / cpb_big)), cpg = std::min(CPG, (ncp + cgroups - 1) / cgroups);
    hipLaunchKernelGGL(k_evb_coupling, dim3(cpb_big, (ncp + cpg - 1) / cpg), dim3(256), 0, so, N, ncp, d_cps, c->d_mol_of.p, w->d_com_atom.p, c->d_xyz.p, c->d_chg.p,
                       w->d_exch.p, box, c->conv, c->d_Foff.p, n3, part_cp, part_stride, w->d_dVspec.p, cpg);
    hipLaunchKernelGGL(k_evb_coupling_finish, dim3(ncp), dim3(64), 0, so, ncp, d_cps, w->d_exch.p, c->d_mass.p, box, c->conv, (const double*)part_cp, part_stride, ncp_big, cpb_big, cpb_small,
                       E_cp_env, w->d_dVspec.p, c->d_Foff.p, n3, d_cpout);
    HIPCHK(hipGetLastError());
    stamp(c, 7, so);
    if (pub_spin) hipLaunchKernelGGL(k_evb_mark, dim3(1), dim3(1), 0, so, w->d_chain_done.p + 1, gmode ? 0LL : w->seq_A + 1, sb);
    if (c->overlap) HIPCHK(hipEventRecord(w->ev_cpl, so));
  }
  if (defer_mol) {      // the principal diabat's per-molecule terms: behind the coupling chain on aux3 (they ran in front of it and held it up by ~80 us)
    if (c->evb_molecules_after_pair && c->ev_pair_done) HIPCHK(hipStreamWaitEvent(c->aux3, c->ev_pair_done, 0));      // ... and behind the pair loop: they fill the blend phase's window instead of taking CUs from the pair loop
    c->shard_now = c->shard_principal && c->n_ranks > 1;
    const int rc_mol = molecule_forces(c, S_ELEC_INTRA, true, true, c->aux3);
    c->shard_now = false;
    CHK(rc_mol);
    CHK(flush_range(c, S_ELEC_INTRA, S_DIH, c->aux3));      // its energy sums on its own stream (fetch B waits for the event, not the main stream)
    HIPCHK(hipEventRecord(c->ev_join2, c->aux3));
    // who waits: fetch B (energies), and in the blend phase only the last kernel - the transforms and the gather of the blended potential
    // ADD to the force array like the per-molecule kernel does, k_evb_blend is the first to read it
    if (!fast) HIPCHK(hipStreamWaitEvent(c->stream, c->ev_join2, 0));      // per-element stores: k_evb_slot0 copies F11
  }
  w->mol_deferred = defer_mol;
  // The Hamiltonian needs energies only, and all of them exist before the all-atom gather (the longest kernel of the EVB phase)
  // starts: fetch them now on stream2 and let the host diagonalise while the gather runs.
  const size_t nE = (size_t)2 * nu + ncp + ngrid + 3 * (size_t)ncp;
  if (nE + 1 > w->hE_cap) {
    if (w->h_E) HIPCHK(hipHostFree(w->h_E));
    w->hE_cap = 2 * nE + 64;      // + 1: the principal reciprocal energy rides along (fetch A)
    HIPCHK(hipHostMalloc((void**)&w->h_E, w->hE_cap * sizeof(double), hipHostMallocDefault));
  }
  // The ground-state VECTOR does not depend on the principal diabat's energy E0: it sits on the whole diagonal (H = H' + E0 I).  So the
  // host diagonalises H' from the energies that exist long before the pair loop ends (unit, coupling and grid energies, fetch A) and
  // the blend kernels queue up directly behind the pair loop; E0 (fetch B: the principal energy slots, complete only after the pair
  // loop) is added to the eigenvalue and to the reported Hamiltonian afterwards, off the critical path.  Both downloads are kernels that
  // write into pinned host memory (no copy commands).
  auto fetch_A = [&]() -> int {      // call after everything that produces a unit / coupling / grid energy has been enqueued
    const int n_recip_part = c->slot_count[S_RECIP];      // principal reciprocal energy: its partial sums are complete on `sg`;
    c->slot_count[S_RECIP] = 0;                           // reduced by the publish kernel below (one launch and its gap fewer in front of the host's turn)
    if (c->overlap) HIPCHK(hipEventRecord(w->ev_E, sg));
    // published from the diabat grid stream, normally the last chain to finish: the waits for the other two chains' events are then
    // already satisfied and the publish kernel starts right behind the last energy kernel (a publish on another stream starts an event hop later)
    // ... behind ONE barrier packet: the descriptor stream (idle since its environment chain) waits for the coupling chain and records the
    // event this stream waits for (two waits in front of the publish kernel were two barrier packets of ~10 us on the way to the host's turn)
    if (c->overlap && !pub_spin) {
      if (!w->ev_chains) HIPCHK(hipEventCreateWithFlags(&w->ev_chains, hipEventDisableTiming));
      if (ncp) HIPCHK(hipStreamWaitEvent(w->stream2, w->ev_cpl, 0));
      HIPCHK(hipEventRecord(w->ev_chains, w->stream2));
      HIPCHK(hipStreamWaitEvent(sg, w->ev_chains, 0));
    }
    const bool lin_e = linear && ngrid > 0;
    const long long seqA = ++w->seq_A;      // (the chains' marks carry this number)
    SolveArgs sa; memset(&sa, 0, sizeof sa);      // (padding bytes too: a captured / replayed launch is compared by its argument bytes)
    size_t lds = 0;
    if (dev_solve) {      // Hamiltonian, ground state and blend coefficients at the end of the kernel that completes the energies - beside the pair loop's tail, not behind it
      const size_t need = 2 + 2 * (size_t)D + (size_t)D * D;
      if (need > w->h_solve_cap) {
        if (w->h_solve) HIPCHK(hipHostFree(w->h_solve));
        w->h_solve_cap = 2 * need + 64;
        HIPCHK(hipHostMalloc((void**)&w->h_solve, w->h_solve_cap * sizeof(double), hipHostMallocDefault));
      }
      if (w->d_coef.alloc(1 + MAX_DIABATS + 2 * MAX_DIABATS)) FAIL(MSEVB_ERR_HIP, "hipMalloc failed (blend coefficients)");
      const int hp = (D + 1) / 2;
      lds = ((size_t)2 * D * D + D + 6 * hp + std::max(D <= 16 ? D * D + 2 * D : 0, n_unit_refs)) * 8 + 16;      // A, V, cplA, round tables (4 hp doubles + 2 hp ints), scratch
      if (!c->attr_evb_solve) {
        const int Dm = MAX_DIABATS, hm = Dm / 2, most = (2 * Dm * Dm + Dm + 6 * hm + std::max(16 * 16 + 32, MAX_UNITS)) * 8 + 16;
        HIPCHK(hipFuncSetAttribute((const void*)k_evb_publish, hipFuncAttributeMaxDynamicSharedMemorySize, most));
        c->attr_evb_solve = true;
      }
      sa.on = 1; sa.method = c->evb_device_solve_method; sa.nref = n_unit_refs;
      sa.sd = (const SolveDesc*)(w->d_desc.p + off_s); sa.unit_idx = (const int*)(w->d_desc.p + off_si); sa.unit_sign = (const double*)(w->d_desc.p + off_ss);
      sa.E = w->d_E.p; sa.e_recip1 = linear ? w->d_Elq.p + 2 * MAX_DIABATS /* zero: the grid energies are differences already */ : c->d_energies.p + S_RECIP;
      sa.dcoef = w->d_coef.p; sa.out_host = w->h_solve; sa.mark_seq = gmode ? 0 : seqA; sa.sb = sb; sa.done_mark = pub_spin ? w->d_chain_done.p + 3 : (long long*)nullptr;
      sa.done_flag = (volatile long long*)(w->h_flag + 6); sa.done_seq = (++w->seq_S, gmode ? 0LL : w->seq_S); sa.fault = c->d_energies.p + S_TIMEOUT;
      if (dev_solve_mr) {
        sa.phase = 1; sa.R = c->n_ranks; sa.rank = c->rank; sa.P = msevb_evb_payload_size(D, c->n_ranks); sa.S = (sa.P - 9) / 2;
        if (c->d_comm_h.alloc((size_t)sa.R * sa.P)) FAIL(MSEVB_ERR_HIP, "hipMalloc failed (comm)");
        if (!c->nccl_comm) {      // virtual rank: the other ranks' payloads as the caller gave them
          if (c->virt_payload.size() != (size_t)sa.R * sa.P) FAIL(MSEVB_ERR_ARG, "msevb_comm_set_virtual_payload: the payloads do not have the size of this call's diabat tree and rank count");
          if (c->virt_payload_dirty) { HIPCHK(hipMemcpyAsync(c->d_comm_h.p, c->virt_payload.data(), c->virt_payload.size() * 8, hipMemcpyHostToDevice, sg)); c->virt_payload_dirty = false; }
        }
        sa.payload = c->d_comm_h.p;
        if (!c->attr_evb_solve_g) {
          const int Dm = MAX_DIABATS, hm = Dm / 2, most = (2 * Dm * Dm + Dm + 6 * hm + std::max(16 * 16 + 32, MAX_UNITS)) * 8 + 16;
          HIPCHK(hipFuncSetAttribute((const void*)k_evb_solve_gathered, hipFuncAttributeMaxDynamicSharedMemorySize, most));
          c->attr_evb_solve_g = true;
        }
      }
    }
    hipLaunchKernelGGL(k_evb_publish, dim3(1), dim3(256), lds, sg, w->d_E.p, (int)nE, w->h_E, c->d_energies.p + S_RECIP, 1, w->h_E + nE, (volatile long long*)w->h_flag, gmode ? 0LL : seqA,
                       lin_e ? ngrid : 0, w->d_Elq.p, w->d_Elq.p + MAX_DIABATS, c->conv, E_grid,
                       (const double*)(c->d_partials.p + (size_t)S_RECIP * MAX_BLOCKS), n_recip_part, c->d_energies.p + S_RECIP,
                       pub_spin ? (const long long*)w->d_chain_done.p : (const long long*)nullptr, gmode ? 0LL : seqA, pub_spin ? (ncp ? 2 : 1) : 0, c->d_energies.p + S_TIMEOUT, sa, sb, (int)SB_A);
    HIPCHK(hipGetLastError());
    stamp(c, 10, sg);
    if (c->overlap) HIPCHK(hipEventRecord(w->ev_fetchA, sg));
    if (dev_solve_mr) {      // the ranks' entries meet (in place: this rank's slot is where phase 1 wrote it), then the same solve on every rank
      if (c->nccl_comm) CHK(msevb_comm_allgather_device(c, sa.P, sg));
      SolveArgs sa2 = sa; sa2.phase = 2;
      hipLaunchKernelGGL(k_evb_solve_gathered, dim3(1), dim3(256), lds, sg, sa2, sb);
      HIPCHK(hipGetLastError());
    }
    if (dev_solve) HIPCHK(hipEventRecord(w->ev_solve, sg));
    return MSEVB_OK;
  };
  SolveArgs sa_none; memset(&sa_none, 0, sizeof sa_none);
  auto fetch_B = [&]() -> int {      // principal energy slots + Verlet displacement flag (behind the pair loop)
    if (!c->overlap) CHK(flush_reductions(c));
    if (c->overlap) {      // the reduction of the partial sums goes to the side stream too: the main stream runs from the pair loop straight into the blend phase
      HIPCHK(hipEventRecord(w->ev_flush, c->stream)); HIPCHK(hipStreamWaitEvent(w->stream2, w->ev_flush, 0));
      if (c->top2_join_pending) { HIPCHK(hipStreamWaitEvent(w->stream2, c->ev_top2, 0)); c->top2_join_pending = false; }
      CHK(flush_range(c, 0, N_SLOTS - 1, w->stream2));
      // (the principal reciprocal energy slot is reduced on the diabat grid stream by fetch A's publish kernel: ev_fetchA, recorded behind ev_E on that stream, covers both)
      HIPCHK(hipStreamWaitEvent(w->stream2, w->ev_fetchA, 0));      // ... by its publish kernel
      if (defer_mol || single_join) HIPCHK(hipStreamWaitEvent(w->stream2, c->ev_join2, 0));      // per-molecule energies (aux3)
    }
    const bool e8 = dev_solve_mr && c->nccl_comm && c->shard_principal;      // the ranks' partial principal energies: summed by a small all-reduce right here (second collective of the call: all-gather, this, the forces)
    if (e8) {
      if (!c->h_comm_e8) { HIPCHK(hipHostMalloc((void**)&c->h_comm_e8, 8 * sizeof(double), hipHostMallocDefault)); memset(c->h_comm_e8, 0, 64); }
      CHK(msevb_comm_allreduce_e8_device(c, ss));
    }
    hipLaunchKernelGGL(k_evb_publish, dim3(1), dim3(256), 0, ss, c->d_energies.p, (int)N_SLOTS, c->h_energies, e8 ? (const double*)c->d_comm_e.p : (const double*)nullptr, e8 ? 8 : 0, e8 ? c->h_comm_e8 : (double*)nullptr,
                       (volatile long long*)(w->h_flag + 2), (++w->seq_B, gmode ? 0LL : w->seq_B), 0, (double*)nullptr, (double*)nullptr, 0.0, (double*)nullptr, (const double*)nullptr, 0, (double*)nullptr,
                       (const long long*)nullptr, 0LL, 0, (double*)nullptr, sa_none, sb, (int)SB_B);
    HIPCHK(hipGetLastError());
    if (c->overlap) HIPCHK(hipEventRecord(w->ev_fetch, ss));
    return MSEVB_OK;
  };
  // ---- 6b. reciprocal space of the owned diabats (calculate_reciprocal_space_pme, :1962)
  if (ngrid && linear) CHK(grid_chain_linear());
  else {
    CHK(pme_front_finish(c));      // (no stencil kernels to put in front of it)
    phase_begin(c, PH_EVB_RECIP, sg);
    stamp(c, 5, sg);
    if (ngrid) {
    hipLaunchKernelGGL(k_evb_grid_init, dim3(2048), dim3(256), 0, sg, K3, ngrid, c->d_Q.p, w->d_grids.p);
    hipLaunchKernelGGL(k_evb_patch, dim3((nra + 3) / 4), dim3(256), 0, sg, nra, d_ra, K, K3, c->d_base.p, c->d_w.p, w->d_sbase.p, w->d_sw.p, w->d_grids.p);
    HIPCHK(hipGetLastError());
    int nbk = 0;
    c->pme_st = c->overlap ? sg : nullptr;
    const int rc_conv = pme_convolve(c, w->d_grids.p, ngrid, part_grid, part_stride, &nbk);      // one batched FFT pair for all owned diabats
    c->pme_st = nullptr;
    CHK(rc_conv);
    hipLaunchKernelGGL(k_sum_rows, dim3(ngrid), dim3(256), 0, sg, ngrid, nbk, part_stride, part_grid, E_grid);
    if (!fast) hipLaunchKernelGGL(k_evb_pack, dim3(2048), dim3(256), 0, sg, K3, ngrid, w->d_grids.p, w->d_pack.p);
      }
    phase_end(c, PH_EVB_RECIP, sg);
    stamp(c, 6, sg);
  }
  w->pack_prefilled = false;
  if (fast && ngrid && linear && fft_lds_usable(c) && !(fft_lds_can_patch(c) && c->evb_fft_patch)) {      // start value of the blended charge grid (Q_1 on rank 0): needs no coefficient, so it goes out in front of the energy fetch (the stream idles there until the other chains have ended)
    if (!w->ev_prefill) HIPCHK(hipEventCreateWithFlags(&w->ev_prefill, hipEventDisableTiming));
    if (c->rank == 0) hipLaunchKernelGGL(k_copy_grid, dim3(256), dim3(256), 0, sg, K3, (const double*)c->d_Q.p, w->d_pack.p);      // (a kernel, not a copy command: those cost ~15 us of queue time around their 4 us in front of the energy fetch)
    else HIPCHK(hipMemsetAsync(w->d_pack.p, 0, K3 * sizeof(double), sg));
    HIPCHK(hipEventRecord(w->ev_prefill, sg));
    w->pack_prefilled = true;
  }
  CHK(fetch_A());
  if (single_join) {      // sg has waited for the environment chain and the couplings (fetch A) and holds the front end, the grid chain and the copy itself: + the per-molecule kernel
    if (!w->ev_side) HIPCHK(hipEventCreateWithFlags(&w->ev_side, hipEventDisableTiming));
    w->side_by_A = pub_spin && linear && fast && c->evb_late_mol_join;
    w->side_wo_mol = c->evb_late_mol_join && fast;      // the per-molecule kernel (aux3, behind the coupling chain) ends ~40 us after the chains: the blend phase starts without it
    if (!w->side_wo_mol) HIPCHK(hipStreamWaitEvent(sg, c->ev_join2, 0));
    HIPCHK(hipEventRecord(w->ev_side, sg));
    w->side_valid = true;
  }
  if (fast && ngrid && !linear) {      // own-stencil corrections of the special atoms -> delta store; behind the energy fetch on this stream: the host's turn does not wait for it
    hipLaunchKernelGGL(k_evb_recip_special, dim3((nra + 3) / 4), dim3(256), 0, sg, nra, d_ra, K, K3, c->d_base.p, c->d_w.p, c->d_dw.p, w->d_sbase.p,
                       w->d_sw.p, w->d_sdw.p, w->d_grids.p, c->conv, c->kk[0][0], c->kk[0][1], c->kk[0][2], c->kk[1][0], c->kk[1][1], c->kk[1][2], c->kk[2][0],
                       c->kk[2][1], c->kk[2][2], w->d_Fdelta.p, n3);
    if (c->overlap) HIPCHK(hipEventRecord(w->ev_rs, sg));
  }
  CHK(fetch_B());
  if (!fast) {      // per-element force stores (msevb_get_evb_element_force, sharding tests)
    if (ngrid) {
      if (c->overlap) HIPCHK(hipStreamWaitEvent(c->stream, w->ev_E, 0));           // packed potentials of the diabat grids
      // all-atom gather, up to 8 grids per pass
      std::vector<int> gslots(ngrid);
      for (int d = 1; d < D; ++d) if (grid_of[d] >= 0) gslots[grid_of[d]] = diag_slot(c, D, d);
      if ((double)K3 * ngrid >= 2147483648.0) FAIL(MSEVB_ERR_LIMIT, "EVB gather: K^3 * n_grids exceeds the 32-bit offset range");
      if (c->overlap) HIPCHK(hipStreamWaitEvent(c->stream, w->ev_join2, 0));      // the gather folds the real-space deltas in
      const int* permp = c->cells_valid ? c->d_perm.p : nullptr;     // cell-sorted traversal keeps the stencils of co-resident waves in L2
      for (int g0 = 0; g0 < ngrid; g0 += 8) {
        const int ng = std::min(8, ngrid - g0);
        auto kern = ng > 4 ? k_stencil_gather<8> : ng > 2 ? k_stencil_gather<4> : ng > 1 ? k_stencil_gather<2> : k_stencil_gather<1>;
        GatherSlots gs; for (int q = 0; q < 8; ++q) gs.s[q] = q < ng ? gslots[g0 + q] : 0;
        hipLaunchKernelGGL(kern, dim3(round_up8((N + 3) / 4)), dim3(256), 0, c->stream, N, K, ng, gs, permp, c->d_chg.p, c->d_base.p,
                           c->d_w.p, c->d_dw.p, w->d_pack.p, ngrid, g0, c->conv, c->kk[0][0], c->kk[0][1], c->kk[0][2], c->kk[1][0], c->kk[1][1],
                           c->kk[1][2], c->kk[2][0], c->kk[2][1], c->kk[2][2], c->d_force_recip.p, c->d_Fdiag.p, w->d_Fdelta.p, n3);
      }
      hipLaunchKernelGGL(k_evb_recip_special, dim3((nra + 3) / 4), dim3(256), 0, c->stream, nra, d_ra, K, K3, c->d_base.p, c->d_w.p, c->d_dw.p, w->d_sbase.p,
                         w->d_sw.p, w->d_sdw.p, w->d_grids.p, c->conv, c->kk[0][0], c->kk[0][1], c->kk[0][2], c->kk[1][0], c->kk[1][1], c->kk[1][2], c->kk[2][0],
                         c->kk[2][1], c->kk[2][2], c->d_Fdiag.p, n3);
      HIPCHK(hipGetLastError());
    }
    if (c->overlap && !ngrid) HIPCHK(hipStreamWaitEvent(c->stream, w->ev_join2, 0));
    hipLaunchKernelGGL(k_evb_slot0, dim3(std::min<size_t>(1024, (n3 + 255) / 256)), dim3(256), 0, c->stream, n3, c->d_force.p, w->d_Fdelta.p, c->d_Fdiag.p, w->d_F11.p);
  }
  // ---- 8. off-diagonal elements: few, latency-bound kernels writing only d_Foff -> side stream, beside the reciprocal space
  //         and the diagonal real-space deltas (forked in step 5b, joined here)
  HIPCHK(hipGetLastError());
  if (ncp && c->overlap && !single_join) HIPCHK(hipStreamWaitEvent(c->stream, w->ev_cpl, 0));      // off-diagonal force stores complete
  if (dev_solve) {      // the blend phase goes out NOW on the main stream, pre-enqueued behind the pair loop: its first kernel starts on the solve kernel's mark - no host turn, no event hop
    if (c->overlap && !pub_spin) HIPCHK(hipStreamWaitEvent(c->stream, w->ev_solve, 0));      // (without device-side marks: one event hop)
    phase_begin(c, PH_EVB_BLEND);
    w->blend_mark = pub_spin ? w->d_chain_done.p + 3 : nullptr; w->blend_mark_seq = (pub_spin && !gmode) ? w->seq_A : 0;      // (no mark, no number: a captured launch's arguments must not change from call to call)
    w->side_by_mark = pub_spin;      // that mark follows every side chain (and the Q_1 copy) of this call on the grid stream: no event waits in the phase
    const int rc_b = fast_blend_phase(c, w, D, nullptr, grid_of.data(), ngrid, nra, d_ra, w->d_coef.p);
    w->side_by_mark = false; w->blend_mark = nullptr;
    CHK(rc_b);
    if (dev_solve_mr && c->nccl_comm) CHK(msevb_comm_allreduce_forces(c));      // third collective of the call, behind the blend on the main stream - enqueued with the step, no host turn in front of it
    phase_end(c, PH_EVB_BLEND);
  }
  if (gmode) {
    if (gmode == 3) {      // learn pass: nothing has been issued and the device has not been touched; keep what was learnt, give the one-shot state back, the call again
      t_step_rec = nullptr; t_lin = nullptr; c->step_sb = nullptr;
      if (LC.bad) { w->graph_broken = true; if (getenv("MSEVB_GRAPH_DEBUG")) fprintf(stderr, "[msevb] linear capture not possible: %s\n", LC.why ? LC.why : "?"); }
      else { memcpy(G.need, LC.need, sizeof G.need); G.nev = LC.nev; G.learned = true; G.learn_serial = w->plan_serial; G.learn_topo = c->topo_epoch; }
      c->refresh_fused_nb = fused_nb_in;
      rearm.ok = true;
      return MSEVB_RETRY;
    }
    if (gmode == 1) {
      // the end of every chain: the side streams leave a final mark each, the main stream waits for the three - whatever follows the call on the main stream
      // (the integrator's second kernel, the next call's step block) follows all of the call
      for (int b = 1; b < LIN_NST; ++b) { lin_add_op(LC, b, 1, 59 + b, 1); lin_flush(LC, b); }      // (marks 60 .. 63)
      for (int b = 1; b < LIN_NST; ++b) lin_add_op(LC, 0, 2, 59 + b, 1);
      LC.head_open = false;
      // the main chain's closing WAITS are not captured: the integrator's second kernel performs them (MD loop), or a kernel of their own behind the chain;
      // whatever else is pending in front of them (a mark behind the chain's last kernel) is
      LinOps tail;
      { const LinOps all = LC.pend[0]; int cut = 0; for (int k = 0; k < all.n; ++k) if (all.kind[k] != 2) cut = k + 1;
        LC.pend[0].n = cut; lin_flush(LC, 0);
        for (int k = cut; k < all.n; ++k) { tail.kind[tail.n] = 2; tail.slot[tail.n] = all.slot[k]; tail.ord[tail.n] = all.ord[k]; tail.n++; } }
      t_step_rec = nullptr; t_lin = nullptr; c->step_sb = nullptr;
      hipGraph_t g[LIN_NST] = {nullptr, nullptr, nullptr, nullptr}; hipGraphExec_t ex[LIN_NST] = {nullptr, nullptr, nullptr, nullptr};
      const double t_inst0 = now_us();
      hipError_t e = hipSuccess;
      for (int b = 0; b < LIN_NST; ++b) { const hipError_t eb = hipStreamEndCapture(LC.st[b], &g[b]); if (eb != hipSuccess && e == hipSuccess) e = eb; }
      size_t nn[LIN_NST] = {0, 0, 0, 0};
      for (int b = 0; b < LIN_NST && e == hipSuccess && !LC.bad; ++b) { if (!g[b]) { e = hipErrorUnknown; break; } (void)hipGraphGetNodes(g[b], nullptr, &nn[b]); e = hipGraphInstantiate(&ex[b], g[b], nullptr, nullptr, 0); }
      for (int b = 0; b < LIN_NST; ++b) if (g[b]) (void)hipGraphDestroy(g[b]);
      if (e != hipSuccess || LC.bad || !ex[LIN_NST - 1]) {      // this runtime cannot capture the call: nothing has been issued - the call again, the ordinary way, and no further attempts
        for (int b = 0; b < LIN_NST; ++b) if (ex[b]) (void)hipGraphExecDestroy(ex[b]);
        (void)hipGetLastError(); w->graph_broken = true;
        fprintf(stderr, "[msevb] hipGraph capture of the MS-EVB force call failed (%s): option step_graph is off for this context\n", LC.bad ? (LC.why ? LC.why : "?") : hipGetErrorString(e));
        c->refresh_fused_nb = fused_nb_in;
        return MSEVB_RETRY;
      }
      for (int b = 0; b < LIN_NST; ++b) G.exec[b] = ex[b];
      G.head = LC.head; G.tail = tail;
      G.trace = gtrace;
      G.hash = grec.hash; G.n = grec.n; G.plan_serial = w->plan_serial; G.topo_epoch = c->topo_epoch; G.alloc_epoch = g_alloc_epoch; c->evb_graph_captures++;
      if (getenv("MSEVB_GRAPH_DEBUG")) fprintf(stderr, "[msevb] MS-EVB call captured as four linear graphs: nodes %zu %zu %zu %zu (main, grid, descriptor, coupling stream); %d events, %d marks and %d waits in %d one-thread kernels (+ the step-block kernel); end of capture + instantiation %.0f us\n", nn[0], nn[1], nn[2], nn[3], LC.nev, LC.n_marks + LIN_NST - 1, LC.n_waits + LIN_NST - 1, LC.n_op_kernels, now_us() - t_inst0);
    } else if (grec.hash != G.hash || grec.n != G.n || g_alloc_epoch != alloc_epoch0) {
      // this call would NOT have enqueued what the graphs hold (an argument, a launch geometry or a buffer changed): nothing has been issued - drop them, repeat the call the ordinary way
      t_step_rec = nullptr; t_lin = nullptr; c->step_sb = nullptr;
      if (getenv("MSEVB_GRAPH_DEBUG")) fprintf(stderr, "[msevb] replay of the captured call differs from the capture (launches %d / %d, allocations %s): call repeated the ordinary way\n", grec.n, G.n, g_alloc_epoch != alloc_epoch0 ? "changed" : "same");
      if (grec.trace) for (size_t k = 0; k < gtrace.size() && k < G.trace.size(); ++k) if (gtrace[k] != G.trace[k]) {
        const size_t tag = (size_t)gtrace[k].first;
        fprintf(stderr, "[msevb]   first difference at operation %zu: %s\n", k, tag < 4096 ? (tag % 16 == 1 ? "event record" : tag % 16 == 2 ? "stream wait" : "fill / copy") : hipKernelNameRefByPtr(gtrace[k].first, c->stream));
        break;
      }
      G.drop(); G.learned = false;
      w->graph_skip_once = true;
      ++c->evb_graph_mismatches;
      if (++w->graph_mismatch_run >= 8) w->graph_broken = true;      // (eight in a row: something in the call changes every time - not worth the dry passes)
      c->refresh_fused_nb = fused_nb_in;
      rearm.ok = true;      // (nothing was issued)
      return MSEVB_RETRY;
    }
    t_step_rec = nullptr; t_lin = nullptr; c->step_sb = nullptr;
    StepBlockVals sv; memset(&sv, 0, sizeof sv);
    sv.v[SB_R] = w->seq_R; sv.v[SB_A] = w->seq_A; sv.v[SB_S] = w->seq_S; sv.v[SB_B] = w->seq_B; sv.v[SB_EPOCH] = c->pair_epoch;
    hipLaunchKernelGGL(k_set_step_block, dim3(1), dim3(64), 0, c->stream, w->d_sb.p, sv, LC.cnt, LC.marks, G.head, LC.fault);      // (every kernel of a side chain that reads the block sits behind a wait for a mark of the main chain)
    { ApiScope aps(4); for (int b = 0; b < LIN_NST; ++b) HIPCHK(hipGraphLaunch(G.exec[b], LC.st[b])); }
    c->evb_graph_launches++; if (gmode == 2) w->graph_mismatch_run = 0;
    c->lin_cnt = LC.cnt; c->lin_marks = LC.marks; c->lin_tail = G.tail;
    if (!c->lin_tail_defer || !dev_solve) {      // nobody downstream will perform them (host-array entry points; a blend phase enqueued by the host behind this): now, as a kernel
      hipLaunchKernelGGL(k_lg_ops, dim3(1), dim3(64), 0, c->stream, LC.cnt, LC.marks, c->lin_tail, LC.fault);
      c->lin_tail.n = 0;
    }
    if (!dev_solve) {      // the host's turn follows (Jacobi, blend phase enqueued the ordinary way): the events that phase waits for were marks inside the graphs - the main chain ends behind all of them
      hipEvent_t evs[] = {w->ev_side, c->ev_join2, w->ev_join2, w->ev_rs, w->ev_E, w->ev_prefill, w->ev_cpl, w->ev_fetchA, w->ev_fetch, c->ev_join};
      for (hipEvent_t e : evs) if (e) HIPCHK(hipEventRecord(e, c->stream));
    }
  }
  HT(8);
  // ---- 9. Hamiltonian without the principal diabat's energy (H' = H - E0 I), from fetch A (or from the device-side solve)
  if (c->overlap) {
    if (spin_wait((volatile long long*)w->h_flag, w->seq_A, w->ev_fetchA, gsync)) HIPCHK(hipErrorUnknown);
  } else HIPCHK(hipStreamSynchronize(c->stream));
  if (spec && *h_verdict != 1) {      // the tree has changed (or the check never ran): this call's side chains worked on a stale topology - the whole call again, the long way
    w->plan_valid = false; w->spec_misses++; c->evb_spec_misses++;
    if (c->overlap && w->side_valid) { HIPCHK(hipStreamWaitEvent(c->stream, w->ev_side, 0)); if (w->side_wo_mol) HIPCHK(hipStreamWaitEvent(c->stream, c->ev_join2, 0)); }
    return MSEVB_RETRY;
  }
  HT(7);      // (7: the wait for fetch A alone; 9: Hamiltonian assembly; 10: Jacobi + blend enqueue / coefficient hand-over; 11: principal energy, wait for fetch B, switch)
  const double E_recip1 = w->h_E[nE];                  // this rank's own (complete) principal reciprocal energy
  const bool sharded = c->shard_principal && c->n_ranks > 1;
  const bool partial = sharded && !c->nccl_comm;       // virtual ranks (tests): the caller sums the ranks' shares
  memset(c->H, 0, sizeof c->H);
  auto unitE = [&](int iu) { return w->h_E[iu] + w->h_E[nu + iu]; };
  const double once = (partial && c->rank != 0) ? 0.0 : 1.0;      // replicated terms count once over virtual ranks
  const double Hp_norep = once * c->e_ref[c->h_mol_type[c->hydronium]];
  c->H[0][0] = Hp_norep + once * unitE(units_of[0][0]);
  for (int d = 1; d < D; ++d) {
    if (!owned(c, d)) continue;
    double e = c->e_ref[c->h_mol_type[c->hydronium]] + eref_delta[d];
    for (int iu : units_of[d]) e += units[iu].sign * unitE(iu);
    e += w->h_E[2 * nu + ncp + grid_of[d]] - (linear ? 0.0 : E_recip1);
    c->H[d][d] = e;
    c->H[c->parent[d]][d] = w->h_E[2 * nu + ncp + ngrid + 3 * cp_of[d] + 2];
    c->cpl_A[d] = w->h_E[2 * nu + ncp + ngrid + 3 * cp_of[d]];      // geometric factor A: the off-diagonal store holds dV, F_pd = -A dV
  }
  HT(9);
  w->last_ngrid = ngrid; w->last_nra = nra; w->last_d_ra = d_ra; w->last_fast = fast;
  for (int d = 0; d < D; ++d) w->last_grid_of[d] = grid_of[d];
  // ---- 10. multi-GPU: Hamiltonian entries of the other ranks' diabats
  bool energies_in = false;
  if (c->nccl_comm && !dev_solve_mr) {
    if (sharded) {      // the ranks' partial principal energies travel with the Hamiltonian entries
      if (c->overlap) { if (spin_wait((volatile long long*)(w->h_flag + 2), w->seq_B, w->ev_fetch, gsync)) HIPCHK(hipErrorUnknown); } else HIPCHK(hipStreamSynchronize(c->stream));
      if (c->rank != 0) c->h_energies[S_RECIP] = 0.0;      // replicated term: counts once
      energies_in = true;
    }
    CHK(msevb_comm_allgather_evb(c, sharded ? 1 : 0));
  }
  // E0 = potential energy of the principal diabat (fetch B); added to the diagonal and to the eigenvalue
  auto principal_energy = [&](double& E0) -> int {
    if (dev_solve_mr && c->nccl_comm && sharded) CHK(msevb_comm_wait_flag(c, (volatile long long*)(w->h_flag + 2), w->seq_B, w->stream2, "EVB energy all-reduce"));
    else if (c->overlap) { if (spin_wait((volatile long long*)(w->h_flag + 2), w->seq_B, w->ev_fetch, gsync)) HIPCHK(hipErrorUnknown); } else HIPCHK(hipStreamSynchronize(c->stream));
    if (nb_check_overflow(c)) return MSEVB_RETRY;      // a list row outgrew its stride in this call's rebuild: the whole force call is repeated
    if (dev_solve_mr && c->nccl_comm && sharded) { for (int k = 0; k < 8; ++k) c->h_energies[S_ELEC_PAIR + k] = c->h_comm_e8[k]; energies_in = true; }      // the sums of the device-side all-reduce (copied out by fetch B's kernel)
    if (sharded && !energies_in && c->rank != 0) c->h_energies[S_RECIP] = 0.0;      // sharded principal diabat: replicated terms count once (rank 0)
    double en[MSEVB_N_ENERGY];
    CHK(total_force_collect(c, en));
    E0 = en[MSEVB_E_POTENTIAL] - (1.0 - once) * c->ewald_self;
    return MSEVB_OK;
  };
  auto remember_plan = [&]() {      // what the next steps start from (speculative step)
    w->pl_D = D; memcpy(w->pl_plog, c->plog, sizeof c->plog); memcpy(w->pl_parent, c->parent, sizeof c->parent);
    w->pl_units_of = units_of; w->pl_grid_of = grid_of; w->pl_cp_of = cp_of; w->pl_eref_delta = eref_delta; w->pl_ngrid = ngrid; w->pl_ncp_big = ncp_big;
    w->plan_valid = true; w->plan_epoch = c->topo_epoch; w->plan_serial++;
  };
  if (c->n_ranks > 1 && !c->nccl_comm && !dev_solve_mr) {   // virtual sharding (tests): leave foreign diabats empty; caller merges
    double E0 = 0.0;
    CHK(principal_energy(E0));
    if (!spec && w->plan_fits && c->evb_speculate) remember_plan();      // (round 4: this return used to skip it - virtual ranks never speculated, and tools/virtual_rank_times.py timed the long way round every call)
    for (int d = 0; d < D; ++d) if (d == 0 || owned(c, d)) c->H[d][d] += E0;
    if (epot_out) *epot_out = c->H[0][0];
    if (fast) w->delta_dirty = true;      // no blend consumed the delta store: msevb_evb_partial_blend may still, the next call re-zeroes it
    if (c->overlap && w->side_valid) { HIPCHK(hipStreamWaitEvent(c->stream, w->ev_side, 0)); if (w->side_wo_mol) HIPCHK(hipStreamWaitEvent(c->stream, c->ev_join2, 0)); }      // nobody else joins the side streams for this call
    rearm.ok = true;
    return MSEVB_OK;
  }
  // ---- 11. diagonalise H', blend (diagonalize_evb_hamiltonian, :242)
  double eg = 0.0;
  if (dev_solve) {      // eigenvalue and ground-state vector of the device's Jacobi (the blend behind it is running already)
    if (dev_solve_mr && c->nccl_comm) CHK(msevb_comm_wait_flag(c, (volatile long long*)(w->h_flag + 6), w->seq_S, c->aux, "EVB all-gather"));      // (behind a collective: a missing peer must end the call, not hang it)
    else if (c->overlap) { if (spin_wait((volatile long long*)(w->h_flag + 6), w->seq_S, w->ev_solve, gsync)) HIPCHK(hipErrorUnknown); } else HIPCHK(hipStreamSynchronize(c->stream));
    const double* hs = w->h_solve;
    if (hs[0] != 0.0) FAIL(MSEVB_ERR_NUMERIC, "too many iterations in jacobi");
    eg = hs[1];
    for (int i = 0; i < D; ++i) c->eigvec[i] = hs[2 + i];
    if (dev_solve_mr) {      // the Hamiltonian every rank solved (H' from the gathered entries) and this rank's coupling factors, for the host's records
      for (int i = 0; i < D; ++i) for (int j = i; j < D; ++j) c->H[i][j] = hs[2 + 2 * D + (size_t)i * D + j];
      for (int d = 1; d < D; ++d) if (owned(c, d)) c->cpl_A[d] = hs[2 + D + d];
    }
  } else {
    std::vector<double> a((size_t)D * D), ev(D), v((size_t)D * D);
    for (int i = 0; i < D; ++i) for (int j = i; j < D; ++j) { a[(size_t)i * D + j] = c->H[i][j]; a[(size_t)j * D + i] = c->H[i][j]; }
    if (jacobi_h(D, a, ev, v) != 0) FAIL(MSEVB_ERR_NUMERIC, "too many iterations in jacobi");
    int g = 0; eg = ev[0];
    for (int i = 1; i < D; ++i) if (ev[i] < eg) { eg = ev[i]; g = i; }
    for (int i = 0; i < D; ++i) c->eigvec[i] = v[(size_t)i * D + g];
  }
  if (!dev_solve) phase_begin(c, PH_EVB_BLEND);
  if (dev_solve) {
  } else if (fast) {
    CHK(fast_blend_phase(c, w, D, c->eigvec, grid_of.data(), ngrid, nra, d_ra));
  } else {
    BlendTerms bt;
    const double w11 = blend_terms(c, D, c->eigvec, bt);
    hipLaunchKernelGGL(k_evb_blend, dim3(std::min<size_t>(2048, (n3 + 255) / 256)), dim3(256), 0, c->stream, n3, bt, (const double*)nullptr, w11, c->d_Fdiag.p, c->d_Foff.p, n3, c->d_force.p, 0, (double*)nullptr, (const int*)nullptr, 0);
  }
  HIPCHK(hipGetLastError());
  if (c->nccl_comm && !dev_solve_mr) CHK(msevb_comm_allreduce_forces(c));
  if (!dev_solve) phase_end(c, PH_EVB_BLEND);
  HT(10);
  {   // the blend is enqueued: now the principal energy (it shifts the whole diagonal, not the ground-state vector)
    double E0 = 0.0;
    CHK(principal_energy(E0));
    for (int d = 0; d < D; ++d) c->H[d][d] += E0;
    eg += E0;
  }
  int pr = 0; double sc = std::fabs(c->eigvec[0]);
  for (int i = 0; i < D; ++i) if (sc < std::fabs(c->eigvec[i])) { sc = std::fabs(c->eigvec[i]); pr = i; }
  c->principal = pr; c->new_hydronium = c->hydronium;
  for (int h = 0; h < c->max_chain; ++h) { if (c->plog[pr][h][0] < 0) break; c->new_hydronium = c->plog[pr][h][3]; }
  if (epot_out) *epot_out = eg;
  // ---- 12. principal-diabat switch (src/ms_evb.f90:218-227): adopt the new topology, rebuild the Verlet list
  if (c->new_hydronium != c->hydronium) {
    c->hop_log.push_back((int)c->evb_calls); c->hop_log.push_back(c->hydronium + 1); c->hop_log.push_back(c->new_hydronium + 1);
    // the new principal topology from the neighbourhood record as it is NOW on the host (a call that started from the recorded tree has not looked at it yet)
    if (spec || !penv.tb) {
      CHK(fetch_region());
      if (fill_plan_tables(c, w->ptab)) FAIL(MSEVB_ERR_LIMIT, "EVB: molecule-type tables exceed the fixed sizes of the enumeration");
      w->region_order.resize(w->region.size());
      for (size_t k = 0; k < w->region.size(); ++k) w->region_order[k] = (int)k;
      penv.tb = &w->ptab; penv.mol_type = c->h_mol_type.data(); penv.mol_natom = c->h_mol_natom.data(); penv.mol_atoms = c->h_mol_atoms.data(); penv.atype = c->h_atype.data();
      penv.chg = c->h_chg.data(); penv.mass = c->h_mass.data(); penv.rec = w->region.data(); penv.order = w->region_order.data(); penv.n_region = (int)w->region.size(); penv.hydronium = c->hydronium;
    }
    w->plan_valid = false;
    PTopo T; T.n_mols = 0; T.n_atoms = 0; T.hydronium = c->hydronium;
    { int perr = PE_OK; if (plan_ensure_mol(penv, T, c->hydronium, &perr) < 0) FAIL(MSEVB_ERR_LIMIT, "EVB: hydronium outside region"); }
    for (int h = 0; h < c->max_chain; ++h) { if (c->plog[pr][h][0] < 0) break; const int pe = plan_apply_hop(penv, T, T.hydronium, c->plog[pr][h][1], c->plog[pr][h][3], c->plog[pr][h][4]); if (pe) FAIL(plan_err_code(pe), plan_error_text(pe)); }
    HIPCHK(hipStreamSynchronize(c->stream));
    for (int im = 0; im < T.n_mols; ++im) {
      const PMol& m = T.mols[im];
      c->h_mol_type[m.id] = m.type; c->h_mol_natom[m.id] = m.natom;
      for (int k = 0; k < MAX_MOL_ATOMS; ++k) c->h_mol_atoms[(size_t)m.id * MAX_MOL_ATOMS + k] = k < m.natom ? m.atoms[k] : -1;
      for (int k = 0; k < m.natom; ++k) {
        const int a = m.atoms[k]; const PAtom* ha = T.find(a);
        c->h_atype[a] = ha->type; c->h_chg[a] = ha->q; c->h_mol_of[a] = m.id;
        HIPCHK(hipMemcpy(c->d_xyz.p + 3 * (size_t)a, ha->x, 24, hipMemcpyHostToDevice));
      }
    }
    c->hydronium = T.hydronium;
    CHK(upload_topology(c));
    CHK(nb_build(c));
    c->flag_verlet_list = 0;
  }
  if (!spec && w->plan_fits && c->evb_speculate && c->new_hydronium == c->hydronium) remember_plan();
  rearm.ok = true;
  HT(11); g_ht.n++;
  if (ht_on && (g_ht.n % 100) == 0) fprintf(stderr, "[msevb] speculative steps %ld, of which the tree had changed %ld\n", w->spec_steps, w->spec_misses);
  return MSEVB_OK;
}

int evb_force_energy(msevb_ctx* c, double* epot_out) {
  c->evb_calls++;
  int rc = evb_force_energy_once(c, epot_out);
  for (int k = 0; k < 6 && rc == MSEVB_RETRY; ++k) rc = evb_force_energy_once(c, epot_out);      // (tree changed -> planned on the device -> speculative call; list overflow; a dropped step graph)
  if (rc == MSEVB_RETRY) FAIL(MSEVB_ERR_LIMIT, "Verlet list rows keep overflowing");
  return rc;
}

// frees the per-context EVB scratch (called by msevb_destroy while the context's streams still exist)
void evb_release(msevb_ctx* c) {
  if (api_prof_on() && c->evb_calls) {
    static const char* nm[5] = {"kernel launches", "event records", "stream waits", "fill / copy commands", "graph launches"};
    fprintf(stderr, "[msevb api us per MS-EVB call (count per call)]");
    for (int k = 0; k < 5; ++k) fprintf(stderr, " %s %.1f (%.1f)", nm[k], g_api_prof.us[k] / c->evb_calls, (double)g_api_prof.n[k] / c->evb_calls);
    fprintf(stderr, "\n");
  }
  if (getenv("MSEVB_HOST_TIMING") && g_ht.n) { fprintf(stderr, "[msevb host us/step]"); for (int k = 0; k < 12; ++k) fprintf(stderr, " %d:%.1f", k, g_ht.t[k] / g_ht.n); fprintf(stderr, "\n"); }
  EvbWork* w = (EvbWork*)c->evb_work;
  if (w) {
    DBuf<double>* db[] = {&w->d_exch, &w->d_part, &w->d_E, &w->d_dVspec, &w->d_sw, &w->d_sdw, &w->d_Fdelta, &w->d_F11, &w->d_pack, &w->d_com_atom};
    for (auto b : db) b->release();
    w->d_stencils.release(); w->d_Elq.release(); w->d_ow.release(); w->d_odw.release(); w->d_obase.release(); w->d_tabs.release(); w->d_sbase.release(); w->d_counter.release(); w->d_grids.release(); w->d_region.release(); w->d_cand.release(); w->d_desc.release();
    hipEvent_t* ev[] = {&w->ev_chains, &w->ev_side, &w->ev_prefill, &w->ev_pairs, &w->ev_fork2, &w->ev_join2, &w->ev_E, &w->ev_fetch, &w->ev_in, &w->ev_region, &w->ev_cpl, &w->ev_flush, &w->ev_fetchA, &w->ev_solve, &w->ev_rs, &w->ev_bterms};
    for (auto e : ev) if (*e) (void)hipEventDestroy(*e);
    if (w->h_region) (void)hipHostFree(w->h_region);
    if (w->h_count) (void)hipHostFree(w->h_count);
    if (w->h_desc) (void)hipHostFree(w->h_desc);
    if (w->h_solve) (void)hipHostFree(w->h_solve);
    if (w->h_flag) (void)hipHostFree(w->h_flag);
    for (auto& g : w->sg) g.drop();
    for (hipEvent_t e : w->ev_gjoin) if (e) (void)hipEventDestroy(e);
    w->d_sb.release(); w->d_lin.release(); w->d_ptab.release(); w->d_plan_order.release(); w->d_plan_near.release(); w->d_plan_cnt.release(); w->d_cps_tmp.release(); w->d_plan_shell.release();
    if (w->h_plan_sum) (void)hipHostFree(w->h_plan_sum);
    if (w->ev_plan) (void)hipEventDestroy(w->ev_plan);
    w->d_coef.release(); w->d_plan.release(); w->d_sphere.release(); w->d_check.release(); w->d_refresh_scratch.release(); w->d_chain_done.release();
    if (w->h_E) (void)hipHostFree(w->h_E);
    delete w;
  }
  c->evb_work = nullptr;
}

extern "C" {

int msevb_ms_evb_calculate_total_force_energy(msevb_ctx* c, double* epot, double* force) {
  if (!c || !c->finalized) { if (c) c->err = "context not finalized"; return MSEVB_ERR_STATE; }
  (void)hipSetDevice(c->device);
  double e = 0;
  CHK(evb_force_energy(c, &e));
  if (force) { pin_user_buffer(c, force, 3 * (size_t)c->N * 8); CHK(download_atoms3(c, force, c->d_force.p)); }
  HIPCHK(hipStreamSynchronize(c->stream));
  if (epot) *epot = e;
  return MSEVB_OK;
}

int msevb_get_evb_state(msevb_ctx* c, int* nd, int* plog, int* cm, double* H, double* ev, int* pr, int* nh) {
  if (!c) return MSEVB_ERR_ARG;
  const int D = c->D;
  if (nd) *nd = D;
  if (plog) for (int d = 0; d < D; ++d) for (int h = 0; h < c->max_chain; ++h) for (int q = 0; q < 5; ++q) { const int v = c->plog[d][h][q]; plog[(d * c->max_chain + h) * 5 + q] = v < 0 ? -1 : v + 1; }
  if (cm) for (int d = 0; d < D; ++d) cm[d] = c->parent[d] < 0 ? -1 : c->parent[d] + 1;
  if (H) for (int i = 0; i < D; ++i) for (int j = 0; j < D; ++j) H[i * D + j] = c->H[i][j];
  if (ev) for (int d = 0; d < D; ++d) ev[d] = c->eigvec[d];
  if (pr) *pr = c->principal + 1;
  if (nh) *nh = c->new_hydronium + 1;
  return MSEVB_OK;
}

int msevb_evb_partial_blend(msevb_ctx* c, const double* eigvec, double* force) {
  if (!c || c->D <= 0 || !eigvec) { if (c) c->err = "bad argument"; return MSEVB_ERR_ARG; }      // force == NULL: the blend stays on the device
  EvbWork* w = (EvbWork*)c->evb_work;
  if (!w) FAIL(MSEVB_ERR_STATE, "no EVB force call yet");
  const size_t n3 = 3 * (size_t)c->N;
  if (w->last_fast) {      // fast path: consumes the delta store and the force accumulator of the last call - ONE call per force call
    CHK(fast_blend_phase(c, w, c->D, eigvec, w->last_grid_of, w->last_ngrid, w->last_nra, w->last_d_ra));
    w->delta_dirty = false;      // the blend re-zeroed every slot it read
  } else {
    if (!w->d_F11.p) FAIL(MSEVB_ERR_STATE, "no EVB force call yet");
    BlendTerms bt;
    const double w11 = blend_terms(c, c->D, eigvec, bt);
    HIPCHK(hipMemcpyAsync(c->d_force.p, w->d_F11.p, n3 * 8, hipMemcpyDeviceToDevice, c->stream));
    hipLaunchKernelGGL(k_evb_blend, dim3(std::min<size_t>(2048, (n3 + 255) / 256)), dim3(256), 0, c->stream, n3, bt, (const double*)nullptr, w11, c->d_Fdiag.p, c->d_Foff.p, n3, c->d_force.p, 0, (double*)nullptr, (const int*)nullptr, 0);
  }
  if (force) HIPCHK(hipMemcpyAsync(force, c->d_force.p, n3 * 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return MSEVB_OK;
}

int msevb_get_evb_element_force(msevb_ctx* c, int i, int j, double* force) {
  if (!c || c->D <= 0 || i < 1 || j < i || j > c->D) { if (c) c->err = "bad element"; return MSEVB_ERR_ARG; }
  if (!c->evb_element_stores) FAIL(MSEVB_ERR_STATE, "per-element force stores are off: msevb_set_option(ctx, \"evb_element_stores\", 1) before the EVB force call");
  const size_t n3 = 3 * (size_t)c->N;
  if (i == j) {      // diagonal: principal force (copy kept by k_evb_slot0) + the stored difference; zero on ranks that do not own the diabat
    EvbWork* w11 = (EvbWork*)c->evb_work;
    if (!w11 || !w11->d_F11.p) FAIL(MSEVB_ERR_STATE, "no EVB force call yet");
    if (i > 1 && !owned(c, i - 1)) { memset(force, 0, n3 * 8); return MSEVB_OK; }
    std::vector<double> d(n3);
    HIPCHK(hipMemcpyAsync(force, w11->d_F11.p, n3 * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(d.data(), c->d_Fdiag.p + (size_t)diag_slot(c, c->D, i - 1) * n3, n3 * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    for (size_t k = 0; k < n3; ++k) force[k] = force[k] + d[k];
    return MSEVB_OK;
  }
  if (c->parent[j - 1] != i - 1) FAIL(MSEVB_ERR_ARG, "Hamiltonian element is zero");
  if (!owned(c, j - 1)) { memset(force, 0, n3 * 8); return MSEVB_OK; }
  HIPCHK(hipMemcpyAsync(force, c->d_Foff.p + (size_t)off_slot(c, c->D, j - 1) * n3, n3 * 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  for (size_t k = 0; k < n3; ++k) force[k] *= -c->cpl_A[j - 1];      // the store holds dV of the coupling; F_pd = -A dV
  return MSEVB_OK;
}

// principal-diabat switches since the last call with clear != 0: log(3, n) = {EVB force call number, old, new hydronium molecule}
int msevb_get_hop_log(msevb_ctx* c, int* n_hops, int* log, int capacity, long* evb_calls, int clear) {
  if (!c) return MSEVB_ERR_ARG;
  const int n = (int)(c->hop_log.size() / 3);
  if (n_hops) *n_hops = n;
  if (evb_calls) *evb_calls = c->evb_calls;
  if (log) for (int k = 0; k < 3 * std::min(n, capacity); ++k) log[k] = c->hop_log[k];
  if (clear) c->hop_log.clear();
  return MSEVB_OK;
}
// virtual ranks: what a real all-gather would deliver - the R payloads of msevb_evb_payload_pack in rank order (this rank's own slot is overwritten by its energy-fetch
// kernel every call).  With them a virtual rank runs the SAME device-side path as a rank with a communicator - solve on the gathered entries, pre-enqueued blend - minus the
// collectives; n_doubles = 0 takes them away again (the caller then merges the ranks' shares with msevb_evb_partial_blend, as before).
int msevb_comm_set_virtual_payload(msevb_ctx* c, const double* all, int n_doubles) {
  if (!c || n_doubles < 0 || (n_doubles > 0 && !all)) return MSEVB_ERR_ARG;
  c->virt_payload.assign(all, all + n_doubles); c->virt_payload_dirty = n_doubles > 0;
  return MSEVB_OK;
}
// ... and what this rank's energy-fetch kernel wrote (all R slots of the buffer as they are on the device; the rank's own slot is the one that is its work): with it
// a test plays the all-gather between virtual ranks itself
int msevb_comm_get_payload(msevb_ctx* c, double* out, int n_doubles) {
  if (!c || !out || n_doubles < 0 || (size_t)n_doubles > c->d_comm_h.n) return MSEVB_ERR_ARG;
  (void)hipSetDevice(c->device);
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(out, c->d_comm_h.p, (size_t)n_doubles * 8, hipMemcpyDeviceToHost));
  return MSEVB_OK;
}
int msevb_comm_set_virtual(msevb_ctx* c, int r, int n) { if (!c || n < 1 || r < 0 || r >= n) return MSEVB_ERR_ARG; c->rank = r; c->n_ranks = n; return MSEVB_OK; }

}  // extern "C"
