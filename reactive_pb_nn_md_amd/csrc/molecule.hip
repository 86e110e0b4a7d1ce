// Per-molecule terms: bonded (bond / angle / dihedral; replaces intra_molecular_energy_force,
// src/intra_bonded_interactions.f90:17-552) and intra-molecular non-bonded with exclusions (replaces
// intra_molecular_pairwise_energy_force, src/pair_int_real_space.f90:386-588, and intra_pme_exclusion :781).
// One thread per molecule (forces are added with fp64 atomics because the kernel may overlap the pair loop); all parameters of a
// molecule type are pre-resolved into one MolTypeDev record (L2 / scalar-cache resident, <= 10 types).
#include "ctx.hpp"
#include "devmath.hpp"

#include "molecule_dev.hpp"

namespace {

constexpr int MOL_TPB = 64;      // threads per block: 3 x 8 x 3 LDS doubles per thread

__global__ __launch_bounds__(MOL_TPB) void k_molecules(int m_lo, int M, const int* __restrict__ mol_type, const int* __restrict__ mol_natom,
                                                   const int* __restrict__ mol_atoms, const MolTypeDev* __restrict__ mts,
                                                   const double* __restrict__ xyz, const double* __restrict__ chg, double* __restrict__ F,
                                                   const double2* __restrict__ etab, double inv_dx, const double* __restrict__ tt,
                                                   const double* __restrict__ dtt, double rc2, double alpha, double erf_factor, double conv,
                                                   double pi_ref, int do_nb, int do_bonded, double* __restrict__ partials, int slot0) {
  __shared__ double red[5][MOL_TPB];
  __shared__ double sx[MAX_MOL_ATOMS * 3 * MOL_TPB], sf[MAX_MOL_ATOMS * 3 * MOL_TPB], sq[MAX_MOL_ATOMS * MOL_TPB];
  const int m = m_lo + blockIdx.x * blockDim.x + threadIdx.x;
  double e[5] = {0, 0, 0, 0, 0};
  if (m < M) {
    const MolTypeDev& mt = mts[mol_type[m]];
    const int n = mol_natom[m];
    const LdsRows<MOL_TPB> x{sx + threadIdx.x}, f{sf + threadIdx.x};
    const LdsVec<MOL_TPB> q{sq + threadIdx.x};
    for (int k = 0; k < n; ++k) {
      const int a = mol_atoms[m * MAX_MOL_ATOMS + k];
      q[k] = chg[a];
      for (int d = 0; d < 3; ++d) { x[k][d] = xyz[3 * a + d]; f[k][d] = 0.0; }
    }
    if (do_nb && n > 1) molecule_nonbonded(mt, n, x, q, f, etab, inv_dx, tt, dtt, rc2, alpha, erf_factor, conv, e[0], e[1]);
    if (do_bonded) molecule_bonded(mt, x, f, pi_ref, e[2], e[3], e[4]);
    for (int k = 0; k < n; ++k) { const int a = mol_atoms[m * MAX_MOL_ATOMS + k]; for (int d = 0; d < 3; ++d) atomicAdd(&F[3 * a + d], f[k][d]); }   // may run beside the pair loop
  }
  for (int s = 0; s < 5; ++s) red[s][threadIdx.x] = e[s];
  __syncthreads();
  for (int st = MOL_TPB / 2; st > 0; st >>= 1) {
    if (threadIdx.x < st) for (int s = 0; s < 5; ++s) red[s][threadIdx.x] += red[s][threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x < 5) partials[(size_t)(slot0 + threadIdx.x) * MAX_BLOCKS + blockIdx.x] = red[threadIdx.x][0];
}

}  // namespace

// slots slot0..slot0+4 = E_elec(intra), E_vdw(intra), E_bond, E_angle, E_dihedral
int molecule_forces(msevb_ctx* c, int slot0, bool do_nb, bool do_bonded, hipStream_t st) {
  int m_lo, m_hi;
  shard_range(c, c->M, m_lo, m_hi);
  const int nb = (m_hi - m_lo + MOL_TPB - 1) / MOL_TPB;
  if (!st) st = c->stream;
  if (nb <= 0) { for (int s = 0; s < 5; ++s) CHK(reduce_partials(c, slot0 + s, 0)); return MSEVB_OK; }      // more ranks than molecules: this rank's share is empty (a launch with no blocks is an error)
  phase_begin(c, PH_MOLECULE, st);
  hipLaunchKernelGGL(k_molecules, dim3(nb), dim3(MOL_TPB), 0, st, m_lo, m_hi, c->d_mol_type.p, c->d_mol_natom.p, c->d_mol_atoms.p, c->d_moltypes.p,
                     c->d_xyz.p, c->d_chg.p, c->d_force.p, c->d_ewald_tab.p, 1.0 / c->erfc_dx, c->d_tt.p, c->d_dtt.p, c->rc * c->rc, c->alpha,
                     2.0 * c->alpha / c->pi_sqrt, c->conv, c->pi, do_nb ? 1 : 0, do_bonded ? 1 : 0, c->d_partials.p, slot0);
  HIPCHK(hipGetLastError());
  phase_end(c, PH_MOLECULE, st);
  for (int s = 0; s < 5; ++s) CHK(reduce_partials(c, slot0 + s, nb));
  return MSEVB_OK;
}
