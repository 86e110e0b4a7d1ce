// Per-molecule terms: bonded (bond / angle / dihedral; replaces intra_molecular_energy_force,
// src/intra_bonded_interactions.f90:17-552) and intra-molecular non-bonded with exclusions (replaces
// intra_molecular_pairwise_energy_force, src/pair_int_real_space.f90:386-588, and intra_pme_exclusion :781).
// Four lanes per molecule (forces are added with fp64 atomics because the kernel may overlap the pair loop); all parameters of a
// molecule type are pre-resolved into one MolTypeDev record (L2 / scalar-cache resident, <= 10 types).
#include "ctx.hpp"
#include "devmath.hpp"

#include "molecule_dev.hpp"

namespace {

constexpr int MOL_LANES = 4;        // lanes per molecule (lane k owns atoms k, k + 4 for the loads and the force write-out).  Eight lanes (rounds 2-3) left five of them idle
                                    // on a water - three pairs, two bonds, one angle - and the kernel is instruction-bound: half the waves, same work per busy lane
constexpr int MOL_TPB = 256;       // 64 molecules per workgroup
static_assert(MAX_MOL_ATOMS % MOL_LANES == 0, "lane k of a molecule's group handles atoms k, k + MOL_LANES, ...");

// a molecule's [atom][3] / [atom] arrays in LDS, shared by the lanes of its group (x[i][d] / q[i] syntax of the molecule functions)
struct MolRows { double* p; __device__ __forceinline__ double* operator[](int i) const { return p + 3 * i; } };
struct MolVec { double* p; __device__ __forceinline__ double& operator[](int i) const { return p[i]; } };

// FOUR LANES PER MOLECULE: the pairs, bonds, angles and dihedrals of a molecule are dealt to the lanes of its group (molecule_dev.hpp, SHARED),
// forces meet in LDS.  One thread per molecule was 512 waves for 32 k molecules - two per CU, each walking a water's three erfc pairs, two bonds
// and an angle one after the other: 47 us of pure latency (77 us beside the pair loop).
__global__ __launch_bounds__(MOL_TPB) void k_molecules(int m_lo, int M, const int* __restrict__ mol_type, const int* __restrict__ mol_natom,
                                                   const int* __restrict__ mol_atoms, const MolTypeDev* __restrict__ mts,
                                                   const double* __restrict__ xyz, const double* __restrict__ chg, double* __restrict__ F,
                                                   const double2* __restrict__ etab, double inv_dx, const double* __restrict__ tt,
                                                   const double* __restrict__ dtt, double rc2, double alpha, double erf_factor, double conv,
                                                   double pi_ref, int do_nb, int do_bonded, double* __restrict__ partials, int slot0) {
  constexpr int MPB = MOL_TPB / MOL_LANES;
  __shared__ double red[5][MOL_TPB / 64];
  __shared__ double sx[MPB][MAX_MOL_ATOMS * 3], sf[MPB][MAX_MOL_ATOMS * 3], sq[MPB][MAX_MOL_ATOMS];
  const int g = threadIdx.x / MOL_LANES, k = threadIdx.x % MOL_LANES;
  const int m = m_lo + blockIdx.x * MPB + g;
  double e[5] = {0, 0, 0, 0, 0};
  const bool live = m < M;
  const int n = live ? mol_natom[m] : 0;
  int a[MAX_MOL_ATOMS / MOL_LANES];
#pragma unroll
  for (int u = 0; u < MAX_MOL_ATOMS / MOL_LANES; ++u) {
    const int kk = k + u * MOL_LANES;
    a[u] = -1;
    if (kk < n) {
      a[u] = mol_atoms[m * MAX_MOL_ATOMS + kk];
      sq[g][kk] = chg[a[u]];
      for (int d = 0; d < 3; ++d) { sx[g][3 * kk + d] = xyz[3 * a[u] + d]; sf[g][3 * kk + d] = 0.0; }
    }
  }
  __syncthreads();
  if (live) {
    const MolTypeDev& mt = mts[mol_type[m]];
    const MolRows x{sx[g]}, f{sf[g]};
    const MolVec q{sq[g]};
    if (do_nb && n > 1) molecule_nonbonded<true>(mt, n, x, q, f, etab, inv_dx, tt, dtt, rc2, alpha, erf_factor, conv, e[0], e[1], k, MOL_LANES);
    if (do_bonded) molecule_bonded<true>(mt, x, f, pi_ref, e[2], e[3], e[4], k, MOL_LANES);
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < MAX_MOL_ATOMS / MOL_LANES; ++u)
    if (a[u] >= 0) for (int d = 0; d < 3; ++d) atomicAdd(&F[3 * a[u] + d], sf[g][3 * (k + u * MOL_LANES) + d]);   // may run beside the pair loop
  for (int s = 0; s < 5; ++s) { double v = e[s]; for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64); if ((threadIdx.x & 63) == 0) red[s][threadIdx.x >> 6] = v; }
  __syncthreads();
  if (threadIdx.x < 5) { double v = 0.0; for (int w = 0; w < MOL_TPB / 64; ++w) v += red[threadIdx.x][w]; partials[(size_t)(slot0 + threadIdx.x) * MAX_BLOCKS + blockIdx.x] = v; }
}

}  // namespace

// slots slot0..slot0+4 = E_elec(intra), E_vdw(intra), E_bond, E_angle, E_dihedral
int molecule_forces(msevb_ctx* c, int slot0, bool do_nb, bool do_bonded, hipStream_t st) {
  int m_lo, m_hi;
  shard_range(c, c->M, m_lo, m_hi);
  const int mpb = MOL_TPB / MOL_LANES, nb = (m_hi - m_lo + mpb - 1) / mpb;
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
