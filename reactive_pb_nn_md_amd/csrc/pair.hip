// Real-space pair loop (replaces pairwise_real_space_verlet, src/pair_int_real_space.f90:135, with its three
// interaction kernels: Ewald erfc table :698/:740, LJ :621, SAPT Buckingham + Tang-Toennies :651).
//
// MI355X mapping: ONE WAVEFRONT PER ATOM.  The atom's own data is wave-uniform, its CSR neighbour row is
// streamed with coalesced 256-B reads (64 lanes x int32), neighbour {x,y,z,q} records are 32-B gathers from the
// cell-sorted array (L2-resident neighbourhoods), the (type_i,type_j) parameter table lives in LDS, and the
// per-atom force is reduced across the 64 lanes with shuffles - the list is a full list, so no fp64 atomics.
// Algorithmic bytes per step: 4 B per list entry + 36 B per atom read, 24 B per atom written.
#include "ctx.hpp"
#include "devmath.hpp"

namespace {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <bool SAPT>
__global__ __launch_bounds__(256) void k_pair(int N, const double4* __restrict__ posq, const int* __restrict__ type_s,
                                              const unsigned* __restrict__ nbr_start, const int* __restrict__ nbr,
                                              const PairParam* __restrict__ pairpar, int nt, const double2* __restrict__ etab, double inv_dx,
                                              const double* __restrict__ tt, const double* __restrict__ dtt, BoxDev box, double rc2,
                                              const int* __restrict__ perm, double* __restrict__ F, double* __restrict__ part_el,
                                              double* __restrict__ part_vdw) {
  extern __shared__ PairParam sp[];
  for (int k = threadIdx.x; k < nt * nt; k += blockDim.x) sp[k] = pairpar[k];
  __shared__ double red[2][4];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
  double e_el = 0.0, e_vdw = 0.0;
  // contiguous chunk of sorted atoms per block, blocks in XCD-aware order (neighbourhoods of one slab stay in one L2)
  const int per_block = (N + gridDim.x - 1) / gridDim.x;
  const int i0 = xcd_block(blockIdx.x, gridDim.x) * per_block, i1 = min(N, i0 + per_block);
  for (int i = i0 + wave; i < i1; i += wpb) {
    const double4 pi = posq[i];
    const int ti = type_s[i];
    const unsigned s = nbr_start[i], e = nbr_start[i + 1];
    double fx = 0.0, fy = 0.0, fz = 0.0;
    // two list entries per lane and trip: both index loads, then both {x,y,z,q} gathers are in flight before the arithmetic
    for (unsigned k = s + lane; k < e; k += 128) {
      const bool two = (k + 64) < e;
      const unsigned e0 = (unsigned)nbr[k], e1 = two ? (unsigned)nbr[k + 64] : e0;
      const double4 pa = posq[e0 & NBR_IDX_MASK], pb = posq[e1 & NBR_IDX_MASK];
      const int ta = (int)(e0 >> NBR_IDX_BITS), tb = (int)(e1 >> NBR_IDX_BITS);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const double4 pj = h ? pb : pa;
        const int tj = h ? tb : ta;
        double dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
        dx -= box.L[0] * floor(dx * box.invL[0] + 0.5);
        dy -= box.L[1] * floor(dy * box.invL[1] + 0.5);
        dz -= box.L[2] * floor(dz * box.invL[2] + 0.5);
        const double r2 = dx * dx + dy * dy + dz * dz;
        if (r2 < rc2 && (h == 0 || two)) {
          double fs, ee, ev;
          pair_interaction<SAPT>(r2, pi.w * pj.w, sp[ti * nt + tj], etab, inv_dx, tt, dtt, fs, ee, ev);
          fx += fs * dx; fy += fs * dy; fz += fs * dz;
          e_el += ee; e_vdw += ev;
        }
      }
    }
    fx = wave_sum(fx); fy = wave_sum(fy); fz = wave_sum(fz);
    if (lane == 0) {
      const int o = perm[i];
      atomicAdd(&F[3 * o], fx); atomicAdd(&F[3 * o + 1], fy); atomicAdd(&F[3 * o + 2], fz);   // the per-molecule kernel may run concurrently
    }
  }
  e_el = wave_sum(e_el);
  e_vdw = wave_sum(e_vdw);
  if (lane == 0) { red[0][wave] = e_el; red[1][wave] = e_vdw; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0, b = 0;
    for (int w = 0; w < wpb; ++w) { a += red[0][w]; b += red[1][w]; }
    part_el[blockIdx.x] = 0.5 * a;   // full list: every pair seen from both atoms
    part_vdw[blockIdx.x] = 0.5 * b;
  }
}

}  // namespace

int pair_forces(msevb_ctx* c, int slot_elec, int slot_vdw) {
  const int N = c->N;
  if (!c->list_valid) FAIL(MSEVB_ERR_STATE, "pair_forces: no Verlet list (call msevb_construct_verlet_list)");
  BoxDev box;
  for (int d = 0; d < 3; ++d) { box.L[d] = c->box[d]; box.invL[d] = 1.0 / c->box[d]; }
  int nblocks = round_up8((N + 7) / 8);      // 8 atoms per block, two per wave
  if (nblocks > 16384) nblocks = 16384;
  phase_begin(c, PH_PAIR);
  // force fields without a live SAPT term (every Buckingham coefficient zero, as in the CH3SO3H/H2O example) run the lean instantiation
  auto kern = c->has_sapt ? k_pair<true> : k_pair<false>;
  hipLaunchKernelGGL(kern, dim3(nblocks), dim3(256), c->nt * c->nt * sizeof(PairParam), c->stream, N, c->d_posq.p, c->d_type_sorted.p,
                     c->d_nbr_start.p, c->d_nbr.p, c->d_pair_par.p, c->nt, c->d_ewald_tab.p, 1.0 / c->erfc_dx, c->d_tt.p, c->d_dtt.p, box,
                     c->rc * c->rc, c->d_perm.p, c->d_force.p, c->d_partials.p + (size_t)slot_elec * MAX_BLOCKS,
                     c->d_partials.p + (size_t)slot_vdw * MAX_BLOCKS);
  HIPCHK(hipGetLastError());
  phase_end(c, PH_PAIR);
  CHK(reduce_partials(c, slot_elec, nblocks));
  CHK(reduce_partials(c, slot_vdw, nblocks));
  return MSEVB_OK;
}
