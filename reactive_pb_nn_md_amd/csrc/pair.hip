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


constexpr unsigned NBR_NONE = 0xffffffffu;     // padding entry of a partial trip

// Minimum-image separation and r^2 with the reference's own operation order and NO FMA contraction (src/pair_int_real_space.f90:222-226:
// dr - box*floor(dr/box + 0.5), dot_product): the cutoff decision r2 < rc2 is a discrete one - coordinates read from a .gro are
// multiples of 0.01 A, so pairs sit EXACTLY on the cutoff (about 30 of them at 98 k atoms, 280 at 985 k) and the last bit of r2 decides
// whether such a pair contributes its 3e-3 kJ/mol.  Everything after the decision may round differently (tolerances of DESIGN.md 5).
__device__ __forceinline__ double min_image_r2(const double4& pi, const double4& pj, const BoxDev& box, double& dx, double& dy, double& dz) {
#pragma clang fp contract(off)
  dx = pi.x - pj.x; dy = pi.y - pj.y; dz = pi.z - pj.z;
  dx = dx - box.L[0] * floor(dx * box.invL[0] + 0.5); dy = dy - box.L[1] * floor(dy * box.invL[1] + 0.5); dz = dz - box.L[2] * floor(dz * box.invL[2] + 0.5);
  return (dx * dx + dy * dy) + dz * dz;
}

// One atom's list row, 64 lanes x 2 entries per trip; the entries of the NEXT trip are loaded before the current trip is evaluated
// and the Ewald table rows of both entries of a trip are fetched before either pair is finished.  `src` is the {x,y,z,q} array the
// entries index: the cell-sorted array in HBM (k_pair) or the cell-neighbourhood tile in LDS (k_pair_tiles).
// PRUNE: when `out` is set, the entries within sqrt(rp2) are compacted into out[0 .. n_out) on the way (the rolling pruned list).
template <bool SAPT, bool PRUNE = false, class PosPtr>
__device__ __forceinline__ void pair_row(PosPtr src, const double4 pi, int ti, unsigned s, unsigned e, int lane, const int* __restrict__ nbr,
                                         const PairParam* sp, int nt, const double2* __restrict__ etab, double inv_dx, int emax,
                                         const double* __restrict__ tt, const double* __restrict__ dtt, const BoxDev& box, double rc2,
                                         double& fx, double& fy, double& fz, double& e_el, double& e_vdw, int* __restrict__ out = nullptr, double rp2 = 0.0,
                                         unsigned* n_out = nullptr) {
#pragma clang fp contract(fast)
  unsigned k = s + lane;
  unsigned e0 = k < e ? (unsigned)nbr[k] : NBR_NONE, e1 = k + 64 < e ? (unsigned)nbr[k + 64] : NBR_NONE;
  for (unsigned k0 = s; k0 < e; k0 += 128) {      // wave-uniform trip count
    const unsigned kn = k + 128;
    const unsigned n0 = kn < e ? (unsigned)nbr[kn] : NBR_NONE, n1 = kn + 64 < e ? (unsigned)nbr[kn + 64] : NBR_NONE;
    const bool va = e0 != NBR_NONE, vb = e1 != NBR_NONE;
    const double4 pa = src[va ? (e0 & NBR_IDX_MASK) : 0u], pb = src[vb ? (e1 & NBR_IDX_MASK) : 0u];
    double dxa, dya, dza, dxb, dyb, dzb;
    const double r2a = min_image_r2(pi, pa, box, dxa, dya, dza), r2b = min_image_r2(pi, pb, box, dxb, dyb, dzb);
    const bool ina = va && r2a < rc2, inb = vb && r2b < rc2;
    if (PRUNE && out) {      // wave-uniform branch
      const bool ka = va && r2a < rp2, kb = vb && r2b < rp2;
      const unsigned long long ma = __ballot(ka), mb = __ballot(kb);
      unsigned tot = *n_out;
      if (ka) out[tot + __builtin_amdgcn_mbcnt_hi((unsigned)(ma >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)ma, 0u))] = (int)e0;
      tot += (unsigned)__popcll(ma);
      if (kb) out[tot + __builtin_amdgcn_mbcnt_hi((unsigned)(mb >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mb, 0u))] = (int)e1;
      *n_out = tot + (unsigned)__popcll(mb);
    }
    EwaldRows wa, wb;
    ewald_fetch(ina ? r2a : rc2, inv_dx, emax, etab, wa);
    ewald_fetch(inb ? r2b : rc2, inv_dx, emax, etab, wb);
    if (ina) {
      double fs, ee, ev;
      pair_interaction_rows<SAPT>(r2a, pi.w * pa.w, sp[ti * nt + (int)(e0 >> NBR_IDX_BITS)], wa, tt, dtt, fs, ee, ev);
      fx += fs * dxa; fy += fs * dya; fz += fs * dza; e_el += ee; e_vdw += ev;
    }
    if (inb) {
      double fs, ee, ev;
      pair_interaction_rows<SAPT>(r2b, pi.w * pb.w, sp[ti * nt + (int)(e1 >> NBR_IDX_BITS)], wb, tt, dtt, fs, ee, ev);
      fx += fs * dxb; fy += fs * dyb; fz += fs * dzb; e_el += ee; e_vdw += ev;
    }
    e0 = n0; e1 = n1; k = kn;
  }
}

// PRUNE: the rolling pruned list.  prune_state[0] (written on this stream by k_top2_final / nb_prune_reset just before) selects for the whole
// launch: 2 = run on the pruned rows nbr2 / count2 (pairs within rc + delta when they were written); 1 = run on the full rows and, on the
// way, rewrite the pruned rows and the position snapshot the displacement check refers to.  The set of pairs inside rc is the same
// either way; ~25 % of the entries of a full row lie outside rc + delta and cost a gather each.
template <bool SAPT, bool PRUNE>
__global__ __launch_bounds__(256) void k_pair(int i_lo, int N, const double4* __restrict__ posq, const int* __restrict__ type_s,
                                              const unsigned* __restrict__ nbr_start, const unsigned* __restrict__ nbr_count, const int* __restrict__ nbr,
                                              const PairParam* __restrict__ pairpar, int nt, const double2* __restrict__ etab, double inv_dx, int emax,
                                              const double* __restrict__ tt, const double* __restrict__ dtt, BoxDev box, double rc2,
                                              const int* __restrict__ perm, double* __restrict__ F, double* __restrict__ part_el,
                                              double* __restrict__ part_vdw, int* __restrict__ prune_state, int* __restrict__ nbr2,
                                              unsigned* __restrict__ count2, double4* __restrict__ posq_prune, double rp2, int N_all) {
  extern __shared__ PairParam sp[];
  for (int k = threadIdx.x; k < nt * nt; k += blockDim.x) sp[k] = pairpar[k];
  __shared__ double red[2][4];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), wpb = blockDim.x >> 6;
  double e_el = 0.0, e_vdw = 0.0;
  // contiguous chunk of sorted atoms per block, blocks in XCD-aware order (neighbourhoods of one slab stay in one L2)
  const int per_block = (N - i_lo + gridDim.x - 1) / gridDim.x;      // [i_lo, N): this rank's share of the sorted atoms
  const int i0 = i_lo + xcd_block(blockIdx.x, gridDim.x) * per_block, i1 = min(N, i0 + per_block);
  const int mode = PRUNE ? __builtin_amdgcn_readfirstlane(prune_state[0]) : 0;
  if (PRUNE && mode != 2) {      // snapshot of ALL positions (the displacement check runs over every atom, also when the rows are sharded over ranks)
    const int all = N_all, chunk = (all + gridDim.x - 1) / gridDim.x, k0 = blockIdx.x * chunk, k1 = min(all, k0 + chunk);
    for (int k = k0 + threadIdx.x; k < k1; k += blockDim.x) posq_prune[k] = posq[k];
    if (blockIdx.x == 0 && threadIdx.x == 0) prune_state[2] = 1;      // seen by the NEXT step's k_top2_final: pruned rows exist for these full rows
  }
  for (int i = i0 + wave; i < i1; i += wpb) {
    const double4 pi = posq[i];
    const int ti = type_s[i];
    const unsigned s = nbr_start[i];
    double fx = 0.0, fy = 0.0, fz = 0.0;
    if (PRUNE && mode == 2) {
      pair_row<SAPT>(posq, pi, ti, s, s + count2[i], lane, nbr2, sp, nt, etab, inv_dx, emax, tt, dtt, box, rc2, fx, fy, fz, e_el, e_vdw);
    } else if (PRUNE) {
      unsigned n2 = 0;
      pair_row<SAPT, true>(posq, pi, ti, s, s + nbr_count[i], lane, nbr, sp, nt, etab, inv_dx, emax, tt, dtt, box, rc2, fx, fy, fz, e_el, e_vdw, nbr2 + s, rp2, &n2);
      if (lane == 0) count2[i] = n2;
    } else {
      pair_row<SAPT>(posq, pi, ti, s, s + nbr_count[i], lane, nbr, sp, nt, etab, inv_dx, emax, tt, dtt, box, rc2, fx, fy, fz, e_el, e_vdw);
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


// LDS-staged variant: ONE WORKGROUP PER CELL.  The {x,y,z,q} records of the cell's whole neighbourhood (its <= 27 neighbour
// cells, contiguous ranges of the cell-sorted array) are staged once into an LDS tile with coalesced 32-B loads; the list
// entries of the cell's atoms are LOCAL indices into that tile, so the per-pair neighbour gather is two ds_read_b128 instead
// of two divergent global loads (the global-gather kernel above is bound by the L1 tag rate: one 64-B line per lane and load).
// Waves of the block take the cell's atoms round-robin; everything else is as in k_pair.
template <bool SAPT, int THREADS>
__global__ __launch_bounds__(THREADS) void k_pair_tiles(int3 nc, const int* __restrict__ cell_start, const double4* __restrict__ posq, const int* __restrict__ type_s,
                                                     const unsigned* __restrict__ nbr_start, const unsigned* __restrict__ nbr_count, const int* __restrict__ nbr, const PairParam* __restrict__ pairpar, int nt,
                                                     const double2* __restrict__ etab, double inv_dx, int emax, const double* __restrict__ tt, const double* __restrict__ dtt,
                                                     BoxDev box, double rc2, const int* __restrict__ perm, double* __restrict__ F, double* __restrict__ part_el,
                                                     double* __restrict__ part_vdw) {
  extern __shared__ double4 tile[];                     // [tile atoms] then the pair-parameter table
  __shared__ int s_lo[27], s_off[28], s_ns;
  __shared__ double red[2][16];
  const int cidx = blockIdx.x;
  if (threadIdx.x == 0) {
    int slots[27];
    const int ns = cell_neighbourhood(cidx, nc.x, nc.y, nc.z, slots);
    int off = 0;
    for (int s = 0; s < ns; ++s) { s_lo[s] = cell_start[slots[s]]; s_off[s] = off; off += cell_start[slots[s] + 1] - s_lo[s]; }
    s_off[ns] = off; s_ns = ns;
  }
  __syncthreads();
  const int ns = s_ns, ntile = s_off[ns];
  PairParam* sp = (PairParam*)(tile + ntile);
  for (int k = threadIdx.x; k < nt * nt; k += blockDim.x) sp[k] = pairpar[k];
  for (int s = 0; s < ns; ++s) {
    const int lo = s_lo[s], n = s_off[s + 1] - s_off[s], o = s_off[s];
    for (int k = threadIdx.x; k < n; k += blockDim.x) tile[o + k] = posq[lo + k];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), wpb = blockDim.x >> 6;
  double e_el = 0.0, e_vdw = 0.0;
  const int i1 = cell_start[cidx + 1];
  for (int i = cell_start[cidx] + wave; i < i1; i += wpb) {
    const double4 pi = posq[i];
    const int ti = type_s[i];
    const unsigned s = nbr_start[i], e = s + nbr_count[i];
    double fx = 0.0, fy = 0.0, fz = 0.0;
    pair_row<SAPT>(tile, pi, ti, s, e, lane, nbr, sp, nt, etab, inv_dx, emax, tt, dtt, box, rc2, fx, fy, fz, e_el, e_vdw);
    fx = wave_sum(fx); fy = wave_sum(fy); fz = wave_sum(fz);
    if (lane == 0) {
      const int o = perm[i];
      atomicAdd(&F[3 * o], fx); atomicAdd(&F[3 * o + 1], fy); atomicAdd(&F[3 * o + 2], fz);
    }
  }
  e_el = wave_sum(e_el);
  e_vdw = wave_sum(e_vdw);
  if (lane == 0) { red[0][wave] = e_el; red[1][wave] = e_vdw; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0, b = 0;
    for (int w = 0; w < wpb; ++w) { a += red[0][w]; b += red[1][w]; }
    part_el[blockIdx.x] = 0.5 * a;
    part_vdw[blockIdx.x] = 0.5 * b;
  }
}

}  // namespace

int pair_forces(msevb_ctx* c, int slot_elec, int slot_vdw) {
  const int N = c->N;
  if (!c->list_valid) FAIL(MSEVB_ERR_STATE, "pair_forces: no Verlet list (call msevb_construct_verlet_list)");
  BoxDev box;
  for (int d = 0; d < 3; ++d) { box.L[d] = c->box[d]; box.invL[d] = 1.0 / c->box[d]; }
  int nblocks;
  phase_begin(c, PH_PAIR);
  if (c->pair_tiles && !(c->shard_now && c->n_ranks > 1)) {
    const int ncell = c->ncell[0] * c->ncell[1] * c->ncell[2];
    if (ncell > MAX_BLOCKS) FAIL(MSEVB_ERR_LIMIT, "pair_forces: too many cells");
    nblocks = ncell;
    const size_t lds = (size_t)c->tile_atoms_max * sizeof(double4) + (size_t)c->nt * c->nt * sizeof(PairParam);
    // the SAPT instantiation needs > 128 VGPRs: 512 threads (8 waves) per cell instead of 1024
    auto kern = c->has_sapt ? k_pair_tiles<true, 512> : k_pair_tiles<false, 1024>;
    static bool attr_set[2] = {false, false};
    if (!attr_set[c->has_sapt ? 1 : 0]) {
      HIPCHK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, PAIR_TILE_LDS_BYTES));
      attr_set[c->has_sapt ? 1 : 0] = true;
    }
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(c->has_sapt ? 512 : 1024), lds, c->stream, make_int3(c->ncell[0], c->ncell[1], c->ncell[2]), c->d_cell_start.p, c->d_posq.p, c->d_type_sorted.p,
                       c->d_nbr_start.p, c->d_nbr_count.p, c->d_nbr.p, c->d_pair_par.p, c->nt, c->d_ewald_tab.p, 1.0 / c->erfc_dx, c->erfc_grid, c->d_tt.p, c->d_dtt.p, box, c->rc * c->rc, c->d_perm.p,
                       c->d_force.p, c->d_partials.p + (size_t)slot_elec * MAX_BLOCKS, c->d_partials.p + (size_t)slot_vdw * MAX_BLOCKS);
  } else {
    int i_lo, i_hi;
    shard_range(c, N, i_lo, i_hi);
    nblocks = round_up8((i_hi - i_lo + 7) / 8);      // 8 atoms per block, two per wave
    if (nblocks > 16384) nblocks = 16384;
    // force fields without a live SAPT term (every Buckingham coefficient zero, as in the CH3SO3H/H2O example) run the lean instantiation
    bool pr = c->prune && c->rc + c->prune_delta < c->rv;      // rolling pruned rows (a delta that reaches the list radius prunes nothing)
    if (pr) {
      if (c->d_nbr2.alloc(c->d_nbr.n) || c->d_nbr2_count.alloc(N) || c->d_posq_prune.alloc(N)) FAIL(MSEVB_ERR_HIP, "hipMalloc failed (pruned pair list)");
      // the pruned rows cover the atom range of the launch that wrote them: another range (sharded EVB call <-> full call) starts over
      if (!c->d_prune_mode.p || c->prune_lo != i_lo || c->prune_hi != i_hi) { CHK(nb_prune_reset(c)); c->prune_lo = i_lo; c->prune_hi = i_hi; }
    }
    auto kern = c->has_sapt ? (pr ? k_pair<true, true> : k_pair<true, false>) : (pr ? k_pair<false, true> : k_pair<false, false>);
    const double rp = c->rc + c->prune_delta;
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(256), c->nt * c->nt * sizeof(PairParam), c->stream, i_lo, i_hi, c->d_posq.p, c->d_type_sorted.p,
                       c->d_nbr_start.p, c->d_nbr_count.p, c->d_nbr.p, c->d_pair_par.p, c->nt, c->d_ewald_tab.p, 1.0 / c->erfc_dx, c->erfc_grid, c->d_tt.p, c->d_dtt.p, box,
                       c->rc * c->rc, c->d_perm.p, c->d_force.p, c->d_partials.p + (size_t)slot_elec * MAX_BLOCKS,
                       c->d_partials.p + (size_t)slot_vdw * MAX_BLOCKS, c->d_prune_mode.p, c->d_nbr2.p, c->d_nbr2_count.p, c->d_posq_prune.p, rp * rp, N);
  }
  HIPCHK(hipGetLastError());
  phase_end(c, PH_PAIR);
  if (c->nbr_stats_copy_pending) {      // row statistics of this call's list rebuild (nb_build_finish), out of the way of the pair loop
    HIPCHK(hipMemcpyAsync(c->h_nbr_stats, c->d_tile_max.p, 8 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    c->nbr_stats_copy_pending = false;
  }
  CHK(reduce_partials(c, slot_elec, nblocks));
  CHK(reduce_partials(c, slot_vdw, nblocks));
  return MSEVB_OK;
}
