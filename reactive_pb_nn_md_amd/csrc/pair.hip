// Real-space pair loop (replaces pairwise_real_space_verlet, src/pair_int_real_space.f90:135, with its three
// interaction kernels: Ewald erfc table :698/:740, LJ :621, SAPT Buckingham + Tang-Toennies :651).
//
// MI355X mapping: ONE WAVEFRONT PER ATOM.  The atom's own data is wave-uniform, its CSR neighbour row is
// streamed with coalesced 256-B reads (64 lanes x int32), neighbour {x,y,z,q} records are 32-B gathers from the
// cell-sorted array (L2-resident neighbourhoods), the (type_i,type_j) parameter table lives in LDS, and the
// per-atom force is reduced across the 64 lanes with shuffles - the list is a full list, so no fp64 atomics.
// Algorithmic bytes per step: 4 B per list entry + 36 B per atom read, 24 B per atom written.
// Two kernels: k_pair (full list, partner records gathered from L2; small boxes) and k_pair_half (half list, LDS tiles; see below).
#include <algorithm>
#include "ctx.hpp"
#include "devmath.hpp"

namespace {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}


// sum over the 16 lanes of each DPP row (lanes 16 r .. 16 r + 15), left in every lane of the row: four cross-lane moves inside the VALU
// (quad swaps, half-row mirror, row mirror), no LDS traffic - __shfl_xor goes through the LDS pipe (ds_bpermute), which the half-list
// pair kernel needs for its tile reads and force adds.
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row_sum16(double v) {
  v += dpp_mov<0xB1>(v);       // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);       // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);      // row_half_mirror: quad q <-> quad 1 - q of each half row
  v += dpp_mov<0x140>(v);      // row_mirror: half row <-> half row
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

// Tang-Toennies rows of pairs on a row boundary (devmath.hpp: tt_arg_exact): the reference's r from its own r^2.  k_pair's r^2 is min_image_r2 already;
// the tile kernels (coordinates relative to the cell centre) go back to the raw positions of the two atoms, as they do for pairs on the cutoff.
__device__ __forceinline__ int tile_to_sorted(int p, const int* s_lo, const int* s_off);
struct TieFixR2 { static constexpr bool on = true; double r2; __device__ __forceinline__ double operator()(double B) const { return tt_arg_exact(r2, B); } };
struct TieFixTile {
  static constexpr bool on = true;
  const double4* __restrict__ posq; int i_sorted, j_tile; const int* s_lo; const int* s_off; const BoxDev* box;
  __device__ __forceinline__ double operator()(double B) const {
    double ux, uy, uz;
    return tt_arg_exact(min_image_r2(posq[i_sorted], posq[tile_to_sorted(j_tile, s_lo, s_off)], *box, ux, uy, uz), B);
  }
};

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
      pair_interaction_rows<SAPT>(r2a, pi.w * pa.w, sp[ti * nt + (int)(e0 >> NBR_IDX_BITS)], wa, tt, dtt, fs, ee, ev, TieFixR2{r2a});
      fx += fs * dxa; fy += fs * dya; fz += fs * dza; e_el += ee; e_vdw += ev;
    }
    if (inb) {
      double fs, ee, ev;
      pair_interaction_rows<SAPT>(r2b, pi.w * pb.w, sp[ti * nt + (int)(e1 >> NBR_IDX_BITS)], wb, tt, dtt, fs, ee, ev, TieFixR2{r2b});
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


// ---------------------------------------------------------------------------------------------------------------------------------
// HALF-LIST LDS PAIR KERNEL: one workgroup per cell of the neighbour grid.
//   * The list is a HALF list (every pair once, Newton's third law): atom i of cell c keeps the partners that come after it in the
//     HALF NEIGHBOURHOOD of c (the cell itself + its 13 forward cells, ctx.hpp), entries = position in that tile (neighbor.hip).
//   * The tile - {x,y,z relative to the cell centre at the periodic image next to it, q} of its ~1 900 atoms, 32 B each - is staged
//     once per workgroup with coalesced loads; the per-pair partner gather is then two ds_read_b128 instead of a divergent 64-B line
//     per lane through L1 (what bounds the global-gather kernel above: 146 M L1 lines per launch at 98 k atoms), no per-pair
//     minimum-image arithmetic is left (the tile IS the minimum image for every pair of the list when there are >= 4 cells per axis),
//     and the Ewald-table rows - the two gathers parity forces on us - are the only per-lane global loads.
//   * -f_ij goes to a force tile in LDS with ds_add_f64 (partners of one row are distinct atoms: no same-address serialisation); the
//     row owner's force is reduced across the wave as before and added to the same tile; the tile is flushed ONCE per workgroup with
//     coalesced fp64 atomics into a force array in SORTED order (Fs, SoA), which k_pair_fold adds to the caller-order force array.
//   * Waves take the cell's atoms from an LDS counter (rows of a half list differ a lot in length).
//   * The cutoff decision is the reference's to the last bit: pairs within 1e-9 relative of r_c^2 are re-decided from the raw coordinates
//     with min_image_r2 (coordinates from a .gro put pairs EXACTLY on the cutoff).
// Half the pair evaluations of the full-list kernel, ~5x fewer L1 line accesses.  Needs >= 4 cells per axis and a half neighbourhood that
// fits the tile (56 B per atom: 2 700 atoms); otherwise the list is built in the full format and k_pair runs.
__device__ __forceinline__ int tile_to_sorted(int p, const int* s_lo, const int* s_off) {
  int s = 0;
  while (s + 1 < HALF_SLOTS && p >= s_off[s + 1]) ++s;
  return s_lo[s] + (p - s_off[s]);
}

template <bool SAPT, bool PRUNE>
__device__ __forceinline__ void pair_row_half(const double4* tile, double* tfx, double* tfy, double* tfz, const double4 pi, int ti, int i_sorted, unsigned s, unsigned e,
                                              int lane, const int* __restrict__ nbr, const PairParam* sp, int nt, const double2* __restrict__ etab, double inv_dx, int emax,
                                              const double* __restrict__ tt, const double* __restrict__ dtt, const BoxDev& box, double rc2, const double4* __restrict__ posq,
                                              const int* s_lo, const int* s_off, double& fx, double& fy, double& fz, double& e_el, double& e_vdw,
                                              int* __restrict__ out, double rp2, unsigned& n_out, unsigned e0, unsigned e1) {
#pragma clang fp contract(fast)
  const double tie = 1e-9 * rc2;
  unsigned k = s + lane;      // e0 / e1 = entries k and k + 64 of the row (NBR_NONE past its end), loaded by the caller while the previous row was evaluated
  for (unsigned k0 = s; k0 < e; k0 += 128) {      // wave-uniform trip count
    const unsigned kn = k + 128;
    const unsigned n0 = kn < e ? (unsigned)nbr[kn] : NBR_NONE, n1 = kn + 64 < e ? (unsigned)nbr[kn + 64] : NBR_NONE;
    const bool va = e0 != NBR_NONE, vb = e1 != NBR_NONE;
    const unsigned ja = va ? (e0 & NBR_IDX_MASK) : 0u, jb = vb ? (e1 & NBR_IDX_MASK) : 0u;
    const double4 pa = tile[ja], pb = tile[jb];
    const double dxa = pi.x - pa.x, dya = pi.y - pa.y, dza = pi.z - pa.z;
    const double dxb = pi.x - pb.x, dyb = pi.y - pb.y, dzb = pi.z - pb.z;
    const double r2a = dxa * dxa + dya * dya + dza * dza, r2b = dxb * dxb + dyb * dyb + dzb * dzb;
    bool ina = va && r2a < rc2, inb = vb && r2b < rc2;
    const bool ta = va && fabs(r2a - rc2) < tie, tb = vb && fabs(r2b - rc2) < tie;
    if (__ballot(ta || tb)) {      // rare: on the cutoff to within rounding - the reference's arithmetic on the raw coordinates decides
      double ux, uy, uz;
      if (ta) ina = min_image_r2(posq[i_sorted], posq[tile_to_sorted((int)ja, s_lo, s_off)], box, ux, uy, uz) < rc2;
      if (tb) inb = min_image_r2(posq[i_sorted], posq[tile_to_sorted((int)jb, s_lo, s_off)], box, ux, uy, uz) < rc2;
    }
    if (PRUNE && out) {      // wave-uniform branch: compact the entries within rc + delta into the pruned row
      const bool ka = va && r2a < rp2, kb = vb && r2b < rp2;
      const unsigned long long ma = __ballot(ka), mb = __ballot(kb);
      unsigned tot = n_out;
      if (ka) out[tot + __builtin_amdgcn_mbcnt_hi((unsigned)(ma >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)ma, 0u))] = (int)e0;
      tot += (unsigned)__popcll(ma);
      if (kb) out[tot + __builtin_amdgcn_mbcnt_hi((unsigned)(mb >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mb, 0u))] = (int)e1;
      n_out = tot + (unsigned)__popcll(mb);
    }
    EwaldRows wa, wb;      // lanes outside the cutoff all fetch the row of r_c: one line for the lot
    ewald_fetch(ina ? r2a : rc2, inv_dx, emax, etab, wa);
    ewald_fetch(inb ? r2b : rc2, inv_dx, emax, etab, wb);
    if (ina) {
      double fs, ee, ev;
      pair_interaction_rows<SAPT>(r2a, pi.w * pa.w, sp[ti * nt + (int)(e0 >> NBR_IDX_BITS)], wa, tt, dtt, fs, ee, ev, TieFixTile{posq, i_sorted, (int)ja, s_lo, s_off, &box});
      const double gx = fs * dxa, gy = fs * dya, gz = fs * dza;
      fx += gx; fy += gy; fz += gz; e_el += ee; e_vdw += ev;
      atomicAdd(&tfx[ja], -gx); atomicAdd(&tfy[ja], -gy); atomicAdd(&tfz[ja], -gz);
    }
    if (inb) {
      double fs, ee, ev;
      pair_interaction_rows<SAPT>(r2b, pi.w * pb.w, sp[ti * nt + (int)(e1 >> NBR_IDX_BITS)], wb, tt, dtt, fs, ee, ev, TieFixTile{posq, i_sorted, (int)jb, s_lo, s_off, &box});
      const double gx = fs * dxb, gy = fs * dyb, gz = fs * dzb;
      fx += gx; fy += gy; fz += gz; e_el += ee; e_vdw += ev;
      atomicAdd(&tfx[jb], -gx); atomicAdd(&tfy[jb], -gy); atomicAdd(&tfz[jb], -gz);
    }
    e0 = n0; e1 = n1; k = kn;
  }
}

// ---- software-pipelined row loop of k_pair_half (launches that run on fixed rows: the pruned rows, or the full rows without pruning) ----
// A wave works through a STREAM of trips (64 list entries each) that crosses row boundaries; three trips are in flight:
//     trip t + 2: its 64 entries are being loaded (global),
//     trip t + 1: stage A done - partner positions read from the tile, r^2, cutoff decision, the two Ewald-table rows REQUESTED,
//     trip t    : stage B - interaction arithmetic with table rows that were requested one trip ago, LDS force adds.
// The wave-per-row loop above waits once per trip for the entries of a new row and once for the table rows; with four waves per SIMD
// (the tile pins one workgroup to a CU) the SIMDs idled 55 % of the time (SQ_WAIT_ANY).  Here the table rows of trip t + 1 and the entries
// of trip t + 2 are under way while trip t computes.
struct TripCtx { int a; int last; };      // wave-uniform: row (atom of the cell) of the trip, 1 = last trip of that row
struct TripState { double dx, dy, dz, r2, qq; EwaldRows w; unsigned e; bool in; };

// Stage A and the arithmetic of stage B are BRANCH-FREE (one basic block): with four waves per SIMD the kernel is bound by the issue
// latency of each wave's dependent fp64 chain, and the compiler interleaves the two independent chains (trip t + 1's geometry, trip t's
// interaction) only inside a block.  Lanes without a pair, or outside the cutoff, compute on r = r_c and their results are discarded;
// the exact cutoff decision for a pair within rounding of r_c (min_image_r2, rare) is made after the arithmetic.
template <bool SAPT>
__device__ __forceinline__ void trip_stage_a(const double4* tile, unsigned e, int a, double rc2, double inv_dx, int emax, const double2* __restrict__ etab, TripState& st) {
#pragma clang fp contract(fast)
  const bool v = e != NBR_NONE;
  const unsigned j = v ? (e & NBR_IDX_MASK) : 0u;
  const double4 pi = tile[a], pj = tile[j];
  st.dx = pi.x - pj.x; st.dy = pi.y - pj.y; st.dz = pi.z - pj.z;
  st.r2 = st.dx * st.dx + st.dy * st.dy + st.dz * st.dz;
  st.qq = pi.w * pj.w;
  st.e = e;
  st.in = v;
  ewald_fetch(v ? fmin(st.r2, rc2) : rc2, inv_dx, emax, etab, st.w);      // lanes outside the cutoff all fetch the row of r_c: one line for the lot
}

template <bool SAPT>
__device__ __forceinline__ void trip_stage_b(const TripState& st, int a, int cs, int ti, const PairParam* sp, int nt, const double* __restrict__ tt, const double* __restrict__ dtt,
                                             const double4* __restrict__ posq, const int* s_lo, const int* s_off, const BoxDev& box, double rc2,
                                             double* tfx, double* tfy, double* tfz, double& fx, double& fy, double& fz, double& e_el, double& e_vdw) {
#pragma clang fp contract(fast)
  const unsigned j = st.e & NBR_IDX_MASK;
  const PairParam& pp = sp[ti * nt + (st.in ? (int)(st.e >> NBR_IDX_BITS) : 0)];
  double fs, ee, ev;
  if (SAPT) pair_interaction_rows<true>(fmin(st.r2, rc2), st.qq, pp, st.w, tt, dtt, fs, ee, ev, TieFixTile{posq, cs + a, (int)j, s_lo, s_off, &box});
  else {      // Ewald + LJ without a branch (pair_interaction_rows' arithmetic; a pair type without LJ gets the factor 0)
    const double inv_r = st.w.inv_r, inv_r2 = inv_r * inv_r;
    const double c2 = (st.w.x1 + 1.0) - (double)st.w.idx, c1 = 1.0 - c2;
    const double evv = c1 * st.w.t0.x + c2 * st.w.t1.x, sv = c1 * st.w.t0.y + c2 * st.w.t1.y;
    const double qr = st.qq * inv_r;
    ee = qr * evv;
    fs = qr * inv_r2 * sv;
    const double lj = pp.type == 0 ? 1.0 : 0.0;
    const double r6i = inv_r2 * inv_r2 * inv_r2, r12i = r6i * r6i;
    ev = lj * (pp.p[0] * r12i - pp.p[1] * r6i);
    fs += lj * inv_r2 * (12.0 * pp.p[0] * r12i - 6.0 * pp.p[1] * r6i);
  }
  bool in = st.in && st.r2 < rc2;
  const bool tie = st.in && fabs(st.r2 - rc2) < 1e-9 * rc2;
  if (__ballot(tie)) {      // rare: on the cutoff to within rounding - the reference's arithmetic on the raw coordinates decides
    double ux, uy, uz;
    if (tie) in = min_image_r2(posq[cs + a], posq[tile_to_sorted((int)j, s_lo, s_off)], box, ux, uy, uz) < rc2;
  }
  if (in) {
    const double gx = fs * st.dx, gy = fs * st.dy, gz = fs * st.dz;
    fx += gx; fy += gy; fz += gz; e_el += ee; e_vdw += ev;
    atomicAdd(&tfx[j], -gx); atomicAdd(&tfy[j], -gy); atomicAdd(&tfz[j], -gz);
  }
}

// the stream of trips of one wave: rows drawn from the workgroup's LDS counter, 64 entries per trip
struct TripCursor {
  int a, na; unsigned s, n, k;      // current row (>= na: no more rows), its first entry, length, next entry offset
};
__device__ __forceinline__ void cursor_draw(TripCursor& cu, int* s_next, const unsigned* s_cnt, const unsigned* __restrict__ nbr_start, int cs, int lane) {
  for (;;) {      // rows without entries are skipped
    int a = 0;
    if (lane == 0) a = atomicAdd(s_next, 1);
    cu.a = __builtin_amdgcn_readfirstlane(a);
    if (cu.a >= cu.na) return;
    cu.n = s_cnt[cu.a];
    if (cu.n == 0) continue;
    cu.s = nbr_start[cs + cu.a]; cu.k = 0;
    return;
  }
}
// next trip of the stream: its context and its entries (a load); ctx.a = -1 when the stream has ended
__device__ __forceinline__ unsigned cursor_next(TripCursor& cu, TripCtx& ctx, const int* __restrict__ rows, int* s_next, const unsigned* s_cnt,
                                                const unsigned* __restrict__ nbr_start, int cs, int lane) {
  if (cu.a >= cu.na) { ctx.a = -1; ctx.last = 0; return NBR_NONE; }
  const unsigned kk = cu.k + (unsigned)lane;
  const unsigned e = kk < cu.n ? (unsigned)rows[cu.s + kk] : NBR_NONE;
  ctx.a = cu.a;
  cu.k += 64;
  ctx.last = cu.k >= cu.n ? 1 : 0;
  if (ctx.last) cursor_draw(cu, s_next, s_cnt, nbr_start, cs, lane);
  return e;
}

template <bool SAPT>
__device__ __forceinline__ void pair_rows_pipelined(const double4* tile, double* tfx, double* tfy, double* tfz, int na, int cs, int lane, int* s_next, const unsigned* s_cnt,
                                                    const unsigned char* s_typ, const unsigned* __restrict__ nbr_start, const int* __restrict__ rows, const PairParam* sp, int nt,
                                                    const double2* __restrict__ etab, double inv_dx, int emax, const double* __restrict__ tt, const double* __restrict__ dtt,
                                                    const BoxDev& box, double rc2, const double4* __restrict__ posq, const int* s_lo, const int* s_off, double& e_el, double& e_vdw) {
  TripCursor cu; cu.na = na; cu.a = 0; cu.s = 0; cu.n = 0; cu.k = 0;
  cursor_draw(cu, s_next, s_cnt, nbr_start, cs, lane);
  TripCtx c0, c1, c2;
  unsigned e1, e2;
  TripState s0, s1;
  const unsigned e0 = cursor_next(cu, c0, rows, s_next, s_cnt, nbr_start, cs, lane);
  e1 = cursor_next(cu, c1, rows, s_next, s_cnt, nbr_start, cs, lane);
  if (c0.a < 0) return;
  trip_stage_a<SAPT>(tile, e0, c0.a, rc2, inv_dx, emax, etab, s0);
  double fx = 0.0, fy = 0.0, fz = 0.0;
  while (c0.a >= 0) {
    e2 = cursor_next(cu, c2, rows, s_next, s_cnt, nbr_start, cs, lane);                                     // entries of trip t + 2: load issued
    trip_stage_a<SAPT>(tile, c1.a >= 0 ? e1 : NBR_NONE, max(c1.a, 0), rc2, inv_dx, emax, etab, s1);      // trip t + 1: table rows requested (a stream that has ended runs an empty trip)
    trip_stage_b<SAPT>(s0, c0.a, cs, s_typ[c0.a], sp, nt, tt, dtt, posq, s_lo, s_off, box, rc2, tfx, tfy, tfz, fx, fy, fz, e_el, e_vdw);      // trip t: arithmetic
    if (c0.last) {      // (wave-uniform) the row is complete: its owner's force, reduced over the 16-lane rows, joins the tile
      fx = row_sum16(fx); fy = row_sum16(fy); fz = row_sum16(fz);
      if ((lane & 15) == 0) { atomicAdd(&tfx[c0.a], fx); atomicAdd(&tfy[c0.a], fy); atomicAdd(&tfz[c0.a], fz); }
      fx = 0.0; fy = 0.0; fz = 0.0;
    }
    c0 = c1; s0 = s1; c1 = c2; e1 = e2;
  }
}

template <bool SAPT, bool PRUNE, int THREADS>
__global__ __launch_bounds__(THREADS) void k_pair_half(int c_lo, int c_hi, int3 nc, int tile_cap, const int* __restrict__ cell_start, const double4* __restrict__ posq,
                                                    const int* __restrict__ type_s, const unsigned* __restrict__ nbr_start, const unsigned* __restrict__ nbr_count,
                                                    const int* __restrict__ nbr, const PairParam* __restrict__ pairpar, int nt, const double2* __restrict__ etab, double inv_dx,
                                                    int emax, const double* __restrict__ tt, const double* __restrict__ dtt, BoxDev box, double rc2, double* __restrict__ Fs,
                                                    double* __restrict__ part_el, double* __restrict__ part_vdw, int* __restrict__ prune_state, int* __restrict__ nbr2,
                                                    unsigned* __restrict__ count2, double4* __restrict__ posq_prune, double rp2, int N_all) {
  extern __shared__ double4 tile[];                     // [tile_cap] positions, then the force tile (3 x [tile_cap]), then the pair-parameter table
  __shared__ int s_lo[HALF_SLOTS], s_off[HALF_SLOTS + 1], s_next;
  __shared__ double red[2][16];
  __shared__ unsigned s_cnt[PAIR_CELL_MAX];      // row lengths of the cell's atoms (the rows this launch runs on)
  __shared__ unsigned char s_typ[PAIR_CELL_MAX];
  double* tfx = (double*)(tile + tile_cap);
  double* tfy = tfx + tile_cap;
  double* tfz = tfy + tile_cap;
  PairParam* sp = (PairParam*)(tfz + tile_cap);
  const int mode = PRUNE ? __builtin_amdgcn_readfirstlane(prune_state[0]) : 0;
  if (PRUNE && mode != 2) {      // snapshot of ALL positions (the displacement check runs over every atom, also when the cells are sharded over ranks)
    const int chunk = (N_all + gridDim.x - 1) / gridDim.x, k0 = blockIdx.x * chunk, k1 = min(N_all, k0 + chunk);
    for (int k = k0 + threadIdx.x; k < k1; k += blockDim.x) posq_prune[k] = posq[k];
    if (blockIdx.x == 0 && threadIdx.x == 0) prune_state[2] = 1;      // seen by the NEXT step's k_top2_final: pruned rows exist for these full rows
  }
  const int cidx = c_lo + xcd_block(blockIdx.x, gridDim.x);      // one XCD works on one contiguous eighth of the cells (a slab): its L2 holds that slab's rows
  if (cidx >= c_hi) { if (threadIdx.x == 0) { part_el[blockIdx.x] = 0.0; part_vdw[blockIdx.x] = 0.0; } return; }
  if (threadIdx.x < HALF_SLOTS + 1) {      // 14 lanes look up their cell, lane 14 waits for the prefix sum below
    int slots[HALF_SLOTS];
    cell_half_neighbourhood(cidx, nc.x, nc.y, nc.z, slots);
    if (threadIdx.x < HALF_SLOTS) { s_lo[threadIdx.x] = cell_start[slots[threadIdx.x]]; s_off[threadIdx.x + 1] = cell_start[slots[threadIdx.x] + 1] - cell_start[slots[threadIdx.x]]; }
  }
  __syncthreads();
  if (threadIdx.x == 0) { int off = 0; for (int s = 0; s < HALF_SLOTS; ++s) { const int n = s_off[s + 1]; s_off[s] = off; off += n; } s_off[HALF_SLOTS] = off; s_next = 0; }
  __syncthreads();
  const int ntile = s_off[HALF_SLOTS], na = s_off[1];
  if (ntile > tile_cap || na > PAIR_CELL_MAX) {      // a half neighbourhood larger than the tile this list was built for: the host sees it in the build statistics and repeats the call
    if (threadIdx.x == 0) { part_el[blockIdx.x] = 0.0; part_vdw[blockIdx.x] = 0.0; }
    return;
  }
  for (int k = threadIdx.x; k < nt * nt; k += blockDim.x) sp[k] = pairpar[k];
  const int cs = s_lo[0];
  {      // stage, ONE pass over the tile (every thread has its loads in flight together): coordinates relative to the cell centre, at the
         // periodic image nearest to it (>= 4 cells per axis: every tile atom is within 2 cells of it)
    const int cx = cidx % nc.x, cy = (cidx / nc.x) % nc.y, cz = cidx / (nc.x * nc.y);
    const double ox = (cx + 0.5) * box.L[0] / nc.x, oy = (cy + 0.5) * box.L[1] / nc.y, oz = (cz + 0.5) * box.L[2] / nc.z;
    for (int p = threadIdx.x; p < ntile; p += blockDim.x) {
      const double4 q = posq[tile_to_sorted(p, s_lo, s_off)];
      double4 t;
      t.x = q.x - ox; t.x = t.x - box.L[0] * floor(t.x * box.invL[0] + 0.5);
      t.y = q.y - oy; t.y = t.y - box.L[1] * floor(t.y * box.invL[1] + 0.5);
      t.z = q.z - oz; t.z = t.z - box.L[2] * floor(t.z * box.invL[2] + 0.5);
      t.w = q.w;
      tile[p] = t;
      tfx[p] = 0.0; tfy[p] = 0.0; tfz[p] = 0.0;
    }
    const unsigned* cnt = (PRUNE && mode == 2) ? count2 : nbr_count;
    for (int a = threadIdx.x; a < na; a += blockDim.x) { s_cnt[a] = cnt[cs + a]; s_typ[a] = (unsigned char)type_s[cs + a]; }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), wpb = blockDim.x >> 6;
  const int* rows = (PRUNE && mode == 2) ? nbr2 : nbr;
  double e_el = 0.0, e_vdw = 0.0;
  if (!(PRUNE && mode != 2)) {
    // fixed rows (pruned rows in use, or no pruning): the software-pipelined stream of 64-entry trips
    pair_rows_pipelined<SAPT>(tile, tfx, tfy, tfz, na, cs, lane, &s_next, s_cnt, s_typ, nbr_start, rows, sp, nt, etab, inv_dx, emax, tt, dtt, box, rc2, posq, s_lo, s_off, e_el, e_vdw);
  } else {
  // A launch that REWRITES the pruned rows (every few steps): wave per row, 128 entries per trip.
  // Rows are handed out by an LDS counter (half-list rows differ a lot in length); a wave draws its NEXT row and loads that row's first 128
  // entries before it evaluates the current one, so the list latency of a row hides behind the arithmetic of its predecessor.
  int a = 0;
  if (lane == 0) a = atomicAdd(&s_next, 1);
  a = __builtin_amdgcn_readfirstlane(a);
  unsigned e0 = NBR_NONE, e1 = NBR_NONE;
  if (a < na) {
    const unsigned s = nbr_start[cs + a], n = s_cnt[a];
    e0 = (unsigned)lane < n ? (unsigned)rows[s + lane] : NBR_NONE; e1 = (unsigned)lane + 64 < n ? (unsigned)rows[s + lane + 64] : NBR_NONE;
  }
  while (a < na) {
    int an = 0;
    if (lane == 0) an = atomicAdd(&s_next, 1);
    an = __builtin_amdgcn_readfirstlane(an);
    unsigned f0 = NBR_NONE, f1 = NBR_NONE;
    if (an < na) {
      const unsigned s = nbr_start[cs + an], n = s_cnt[an];
      f0 = (unsigned)lane < n ? (unsigned)rows[s + lane] : NBR_NONE; f1 = (unsigned)lane + 64 < n ? (unsigned)rows[s + lane + 64] : NBR_NONE;
    }
    const int i = cs + a;
    const double4 pi = tile[a];
    const int ti = s_typ[a];
    const unsigned s = nbr_start[i];
    double fx = 0.0, fy = 0.0, fz = 0.0;
    unsigned n2 = 0;
    pair_row_half<SAPT, true>(tile, tfx, tfy, tfz, pi, ti, i, s, s + s_cnt[a], lane, rows, sp, nt, etab, inv_dx, emax, tt, dtt, box, rc2, posq, s_lo, s_off, fx, fy, fz, e_el, e_vdw, nbr2 + s, rp2, n2, e0, e1);
    if (lane == 0) count2[i] = n2;
    fx = row_sum16(fx); fy = row_sum16(fy); fz = row_sum16(fz);      // the four row sums meet in the LDS add
    if ((lane & 15) == 0) { atomicAdd(&tfx[a], fx); atomicAdd(&tfy[a], fy); atomicAdd(&tfz[a], fz); }
    a = an; e0 = f0; e1 = f1;
  }
  }
  e_el = wave_sum(e_el);
  e_vdw = wave_sum(e_vdw);
  if (lane == 0) { red[0][wave] = e_el; red[1][wave] = e_vdw; }
  __syncthreads();      // ... and every wave's LDS force adds are done
  if (threadIdx.x == 0) {
    double a_ = 0, b_ = 0;
    for (int w = 0; w < wpb; ++w) { a_ += red[0][w]; b_ += red[1][w]; }
    part_el[blockIdx.x] = a_;      // half list: every pair once
    part_vdw[blockIdx.x] = b_;
  }
  // flush the force tile: runs of consecutive sorted atoms per cell, one coalesced fp64 atomic per component (partners no row reached stay out)
  for (int p = threadIdx.x; p < ntile; p += blockDim.x) {
    const double vx = tfx[p], vy = tfy[p], vz = tfz[p];
    if (vx != 0.0 || vy != 0.0 || vz != 0.0) {
      const int j = tile_to_sorted(p, s_lo, s_off);
      atomicAdd(&Fs[j], vx); atomicAdd(&Fs[(size_t)N_all + j], vy); atomicAdd(&Fs[2 * (size_t)N_all + j], vz);
    }
  }
}

// caller-order force += sorted-order accumulator of k_pair_half (atomic: the per-molecule kernel and the reciprocal gather may be running), accumulator re-zeroed
__global__ __launch_bounds__(256) void k_pair_fold(int N, const int* __restrict__ perm, double* __restrict__ Fs, double* __restrict__ F) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= N) return;
  const double vx = Fs[k], vy = Fs[(size_t)N + k], vz = Fs[2 * (size_t)N + k];
  Fs[k] = 0.0; Fs[(size_t)N + k] = 0.0; Fs[2 * (size_t)N + k] = 0.0;
  const int o = perm[k];
  atomicAdd(&F[3 * o], vx); atomicAdd(&F[3 * o + 1], vy); atomicAdd(&F[3 * o + 2], vz);
}

}  // namespace

int pair_forces(msevb_ctx* c, int slot_elec, int slot_vdw) {
  const int N = c->N;
  if (!c->list_valid) FAIL(MSEVB_ERR_STATE, "pair_forces: no Verlet list (call msevb_construct_verlet_list)");
  BoxDev box;
  for (int d = 0; d < 3; ++d) { box.L[d] = c->box[d]; box.invL[d] = 1.0 / c->box[d]; }
  int nblocks;
  phase_begin(c, PH_PAIR);
  if (c->pair_tiles) {      // half list, tile-local entries: the LDS kernel, one workgroup per cell
    const int ncell = c->ncell[0] * c->ncell[1] * c->ncell[2];
    int c_lo = 0, c_hi = ncell;
    if (c->shard_now && c->n_ranks > 1) { c_lo = (int)((long)ncell * c->rank / c->n_ranks); c_hi = (int)((long)ncell * (c->rank + 1) / c->n_ranks); }      // this rank's slab of cells
    nblocks = round_up8(std::max(c_hi - c_lo, 1));
    if (nblocks > MAX_BLOCKS) FAIL(MSEVB_ERR_LIMIT, "pair_forces: too many cells");
    const size_t lds = (size_t)c->half_tile_cap * PAIR_TILE_ATOM_BYTES + (size_t)c->nt * c->nt * sizeof(PairParam);
    const bool pr = c->prune && c->rc + c->prune_delta < c->rv;
    if (pr) {
      const bool flat = pair_flat_usable(c);      // (k_pair_flat keeps its list in d_flat)
      if ((!flat && (c->d_nbr2.alloc(c->d_nbr.n) || c->d_nbr2_count.alloc(N))) || c->d_posq_prune.alloc(N)) FAIL(MSEVB_ERR_HIP, "hipMalloc failed (pruned pair list)");
      if (!c->d_prune_mode.p || c->prune_lo != -2 - c_lo || c->prune_hi != c_hi) { CHK(nb_prune_reset(c)); c->prune_lo = -2 - c_lo; c->prune_hi = c_hi; }      // (cell ranges are kept apart from the atom ranges of k_pair)
    }
    if (!c->d_fsorted.p) {
      if (c->d_fsorted.alloc(3 * (size_t)N)) FAIL(MSEVB_ERR_HIP, "hipMalloc failed (sorted force accumulator)");
      HIPCHK(hipMemsetAsync(c->d_fsorted.p, 0, 3 * (size_t)N * 8, c->stream));
      c->fsorted_pending = false;
    }
    if (c->fsorted_pending) {      // a force call whose blend never ran (virtual-rank tests, a failed call) left its pair forces in the accumulator: drop them
      HIPCHK(hipMemsetAsync(c->d_fsorted.p, 0, 3 * (size_t)N * 8, c->stream));
      c->fsorted_pending = false;
    }
    if (pr && pair_flat_usable(c)) {      // flat per-cell list (pair_flat.hip)
      CHK(pair_flat_launch(c, c_lo, c_hi, slot_elec, slot_vdw, &nblocks));
    } else {
    // the SAPT instantiation needs > 128 VGPRs: 512 threads (8 waves) per cell instead of 1024
    void (*kern)(int, int, int3, int, const int*, const double4*, const int*, const unsigned*, const unsigned*, const int*, const PairParam*, int, const double2*, double, int,
                 const double*, const double*, BoxDev, double, double*, double*, double*, int*, int*, unsigned*, double4*, double, int) =
        c->has_sapt ? (pr ? k_pair_half<true, true, 512> : k_pair_half<true, false, 512>)
                    : (c->pair_threads == 512 ? (pr ? k_pair_half<false, true, 512> : k_pair_half<false, false, 512>) : (pr ? k_pair_half<false, true, 1024> : k_pair_half<false, false, 1024>));
    const int nthr = (c->has_sapt || c->pair_threads == 512) ? 512 : 1024;
    const int which = (c->has_sapt ? 2 : 0) + (pr ? 1 : 0) + (nthr == 512 && !c->has_sapt ? 4 : 0);
    if (!c->attr_pair_half[which]) {
      HIPCHK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, PAIR_TILE_LDS_BYTES));
      c->attr_pair_half[which] = true;
    }
    const double rp = c->rc + c->prune_delta;
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(nthr), lds, c->stream, c_lo, c_hi, make_int3(c->ncell[0], c->ncell[1], c->ncell[2]), c->half_tile_cap,
                       c->d_cell_start.p, c->d_posq.p, c->d_type_sorted.p, c->d_nbr_start.p, c->d_nbr_count.p, c->d_nbr.p, c->d_pair_par.p, c->nt, c->d_ewald_tab.p, 1.0 / c->erfc_dx,
                       c->erfc_grid, c->d_tt.p, c->d_dtt.p, box, c->rc * c->rc, c->d_fsorted.p, c->d_partials.p + (size_t)slot_elec * MAX_BLOCKS,
                       c->d_partials.p + (size_t)slot_vdw * MAX_BLOCKS, c->d_prune_mode.p, c->d_nbr2.p, c->d_nbr2_count.p, c->d_posq_prune.p, rp * rp, N);
    }
    if (c->defer_fold_now) c->fsorted_pending = true;      // MS-EVB fast path: k_evb_blend adds the accumulator to the force array (and re-zeroes it)
    else hipLaunchKernelGGL(k_pair_fold, dim3((N + 255) / 256), dim3(256), 0, c->stream, N, c->d_perm.p, c->d_fsorted.p, c->d_force.p);
  } else {
    int i_lo, i_hi;
    shard_range(c, N, i_lo, i_hi);
    nblocks = round_up8(std::max((i_hi - i_lo + 7) / 8, 1));      // 8 atoms per block, two per wave (an empty share still launches: its blocks write zero partial sums)
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
