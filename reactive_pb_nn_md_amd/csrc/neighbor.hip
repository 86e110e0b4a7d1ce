// Device neighbour structure (replaces construct_verlet_list_grid, src/general_routines.f90:1408, and
// update_verlet_displacements, :1259).
//
// Layout in HBM: atoms are sorted by cell (cell edge >= verlet_cutoff) so that neighbours of consecutive atoms
// are consecutive in memory; the list is a FULL list in sorted numbering (every pair stored for both atoms, so the
// pair kernel owns each atom's force) with fixed-stride rows: row i = nbr[i * cap .. i * cap + count[i]), which lets a
// rebuild run in one pass (no count + scan + fill).  As a set of unordered pairs {different molecule,
// r < verlet_cutoff} it is identical to the reference's half list.
#include <hipcub/hipcub.hpp>

#include "ctx.hpp"

namespace {

__global__ void k_cell_assign(int N, const double* __restrict__ xyz, BoxDev box, int3 nc, int* __restrict__ keys, int* __restrict__ vals) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  int cc[3];
  const int ncv[3] = {nc.x, nc.y, nc.z};
  for (int d = 0; d < 3; ++d) {
    double f = xyz[3 * i + d] * box.invL[d];
    f = f - floor(f);
    int v = (int)(f * ncv[d]);
    if (v >= ncv[d]) v = ncv[d] - 1;
    if (v < 0) v = 0;
    cc[d] = v;
  }
  keys[i] = cc[0] + nc.x * (cc[1] + nc.y * cc[2]);
  vals[i] = i;
}

// ---- stable counting sort of the atoms by cell (what hipcub's radix sort delivers - ascending cell, ties in atom order - in four kernels instead
// of ten: the radix passes are 5 us each with 10 us between them, 125 us of every list rebuild)
constexpr int CS_APB = 256;      // atoms per block of the histogram / scatter kernels
// cell of every atom (k_cell_assign's arithmetic) + per-block histogram row hist[b][cell]
__global__ __launch_bounds__(256) void k_cs_hist(int N, const double* __restrict__ xyz, BoxDev box, int3 nc, int ncell, int* __restrict__ keys, int* __restrict__ hist) {
  extern __shared__ int s_h[];
  for (int k = threadIdx.x; k < ncell; k += blockDim.x) s_h[k] = 0;
  __syncthreads();
  const int i = blockIdx.x * CS_APB + threadIdx.x;
  if (i < N) {
    int cc[3];
    const int ncv[3] = {nc.x, nc.y, nc.z};
    for (int d = 0; d < 3; ++d) {
      double f = xyz[3 * i + d] * box.invL[d];
      f = f - floor(f);
      int v = (int)(f * ncv[d]);
      if (v >= ncv[d]) v = ncv[d] - 1;
      if (v < 0) v = 0;
      cc[d] = v;
    }
    const int key = cc[0] + nc.x * (cc[1] + nc.y * cc[2]);
    keys[i] = key;
    atomicAdd(&s_h[key], 1);
  }
  __syncthreads();
  for (int k = threadIdx.x; k < ncell; k += blockDim.x) hist[(size_t)blockIdx.x * ncell + k] = s_h[k];
}
// hist[b][cell] -> number of the cell's atoms in the blocks before b: one wavefront per cell, lane l takes a contiguous run of blocks, the
// runs' totals meet in a wave scan (a thread per cell walking all blocks was 97 us of dependent loads)
__global__ __launch_bounds__(64) void k_cs_scan(int nblk, int ncell, int* __restrict__ hist, int* __restrict__ cell_total) {
  const int k = blockIdx.x, lane = threadIdx.x;
  const int per = (nblk + 63) / 64, b0 = lane * per, b1 = min(b0 + per, nblk);
  int v[16];      // per <= 16: N <= 262144 atoms in blocks of 256
  int sum = 0;
#pragma unroll
  for (int q = 0; q < 16; ++q) { v[q] = (q < per && b0 + q < b1) ? hist[(size_t)(b0 + q) * ncell + k] : 0; sum += v[q]; }
  int incl = sum;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
  int run = incl - sum;
#pragma unroll
  for (int q = 0; q < 16; ++q) if (q < per && b0 + q < b1) { hist[(size_t)(b0 + q) * ncell + k] = run; run += v[q]; }
  if (lane == 63) cell_total[k] = incl;
}
// cell_start[] = exclusive prefix of the cells' totals (<= 4096 cells, one workgroup)
__global__ __launch_bounds__(1024) void k_cs_cells(int ncell, int N, const int* __restrict__ cell_total, int* __restrict__ cell_start) {
  __shared__ int s_w[16];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  int v[4], sum = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) { const int k = 4 * t + q; v[q] = k < ncell ? cell_total[k] : 0; sum += v[q]; }
  int incl = sum;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const int x = __shfl_up(incl, o, 64); if (lane >= o) incl += x; }
  if (lane == 63) s_w[wave] = incl;
  __syncthreads();
  int base = 0;
  for (int w = 0; w < wave; ++w) base += s_w[w];
  int run = base + incl - sum;
#pragma unroll
  for (int q = 0; q < 4; ++q) { const int k = 4 * t + q; if (k < ncell) cell_start[k] = run; run += v[q]; }
  if (t == 0) cell_start[ncell] = N;
}
// one wavefront per block of CS_APB atoms, chunks of 64 in atom order: position = cell_start + atoms of the cell in earlier blocks + earlier
// atoms of the cell in this block (lanes of a chunk rank themselves with 64 shuffles, one lane per cell advances the cell's counter)
__global__ __launch_bounds__(64) void k_cs_scatter(int N, int ncell, const int* __restrict__ keys, const int* __restrict__ hist, const int* __restrict__ cell_start,
                                                   int* __restrict__ perm, int* __restrict__ cell_of) {
  extern __shared__ int s_c[];
  const int lane = threadIdx.x;
  for (int k = lane; k < ncell; k += 64) s_c[k] = cell_start[k] + hist[(size_t)blockIdx.x * ncell + k];
  __syncthreads();
  for (int c0 = 0; c0 < CS_APB; c0 += 64) {
    const int i = blockIdx.x * CS_APB + c0 + lane;
    const int key = i < N ? keys[i] : -1;
    int rank = 0, cnt = 0;
    for (int l = 0; l < 64; ++l) { const int kl = __shfl(key, l, 64); if (kl == key) { ++cnt; if (l < lane) ++rank; } }
    int base = 0;
    if (key >= 0) base = s_c[key];      // (every lane of the cell's group reads the counter before its first lane advances it: one wavefront, in-order LDS)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
    if (key >= 0 && rank == 0) s_c[key] = base + cnt;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (key >= 0) { perm[base + rank] = i; cell_of[base + rank] = key; }
  }
}

__global__ void k_cell_bounds(int N, int ncell, const int* __restrict__ keys, int* __restrict__ cell_start) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= N) return;
  int key = keys[k];
  int prev = k == 0 ? -1 : keys[k - 1];
  for (int cidx = prev + 1; cidx <= key; ++cidx) cell_start[cidx] = k;
  if (k == N - 1) for (int cidx = key + 1; cidx <= ncell; ++cidx) cell_start[cidx] = N;
}

__global__ void k_gather_sorted(int N, const int* __restrict__ perm, const double* __restrict__ xyz, const double* __restrict__ chg,
                                const int* __restrict__ atype, const int* __restrict__ mol_of, double4* __restrict__ posq,
                                int* __restrict__ type_s, int* __restrict__ mol_s, int* __restrict__ inv_perm,
                                double* __restrict__ F_zero = nullptr, int* __restrict__ prune_mode = nullptr, int* __restrict__ stats_zero = nullptr) {
  // the small resets of a list rebuild ride along (each was a fill COMMAND of its own on the stream, 7-17 us apiece in front of the row kernel):
  // force accumulator zeroed (rebuild inside a force call), pair loop told to prune (nb_prune_reset), row statistics zeroed
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k == 0 && prune_mode) { prune_mode[0] = 1; prune_mode[2] = 0; }
  if (k < 8 && stats_zero) stats_zero[k] = 0;
  if (k >= N) return;
  int i = perm[k];
  posq[k] = make_double4(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], chg[i]);
  type_s[k] = atype[i];
  mol_s[k] = mol_of[i];
  if (inv_perm) inv_perm[i] = k;
  if (F_zero) { F_zero[3 * i] = 0.0; F_zero[3 * i + 1] = 0.0; F_zero[3 * i + 2] = 0.0; }
}

// out[0]: largest number of atoms in the neighbourhood (distinct neighbour cells) of any cell = LDS tile of the full-list build;
// out[1]: the same for the HALF neighbourhood (cell + 13 forward cells) = LDS tile of the half-list build and of the half-list pair kernel
__global__ void k_tile_max(int ncell, int3 nc, const int* __restrict__ cell_start, int* __restrict__ out) {
  const int cidx = blockIdx.x * blockDim.x + threadIdx.x;
  if (cidx >= ncell) return;
  int slots[27];
  const int ns = cell_neighbourhood(cidx, nc.x, nc.y, nc.z, slots);
  int tot = 0;
  for (int s = 0; s < ns; ++s) tot += cell_start[slots[s] + 1] - cell_start[slots[s]];
  atomicMax(out, tot);
  atomicMax(out + 6, cell_start[cidx + 1] - cell_start[cidx]);      // most atoms in one cell (the half-list pair kernels keep row metadata for PAIR_CELL_MAX)
  if (nc.x >= 3 && nc.y >= 3 && nc.z >= 3) {
    cell_half_neighbourhood(cidx, nc.x, nc.y, nc.z, slots);
    tot = 0;
    for (int s = 0; s < HALF_SLOTS; ++s) tot += cell_start[slots[s] + 1] - cell_start[slots[s]];
    atomicMax(out + 1, tot);
    int lo7 = 0;
    for (int s = 0; s < 7; ++s) lo7 += cell_start[slots[s] + 1] - cell_start[slots[s]];
    atomicMax(out + 7, max(lo7, cell_start[cidx + 1] - cell_start[cidx] + tot - lo7));      // largest sub-tile of the split pair kernel
  }
}

// One wavefront per sorted atom.  Rows have a fixed stride `cap` (row i starts at i * cap), so a rebuild is ONE pass: hits are
// written compactly (ballot + mbcnt) while they fit, the true count goes to count[i], and the host re-runs with a larger stride only
// when max(count) > cap.  FILL = false only counts (first build: sizes cap).
template <bool FILL>
__global__ __launch_bounds__(256) void k_neighbors(int N, const double4* __restrict__ posq, const int* __restrict__ mol_s, const int* __restrict__ type_s,
                                                   const int* __restrict__ keys, const int* __restrict__ cell_start, BoxDev box, int3 nc,
                                                   double rv2, unsigned cap, unsigned* __restrict__ count, unsigned* __restrict__ start,
                                                   int* __restrict__ nbr) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (i >= N) return;
  const double4 pi = posq[i];
  const int mi = mol_s[i];
  int slots[27];
  const int ns = cell_neighbourhood(keys[i], nc.x, nc.y, nc.z, slots);
  unsigned total = 0;
  const unsigned base = (unsigned)i * cap;
  for (int sl = 0; sl < ns; ++sl) {
    const int s = cell_start[slots[sl]], e = cell_start[slots[sl] + 1];
    for (int j0 = s; j0 < e; j0 += 64) {
      const int j = j0 + lane;
      bool hit = false;
      if (j < e && j != i && mol_s[j] != mi) {
        const double4 pj = posq[j];
        double dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
        dx -= box.L[0] * floor(dx * box.invL[0] + 0.5);
        dy -= box.L[1] * floor(dy * box.invL[1] + 0.5);
        dz -= box.L[2] * floor(dz * box.invL[2] + 0.5);
        hit = (dx * dx + dy * dy + dz * dz) < rv2;
      }
      const unsigned long long m = __ballot(hit);
      if (FILL && hit) {
        const unsigned slot = total + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        if (slot < cap) nbr[base + slot] = (int)((unsigned)j | ((unsigned)type_s[j] << NBR_IDX_BITS));
      }
      total += (unsigned)__popcll(m);
    }
  }
  if (lane == 0) { count[i] = total; if (FILL) start[i] = base; }
}

// LDS-staged list build: ONE WORKGROUP PER CELL.  The cell's whole neighbourhood (its <= 27 neighbour cells, ~3 600 atoms at liquid
// density) is staged once in LDS as {x, y, z (fp32, relative to the cell centre, moved to the periodic image next to this cell), molecule}
// + the finished list entry word; the 16 wavefronts take the cell's atoms round-robin and scan the tile 64 candidates per trip.
// The scan is a SINGLE-PRECISION pre-filter: |r2_f32 - rv2| >= 1e-3 rv2 decides (the fp32 error of r2 is < 3e-4 A^2 for coordinates
// < 20 A); the ~1 candidate per atom inside that band is decided in fp64 exactly as the wave-per-atom kernel does, including the
// tie-zone re-evaluation with the reference's own arithmetic (coordinates read from a .gro are multiples of 0.01 A: exact ties occur).
// Same traversal order, same decisions => the same rows.  20 B per tile atom: two workgroups share a CU.  A neighbourhood that does
// not fit the tile (tile_cap atoms) is scanned from global memory by the same loop.  (The fp64 version of this scan cost 0.83 ms per
// rebuild at 98 k atoms, the wave-per-atom kernel with its per-candidate floor()s 0.95 ms.)
__device__ __forceinline__ bool nbr_decide_fp64(const double4& pr, double xi, double yi, double zi, const double4& q, double shx, double shy, double shz,
                                                const BoxDev& box, double rv2) {
  const double xj = (q.x - box.L[0] * floor(q.x * box.invL[0])) + shx, yj = (q.y - box.L[1] * floor(q.y * box.invL[1])) + shy, zj = (q.z - box.L[2] * floor(q.z * box.invL[2])) + shz;
  const double dx = xi - xj, dy = yi - yj, dz = zi - zj;
  const double r2 = dx * dx + dy * dy + dz * dz;
  bool hit = r2 < rv2;
  if (fabs(r2 - rv2) < 1e-9 * rv2) {      // tie zone: decide with the reference's arithmetic on the raw coordinates
    double ex = pr.x - q.x, ey = pr.y - q.y, ez = pr.z - q.z;
    ex -= box.L[0] * floor(ex * box.invL[0] + 0.5);
    ey -= box.L[1] * floor(ey * box.invL[1] + 0.5);
    ez -= box.L[2] * floor(ez * box.invL[2] + 0.5);
    hit = (ex * ex + ey * ey + ez * ez) < rv2;
  }
  return hit;
}

// HALF: the tile is the HALF neighbourhood (the cell + its 13 forward cells, cell_half_neighbourhood order), an atom keeps the partners
// that come AFTER it in the tile (same cell: larger sorted index; forward cells: all) - every pair once - and the entries hold the
// partner's position in the tile: the list of the half-list LDS pair kernel (pair.hip).  The set of unordered pairs is the same.
template <bool HALF, bool SPLIT>      // SPLIT: also count where every row crosses into the eighth cell of its tile (only the split pair kernel, option pair_split, reads that)
__global__ __launch_bounds__(1024, 2) void k_neighbors_tiles(int3 nc, const int* __restrict__ cell_start, const double4* __restrict__ posq, const int* __restrict__ mol_s,
                                                          const int* __restrict__ type_s, BoxDev box, double rv2, unsigned cap, int tile_cap,
                                                          unsigned* __restrict__ count, unsigned* __restrict__ start, int* __restrict__ nbr, int* __restrict__ stats, unsigned* __restrict__ split) {
  extern __shared__ float4 t4[];                          // {x, y, z, molecule (int bits)}[tile_cap], then the entry words int[tile_cap]
  __shared__ int s_lo[27], s_off[28], s_ns;
  __shared__ double s_shift[27][3];
  const int cidx = blockIdx.x;
  const int cx = cidx % nc.x, cy = (cidx / nc.x) % nc.y, cz = cidx / (nc.x * nc.y);
  if (threadIdx.x == 0) {
    int slots[27];
    int ns;
    if (HALF) { cell_half_neighbourhood(cidx, nc.x, nc.y, nc.z, slots); ns = HALF_SLOTS; }
    else ns = cell_neighbourhood(cidx, nc.x, nc.y, nc.z, slots);
    int off = 0;
    for (int s = 0; s < ns; ++s) {
      s_lo[s] = cell_start[slots[s]]; s_off[s] = off; off += cell_start[slots[s] + 1] - s_lo[s];
      // periodic image of the neighbour cell that touches this cell (the host guarantees >= 4 cells per axis)
      const int d[3] = {slots[s] % nc.x - cx, (slots[s] / nc.x) % nc.y - cy, slots[s] / (nc.x * nc.y) - cz};
      for (int q = 0; q < 3; ++q) s_shift[s][q] = d[q] > 1 ? -box.L[q] : (d[q] < -1 ? box.L[q] : 0.0);
    }
    s_off[ns] = off; s_ns = ns;
  }
  __syncthreads();
  const int ns = s_ns, ntile = s_off[ns];
  const int off7 = HALF ? s_off[7] : 0;      // tile position of the eighth cell: rows are cut there for the two workgroups per cell of the split pair kernel (pair_flat.hip)
  const bool in_lds = ntile <= tile_cap;
  int* te = (int*)(t4 + tile_cap + 2);      // (+ 2 spare float4: the last pair of an odd-sized tile reads one entry past it)
  // cell centre: tile coordinates are within +-1.5 cells of it
  const double ox = (cx + 0.5) * box.L[0] / nc.x, oy = (cy + 0.5) * box.L[1] / nc.y, oz = (cz + 0.5) * box.L[2] / nc.z;
  if (in_lds) {      // ONE pass over the tile: every thread has its loads in flight together (a loop over the slots waits for memory once per slot)
    for (int q = threadIdx.x; q < ntile; q += blockDim.x) {
      int s = 0;
      while (s + 1 < ns && q >= s_off[s + 1]) ++s;
      const int j = s_lo[s] + (q - s_off[s]);
      const double4 p = posq[j];
      float4 t;
      t.x = (float)(((p.x - box.L[0] * floor(p.x * box.invL[0])) + s_shift[s][0]) - ox);
      t.y = (float)(((p.y - box.L[1] * floor(p.y * box.invL[1])) + s_shift[s][1]) - oy);
      t.z = (float)(((p.z - box.L[2] * floor(p.z * box.invL[2])) + s_shift[s][2]) - oz);
      t.w = __int_as_float(mol_s[j]);
      t4[q] = t;
      te[q] = (int)((HALF ? (unsigned)q : (unsigned)j) | ((unsigned)type_s[j] << NBR_IDX_BITS));
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), wpb = blockDim.x >> 6;
  const int i1 = cell_start[cidx + 1];
  const float rv2f = (float)rv2, band = 1e-3f * (float)rv2;
  unsigned wmax = 0; unsigned long long wsum = 0;
  for (int i = cell_start[cidx] + wave; i < i1; i += wpb) {
    const double4 pr = posq[i];
    const double xi = pr.x - box.L[0] * floor(pr.x * box.invL[0]), yi = pr.y - box.L[1] * floor(pr.y * box.invL[1]), zi = pr.z - box.L[2] * floor(pr.z * box.invL[2]);
    const float fx = (float)(xi - ox), fy = (float)(yi - oy), fz = (float)(zi - oz);
    const int mi = mol_s[i];
    unsigned total = 0, total_lo = 0;
    const unsigned base = (unsigned)i * cap;
    const int pfirst = HALF ? i - cell_start[cidx] + 1 : 0;      // half list: first tile position that counts as a partner of this atom
    if (in_lds) {
      // two ADJACENT candidates per lane and trip (128 per trip): one 32-byte and one 8-byte LDS read per lane, the fp32 arithmetic of
      // the pair packs into v_pk_* instructions; the compaction keeps the candidates' order (slot = hits before this lane's pair, then A, then B)
      for (int p0 = HALF ? (pfirst & ~127) : 0; p0 < ntile; p0 += 128) {
        const int pA = p0 + 2 * lane, pB = pA + 1;
        const int pr_ = pA < ntile ? pA : 0;      // even index; entry pr_ + 1 <= ntile exists (the tile has one spare entry)
        const float4 tA = t4[pr_], tB = t4[pr_ + 1];
        const int2 en = *(const int2*)(te + pr_);
        const float dxA = fx - tA.x, dyA = fy - tA.y, dzA = fz - tA.z, dxB = fx - tB.x, dyB = fy - tB.y, dzB = fz - tB.z;
        const float r2A = dxA * dxA + dyA * dyA + dzA * dzA, r2B = dxB * dxB + dyB * dyB + dzB * dzB;
        const bool cA = pA < ntile && pA >= pfirst && __float_as_int(tA.w) != mi, cB = pB < ntile && pB >= pfirst && __float_as_int(tB.w) != mi;
        bool hA = cA && r2A < rv2f, hB = cB && r2B < rv2f;
        const bool bA = cA && fabsf(r2A - rv2f) < band, bB = cB && fabsf(r2B - rv2f) < band;
        if (__ballot(bA || bB)) {      // rare (about one candidate per atom): too close to call in single precision
          for (int q = 0; q < 2; ++q) {
            if (q == 0 ? bA : bB) {
              const int p = q == 0 ? pA : pB;
              int s2 = 0;
              while (p >= s_off[s2 + 1]) ++s2;
              const int j = s_lo[s2] + (p - s_off[s2]);
              const bool hit = nbr_decide_fp64(pr, xi, yi, zi, posq[j], s_shift[s2][0], s_shift[s2][1], s_shift[s2][2], box, rv2);
              if (q == 0) hA = hit; else hB = hit;
            }
          }
        }
        const unsigned long long mA = __ballot(hA), mB = __ballot(hB);
        const unsigned before = __builtin_amdgcn_mbcnt_hi((unsigned)(mA >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mA, 0u)) +
                                __builtin_amdgcn_mbcnt_hi((unsigned)(mB >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mB, 0u));
        const unsigned slotA = total + before, slotB = slotA + (hA ? 1u : 0u);
        if (hA && slotA < cap) nbr[base + slotA] = en.x;
        if (hB && slotB < cap) nbr[base + slotB] = en.y;
        total += (unsigned)(__popcll(mA) + __popcll(mB));
        if (SPLIT && p0 < off7) total_lo += (unsigned)(__popcll(__ballot(hA && pA < off7)) + __popcll(__ballot(hB && pB < off7)));
      }
    } else {           // neighbourhood larger than the tile: the same decisions from global memory
      for (int sl = 0; sl < ns; ++sl) {
        const int lo = s_lo[sl], o = s_off[sl], n = s_off[sl + 1] - o;
        const double shx = s_shift[sl][0], shy = s_shift[sl][1], shz = s_shift[sl][2];
        for (int p0 = 0; p0 < n; p0 += 64) {
          const int p = p0 + lane, j = lo + p;
          bool hit = false;
          if (p < n && o + p >= pfirst && mol_s[j] != mi) hit = nbr_decide_fp64(pr, xi, yi, zi, posq[j], shx, shy, shz, box, rv2);
          const unsigned long long m = __ballot(hit);
          if (hit) {
            const unsigned slot = total + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
            const unsigned idx = HALF ? (unsigned)(o + p) : (unsigned)j;
            if (slot < cap) nbr[base + slot] = (int)(idx | ((unsigned)type_s[j] << NBR_IDX_BITS));
          }
          total += (unsigned)__popcll(m);
          if (SPLIT) total_lo += (unsigned)__popcll(__ballot(hit && o + p < off7));
        }
      }
    }
    if (lane == 0) { count[i] = stats ? min(total, cap) : total; start[i] = base; if (SPLIT) split[i] = min(total_lo, cap); }      // with in-kernel statistics the row is cut here (see k_count_stats)
    wmax = max(wmax, total); wsum += total;
  }
  if (stats) {      // {largest neighbourhood, -, longest row (uncut), -, sum of the row lengths (64 bit)}: one atomic each per workgroup
    __shared__ unsigned s_max[16];
    __shared__ unsigned long long s_sum[16];
    if (lane == 0) { s_max[wave] = wmax; s_sum[wave] = wsum; }
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned m = 0; unsigned long long sm = 0;
      for (int k = 0; k < wpb; ++k) { m = max(m, s_max[k]); sm += s_sum[k]; }
      atomicMax((unsigned*)stats + 2, m); atomicAdd((unsigned long long*)(stats + 4), sm); atomicMax(stats + (HALF ? 1 : 0), ntile); atomicMax(stats + 6, i1 - cell_start[cidx]);
      if (HALF) atomicMax(stats + 7, max(s_off[7], s_off[1] + ntile - s_off[7]));
    }
  }
}

// row statistics {max, pad, sum (64 bit)}; rows longer than the stride (they were not written past it) are cut to it, so that a pair
// loop enqueued before the host has seen the statistics stays inside its rows - the host then rebuilds with a larger stride
__global__ void k_count_stats(int N, unsigned* __restrict__ count, unsigned cap, unsigned* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned v = i < N ? count[i] : 0u;
  if (cap && v > cap) count[i] = cap;
  unsigned mx = v;
  unsigned long long sm = v;
  for (int o = 32; o > 0; o >>= 1) { mx = max(mx, (unsigned)__shfl_xor((int)mx, o, 64)); sm += (unsigned long long)__shfl_xor((long long)sm, o, 64); }
  if ((threadIdx.x & 63) == 0) { atomicMax(&out[0], mx); atomicAdd((unsigned long long*)(out + 2), sm); }
}

// accumulate displacements since the last list build and reduce the two largest norms per block
__global__ __launch_bounds__(256) void k_displacements(int N, const double* __restrict__ xyz, double* __restrict__ store, double* __restrict__ disp,
                                                       BoxDev box, int initialize, double* __restrict__ part1, double* __restrict__ part2) {
  __shared__ double s1[256], s2[256];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  double nrm = 0.0;
  if (i < N) {
    double a[3];
    for (int d = 0; d < 3; ++d) {
      const double x = xyz[3 * i + d];
      if (initialize) { disp[3 * i + d] = 0.0; a[d] = 0.0; }
      else {
        const double dc = x - store[3 * i + d];
        const double sh = floor(dc * box.invL[d] + 0.5) * box.L[d];
        a[d] = disp[3 * i + d] + (dc - sh);
        disp[3 * i + d] = a[d];
      }
      store[3 * i + d] = x;
    }
    nrm = sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
  }
  s1[threadIdx.x] = nrm;
  s2[threadIdx.x] = 0.0;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (threadIdx.x < st) {
      double a1 = s1[threadIdx.x], a2 = s2[threadIdx.x], b1 = s1[threadIdx.x + st], b2 = s2[threadIdx.x + st];
      double m1 = fmax(a1, b1), m2 = fmax(fmin(a1, b1), fmax(a2, b2));
      s1[threadIdx.x] = m1;
      s2[threadIdx.x] = m2;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { part1[blockIdx.x] = s1[0]; part2[blockIdx.x] = s2[0]; }
}

// Per-step refresh between list builds, one launch: thread k = sorted slot k (atom i = perm[k]) re-gathers {x,y,z,q}, type and
// molecule into the cell-sorted arrays, accumulates the atom's displacement since the last build (update_verlet_displacements,
// src/general_routines.f90:1259) with the per-block top-2 norms, and zeroes the atom's force accumulator.
__global__ __launch_bounds__(256) void k_step_refresh(int N, const int* __restrict__ perm, const double* __restrict__ xyz, const double* __restrict__ chg,
                                                      const int* __restrict__ atype, const int* __restrict__ mol_of, double4* __restrict__ posq, int* __restrict__ type_s,
                                                      int* __restrict__ mol_s, double* __restrict__ store, double* __restrict__ disp, BoxDev box,
                                                      double* __restrict__ F, double* __restrict__ part1, double* __restrict__ part2,
                                                      const double4* __restrict__ posq_prune, double* __restrict__ part3, double* __restrict__ part4) {
  __shared__ double s1[256], s2[256];
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  double nrm = 0.0, nrm_p = 0.0;
  if (k < N) {
    const int i = perm[k];
    const double x[3] = {xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
    if (posq_prune) {      // displacement since the pair list was pruned (snapshot taken by the pair loop, sorted order)
      const double4 q = posq_prune[k];
      double e[3] = {x[0] - q.x, x[1] - q.y, x[2] - q.z};
      for (int d = 0; d < 3; ++d) e[d] -= box.L[d] * floor(e[d] * box.invL[d] + 0.5);
      nrm_p = sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
    }
    posq[k] = make_double4(x[0], x[1], x[2], chg[i]);
    type_s[k] = atype[i];
    mol_s[k] = mol_of[i];
    double a[3];
    for (int d = 0; d < 3; ++d) {
      const double dc = x[d] - store[3 * i + d];
      const double sh = floor(dc * box.invL[d] + 0.5) * box.L[d];
      a[d] = disp[3 * i + d] + (dc - sh);
      disp[3 * i + d] = a[d];
      store[3 * i + d] = x[d];
      F[3 * i + d] = 0.0;
    }
    nrm = sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
  }
  s1[threadIdx.x] = nrm;
  s2[threadIdx.x] = 0.0;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (threadIdx.x < st) {
      double a1 = s1[threadIdx.x], a2 = s2[threadIdx.x], b1 = s1[threadIdx.x + st], b2 = s2[threadIdx.x + st];
      double m1 = fmax(a1, b1), m2 = fmax(fmin(a1, b1), fmax(a2, b2));
      s1[threadIdx.x] = m1;
      s2[threadIdx.x] = m2;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { part1[blockIdx.x] = s1[0]; part2[blockIdx.x] = s2[0]; }
  if (!posq_prune) return;
  __syncthreads();
  s1[threadIdx.x] = nrm_p;
  s2[threadIdx.x] = 0.0;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (threadIdx.x < st) {
      double a1 = s1[threadIdx.x], a2 = s2[threadIdx.x], b1 = s1[threadIdx.x + st], b2 = s2[threadIdx.x + st];
      s1[threadIdx.x] = fmax(a1, b1);
      s2[threadIdx.x] = fmax(fmin(a1, b1), fmax(a2, b2));
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { part3[blockIdx.x] = s1[0]; part4[blockIdx.x] = s2[0]; }
}

// blockIdx.x = 0: displacement since the last list build -> out[0..1] (the host's rebuild flag); blockIdx.x = 1: displacement since the
// last pruning -> out[2..3] and the pair loop's mode for THIS step: two atoms approach by at most d1 + d2, so the pruned rows (all pairs
// within rc + delta when they were written) still hold every pair within rc while d1 + d2 < delta
__global__ void k_top2_final(int nb, const double* __restrict__ part1, const double* __restrict__ part2, double* __restrict__ out,
                             const double* __restrict__ part3, const double* __restrict__ part4, double prune_delta, int* __restrict__ prune_mode) {
  __shared__ double s1[256], s2[256];
  if (blockIdx.x == 1) { part1 = part3; part2 = part4; out += 2; }
  double m1 = 0, m2 = 0;
  for (int k = threadIdx.x; k < nb; k += 256) {
    double b1 = part1[k], b2 = part2[k];
    double n1 = fmax(m1, b1), n2 = fmax(fmin(m1, b1), fmax(m2, b2));
    m1 = n1; m2 = n2;
  }
  s1[threadIdx.x] = m1; s2[threadIdx.x] = m2;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (threadIdx.x < st) {
      double a1 = s1[threadIdx.x], a2 = s2[threadIdx.x], b1 = s1[threadIdx.x + st], b2 = s2[threadIdx.x + st];
      s1[threadIdx.x] = fmax(a1, b1);
      s2[threadIdx.x] = fmax(fmin(a1, b1), fmax(a2, b2));
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[0] = s1[0]; out[1] = s2[0];
    // prune_mode[2]: pruned rows exist for the CURRENT full rows (cleared by nb_prune_reset, set by a pair loop that wrote them)
    if (blockIdx.x == 1 && prune_mode) { const int use = prune_mode[2] != 0 && (s1[0] + s2[0]) < prune_delta; prune_mode[0] = use ? 2 : 1; prune_mode[1] += use ? 0 : 1; }      // (prune_mode == nullptr: the pair kernel decides for itself)
  }
}

}  // namespace

static BoxDev make_box(const msevb_ctx* c) {
  BoxDev b;
  for (int d = 0; d < 3; ++d) { b.L[d] = c->box[d]; b.invL[d] = 1.0 / c->box[d]; }
  return b;
}

int nb_refresh_sorted(msevb_ctx* c) {
  const int N = c->N;
  const bool reset_here = !c->prune_freeze && c->prune && c->d_prune_mode.p;      // nb_prune_reset's two stores, by the kernel
  c->stats_zeroed_now = c->d_tile_max.p != nullptr;
  hipLaunchKernelGGL(k_gather_sorted, dim3((N + 255) / 256), dim3(256), 0, c->stream, N, c->d_perm.p, c->d_xyz.p, c->d_chg.p, c->d_atype.p,
                     c->d_mol_of.p, c->d_posq.p, c->d_type_sorted.p, c->d_mol_sorted.p, c->d_inv_perm.p,
                     c->rebuild_zero_force_now ? c->d_force.p : (double*)nullptr, reset_here ? c->d_prune_mode.p : (int*)nullptr, c->d_tile_max.p);
  HIPCHK(hipGetLastError());
  c->rebuild_zero_force_now = false;      // done (the caller looks)
  if (reset_here || c->prune_freeze) return MSEVB_OK;
  return nb_prune_reset(c);      // positions were replaced wholesale: whatever pruned rows exist are not known to cover them
}

// the next pair loop runs on the full rows and writes fresh pruned ones
int nb_prune_reset(msevb_ctx* c) {
  if (!c->prune) return MSEVB_OK;
  if (!c->d_prune_mode.p) {
    if (c->d_prune_mode.alloc(4)) FAIL(MSEVB_ERR_HIP, "hipMalloc failed (prune state)");
    HIPCHK(hipMemsetAsync(c->d_prune_mode.p, 0, 4 * sizeof(int), c->stream));
  }
  HIPCHK(hipMemsetD32Async((hipDeviceptr_t)c->d_prune_mode.p, 1, 1, c->stream));            // mode: prune
  HIPCHK(hipMemsetD32Async((hipDeviceptr_t)(c->d_prune_mode.p + 2), 0, 1, c->stream));      // no pruned rows for these full rows yet
  return MSEVB_OK;
}

// displacement check + sorted-array refresh + force zeroing of a step that keeps the current list (one kernel + the tiny top-2 finish)
int nb_step_refresh(msevb_ctx* c, hipStream_t st, bool with_top2) {
  if (!st) st = c->stream;
  const int N = c->N, nb = (N + 255) / 256;
  const bool pr = c->prune && c->d_posq_prune.p && c->d_prune_mode.p;
  static_assert(S_MAXD2 == S_MAXD1 + 1 && S_PRD1 == S_MAXD1 + 2 && S_PRD2 == S_MAXD1 + 3, "displacement slots are contiguous");
  hipLaunchKernelGGL(k_step_refresh, dim3(nb), dim3(256), 0, st, N, c->d_perm.p, c->d_xyz.p, c->d_chg.p, c->d_atype.p, c->d_mol_of.p, c->d_posq.p,
                     c->d_type_sorted.p, c->d_mol_sorted.p, c->d_xyz_store.p, c->d_disp.p, make_box(c), c->d_force.p,
                     c->d_partials.p + (size_t)S_MAXD1 * MAX_BLOCKS, c->d_partials.p + (size_t)S_MAXD2 * MAX_BLOCKS,
                     pr ? c->d_posq_prune.p : (const double4*)nullptr, c->d_partials.p + (size_t)S_PRD1 * MAX_BLOCKS, c->d_partials.p + (size_t)S_PRD2 * MAX_BLOCKS);
  if (with_top2)
  hipLaunchKernelGGL(k_top2_final, dim3(pr ? 2 : 1), dim3(256), 0, st, nb, c->d_partials.p + (size_t)S_MAXD1 * MAX_BLOCKS,
                     c->d_partials.p + (size_t)S_MAXD2 * MAX_BLOCKS, c->d_energies.p + S_MAXD1,
                     c->d_partials.p + (size_t)S_PRD1 * MAX_BLOCKS, c->d_partials.p + (size_t)S_PRD2 * MAX_BLOCKS, c->prune_delta, c->d_prune_mode.p);
  HIPCHK(hipGetLastError());
  return MSEVB_OK;
}

int nb_step_top2(msevb_ctx* c, int nb, hipStream_t st, bool decide) {
  if (!st) st = c->stream;
  const bool pr = c->prune && c->d_posq_prune.p && c->d_prune_mode.p;
  hipLaunchKernelGGL(k_top2_final, dim3(pr ? 2 : 1), dim3(256), 0, st, nb, c->d_partials.p + (size_t)S_MAXD1 * MAX_BLOCKS,
                     c->d_partials.p + (size_t)S_MAXD2 * MAX_BLOCKS, c->d_energies.p + S_MAXD1,
                     c->d_partials.p + (size_t)S_PRD1 * MAX_BLOCKS, c->d_partials.p + (size_t)S_PRD2 * MAX_BLOCKS, c->prune_delta, decide ? c->d_prune_mode.p : (int*)nullptr);
  HIPCHK(hipGetLastError());
  return MSEVB_OK;
}

// A (re)build in two halves so that the caller can start everything that only needs the CELL structure (PME spreading, per-molecule
// terms) on the side streams while the rows are still being filled: nb_build_begin = cell sort + sorted arrays, nb_build_finish = rows
// (+ the one host synchronisation of a build).  nb_build = both.
int nb_build(msevb_ctx* c) { CHK(nb_build_begin(c)); return nb_build_finish(c, false); }

int nb_build_begin(msevb_ctx* c) {
  const int N = c->N;
  if ((unsigned)N > NBR_IDX_MASK || c->nt > 256) FAIL(MSEVB_ERR_LIMIT, "Verlet list entries hold 24-bit atom indices and 8-bit atom types");
  phase_begin(c, PH_VERLET);
  for (int d = 0; d < 3; ++d) { int n = (int)floor(c->box[d] / c->rv); c->ncell[d] = n < 1 ? 1 : n; }
  const int3 nc = make_int3(c->ncell[0], c->ncell[1], c->ncell[2]);
  const int ncell = nc.x * nc.y * nc.z;
  const BoxDev box = make_box(c);
  if (c->d_cell_of.alloc(N) || c->d_perm.alloc(N) || c->d_cell_keys_tmp.alloc(N) || c->d_perm_tmp.alloc(N) || c->d_cell_start.alloc(ncell + 1) ||
      c->d_nbr_count.alloc(N + 1) || c->d_nbr_start.alloc(N + 1) || c->d_nbr_split.alloc(N + 1) || c->d_inv_perm.alloc(N))
    FAIL(MSEVB_ERR_HIP, "hipMalloc failed (neighbour structure)");
  if (c->cell_sort_own && N <= 262144 && ncell <= 4096) {      // four kernels; the same permutation as the radix sort below (stable)
    const int nblk = (N + CS_APB - 1) / CS_APB;
    if (c->d_sort_hist.alloc((size_t)nblk * ncell + ncell)) FAIL(MSEVB_ERR_HIP, "hipMalloc failed (sort workspace)");
    hipLaunchKernelGGL(k_cs_hist, dim3(nblk), dim3(256), (size_t)ncell * sizeof(int), c->stream, N, c->d_xyz.p, box, nc, ncell, c->d_cell_keys_tmp.p, c->d_sort_hist.p);
    hipLaunchKernelGGL(k_cs_scan, dim3(ncell), dim3(64), 0, c->stream, nblk, ncell, c->d_sort_hist.p, c->d_sort_hist.p + (size_t)nblk * ncell);
    hipLaunchKernelGGL(k_cs_cells, dim3(1), dim3(1024), 0, c->stream, ncell, N, (const int*)(c->d_sort_hist.p + (size_t)nblk * ncell), c->d_cell_start.p);
    hipLaunchKernelGGL(k_cs_scatter, dim3(nblk), dim3(64), (size_t)ncell * sizeof(int), c->stream, N, ncell, c->d_cell_keys_tmp.p, c->d_sort_hist.p, c->d_cell_start.p, c->d_perm.p, c->d_cell_of.p);
    HIPCHK(hipGetLastError());
    CHK(nb_refresh_sorted(c));
    c->list_valid = false;
    c->cells_valid = true;
    return MSEVB_OK;
  }
  hipLaunchKernelGGL(k_cell_assign, dim3((N + 255) / 256), dim3(256), 0, c->stream, N, c->d_xyz.p, box, nc, c->d_cell_keys_tmp.p, c->d_perm_tmp.p);
  int bits = 1;
  while ((1 << bits) < ncell) ++bits;
  size_t tmp = 0;
  HIPCHK(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp, c->d_cell_keys_tmp.p, c->d_cell_of.p, c->d_perm_tmp.p, c->d_perm.p, N, 0, bits, c->stream));
  if (c->d_sort_tmp.alloc(tmp + 256)) FAIL(MSEVB_ERR_HIP, "hipMalloc failed (sort workspace)");
  size_t tb = c->d_sort_tmp.n;
  HIPCHK(hipcub::DeviceRadixSort::SortPairs(c->d_sort_tmp.p, tb, c->d_cell_keys_tmp.p, c->d_cell_of.p, c->d_perm_tmp.p, c->d_perm.p, N, 0, bits, c->stream));
  hipLaunchKernelGGL(k_cell_bounds, dim3((N + 255) / 256), dim3(256), 0, c->stream, N, ncell, c->d_cell_of.p, c->d_cell_start.p);
  CHK(nb_refresh_sorted(c));
  c->list_valid = false;      // cells are current, rows are not yet
  c->cells_valid = true;
  return MSEVB_OK;
}

// defer: the row statistics of a REbuild travel to pinned host memory without a synchronisation; the caller looks at them with
// nb_check_overflow() once it has waited for something later on the stream, and repeats its force call in the (rare) overflow case.
int nb_build_finish(msevb_ctx* c, bool defer) {
  const int N = c->N;
  const int3 nc = make_int3(c->ncell[0], c->ncell[1], c->ncell[2]);
  const int ncell = nc.x * nc.y * nc.z;
  const BoxDev box = make_box(c);
  const int wpb = 4;
  const dim3 grid((N + wpb - 1) / wpb), block(64 * wpb);
  const double rv2 = c->rv * c->rv;
  // tile size of the LDS-staged pair kernel + row statistics scratch: {tile_max, pad, count_max, pad, count_sum(64-bit)}
  if (c->d_tile_max.alloc(8)) FAIL(MSEVB_ERR_HIP, "hipMalloc failed (list statistics)");
  int* d_stats = c->d_tile_max.p;
  if (!c->h_nbr_stats) HIPCHK(hipHostMalloc((void**)&c->h_nbr_stats, 8 * sizeof(int), hipHostMallocDefault));
  int* h_stats = c->h_nbr_stats;
  c->nbr_check_pending = false;
  auto stats = [&](bool with_tile, bool sync = true) -> int {       // one synchronisation per (re)build, unless deferred
    HIPCHK(hipMemsetAsync(d_stats, 0, 8 * sizeof(int), c->stream));
    if (with_tile) hipLaunchKernelGGL(k_tile_max, dim3((ncell + 255) / 256), dim3(256), 0, c->stream, ncell, nc, c->d_cell_start.p, d_stats);
    hipLaunchKernelGGL(k_count_stats, dim3((N + 255) / 256), dim3(256), 0, c->stream, N, c->d_nbr_count.p, c->nbr_cap, (unsigned*)d_stats + 2);
    HIPCHK(hipMemcpyAsync(h_stats, d_stats, 8 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    if (sync) HIPCHK(hipStreamSynchronize(c->stream));
    return MSEVB_OK;
  };
  bool inline_stats = false;      // the fill kernel left {tile max, half tile max, row max, row sum} in d_stats and cut overlong rows itself
  const bool tile_build_ok = ncell >= NBR_TILE_MIN_CELLS && c->nbr_tiles && nc.x >= 4 && nc.y >= 4 && nc.z >= 4;      // workgroup per cell, neighbourhood staged in LDS
  c->nbr_tiles_used = tile_build_ok;
  // Format of THIS build: half list with tile-local entries (the LDS pair kernel, pair.hip) when the largest half neighbourhood the
  // previous build saw (+ 8 %) fits the pair kernel's LDS tile; a build that meets a larger one is repeated (nb_check_overflow).
  auto decide_format = [&]() {
    int cap = ((int)(c->half_tile_max * 1.08) + 63) & ~63;
    if (cap < c->half_tile_cap_seen && (size_t)c->half_tile_cap_seen * PAIR_TILE_ATOM_BYTES + (size_t)c->nt * c->nt * sizeof(PairParam) + 2048 <= (size_t)PAIR_TILE_LDS_BYTES) cap = c->half_tile_cap_seen;      // never smaller than a tile this context has used: the pair kernel's launch (pitch, capacity) stays what a captured force call holds (evb.hip, step graph)
    const size_t bytes = (size_t)cap * PAIR_TILE_ATOM_BYTES + (size_t)c->nt * c->nt * sizeof(PairParam) + 2048;
    c->pair_tiles = tile_build_ok && c->pair_tiles_opt != 0 && c->half_tile_max > 0 && bytes <= (size_t)PAIR_TILE_LDS_BYTES && c->cell_atoms_max <= PAIR_CELL_MAX;
    c->half_tile_cap = c->pair_tiles ? cap : 0;
    if (c->pair_tiles) c->half_tile_cap_seen = cap;
  };
  auto fill = [&](bool want_inline = false) -> int {
    if (c->d_nbr.alloc((size_t)N * c->nbr_cap)) FAIL(MSEVB_ERR_HIP, "hipMalloc failed (neighbour list)");
    inline_stats = false;
    if (tile_build_ok) {
      auto kern = c->pair_tiles ? (c->pair_split_opt ? k_neighbors_tiles<true, true> : k_neighbors_tiles<true, false>) : k_neighbors_tiles<false, false>;
      const int which = c->pair_tiles ? (c->pair_split_opt ? 2 : 1) : 0;
      if (!c->attr_nbr_tiles[which]) { HIPCHK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, NBR_TILE_CAP * NBR_TILE_BYTES + 64)); c->attr_nbr_tiles[which] = true; }      // (once per context: the call costs ~30 us of host time in front of the row kernel)
      // tile: sized from the largest neighbourhood the last build saw
      // (a larger one is scanned from global memory).  Up to NBR_TILE_CAP2 atoms two workgroups share a CU's 160 KB.
      int tile_cap = NBR_TILE_CAP2;
      const int seen = c->pair_tiles ? c->half_tile_max : c->tile_atoms_max;
      if (seen > 0 && seen * 1.03 > NBR_TILE_CAP2) tile_cap = std::min(NBR_TILE_CAP, ((int)(seen * 1.08) + 63) & ~63);
      if (want_inline) { if (!c->stats_zeroed_now) HIPCHK(hipMemsetAsync(d_stats, 0, 8 * sizeof(int), c->stream)); c->stats_zeroed_now = false; inline_stats = true; }      // (k_gather_sorted of this rebuild has zeroed them already; a repeated fill zeroes them itself)
      hipLaunchKernelGGL(kern, dim3(ncell), dim3(1024), (size_t)tile_cap * NBR_TILE_BYTES + 64, c->stream, nc, c->d_cell_start.p, c->d_posq.p, c->d_mol_sorted.p, c->d_type_sorted.p, box, rv2,
                         c->nbr_cap, tile_cap, c->d_nbr_count.p, c->d_nbr_start.p, c->d_nbr.p, want_inline ? d_stats : (int*)nullptr, c->d_nbr_split.p);
    } else
      hipLaunchKernelGGL((k_neighbors<true>), grid, block, 0, c->stream, N, c->d_posq.p, c->d_mol_sorted.p, c->d_type_sorted.p, c->d_cell_of.p, c->d_cell_start.p, box, nc, rv2,
                         c->nbr_cap, c->d_nbr_count.p, c->d_nbr_start.p, c->d_nbr.p);
    HIPCHK(hipGetLastError());
    return MSEVB_OK;
  };
  auto set_cap = [&](unsigned maxcount) { c->nbr_cap = ((unsigned)(maxcount * 1.15) + 64u + 63u) & ~63u; };
  if (c->nbr_cap == 0 || (c->pair_tiles_opt != 0 && tile_build_ok && c->half_tile_max == 0)) {      // first build: a counting pass sizes the row stride (full rows: an upper bound of the half rows) and the tiles
    hipLaunchKernelGGL((k_neighbors<false>), grid, block, 0, c->stream, N, c->d_posq.p, c->d_mol_sorted.p, c->d_type_sorted.p, c->d_cell_of.p, c->d_cell_start.p, box, nc, rv2,
                       0u, c->d_nbr_count.p, (unsigned*)nullptr, (int*)nullptr);
    const unsigned cap_before = c->nbr_cap;
    c->nbr_cap = 0;      // k_count_stats must not cut the counted rows
    CHK(stats(true));
    if (cap_before == 0 || (unsigned)h_stats[2] > cap_before) set_cap((unsigned)h_stats[2]); else c->nbr_cap = cap_before;
    c->tile_atoms_max = h_stats[0]; c->half_tile_max = h_stats[1]; c->cell_atoms_max = h_stats[6]; c->sub_tile_max = h_stats[7];
    decide_format();
    CHK(fill());
    CHK(stats(false));
  } else {                    // rebuild: one pass; the stride only grows when a row overflowed
    decide_format();
    CHK(fill(defer));
    if (defer) {      // statistics without a synchronisation: by the fill kernel itself when it is the tile kernel (nothing between it and the pair loop)
      if (inline_stats) c->nbr_stats_copy_pending = true;      // downloaded behind the pair loop (pair_forces): the host only looks at its next wait
      else CHK(stats(true, false));      // + the largest neighbourhood, for the next build's tile
      c->nbr_check_pending = true;
    } else {
      CHK(stats(true));
      c->tile_atoms_max = h_stats[0];
      const bool tile_grew = c->pair_tiles && (h_stats[1] > c->half_tile_cap || h_stats[6] > PAIR_CELL_MAX);      // a half neighbourhood outgrew the pair kernel's LDS tile, or one cell its row metadata
      if (h_stats[1] > 0) c->half_tile_max = h_stats[1];
      c->cell_atoms_max = h_stats[6]; if (h_stats[7] > 0) c->sub_tile_max = h_stats[7];
      if ((unsigned)h_stats[2] > c->nbr_cap || tile_grew) {
        if ((unsigned)h_stats[2] > c->nbr_cap) set_cap((unsigned)h_stats[2]);
        decide_format(); CHK(fill()); CHK(stats(false));
      }
    }
  }
  if ((double)N * c->nbr_cap >= 4294967296.0) FAIL(MSEVB_ERR_LIMIT, "Verlet list exceeds 32-bit row offsets");
  if (!c->nbr_check_pending) c->n_nbr = (size_t)(((unsigned long long)(unsigned)h_stats[5] << 32) | (unsigned)h_stats[4]);
  HIPCHK(hipGetLastError());
  phase_end(c, PH_VERLET);
  c->list_valid = true;
  int flag;
  CHK(nb_check_displacements(c, 1, &flag));
  return MSEVB_OK;
}

// After a deferred rebuild, once the host has waited for work enqueued behind it: 1 = a row overflowed its stride (the stride has been
// enlarged, the list is marked stale, the caller repeats the force call), 0 = the list is complete.
int nb_check_overflow(msevb_ctx* c) {
  if (c->h_energies[S_BRICK] != 0.0) {      // the brick spread overflowed (denser than any liquid): tile kernel from now on, this force call again
    c->spread_bricks = false; c->h_energies[S_BRICK] = 0.0;
    (void)hipMemsetAsync(c->d_energies.p + S_BRICK, 0, sizeof(double), c->stream);
    return 1;
  }
  if (!c->nbr_check_pending) return 0;
  c->nbr_check_pending = false;
  const int* h = c->h_nbr_stats;
  if (h[0] > 0) c->tile_atoms_max = h[0];
  const bool tile_grew = c->pair_tiles && (h[1] > c->half_tile_cap || h[6] > PAIR_CELL_MAX);      // a half neighbourhood outgrew the LDS tile of the pair kernel, or one cell its row metadata (the kernel skipped that cell)
  if (h[1] > 0) c->half_tile_max = h[1];
  if (h[6] > 0) c->cell_atoms_max = h[6];
  if (h[7] > 0) c->sub_tile_max = h[7];
  if ((unsigned)h[2] > c->nbr_cap || tile_grew) {
    if ((unsigned)h[2] > c->nbr_cap) c->nbr_cap = ((unsigned)((unsigned)h[2] * 1.15) + 64u + 63u) & ~63u;
    c->list_valid = false;
    return 1;
  }
  c->n_nbr = (size_t)(((unsigned long long)(unsigned)h[5] << 32) | (unsigned)h[4]);
  return 0;
}

// a deferred rebuild whose statistics nobody looked at (the call that made it failed before its end): look now
int nb_settle(msevb_ctx* c) {
  if (!c->nbr_check_pending) return MSEVB_OK;
  HIPCHK(hipStreamSynchronize(c->stream));
  (void)nb_check_overflow(c);      // on overflow the list is marked stale and the next use rebuilds it
  return MSEVB_OK;
}

// initialize != 0: reset the displacement store (no host sync).  Otherwise the flag is available in
// c->h_energies[S_MAXD1..2] after the next fetch_energies().
int nb_check_displacements(msevb_ctx* c, int initialize, int* flag) {
  const int N = c->N, nb = (N + 255) / 256;
  const BoxDev box = make_box(c);
  if (c->d_xyz_store.alloc(3 * (size_t)N) || c->d_disp.alloc(3 * (size_t)N)) FAIL(MSEVB_ERR_HIP, "hipMalloc failed (displacements)");
  hipLaunchKernelGGL(k_displacements, dim3(nb), dim3(256), 0, c->stream, N, c->d_xyz.p, c->d_xyz_store.p, c->d_disp.p, box, initialize,
                     c->d_partials.p + (size_t)S_MAXD1 * MAX_BLOCKS, c->d_partials.p + (size_t)S_MAXD2 * MAX_BLOCKS);
  hipLaunchKernelGGL(k_top2_final, dim3(1), dim3(256), 0, c->stream, nb, c->d_partials.p + (size_t)S_MAXD1 * MAX_BLOCKS,
                     c->d_partials.p + (size_t)S_MAXD2 * MAX_BLOCKS, c->d_energies.p + S_MAXD1, (const double*)nullptr, (const double*)nullptr, 0.0, (int*)nullptr);
  HIPCHK(hipGetLastError());
  if (flag) *flag = 0;
  return MSEVB_OK;
}
