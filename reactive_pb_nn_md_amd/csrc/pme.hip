// Smooth PME reciprocal space (replaces pme_reciprocal_space_energy_force, src/pme.f90:28-179):
//   k_pme_prep   scaled coordinates (create_scaled_direct_coordinates, src/general_routines.f90:497) and the per-atom
//                B-spline stencil: nearpt = floor(u), 6 table rows per axis (bit-exact index arithmetic of src/pme.f90:220-247,
//                :412-414) -> weights w[3][6] and derivative weights dw[3][6]
//   k_pme_spread grid_Q (src/pme.f90:184): 216 grid points per atom, fp64 atomics into the real part of the complex grid
//   rocFFT       3-D in-place double-complex forward / backward, unnormalised (MKL DFTI in the reference, :85/:113)
//   k_pme_conv   FQ *= CB fused with the energy 0.5*sum_m CB(m)|FQ(m)|^2 (== 0.5*sum Q*theta, :128, by Parseval)
//   k_pme_gather derivative_grid_Q (src/pme.f90:346): F = -K kk^T sum fac*theta*conv, one wavefront per atom
// HBM layout: complex grid double2[K^3] with n1 fastest (reference Q(n1+1,n2+1,n3+1)); per-atom stencil 36 doubles.
#include "ctx.hpp"

namespace {

// index arithmetic must round exactly like the reference: no FMA contraction in this function.
// One thread per (atom, axis, tap): t = (3 i + l) 6 + k is also the element index of w / dw / splindex, so every store is
// coalesced (the thread-per-atom version wrote 36 doubles at a 288-B stride and cost 8x its bytes in HBM write traffic).
// splindex (the B6 table rows, kept for msevb_get_pme_indices) is only written when asked for.
__global__ __launch_bounds__(256) void k_pme_prep(int N, int K, int G, const double* __restrict__ xyz, double k00, double k01,
                           double k02, double k10, double k11, double k12, double k20, double k21, double k22,
                           const double* __restrict__ B6, const double* __restrict__ B5, double* __restrict__ uscale, int* __restrict__ base,
                           int* __restrict__ splindex, double* __restrict__ w, double* __restrict__ dw, double* __restrict__ fault,
                           const int* __restrict__ inv_perm, int* __restrict__ base_s, double* __restrict__ w_s) {
#pragma clang fp contract(off)
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 18 * N) return;
  const int i = t / 18, r = t - 18 * i, l = r / 6, k = r - 6 * l;
  const double x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
  const double ka = l == 0 ? k00 : l == 1 ? k10 : k20, kb = l == 0 ? k01 : l == 1 ? k11 : k21, kc = l == 0 ? k02 : l == 1 ? k12 : k22;
  double u = (double)K * ((ka * x + kb * y) + kc * z);
  if (u < 0.0) u = u + (double)K; else if (u >= (double)K) u = u - (double)K;
  if (!(u >= 0.0 && u < (double)K)) { *fault = 16.0 * (double)(i + 1) + 1.0; u = 0.5; }      // more than one box away, or not a number: flag it (code 1 + 16 x (atom + 1): total_force_collect names the atom), stay inside the grid
  if (fabs(fmod(u, 1.0)) < 1e-6) u = u + 1e-6;
  const int np = (int)floor(u);
  if (k == 0 && uscale) { uscale[3 * i + l] = u; base[3 * i + l] = np; }      // (uscale == nullptr: weights only, beside k_pme_prep_light - pme_side_weights)
  const int n = np - k;
  const double a1 = u - (double)n, a2 = a1 - 1.0;
  const int g6 = (int)ceil(a1 / 6.0 * (double)G);
  if (splindex) splindex[t] = g6;
  const double wv = B6[g6 - 1];
  w[t] = wv;
  if (inv_perm) {      // the same values in CELL-SORTED atom order: the brick spread scans its neighbourhood with coalesced reads
    const int ks = inv_perm[i];
    w_s[18 * (size_t)ks + r] = wv;
    if (k == 0) base_s[3 * ks + l] = np;
  }
  double d = 0.0;
  if (a1 < 5.0) d = B5[(int)ceil(a1 / 5.0 * (double)G) - 1];
  if (0.0 < a2) d = d - B5[(int)ceil(a2 / 5.0 * (double)G) - 1];
  dw[t] = d;
}

// ---- table rows in integer arithmetic -------------------------------------------------------------------------------
// The reference's row of tap k is ceil(a / 6 * G) with a = u - (nearpt - k) = f + k, f = u - nearpt in (0, 1) (src/pme.f90:220-247; exact in
// fp64), and ceil(a / 5 * G), ceil((a - 1) / 5 * G) for the two B5 rows of the derivative weight (:412-414).  With G = 6 q6 + rho6:
//     (f + k) G / 6 = X + k q6 + k rho6 / 6,   X = f G / 6  =>  row = k q6 + floor(k rho6 / 6) + floor(X) + 1 + [6 frac(X) + (k rho6 mod 6) >= 6]
// as long as 6 frac(X) is not within 1e-8 of an integer (then the reference's two roundings decide: `rare`, the callers take its own
// expression for the whole wavefront).  So ONE product, floor and class per atom and axis (here, by k_pme_prep_light) replace a product, a
// ceil, a subtraction and two compares per TAP in the tiled kernels (36 v_ceil_f64 per lane and trip of the gather, round 3).  Same for G / 5.
// seed.x = floor(X) | class6 << 16 | rare << 31,  seed.y = floor(Y) | class5 << 16,  Y = f G / 5
__device__ __forceinline__ int2 pme_row_seed(double f, double G) {
#pragma clang fp contract(off)
  const double X = f * (G / 6.0), Y = f * (G / 5.0);
  const double c6 = floor(X), c5 = floor(Y);
  const double s6 = (X - c6) * 6.0, s5 = (Y - c5) * 5.0;
  const double k6 = floor(s6), k5 = floor(s5);
  const bool rare = s6 - k6 < 1e-8 || s6 - k6 > 1.0 - 1e-8 || s5 - k5 < 1e-8 || s5 - k5 > 1.0 - 1e-8;
  return make_int2((int)c6 | ((int)k6 << 16) | (rare ? (int)0x80000000u : 0), (int)c5 | ((int)k5 << 16));
}
// per-tap constants of a lane (its tap k is fixed): row = A + floor part of the seed + [class + m >= modulus]
struct RowTap { int A6, m6, A5a, m5a, A5b, m5b; };
__device__ __forceinline__ RowTap pme_row_tap(int k, int Gi) {
  const int q6 = Gi / 6, r6 = Gi % 6, q5 = Gi / 5, r5 = Gi % 5, kb = k - 1;      // (kb = -1 for tap 0: its (a - 1) term does not apply, the row is clamped into the table)
  RowTap t;
  t.A6 = k * q6 + (k * r6) / 6 + 1; t.m6 = (k * r6) % 6;
  t.A5a = k * q5 + (k * r5) / 5 + 1; t.m5a = (k * r5) % 5;
  t.A5b = kb >= 0 ? kb * q5 + (kb * r5) / 5 + 1 : -Gi; t.m5b = kb >= 0 ? (kb * r5) % 5 : 0;
  return t;
}
__device__ __forceinline__ int pme_row6(int2 sd, const RowTap& t) { return t.A6 + (sd.x & 0xffff) + ((((sd.x >> 16) & 7) + t.m6) >= 6 ? 1 : 0); }
__device__ __forceinline__ int pme_row5a(int2 sd, const RowTap& t) { return t.A5a + (sd.y & 0xffff) + ((((sd.y >> 16) & 7) + t.m5a) >= 5 ? 1 : 0); }
__device__ __forceinline__ int pme_row5b(int2 sd, const RowTap& t) { return t.A5b + (sd.y & 0xffff) + ((((sd.y >> 16) & 7) + t.m5b) >= 5 ? 1 : 0); }

// The same scaled coordinate and nearest grid point WITHOUT the 36 weights per atom (option pme_weightless, the default on the tiled
// spread / gather): one thread per (atom, axis), 36 B written per atom instead of 324 B.  The tiled kernels below recompute the weights
// from uscale / base and the spline tables with the arithmetic of k_pme_prep (pme_tap_w / pme_taps3), so the values are the same bits.
__global__ __launch_bounds__(256) void k_pme_prep_light(int N, int K, const double* __restrict__ xyz, double k00, double k01, double k02, double k10, double k11,
                                                        double k12, double k20, double k21, double k22, double* __restrict__ uscale, int* __restrict__ base,
                                                        double* __restrict__ fault, const int* __restrict__ inv_perm, double* __restrict__ uscale_s,
                                                        int* __restrict__ base_s, int2* __restrict__ rows_s, double G) {      // *_s: the same values in cell-sorted atom order (what the tiled kernels read, coalesced)
#pragma clang fp contract(off)
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 3 * N) return;
  const int i = t / 3, l = t - 3 * i;
  const double x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
  const double ka = l == 0 ? k00 : l == 1 ? k10 : k20, kb = l == 0 ? k01 : l == 1 ? k11 : k21, kc = l == 0 ? k02 : l == 1 ? k12 : k22;
  double u = (double)K * ((ka * x + kb * y) + kc * z);
  if (u < 0.0) u = u + (double)K; else if (u >= (double)K) u = u - (double)K;
  if (!(u >= 0.0 && u < (double)K)) { *fault = 16.0 * (double)(i + 1) + 1.0; u = 0.5; }
  if (fabs(fmod(u, 1.0)) < 1e-6) u = u + 1e-6;
  const int np = (int)floor(u);
  uscale[t] = u; base[t] = np;
  const int ts = 3 * inv_perm[i] + l;
  uscale_s[ts] = u; base_s[ts] = np;
  rows_s[ts] = pme_row_seed(u - (double)np, G);
}

// Table rows without the fp64 divisions: (int)ceil(a / div * G) equals ceil(a * (G / div)) unless the product lies within ~1e-10 of an integer
// (the reference rounds twice, the product once: both are a few 1e-11 from the exact quotient at G = 1e5).  A wavefront in which any lane comes
// closer than 1e-9 takes the reference's own expression for all its lanes - a scalar branch with no memory operation behind it, so the
// table loads of all taps go out together.
__device__ __forceinline__ double pme_tap_w(double u, int np, int k, double G, const double* __restrict__ B6) {
  const double a1 = u - (double)(np - k);
  const double t = a1 * (G / 6.0), ct = ceil(t), gap = ct - t;
  int r = (int)ct;
  if (__builtin_expect(__any(gap < 1e-9 || gap > 1.0 - 1e-9), 0)) r = (int)ceil(a1 / 6.0 * G);
  return B6[r - 1];
}
// weight and derivative weight of tap k of the three axes (u, np: scaled coordinate and nearest grid point per axis) - the values k_pme_prep
// stores (src/pme.f90:220-247, :412-414) - in two halves, so that a caller can put work between the nine table loads and their first use:
// pme_taps3_issue computes the rows and starts the loads (rows of terms that do not apply are clamped into the table), pme_taps3_finish drops
// the values of those terms and forms the derivative weight.
struct TapRaw { double v6[3], va[3], vb[3]; int mask; };      // mask: bit l = (a1 < 5) of axis l, bit 3 + l = (0 < a2)
__device__ __forceinline__ void pme_taps3_issue(double u1, double u2, double u3, int n1, int n2, int n3, int k, double G, const double* __restrict__ B6,
                                                const double* __restrict__ B5, TapRaw& r) {
  const double u[3] = {u1, u2, u3};
  const int np[3] = {n1, n2, n3};
  const double G6 = G / 6.0, G5 = G / 5.0;
  double a1[3], a2[3];
  int r6[3], ra[3], rb[3];
  bool rare = false;
#pragma unroll
  for (int l = 0; l < 3; ++l) {
    a1[l] = u[l] - (double)(np[l] - k); a2[l] = a1[l] - 1.0;
    const double t6 = a1[l] * G6, c6 = ceil(t6), ta = a1[l] * G5, ca = ceil(ta), tb = a2[l] * G5, cb = ceil(tb);
    const double g6 = c6 - t6, ga = ca - ta, gb = cb - tb;
    rare |= g6 < 1e-9 || g6 > 1.0 - 1e-9 || ga < 1e-9 || ga > 1.0 - 1e-9 || gb < 1e-9 || gb > 1.0 - 1e-9;
    r6[l] = (int)c6; ra[l] = (int)ca; rb[l] = (int)cb;
  }
  if (__builtin_expect(__any(rare), 0)) {
#pragma unroll
    for (int l = 0; l < 3; ++l) { r6[l] = (int)ceil(a1[l] / 6.0 * G); ra[l] = (int)ceil(a1[l] / 5.0 * G); rb[l] = (int)ceil(a2[l] / 5.0 * G); }
  }
  const int Gi = (int)G;
  r.mask = 0;
#pragma unroll
  for (int l = 0; l < 3; ++l) {
    r.v6[l] = B6[min(max(r6[l], 1), Gi) - 1]; r.va[l] = B5[min(max(ra[l], 1), Gi) - 1]; r.vb[l] = B5[min(max(rb[l], 1), Gi) - 1];
    r.mask |= (a1[l] < 5.0 ? 1 << l : 0) | (0.0 < a2[l] ? 8 << l : 0);
  }
}
// the same from the row seeds of k_pme_prep_light: integer adds per tap; a wavefront with a seed in a tie zone reads (u, nearpt) of its atoms (usrc, bsrc
// + 3 x the atom's position in the cell-sorted order) and takes the arithmetic above - rare, and the only global load of the function
__device__ __forceinline__ void pme_taps3_issue_rows(int2 s1, int2 s2, int2 s3, const RowTap& t, int k, int Gi, const double* __restrict__ B6, const double* __restrict__ B5,
                                                     const double* __restrict__ usrc, const int* __restrict__ bsrc, int at3, TapRaw& r) {
  if (__builtin_expect(__any((s1.x | s2.x | s3.x) < 0), 0)) {
    pme_taps3_issue(usrc[at3], usrc[at3 + 1], usrc[at3 + 2], bsrc[at3], bsrc[at3 + 1], bsrc[at3 + 2], k, (double)Gi, B6, B5, r);
    return;
  }
  const int2 sd[3] = {s1, s2, s3};
#pragma unroll
  for (int l = 0; l < 3; ++l) {
    r.v6[l] = B6[min(max(pme_row6(sd[l], t), 1), Gi) - 1]; r.va[l] = B5[min(max(pme_row5a(sd[l], t), 1), Gi) - 1]; r.vb[l] = B5[min(max(pme_row5b(sd[l], t), 1), Gi) - 1];
  }
  r.mask = (k < 5 ? 7 : 0) | (k >= 1 ? 56 : 0);      // a1 = f + k < 5 <=> k <= 4; 0 < a2 = f + k - 1 <=> k >= 1 (0 < f < 1: k_pme_prep_light keeps u off the integers)
}
__device__ __forceinline__ void pme_taps3_finish(const TapRaw& r, double (&w)[3], double (&dw)[3]) {
#pragma unroll
  for (int l = 0; l < 3; ++l) { w[l] = r.v6[l]; dw[l] = ((r.mask >> l) & 1 ? r.va[l] : 0.0) - ((r.mask >> (3 + l)) & 1 ? r.vb[l] : 0.0); }
}

// one thread per (atom, k3, k2); the 6 k1 taps are consecutive grid points (modulo wrap)
__global__ __launch_bounds__(256) void k_pme_spread(int N, int K, const double* __restrict__ chg, const int* __restrict__ base,
                                                    const double* __restrict__ w, double* __restrict__ grid, int gs) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)N * 36) return;
  const int i = (int)(t / 36), r = (int)(t % 36), k3 = r / 6, k2 = r % 6;
  const double q = chg[i];
  if (q == 0.0) return;
  int n3 = base[3 * i + 2] - k3; if (n3 < 0) n3 += K;
  int n2 = base[3 * i + 1] - k2; if (n2 < 0) n2 += K;
  const double w23 = w[(3 * i + 1) * 6 + k2], w3 = w[(3 * i + 2) * 6 + k3];
  const int b1 = base[3 * i];
  double* row = grid + (size_t)gs * K * (n2 + (size_t)K * n3);
  for (int k1 = 0; k1 < 6; ++k1) {
    int n1 = b1 - k1; if (n1 < 0) n1 += K;
    const double v = q * w[(3 * i) * 6 + k1] * w23 * w3;
    atomicAdd(&row[gs * n1], v);
  }
}


__host__ __device__ inline int spread_plane_stride(int3 T) { const int rows = (T.x | 1) * T.y; return rows + (6 - rows % 16 + 16) % 16; }
constexpr int SPREAD_CHUNK = 64;      // atoms of a cell staged in LDS at a time (10.5 KB)

// n = q d + r for 0 <= n < 2^23 and a divisor known only at run time: the quotient estimated with a single-precision reciprocal (off by at
// most one at these sizes) and corrected by the remainder - eight instructions where the compiler's 32-bit division takes ~35; the tile loops
// below did two to five of those per grid point
__device__ __forceinline__ void divmod_small(int n, int d, float inv_d, int& q, int& r) {
  q = (int)((float)n * inv_d); r = n - q * d;
  if (r < 0) { --q; r += d; }
  if (r >= d) { ++q; r -= d; }
}
// LDS-tiled spread: one workgroup per neighbour cell (atoms are cell-sorted).  The cell's atoms scatter
// their 216-point stencils into an LDS tile with ds_add_f64, the tile is flushed once with global fp64 atomics
// (non-zero points only): ~20x fewer HBM atomics than one per stencil point.  Atoms that drifted out of the tile since the
// last list build fall back to a direct global atomic.
__global__ __launch_bounds__(512) void k_pme_spread_tiles(int K, int3 nc, int3 T, const int* __restrict__ cell_start, const int* __restrict__ perm,
                                                          const double* __restrict__ chg, const int* __restrict__ base, const double* __restrict__ w,
                                                          double* __restrict__ grid, int gs,      // gs = 2: real parts of a complex grid, 1: real grid
                                                          const double* __restrict__ uscale, const double* __restrict__ B6, int G, const int2* __restrict__ rows_s) {      // w == nullptr: weights from the table (row seeds of k_pme_prep_light)
  extern __shared__ double tile[];
  const int cidx = blockIdx.x;
  const int cx = cidx % nc.x, cy = (cidx / nc.x) % nc.y, cz = cidx / (nc.x * nc.y);
  // tile origin: first grid point that can receive charge from an atom inside the cell, minus one point of drift margin
  const int ox = (int)floor((double)cx * K / nc.x) - 6, oy = (int)floor((double)cy * K / nc.y) - 6, oz = (int)floor((double)cz * K / nc.z) - 6;
  // row stride padded to an odd number of doubles: with T.x = 16 a row is exactly the 32 LDS banks wide and the 36 taps (k2,k3) of an
  // atom, which differ only in their row, all hit the same two banks (111 us stand-alone, bank-conflict bound)
  // ... and the plane stride to 6 (mod 16) doubles, which spreads the 6 x 6 taps of an atom evenly over the 16 double-wide bank pairs
  const int P = T.x | 1, P2 = spread_plane_stride(T);
  const int ntile = P2 * T.z;
#ifdef PME_GATHER_DIAG
  const unsigned long long sd0 = __builtin_amdgcn_s_memtime();
#endif
  for (int k = threadIdx.x; k < ntile; k += blockDim.x) tile[k] = 0.0;
  // per-atom inputs of a chunk of the cell's atoms are staged in LDS first: the scatter loop then has no global loads (it used to wait
  // for six dependent weight loads per item at < 3 waves per SIMD - latency bound, 90 us)
  double* sw = tile + ntile;                         // [SPREAD_CHUNK][18] weights (3 axes x 6 taps)
  double* sq = sw + SPREAD_CHUNK * 18;               // [SPREAD_CHUNK] charges
  int* sb = (int*)(sq + SPREAD_CHUNK);               // [SPREAD_CHUNK][3] base grid points
  const int s = cell_start[cidx], e = cell_start[cidx + 1];
  for (int c0 = s; c0 < e; c0 += SPREAD_CHUNK) {
    const int n = min(SPREAD_CHUNK, e - c0);
    __syncthreads();      // tile zeroed / previous chunk consumed
    if (w) for (int k = threadIdx.x; k < n * 18; k += blockDim.x) { const int a = k / 18, r = k - 18 * a; sw[k] = w[(size_t)perm[c0 + a] * 18 + r]; }
    else for (int k0 = 0; k0 < n * 18; k0 += blockDim.x) {      // (trip count uniform over the workgroup: the tie-zone test below is a wave vote)
      const int k = k0 + threadIdx.x, kk = min(k, n * 18 - 1);
      const int a = kk / 18, r = kk - 18 * a, l = r / 6, tap = r - 6 * l, i3 = 3 * (c0 + a) + l;      // (seeds / uscale / base in cell-sorted order here)
      const int2 sd = rows_s[i3];
      int row = tap * (G / 6) + (tap * (G % 6)) / 6 + 1 + (sd.x & 0xffff) + ((((sd.x >> 16) & 7) + (tap * (G % 6)) % 6) >= 6 ? 1 : 0);
      double wv;
      if (__builtin_expect(__any(sd.x < 0), 0)) wv = pme_tap_w(uscale[i3], base[i3], tap, (double)G, B6);
      else wv = B6[min(max(row, 1), G) - 1];
      if (k < n * 18) sw[k] = wv;
    }
    if (w) for (int k = threadIdx.x; k < n * 3; k += blockDim.x) { const int a = k / 3, r = k - 3 * a; sb[k] = base[3 * perm[c0 + a] + r]; }
    else for (int k = threadIdx.x; k < n * 3; k += blockDim.x) sb[k] = base[3 * c0 + k];
    for (int a = threadIdx.x; a < n; a += blockDim.x) sq[a] = chg[perm[c0 + a]];
    __syncthreads();
    for (int t = threadIdx.x; t < n * 36; t += blockDim.x) {
      const int a = t / 36, r = t % 36, k3 = r / 6, k2 = r % 6;
      const double q = sq[a];
      if (q == 0.0) continue;
      const int b1 = sb[3 * a], b2 = sb[3 * a + 1], b3 = sb[3 * a + 2];
      // tile-local coordinates: base - tap - origin lies in (-K, K + 12): one conditional add and one conditional subtract wrap it
      int l3 = b3 - k3 - oz; l3 += l3 < 0 ? K : 0; l3 -= l3 >= K ? K : 0;
      int l2 = b2 - k2 - oy; l2 += l2 < 0 ? K : 0; l2 -= l2 >= K ? K : 0;
      const bool in23 = l2 < T.y && l3 < T.z;
      double* trow = tile + P * l2 + P2 * l3;
      const double w2 = sw[18 * a + 6 + k2], w3 = sw[18 * a + 12 + k3];
      int l1 = b1 - ox; l1 += l1 < 0 ? K : 0; l1 -= l1 >= K ? K : 0;      // tap 0; the taps walk down with wrap-around
      for (int k1 = 0; k1 < 6; ++k1) {
        const double v = q * sw[18 * a + k1] * w2 * w3;
        if (in23 && l1 < T.x) atomicAdd(&trow[l1], v);
        else {      // the atom drifted out of the tile since the last list build: straight to the grid
          int n1 = b1 - k1; if (n1 < 0) n1 += K;
          int n2 = b2 - k2; if (n2 < 0) n2 += K;
          int n3 = b3 - k3; if (n3 < 0) n3 += K;
          atomicAdd(&grid[gs * (n1 + (size_t)K * (n2 + (size_t)K * n3))], v);
        }
        l1 = l1 == 0 ? K - 1 : l1 - 1;
      }
    }
  }
  __syncthreads();
#ifdef PME_GATHER_DIAG
  const unsigned long long sd1 = __builtin_amdgcn_s_memtime();
#endif
  const float inv_P2 = __builtin_amdgcn_rcpf((float)P2), inv_P = __builtin_amdgcn_rcpf((float)P);
  for (int k = threadIdx.x; k < ntile; k += blockDim.x) {
    const double v = tile[k];
    if (v != 0.0) {      // padding points stay zero
      int cc, r, b, a;
      divmod_small(k, P2, inv_P2, cc, r); divmod_small(r, P, inv_P, b, a);
      int n1 = ox + a; n1 += n1 < 0 ? K : 0; n1 -= n1 >= K ? K : 0;      // (origin + offset lies in (-K, 2 K): the tile is no wider than the grid)
      int n2 = oy + b; n2 += n2 < 0 ? K : 0; n2 -= n2 >= K ? K : 0;
      int n3 = oz + cc; n3 += n3 < 0 ? K : 0; n3 -= n3 >= K ? K : 0;
      atomicAdd(&grid[gs * (n1 + (size_t)K * (n2 + (size_t)K * n3))], v);
    }
  }
#ifdef PME_GATHER_DIAG
  const unsigned long long sd2 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (threadIdx.x == 0 && (blockIdx.x % 61) == 5)
    printf("spread diag block %d atoms %d: zero + atoms %llu clk (100 MHz), flush issue %llu, flush drain %llu\n", blockIdx.x, e - s, sd1 - sd0, sd2 - sd1, __builtin_amdgcn_s_memtime() - sd2);
#endif
}

// MFMA spread (north_star: "MFMA used only for the B-spline -> charge-grid spreading/gathering contraction").  One workgroup (4 waves) per
// neighbour cell; the cell's tile is 16 x 16 x 16 grid points (needs ceil(K / cells) + 8 <= 16 on every axis).  The charge of the tile is
// the contraction over the cell's atoms of three DENSE 16-entry weight vectors (6 non-zero B-spline taps each, placed at the atom's
// position inside the tile):
//     tile[z][y][x] = sum_a (q_a Wz_a[z] Wy_a[y]) * Wx_a[x]
// i.e. for every plane z a GEMM  D_z[y][x] = sum_a A_z[y][a] B[a][x]  with A_z[y][a] = q_a Wz_a[z] Wy_a[y], B[a][x] = Wx_a[x]:
// M = N = 16, K = atoms of the cell -> v_mfma_f64_16x16x4_f64, 4 atoms per instruction.  Operand maps (cdna_hip_programming.md):
// A[row = lane & 15][k = lane >> 4], B[k = lane >> 4][col = lane & 15], D[row = (lane >> 4) + 4 r][col = lane & 15], r = 0..3.
// Wave w owns the planes z = 4 w .. 4 w + 3 (16 accumulator doubles per lane); per 4 atoms a lane reads Wy, Wx once and q Wz four times
// from LDS, multiplies four times and issues four MFMAs.  The accumulators go straight to the grid with fp64 atomics, x fastest
// (16 consecutive doubles per quarter wave), zeros skipped.  Atoms whose stencil left the tile since the last list build are spread
// point by point as in k_pme_spread_tiles.  ~19x the multiply-adds of the direct stencil loop, but no LDS atomics and no bank conflicts.
typedef double mfma_d4 __attribute__((ext_vector_type(4)));
constexpr int SPREAD_MFMA_CHUNK = 192;      // atoms of a cell staged at a time (3 x 16 doubles each: 72 KB)
__global__ __launch_bounds__(256) void k_pme_spread_mfma(int K, int3 nc, const int* __restrict__ cell_start, const int* __restrict__ perm,
                                                         const double* __restrict__ chg, const int* __restrict__ base, const double* __restrict__ w,
                                                         double2* __restrict__ grid) {
  extern __shared__ double sm[];
  double* Wx = sm;                                   // [CHUNK][16]
  double* Wy = Wx + SPREAD_MFMA_CHUNK * 16;          // [CHUNK][16]
  double* Wz = Wy + SPREAD_MFMA_CHUNK * 16;          // [CHUNK][16], times the charge
  __shared__ int s_out[SPREAD_MFMA_CHUNK], s_nout;
  const int cidx = blockIdx.x;
  const int cx = cidx % nc.x, cy = (cidx / nc.x) % nc.y, cz = cidx / (nc.x * nc.y);
  const int ox = (int)floor((double)cx * K / nc.x) - 6, oy = (int)floor((double)cy * K / nc.y) - 6, oz = (int)floor((double)cz * K / nc.z) - 6;
  const int s = cell_start[cidx], e = cell_start[cidx + 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  mfma_d4 acc[4];
  for (int r = 0; r < 4; ++r) acc[r] = (mfma_d4){0.0, 0.0, 0.0, 0.0};
  for (int c0 = s; c0 < e; c0 += SPREAD_MFMA_CHUNK) {
    const int n = min(SPREAD_MFMA_CHUNK, e - c0), n4 = (n + 3) & ~3;
    __syncthreads();      // previous chunk consumed
    if (threadIdx.x == 0) s_nout = 0;
    for (int k = threadIdx.x; k < n4 * 48; k += blockDim.x) sm[(k / (n4 * 16)) * SPREAD_MFMA_CHUNK * 16 + k % (n4 * 16)] = 0.0;
    __syncthreads();
    for (int t = threadIdx.x; t < n * 3; t += blockDim.x) {      // one thread per (atom, axis): six taps into the dense vector
      const int a = t / 3, d = t - 3 * a, i = perm[c0 + a];
      const int b1 = base[3 * i], b2 = base[3 * i + 1], b3 = base[3 * i + 2];
      int l1 = b1 - ox; l1 += l1 < 0 ? K : 0; l1 -= l1 >= K ? K : 0;
      int l2 = b2 - oy; l2 += l2 < 0 ? K : 0; l2 -= l2 >= K ? K : 0;
      int l3 = b3 - oz; l3 += l3 < 0 ? K : 0; l3 -= l3 >= K ? K : 0;
      const double q = chg[i];
      const bool inside = l1 >= 5 && l1 < 16 && l2 >= 5 && l2 < 16 && l3 >= 5 && l3 < 16;
      if (!inside) { if (d == 0 && q != 0.0) s_out[atomicAdd(&s_nout, 1)] = i; continue; }      // its dense vectors stay zero
      const int l = d == 0 ? l1 : d == 1 ? l2 : l3;
      double* W = (d == 0 ? Wx : d == 1 ? Wy : Wz) + a * 16;
      const double sc = d == 2 ? q : 1.0;
      for (int k = 0; k < 6; ++k) W[l - k] = sc * w[(size_t)i * 18 + d * 6 + k];
    }
    __syncthreads();
    const int col = lane & 15, kq = lane >> 4;
    for (int a0 = 0; a0 < n4; a0 += 4) {
      const int a = a0 + kq;
      const double bx = Wx[a * 16 + col], ay = Wy[a * 16 + col];
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = __builtin_amdgcn_mfma_f64_16x16x4f64(Wz[a * 16 + 4 * wave + r] * ay, bx, acc[r], 0, 0, 0);
    }
    __syncthreads();
    for (int o = 0; o < s_nout; ++o) {      // atoms that drifted out of the tile: straight to the grid, 216 taps over the block
      const int i = s_out[o];
      const double q = chg[i];
      for (int t = threadIdx.x; t < 216; t += blockDim.x) {
        const int k1 = t % 6, k2 = (t / 6) % 6, k3 = t / 36;
        int n1 = base[3 * i] - k1; if (n1 < 0) n1 += K;
        int n2 = base[3 * i + 1] - k2; if (n2 < 0) n2 += K;
        int n3 = base[3 * i + 2] - k3; if (n3 < 0) n3 += K;
        atomicAdd(&grid[n1 + (size_t)K * (n2 + (size_t)K * n3)].x, q * w[(size_t)i * 18 + k1] * w[(size_t)i * 18 + 6 + k2] * w[(size_t)i * 18 + 12 + k3]);
      }
    }
  }
  // D[row = y = (lane >> 4) + 4 v][col = x = lane & 15] of plane z = 4 wave + r
  int n1 = ox + (lane & 15); n1 += n1 < 0 ? K : 0; n1 -= n1 >= K ? K : 0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    int n3 = oz + 4 * wave + r; n3 += n3 < 0 ? K : 0; n3 -= n3 >= K ? K : 0;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const double val = acc[r][v];
      if (val != 0.0) {
        int n2 = oy + (lane >> 4) + 4 * v; n2 += n2 < 0 ? K : 0; n2 -= n2 >= K ? K : 0;
        atomicAdd(&grid[n1 + (size_t)K * (n2 + (size_t)K * n3)].x, val);
      }
    }
  }
}

// OWNER-COMPUTES spread on the matrix cores: one workgroup per neighbour cell, which owns the BRICK of grid points
// [floor(c K / n), floor((c + 1) K / n)) per axis (<= 8 points) - the bricks partition the grid, so every grid point is WRITTEN once
// (plain stores of {Q, 0} into the complex grid and of Q into the real copy): no global atomics (the halo flush of the per-cell tiles
// was ~0.4 M atomic L2 requests per launch, the whole duration of both tile kernels), no zero-fill, no copy pass.
//   1. scan the cell's neighbourhood (<= 27 cells, cell-sorted base points: coalesced) and keep the atoms whose 6-tap stencil reaches
//      into the brick on every axis (~5 x the cell's own atoms);
//   2. per chunk of kept atoms, three dense 8-entry weight vectors per atom (its taps inside the brick, zeros elsewhere), then
//      brick[z][y][x] = sum_a (q_a Wz_a[z] Wy_a[y]) Wx_a[x] as v_mfma_f64_16x16x4_f64: rows = 2 planes x 8 y, columns = x (8 of 16 used),
//      4 atoms per instruction; wave w owns the planes 2 w, 2 w + 1.
constexpr int BRICK_SEL = 2048;       // most atoms a brick can take (a liquid gives ~700)
constexpr int BRICK_CHUNK = 192;
__global__ __launch_bounds__(256) void k_pme_spread_bricks(int K, int3 nc, const int* __restrict__ cell_start, const int* __restrict__ base_s, const double* __restrict__ w_s,
                                                           const double4* __restrict__ posq, double2* __restrict__ grid, double* __restrict__ Q, double* __restrict__ overflow) {
  __shared__ int s_lo[27], s_n[27], s_ns, s_cnt[4];
  __shared__ int sel_k[BRICK_SEL];      // four wave-private lists of BRICK_SEL / 4 entries (no shared counter: a wave counts in a register)
  __shared__ unsigned sel_l[BRICK_SEL];
  extern __shared__ double Wd[];      // [3][BRICK_CHUNK][8]
  double* Wx = Wd; double* Wy = Wx + BRICK_CHUNK * 8; double* Wz = Wy + BRICK_CHUNK * 8;
  const int cidx = blockIdx.x;
  const int cx = cidx % nc.x, cy = (cidx / nc.x) % nc.y, cz = cidx / (nc.x * nc.y);
  const int p0x = (int)floor((double)cx * K / nc.x), p0y = (int)floor((double)cy * K / nc.y), p0z = (int)floor((double)cz * K / nc.z);
  const int nbx = (cx + 1 == nc.x ? K : (int)floor((double)(cx + 1) * K / nc.x)) - p0x, nby = (cy + 1 == nc.y ? K : (int)floor((double)(cy + 1) * K / nc.y)) - p0y,
            nbz = (cz + 1 == nc.z ? K : (int)floor((double)(cz + 1) * K / nc.z)) - p0z;
  if (threadIdx.x < 27) {      // one lane per neighbour cell (a single thread walked 54 dependent loads)
    int slots[27];
    const int ns = cell_neighbourhood(cidx, nc.x, nc.y, nc.z, slots);
    if ((int)threadIdx.x < ns) { const int lo = cell_start[slots[threadIdx.x]]; s_lo[threadIdx.x] = lo; s_n[threadIdx.x] = cell_start[slots[threadIdx.x] + 1] - lo; }
    if (threadIdx.x == 0) s_ns = ns;
  }
  __syncthreads();
  const int ns = s_ns;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // 1. selection: every wave takes whole neighbour cells (no search for the cell of a candidate); the base points of a cell's atoms are
  //    consecutive int triples, and the next trip's are loaded before the current trip is compacted
  int wcount = 0;      // wave-uniform
  const int wlist = wave * (BRICK_SEL / 4);
  for (int sl = wave; sl < ns; sl += 4) {
    const int lo = s_lo[sl], n = s_n[sl];
    int ks = lo + min(lane, n - 1);
    int b1 = base_s[3 * ks], b2 = base_s[3 * ks + 1], b3 = base_s[3 * ks + 2];
    for (int q0 = 0; q0 < n; q0 += 64) {
      const int q = q0 + lane, ksn = lo + min(q + 64, n - 1);
      const int c1 = base_s[3 * ksn], c2 = base_s[3 * ksn + 1], c3 = base_s[3 * ksn + 2];      // next trip (the last one reloads the cell's last atom)
      int l1 = b1 - p0x; l1 += l1 < 0 ? K : 0;
      int l2 = b2 - p0y; l2 += l2 < 0 ? K : 0;
      int l3 = b3 - p0z; l3 += l3 < 0 ? K : 0;
      const bool keep = q < n && l1 < nbx + 5 && l2 < nby + 5 && l3 < nbz + 5;      // a tap of every axis lands inside the brick
      const unsigned long long m = __ballot(keep);
      if (keep) {
        const int slot = wcount + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        if (slot < BRICK_SEL / 4) { sel_k[wlist + slot] = ks; sel_l[wlist + slot] = (unsigned)l1 | ((unsigned)l2 << 8) | ((unsigned)l3 << 16); }
      }
      wcount += (int)__popcll(m);
      ks = ksn; b1 = c1; b2 = c2; b3 = c3;
    }
  }
  if (lane == 0) s_cnt[wave] = wcount;
  __syncthreads();
  const int n0 = s_cnt[0], n1 = s_cnt[1], n2 = s_cnt[2], n3 = s_cnt[3];
  if (max(max(n0, n1), max(n2, n3)) > BRICK_SEL / 4) { if (threadIdx.x == 0) *overflow = 1.0; return; }      // sticky energy slot S_BRICK: the host switches to the tile kernel and repeats the force call
  const int nsel = n0 + n1 + n2 + n3;
  // position g of the concatenated lists -> entry of a wave's list
  auto sel_at = [&](int g) { return g < n0 ? g : g < n0 + n1 ? BRICK_SEL / 4 + (g - n0) : g < n0 + n1 + n2 ? 2 * (BRICK_SEL / 4) + (g - n0 - n1) : 3 * (BRICK_SEL / 4) + (g - n0 - n1 - n2); };
  mfma_d4 acc = (mfma_d4){0.0, 0.0, 0.0, 0.0}, acc2 = (mfma_d4){0.0, 0.0, 0.0, 0.0};      // two accumulation chains: back-to-back MFMAs into one accumulator wait for each other
  for (int c0 = 0; c0 < nsel; c0 += BRICK_CHUNK) {      // 2. contraction, BRICK_CHUNK atoms at a time
    const int n = min(BRICK_CHUNK, nsel - c0), n4 = (n + 7) & ~7;
    __syncthreads();      // previous chunk consumed
    for (int t = threadIdx.x; t < n4 * 3; t += blockDim.x) {      // one thread per (atom, axis): the taps inside the brick, zeros elsewhere
      const int a = t / 3, d = t - 3 * a;
      double* W = (d == 0 ? Wx : d == 1 ? Wy : Wz) + a * 8;
      if (a < n) {
        const int se = sel_at(c0 + a);
        const int ks = sel_k[se];
        const int l = (int)((sel_l[se] >> (8 * d)) & 255u);
        const double sc = d == 2 ? posq[ks].w : 1.0;
        const double* wp = w_s + 18 * (size_t)ks + 6 * d;
        double w6[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) w6[k] = wp[k];
#pragma unroll
        for (int q = 0; q < 8; ++q) {      // brick point q holds tap k = l - q (static register indexing: a select chain, no scratch)
          const int k = l - q;
          double v = 0.0;
#pragma unroll
          for (int kk = 0; kk < 6; ++kk) v = k == kk ? w6[kk] : v;
          W[q] = sc * v;
        }
      } else {
#pragma unroll
        for (int q = 0; q < 8; ++q) W[q] = 0.0;
      }
    }
    __syncthreads();
    const int row = lane & 15, kq = lane >> 4, zp = row >> 3, y = row & 7;
    for (int a0 = 0; a0 < n4; a0 += 8) {
      const int a = a0 + kq, b = a + 4;
      const double av = Wz[a * 8 + 2 * wave + zp] * Wy[a * 8 + y], av2 = Wz[b * 8 + 2 * wave + zp] * Wy[b * 8 + y];
      const double bv = row < 8 ? Wx[a * 8 + row] : 0.0, bv2 = row < 8 ? Wx[b * 8 + row] : 0.0;      // (row doubles as the column index of the B operand: lane & 15)
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(av2, bv2, acc2, 0, 0, 0);
    }
  }
  acc += acc2;
  // D[row = (lane >> 4) + 4 v][col = lane & 15]: row = plane (row >> 3) of this wave, y = row & 7; col = x
  const int x = lane & 15;
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const int r = (lane >> 4) + 4 * v, z = 2 * wave + (r >> 3), y = r & 7;
    if (x < nbx && y < nby && z < nbz) {
      const size_t idx = (size_t)(p0x + x) + (size_t)K * ((size_t)(p0y + y) + (size_t)K * (size_t)(p0z + z));
      grid[idx] = make_double2(acc[v], 0.0);
      Q[idx] = acc[v];
    }
  }
}

// blockIdx.y = grid (batch) index; energy partials of grid g go to part[g*part_stride + blockIdx.x]
__global__ __launch_bounds__(256) void k_pme_conv(size_t K3, const double* __restrict__ CB, double2* __restrict__ grids, int part_stride,
                                                  double* __restrict__ part, double half_conv) {
  __shared__ double red[256];
  double e = 0.0;
  double2* grid = grids + (size_t)blockIdx.y * K3;
  part += (size_t)blockIdx.y * part_stride;
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < K3; k += (size_t)gridDim.x * blockDim.x) {
    double2 v = grid[k];
    const double cb = CB[k];
    e += cb * (v.x * v.x + v.y * v.y);
    v.x *= cb; v.y *= cb;
    grid[k] = v;
  }
  red[threadIdx.x] = e;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) { if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st]; __syncthreads(); }
  if (threadIdx.x == 0) part[blockIdx.x] = red[0] * half_conv;
}

__global__ void k_copy_real(size_t K3, const double2* __restrict__ grid, double* __restrict__ Q) {
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < K3; k += (size_t)gridDim.x * blockDim.x) Q[k] = grid[k].x;
}

__global__ void k_real_to_complex(size_t K3, const double* __restrict__ in, double2* __restrict__ out) {
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < K3; k += (size_t)gridDim.x * blockDim.x) out[k] = make_double2(in[k], 0.0);
}

// real part of the convolved grid, contiguous (the gathers read 6 consecutive doubles per stencil row)
__global__ void k_pme_real(size_t K3, const double2* __restrict__ grid, double* __restrict__ theta) {
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < K3; k += (size_t)gridDim.x * blockDim.x) theta[k] = grid[k].x;
}

// derivative_grid_Q: 8 lanes per atom (8 atoms per wavefront); lane s < 6 owns the k3 = s plane of the stencil (36 taps, 6 rows
// of 6 consecutive grid points), 3-step shuffle reduction inside the 8-lane group, F = -K kk^T Ftilde.
__global__ __launch_bounds__(256) void k_pme_gather(int N, int K, const double* __restrict__ chg, const int* __restrict__ base,
                                                    const double* __restrict__ w, const double* __restrict__ dw, const double* __restrict__ theta,
                                                    double conv, double k00, double k01, double k02, double k10, double k11, double k12,
                                                    double k20, double k21, double k22, double* __restrict__ F_recip, double* __restrict__ F, int atomic, int add_lo, int add_hi) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = t >> 3, sub = t & 7;
  double f1 = 0.0, f2 = 0.0, f3 = 0.0;
  if (i < N && sub < 6) {
    const int b1 = base[3 * i], b2 = base[3 * i + 1];
    int n3 = base[3 * i + 2] - sub; if (n3 < 0) n3 += K;
    const double* wi = w + (size_t)i * 18;
    const double* dwi = dw + (size_t)i * 18;
    double w1[6], d1[6]; int n1[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) { w1[k] = wi[k]; d1[k] = dwi[k]; int n = b1 - k; if (n < 0) n += K; n1[k] = n; }
    const double w3 = wi[12 + sub], d3 = dwi[12 + sub];
#pragma unroll
    for (int k2 = 0; k2 < 6; ++k2) {
      int n2 = b2 - k2; if (n2 < 0) n2 += K;
      const double* row = theta + (size_t)K * (n2 + (size_t)K * n3);
      const double w2 = wi[6 + k2], d2 = dwi[6 + k2];
      double s0 = 0.0, s1 = 0.0;    // sum_k1 w1*theta, sum_k1 d1*theta
#pragma unroll
      for (int k = 0; k < 6; ++k) { const double th = row[n1[k]]; s0 += w1[k] * th; s1 += d1[k] * th; }
      f1 += s1 * w2 * w3; f2 += s0 * d2 * w3; f3 += s0 * w2 * d3;
    }
  }
#pragma unroll
  for (int o = 4; o > 0; o >>= 1) { f1 += __shfl_xor(f1, o, 64); f2 += __shfl_xor(f2, o, 64); f3 += __shfl_xor(f3, o, 64); }
  if (i < N && sub == 0) {
    const double s = chg[i] * conv;
    f1 *= s; f2 *= s; f3 *= s;
    const double Kd = (double)K;
    const double fx = -Kd * (k00 * f1 + k10 * f2 + k20 * f3);
    const double fy = -Kd * (k01 * f1 + k11 * f2 + k21 * f3);
    const double fz = -Kd * (k02 * f1 + k12 * f2 + k22 * f3);
    F_recip[3 * i] = fx; F_recip[3 * i + 1] = fy; F_recip[3 * i + 2] = fz;
    if (F && i >= add_lo && i < add_hi) {      // the total force takes this atom only on the rank that owns it
      if (atomic) { atomicAdd(&F[3 * i], fx); atomicAdd(&F[3 * i + 1], fy); atomicAdd(&F[3 * i + 2], fz); }     // beside the pair loop
      else { F[3 * i] += fx; F[3 * i + 1] += fy; F[3 * i + 2] += fz; }
    }
  }
}

// LDS-tiled gather, the counterpart of k_pme_spread_tiles: one workgroup per neighbour cell stages the potential of the cell's tile
// (same origin, size and bank-conflict padding) once, the 8-lane groups then take the cell's atoms round-robin and read their 216 taps
// from LDS instead of gathering them from L2.  Same loops and summation order per atom as k_pme_gather (bit-identical forces); an atom
// whose stencil left the tile since the last list build is done from global memory by the same code.
__global__ __launch_bounds__(512) void k_pme_gather_tiles(int K, int3 nc, int3 T, const int* __restrict__ cell_start, const int* __restrict__ perm,
                                                          const double* __restrict__ chg, const int* __restrict__ base, const double* __restrict__ w,
                                                          const double* __restrict__ dw, const double* __restrict__ theta, double conv, double k00, double k01,
                                                          double k02, double k10, double k11, double k12, double k20, double k21, double k22,
                                                          double* __restrict__ F_recip, double* __restrict__ F, int atomic, int add_lo, int add_hi) {
  extern __shared__ double tile[];
  const int cidx = blockIdx.x;
  const int cx = cidx % nc.x, cy = (cidx / nc.x) % nc.y, cz = cidx / (nc.x * nc.y);
  const int ox = (int)floor((double)cx * K / nc.x) - 6, oy = (int)floor((double)cy * K / nc.y) - 6, oz = (int)floor((double)cz * K / nc.z) - 6;
  const int P = T.x | 1, P2 = spread_plane_stride(T);
  {      // stage: x fastest (coalesced rows of T.x points); four loads in flight per thread (one per trip left the fill latency-bound: 16 dependent trips)
    const int ntile = T.x * T.y * T.z, txy = T.x * T.y;
    const float inv_txy = __builtin_amdgcn_rcpf((float)txy), inv_tx = __builtin_amdgcn_rcpf((float)T.x);
    for (int r0 = threadIdx.x; r0 < ntile; r0 += 4 * blockDim.x) {
      double v[4]; int dst[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int r = r0 + u * blockDim.x;
        dst[u] = -1;
        if (r < ntile) {
          int cc, rem, b, a;
          divmod_small(r, txy, inv_txy, cc, rem); divmod_small(rem, T.x, inv_tx, b, a);
          int n1 = ox + a; n1 += n1 < 0 ? K : 0; n1 -= n1 >= K ? K : 0;
          int n2 = oy + b; n2 += n2 < 0 ? K : 0; n2 -= n2 >= K ? K : 0;
          int n3 = oz + cc; n3 += n3 < 0 ? K : 0; n3 -= n3 >= K ? K : 0;
          v[u] = theta[n1 + (size_t)K * (n2 + (size_t)K * n3)];
          dst[u] = a + P * b + P2 * cc;
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) if (dst[u] >= 0) tile[dst[u]] = v[u];
    }
  }
  __syncthreads();
  const int s = cell_start[cidx], e = cell_start[cidx + 1];
  const int sub = threadIdx.x & 7, grp = threadIdx.x >> 3, ngrp = blockDim.x >> 3;
  for (int a0 = s; a0 < e; a0 += ngrp) {      // uniform trip count per wave (shuffles below)
    const int ia = a0 + grp;
    const bool live = ia < e;
    const int i = live ? perm[ia] : 0;
    double f1 = 0.0, f2 = 0.0, f3 = 0.0;
    if (live && sub < 6) {
      const int b1 = base[3 * i], b2 = base[3 * i + 1], b3 = base[3 * i + 2];
      // tile-local coordinate of tap 0 per axis; the 6 taps walk down: all inside the tile <=> 5 <= l < T
      int l1 = b1 - ox; l1 += l1 < 0 ? K : 0; l1 -= l1 >= K ? K : 0;
      int l2 = b2 - oy; l2 += l2 < 0 ? K : 0; l2 -= l2 >= K ? K : 0;
      int l3 = b3 - oz; l3 += l3 < 0 ? K : 0; l3 -= l3 >= K ? K : 0;
      const bool in_tile = l1 >= 5 && l1 < T.x && l2 >= 5 && l2 < T.y && l3 >= 5 && l3 < T.z;
      const double* wi = w + (size_t)i * 18;
      const double* dwi = dw + (size_t)i * 18;
      double w1[6], d1[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) { w1[k] = wi[k]; d1[k] = dwi[k]; }
      const double w3 = wi[12 + sub], d3 = dwi[12 + sub];
      if (in_tile) {
        const double* plane = tile + P2 * (l3 - sub) + l1;
#pragma unroll
        for (int k2 = 0; k2 < 6; ++k2) {
          const double* row = plane + P * (l2 - k2);
          const double w2 = wi[6 + k2], d2 = dwi[6 + k2];
          double s0 = 0.0, s1 = 0.0;
#pragma unroll
          for (int k = 0; k < 6; ++k) { const double th = row[-k]; s0 += w1[k] * th; s1 += d1[k] * th; }
          f1 += s1 * w2 * w3; f2 += s0 * d2 * w3; f3 += s0 * w2 * d3;
        }
      } else {
        int n3 = b3 - sub; if (n3 < 0) n3 += K;
        int n1[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) { int n = b1 - k; if (n < 0) n += K; n1[k] = n; }
#pragma unroll
        for (int k2 = 0; k2 < 6; ++k2) {
          int n2 = b2 - k2; if (n2 < 0) n2 += K;
          const double* row = theta + (size_t)K * (n2 + (size_t)K * n3);
          const double w2 = wi[6 + k2], d2 = dwi[6 + k2];
          double s0 = 0.0, s1 = 0.0;
#pragma unroll
          for (int k = 0; k < 6; ++k) { const double th = row[n1[k]]; s0 += w1[k] * th; s1 += d1[k] * th; }
          f1 += s1 * w2 * w3; f2 += s0 * d2 * w3; f3 += s0 * w2 * d3;
        }
      }
    }
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) { f1 += __shfl_xor(f1, o, 64); f2 += __shfl_xor(f2, o, 64); f3 += __shfl_xor(f3, o, 64); }
    if (live && sub == 0) {
      const double sc = chg[i] * conv;
      f1 *= sc; f2 *= sc; f3 *= sc;
      const double Kd = (double)K;
      const double fx = -Kd * (k00 * f1 + k10 * f2 + k20 * f3);
      const double fy = -Kd * (k01 * f1 + k11 * f2 + k21 * f3);
      const double fz = -Kd * (k02 * f1 + k12 * f2 + k22 * f3);
      F_recip[3 * i] = fx; F_recip[3 * i + 1] = fy; F_recip[3 * i + 2] = fz;
      if (F && i >= add_lo && i < add_hi) {
        if (atomic) { atomicAdd(&F[3 * i], fx); atomicAdd(&F[3 * i + 1], fy); atomicAdd(&F[3 * i + 2], fz); }
        else { F[3 * i] += fx; F[3 * i + 1] += fy; F[3 * i + 2] += fz; }
      }
    }
  }
}

// The same gather WITHOUT the weight store (k_pme_prep_light: uscale / base in cell-sorted order).  Lane `sub` of an atom's 8-lane group takes
// tap `sub` of the three axes from the spline tables (pme_taps3: the bits k_pme_prep stores); the group exchanges the x and y
// values through 192 B of LDS of its own (the group sits inside one wavefront: in-order LDS, no workgroup barrier), the z tap stays with its
// lane.  The table loads of an atom go out one trip ahead: trip 0's before the tile fill, from global (u, nearpt); the later ones before the
// stencil sums of the trip in front of them, from the cell's (u, nearpt) staged in LDS beside the tile - no global load waits inside the loop.
// Same sums in the same order as k_pme_gather_tiles: bit-identical forces.
// FOLD (MS-EVB fast path): the pair loop's force accumulator in cell-sorted order (d_fsorted, [3][N]) joins the atom's reciprocal force in the ONE
// atomic add this kernel makes per component anyway and is re-zeroed - the atoms are walked in that order here, so the read is coalesced; with the
// blend terms added from a side stream (k_evb_blend_terms, evb.hip) no k_evb_blend pass over the force array is left on the step's critical path.
// (Round 4 also tried the whole blend in here: 68 us against 40 + 15 - the force STORES are in atom order, scattered 24-byte reads from this kernel.)
template <bool FOLD>
__global__ __launch_bounds__(512, 2) void k_pme_gather_tiles_wl(int K, int3 nc, int3 T, const int* __restrict__ cell_start, const int* __restrict__ perm,
                                                             const double* __restrict__ chg, const int* __restrict__ base_s, const double* __restrict__ uscale_s, const int2* __restrict__ rows_s,
                                                             const double* __restrict__ B6, const double* __restrict__ B5, int G, int cap,
                                                             const double* __restrict__ theta, double conv, double k00, double k01,
                                                             double k02, double k10, double k11, double k12, double k20, double k21, double k22,
                                                             double* __restrict__ F_recip, double* __restrict__ F, int atomic, int add_lo, int add_hi, double* __restrict__ Fs, int NFs) {
  extern __shared__ double tile[];
#ifdef PME_GATHER_DIAG
  const unsigned long long gd0 = __builtin_amdgcn_s_memtime();
#endif
  const int cidx = blockIdx.x;
  const int cx = cidx % nc.x, cy = (cidx / nc.x) % nc.y, cz = cidx / (nc.x * nc.y);
  const int ox = (int)floor((double)cx * K / nc.x) - 6, oy = (int)floor((double)cy * K / nc.y) - 6, oz = (int)floor((double)cz * K / nc.z) - 6;
  const int P = T.x | 1, P2 = spread_plane_stride(T);
  const int sub = threadIdx.x & 7, grp = threadIdx.x >> 3, ngrp = blockDim.x >> 3;
  double* const xw = tile + P2 * T.z + 24 * grp;      // {w_x[6], w_y[6], dw_x[6], dw_y[6]} of the group's atom
  int2* const cu = (int2*)(tile + P2 * T.z + 24 * ngrp);     // [3 cap] row seeds of the cell's atoms (pme_row_seed) ...
  int* const cb = (int*)(cu + 3 * cap);                // [3 cap] ... their nearest grid points,
  double* const cq = (double*)(cb + 3 * cap + (cap & 1));      // [cap] charges
  int* const ci = (int*)(cq + cap);                    // [cap] and atom indices
  const int s = cell_start[cidx], e = cell_start[cidx + 1];
  const RowTap rt = pme_row_tap(sub < 6 ? sub : 0, G);      // this lane's tap: constants of its table rows
  // the atom one trip ahead (registers): raw table values of tap `sub` of the three axes, nearest grid points, atom, charge.  Lanes without an
  // atom (sub >= 6, no atom left) look up a harmless coordinate: the loads go out unconditionally, no branch between them and their first use
  TapRaw nt;
  double nq = 0.0;
  int nb1 = 0, nb2 = 0, nb3 = 0, ni = 0;
  // Load order of the prologue (the memory counter retires in order): trip 0's (u, nearpt) first, then the cell's staging values into registers
  // (thread k takes atom k), then - as soon as trip 0's coordinates are there - its nine table loads and the charges, then the tile; the LDS
  // stores of the staging values come behind the tile fill, so one wait covers everything that has been in flight beside it.
  const int n_cell = e - s;
  const bool staged = n_cell <= cap && n_cell <= (int)blockDim.x;      // (a cell larger than the host knew: the later trips read global memory as trip 0 does)
  const int iac = s + grp < e ? s + grp : max(e - 1, 0);      // (no atom for this group: any valid one, never used)
  nb1 = base_s[3 * iac]; nb2 = base_s[3 * iac + 1]; nb3 = base_s[3 * iac + 2];
  const int2 tr1 = rows_s[3 * iac], tr2 = rows_s[3 * iac + 1], tr3 = rows_s[3 * iac + 2];
  ni = perm[iac];
  const bool st_on = staged && (int)threadIdx.x < n_cell;
  const int ks = st_on ? s + (int)threadIdx.x : iac;
  const int2 su1 = rows_s[3 * ks], su2 = rows_s[3 * ks + 1], su3 = rows_s[3 * ks + 2];
  const int sb1 = base_s[3 * ks], sb2 = base_s[3 * ks + 1], sb3 = base_s[3 * ks + 2], si = perm[ks];
  // the tile's loads go out BEHIND the loads above and IN FRONT of the loads that wait for them (table rows, charges): the counter retires in order, so
  // the wait for trip 0's coordinates leaves the tile loads in flight - one round of memory latency for both (they were two: 2-5 + 4-9 us per workgroup, measured)
  const int ntile = T.x * T.y * T.z, txy = T.x * T.y;
  const float inv_txy = __builtin_amdgcn_rcpf((float)txy), inv_tx = __builtin_amdgcn_rcpf((float)T.x);
  auto tile_index = [&](int r, size_t& src, int& dst) {      // point r of the tile (x fastest: coalesced rows of T.x points) -> grid point, LDS slot
    int cc, rem, b, a;
    divmod_small(r, txy, inv_txy, cc, rem); divmod_small(rem, T.x, inv_tx, b, a);
    int n1 = ox + a; n1 += n1 < 0 ? K : 0; n1 -= n1 >= K ? K : 0;
    int n2 = oy + b; n2 += n2 < 0 ? K : 0; n2 -= n2 >= K ? K : 0;
    int n3 = oz + cc; n3 += n3 < 0 ? K : 0; n3 -= n3 >= K ? K : 0;
    src = n1 + (size_t)K * (n2 + (size_t)K * n3);
    dst = a + P * b + P2 * cc;
  };
  constexpr int FILL = 13;      // loads in flight per thread (a 16^3 tile is 8 per thread at 512 threads, 12.8 at 320)
  double v0[FILL];
#pragma unroll
  for (int u = 0; u < FILL; ++u) {
    const int r = (int)threadIdx.x + u * (int)blockDim.x;
    v0[u] = 0.0;
    if (r < ntile) { size_t src; int dst; tile_index(r, src, dst); v0[u] = theta[src]; }
  }
  pme_taps3_issue_rows(tr1, tr2, tr3, rt, sub < 6 ? sub : 0, G, B6, B5, uscale_s, base_s, 3 * iac, nt);
  nq = chg[ni];
  const double sq = chg[si];
#ifdef PME_GATHER_DIAG
  const unsigned long long gd1 = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll
  for (int u = 0; u < FILL; ++u) {
    const int r = (int)threadIdx.x + u * (int)blockDim.x;
    if (r < ntile) { size_t src; int dst; tile_index(r, src, dst); tile[dst] = v0[u]; }
  }
  for (int r0 = (int)threadIdx.x + FILL * (int)blockDim.x; r0 < ntile; r0 += FILL * blockDim.x) {      // (larger tiles: the rest, FILL at a time)
    double v[FILL]; int dst[FILL];
#pragma unroll
    for (int u = 0; u < FILL; ++u) {
      const int r = r0 + u * blockDim.x;
      dst[u] = -1;
      if (r < ntile) { size_t src; tile_index(r, src, dst[u]); v[u] = theta[src]; }
    }
#pragma unroll
    for (int u = 0; u < FILL; ++u) if (dst[u] >= 0) tile[dst[u]] = v[u];
  }
  if (st_on) {
    const int k = threadIdx.x;
    cu[3 * k] = su1; cu[3 * k + 1] = su2; cu[3 * k + 2] = su3; cb[3 * k] = sb1; cb[3 * k + 1] = sb2; cb[3 * k + 2] = sb3; ci[k] = si; cq[k] = sq;
  }
  __syncthreads();
#ifdef PME_GATHER_DIAG
  const unsigned long long gd2 = __builtin_amdgcn_s_memtime();
#endif
  for (int a0 = s; a0 < e; a0 += ngrp) {      // uniform trip count per wave (shuffles below)
    const int ia = a0 + grp;
    const bool live = ia < e;
    const int i = ni, b1 = nb1, b2 = nb2, b3 = nb3;
    const double q = nq;
    double tw[3], td[3];
    pme_taps3_finish(nt, tw, td);      // (first use of the table values that went out one trip ago)
    const double w3 = tw[2], d3 = td[2];
    if (live && sub < 6) { xw[sub] = tw[0]; xw[6 + sub] = tw[1]; xw[12 + sub] = td[0]; xw[18 + sub] = td[1]; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    double w1[6], d1[6];
    if (live && sub < 6) {
#pragma unroll
      for (int k = 0; k < 6; ++k) { w1[k] = xw[k]; d1[k] = xw[12 + k]; }
    }
    {      // the next trip's atom (the last atom of the cell again when there is none): table loads in flight during this trip's sums
      const int jc = min(ia + ngrp, e - 1) - s;
      int2 u1, u2, u3;
      if (staged) { ni = ci[jc]; nq = cq[jc]; nb1 = cb[3 * jc]; nb2 = cb[3 * jc + 1]; nb3 = cb[3 * jc + 2]; u1 = cu[3 * jc]; u2 = cu[3 * jc + 1]; u3 = cu[3 * jc + 2]; }
      else {
        const int jg = s + jc;
        ni = perm[jg]; nq = chg[ni]; nb1 = base_s[3 * jg]; nb2 = base_s[3 * jg + 1]; nb3 = base_s[3 * jg + 2]; u1 = rows_s[3 * jg]; u2 = rows_s[3 * jg + 1]; u3 = rows_s[3 * jg + 2];
      }
      pme_taps3_issue_rows(u1, u2, u3, rt, sub < 6 ? sub : 0, G, B6, B5, uscale_s, base_s, 3 * (s + jc), nt);
    }
    double f1 = 0.0, f2 = 0.0, f3 = 0.0;
    if (live && sub < 6) {
      // tile-local coordinate of tap 0 per axis; the 6 taps walk down: all inside the tile <=> 5 <= l < T
      int l1 = b1 - ox; l1 += l1 < 0 ? K : 0; l1 -= l1 >= K ? K : 0;
      int l2 = b2 - oy; l2 += l2 < 0 ? K : 0; l2 -= l2 >= K ? K : 0;
      int l3 = b3 - oz; l3 += l3 < 0 ? K : 0; l3 -= l3 >= K ? K : 0;
      const bool in_tile = l1 >= 5 && l1 < T.x && l2 >= 5 && l2 < T.y && l3 >= 5 && l3 < T.z;
      if (in_tile) {
        const double* plane = tile + P2 * (l3 - sub) + l1;
#pragma unroll
        for (int k2 = 0; k2 < 6; ++k2) {
          const double* row = plane + P * (l2 - k2);
          const double w2 = xw[6 + k2], d2 = xw[18 + k2];
          double s0 = 0.0, s1 = 0.0;
#pragma unroll
          for (int k = 0; k < 6; ++k) { const double th = row[-k]; s0 += w1[k] * th; s1 += d1[k] * th; }
          f1 += s1 * w2 * w3; f2 += s0 * d2 * w3; f3 += s0 * w2 * d3;
        }
      } else {      // the stencil left the tile since the last list build: from global memory (rare: loops kept rolled, registers belong to the path above)
        int n3 = b3 - sub; if (n3 < 0) n3 += K;
#pragma unroll 1
        for (int k2 = 0; k2 < 6; ++k2) {
          int n2 = b2 - k2; if (n2 < 0) n2 += K;
          const double* row = theta + (size_t)K * (n2 + (size_t)K * n3);
          const double w2 = xw[6 + k2], d2 = xw[18 + k2];
          double s0 = 0.0, s1 = 0.0;
#pragma unroll 1
          for (int k = 0; k < 6; ++k) { int n1 = b1 - k; if (n1 < 0) n1 += K; const double th = row[n1]; s0 += xw[k] * th; s1 += xw[12 + k] * th; }
          f1 += s1 * w2 * w3; f2 += s0 * d2 * w3; f3 += s0 * w2 * d3;
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();      // (the next trip's taps overwrite xw)
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) { f1 += __shfl_xor(f1, o, 64); f2 += __shfl_xor(f2, o, 64); f3 += __shfl_xor(f3, o, 64); }
    if (live && sub == 0) {
      const double sc = q * conv;
      f1 *= sc; f2 *= sc; f3 *= sc;
      const double Kd = (double)K;
      const double fx = -Kd * (k00 * f1 + k10 * f2 + k20 * f3);
      const double fy = -Kd * (k01 * f1 + k11 * f2 + k21 * f3);
      const double fz = -Kd * (k02 * f1 + k12 * f2 + k22 * f3);
      F_recip[3 * i] = fx; F_recip[3 * i + 1] = fy; F_recip[3 * i + 2] = fz;
      if (FOLD) {      // + the pair loop's accumulator of this atom (sorted position ia), consumed
        double* a = Fs + ia;
        const double g1 = a[0], g2 = a[NFs], g3 = a[2 * (size_t)NFs];
        a[0] = 0.0; a[NFs] = 0.0; a[2 * (size_t)NFs] = 0.0;
        const bool add = i >= add_lo && i < add_hi;
        atomicAdd(&F[3 * i], g1 + (add ? fx : 0.0)); atomicAdd(&F[3 * i + 1], g2 + (add ? fy : 0.0)); atomicAdd(&F[3 * i + 2], g3 + (add ? fz : 0.0));
      } else
      if (F && i >= add_lo && i < add_hi) {
        if (atomic) { atomicAdd(&F[3 * i], fx); atomicAdd(&F[3 * i + 1], fy); atomicAdd(&F[3 * i + 2], fz); }
        else { F[3 * i] += fx; F[3 * i + 1] += fy; F[3 * i + 2] += fz; }
      }
    }
  }
#ifdef PME_GATHER_DIAG
  const unsigned long long gd3 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (threadIdx.x == 0 && (blockIdx.x % 61) == 5)
    printf("gather diag block %d atoms %d: prologue %llu clk (100 MHz), fill %llu, atom loop %llu, drain %llu\n", blockIdx.x, e - s, gd1 - gd0, gd2 - gd1, gd3 - gd2, __builtin_amdgcn_s_memtime() - gd3);
#endif
}

// MFMA gather (option gather_mfma; north_star: "... spreading/GATHERING contraction"): the transpose of k_pme_spread_mfma.  One workgroup
// (4 waves) per neighbour cell, the cell's 16^3 potential tile in LDS.  For 16 atoms at a time and every plane z of the tile two GEMMs
//     G_z[y][a] = sum_x theta_z[y][x] Wx_a[x],     H_z[y][a] = sum_x theta_z[y][x] dWx_a[x]          (M = y = 16, N = atoms = 16, K = x = 16)
// on v_mfma_f64_16x16x4_f64 (A[row = y = lane & 15][k = lane >> 4] = theta_z[y][4 j + k], B[k][col = a = lane & 15] = the atom's DENSE x vector,
// built from its six taps; D[row = y = (lane >> 4) + 4 v][col = a]), then per lane the y and z contractions of derivative_grid_Q
// (src/pme.f90:346-493):  f1 += Wz[z] sum_y Wy[y] H,  f2 += Wz[z] sum_y dWy[y] G,  f3 += dWz[z] sum_y Wy[y] G, and a 4-lane sum over the y quarters.
// A wave takes the cell's 16-atom blocks round-robin.  16/6 of the direct loop's multiply-adds along x and all 16 x 16 (y, z) pairs instead of
// 6 x 6: the matrix cores of gfx950 run fp64 at the rate of the vector units, so this is a measurement, not the default.
// Atoms whose stencil left the tile since the last list build are gathered from global memory, 8 lanes per atom, as k_pme_gather does.
constexpr int GM_ROW = 17, GM_PLANE = 17 * 16;      // LDS pitches of the potential tile (doubles): the 16 rows an A operand reads fall into distinct banks
__global__ __launch_bounds__(256) void k_pme_gather_mfma(int K, int3 nc, const int* __restrict__ cell_start, const int* __restrict__ perm,
                                                         const double* __restrict__ chg, const int* __restrict__ base, const double* __restrict__ w,
                                                         const double* __restrict__ dw, const double* __restrict__ theta, double conv, double k00, double k01,
                                                         double k02, double k10, double k11, double k12, double k20, double k21, double k22,
                                                         double* __restrict__ F_recip, double* __restrict__ F, int atomic, int add_lo, int add_hi) {
  extern __shared__ double sm[];
  double* tile = sm;                                   // [16][16][17]
  double* sw_all = sm + 16 * GM_PLANE;                 // [4 waves][16 atoms][36]: w[3][6], dw[3][6]
  int* sl_all = (int*)(sw_all + 4 * 16 * 36);          // [4][16][4]: tile-local tap-0 coordinates, atom (or -1)
  __shared__ int s_out[512], s_nout;
  const int cidx = blockIdx.x;
  const int cx = cidx % nc.x, cy = (cidx / nc.x) % nc.y, cz = cidx / (nc.x * nc.y);
  const int ox = (int)floor((double)cx * K / nc.x) - 6, oy = (int)floor((double)cy * K / nc.y) - 6, oz = (int)floor((double)cz * K / nc.z) - 6;
  const int s = cell_start[cidx], e = cell_start[cidx + 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) s_nout = 0;
  for (int r0 = threadIdx.x; r0 < 4096; r0 += 4 * blockDim.x) {      // the whole 16^3 tile, x fastest, four loads in flight per thread
    double v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int r = r0 + u * blockDim.x, a = r & 15, b = (r >> 4) & 15, cc = r >> 8;
      int n1 = ox + a; n1 += n1 < 0 ? K : 0; n1 -= n1 >= K ? K : 0;
      int n2 = oy + b; n2 += n2 < 0 ? K : 0; n2 -= n2 >= K ? K : 0;
      int n3 = oz + cc; n3 += n3 < 0 ? K : 0; n3 -= n3 >= K ? K : 0;
      v[u] = theta[n1 + (size_t)K * (n2 + (size_t)K * n3)];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int r = r0 + u * blockDim.x; tile[(r & 15) + GM_ROW * ((r >> 4) & 15) + GM_PLANE * (r >> 8)] = v[u]; }
  }
  __syncthreads();
  double* sw = sw_all + wave * 16 * 36;
  int* sl = sl_all + wave * 16 * 4;
  const int col = lane & 15, kq = lane >> 4;
  const double Kd = (double)K;
  for (int b0 = s + 16 * wave; b0 < e; b0 += 64) {      // this wave's 16-atom blocks
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();      // (the block before has been read)
    if (lane < 16) {
      const int ia = b0 + lane;
      int i = -1, l1 = 5, l2 = 5, l3 = 5;
      if (ia < e) {
        i = perm[ia];
        l1 = base[3 * i] - ox; l1 += l1 < 0 ? K : 0; l1 -= l1 >= K ? K : 0;
        l2 = base[3 * i + 1] - oy; l2 += l2 < 0 ? K : 0; l2 -= l2 >= K ? K : 0;
        l3 = base[3 * i + 2] - oz; l3 += l3 < 0 ? K : 0; l3 -= l3 >= K ? K : 0;
        if (!(l1 >= 5 && l1 < 16 && l2 >= 5 && l2 < 16 && l3 >= 5 && l3 < 16)) { const int slot = atomicAdd(&s_nout, 1); if (slot < 512) s_out[slot] = i; i = -1; l1 = l2 = l3 = 5; }      // (the host launches this kernel for cells of <= 512 atoms)
      }
      sl[4 * lane] = l1; sl[4 * lane + 1] = l2; sl[4 * lane + 2] = l3; sl[4 * lane + 3] = i;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int k = lane; k < 16 * 36; k += 64) {
      const int a = k / 36, r = k - 36 * a, i = sl[4 * a + 3];
      sw[k] = i < 0 ? 0.0 : r < 18 ? w[(size_t)i * 18 + r] : dw[(size_t)i * 18 + r - 18];      // (an atom that is not done here has zero weights)
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int l1 = sl[4 * col], l2 = sl[4 * col + 1], l3 = sl[4 * col + 2], i = sl[4 * col + 3];
    const double* swa = sw + 36 * col;
    double bx[4], bdx[4], wy[4], dwy[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int dx = l1 - (4 * j + kq), dy = l2 - (kq + 4 * j);
      const bool okx = dx >= 0 && dx < 6, oky = dy >= 0 && dy < 6;
      const int cxk = okx ? dx : 0, cyk = oky ? dy : 0;
      bx[j] = okx ? swa[cxk] : 0.0; bdx[j] = okx ? swa[18 + cxk] : 0.0;
      wy[j] = oky ? swa[6 + cyk] : 0.0; dwy[j] = oky ? swa[24 + cyk] : 0.0;
    }
    double f1 = 0.0, f2 = 0.0, f3 = 0.0;
    for (int z = 0; z < 16; ++z) {
      const int dz = l3 - z;
      const bool okz = dz >= 0 && dz < 6;
      if (!__any(okz)) continue;      // no atom of the block reaches this plane
      const double wz = okz ? swa[12 + dz] : 0.0, dwz = okz ? swa[30 + dz] : 0.0;
      mfma_d4 G = (mfma_d4){0.0, 0.0, 0.0, 0.0}, H = (mfma_d4){0.0, 0.0, 0.0, 0.0};
      const double* arow = tile + GM_PLANE * z + GM_ROW * col + kq;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const double av = arow[4 * j];
        G = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bx[j], G, 0, 0, 0);
        H = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bdx[j], H, 0, 0, 0);
      }
      double p1 = 0.0, p2 = 0.0, p3 = 0.0;
#pragma unroll
      for (int v = 0; v < 4; ++v) { p1 += wy[v] * H[v]; p2 += dwy[v] * G[v]; p3 += wy[v] * G[v]; }
      f1 += wz * p1; f2 += wz * p2; f3 += dwz * p3;
    }
    f1 += __shfl_xor(f1, 16, 64); f2 += __shfl_xor(f2, 16, 64); f3 += __shfl_xor(f3, 16, 64);
    f1 += __shfl_xor(f1, 32, 64); f2 += __shfl_xor(f2, 32, 64); f3 += __shfl_xor(f3, 32, 64);
    if (kq == 0 && i >= 0) {
      const double sc = chg[i] * conv;
      f1 *= sc; f2 *= sc; f3 *= sc;
      const double fx = -Kd * (k00 * f1 + k10 * f2 + k20 * f3);
      const double fy = -Kd * (k01 * f1 + k11 * f2 + k21 * f3);
      const double fz = -Kd * (k02 * f1 + k12 * f2 + k22 * f3);
      F_recip[3 * i] = fx; F_recip[3 * i + 1] = fy; F_recip[3 * i + 2] = fz;
      if (F && i >= add_lo && i < add_hi) {
        if (atomic) { atomicAdd(&F[3 * i], fx); atomicAdd(&F[3 * i + 1], fy); atomicAdd(&F[3 * i + 2], fz); }
        else { F[3 * i] += fx; F[3 * i + 1] += fy; F[3 * i + 2] += fz; }
      }
    }
  }
  __syncthreads();
  const int nout = min(s_nout, 512);
  const int sub = threadIdx.x & 7;
  for (int o0 = 0; o0 < nout; o0 += 32) {      // stencils that left the tile: from global memory, 8 lanes per atom (uniform trip count per wave: shuffles)
    const int o = o0 + (threadIdx.x >> 3);
    const bool live = o < nout;
    const int i = live ? s_out[o] : 0;
    double f1 = 0.0, f2 = 0.0, f3 = 0.0;
    if (live && sub < 6) {
      const int b1 = base[3 * i], b2 = base[3 * i + 1], b3 = base[3 * i + 2];
      const double* wi = w + (size_t)i * 18;
      const double* dwi = dw + (size_t)i * 18;
      int n3 = b3 - sub; if (n3 < 0) n3 += K;
      for (int k2 = 0; k2 < 6; ++k2) {
        int n2 = b2 - k2; if (n2 < 0) n2 += K;
        const double* row = theta + (size_t)K * (n2 + (size_t)K * n3);
        double s0 = 0.0, s1 = 0.0;
        for (int k = 0; k < 6; ++k) { int n1 = b1 - k; if (n1 < 0) n1 += K; const double th = row[n1]; s0 += wi[k] * th; s1 += dwi[k] * th; }
        f1 += s1 * wi[6 + k2] * wi[12 + sub]; f2 += s0 * dwi[6 + k2] * wi[12 + sub]; f3 += s0 * wi[6 + k2] * dwi[12 + sub];
      }
    }
#pragma unroll
    for (int q = 4; q > 0; q >>= 1) { f1 += __shfl_xor(f1, q, 64); f2 += __shfl_xor(f2, q, 64); f3 += __shfl_xor(f3, q, 64); }
    if (live && sub == 0) {
      const double sc = chg[i] * conv;
      f1 *= sc; f2 *= sc; f3 *= sc;
      const double fx = -Kd * (k00 * f1 + k10 * f2 + k20 * f3);
      const double fy = -Kd * (k01 * f1 + k11 * f2 + k21 * f3);
      const double fz = -Kd * (k02 * f1 + k12 * f2 + k22 * f3);
      F_recip[3 * i] = fx; F_recip[3 * i + 1] = fy; F_recip[3 * i + 2] = fz;
      if (F && i >= add_lo && i < add_hi) {
        if (atomic) { atomicAdd(&F[3 * i], fx); atomicAdd(&F[3 * i + 1], fy); atomicAdd(&F[3 * i + 2], fz); }
        else { F[3 * i] += fx; F[3 * i + 1] += fy; F[3 * i + 2] += fz; }
      }
    }
  }
}

}  // namespace

// stream of the PME front end (prep ... inverse FFT): the side stream while the principal pair loop runs on the main one
static inline hipStream_t pme_stream(const msevb_ctx* c) { return c->pme_st ? c->pme_st : c->stream; }

#define FFTCHK(call) do { if ((call) != rocfft_status_success) FAIL(MSEVB_ERR_FFT, std::string("rocFFT failure: ") + #call); } while (0)

static bool g_rocfft_ready = false;

int pme_setup(msevb_ctx* c) {
  const int K = c->K, N = c->N;
  const size_t K3 = (size_t)K * K * K;
  if (!g_rocfft_ready) { if (rocfft_setup() != rocfft_status_success) FAIL(MSEVB_ERR_FFT, "rocfft_setup failed"); g_rocfft_ready = true; }
  const int batch = c->fft_batch < 1 ? 1 : c->fft_batch;
  if (c->d_grid.alloc(K3 * batch) || c->d_Q.alloc(K3) || c->d_theta.alloc(K3) || c->d_w.alloc(18 * (size_t)N) || c->d_dw.alloc(18 * (size_t)N) || c->d_uscale.alloc(3 * (size_t)N) ||
      c->d_base.alloc(3 * (size_t)N) || c->d_base_s.alloc(3 * (size_t)N) || c->d_uscale_s.alloc(3 * (size_t)N) || c->d_rows_s.alloc(3 * (size_t)N) || c->d_w_s.alloc(18 * (size_t)N) || c->d_splindex.alloc(18 * (size_t)N) || c->d_force_recip.alloc(3 * (size_t)N))
    FAIL(MSEVB_ERR_HIP, "hipMalloc failed (PME)");
  size_t len[3] = {(size_t)K, (size_t)K, (size_t)K};
  pme_teardown(c);
  FFTCHK(rocfft_plan_create(&c->plan_fwd, rocfft_placement_inplace, rocfft_transform_type_complex_forward, rocfft_precision_double, 3, len, 1, nullptr));
  FFTCHK(rocfft_plan_create(&c->plan_bwd, rocfft_placement_inplace, rocfft_transform_type_complex_inverse, rocfft_precision_double, 3, len, 1, nullptr));
  size_t wf = 0, wb = 0;
  FFTCHK(rocfft_plan_get_work_buffer_size(c->plan_fwd, &wf));
  FFTCHK(rocfft_plan_get_work_buffer_size(c->plan_bwd, &wb));
  if (wb > wf) wf = wb;
  FFTCHK(rocfft_execution_info_create(&c->fft_info));
  if (wf) {
    if (c->d_fft_work.alloc(wf)) FAIL(MSEVB_ERR_HIP, "hipMalloc failed (FFT work buffer)");
    FFTCHK(rocfft_execution_info_set_work_buffer(c->fft_info, c->d_fft_work.p, wf));
  }
  FFTCHK(rocfft_execution_info_set_stream(c->fft_info, c->stream));
  CHK(fft_lds_setup(c));
  return pme_build_green(c);
}

// Real-space kernel of the reciprocal sum, G = IFFT(CB) (unnormalised backward transform, as theta = IFFT(CB FFT(Q)) is): theta = G * Q as a
// periodic convolution.  The MS-EVB path evaluates the diabats' charge DIFFERENCES through it (evb.hip).  Rebuilt with the box (CB).
int pme_build_green(msevb_ctx* c) {
  const size_t K3 = (size_t)c->K * c->K * c->K;
  if (!c->plan_bwd || !c->d_CB.p) return MSEVB_OK;
  if (c->d_green.alloc(K3)) FAIL(MSEVB_ERR_HIP, "hipMalloc failed (PME real-space kernel)");
  hipLaunchKernelGGL(k_real_to_complex, dim3(1024), dim3(256), 0, c->stream, K3, c->d_CB.p, c->d_grid.p);
  void* buf[1] = {c->d_grid.p};
  FFTCHK(rocfft_execution_info_set_stream(c->fft_info, c->stream));
  FFTCHK(rocfft_execute(c->plan_bwd, buf, nullptr, c->fft_info));
  hipLaunchKernelGGL(k_copy_real, dim3(1024), dim3(256), 0, c->stream, K3, c->d_grid.p, c->d_green.p);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(c->stream));
  return MSEVB_OK;
}

void pme_teardown(msevb_ctx* c) {
  if (c->plan_fwd) { rocfft_plan_destroy(c->plan_fwd); c->plan_fwd = nullptr; }
  if (c->plan_bwd) { rocfft_plan_destroy(c->plan_bwd); c->plan_bwd = nullptr; }
  for (auto& kv : c->plans_b) { if (kv.second.first) rocfft_plan_destroy(kv.second.first); if (kv.second.second) rocfft_plan_destroy(kv.second.second); }
  c->plans_b.clear();
  if (c->fft_info) { rocfft_execution_info_destroy(c->fft_info); c->fft_info = nullptr; }
  if (c->fft_info_b) { rocfft_execution_info_destroy(c->fft_info_b); c->fft_info_b = nullptr; }
}

// The plan pair for `batch` grids, created on first use and kept (c->plans_b).  The first batched call also creates the pairs for the batch
// sizes around it: a proton-transfer tree gains and loses a diabat every few dozen steps, and a plan pair costs milliseconds to make.
static int pme_batched_plan_make(msevb_ctx* c, int batch) {
  if (c->plans_b.count(batch)) return MSEVB_OK;
  size_t len[3] = {(size_t)c->K, (size_t)c->K, (size_t)c->K};
  rocfft_plan f = nullptr, b = nullptr;
  FFTCHK(rocfft_plan_create(&f, rocfft_placement_inplace, rocfft_transform_type_complex_forward, rocfft_precision_double, 3, len, batch, nullptr));
  FFTCHK(rocfft_plan_create(&b, rocfft_placement_inplace, rocfft_transform_type_complex_inverse, rocfft_precision_double, 3, len, batch, nullptr));
  c->plans_b[batch] = std::make_pair(f, b);
  size_t wf = 0, wb = 0;
  FFTCHK(rocfft_plan_get_work_buffer_size(f, &wf));
  FFTCHK(rocfft_plan_get_work_buffer_size(b, &wb));
  if (wb > wf) wf = wb;
  // the batched plans have their OWN work buffer and execution info: growing a shared buffer would free the one whose address the
  // captured graph of the single-grid front end replays with (K = 128 needs work buffers; K = 64 does not - config C faulted here)
  if (!c->fft_info_b) FFTCHK(rocfft_execution_info_create(&c->fft_info_b));
  if (wf > c->d_fft_work_b.n) {
    HIPCHK(hipDeviceSynchronize());
    if (c->d_fft_work_b.alloc(wf)) FAIL(MSEVB_ERR_HIP, "hipMalloc failed (FFT work buffer)");
    FFTCHK(rocfft_execution_info_set_work_buffer(c->fft_info_b, c->d_fft_work_b.p, c->d_fft_work_b.n));
  }
  return MSEVB_OK;
}

static int pme_batched_plans(msevb_ctx* c, int batch, rocfft_plan* pf, rocfft_plan* pb) {
  if (!c->plans_b.count(batch)) {
    const bool first = c->plans_b.empty();
    if (int rc = pme_batched_plan_make(c, batch)) return rc;
    if (first && c->K <= 64) {      // small grids only: at K = 128 a work buffer per size would be 100s of MB and the step is 8 ms anyway
      const int hi = batch + 3 > 10 ? batch + 3 : 10;
      for (int b = 2; b <= hi; ++b) if (int rc = pme_batched_plan_make(c, b)) return rc;
    }
  }
  *pf = c->plans_b[batch].first; *pb = c->plans_b[batch].second;
  return MSEVB_OK;
}

// theta_g = IFFT(CB * FFT(grid_g)) in place for `batch` contiguous grids; the energy partials of grid g go to
// part[g*part_stride + 0..nb) (returns nb through *nblocks)
int pme_convolve(msevb_ctx* c, double2* grids, int batch, double* part, int part_stride, int* nblocks) {
  const size_t K3 = (size_t)c->K * c->K * c->K;
  rocfft_plan pf = c->plan_fwd, pb = c->plan_bwd;
  if (batch > 1) {
    if (int rc = pme_batched_plans(c, batch, &pf, &pb)) return rc;
  }
  void* buf[1] = {grids};
  rocfft_execution_info info = batch > 1 ? c->fft_info_b : c->fft_info;
  FFTCHK(rocfft_execution_info_set_stream(info, pme_stream(c)));
  FFTCHK(rocfft_execute(pf, buf, nullptr, info));
  int nb = (int)((K3 + 255) / 256);
  if (nb > 512) nb = 512;
  hipLaunchKernelGGL(k_pme_conv, dim3(nb, batch), dim3(256), 0, pme_stream(c), K3, c->d_CB.p, grids, part_stride, part, 0.5 * c->conv);
  FFTCHK(rocfft_execute(pb, buf, nullptr, info));
  HIPCHK(hipGetLastError());
  if (nblocks) *nblocks = nb;
  return MSEVB_OK;
}

// whether the tiled spread / gather serve this context (workgroup per neighbour cell, tile in LDS)
bool pme_tiles_ok(const msevb_ctx* c) {
  int3 T;
  T.x = (int)ceil((double)c->K / c->ncell[0]) + 8; T.y = (int)ceil((double)c->K / c->ncell[1]) + 8; T.z = (int)ceil((double)c->K / c->ncell[2]) + 8;
  const size_t tile_bytes = (size_t)spread_plane_stride(T) * T.z * sizeof(double) + (size_t)SPREAD_CHUNK * (18 * 8 + 8 + 12);
  return c->cells_valid && c->spread_tiles && tile_bytes <= 64 * 1024 && T.x <= c->K && T.y <= c->K && T.z <= c->K;
}
bool pme_gather_can_fold(const msevb_ctx* c) {      // mirrors the dispatch of pme_gather below
  const bool mfma = c->gather_mfma && c->cells_valid && c->weights_valid;
  return !mfma && pme_tiles_ok(c) && !c->weights_valid && !c->weights_side;
}
// no weight store this call: the tiled kernels run, nothing else that reads d_w / d_dw does (the caller says so: weights_needed_now)
bool pme_weightless_now(const msevb_ctx* c) {
  const bool other_spread = c->cells_valid && (c->spread_bricks || c->spread_mfma || c->gather_mfma);      // (kernels that read the stored weights)
  return c->pme_weightless && !c->weights_needed_now && pme_tiles_ok(c) && !other_spread;
}

// MS-EVB fast path with a weight-free front end: the 36 stored weights per atom all the same, by k_pme_prep (weights only) on an idle side stream
// at the very start of the step (`after` = the new positions), for the ONE gather of the blend phase - which sits on the step's critical path
// and takes 30 us with stored weights, 40 recomputing table rows.  pme_gather waits for ev_weights.
int pme_side_weights(msevb_ctx* c, hipStream_t st, hipEvent_t after) {
  if (!c->pme_side_weights) return MSEVB_OK;
  if (!c->ev_weights) HIPCHK(hipEventCreateWithFlags(&c->ev_weights, hipEventDisableTiming));
  HIPCHK(hipStreamWaitEvent(st, after, 0));      // (the new positions)
  const long nt = 18L * c->N;
  hipLaunchKernelGGL(k_pme_prep, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, st, c->N, c->K, c->spline_grid, c->d_xyz.p, c->kk[0][0], c->kk[0][1],
                     c->kk[0][2], c->kk[1][0], c->kk[1][1], c->kk[1][2], c->kk[2][0], c->kk[2][1], c->kk[2][2], c->d_B6.p, c->d_B5.p, (double*)nullptr,
                     (int*)nullptr, (int*)nullptr, c->d_w.p, c->d_dw.p, c->d_energies.p + S_FAULT, (const int*)nullptr, (int*)nullptr, (double*)nullptr);
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(c->ev_weights, st));
  return MSEVB_OK;
}

int pme_prepare_atoms(msevb_ctx* c, bool with_indices) {
  const int N = c->N;
  const long nt = 18L * N;
  if (!with_indices && pme_weightless_now(c)) {
    hipLaunchKernelGGL(k_pme_prep_light, dim3((unsigned)((3 * N + 255) / 256)), dim3(256), 0, pme_stream(c), N, c->K, c->d_xyz.p, c->kk[0][0], c->kk[0][1], c->kk[0][2], c->kk[1][0],
                       c->kk[1][1], c->kk[1][2], c->kk[2][0], c->kk[2][1], c->kk[2][2], c->d_uscale.p, c->d_base.p, c->d_energies.p + S_FAULT, c->d_inv_perm.p, c->d_uscale_s.p, c->d_base_s.p, c->d_rows_s.p, (double)c->spline_grid);
    HIPCHK(hipGetLastError());
    c->weights_valid = false; c->weights_side = false;
    return MSEVB_OK;
  }
  c->weights_valid = true; c->weights_side = false;
  hipLaunchKernelGGL(k_pme_prep, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, pme_stream(c), N, c->K, c->spline_grid, c->d_xyz.p, c->kk[0][0], c->kk[0][1],
                     c->kk[0][2], c->kk[1][0], c->kk[1][1], c->kk[1][2], c->kk[2][0], c->kk[2][1], c->kk[2][2], c->d_B6.p, c->d_B5.p, c->d_uscale.p,
                     c->d_base.p, with_indices ? c->d_splindex.p : (int*)nullptr, c->d_w.p, c->d_dw.p, c->d_energies.p + S_FAULT,
                     (c->cells_valid && c->spread_bricks) ? c->d_inv_perm.p : (const int*)nullptr, c->d_base_s.p, c->d_w_s.p);
  HIPCHK(hipGetLastError());
  return MSEVB_OK;
}

int pme_gather(msevb_ctx* c, const double2* grid, const double* chg, double* F_recip, double* F_add, bool atomic, const double* theta, double* Fs_fold, bool* folded) {
  if (folded) *folded = false;
  const int N = c->N;
  int add_lo, add_hi;
  shard_range(c, N, add_lo, add_hi);
  {
    int3 T;
    T.x = (int)ceil((double)c->K / c->ncell[0]) + 8; T.y = (int)ceil((double)c->K / c->ncell[1]) + 8; T.z = (int)ceil((double)c->K / c->ncell[2]) + 8;
    const size_t tile_bytes = (size_t)spread_plane_stride(T) * T.z * sizeof(double);
    if (c->gather_mfma && c->cells_valid && c->weights_valid && T.x <= 16 && T.y <= 16 && T.z <= 16 && c->K >= 16 && c->cell_atoms_max <= 512) {      // 16^3 tile per cell, x contraction on the matrix cores
      const int ncell = c->ncell[0] * c->ncell[1] * c->ncell[2];
      const size_t lds = (size_t)16 * GM_PLANE * 8 + 4 * 16 * 36 * 8 + 4 * 16 * 4 * 4;
      hipLaunchKernelGGL(k_pme_gather_mfma, dim3(ncell), dim3(256), lds, c->stream, c->K, make_int3(c->ncell[0], c->ncell[1], c->ncell[2]), c->d_cell_start.p, c->d_perm.p, chg, c->d_base.p,
                         c->d_w.p, c->d_dw.p, theta ? theta : c->d_theta.p, c->conv, c->kk[0][0], c->kk[0][1], c->kk[0][2], c->kk[1][0], c->kk[1][1], c->kk[1][2], c->kk[2][0], c->kk[2][1],
                         c->kk[2][2], F_recip, F_add, atomic ? 1 : 0, add_lo, add_hi);
      HIPCHK(hipGetLastError());
      return MSEVB_OK;
    }
    if (c->cells_valid && c->spread_tiles && tile_bytes <= 64 * 1024 && T.x <= c->K && T.y <= c->K && T.z <= c->K) {      // workgroup per cell, potential staged in LDS
      const int ncell = c->ncell[0] * c->ncell[1] * c->ncell[2];
      if (!c->weights_valid && c->weights_side) HIPCHK(hipStreamWaitEvent(c->stream, c->ev_weights, 0));      // (k_pme_weights filled them in on a side stream meanwhile)
      if (!c->weights_valid && !c->weights_side) {      // k_pme_prep_light ran: per workgroup 192 B of LDS per 8-lane group + 48 B per atom of the largest cell
        int cap = std::min(((std::max((int)(c->cell_atoms_max * 1.05) + 1, 16) + 15) & ~15), 1024);      // (the largest cell of the last build whose statistics the host has seen; in steps of 16 and never smaller than before: the launch stays what a captured force call holds)
        cap = c->gather_cap_seen = std::max(cap, c->gather_cap_seen);
        // 256 threads (32 atoms per trip) when the cells hold <= 256 atoms: 49 KB of LDS per workgroup, three per CU - all 729 workgroups of configs[2] resident at once instead of
        // 1.4 rounds of two 512-thread ones (0.4021 against 0.4121 ms per step; 320 and 384 threads: 0.4209 / 0.4170 - HISTORY.md E)
        const int gthreads = c->pme_gather_threads > 0 ? c->pme_gather_threads : (c->cell_atoms_max <= 256 ? 256 : c->pme_tile_threads);
        const size_t lds = tile_bytes + (size_t)gthreads * 24 + (size_t)cap * 48 + 16;
        if (lds > 160 * 1024) FAIL(MSEVB_ERR_LIMIT, "pme_gather: cell too large for the weight-free gather (set option pme_weightless=0)");
        const bool fuse = Fs_fold && folded && F_add && atomic;
        auto kern = fuse ? k_pme_gather_tiles_wl<true> : k_pme_gather_tiles_wl<false>;
        int& attr = fuse ? c->attr_gather_wl_b : c->attr_gather_wl;
        if (lds > 64 * 1024 && attr < (int)lds) { HIPCHK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); attr = (int)lds; }
        hipLaunchKernelGGL(kern, dim3(ncell), dim3(gthreads), lds, c->stream, c->K, make_int3(c->ncell[0], c->ncell[1], c->ncell[2]), T, c->d_cell_start.p, c->d_perm.p,
                           chg, c->d_base_s.p, c->d_uscale_s.p, c->d_rows_s.p, c->d_B6.p, c->d_B5.p, c->spline_grid, cap, theta ? theta : c->d_theta.p, c->conv, c->kk[0][0], c->kk[0][1], c->kk[0][2], c->kk[1][0],
                           c->kk[1][1], c->kk[1][2], c->kk[2][0], c->kk[2][1], c->kk[2][2], F_recip, F_add, atomic ? 1 : 0, add_lo, add_hi, fuse ? Fs_fold : (double*)nullptr, c->N);
        if (fuse) *folded = true;
      } else
      hipLaunchKernelGGL(k_pme_gather_tiles, dim3(ncell), dim3(c->pme_tile_threads), tile_bytes, c->stream, c->K, make_int3(c->ncell[0], c->ncell[1], c->ncell[2]), T, c->d_cell_start.p, c->d_perm.p,
                         chg, c->d_base.p, c->d_w.p, c->d_dw.p, theta ? theta : c->d_theta.p, c->conv, c->kk[0][0], c->kk[0][1], c->kk[0][2], c->kk[1][0], c->kk[1][1], c->kk[1][2],
                         c->kk[2][0], c->kk[2][1], c->kk[2][2], F_recip, F_add, atomic ? 1 : 0, add_lo, add_hi);
      HIPCHK(hipGetLastError());
      return MSEVB_OK;
    }
  }
  if (!c->weights_valid && !c->weights_side) FAIL(MSEVB_ERR_STATE, "pme_gather: the per-atom weights of this step were not stored (option pme_weightless) and the tiled gather does not serve this context");
  hipLaunchKernelGGL(k_pme_gather, dim3((N * 8 + 255) / 256), dim3(256), 0, c->stream, N, c->K, chg, c->d_base.p, c->d_w.p, c->d_dw.p, theta ? theta : c->d_theta.p, c->conv, c->kk[0][0],
                     c->kk[0][1], c->kk[0][2], c->kk[1][0], c->kk[1][1], c->kk[1][2], c->kk[2][0], c->kk[2][1], c->kk[2][2], F_recip, F_add, atomic ? 1 : 0, add_lo, add_hi);
  HIPCHK(hipGetLastError());
  return MSEVB_OK;
}

int pme_forces(msevb_ctx* c, int slot_recip, bool join_aux) { CHK(pme_front(c, slot_recip, join_aux)); return pme_tail(c, join_aux); }

// front end: stencils, spreading, FFT, convolution (+ energy), inverse FFT, real part - on `aux` when on_aux
int pme_front(msevb_ctx* c, int slot_recip, bool join_aux) {
  const int N = c->N, K = c->K;
  const size_t K3 = (size_t)K * K * K;
  // join_aux: the caller has forked the side streams; the front end (stencils, spreading, FFTs, convolution) goes to `aux` beside the
  // pair loop on the main stream; the gather follows on the main stream behind the pair loop (atomic adds: the per-molecule kernel
  // may still be running on aux2), which leaves `aux` free for the diabat grids of the EVB phase
  c->pme_st = join_aux ? c->aux : nullptr;
  const hipStream_t st = pme_stream(c);
  // The front end is the same ~14 launches with the same arguments every step (all its inputs live at fixed device addresses): on the
  // side stream it is captured once into a hipGraph and replayed with one launch (the ~14 launches cost the host ~80 us per step, and
  // the host is what decides when the EVB chains behind them can start).  Steps that sample the phase timers launch directly.
  struct Key { const void* p[9]; int v[8]; } key = {{c->d_grid.p, c->d_Q.p, c->d_theta.p, c->d_w.p, c->d_base.p, c->d_cell_start.p, c->d_perm.p, c->d_chg.p, c->d_fft_work.p},
                                               {N, K, c->ncell[0], c->ncell[1], c->ncell[2], c->cells_valid ? 1 : 0, (c->spread_tiles ? 1 : 0) + (c->spread_mfma ? 2 : 0) + (c->spread_bricks ? 4 : 0) + (fft_lds_usable(c) ? 8 : 0) + (c->pme_split_now ? 16 : 0) + (pme_weightless_now(c) ? 32 : 0), slot_recip}};
  const bool use_graph = join_aux && c->graphs && !c->timing_now && !t_step_rec;      // (inside a captured / replayed force call the launches go into THAT graph)
  bool capturing = false;
  int nbk = 0;
  if (use_graph && c->pme_graph_exec && memcmp(&key, c->pme_graph_key, sizeof key) == 0) {
    { ApiScope aps(4); HIPCHK(hipGraphLaunch((hipGraphExec_t)c->pme_graph_exec, st)); }
    nbk = c->pme_graph_nbk;
  } else {
    if (use_graph) {
      if (c->pme_graph_exec) { (void)hipGraphExecDestroy((hipGraphExec_t)c->pme_graph_exec); c->pme_graph_exec = nullptr; }
      capturing = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) == hipSuccess;
    }
    phase_begin(c, PH_SPREAD, st);
    CHK(pme_prepare_atoms(c, false));
    bool bricks = false;
    const bool own = fft_lds_usable(c);      // 64^3: three-kernel solve on the REAL charge grid (fft_lds.hip) - spread straight into Q, no real<->complex passes
    {
      int3 T;
      T.x = (int)ceil((double)K / c->ncell[0]) + 8; T.y = (int)ceil((double)K / c->ncell[1]) + 8; T.z = (int)ceil((double)K / c->ncell[2]) + 8;
      const size_t tile_bytes = (size_t)spread_plane_stride(T) * T.z * sizeof(double) + (size_t)SPREAD_CHUNK * (18 * 8 + 8 + 12);      // tile + staged atom inputs
      bricks = c->cells_valid && c->spread_bricks && K >= 16 && T.x <= 16 && T.y <= 16 && T.z <= 16 && c->d_inv_perm.p;      // bricks of <= 8 points per axis
      const bool mfma = !bricks && c->cells_valid && c->spread_mfma && T.x <= 16 && T.y <= 16 && T.z <= 16 && K >= 16;
      const bool to_real = own && !bricks && !mfma;
      double* g = to_real ? c->d_Q.p : (double*)c->d_grid.p;
      const int gs = to_real ? 1 : 2;
      if (!bricks) HIPCHK(hipMemsetAsync(g, 0, K3 * sizeof(double) * gs, st));
      if (bricks) {      // owner computes: no zero-fill, no atomics, no real-copy pass
        const int ncell = c->ncell[0] * c->ncell[1] * c->ncell[2];
        hipLaunchKernelGGL(k_pme_spread_bricks, dim3(ncell), dim3(256), (size_t)3 * BRICK_CHUNK * 8 * sizeof(double), st, K, make_int3(c->ncell[0], c->ncell[1], c->ncell[2]),
                           c->d_cell_start.p, c->d_base_s.p, c->d_w_s.p, c->d_posq.p, c->d_grid.p, c->d_Q.p, c->d_energies.p + S_BRICK);
      } else if (mfma) {      // 16^3 tile per cell: the contraction over the cell's atoms on the matrix cores
        const int ncell = c->ncell[0] * c->ncell[1] * c->ncell[2];
        if (!c->attr_spread_mfma) { HIPCHK(hipFuncSetAttribute((const void*)k_pme_spread_mfma, hipFuncAttributeMaxDynamicSharedMemorySize, SPREAD_MFMA_CHUNK * 48 * 8)); c->attr_spread_mfma = true; }
        hipLaunchKernelGGL(k_pme_spread_mfma, dim3(ncell), dim3(256), (size_t)SPREAD_MFMA_CHUNK * 48 * 8, st, K, make_int3(c->ncell[0], c->ncell[1], c->ncell[2]),
                           c->d_cell_start.p, c->d_perm.p, c->d_chg.p, c->d_base.p, c->d_w.p, c->d_grid.p);
      } else if (c->cells_valid && c->spread_tiles && tile_bytes <= 64 * 1024 && T.x <= K && T.y <= K && T.z <= K) {
        const int ncell = c->ncell[0] * c->ncell[1] * c->ncell[2];
        hipLaunchKernelGGL(k_pme_spread_tiles, dim3(ncell), dim3(c->pme_tile_threads), tile_bytes, st, K, make_int3(c->ncell[0], c->ncell[1], c->ncell[2]), T,
                           c->d_cell_start.p, c->d_perm.p, c->d_chg.p, c->weights_valid ? c->d_base.p : c->d_base_s.p, c->weights_valid ? c->d_w.p : (const double*)nullptr, g, gs, c->d_uscale_s.p, c->d_B6.p, c->spline_grid, c->d_rows_s.p);
      } else {
        const long nt = (long)N * 36;
        hipLaunchKernelGGL(k_pme_spread, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, st, N, K, c->d_chg.p, c->d_base.p, c->d_w.p, g, gs);
      }
      int nbc = (int)((K3 + 255) / 256);
      if (nbc > 2048) nbc = 2048;
      if (!bricks && !to_real) hipLaunchKernelGGL(k_copy_real, dim3(nbc), dim3(256), 0, st, K3, c->d_grid.p, c->d_Q.p);      // (the brick kernel writes Q itself)
    }
    HIPCHK(hipGetLastError());
    phase_end(c, PH_SPREAD, st);
    if (c->pme_split_now && own) {      // the caller enqueues the solve itself (pme_front_finish), behind work of its own on this stream
    } else {
    phase_begin(c, PH_FFT, st);
    if (own) CHK(fft_lds_solve(c, c->d_Q.p, c->d_grid.p, c->d_theta.p, c->d_partials.p + (size_t)slot_recip * MAX_BLOCKS, &nbk, st));
    else {
      CHK(pme_convolve(c, c->d_grid.p, 1, c->d_partials.p + (size_t)slot_recip * MAX_BLOCKS, MAX_BLOCKS, &nbk));
      hipLaunchKernelGGL(k_pme_real, dim3(1024), dim3(256), 0, st, K3, c->d_grid.p, c->d_theta.p);
    }
    phase_end(c, PH_FFT, st);
    }
    if (capturing) {
      hipGraph_t g = nullptr;
      if (hipStreamEndCapture(st, &g) == hipSuccess && g) {
        hipGraphExec_t ex = nullptr;
        if (hipGraphInstantiate(&ex, g, nullptr, nullptr, 0) == hipSuccess && ex) {
          c->pme_graph_exec = ex; memcpy(c->pme_graph_key, &key, sizeof key); c->pme_graph_nbk = nbk;
          HIPCHK(hipGraphLaunch(ex, st));       // the capture itself executed nothing
        } else c->graphs = false;
        (void)hipGraphDestroy(g);
      } else c->graphs = false;
      if (!c->graphs) FAIL(MSEVB_ERR_HIP, "hipGraph capture of the PME front end failed (set option graphs=0)");
    }
  }
  c->pme_st = nullptr;
  if (c->pme_split_now && fft_lds_usable(c)) { c->pme_split_pending = true; c->pme_split_slot = slot_recip; return MSEVB_OK; }
  CHK(reduce_partials(c, slot_recip, nbk));
  if (join_aux) HIPCHK(hipEventRecord(c->ev_join, c->aux));
  return MSEVB_OK;
}

// second half of a split front end (MS-EVB call): the solve, on `aux`, behind whatever the caller has put there since
int pme_front_finish(msevb_ctx* c) {
  if (!c->pme_split_pending) return MSEVB_OK;
  c->pme_split_pending = false;
  int nbk = 0;
  phase_begin(c, PH_FFT, c->aux);
  CHK(fft_lds_solve(c, c->d_Q.p, c->d_grid.p, c->d_theta.p, c->d_partials.p + (size_t)c->pme_split_slot * MAX_BLOCKS, &nbk, c->aux));
  phase_end(c, PH_FFT, c->aux);
  CHK(reduce_partials(c, c->pme_split_slot, nbk));
  HIPCHK(hipEventRecord(c->ev_join, c->aux));
  return MSEVB_OK;
}

// tail on the main stream: wait for the front end, gather (atomic adds when other force kernels may still be running)
int pme_tail(msevb_ctx* c, bool join_aux) {
  if (join_aux && !c->skip_side_joins_now) HIPCHK(hipStreamWaitEvent(c->stream, c->ev_join, 0));      // (MS-EVB fast path: ONE wait for all side streams in front of the blend phase)
  if (c->skip_gather_now) return MSEVB_OK;      // EVB fast path: the reciprocal force comes with the gather of the blended potential
  phase_begin(c, PH_GATHER);
  CHK(pme_gather(c, c->d_grid.p, c->d_chg.p, c->d_force_recip.p, c->d_force.p, join_aux));
  phase_end(c, PH_GATHER);
  return MSEVB_OK;
}
