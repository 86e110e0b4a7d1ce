// Real-space pair loop on a FLAT per-cell pair list (replaces pairwise_real_space_verlet, src/pair_int_real_space.f90:135-321, with
// the Ewald table interpolation :698-760, Lennard-Jones :621-645 and - SAPT instantiation - the Buckingham / Tang-Toennies term :651-690; k_pair_half serves > 16 atom types or pruning off).
//
// One 1024-thread workgroup per cell of the neighbour grid, as k_pair_half (pair.hip), but the list the force loop reads is not a set of
// per-atom rows any more:
//   * FLAT LIST.  All pairs (i in the cell, j in the cell's half neighbourhood, r < r_c + delta) of a cell are one contiguous array of
//     32-bit entries { pair-type index (10 bits) | a = i's position in the cell (9 bits) | j = partner's position in the tile (12 bits) },
//     sorted by a, every row padded to a multiple of 16 entries with entries that name a far-away dummy atom.  A trip is 64 consecutive
//     entries: no row cursor, no scalar bookkeeping, every wave of the workgroup gets the same number of trips (a contiguous chunk), and
//     a row's end wastes at most 15 lanes instead of 63 (lane utilisation 96 % against 77 % for 64-entry trips of single rows).
//   * The 16 lanes of a DPP row always belong to ONE atom i: the force on i accumulates in registers across trips and is summed over the
//     DPP row (row_sum16) and added to the LDS force tile only when the row's atom changes (one trip in three).
//   * TILE as structure of arrays in LDS (x[], y[], z[], q[], fx[], fy[], fz[] at a compile-time pitch behind the 8 KB Lennard-Jones table, from LDS
//     address 0: the kernel has no static LDS): 8 j is the address of the two ds_read2st64_b64 of a partner's {x,y,z,q}, 8 j + constant that of the
//     three ds_add_f64 of its force (the array starts are instruction offsets); the (type_i,type_j) Lennard-Jones
//     table holds {C12, C6, 12 C12, 6 C6} (32 B, zeros for pairs without a Lennard-Jones term: no select in the loop).
//   * No validity masks: padding entries point at a tile slot 1e4 A away, r^2 is clamped to r_c^2 for the table look-up, and everything
//     that accumulates sits behind ONE exec mask (r^2 < r_c^2).  r and 1/r come from v_rsq_f64 + one coupled Newton step.
//   * The list is written by the kernel itself on the steps whose displacement test (k_top2_final, neighbor.hip) says the old one no
//     longer covers r_c: the builder's rows are cut into 64-entry pieces, every wave takes an equal share of the pieces, keeps the entries within
//     r_c + delta (single-precision copy of the tile, batched loads) and writes them - converted, padded - straight into ITS segment of the cell's
//     list, which it then reads back in its force loop; the launches that keep the list find their segment in d_cell_trips.
// Software pipeline of the force loop as in k_pair_half: entries of trip t + 2 loading, table rows of trip t + 1 requested, trip t computing.
#include <algorithm>
#include "ctx.hpp"

// PF_EXP (diagnostic builds only, tools/pair_exp.sh): bit 0 no Ewald-table loads, bit 1 no LDS force adds for the partners, bit 2 no Lennard-Jones table reads,
// bit 3 partner positions not read from the tile, bit 8 shader-clock stamps of the kernel's phases (printed by a few workgroups), bit 5 no force loop at all, bit 6 no flush of the force tile - wrong forces, the run time says what each of them costs
#ifndef PF_EXP
#define PF_EXP 0
#endif

namespace {

// (the list entries - 75 MB per launch at 98 k atoms - are loaded with the default cache policy: with the non-temporal hint the stage call took
// 240 instead of 194 us, the list no longer being served from the 256 MB Infinity Cache, where it survives from one step to the next)
__device__ __forceinline__ unsigned pf_ld_entry(const int* p) { return (unsigned)*p; }

// Lennard-Jones table at LDS address 0: 256 pair types x {C12, C6, 12 C12, 6 C6}; the split geometry (pitch 1344, two workgroups per CU) has room for 144
constexpr int pf_lj_bytes(int P) { return P == 1344 ? 4608 : 8192; }
constexpr int PF_J_BITS = 12, PF_A_BITS = 9, PF_A_SHIFT = PF_J_BITS, PF_T_SHIFT = PF_J_BITS + PF_A_BITS;
static_assert((1 << PF_A_BITS) == PAIR_CELL_MAX, "row field of a flat-list entry");

template <int CTRL>
__device__ __forceinline__ double pf_dpp(double v) {
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double pf_row_sum16(double v) {      // sum over the 16 lanes of a DPP row, left in every lane of the row
  v += pf_dpp<0xB1>(v); v += pf_dpp<0x4E>(v); v += pf_dpp<0x141>(v); v += pf_dpp<0x140>(v);
  return v;
}
__device__ __forceinline__ double pf_wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// LDS by ADDRESS: the kernel's only LDS is its dynamic block, which therefore starts at LDS address 0 (checked at kernel entry), and the hot
// loop addresses it with plain integers - through the symbol the compiler adds its (unknown until link time) base to every address in a VALU
// instruction of its own and cannot fold the array starts into the instructions' offset fields.
typedef __attribute__((address_space(3))) double pf_lds_f64;
__device__ __forceinline__ double pf_ld(unsigned addr) { return *(const pf_lds_f64*)(size_t)addr; }
__device__ __forceinline__ void pf_lds_add(unsigned addr, double v) {      // ds_add_f64, no return value
  (void)__hip_atomic_fetch_add((pf_lds_f64*)(size_t)addr, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// address of an atom's fx in the force tile (fy, fz follow at the pitch: instruction offsets) from 8 x its tile position; opaque to the
// compiler, which otherwise folds the three array starts into three separate 32-bit constants and adds each in an instruction of its own
template <int P>
__device__ __forceinline__ unsigned pf_force_addr(unsigned a8) { unsigned f = a8 + (pf_lj_bytes(P) + 32u * P); asm("" : "+v"(f)); return f; }

struct PairFlatArgs {
  int c_lo, c_hi; int3 nc; int tile_cap;
  const int* cell_start; const double4* posq; const int* type_s;
  const unsigned* nbr_count; const unsigned* nbr_split; const int* nbr;      // the builder's rows (tile-local half list, neighbor.hip)
  const PairParam* pairpar; int nt;
  const double2* etab; double inv_dx, dx, delta /* alpha dx */;
  BoxDev box; double rc2;
  double* Fs; double* part_el; double* part_vdw;
  int* prune_state; int* flat; uint2* cell_trips; unsigned stride;
  double4* posq_prune; double rp2; int N_all;
  // self_nb > 0: the kernel takes k_top2_final's decision itself (a launch and a gap less in front of the pair loop): every workgroup reduces the self_nb
  // per-block top-2 displacements since the last rewrite and compares with prune_delta; prune_state[2] carries the NUMBER (epoch) of the launch that wrote
  // the list, so that workgroups starting after workgroup 0 has stamped it still agree with it
  int self_nb, epoch; const double* prd1; const double* prd2; double prune_delta;
  const long long* sb;      // step graph (ctx.hpp): the launch number comes from the device-resident step block (sb[SB_EPOCH]) instead of `epoch` - a captured launch's arguments never change
  const double* tt; const double* dtt;      // Tang-Toennies damping tables (SAPT instantiation)
};

struct PfTrip {      // what stage A leaves for stage B (per lane)
  double dx, dy, dz, r2, qq, inv_r, c1, xb;
  double2 t0, t1;
  unsigned ja, aa, pa;      // byte offsets: 8 j, 8 a, 32 x pair type
};

// Stage A of a trip in two parts, so that the LDS reads of trip t + 1 are under way while trip t's arithmetic (stage B) runs:
//   pf_stage_a_read: decode, partner and owner {x,y,z,q} requested from the tile;
//   pf_stage_a_math: r^2, r and 1/r, the Ewald-table rows requested.
// The table is the reference's: row(r) = ceil(r / dx), value = c1 T[row] + c2 T[row + 1] with c2 = (r/dx + 1) - row (src/pair_int_real_space.f90:751-756;
// T[k] tabulates r = k dx, so the interpolant is taken one interval up - reproduced as it is).
// TAYLOR: ONE row {T_B, E_B} of a second table is fetched instead of two rows {T, S}: both reference tables are built from erfc and its own derivative,
// S = T + x E with E = (2/sqrt(pi)) exp(-x^2) conv = -dT/dx (tables.cpp), and the higher derivatives of T and S are E times polynomials in x, so the
// lower row follows from the upper one to 1e-16 (third-order expansion over one table interval, alpha dx = 2.7e-5).
struct PfRaw { double xj, yj, zj, qj, xi, yi, zi, qi; unsigned ja, aa, pa; };

// (round 5, measured: the row atom's {x, y, z, q} kept in registers and re-read only in trips in which a DPP row moves on - two LDS instructions less in two trips out of
// three, 113 registers instead of 107 - changed nothing: 0.166 against 0.164 ms per stage call.  The LDS is not what a reading launch waits for.)
template <int P>
__device__ __forceinline__ void pf_stage_a_read(unsigned e, PfRaw& w) {
  w.ja = (e & ((1u << PF_J_BITS) - 1u)) << 3;
  w.aa = (e >> (PF_A_SHIFT - 3)) & (((1u << PF_A_BITS) - 1u) << 3);
  w.pa = ((e >> (PF_T_SHIFT - 5)) & (0x3ffu << 5)) | (e & 0x80000000u);      // bit 31: the pair type has a Lennard-Jones term (set by the rewrite)
  constexpr unsigned X = pf_lj_bytes(P);
#if PF_EXP & 8
  w.xj = (double)(w.ja >> 3) * 0.01; w.yj = 3.0; w.zj = 4.0; w.qj = 0.4;
#else
  w.xj = pf_ld(w.ja + X); w.yj = pf_ld(w.ja + X + 8 * P); w.zj = pf_ld(w.ja + X + 16 * P); w.qj = pf_ld(w.ja + X + 24 * P);
#endif
  w.xi = pf_ld(w.aa + X); w.yi = pf_ld(w.aa + X + 8 * P); w.zi = pf_ld(w.aa + X + 16 * P); w.qi = pf_ld(w.aa + X + 24 * P);
}

template <bool TAYLOR>
__device__ __forceinline__ void pf_stage_a_math(const PfRaw& w, double rc2, double inv_dx, double delta, const char* etab, PfTrip& s) {
#pragma clang fp contract(fast)
  s.ja = w.ja; s.aa = w.aa; s.pa = w.pa;
  s.dx = w.xi - w.xj; s.dy = w.yi - w.yj; s.dz = w.zi - w.zj;
  s.r2 = s.dx * s.dx + s.dy * s.dy + s.dz * s.dz;
  s.qq = w.qi * w.qj;
  const double a = fmin(s.r2, rc2);
  // r = sqrt(a), 1/r: v_rsq_f64 seed + ONE coupled Newton (Goldschmidt) step: relative error ~1e-15, far inside the tolerances of DESIGN.md 5
  const double y0 = __builtin_amdgcn_rsq(a);
  double g = a * y0, h = 0.5 * y0;
  const double d = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, d, g); h = __builtin_fma(h, d, h);
  s.inv_r = h + h;
  const double x1 = g * inv_dx, cx = ceil(x1);
  s.c1 = cx - x1;
  if (TAYLOR) s.xb = __builtin_fma(cx, delta, delta);      // alpha r_B, r_B = (row + 1) dx
  const int row = max((int)cx, 1);      // (r = 0 / NaN must not read in front of the table; the table has one spare row behind r_c)
  const unsigned off = (unsigned)row << 4;
#if PF_EXP & 1
  s.t1 = make_double2((double)off, 2.0); s.t0 = make_double2(1.0, (double)off);
#else
  s.t1 = *(const double2*)(etab + off);
  if (!TAYLOR) s.t0 = *(const double2*)(etab + off - 16);
#endif
}

__device__ __forceinline__ int pf_tile_to_sorted(int p, const int* s_lo, const int* s_off, int ns) {
  int s = 0;
  while (s + 1 < ns && p >= s_off[s + 1]) ++s;
  return s_lo[s] + (p - s_off[s]);
}

// the reference's own r^2 (src/pair_int_real_space.f90:222-226), for pairs within rounding of the cutoff
__device__ __forceinline__ double pf_exact_r2(const double4& pi, const double4& pj, const BoxDev& box) {
#pragma clang fp contract(off)
  double dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
  dx = dx - box.L[0] * floor(dx * box.invL[0] + 0.5); dy = dy - box.L[1] * floor(dy * box.invL[1] + 0.5); dz = dz - box.L[2] * floor(dz * box.invL[2] + 0.5);
  return (dx * dx + dy * dy) + dz * dz;
}

struct PfSmall { double red[2][16]; int lo[HALF_SLOTS], off[HALF_SLOTS + 1], full, off7; unsigned tt0, tt; unsigned ljm[16] /* per atom type: bit t set = the pair with type t has a van der Waals term (Lennard-Jones, or SAPT with a non-zero coefficient) */; unsigned char vkind[256] /* per pair type: 1 = SAPT */; unsigned char typ[PAIR_CELL_MAX]; };
struct PfRows { unsigned off2[PAIR_CELL_MAX], cnt[PAIR_CELL_MAX], beg[PAIR_CELL_MAX]; };      // per-row scratch of the pruning pass: in the third force array, behind the single-precision copy of the tile

struct PfAcc { double fx, fy, fz, e_el, e_vdw; unsigned a_prev;
};

template <int P>
__device__ __forceinline__ void pf_flush_row(PfAcc& acc, int lane) {      // every active lane's DPP row is complete
  const double sx = pf_row_sum16(acc.fx), sy = pf_row_sum16(acc.fy), sz = pf_row_sum16(acc.fz);
  if ((lane & 15) == 0) { const unsigned f = pf_force_addr<P>(acc.a_prev); pf_lds_add(f, sx); pf_lds_add(f + 8 * P, sy); pf_lds_add(f + 16 * P, sz); }
  acc.fx = 0.0; acc.fy = 0.0; acc.fz = 0.0;
}

// Stage B: interaction arithmetic of a trip whose table rows were requested a trip ago; LDS force adds.
// argument of the Tang-Toennies row lookup in the reference's arithmetic (vdw_interaction, devmath.hpp: r = sqrt(r^2) of the raw coordinates)
__device__ __forceinline__ double pf_exact_tt_arg(const double4& pi, const double4& pj, const BoxDev& box, double B) {
#pragma clang fp contract(off)
  const double r = sqrt(pf_exact_r2(pi, pj, box));
  return B * r / TT_MAX * (double)TT_GRID;
}

template <int P, bool TAYLOR, bool SAPT>
__device__ __forceinline__ void pf_stage_b(const PfTrip& s, int lane, double rc2, double rc2_lo, double rc2_hi, double dxg, double delta, const BoxDev& box,
                                           const double4* __restrict__ posq, int cs, const int* s_lo, const int* s_off, int ns, PfAcc& acc,
                                           const unsigned char* vkind, const PairParam* __restrict__ pairpar, const double* __restrict__ tt, const double* __restrict__ dtt) {
#pragma clang fp contract(fast)
  if (__ballot(s.aa != acc.a_prev)) {      // (wave-uniform test) some DPP row has moved on to its next atom: its owner's force joins the tile
    if (s.aa != acc.a_prev) { pf_flush_row<P>(acc, lane); acc.a_prev = s.aa; }
  }
  double ev, sv;      // erfc and force-scale interpolants
  if (TAYLOR) {
    const double gE = (s.c1 * delta) * s.t1.y;        // alpha (r_B - r - dx) E_B: the fetched row is {T_B, E_B}, E = (2/sqrt(pi)) exp(-x^2) conv = -dT/dx (d_ewald_tay)
    const double x = s.xb, d2 = delta * delta;
    const double bt = __builtin_fma(x, __builtin_fma(x, d2 * (2.0 / 3.0), delta), 1.0 - d2 * (1.0 / 3.0));
    const double bs = __builtin_fma(x, __builtin_fma(x, __builtin_fma(x, __builtin_fma(x, d2 * (4.0 / 3.0), 2.0 * delta), 2.0 - d2 * (10.0 / 3.0)), -2.0 * delta), d2 * (2.0 / 3.0));
    ev = __builtin_fma(gE, bt, s.t1.x);
    sv = __builtin_fma(gE, bs, __builtin_fma(x, s.t1.y, s.t1.x));      // S_B = T_B + x_B E_B
  } else {
    const double c2 = 1.0 - s.c1;
    ev = s.c1 * s.t0.x + c2 * s.t1.x;
    sv = s.c1 * s.t0.y + c2 * s.t1.y;
  }
  const double inv_r2 = s.inv_r * s.inv_r, qr = s.qq * s.inv_r;
  double evdw = 0.0, flj = 0.0;
  // Lennard-Jones only in trips that hold a pair with such a term (wave-uniform test of the entries' flag bit): in water two rows in three belong
  // to hydrogens, whose pairs have none - their trips skip four LDS reads and eight instructions per entry
  if (__ballot((int)s.pa < 0)) {
    const unsigned pa = s.pa & 0x7fffffffu;
#if PF_EXP & 4
    const double C12 = (double)pa, C6 = 2.0, C12x12 = 3.0, C6x6 = 4.0;
#else
    const double C12 = pf_ld(pa), C6 = pf_ld(pa + 8), C12x12 = pf_ld(pa + 16), C6x6 = pf_ld(pa + 24);      // (zeros for a pair type without the term)
#endif
    const double r6i = inv_r2 * inv_r2 * inv_r2;
    evdw = __builtin_fma(C12, r6i, -C6) * r6i;                                  // C12/r^12 - C6/r^6
    flj = __builtin_fma(C12x12, r6i, -C6x6) * r6i * inv_r2;
    if (SAPT) {      // Buckingham + Tang-Toennies damped dispersion (pairwise_real_space_sapt, src/pair_int_real_space.f90:651-690; vdw_interaction of devmath.hpp): pair types of kind 1,
      const bool sp = vkind[pa >> 5] == 1;      // whose Lennard-Jones row is zero; their six coefficients come from the parameter table in global memory (a handful of 64-byte rows: cache hits)
      if (__ballot(sp)) {
        if (sp) {
          const PairParam& pp = pairpar[pa >> 5];
          const double r = s.r2 * s.inv_r, A = pp.p[0], B = pp.p[1];      // (r^2 < r_c^2 for every lane that accumulates: inv_r is 1/r there)
          // The damping table is a STEP function of B r (row = ceil(B r / 50 * 1000), no interpolation: the reference's): a pair whose B r sits on a row
          // boundary to within rounding - lattice-like distances, 3.5 A at B = 3.5 - takes the row the REFERENCE's arithmetic on the raw coordinates
          // gives (as the cutoff test above does); the tile coordinates differ from those in the last bits
          double xi = B * r / TT_MAX * (double)TT_GRID;
          const bool tie = fabs(xi - rint(xi)) <= 1e-9 * fmax(xi, 1.0);
          if (__builtin_expect(tie, 0)) xi = pf_exact_tt_arg(posq[cs + (int)(s.aa >> 3)], posq[pf_tile_to_sorted((int)(s.ja >> 3), s_lo, s_off, ns)], box, B);
          const int ti = min(max((int)ceil(xi), 1), TT_GRID);
          const double* t = tt + 4 * (ti - 1);
          const double* dt = dtt + 4 * (ti - 1);
          const double r8i = r6i * inv_r2, r10i = r8i * inv_r2, r12i = r10i * inv_r2;
          const double ex = exp(-B * r);
          evdw = A * ex - t[0] * pp.p[2] * r6i - t[1] * pp.p[3] * r8i - t[2] * pp.p[4] * r10i - t[3] * pp.p[5] * r12i;
          const double fac = r * A * B * ex + (r * B * dt[0] - 6.0 * t[0]) * pp.p[2] * r6i + (r * B * dt[1] - 8.0 * t[1]) * pp.p[3] * r8i +
                             (r * B * dt[2] - 10.0 * t[2]) * pp.p[4] * r10i + (r * B * dt[3] - 12.0 * t[3]) * pp.p[5] * r12i;
          flj = inv_r2 * fac;
        }
      }
    }
  }
  const double fs = __builtin_fma(qr * inv_r2, sv, flj);      // force on i = fs * dr
  bool in = s.r2 < rc2_lo;
  const unsigned long long m_in = __ballot(in), m_near = __ballot(s.r2 < rc2_hi);
  if (m_in != m_near) {      // rare: on the cutoff to within rounding - the reference's arithmetic on the raw coordinates decides
    if (((m_near & ~m_in) >> lane) & 1ull) in = pf_exact_r2(posq[cs + (int)(s.aa >> 3)], posq[pf_tile_to_sorted((int)(s.ja >> 3), s_lo, s_off, ns)], box) < rc2;
  }
  if (in) {
    const double gx = fs * s.dx, gy = fs * s.dy, gz = fs * s.dz;
    acc.fx += gx; acc.fy += gy; acc.fz += gz;
    acc.e_el = __builtin_fma(qr, ev, acc.e_el); acc.e_vdw += evdw;
#if PF_EXP & 2
    acc.e_vdw += gx + gy + gz;
#else
    const unsigned f = pf_force_addr<P>(s.ja);
    pf_lds_add(f, -gx); pf_lds_add(f + 8 * P, -gy); pf_lds_add(f + 16 * P, -gz);
#endif
  }
}

// LDS (all dynamic, from address 0): Lennard-Jones table (8 KB; 4 KB in the split geometry); then positions x, y, z, q and the force tile fx, fy, fz (P doubles
// each; slot ntile = the dummy partner); then the small arrays of PfSmall.
// SPLIT: TWO workgroups per cell - workgroup h = 0 works on the pairs whose partner lies in the cell itself or in the first 6 forward cells, h = 1 on the
// last 7 forward cells (the builder's rows are in tile order and carry the position of the cut, d_nbr_split) - so that a sub-tile (the cell + 6 or 7 cells,
// ~1 100 atoms) needs < 80 KB and two workgroups share a CU: 24 waves per CU instead of 16, and one workgroup stages or flushes while the other computes.
template <int P, bool TAYLOR, bool SPLIT, int THREADS, int MINW, bool PIPE, bool SAPT = false>
__global__ __launch_bounds__(THREADS, MINW) void k_pair_flat(const PairFlatArgs A) {
  constexpr int NS = SPLIT ? 8 : HALF_SLOTS;      // cells in the (sub-)tile
  constexpr int LJB = pf_lj_bytes(P);
  extern __shared__ double4 pf_lds4[];
  char* lds = (char*)pf_lds4;
  PfSmall& sm = *(PfSmall*)(lds + LJB + 56 * P);
  int* s_lo = sm.lo; int* s_off = sm.off; unsigned char* s_typ = sm.typ;
  static_assert(sizeof(PfRows) <= 8 * P, "pruning scratch fits the third force array");
  PfRows& pr = *(PfRows*)(lds + LJB + 32 * P + 16 * P);
  unsigned* s_off2 = pr.off2; unsigned* s_cnt = pr.cnt; unsigned* s_beg = pr.beg;
  if ((unsigned)(size_t)(__attribute__((address_space(3))) char*)pf_lds4 != 0u) __builtin_trap();      // (see pf_ld)
#if PF_EXP & 256
  const unsigned long long ts0 = __builtin_amdgcn_s_memtime();
#endif
  int mode;
  if (A.self_nb > 0) {
    const int w = A.prune_state[2];
    double m1 = 0.0, m2 = 0.0;      // same top-2 merge as k_top2_final (maxima: the order does not matter, same bits)
    for (int k = threadIdx.x; k < A.self_nb; k += THREADS) {
      const double b1 = A.prd1[k], b2 = A.prd2[k];
      const double n1 = fmax(m1, b1), n2 = fmax(fmin(m1, b1), fmax(m2, b2));
      m1 = n1; m2 = n2;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const double b1 = __shfl_xor(m1, o, 64), b2 = __shfl_xor(m2, o, 64);
      const double n1 = fmax(m1, b1), n2 = fmax(fmin(m1, b1), fmax(m2, b2));
      m1 = n1; m2 = n2;
    }
    if ((threadIdx.x & 63) == 0) { sm.red[0][threadIdx.x >> 6] = m1; sm.red[1][threadIdx.x >> 6] = m2; }
    __syncthreads();
    m1 = sm.red[0][0]; m2 = sm.red[1][0];
#pragma unroll
    for (int k = 1; k < THREADS / 64; ++k) {
      const double b1 = sm.red[0][k], b2 = sm.red[1][k];
      const double n1 = fmax(m1, b1), n2 = fmax(fmin(m1, b1), fmax(m2, b2));
      m1 = n1; m2 = n2;
    }
    __syncthreads();      // (sm.red is the energy reduction's scratch later on)
    mode = __builtin_amdgcn_readfirstlane((w != 0 && w != (A.sb ? (int)A.sb[SB_EPOCH] : A.epoch) && (m1 + m2) < A.prune_delta) ? 2 : 1);
  } else mode = __builtin_amdgcn_readfirstlane(A.prune_state[0]);      // 2: the flat list still covers r_c; otherwise it is rewritten first
  if (mode != 2) {      // snapshot of ALL positions (the displacement check runs over every atom, also when the cells are sharded over ranks)
    const int chunk = (A.N_all + gridDim.x - 1) / gridDim.x, k0 = blockIdx.x * chunk, k1 = min(A.N_all, k0 + chunk);
    for (int k = k0 + threadIdx.x; k < k1; k += blockDim.x) A.posq_prune[k] = A.posq[k];
    if (blockIdx.x == 0 && threadIdx.x == 0) A.prune_state[2] = A.sb ? (int)A.sb[SB_EPOCH] : A.epoch;      // a list exists for the current rows (k_top2_final / the next launches look), written by launch `epoch`
  }
  const int lb = xcd_block(blockIdx.x, gridDim.x);      // one XCD works on one contiguous eighth of the cells (a slab)
  const int cidx = A.c_lo + (SPLIT ? lb >> 1 : lb), h = SPLIT ? (lb & 1) : 0;
  if (cidx >= A.c_hi) { if (threadIdx.x == 0) { A.part_el[blockIdx.x] = 0.0; A.part_vdw[blockIdx.x] = 0.0; } return; }
  if (threadIdx.x < 64) {      // the cells of the (sub-)tile: first atom and tile offset of each (prefix sums across the lanes)
    int slots[HALF_SLOTS];
    cell_half_neighbourhood(cidx, A.nc.x, A.nc.y, A.nc.z, slots);
    const int tid = threadIdx.x, sl = min(tid, HALF_SLOTS - 1);
    const int lo = A.cell_start[slots[sl]], n = tid < HALF_SLOTS ? A.cell_start[slots[sl] + 1] - lo : 0;
    const bool mine = tid < HALF_SLOTS && (!SPLIT || (h == 0 ? tid <= 6 : (tid == 0 || tid >= 7)));
    const int nm = mine ? n : 0;
    int incf = n, incm = nm;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) { const int t = __shfl_up(incf, o, 64), u = __shfl_up(incm, o, 64); if (tid >= o) { incf += t; incm += u; } }
    const unsigned long long mm = __ballot(mine);
    const int k = __popcll(mm & ((1ull << tid) - 1ull));      // position of this cell in the (sub-)tile
    if (mine) { s_lo[k] = lo; s_off[k] = incm - nm; }
    if (tid == HALF_SLOTS - 1) { sm.full = incf; s_off[NS] = incm; if (SPLIT && h == 0) { s_lo[NS - 1] = 0; s_off[NS - 1] = incm; } }
    if (tid == 6) sm.off7 = incf;
  }
  __syncthreads();
  const int ntile = s_off[NS], na = s_off[1], cs = s_lo[0];
  if (sm.full > A.tile_cap || ntile >= P || na > PAIR_CELL_MAX) {      // larger than the tile this list was built for: the host sees it in the build statistics and repeats the call
    if (threadIdx.x == 0) { A.part_el[blockIdx.x] = 0.0; A.part_vdw[blockIdx.x] = 0.0; }
    return;
  }
  double* tp = (double*)(lds + LJB); double* tf = tp + 4 * P;
  // tile position -> sorted atom: p + (lo[s] - off[s]) for the cell s that holds p, as a sum of selects on scalar operands (a search through the LDS
  // copies is a chain of dependent reads per atom)
  int t2s_off[NS], t2s_add[NS];
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    t2s_off[k] = __builtin_amdgcn_readfirstlane(s_off[k]);
    const int d = __builtin_amdgcn_readfirstlane(s_lo[k] - s_off[k]);
    t2s_add[k] = k ? d - __builtin_amdgcn_readfirstlane(s_lo[k - 1] - s_off[k - 1]) : d;
  }
  auto tile_to_sorted = [&](int p) -> int {
    int j = p + t2s_add[0];
#pragma unroll
    for (int k = 1; k < NS; ++k) j += p >= t2s_off[k] ? t2s_add[k] : 0;
    return j;
  };
  {      // stage the tile: coordinates relative to the cell centre, at the periodic image nearest to it (>= 4 cells per axis: every tile atom is within 2 cells of it)
    const int cx = cidx % A.nc.x, cy = (cidx / A.nc.x) % A.nc.y, cz = cidx / (A.nc.x * A.nc.y);
    const double ox = (cx + 0.5) * A.box.L[0] / A.nc.x, oy = (cy + 0.5) * A.box.L[1] / A.nc.y, oz = (cz + 0.5) * A.box.L[2] / A.nc.z;
    for (int p = threadIdx.x; p <= ntile; p += blockDim.x) {
      double4 t = make_double4(1.0e4, 1.0e4, 1.0e4, 0.0);      // slot ntile: the partner of the padding entries
      if (p < ntile) {
        const double4 q = A.posq[tile_to_sorted(p)];
        t.x = q.x - ox; t.x = t.x - A.box.L[0] * floor(t.x * A.box.invL[0] + 0.5);
        t.y = q.y - oy; t.y = t.y - A.box.L[1] * floor(t.y * A.box.invL[1] + 0.5);
        t.z = q.z - oz; t.z = t.z - A.box.L[2] * floor(t.z * A.box.invL[2] + 0.5);
        t.w = q.w;
      }
      tp[p] = t.x; tp[P + p] = t.y; tp[2 * P + p] = t.z; tp[3 * P + p] = t.w;
      if (mode != 2) ((float4*)tf)[p] = make_float4((float)t.x, (float)t.y, (float)t.z, 0.0f);      // single-precision copy for the pruning pass, in the (still unused) force tile
      else { tf[p] = 0.0; tf[P + p] = 0.0; tf[2 * P + p] = 0.0; }
    }
    for (int a = threadIdx.x; a < na; a += blockDim.x) s_typ[a] = (unsigned char)A.type_s[cs + a];
    double4* ppt = (double4*)lds;
    for (int k = threadIdx.x; k < A.nt * A.nt; k += blockDim.x) {
      const PairParam q = A.pairpar[k];
      ppt[k] = q.type == 0 ? make_double4(q.p[0], q.p[1], 12.0 * q.p[0], 6.0 * q.p[1]) : make_double4(0.0, 0.0, 0.0, 0.0);
    }
    if (mode != 2 && (int)threadIdx.x < A.nt) {
      unsigned m = 0;
      for (int t = 0; t < A.nt; ++t) { const PairParam q = A.pairpar[threadIdx.x * A.nt + t]; if ((q.type == 0 && (q.p[0] != 0.0 || q.p[1] != 0.0)) || (SAPT && q.type == 1 && q.nonzero)) m |= 1u << t; }
      sm.ljm[threadIdx.x] = m;
    }
    if (SAPT) for (int k = threadIdx.x; k < A.nt * A.nt; k += blockDim.x) { const PairParam q = A.pairpar[k]; sm.vkind[k] = (q.type == 1 && q.nonzero) ? 1 : 0; }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int wpb = THREADS / 64;
#if PF_EXP & 256
  const unsigned long long ts1 = __builtin_amdgcn_s_memtime();
#endif
  // The cell's list is one SEGMENT per wave (of either workgroup): a wave writes its segment (when the list is rewritten) and reads it back in its force loop,
  // here and in the launches that keep the list - no scratch copy.  Room: the cell's rows at (stride + 96) + 3584.
  int* cell_flat = A.flat + ((size_t)cs * (A.stride + 96u) + 3584 * (size_t)cidx);
  uint2* meta = A.cell_trips + ((size_t)cidx * 2 + h) * 16 + wave;
  unsigned seg_off, seg_trips;
  if (mode != 2) {
    // ---- the builder's rows (this workgroup's part of each) cut into 64-entry pieces, dealt to the waves in equal numbers; exclusive prefix of the pieces per row
    if (wave == 0) {      // 8 rows per lane, shuffles across the wave
      unsigned loc[8], sum = 0, sum0 = 0;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int a = 8 * lane + q;
        unsigned n = 0, b = 0;
        if (a < na) {
          const unsigned nall = A.nbr_count[cs + a], sp = SPLIT ? min(A.nbr_split[cs + a], nall) : 0u;
          n = SPLIT ? (h ? nall - sp : sp) : nall; b = (SPLIT && h) ? sp : 0u;
          s_cnt[a] = n; s_beg[a] = b;
          if (SPLIT && h) sum0 += (sp + 63u) >> 6;      // (pieces of the other workgroup: its segments come first in the cell's list)
        }
        loc[q] = (n + 63u) >> 6; sum += loc[q];
      }
      unsigned inc = sum;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { const unsigned t = (unsigned)__shfl_up((int)inc, o, 64); if (lane >= o) inc += t; }
      unsigned run = inc - sum;
#pragma unroll
      for (int q = 0; q < 8; ++q) { const int a = 8 * lane + q; if (a < na) s_off2[a] = run; run += loc[q]; }
      if (lane == 63) sm.tt = inc;
      if (SPLIT && h) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sum0 += (unsigned)__shfl_xor((int)sum0, o, 64);
        if (lane == 0) sm.tt0 = sum0;
      }
    }
    __syncthreads();
#if PF_EXP & 256
    const unsigned long long tsA = __builtin_amdgcn_s_memtime();
#endif
    const unsigned TT = sm.tt;
    unsigned ft = (unsigned)(((unsigned long long)TT * (unsigned)wave) / (unsigned)wpb);
    const unsigned ft1 = (unsigned)(((unsigned long long)TT * (unsigned)(wave + 1)) / (unsigned)wpb);
    int a = 0;
    if (ft < ft1) {      // the row that holds piece ft: the last one whose prefix is <= ft
      int lo = 0, hi = na;
      while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (s_off2[mid] <= ft) lo = mid; else hi = mid; }
      a = lo;
    }
    // Upper bound of what the waves before this one write: their pieces (64 each), 15 padding entries per row part (a rows + one split per wave), 63 per
    // segment end; in front of workgroup 1's segments, all of workgroup 0's
    const unsigned base_h = (SPLIT && h) ? 64u * (sm.tt0 + (unsigned)wpb + (((unsigned)na + (unsigned)wpb + 3u) >> 2)) : 0u;
    seg_off = base_h + 64u * (ft + (unsigned)wave + (((unsigned)a + (unsigned)wave + 3u) >> 2));
    int* out = cell_flat + seg_off;
    // The wave's pieces as a stream {row, offset in the row}; PF_BATCH pieces are loaded together and the next batch is requested before the current one is
    // looked at (the rows come from HBM: 2 x 4 pieces of 256 B per wave in flight, 32 KB per CU; twice as many measured the same).
    constexpr int PF_BATCH = 4;
    const float4* tq = (const float4*)tf;
    const float rp2f = (float)A.rp2 + 2e-3f;
    struct Piece { int a; unsigned k0, n; };      // (wave-uniform)
    int pa = a; unsigned pk = ft < ft1 ? (ft - s_off2[a]) << 6 : 0u, pn = ft < ft1 ? s_cnt[a] : 0u, left = ft1 - ft;
    const int* rows = A.nbr + (size_t)cs * A.stride;      // (row i of the builder starts at i x stride)
    const unsigned jshift = (SPLIT && h) ? (unsigned)(sm.off7 - na) : 0u;      // tile position -> sub-tile position for the partners of workgroup 1
    auto next_piece = [&](Piece& pc) -> unsigned {      // returns the piece's entry of this lane (0xffffffff: none)
      if (left == 0) { pc.a = -1; pc.k0 = 0; pc.n = 0; return 0xffffffffu; }
      while (pk >= pn) { ++pa; pk = 0; pn = s_cnt[pa]; }      // (rows without entries hold no piece)
      pc.a = pa; pc.k0 = pk; pc.n = pn;
      const unsigned k = pk + lane;
      const unsigned e = k < pn ? (unsigned)rows[(size_t)pa * A.stride + s_beg[pa] + k] : 0xffffffffu;
      pk += 64; --left;
      return e;
    };
    Piece pc[PF_BATCH], pd[PF_BATCH]; unsigned ec[PF_BATCH], ed[PF_BATCH];
#pragma unroll
    for (int q = 0; q < PF_BATCH; ++q) ec[q] = next_piece(pc[q]);
    unsigned o = 0, tot = 0;
    int a_last = a;
    while (pc[0].a >= 0) {
#pragma unroll
      for (int q = 0; q < PF_BATCH; ++q) ed[q] = next_piece(pd[q]);      // the batch after this one: loads issued
#pragma unroll
      for (int q = 0; q < PF_BATCH; ++q) {
        if (pc[q].a < 0) break;
        const int ra = pc[q].a;
        const unsigned e = ec[q];
        const bool v = e != 0xffffffffu;
        const unsigned j = v ? (e & NBR_IDX_MASK) - jshift : (unsigned)ntile;
        const float4 fi = tq[ra], fj = tq[j];      // tile coordinates are within 2.5 cells of the cell centre: r^2 is good to 2e-4 A^2 in single precision,
        const float ddx = fi.x - fj.x, ddy = fi.y - fj.y, ddz = fi.z - fj.z;      // and an entry too many in the list costs nothing but its lane
        const bool keep = v && (ddx * ddx + ddy * ddy + ddz * ddz) < rp2f;
        const unsigned long long m = __ballot(keep);
        const unsigned ti = (unsigned)s_typ[ra], ljm = sm.ljm[ti];
        const unsigned head = (ti * (unsigned)A.nt) << PF_T_SHIFT | ((unsigned)ra << PF_A_SHIFT);
        if (keep) out[o + tot + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u))] = (int)((head + ((e >> NBR_IDX_BITS) << PF_T_SHIFT) + j) | (((ljm >> (e >> NBR_IDX_BITS)) & 1u) << 31));
        tot += (unsigned)__popcll(m);
        const int na_next = q + 1 < PF_BATCH ? pc[q + 1].a : pd[0].a;      // the row of the piece that follows (-1: none)
        if (na_next != ra) {      // the row part ends: padded to 16 entries (dummy partner, same row)
          const unsigned np = (tot + 15u) & ~15u;
          if (tot + lane < np) out[o + tot + lane] = (int)(((unsigned)ra << PF_A_SHIFT) + (unsigned)ntile);
          o += np; tot = 0; a_last = ra;
        }
      }
#pragma unroll
      for (int q = 0; q < PF_BATCH; ++q) { pc[q] = pd[q]; ec[q] = ed[q]; }
    }
    seg_trips = (o + 63u) >> 6;
    if (o + lane < (seg_trips << 6)) out[o + lane] = (int)(((unsigned)a_last << PF_A_SHIFT) + (unsigned)ntile);
    if (lane == 0) *meta = make_uint2(seg_off, seg_trips);
#if PF_EXP & 256
    const unsigned long long tsB = __builtin_amdgcn_s_memtime();
#endif
    __syncthreads();      // (the stores of the wave's segment have left it, which is all its own reads of the segment need: same CU, write-through L1)
    for (int p = threadIdx.x; p <= ntile; p += blockDim.x) { tf[p] = 0.0; tf[P + p] = 0.0; tf[2 * P + p] = 0.0; }      // the force tile held the single-precision copy
    __syncthreads();
#if PF_EXP & 256
    if (lane == 0 && (wave == 0 || wave == 15) && (blockIdx.x % 97) == 5)
      printf("pf diag REWRITE block %d wave %d pieces %u: row prefix %llu clk, pieces %llu, wait + zero the force tile %llu\n", blockIdx.x, wave, ft1 - ft, tsA - ts1, tsB - tsA, __builtin_amdgcn_s_memtime() - tsB);
#endif
  } else {
    const uint2 sg = *meta;
    seg_off = sg.x; seg_trips = sg.y;
  }
  // ---- force loop over the wave's segment
  const unsigned t0 = 0u, t1 = seg_trips;
  const double rc2_lo = A.rc2 * (1.0 - 1e-9), rc2_hi = A.rc2 * (1.0 + 1e-9);
  const char* etab = (const char*)A.etab;
  PfAcc acc;
  acc.fx = 0.0; acc.fy = 0.0; acc.fz = 0.0; acc.e_el = 0.0; acc.e_vdw = 0.0; acc.a_prev = 0u;
  if (t0 < t1 && !(PF_EXP & 32)) {
    // Entry loads are UNCONDITIONAL (behind the wave's last trip they read the next wave's entries or the list's padding - the allocation has trips to
    // spare - and a select discards them): a load inside a branch makes the compiler wait for ALL outstanding loads at the join, table rows included.
    const int* src = cell_flat + seg_off + lane;
    const unsigned e_none = (unsigned)ntile;
    PfRaw w; PfTrip sa;
    if (PIPE) {
      // Two trips in flight per wave: geometry of trip t + 1 (its table rows requested) in front of the arithmetic of trip t, whose rows were requested a
      // trip ago; the entries of trip t + 2 loading.  (A third stage - tile reads of trip t + 2 issued before the arithmetic of trip t - measured the same
      // and costs 20 registers: the loop is bound by the throughput of the LDS, the vector-memory address path and the fp64 pipe together, not by a wait.)
      PfTrip sb;
      unsigned ea = pf_ld_entry(src + 64), eb;
      pf_stage_a_read<P>(pf_ld_entry(src), w);
      ea = t0 + 1 < t1 ? ea : e_none;
      pf_stage_a_math<TAYLOR>(w, A.rc2, A.inv_dx, A.delta, etab, sa);
      acc.a_prev = sa.aa;
      for (unsigned t = t0; t < t1; t += 2) {      // two trips per turn: the states swap roles instead of being copied
        eb = pf_ld_entry(src + 128);                                                                                  // entries of trip t + 2
        pf_stage_a_read<P>(ea, w);
        pf_stage_a_math<TAYLOR>(w, A.rc2, A.inv_dx, A.delta, etab, sb);                                               // trip t + 1: geometry, table rows requested (behind the last trip: padding)
        pf_stage_b<P, TAYLOR, SAPT>(sa, lane, A.rc2, rc2_lo, rc2_hi, A.dx, A.delta, A.box, A.posq, cs, s_lo, s_off, NS, acc, sm.vkind, A.pairpar, A.tt, A.dtt);      // trip t: arithmetic
        eb = t + 2 < t1 ? eb : e_none;
        if (t + 1 >= t1) break;
        ea = pf_ld_entry(src + 192);
        pf_stage_a_read<P>(eb, w);
        pf_stage_a_math<TAYLOR>(w, A.rc2, A.inv_dx, A.delta, etab, sa);
        pf_stage_b<P, TAYLOR, SAPT>(sb, lane, A.rc2, rc2_lo, rc2_hi, A.dx, A.delta, A.box, A.posq, cs, s_lo, s_off, NS, acc, sm.vkind, A.pairpar, A.tt, A.dtt);
        ea = t + 3 < t1 ? ea : e_none;
        src += 128;
      }
    } else {
      // One trip at a time (80 registers: six waves per SIMD): the table rows are requested and waited for inside the trip, the other waves cover the wait
      unsigned e = pf_ld_entry(src);
      bool first = true;
      for (unsigned t = t0; t < t1; ++t) {
        const unsigned en = pf_ld_entry(src + 64);      // entries of the next trip
        pf_stage_a_read<P>(e, w);
        pf_stage_a_math<TAYLOR>(w, A.rc2, A.inv_dx, A.delta, etab, sa);
        if (first) { acc.a_prev = sa.aa; first = false; }
        pf_stage_b<P, TAYLOR, SAPT>(sa, lane, A.rc2, rc2_lo, rc2_hi, A.dx, A.delta, A.box, A.posq, cs, s_lo, s_off, NS, acc, sm.vkind, A.pairpar, A.tt, A.dtt);
        e = en; src += 64;
      }
    }
    pf_flush_row<P>(acc, lane);
  }
#if PF_EXP & 256
  const unsigned long long ts2 = __builtin_amdgcn_s_memtime();
#endif
  const double e_el = pf_wave_sum(acc.e_el), e_vdw = pf_wave_sum(acc.e_vdw);
  if (lane == 0) { sm.red[0][wave] = e_el; sm.red[1][wave] = e_vdw; }
  __syncthreads();      // ... and every wave's LDS force adds are done
#if PF_EXP & 256
  const unsigned long long ts3 = __builtin_amdgcn_s_memtime();
#endif
  if (threadIdx.x == 0) {
    double a_ = 0, b_ = 0;
    for (int w = 0; w < wpb; ++w) { a_ += sm.red[0][w]; b_ += sm.red[1][w]; }
    A.part_el[blockIdx.x] = a_;      // half list: every pair once
    A.part_vdw[blockIdx.x] = b_;
  }
  // flush the force tile: runs of consecutive sorted atoms per cell, one coalesced fp64 atomic per component (partners no pair reached stay out)
  if (!(PF_EXP & 64))
  for (int p = threadIdx.x; p < ntile; p += blockDim.x) {
    const double vx = tf[p], vy = tf[P + p], vz = tf[2 * P + p];
    if (vx != 0.0 || vy != 0.0 || vz != 0.0) {
      const int j = tile_to_sorted(p);
      atomicAdd(&A.Fs[j], vx); atomicAdd(&A.Fs[(size_t)A.N_all + j], vy); atomicAdd(&A.Fs[2 * (size_t)A.N_all + j], vz);
    }
  }
#if PF_EXP & 256
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane == 0 && (wave == 0 || wave == wpb - 1) && (blockIdx.x % 97) == 5)
    printf("pf diag block %d mode %d wave %d trips %u: staging %llu clk, loop %llu, wait for the other waves %llu, flush %llu\n", blockIdx.x, mode, wave, t1, ts1 - ts0, ts2 - ts1, ts3 - ts2, __builtin_amdgcn_s_memtime() - ts3);
#endif
}

template <int P, bool TAYLOR, bool SPLIT, int THREADS, int MINW, bool PIPE, bool SAPT = false>
int pf_launch(msevb_ctx* c, const PairFlatArgs& A, int nblocks, bool& attr_done) {
  auto kern = k_pair_flat<P, TAYLOR, SPLIT, THREADS, MINW, PIPE, SAPT>;
  constexpr int lds = pf_lj_bytes(P) + 56 * P + (int)sizeof(PfSmall);
  if (!attr_done) {
    HIPCHK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, dim3(nblocks), dim3(THREADS), lds, c->stream, A);
  if (c->prune_freeze) HIPCHK(hipMemsetD32Async((hipDeviceptr_t)c->d_prune_mode.p, 2, 1, c->stream));      // diagnostics: the next stage call reads this list
  return MSEVB_OK;
}

}  // namespace

// the flat-list kernel serves: half list on tiles, rolling pruning on, no live SAPT term, <= 16 atom types, tile + dummy slot within the largest pitch
bool pair_flat_usable(const msevb_ctx* c) {
  return c->pair_flat_opt != 0 && c->pair_tiles && (!c->has_sapt || c->pair_taylor) && c->nt <= 16 && c->half_tile_cap + 1 <= 2560 && c->prune && c->rc + c->prune_delta < c->rv;
}
// two workgroups per cell: the sub-tiles the last build saw (+ 3 %) fit the 1344-atom pitch, and the Lennard-Jones table 4.5 KB
bool pair_flat_split(const msevb_ctx* c) {
  return c->pair_split_opt != 0 && !c->has_sapt && c->nt <= 12 && c->sub_tile_max > 0 && (int)(c->sub_tile_max * 1.03) + 1 < 1344;
}

static void pf_args(msevb_ctx* c, int c_lo, int c_hi, int slot_elec, int slot_vdw, PairFlatArgs& A) {
  const int N = c->N;
  A.c_lo = c_lo; A.c_hi = c_hi; A.nc = make_int3(c->ncell[0], c->ncell[1], c->ncell[2]); A.tile_cap = c->half_tile_cap;
  A.cell_start = c->d_cell_start.p; A.posq = c->d_posq.p; A.type_s = c->d_type_sorted.p;
  A.nbr_count = c->d_nbr_count.p; A.nbr_split = c->d_nbr_split.p; A.nbr = c->d_nbr.p;
  A.pairpar = c->d_pair_par.p; A.nt = c->nt;
  A.etab = c->pair_taylor ? c->d_ewald_tay.p : c->d_ewald_tab.p; A.inv_dx = 1.0 / c->erfc_dx; A.dx = c->erfc_dx; A.delta = c->alpha * c->erfc_dx;
  for (int d = 0; d < 3; ++d) { A.box.L[d] = c->box[d]; A.box.invL[d] = 1.0 / c->box[d]; }
  A.rc2 = c->rc * c->rc;
  A.Fs = c->d_fsorted.p; A.part_el = c->d_partials.p + (size_t)slot_elec * MAX_BLOCKS; A.part_vdw = c->d_partials.p + (size_t)slot_vdw * MAX_BLOCKS;
  A.prune_state = c->d_prune_mode.p; A.flat = c->d_flat.p; A.cell_trips = c->d_cell_trips.p; A.stride = c->nbr_cap;
  A.posq_prune = c->d_posq_prune.p;
  A.tt = c->d_tt.p; A.dtt = c->d_dtt.p;
  const double rp = c->rc + c->prune_delta;
  A.rp2 = rp * rp; A.N_all = N;
  A.prd1 = c->d_partials.p + (size_t)S_PRD1 * MAX_BLOCKS; A.prd2 = c->d_partials.p + (size_t)S_PRD2 * MAX_BLOCKS; A.prune_delta = c->prune_delta;
  A.self_nb = 0; A.epoch = 0; A.sb = nullptr;
}

// called by pair_forces (pair.hip) in place of the k_pair_half launch; the caller has sized the pruning state and the sorted-order accumulator
int pair_flat_launch(msevb_ctx* c, int c_lo, int c_hi, int slot_elec, int slot_vdw, int* nblocks_out) {
  const int N = c->N, ncell = c->ncell[0] * c->ncell[1] * c->ncell[2];
  const size_t need = (size_t)N * (c->nbr_cap + 96) + 3584 * (size_t)ncell + 6 * 64;      // (+ the trips the force loop's unconditional entry loads may reach behind the last cell's list)
  if (c->d_flat.n < need || c->d_cell_trips.n < 32 * (size_t)ncell) {      // (only behind a rebuild that enlarged the rows: the pruning state has been reset with it)
    if (c->d_flat.alloc(need) || c->d_cell_trips.alloc(32 * (size_t)ncell)) FAIL(MSEVB_ERR_HIP, "hipMalloc failed (flat pair list)");
  }
  const bool split = pair_flat_split(c);
  { static const bool dbg = getenv("MSEVB_PAIR_DEBUG") != nullptr; if (dbg) fprintf(stderr, "pair_flat: half tile max %d cap %d, sub-tile max %d, largest cell %d, nt %d, split %d\n", c->half_tile_max, c->half_tile_cap, c->sub_tile_max, c->cell_atoms_max, c->nt, (int)split); }
  if (split != c->pair_split_now) { CHK(nb_prune_reset(c)); c->pair_split_now = split; }      // (the two geometries keep different lists)
  const int nblocks = round_up8(std::max((c_hi - c_lo) * (split ? 2 : 1), 1));
  if (nblocks > MAX_BLOCKS) FAIL(MSEVB_ERR_LIMIT, "pair_forces: too many cells");
  *nblocks_out = nblocks;
  PairFlatArgs A;
  memset(&A, 0, sizeof A);      // (padding bytes too: captured / replayed launches are compared by their argument bytes)
  pf_args(c, c_lo, c_hi, slot_elec, slot_vdw, A);
  if (++c->pair_epoch <= 0) c->pair_epoch = 1;
  A.epoch = c->step_sb ? 0 : c->pair_epoch; A.sb = c->step_sb; A.self_nb = c->pair_self_nb; c->pair_self_nb = 0;      // (one-shot: set by total_force_enqueue for the launch it is about to make)
  const bool ty = c->pair_taylor;
  bool* at = c->attr_pair_flat;
  if (split) {
    if (c->pair_split_opt == 2) return ty ? pf_launch<1344, true, true, 512, 4, true>(c, A, nblocks, at[8]) : pf_launch<1344, false, true, 512, 4, true>(c, A, nblocks, at[9]);      // (2 x 8 waves, pipelined loop)
    return ty ? pf_launch<1344, true, true, 768, 6, false>(c, A, nblocks, at[10]) : pf_launch<1344, false, true, 768, 6, false>(c, A, nblocks, at[11]);      // 2 x 12 waves, 80 registers
  }
  const int need_slots = c->half_tile_cap + 1;
  if (c->has_sapt) {      // a live SAPT term: the same kernel with the Buckingham / Tang-Toennies branch (one-row Ewald table, one workgroup per cell, one trip at a time: with two in flight the branch spills 22 registers)
    if (need_slots <= 2048) return pf_launch<2048, true, false, 1024, 4, false, true>(c, A, nblocks, at[16]);
    if (need_slots <= 2304) return pf_launch<2304, true, false, 1024, 4, false, true>(c, A, nblocks, at[17]);
    return pf_launch<2560, true, false, 1024, 4, false, true>(c, A, nblocks, at[18]);
  }
  if (!c->pair_pipe) {      // option pair_pipe = 0: one trip at a time in <= 80 registers - the four waves per SIMD of a tile's workgroup then leave 192 registers per lane for waves of the side chains
    if (need_slots <= 2048) return ty ? pf_launch<2048, true, false, 1024, 6, false>(c, A, nblocks, at[12]) : pf_launch<2048, false, false, 1024, 6, false>(c, A, nblocks, at[13]);
    return ty ? pf_launch<2560, true, false, 1024, 6, false>(c, A, nblocks, at[14]) : pf_launch<2560, false, false, 1024, 6, false>(c, A, nblocks, at[15]);
  }
  if (need_slots <= 1024) return ty ? pf_launch<1024, true, false, 1024, 4, true>(c, A, nblocks, at[0]) : pf_launch<1024, false, false, 1024, 4, true>(c, A, nblocks, at[1]);
  if (need_slots <= 2048) return ty ? pf_launch<2048, true, false, 1024, 4, true>(c, A, nblocks, at[2]) : pf_launch<2048, false, false, 1024, 4, true>(c, A, nblocks, at[3]);
  if (need_slots <= 2304) return ty ? pf_launch<2304, true, false, 1024, 4, true>(c, A, nblocks, at[6]) : pf_launch<2304, false, false, 1024, 4, true>(c, A, nblocks, at[7]);      // (the largest pitch that keeps all four position arrays within the 16-bit offset field)
  return ty ? pf_launch<2560, true, false, 1024, 4, true>(c, A, nblocks, at[4]) : pf_launch<2560, false, false, 1024, 4, true>(c, A, nblocks, at[5]);
}
