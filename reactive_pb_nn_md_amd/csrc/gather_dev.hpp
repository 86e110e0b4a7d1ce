// Stencil gather kernel of the per-diabat reciprocal force (evb.hip).
#pragma once
#include "ctx.hpp"

namespace {

struct GatherSlots { int s[8]; };     // force-store slot of each grid of a pass (kernel argument)

// All-atom reciprocal force of up to NGP owned diabats in one pass (update_reciprocal_space_force_dQ_dr for the atoms whose
// stencil did not change, src/ms_evb.f90:2055-2084).  The store holds DIFFERENCES to the principal force:
// Fstore[slot_g] = (f_g - Frecip1) + Fdelta[slot_g]  (F_gg = F11 + Fstore[slot_g]; the blend needs F11 only once because sum c_d^2 = 1).
// ONE WAVEFRONT PER ATOM; lane = (row r, grid g, tap k1): a stencil row of
// the packed potential [point][grid] is 6 x ngtot consecutive doubles, so a wave-wide load touches 3-6 cache lines instead of
// one line per lane (the per-lane-plane layout this replaces was bound by the L1 tag rate: 48 lines per load).  With NGP = 8
// the row index is wave-uniform and the k2/k3 weights come through scalar loads.
template <int NGP>
__global__ __launch_bounds__(256) void k_stencil_gather(int N, int K, int ng, GatherSlots slots, const int* __restrict__ perm,
                                                    const double* __restrict__ chg, const int* __restrict__ base, const double* __restrict__ w,
                                                    const double* __restrict__ dw, const double* __restrict__ pack, int ngtot, int g0, double conv, double k00,
                                                    double k01, double k02, double k10, double k11, double k12, double k20, double k21, double k22,
                                                    const double* Frecip1, double* __restrict__ Fstore, double* __restrict__ Fdelta, size_t fstride) {
  constexpr int RPW = 64 / (8 * NGP);    // stencil rows per wave-wide load
  const int lane = threadIdx.x & 63;
  const int s = xcd_block(blockIdx.x, gridDim.x) * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (s >= N) return;
  const int i = perm ? perm[s] : s;
  const int k1 = lane & 7, g = (lane >> 3) & (NGP - 1), rsub = lane / (8 * NGP);
  const bool live = k1 < 6 && g < ng;
  const int b1 = base[3 * i], b2 = base[3 * i + 1], b3 = base[3 * i + 2];
  const double* wi = w + (size_t)i * 18;
  const double* dwi = dw + (size_t)i * 18;
  int n1 = b1 - (live ? k1 : 0); if (n1 < 0) n1 += K;
  const double* col = pack + (size_t)n1 * ngtot + g0 + (live ? g : 0);
  const unsigned Kng = (unsigned)K * (unsigned)ngtot;      // 32-bit offsets: the host checks K^3 * ngtot < 2^31
  double aa = 0.0, ab = 0.0, ac = 0.0;
  {
#pragma clang fp contract(fast)
    if constexpr (RPW == 1) {       // wave-uniform rows: plane sums first (2 FMAs per row + 3 per plane, no weight products)
#pragma unroll
      for (int k3 = 0; k3 < 6; ++k3) {
        int n3 = b3 - k3; if (n3 < 0) n3 += K;
        double sa = 0.0, sb = 0.0;
#pragma unroll
        for (int k2 = 0; k2 < 6; ++k2) {
          int n2 = b2 - k2; if (n2 < 0) n2 += K;
          const double th = col[((unsigned)n3 * (unsigned)K + (unsigned)n2) * Kng];
          sa += wi[6 + k2] * th; sb += dwi[6 + k2] * th;
        }
        aa += wi[12 + k3] * sa; ab += wi[12 + k3] * sb; ac += dwi[12 + k3] * sa;
      }
    } else {
#pragma unroll
      for (int it = 0; it < (36 + RPW - 1) / RPW; ++it) {
        const int rr = it * RPW + rsub;
        if (RPW < 8 || rr < 36) {
          const int k3 = rr / 6, k2 = rr - 6 * k3;
          int n2 = b2 - k2; if (n2 < 0) n2 += K;
          int n3 = b3 - k3; if (n3 < 0) n3 += K;
          const double th = col[((unsigned)n3 * (unsigned)K + (unsigned)n2) * Kng];
          const double w2 = wi[6 + k2], d2 = dwi[6 + k2], w3 = wi[12 + k3], d3 = dwi[12 + k3];
          aa += (w2 * w3) * th; ab += (d2 * w3) * th; ac += (w2 * d3) * th;
        }
      }
    }
  }
  double f1 = live ? dwi[k1] * aa : 0.0, f2 = live ? wi[k1] * ab : 0.0, f3 = live ? wi[k1] * ac : 0.0;
#pragma unroll
  for (int o = 1; o < 8; o <<= 1) { f1 += __shfl_xor(f1, o, 64); f2 += __shfl_xor(f2, o, 64); f3 += __shfl_xor(f3, o, 64); }
#pragma unroll
  for (int o = 8 * NGP; o < 64; o <<= 1) { f1 += __shfl_xor(f1, o, 64); f2 += __shfl_xor(f2, o, 64); f3 += __shfl_xor(f3, o, 64); }
  if (k1 == 0 && rsub == 0 && g < ng) {
    const double sc = chg[i] * conv, Kd = (double)K;
    f1 *= sc; f2 *= sc; f3 *= sc;
    const double fx = -Kd * (k00 * f1 + k10 * f2 + k20 * f3), fy = -Kd * (k01 * f1 + k11 * f2 + k21 * f3), fz = -Kd * (k02 * f1 + k12 * f2 + k22 * f3);
    {
      // Fdelta: real-space / bonded / repulsion delta of this diabat, accumulated concurrently on another stream; consumed and
      // re-zeroed here so that it is clean for the next step
      double* F = Fstore + (size_t)slots.s[g] * fstride;
      double* D = Fdelta + (size_t)slots.s[g] * fstride;
      F[3 * i] = (fx - Frecip1[3 * i]) + D[3 * i];
      F[3 * i + 1] = (fy - Frecip1[3 * i + 1]) + D[3 * i + 1];
      F[3 * i + 2] = (fz - Frecip1[3 * i + 2]) + D[3 * i + 2];
      D[3 * i] = 0.0; D[3 * i + 1] = 0.0; D[3 * i + 2] = 0.0;
    }
  }
}

}  // namespace
