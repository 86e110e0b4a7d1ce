// fft_lds.hip - the reciprocal-space solve of one 64^3 grid, theta = IFFT( CB . FFT(Q) ), in THREE kernels
// (reference: src/pme.f90:73-125 - DftiComputeForward, FQ = FQ*CB with E = 1/2 sum CB |FQ|^2, DftiComputeBackward, both unnormalised).
//
// rocFFT runs a 64^3 double-complex transform as three kernels of 7-34 us each: at 4 MB the transform is latency bound, not bandwidth
// bound, and the pair of them + the convolution + the real<->complex passes were ~105 us alone, twice per MS-EVB step (principal charge
// grid; blended charge grid of the Hellmann-Feynman force - the second one sits on the critical path).  Here:
//   k_fft_xy_fwd : one workgroup per z plane.  Reads the REAL charge plane (32 KB), 64 x-lines then 64 y-lines through LDS, writes the
//                  complex plane (x fastest, coalesced).
//   k_fft_z_conv : one workgroup per (16 x, y) bundle of z-lines: forward along z, times CB (+ the energy partial), backward along z,
//                  in place - a z-line never leaves its registers between the two transforms.
//   k_fft_xy_inv : one workgroup per z plane, y-lines then x-lines backward, writes the REAL part (the potential is real: Q is real and
//                  CB is even) - no complex->real pass behind it.
// A 64-point line is 8 x 8 (Cooley-Tukey): a thread holds x[j + 8 m], m = 0..7, does a radix-8 butterfly in registers, twiddles by
// W64^(j k2), the 8 threads of a line exchange through LDS, second radix-8 butterfly: the thread then holds X[j + 8 k1], the layout it
// started with - so the convolution and the backward transform need no further exchange.
// Rounding differs from rocFFT / MKL in the last bits only (tested against the rocFFT path).
#include "ctx.hpp"

namespace {

constexpr int FK = 64;
constexpr int LS = 72;      // LDS row stride in doubles: 64 + 8 (rows 8 banks apart; 8 x 9 exchange layout fits)
constexpr double RH = 0.70710678118654752440;

__device__ double2 g_tw64[64];      // W64^m = (cos, sin)(2 pi m / 64), host-computed

// radix-4 / radix-8 butterflies, natural order in and out; S = -1 forward (e^{-i...}), +1 backward
template <int S> __device__ __forceinline__ void dft4(double& r0, double& i0, double& r1, double& i1, double& r2, double& i2, double& r3, double& i3) {
  const double t0r = r0 + r2, t0i = i0 + i2, t1r = r0 - r2, t1i = i0 - i2;
  const double t2r = r1 + r3, t2i = i1 + i3, t3r = r1 - r3, t3i = i1 - i3;
  r0 = t0r + t2r; i0 = t0i + t2i;
  r2 = t0r - t2r; i2 = t0i - t2i;
  // b1 = t1 + i S t3, b3 = t1 - i S t3
  r1 = t1r - S * t3i; i1 = t1i + S * t3r;
  r3 = t1r + S * t3i; i3 = t1i - S * t3r;
}

template <int S> __device__ __forceinline__ void dft8(double (&re)[8], double (&im)[8]) {
  double er0 = re[0], ei0 = im[0], er1 = re[2], ei1 = im[2], er2 = re[4], ei2 = im[4], er3 = re[6], ei3 = im[6];
  double or0 = re[1], oi0 = im[1], or1 = re[3], oi1 = im[3], or2 = re[5], oi2 = im[5], or3 = re[7], oi3 = im[7];
  dft4<S>(er0, ei0, er1, ei1, er2, ei2, er3, ei3);
  dft4<S>(or0, oi0, or1, oi1, or2, oi2, or3, oi3);
  // O[k] *= W8^k: W8^1 = (c, S c), W8^2 = (0, S), W8^3 = (-c, S c)
  const double a1r = RH * (or1 - S * oi1), a1i = RH * (S * or1 + oi1);
  const double a2r = -S * oi2, a2i = S * or2;
  const double a3r = -RH * (or3 + S * oi3), a3i = RH * (S * or3 - oi3);
  re[0] = er0 + or0; im[0] = ei0 + oi0; re[4] = er0 - or0; im[4] = ei0 - oi0;
  re[1] = er1 + a1r; im[1] = ei1 + a1i; re[5] = er1 - a1r; im[5] = ei1 - a1i;
  re[2] = er2 + a2r; im[2] = ei2 + a2i; re[6] = er2 - a2r; im[6] = ei2 - a2i;
  re[3] = er3 + a3r; im[3] = ei3 + a3i; re[7] = er3 - a3r; im[7] = ei3 - a3i;
}

// A[k2] *= W64^(S j k2)
template <int S> __device__ __forceinline__ void twiddle(double (&re)[8], double (&im)[8], const double2 (&tw)[8]) {
#pragma unroll
  for (int k = 1; k < 8; ++k) {
    const double c = tw[k].x, s = S * tw[k].y;
    const double r = re[k] * c - im[k] * s, i = re[k] * s + im[k] * c;
    re[k] = r; im[k] = i;
  }
}

__device__ __forceinline__ void load_tw(int j, double2 (&tw)[8]) {
#pragma unroll
  for (int k = 0; k < 8; ++k) tw[k] = g_tw64[(j * k) & 63];
}

// ---- x then y, forward, real input ----
// PATCH: the plane is staged in LDS first and the stencils of P that touch it are added there (coefficient x charge x tap weights, one thread per
// stencil, 36 LDS atomics each) - the blended charge grid Q_1 + sum_d c_d^2 DQ_d of the MS-EVB step exists only inside this kernel (round 3:
// a 2 MB copy of Q_1 in front of the energy fetch + k_evb_blend_patch in front of the transform).  Q == nullptr: a zero plane (ranks other than 0).
template <bool PATCH>
__global__ __launch_bounds__(512) void k_fft_xy_fwd(const double* __restrict__ Q, double2* __restrict__ grid, const FftPatch P) {
  extern __shared__ double lds[];
  double* pr = lds;                 // [64][LS] real parts
  double* pi = lds + FK * LS;       // imaginary parts
  const int z = blockIdx.x, t = threadIdx.x;
  double re[8], im[8];
  double2 tw[8];
  if (PATCH) {
    if (P.wait_mark) {      // (as evb_wait_mark, evb.hip: 2 s limit, then the sticky fault flag)
      if (t == 0) {
        const long long t0 = wall_clock64();
        const long long want = P.wait_sb ? P.wait_sb[SB_A] : P.wait_seq;
        while (__hip_atomic_load(P.wait_mark, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != want) {      // (relaxed polls, one acquire behind them)
          if (wall_clock64() - t0 > 200000000LL) { if (P.fault) *P.fault = 2.0; break; }
          __builtin_amdgcn_s_sleep(8);
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
      }
      __syncthreads();
    }
    const double* plane = Q ? Q + (size_t)FK * FK * z : nullptr;
#pragma unroll
    for (int m = 0; m < 8; ++m) { const int k = t + 512 * m, y = k >> 6, x = k & 63; pr[y * LS + x] = plane ? plane[k] : 0.0; }
    __syncthreads();
    for (int s0 = 0; s0 < P.nst; s0 += 512) {
      const int si = s0 + t;
      if (si < P.nst) {
        const Stencil& S = P.st[si];
        int k3 = S.base[2] - z; if (k3 < 0) k3 += FK;
        if (k3 < 6 && S.q != 0.0) {
          const double q = S.q * (P.dc ? P.dc[1 + S.grid] : P.c[S.grid]) * S.w[12 + k3];
          if (q != 0.0) {
            for (int k2 = 0; k2 < 6; ++k2) {
              int n2 = S.base[1] - k2; if (n2 < 0) n2 += FK;
              const double q2 = q * S.w[6 + k2];
              for (int k1 = 0; k1 < 6; ++k1) { int n1 = S.base[0] - k1; if (n1 < 0) n1 += FK; atomicAdd(&pr[n2 * LS + n1], q2 * S.w[k1]); }
            }
          }
        }
      }
    }
    __syncthreads();
  }
  {      // x lines: 8 threads per line y
    const int y = t >> 3, j = t & 7;
    load_tw(j, tw);
    if (PATCH) {
#pragma unroll
      for (int m = 0; m < 8; ++m) { re[m] = pr[y * LS + j + 8 * m]; im[m] = 0.0; }
      __syncthreads();      // (the exchange below overwrites the staged plane)
    } else {
      const double* row = Q + (size_t)FK * (y + (size_t)FK * z);
#pragma unroll
      for (int m = 0; m < 8; ++m) { re[m] = row[j + 8 * m]; im[m] = 0.0; }
    }
    dft8<-1>(re, im);
    twiddle<-1>(re, im, tw);
    double* rr = pr + y * LS + j * 9; double* ri = pi + y * LS + j * 9;
#pragma unroll
    for (int k = 0; k < 8; ++k) { rr[k] = re[k]; ri[k] = im[k]; }
    __syncthreads();
#pragma unroll
    for (int n = 0; n < 8; ++n) { re[n] = pr[y * LS + n * 9 + j]; im[n] = pi[y * LS + n * 9 + j]; }
    dft8<-1>(re, im);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k) { pr[y * LS + j + 8 * k] = re[k]; pi[y * LS + j + 8 * k] = im[k]; }      // X[kx = j + 8 k] of line y
    __syncthreads();
  }
  {      // y lines: thread (kx, j), kx fastest
    const int kx = t & 63, j = t >> 6;
    load_tw(j, tw);
#pragma unroll
    for (int m = 0; m < 8; ++m) { re[m] = pr[(j + 8 * m) * LS + kx]; im[m] = pi[(j + 8 * m) * LS + kx]; }
    dft8<-1>(re, im);
    twiddle<-1>(re, im, tw);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k) { pr[(j * 8 + k) * LS + kx] = re[k]; pi[(j * 8 + k) * LS + kx] = im[k]; }
    __syncthreads();
#pragma unroll
    for (int n = 0; n < 8; ++n) { re[n] = pr[(n * 8 + j) * LS + kx]; im[n] = pi[(n * 8 + j) * LS + kx]; }
    dft8<-1>(re, im);
    double2* plane = grid + (size_t)FK * FK * z;
#pragma unroll
    for (int k = 0; k < 8; ++k) plane[(size_t)FK * (j + 8 * k) + kx] = make_double2(re[k], im[k]);
  }
}

// ---- z forward, x CB (+ energy), z backward: 16 x-neighbours of one y per workgroup, 8 threads per z-line ----
__global__ __launch_bounds__(128) void k_fft_z_conv(double2* __restrict__ grid, const double* __restrict__ CB, double* __restrict__ part, double half_conv) {
  __shared__ double pr[16 * 73], pi[16 * 73];
  __shared__ double red[2];
  const int t = threadIdx.x, lx = t & 15, j = t >> 4;
  const int x = (blockIdx.x & 3) * 16 + lx, y = blockIdx.x >> 2;
  const size_t col = (size_t)x + (size_t)FK * y, zs = (size_t)FK * FK;
  double re[8], im[8];
  double2 tw[8];
  load_tw(j, tw);
  double cb[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) { const double2 v = grid[col + zs * (j + 8 * m)]; re[m] = v.x; im[m] = v.y; }
#pragma unroll
  for (int m = 0; m < 8; ++m) cb[m] = CB[col + zs * (j + 8 * m)];
  double* er = pr + lx * 73; double* ei = pi + lx * 73;
  dft8<-1>(re, im);
  twiddle<-1>(re, im, tw);
#pragma unroll
  for (int k = 0; k < 8; ++k) { er[j * 9 + k] = re[k]; ei[j * 9 + k] = im[k]; }
  __syncthreads();
#pragma unroll
  for (int n = 0; n < 8; ++n) { re[n] = er[n * 9 + j]; im[n] = ei[n * 9 + j]; }
  dft8<-1>(re, im);      // F[kz = j + 8 k]
  double e = 0.0;
#pragma unroll
  for (int k = 0; k < 8; ++k) { e += cb[k] * (re[k] * re[k] + im[k] * im[k]); re[k] *= cb[k]; im[k] *= cb[k]; }
  dft8<1>(re, im);
  twiddle<1>(re, im, tw);
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 8; ++k) { er[j * 9 + k] = re[k]; ei[j * 9 + k] = im[k]; }
  __syncthreads();
#pragma unroll
  for (int n = 0; n < 8; ++n) { re[n] = er[n * 9 + j]; im[n] = ei[n * 9 + j]; }
  dft8<1>(re, im);
#pragma unroll
  for (int m = 0; m < 8; ++m) grid[col + zs * (j + 8 * m)] = make_double2(re[m], im[m]);
  if (part) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) e += __shfl_xor(e, o, 64);
    if ((t & 63) == 0) red[t >> 6] = e;
    __syncthreads();
    if (t == 0) part[blockIdx.x] = (red[0] + red[1]) * half_conv;
  }
}

// ---- y then x, backward, real output ----
__global__ __launch_bounds__(512) void k_fft_xy_inv(const double2* __restrict__ grid, double* __restrict__ theta) {
  extern __shared__ double lds[];
  double* pr = lds;
  double* pi = lds + FK * LS;
  const int z = blockIdx.x, t = threadIdx.x;
  double re[8], im[8];
  double2 tw[8];
  {
    const int kx = t & 63, j = t >> 6;
    load_tw(j, tw);
    const double2* plane = grid + (size_t)FK * FK * z;
#pragma unroll
    for (int m = 0; m < 8; ++m) { const double2 v = plane[(size_t)FK * (j + 8 * m) + kx]; re[m] = v.x; im[m] = v.y; }
    dft8<1>(re, im);
    twiddle<1>(re, im, tw);
#pragma unroll
    for (int k = 0; k < 8; ++k) { pr[(j * 8 + k) * LS + kx] = re[k]; pi[(j * 8 + k) * LS + kx] = im[k]; }
    __syncthreads();
#pragma unroll
    for (int n = 0; n < 8; ++n) { re[n] = pr[(n * 8 + j) * LS + kx]; im[n] = pi[(n * 8 + j) * LS + kx]; }
    dft8<1>(re, im);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k) { pr[(j + 8 * k) * LS + kx] = re[k]; pi[(j + 8 * k) * LS + kx] = im[k]; }      // row y = j + 8 k, column kx
    __syncthreads();
  }
  {
    const int y = t >> 3, j = t & 7;
    load_tw(j, tw);
#pragma unroll
    for (int m = 0; m < 8; ++m) { re[m] = pr[y * LS + j + 8 * m]; im[m] = pi[y * LS + j + 8 * m]; }
    dft8<1>(re, im);
    twiddle<1>(re, im, tw);
    __syncthreads();
    double* rr = pr + y * LS + j * 9; double* ri = pi + y * LS + j * 9;
#pragma unroll
    for (int k = 0; k < 8; ++k) { rr[k] = re[k]; ri[k] = im[k]; }
    __syncthreads();
#pragma unroll
    for (int n = 0; n < 8; ++n) { re[n] = pr[y * LS + n * 9 + j]; im[n] = pi[y * LS + n * 9 + j]; }
    dft8<1>(re, im);
    double* row = theta + (size_t)FK * (y + (size_t)FK * z);
#pragma unroll
    for (int k = 0; k < 8; ++k) row[j + 8 * k] = re[k];
  }
}


// =====================================================================================================================================
// 128^3 (BASELINE.json configs[4]): a z plane is 256 KB, so the three axes go one at a time - five kernels, each streaming the 33.5 MB
// complex grid once in and once out (x forward reads the 16.8 MB REAL charge grid, x backward writes the REAL potential):
//   k_fft128_x<-1>  : 16 x-lines per workgroup, 8 threads per line
//   k_fft128_yz<-1> : y lines, 16 x-neighbours per workgroup (256 B contiguous per row), in place
//   k_fft128_z_conv : z lines forward, x CB (+ energy partial), backward - the line stays in its registers
//   k_fft128_yz<+1>, k_fft128_x<+1>
// A 128-point line is 8 x 16: the thread j of a line holds x[j + 8 m], m = 0..15, does a 16-point DFT over m in registers (two radix-8
// butterflies + W16), twiddles by W128^(j k2), the 8 threads exchange through LDS, thread j takes k2 = j and j + 8 and does two radix-8
// butterflies over the threads' index: X[k2 + 16 k1] - which is X[j + 8 m'] with m' = 2 k1 + (k2 >= 8), the layout it started with.
constexpr int FN = 128;
constexpr double C16 = 0.92387953251128675613, S16 = 0.38268343236508977173;      // cos, sin of pi / 8
__device__ double2 g_tw128[128];      // W128^m = (cos, sin)(2 pi m / 128)

template <int S> __device__ __forceinline__ void dft16(double (&re)[16], double (&im)[16]) {
  double er[8], ei[8], orr[8], oi[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) { er[m] = re[2 * m]; ei[m] = im[2 * m]; orr[m] = re[2 * m + 1]; oi[m] = im[2 * m + 1]; }
  dft8<S>(er, ei);
  dft8<S>(orr, oi);
  // O[k] *= W16^k = (cos(k pi/8), S sin(k pi/8))
  const double wc[8] = {1.0, C16, RH, S16, 0.0, -S16, -RH, -C16}, ws[8] = {0.0, S16, RH, C16, 1.0, C16, RH, S16};
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const double c = wc[k], sn = S * ws[k];
    const double tr = orr[k] * c - oi[k] * sn, ti = orr[k] * sn + oi[k] * c;
    re[k] = er[k] + tr; im[k] = ei[k] + ti; re[k + 8] = er[k] - tr; im[k + 8] = ei[k] - ti;
  }
}

// one 128-point line held by 8 threads (j = 0..7; in and out: element j + 8 m in slot m); xr / xi: the line's LDS scratch, 8 x 17 doubles each.
// The caller's workgroup must reach this call with all its threads (two barriers inside).
template <int S> __device__ __forceinline__ void line128(double (&re)[16], double (&im)[16], int j, double* xr, double* xi) {
  dft16<S>(re, im);
#pragma unroll
  for (int k = 1; k < 16; ++k) {
    const double2 w = g_tw128[(j * k) & 127];
    const double c = w.x, sn = S * w.y;
    const double r = re[k] * c - im[k] * sn, i = re[k] * sn + im[k] * c;
    re[k] = r; im[k] = i;
  }
  __syncthreads();      // (scratch of the transform before this one has been read)
#pragma unroll
  for (int k = 0; k < 16; ++k) { xr[j * 17 + k] = re[k]; xi[j * 17 + k] = im[k]; }
  __syncthreads();
  double ar[8], ai[8], br[8], bi[8];
#pragma unroll
  for (int n = 0; n < 8; ++n) { ar[n] = xr[n * 17 + j]; ai[n] = xi[n * 17 + j]; br[n] = xr[n * 17 + j + 8]; bi[n] = xi[n * 17 + j + 8]; }
  dft8<S>(ar, ai);
  dft8<S>(br, bi);
#pragma unroll
  for (int k = 0; k < 8; ++k) { re[2 * k] = ar[k]; im[2 * k] = ai[k]; re[2 * k + 1] = br[k]; im[2 * k + 1] = bi[k]; }
}

constexpr int L128 = 8 * 17 + 1;      // scratch doubles per line and part (odd: neighbouring lines start in different banks)

// x lines: workgroup = 16 consecutive y of one z, 8 threads per line.  S = -1: real charge grid in, complex out; S = +1: complex in, real part out.
template <int S> __global__ __launch_bounds__(128) void k_fft128_x(const double* __restrict__ Q, double2* __restrict__ grid, double* __restrict__ theta) {
  __shared__ double pr[16 * L128], pi[16 * L128];
  const int t = threadIdx.x, j = t & 7, l = t >> 3;
  const size_t row = (size_t)FN * ((size_t)(blockIdx.x & 7) * 16 + l + (size_t)FN * (blockIdx.x >> 3));
  double re[16], im[16];
  if (S < 0) {
#pragma unroll
    for (int m = 0; m < 16; ++m) { re[m] = Q[row + j + 8 * m]; im[m] = 0.0; }
  } else {
#pragma unroll
    for (int m = 0; m < 16; ++m) { const double2 v = grid[row + j + 8 * m]; re[m] = v.x; im[m] = v.y; }
  }
  line128<S>(re, im, j, pr + l * L128, pi + l * L128);
  if (S < 0) {
#pragma unroll
    for (int m = 0; m < 16; ++m) grid[row + j + 8 * m] = make_double2(re[m], im[m]);
  } else {
#pragma unroll
    for (int m = 0; m < 16; ++m) theta[row + j + 8 * m] = re[m];
  }
}

// y lines in place: workgroup = 16 neighbouring x of one z
template <int S> __global__ __launch_bounds__(128) void k_fft128_y(double2* __restrict__ grid) {
  __shared__ double pr[16 * L128], pi[16 * L128];
  const int t = threadIdx.x, lx = t & 15, j = t >> 4;
  const size_t col = (size_t)(blockIdx.x & 7) * 16 + lx + (size_t)FN * FN * (blockIdx.x >> 3);
  double re[16], im[16];
#pragma unroll
  for (int m = 0; m < 16; ++m) { const double2 v = grid[col + (size_t)FN * (j + 8 * m)]; re[m] = v.x; im[m] = v.y; }
  line128<S>(re, im, j, pr + lx * L128, pi + lx * L128);
#pragma unroll
  for (int m = 0; m < 16; ++m) grid[col + (size_t)FN * (j + 8 * m)] = make_double2(re[m], im[m]);
}

// z lines forward, x CB (+ energy partial: 1/2 conv sum CB |F|^2, src/pme.f90:100-125), z lines backward; workgroup = 16 neighbouring x of one y
__global__ __launch_bounds__(128) void k_fft128_z_conv(double2* __restrict__ grid, const double* __restrict__ CB, double* __restrict__ part, double half_conv) {
  __shared__ double pr[16 * L128], pi[16 * L128];
  __shared__ double red[2];
  const int t = threadIdx.x, lx = t & 15, j = t >> 4;
  const size_t col = (size_t)(blockIdx.x & 7) * 16 + lx + (size_t)FN * (blockIdx.x >> 3), zs = (size_t)FN * FN;
  double re[16], im[16];
#pragma unroll
  for (int m = 0; m < 16; ++m) { const double2 v = grid[col + zs * (j + 8 * m)]; re[m] = v.x; im[m] = v.y; }
  line128<-1>(re, im, j, pr + lx * L128, pi + lx * L128);
  double e = 0.0;
#pragma unroll
  for (int m = 0; m < 16; ++m) { const double cb = CB[col + zs * (j + 8 * m)]; e += cb * (re[m] * re[m] + im[m] * im[m]); re[m] *= cb; im[m] *= cb; }
  line128<1>(re, im, j, pr + lx * L128, pi + lx * L128);
#pragma unroll
  for (int m = 0; m < 16; ++m) grid[col + zs * (j + 8 * m)] = make_double2(re[m], im[m]);
  if (part) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) e += __shfl_xor(e, o, 64);
    if ((t & 63) == 0) red[t >> 6] = e;
    __syncthreads();
    if (t == 0) part[blockIdx.x] = (red[0] + red[1]) * half_conv;
  }
}

}  // namespace

bool fft_lds_usable(const msevb_ctx* c) { return c->fft_own && (c->K == FK || c->K == FN); }

// once per context (the twiddle table is a per-device symbol; LDS beyond 64 KB needs the attribute)
int fft_lds_setup(msevb_ctx* c) {
  if (c->K == FN) {
    double2 tw[128];
    for (int m = 0; m < 128; ++m) { const double a = 2.0 * M_PI * m / 128.0; tw[m] = make_double2(cos(a), sin(a)); }
    tw[0] = make_double2(1, 0); tw[32] = make_double2(0, 1); tw[64] = make_double2(-1, 0); tw[96] = make_double2(0, -1);
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_tw128), tw, sizeof tw));
    return MSEVB_OK;
  }
  if (c->K != FK) return MSEVB_OK;
  double2 tw[64];
  for (int m = 0; m < 64; ++m) { const double a = 2.0 * M_PI * m / 64.0; tw[m] = make_double2(cos(a), sin(a)); }
  // exact values where they are exact
  tw[0] = make_double2(1, 0); tw[16] = make_double2(0, 1); tw[32] = make_double2(-1, 0); tw[48] = make_double2(0, -1);
  HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_tw64), tw, sizeof tw));
  const int lds = 2 * FK * LS * (int)sizeof(double);
  HIPCHK(hipFuncSetAttribute((const void*)k_fft_xy_fwd<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  HIPCHK(hipFuncSetAttribute((const void*)k_fft_xy_fwd<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  HIPCHK(hipFuncSetAttribute((const void*)k_fft_xy_inv, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  return MSEVB_OK;
}

// theta = Re IFFT(CB FFT(Q)); Q and theta real [64^3] (may be the same buffer), work complex [64^3]; part (optional): 256 energy partials
bool fft_lds_can_patch(const msevb_ctx* c) { return fft_lds_usable(c) && c->K == FK; }
int fft_lds_solve(msevb_ctx* c, const double* Q, double2* work, double* theta, double* part, int* nblocks, hipStream_t st, const FftPatch* patch) {
  if (patch && c->K != FK) FAIL(MSEVB_ERR_STATE, "fft_lds_solve: a charge-grid patch needs the 64^3 transform");
  if (c->K == FN) {      // five kernels, 1024 workgroups of 128 threads each
    hipLaunchKernelGGL(k_fft128_x<-1>, dim3(FN * 8), dim3(128), 0, st, Q, work, (double*)nullptr);
    hipLaunchKernelGGL(k_fft128_y<-1>, dim3(FN * 8), dim3(128), 0, st, work);
    hipLaunchKernelGGL(k_fft128_z_conv, dim3(FN * 8), dim3(128), 0, st, work, c->d_CB.p, part, 0.5 * c->conv);
    hipLaunchKernelGGL(k_fft128_y<1>, dim3(FN * 8), dim3(128), 0, st, work);
    hipLaunchKernelGGL(k_fft128_x<1>, dim3(FN * 8), dim3(128), 0, st, (const double*)nullptr, work, theta);
    HIPCHK(hipGetLastError());
    if (nblocks) *nblocks = FN * 8;
    return MSEVB_OK;
  }
  const int lds = 2 * FK * LS * (int)sizeof(double);
  static const FftPatch no_patch = {};
  if (patch) hipLaunchKernelGGL(k_fft_xy_fwd<true>, dim3(FK), dim3(512), lds, st, Q, work, *patch);
  else hipLaunchKernelGGL(k_fft_xy_fwd<false>, dim3(FK), dim3(512), lds, st, Q, work, no_patch);
  hipLaunchKernelGGL(k_fft_z_conv, dim3(FK * 4), dim3(128), 0, st, work, c->d_CB.p, part, 0.5 * c->conv);
  hipLaunchKernelGGL(k_fft_xy_inv, dim3(FK), dim3(512), lds, st, work, theta);
  HIPCHK(hipGetLastError());
  if (nblocks) *nblocks = FK * 4;
  return MSEVB_OK;
}
