// Stage-level PME entry points with the reference's own granularity (SURVEY.md 8b "hot-path signatures to preserve"):
//   msevb_grid_Q            grid_Q            src/pme.f90:184
//   msevb_modify_Q_grid     modify_Q_grid     src/pme.f90:275
//   msevb_derivative_grid_Q derivative_grid_Q src/pme.f90:346 (for a range of atoms instead of one per call)
// They take what the reference procedures take - SCALED coordinates (create_scaled_direct_coordinates), charges, a host grid - so a
// Fortran caller of those procedures can switch to the GPU one at a time.  The force path of the library does not use them (it keeps
// everything on the device and fuses these steps, pme.hip / evb.hip); each call here moves its grid over PCIe.
#include "ctx.hpp"

namespace {

// stencil of an atom from its scaled coordinate u (the index arithmetic of src/pme.f90:220-247, :412-414; no FMA contraction)
__global__ __launch_bounds__(256) void k_stencil_from_u(int n, int K, int G, const double* __restrict__ u3, const double* __restrict__ B6,
                                                        const double* __restrict__ B5, int* __restrict__ base, double* __restrict__ w,
                                                        double* __restrict__ dw, int* __restrict__ bad) {
#pragma clang fp contract(off)
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 18 * n) return;
  const int i = t / 18, r = t - 18 * i, l = r / 6, k = r - 6 * l;
  double u = u3[3 * i + l];
  if (!(u >= 0.0 && u < (double)K)) { *bad = 1; u = 0.5; }      // the reference would index outside its grid
  const int np = (int)floor(u);
  if (k == 0) base[3 * i + l] = np;
  const double a1 = u - (double)(np - k), a2 = a1 - 1.0;
  // a scaled coordinate that IS an integer gives a1 = 0 at k = 0 and row 0 (the reference reads B6_spline(0), outside its table): the callers of
  // these entry points hand in arbitrary coordinates, so the rows are held inside the tables (row 1 = the table's value next to zero)
  w[t] = B6[min(max((int)ceil(a1 / 6.0 * (double)G), 1), G) - 1];
  double d = 0.0;
  if (a1 < 5.0) d = B5[min(max((int)ceil(a1 / 5.0 * (double)G), 1), G) - 1];
  if (0.0 < a2) d = d - B5[min(max((int)ceil(a2 / 5.0 * (double)G), 1), G) - 1];
  dw[t] = d;
}

// one thread per (atom, k3, k2): the 6 k1 taps into a REAL grid; sign = +1 / -1 (modify_Q_grid's operation); charges below `small` skipped
__global__ __launch_bounds__(256) void k_spread_real(int n, int K, const double* __restrict__ chg, const int* __restrict__ base, const double* __restrict__ w,
                                                     double sign, double small, double* __restrict__ Q) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)n * 36) return;
  const int i = (int)(t / 36), r = (int)(t % 36), k3 = r / 6, k2 = r % 6;
  const double q = chg[i];
  if (!(fabs(q) > small)) return;
  int n3 = base[3 * i + 2] - k3; if (n3 < 0) n3 += K;
  int n2 = base[3 * i + 1] - k2; if (n2 < 0) n2 += K;
  const double w23 = w[(3 * i + 1) * 6 + k2] * w[(3 * i + 2) * 6 + k3];
  double* row = Q + (size_t)K * (n2 + (size_t)K * n3);
  for (int k1 = 0; k1 < 6; ++k1) {
    int n1 = base[3 * i] - k1; if (n1 < 0) n1 += K;
    atomicAdd(&row[n1], sign * (q * w[(3 * i) * 6 + k1] * w23));
  }
}

// derivative_grid_Q for atoms [0, n): 8 lanes per atom as k_pme_gather; kk and conv are the caller's
__global__ __launch_bounds__(256) void k_gather_real(int n, int K, const double* __restrict__ chg, const int* __restrict__ base, const double* __restrict__ w,
                                                     const double* __restrict__ dw, const double* __restrict__ theta, double conv, double k00, double k01,
                                                     double k02, double k10, double k11, double k12, double k20, double k21, double k22, double* __restrict__ F) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = t >> 3, sub = t & 7;
  double f1 = 0.0, f2 = 0.0, f3 = 0.0;
  if (i < n && sub < 6) {
    int n3 = base[3 * i + 2] - sub; if (n3 < 0) n3 += K;
    const double* wi = w + (size_t)i * 18;
    const double* dwi = dw + (size_t)i * 18;
    for (int k2 = 0; k2 < 6; ++k2) {
      int n2 = base[3 * i + 1] - k2; if (n2 < 0) n2 += K;
      const double* row = theta + (size_t)K * (n2 + (size_t)K * n3);
      double s0 = 0.0, s1 = 0.0;
      for (int k = 0; k < 6; ++k) { int n1 = base[3 * i] - k; if (n1 < 0) n1 += K; const double th = row[n1]; s0 += wi[k] * th; s1 += dwi[k] * th; }
      f1 += s1 * wi[6 + k2] * wi[12 + sub]; f2 += s0 * dwi[6 + k2] * wi[12 + sub]; f3 += s0 * wi[6 + k2] * dwi[12 + sub];
    }
  }
  for (int o = 4; o > 0; o >>= 1) { f1 += __shfl_xor(f1, o, 64); f2 += __shfl_xor(f2, o, 64); f3 += __shfl_xor(f3, o, 64); }
  if (i < n && sub == 0) {
    const double s = chg[i] * conv;
    f1 *= s; f2 *= s; f3 *= s;
    const double Kd = (double)K;
    F[3 * i] = -Kd * (k00 * f1 + k10 * f2 + k20 * f3);
    F[3 * i + 1] = -Kd * (k01 * f1 + k11 * f2 + k21 * f3);
    F[3 * i + 2] = -Kd * (k02 * f1 + k12 * f2 + k22 * f3);
  }
}

struct StageBufs { DBuf<double> u, q, w, dw, grid, F; DBuf<int> base, bad; };

int stage_stencils(msevb_ctx* c, StageBufs& b, const double* chg, const double* xyz_scale, int n) {
  if (b.u.alloc(3 * (size_t)n) || b.q.alloc(n) || b.w.alloc(18 * (size_t)n) || b.dw.alloc(18 * (size_t)n) || b.base.alloc(3 * (size_t)n) || b.bad.alloc(1))
    FAIL(MSEVB_ERR_HIP, "hipMalloc failed (PME stage call)");
  HIPCHK(hipMemcpyAsync(b.u.p, xyz_scale, 3 * (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(b.q.p, chg, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemsetAsync(b.bad.p, 0, sizeof(int), c->stream));
  hipLaunchKernelGGL(k_stencil_from_u, dim3((18 * n + 255) / 256), dim3(256), 0, c->stream, n, c->K, c->spline_grid, b.u.p, c->d_B6.p, c->d_B5.p, b.base.p, b.w.p, b.dw.p, b.bad.p);
  HIPCHK(hipGetLastError());
  return MSEVB_OK;
}
int stage_check(msevb_ctx* c, StageBufs& b) {
  int bad = 0;
  HIPCHK(hipMemcpyAsync(&bad, b.bad.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  if (bad) FAIL(MSEVB_ERR_NUMERIC, "scaled coordinate outside [0, pme_grid) (create_scaled_direct_coordinates wraps once)");
  return MSEVB_OK;
}
void stage_free(StageBufs& b) { b.u.release(); b.q.release(); b.w.release(); b.dw.release(); b.grid.release(); b.F.release(); b.base.release(); b.bad.release(); }

}  // namespace

extern "C" {

int msevb_modify_Q_grid(msevb_ctx* c, double* Q, const double* chg, const double* xyz_scale, int n_atom, int operation) {
  if (!c) return MSEVB_ERR_ARG;
  if (!c->finalized) FAIL(MSEVB_ERR_STATE, "call msevb_finalize first");
  if (!Q || !chg || !xyz_scale || n_atom < 0) FAIL(MSEVB_ERR_ARG, "msevb_modify_Q_grid: null argument");
  HIPCHK(hipSetDevice(c->device));
  const size_t K3 = (size_t)c->K * c->K * c->K;
  StageBufs b;
  int rc = MSEVB_OK;
  auto body = [&]() -> int {
    if (b.grid.alloc(K3)) FAIL(MSEVB_ERR_HIP, "hipMalloc failed (PME stage call)");
    if (operation == 0) HIPCHK(hipMemsetAsync(b.grid.p, 0, K3 * 8, c->stream));      // grid_Q: Q = 0 first (src/pme.f90:200)
    else HIPCHK(hipMemcpyAsync(b.grid.p, Q, K3 * 8, hipMemcpyHostToDevice, c->stream));
    if (n_atom > 0) {
      CHK(stage_stencils(c, b, chg, xyz_scale, n_atom));
      // grid_Q spreads every atom (:212-262); modify_Q_grid skips |q| <= 1e-6 (:296)
      hipLaunchKernelGGL(k_spread_real, dim3((unsigned)(((long)n_atom * 36 + 255) / 256)), dim3(256), 0, c->stream, n_atom, c->K, b.q.p, b.base.p, b.w.p,
                         operation < 0 ? -1.0 : 1.0, operation == 0 ? -1.0 : 1e-6, b.grid.p);
      HIPCHK(hipGetLastError());
    }
    HIPCHK(hipMemcpyAsync(Q, b.grid.p, K3 * 8, hipMemcpyDeviceToHost, c->stream));
    return n_atom > 0 ? stage_check(c, b) : (hipStreamSynchronize(c->stream) == hipSuccess ? MSEVB_OK : MSEVB_ERR_HIP);
  };
  rc = body();
  stage_free(b);
  return rc;
}

int msevb_grid_Q(msevb_ctx* c, double* Q, const double* chg, const double* xyz_scale, int n_atom) { return msevb_modify_Q_grid(c, Q, chg, xyz_scale, n_atom, 0); }

int msevb_derivative_grid_Q(msevb_ctx* c, double* force, const double* FQ, const double* chg, const double* xyz_scale, int n_atom, const double* kk, double conv) {
  if (!c) return MSEVB_ERR_ARG;
  if (!c->finalized) FAIL(MSEVB_ERR_STATE, "call msevb_finalize first");
  if (!force || !FQ || !chg || !xyz_scale || !kk || n_atom < 1) FAIL(MSEVB_ERR_ARG, "msevb_derivative_grid_Q: null argument");
  HIPCHK(hipSetDevice(c->device));
  const size_t K3 = (size_t)c->K * c->K * c->K;
  StageBufs b;
  auto body = [&]() -> int {
    if (b.grid.alloc(K3) || b.F.alloc(3 * (size_t)n_atom)) FAIL(MSEVB_ERR_HIP, "hipMalloc failed (PME stage call)");
    HIPCHK(hipMemcpyAsync(b.grid.p, FQ, K3 * 8, hipMemcpyHostToDevice, c->stream));
    CHK(stage_stencils(c, b, chg, xyz_scale, n_atom));
    // kk is the Fortran array kk(3,3), column-major: kk(i,j) = kk[(i-1) + 3 (j-1)]; force(l) = -K sum_i kk(i,l) Ftilde(i) (src/pme.f90:489-493)
    hipLaunchKernelGGL(k_gather_real, dim3((n_atom * 8 + 255) / 256), dim3(256), 0, c->stream, n_atom, c->K, b.q.p, b.base.p, b.w.p, b.dw.p, b.grid.p, conv,
                       kk[0], kk[3], kk[6], kk[1], kk[4], kk[7], kk[2], kk[5], kk[8], b.F.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(force, b.F.p, 3 * (size_t)n_atom * 8, hipMemcpyDeviceToHost, c->stream));
    return stage_check(c, b);
  };
  const int rc = body();
  stage_free(b);
  return rc;
}

}  // extern "C"
