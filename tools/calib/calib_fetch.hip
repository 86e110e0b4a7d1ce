// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE for the access widths libmsevb_hip's kernels use (MI355X_MICROARCH.md, HBM section:
// "other access widths are uncalibrated: calibrate on a known byte count in your own access pattern").  Every kernel moves exactly
// BYTES bytes; tools/calib/run.sh runs one --pmc pass per counter and prints counter / BYTES per kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr size_t BYTES = 512ull << 20;

template <class T> __global__ void k_read(const T* __restrict__ p, size_t n, T* __restrict__ sink) {      // coalesced stream, sizeof(T) bytes per lane
  T acc = p[0];
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const T v = p[i]; if (((const char*)&v)[0] == 77) acc = v; }
  if (((const char*)&acc)[1] == 99) sink[0] = acc;
}
template <class T> __global__ void k_write(T* __restrict__ p, size_t n, T v) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void k_atomic_f64(double* __restrict__ p, size_t n) {      // one fp64 atomic per element, distinct addresses
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) atomicAdd(&p[i], 1.0);
}
__global__ void k_gather_32B(const double4* __restrict__ p, size_t n, double4* __restrict__ sink) {      // 32-byte records at pseudo-random indices (the tile staging of the pair loop)
  double4 acc = p[0];
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const size_t j = (i * 2654435761ull) % n; const double4 v = p[j]; if (v.x == 77.0) acc = v; }
  if (acc.y == 99.0) sink[0] = acc;
}

int main() {
  char* buf; CK(hipMalloc(&buf, BYTES + 4096)); CK(hipMemset(buf, 0, BYTES));
  char* sink; CK(hipMalloc(&sink, 4096));
  const dim3 g(256 * 16), b(256);
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(k_read<unsigned>, g, b, 0, 0, (const unsigned*)buf, BYTES / 4, (unsigned*)sink);
    hipLaunchKernelGGL(k_read<uint2>, g, b, 0, 0, (const uint2*)buf, BYTES / 8, (uint2*)sink);
    hipLaunchKernelGGL(k_read<double2>, g, b, 0, 0, (const double2*)buf, BYTES / 16, (double2*)sink);
    hipLaunchKernelGGL(k_read<double4>, g, b, 0, 0, (const double4*)buf, BYTES / 32, (double4*)sink);
    hipLaunchKernelGGL(k_gather_32B, g, b, 0, 0, (const double4*)buf, BYTES / 32, (double4*)sink);
    hipLaunchKernelGGL(k_write<unsigned>, g, b, 0, 0, (unsigned*)buf, BYTES / 4, 0u);
    hipLaunchKernelGGL(k_write<double>, g, b, 0, 0, (double*)buf, BYTES / 8, 0.0);
    hipLaunchKernelGGL(k_write<double2>, g, b, 0, 0, (double2*)buf, BYTES / 16, make_double2(0.0, 0.0));
    hipLaunchKernelGGL(k_atomic_f64, g, b, 0, 0, (double*)buf, BYTES / 8);
  }
  CK(hipDeviceSynchronize());
  printf("bytes per kernel launch: %zu\n", BYTES);
  return 0;
}
