// Probe of hipGraph behaviour on the GPU box (design input for capturing the MS-EVB step):
//   (1) host cost of launching a 4-branch, ~40-kernel DAG as stream launches vs. as one hipGraphLaunch,
//   (2) whether the branches of a launched graph overlap on the GPU like four application streams do,
//   (3) whether a kernel of one branch may spin on a mark a kernel of another branch sets (device-side wait inside a graph).
// build: hipcc -O2 --offload-arch=gfx950 tools/graph_probe.hip -o gpurun_out/graph_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void k_busy(long long ticks, double* sink) {      // every workgroup stays for `ticks` of the 100 MHz clock
  const long long t0 = wall_clock64();
  double x = threadIdx.x;
  while (wall_clock64() - t0 < ticks) x = x * 1.0000001 + 1e-9;
  if (x == 123.456) *sink = x;
}
__global__ void k_small(int n, double* p) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1.0; }
__global__ void k_mark(long long* m, long long v) { __hip_atomic_store(m, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
__global__ void k_wait(const long long* m, long long v, int* timed_out) {
  const long long t0 = wall_clock64();
  while (__hip_atomic_load(m, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != v) {
    if (wall_clock64() - t0 > 50000000LL) { *timed_out = 1; break; }      // 0.5 s
    __builtin_amdgcn_s_sleep(8);
  }
}
__global__ void k_stamp(unsigned long long* out) { *out = wall_clock64(); }
// linear graphs (one per stream): the step number lives on the device, one counter per stream, advanced by the first kernel of the stream's graph; a mark carries the
// producer's step number, a waiter compares with its own - nothing in a kernel node's arguments changes from launch to launch
__global__ void k_tick(long long* cnt) { *cnt = *cnt + 1; }
__global__ void k_mark_cnt(long long* m, const long long* cnt) { __hip_atomic_store(m, *cnt, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
__global__ void k_wait_cnt(const long long* m, const long long* cnt, int* timed_out) {
  const long long want = *cnt, t0 = wall_clock64();
  while (__hip_atomic_load(m, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) {
    if (wall_clock64() - t0 > 50000000LL) { *timed_out = 1; break; }
    __builtin_amdgcn_s_sleep(8);
  }
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Ctx { hipStream_t s[4]; hipEvent_t fork, join[4], mid[4]; double* buf; long long* mark; int* to; unsigned long long* stamps; };

// the DAG: main forks three side branches; every branch = `big` (long kernel on a quarter of the GPU) + `nsmall` small kernels; side branch 1 ends in a mark,
// the main branch (optionally) waits for that mark on the device; everything joins on main
static int enqueue(Ctx& c, int nsmall, long long big_ticks, bool spin, long long seq) {
  CK(hipEventRecord(c.fork, c.s[0]));
  for (int b = 1; b < 4; ++b) CK(hipStreamWaitEvent(c.s[b], c.fork, 0));
  hipLaunchKernelGGL(k_stamp, dim3(1), dim3(1), 0, c.s[0], c.stamps);
  for (int b = 0; b < 4; ++b) {
    hipLaunchKernelGGL(k_busy, dim3(64), dim3(256), 0, c.s[b], big_ticks, c.buf);
    hipLaunchKernelGGL(k_stamp, dim3(1), dim3(1), 0, c.s[b], c.stamps + 1 + b);
    for (int k = 0; k < nsmall; ++k) hipLaunchKernelGGL(k_small, dim3(64), dim3(256), 0, c.s[b], 64 * 256, c.buf + 65536 * (b + 1));
  }
  if (spin) {
    hipLaunchKernelGGL(k_mark, dim3(1), dim3(1), 0, c.s[1], c.mark, seq);
    hipLaunchKernelGGL(k_wait, dim3(1), dim3(1), 0, c.s[0], c.mark, seq, c.to);
  }
  for (int b = 1; b < 4; ++b) { CK(hipEventRecord(c.join[b], c.s[b])); CK(hipStreamWaitEvent(c.s[0], c.join[b], 0)); }
  hipLaunchKernelGGL(k_stamp, dim3(1), dim3(1), 0, c.s[0], c.stamps + 5);
  return 0;
}

int main() {
  Ctx c;
  for (int b = 0; b < 4; ++b) CK(hipStreamCreateWithFlags(&c.s[b], hipStreamNonBlocking));
  CK(hipEventCreateWithFlags(&c.fork, hipEventDisableTiming));
  for (int b = 0; b < 4; ++b) { CK(hipEventCreateWithFlags(&c.join[b], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&c.mid[b], hipEventDisableTiming)); }
  CK(hipMalloc(&c.buf, 8 * 65536 * 8)); CK(hipMemset(c.buf, 0, 8 * 65536 * 8));
  CK(hipMalloc(&c.mark, 8)); CK(hipMemset(c.mark, 0, 8));
  CK(hipMalloc(&c.to, 4)); CK(hipMemset(c.to, 0, 4));
  CK(hipHostMalloc((void**)&c.stamps, 64, hipHostMallocDefault));
  CK(hipDeviceSynchronize());
  const int nsmall = 9;      // 4 x (1 + 1 + 9) + 2 = 46 kernels
  const long long big = 10000;      // 100 us
  for (int spin = 0; spin < 2; ++spin) {
    long long seq = 0;
    // ---- stream launches
    for (int w = 0; w < 5; ++w) { if (enqueue(c, nsmall, big, spin, ++seq)) return 1; }
    CK(hipDeviceSynchronize());
    double host = 0, wall0 = now_us();
    const int reps = 50;
    double gpu_span = 0, gpu_big[4] = {0, 0, 0, 0};
    for (int r = 0; r < reps; ++r) {
      const double t0 = now_us();
      if (enqueue(c, nsmall, big, spin, ++seq)) return 1;
      host += now_us() - t0;
      CK(hipStreamSynchronize(c.s[0]));
      gpu_span += (c.stamps[5] - c.stamps[0]) * 0.01;
      for (int b = 0; b < 4; ++b) gpu_big[b] += (c.stamps[1 + b] - c.stamps[0]) * 0.01;
    }
    double wall = now_us() - wall0;
    printf("streams spin=%d: host enqueue %.1f us / DAG, wall %.1f us / DAG, GPU span %.1f us, long kernels end at %.0f %.0f %.0f %.0f us\n", spin, host / reps, wall / reps, gpu_span / reps, gpu_big[0] / reps, gpu_big[1] / reps, gpu_big[2] / reps, gpu_big[3] / reps);
    // ---- the same DAG captured once, launched as a graph (the mark value is baked in: the waiter compares with a constant, the marker re-stores it - only the first launch really waits;
    //      enough to see whether the waiter can deadlock: the mark is reset before every launch)
    hipGraph_t g = nullptr; hipGraphExec_t ex = nullptr;
    CK(hipStreamBeginCapture(c.s[0], hipStreamCaptureModeRelaxed));
    if (enqueue(c, nsmall, big, spin, 777)) return 1;
    CK(hipStreamEndCapture(c.s[0], &g));
    size_t nn = 0; CK(hipGraphGetNodes(g, nullptr, &nn));
    const double ti0 = now_us();
    CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
    const double ti1 = now_us();
    for (int w = 0; w < 5; ++w) { CK(hipMemsetAsync(c.mark, 0, 8, c.s[0])); CK(hipGraphLaunch(ex, c.s[0])); }
    CK(hipDeviceSynchronize());
    host = 0; gpu_span = 0; for (int b = 0; b < 4; ++b) gpu_big[b] = 0; wall0 = now_us();
    for (int r = 0; r < reps; ++r) {
      CK(hipMemsetAsync(c.mark, 0, 8, c.s[0]));
      const double t0 = now_us();
      CK(hipGraphLaunch(ex, c.s[0]));
      host += now_us() - t0;
      CK(hipStreamSynchronize(c.s[0]));
      gpu_span += (c.stamps[5] - c.stamps[0]) * 0.01;
      for (int b = 0; b < 4; ++b) gpu_big[b] += (c.stamps[1 + b] - c.stamps[0]) * 0.01;
    }
    wall = now_us() - wall0;
    int to = 0; CK(hipMemcpy(&to, c.to, 4, hipMemcpyDeviceToHost));
    printf("graph   spin=%d: %zu nodes, instantiate %.0f us, host launch %.1f us / DAG, wall %.1f us / DAG, GPU span %.1f us, long kernels end at %.0f %.0f %.0f %.0f us, device wait timed out: %d\n", spin, nn, ti1 - ti0, host / reps, wall / reps,
           gpu_span / reps, gpu_big[0] / reps, gpu_big[1] / reps, gpu_big[2] / reps, gpu_big[3] / reps, to);
    CK(hipMemset(c.to, 0, 4));
    CK(hipGraphExecDestroy(ex)); CK(hipGraphDestroy(g));
  }
  // ---- the same work as FOUR LINEAR graphs, one per stream, captured separately; fork / join / the cross-branch wait through device-side marks
  {
    long long* cnt; long long* marks;      // cnt[4] step counters, marks[8]: 0 fork, 1..3 join of branch b, 4 the cross-branch mark
    CK(hipMalloc(&cnt, 4 * 8)); CK(hipMemset(cnt, 0, 32)); CK(hipMalloc(&marks, 8 * 8)); CK(hipMemset(marks, 0, 64)); CK(hipDeviceSynchronize());
    hipGraph_t g[4]; hipGraphExec_t ex[4]; size_t nn[4];
    for (int b = 0; b < 4; ++b) {
      CK(hipStreamBeginCapture(c.s[b], hipStreamCaptureModeRelaxed));
      hipLaunchKernelGGL(k_tick, dim3(1), dim3(1), 0, c.s[b], cnt + b);
      if (b == 0) { hipLaunchKernelGGL(k_mark_cnt, dim3(1), dim3(1), 0, c.s[0], marks + 0, cnt + 0); hipLaunchKernelGGL(k_stamp, dim3(1), dim3(1), 0, c.s[0], c.stamps); }
      else hipLaunchKernelGGL(k_wait_cnt, dim3(1), dim3(1), 0, c.s[b], marks + 0, cnt + b, c.to);
      hipLaunchKernelGGL(k_busy, dim3(64), dim3(256), 0, c.s[b], big, c.buf);
      hipLaunchKernelGGL(k_stamp, dim3(1), dim3(1), 0, c.s[b], c.stamps + 1 + b);
      for (int k = 0; k < nsmall; ++k) hipLaunchKernelGGL(k_small, dim3(64), dim3(256), 0, c.s[b], 64 * 256, c.buf + 65536 * (b + 1));
      if (b == 1) hipLaunchKernelGGL(k_mark_cnt, dim3(1), dim3(1), 0, c.s[1], marks + 4, cnt + 1);
      if (b == 0) {
        hipLaunchKernelGGL(k_wait_cnt, dim3(1), dim3(1), 0, c.s[0], marks + 4, cnt + 0, c.to);
        for (int q = 1; q < 4; ++q) hipLaunchKernelGGL(k_wait_cnt, dim3(1), dim3(1), 0, c.s[0], marks + q, cnt + 0, c.to);
        hipLaunchKernelGGL(k_stamp, dim3(1), dim3(1), 0, c.s[0], c.stamps + 5);
      } else hipLaunchKernelGGL(k_mark_cnt, dim3(1), dim3(1), 0, c.s[b], marks + b, cnt + b);
      CK(hipStreamEndCapture(c.s[b], &g[b]));
      CK(hipGraphGetNodes(g[b], nullptr, &nn[b]));
      CK(hipGraphInstantiate(&ex[b], g[b], nullptr, nullptr, 0));
    }
    for (int order = 0; order < 2; ++order) {      // order 1: the side graphs first, the main one (which sets the fork mark) last
      for (int w = 0; w < 5; ++w) { for (int q = 0; q < 4; ++q) { const int b = order ? 3 - q : q; CK(hipGraphLaunch(ex[b], c.s[b])); } }
      CK(hipDeviceSynchronize());
      const int reps = 50;
      double host = 0, gpu_span = 0, gpu_big[4] = {0, 0, 0, 0}, wall0 = now_us();
      for (int r = 0; r < reps; ++r) {
        const double t0 = now_us();
        for (int q = 0; q < 4; ++q) { const int b = order ? 3 - q : q; CK(hipGraphLaunch(ex[b], c.s[b])); }
        host += now_us() - t0;
        for (int b = 0; b < 4; ++b) CK(hipStreamSynchronize(c.s[b]));
        gpu_span += (c.stamps[5] - c.stamps[0]) * 0.01;
        for (int b = 0; b < 4; ++b) gpu_big[b] += (c.stamps[1 + b] - c.stamps[0]) * 0.01;
      }
      const double wall = now_us() - wall0;
      int to = 0; CK(hipMemcpy(&to, c.to, 4, hipMemcpyDeviceToHost));
      printf("4 linear graphs (launch order %s): nodes %zu %zu %zu %zu, host launch %.1f us / DAG, wall %.1f us / DAG, GPU span %.1f us, long kernels end at %.0f %.0f %.0f %.0f us, device wait timed out: %d\n", order ? "side first" : "main first",
             nn[0], nn[1], nn[2], nn[3], host / reps, wall / reps, gpu_span / reps, gpu_big[0] / reps, gpu_big[1] / reps, gpu_big[2] / reps, gpu_big[3] / reps, to);
    }
    // one linear graph alone: host cost per node
    { const int reps = 200; CK(hipMemset(marks, 0x7f, 8)); CK(hipDeviceSynchronize()); double host = 0; for (int r = 0; r < reps; ++r) { const double t0 = now_us(); CK(hipGraphLaunch(ex[2], c.s[2])); host += now_us() - t0; CK(hipStreamSynchronize(c.s[2])); }
      printf("one linear graph of %zu kernel nodes: host launch %.1f us\n", nn[2], host / reps); }
  }
  // ---- serial reference: the four long kernels one after the other would take 4 x 100 us
  printf("(long kernel = 100 us on 64 workgroups; four of them overlap if their branches run concurrently)\n");
  return 0;
}
