// Private definitions of libmsevb_hip.so (gfx950 only).  Public ABI: include/msevb_hip.h
#pragma once
#include <hip/hip_runtime.h>
#include <rocfft/rocfft.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <cstring>
#include <ctime>
#include <string>
#include <utility>
#include <map>
#include <mutex>
#include <vector>

#include "../../include/msevb_hip.h"

constexpr int MAX_MOL_ATOMS = 8;
constexpr int MAX_TYPES = 32;
constexpr int MAX_MOLTYPES = 10;
constexpr int MAX_BONDED = 16;
constexpr int TT_GRID = 1000;
constexpr double TT_MAX = 50.0;
constexpr int MAX_INTERACTION = 15;
constexpr int MAX_DIABATS = 80;
constexpr int MAX_CHAIN = 4;
constexpr int WAVE = 64;

#define HIPCHK(call)                                                                                       \
  do {                                                                                                     \
    hipError_t e_ = (call);                                                                                \
    if (e_ != hipSuccess) {                                                                                \
      char b_[512];                                                                                        \
      snprintf(b_, sizeof b_, "%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_));         \
      c->err = b_;                                                                                         \
      return MSEVB_ERR_HIP;                                                                                \
    }                                                                                                      \
  } while (0)
#define FAIL(code, msg) do { c->err = (msg); return (code); } while (0)
#define CHK(call) do { int r_ = (call); if (r_ != MSEVB_OK) return r_; } while (0)

// ---- capture / replay of a whole MS-EVB force call ("step graph", option step_graph; evb.hip) ------------------------------------
// Between two changes of the diabat tree a force call enqueues the same ~35 kernels, event records / waits and fills with the same arguments (what changes from
// call to call - sequence numbers, the pair list's epoch - is read from a device-resident step block).  The call's host code is therefore run in one of these modes:
//   0 direct   every launch / event operation goes to its stream (rebuild steps, calls that re-enumerate the tree, timed calls, step_graph = 0);
//   3 learn    a dry pass, nothing issued: which event records are waited for on ANOTHER stream (only those become device-side marks); the call is repeated;
//   1 capture  every stream of the call is in a stream capture of its OWN (four linear graphs, LinCap below): the same calls build the graphs; every launch is also
//              hashed (function, geometry, stream, argument bytes);
//   2 replay   the same host code runs again for its bookkeeping, launches and event operations are NOT issued but hashed; if the hash equals the capture's,
//              the call is four hipGraphLaunch calls - otherwise the call is repeated in mode 0 (nothing had been issued) and the graphs are dropped.
// (Round 5's first form - ONE graph holding all four branches - saved no host time: this runtime enqueues a graph with branches node by node.)
// All device work of the library goes through the wrappers below (the HIP names are redirected to them at the end of this block).
enum StepBlockSlot { SB_R = 0, SB_A, SB_S, SB_B, SB_EPOCH, SB_N = 8 };      // device-resident step block: what changes between two launches of a captured force call
struct StepRec { int mode = 0; unsigned long long hash = 0xcbf29ce484222325ull; int n = 0; std::vector<std::pair<const void*, unsigned long long>>* trace = nullptr; /* MSEVB_GRAPH_DEBUG: (kernel or operation tag, running hash) per operation */ };
extern thread_local StepRec* t_step_rec;      // non-null only inside a captured / replayed force call of the calling thread (api.hip)
extern std::atomic<int> g_live_contexts;
extern long g_alloc_epoch;                    // bumped by every (re)allocation or release of a device buffer: addresses baked into a captured graph may be stale
inline void step_hash_bytes(StepRec& r, const void* p, size_t n) {
  const unsigned char* b = (const unsigned char*)p;
  unsigned long long h = r.hash;
  size_t k = 0;
  for (; k + 8 <= n; k += 8) { unsigned long long w; memcpy(&w, b + k, 8); h = (h ^ w) * 0x100000001b3ull; h ^= h >> 29; }
  for (; k < n; ++k) h = (h ^ b[k]) * 0x100000001b3ull;
  r.hash = h;
}
template <class T> inline void step_hash_one(StepRec& r, const T& v) { step_hash_bytes(r, &v, sizeof(T)); }
inline void step_hash_one(StepRec& r, std::nullptr_t) { const void* z = nullptr; step_hash_bytes(r, &z, sizeof z); }
// MSEVB_API_TIMING=1 (diagnostic): host time and call count per kind of runtime call the library's wrappers issue (0 kernel launch, 1 event record, 2 stream wait,
// 3 fill / copy command, 4 graph launch), printed per MS-EVB call when the context is released - where the host's enqueue time of a step goes
struct ApiProf { int on = -1; double us[5] = {0, 0, 0, 0, 0}; long n[5] = {0, 0, 0, 0, 0}; };
extern ApiProf g_api_prof;
inline bool api_prof_on() { if (g_api_prof.on < 0) g_api_prof.on = getenv("MSEVB_API_TIMING") ? 1 : 0; return g_api_prof.on == 1; }
inline double api_now_us() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3; }
struct ApiScope { int k; double t0; explicit ApiScope(int kind) : k(api_prof_on() ? kind : -1), t0(k >= 0 ? api_now_us() : 0.0) {} ~ApiScope() { if (k >= 0) { g_api_prof.us[k] += api_now_us() - t0; g_api_prof.n[k]++; } } };
// ---- the captured call as FOUR LINEAR graphs, one per stream (round 5; tools/graph_probe.hip, profiles/r05_graph_probe.txt) ---------------------------------------
// One hipGraph that holds the call's four branches costs the calling thread ~3 us per node on this runtime (it is enqueued node by node, like the stream launches
// it replaces); a graph that is ONE chain of kernel nodes costs ~3 us for the whole chain (14 nodes: 3.0 us; four chains of 14-19 nodes: 10.7 us).  So every stream of
// the call is captured on its own (four captures, no edges between them) and what used to order the streams - event record on one, wait on another - becomes a pair
// of operations of one-thread kernels inside the chains (k_lg_ops): a mark stores {step number of the producing stream} x 64 + {ordinal of this record of the event in
// the call}, a wait polls for that value; what a stream does between two of its kernels is ONE such kernel.  The step number lives on the device, one counter per
// stream, advanced by the first operation of the stream's chain: no node's arguments change from launch to launch.  Records nobody waits for on another
// stream emit nothing (found by a dry "learn" pass over the call, mode 3, before the capture pass).
// what a stream does between two of its kernels, in order: up to 8 operations {kind, slot, ordinal} in ONE one-thread kernel (kind 0 = advance the stream's step counter,
// 1 = store a mark, 2 = wait for a mark).  Order matters (a record in front of a wait must stay in front of it: two streams that each mark first and wait second would
// wait for each other otherwise), so the list is kept as issued.
struct LinOps { int n = 0; int kind[8] = {0, 0, 0, 0, 0, 0, 0, 0}; int slot[8] = {0, 0, 0, 0, 0, 0, 0, 0}; int ord[8] = {0, 0, 0, 0, 0, 0, 0, 0}; };
constexpr int LIN_NST = 4;      // main, grid, descriptor, coupling stream (a fifth stream was tried in round 5 - the stencil kernels beside the charge spreading - and made the step 20 % slower: the runtime's streams share four hardware queues, HISTORY.md F)
struct LinCap {
  hipStream_t st[LIN_NST] = {nullptr, nullptr, nullptr, nullptr};
  long long* cnt = nullptr; long long* marks = nullptr; double* fault = nullptr;      // device: cnt[8], marks[64]; the sticky time-out flag (S_TIMEOUT)
  int nev = 0; hipEvent_t ev[60]; int ev_stream[60]; int ev_ord[60]; int ev_kpos[60];      // ev_kpos: kernels the recording stream had been given when the (latest) record was made
  int kcount[LIN_NST] = {0, 0, 0, 0}; int waited_kpos[LIN_NST][LIN_NST];      // per stream: kernels given so far; per (waiting, producing) stream: the latest position already waited for - a wait for an earlier position of the same stream adds nothing
  unsigned long long need[60];      // bit `ord` of need[slot]: that record of the event is waited for by another stream (learn pass -> capture pass)
  LinOps pend[LIN_NST];             // operations a stream has collected since its last kernel
  bool head_open = true; LinOps head;      // the main stream's operations in front of its first kernel: not captured - the step-block kernel, launched in front of the main chain at every replay, performs them
  bool bad = false, verbose = false; const char* why = nullptr; int n_marks = 0, n_waits = 0, n_op_kernels = 0;
  LinCap() { for (auto& r : waited_kpos) for (int& v : r) v = -1; }
  int stream_index(hipStream_t s) const { for (int b = 0; b < LIN_NST; ++b) if (st[b] && st[b] == s) return b; return -1; }
  int slot_of(hipEvent_t e) const { for (int k = 0; k < nev; ++k) if (ev[k] == e) return k; return -1; }
};
extern thread_local LinCap* t_lin;      // non-null during the learn (mode 3) and capture (mode 1) passes of a call (evb.hip)
#ifdef __HIPCC__
__device__ __forceinline__ void lin_run_ops(const LinOps& a, long long* cnt, long long* marks, double* fault) {
  const long long t0 = wall_clock64();
  bool waited = false;
  for (int k = 0; k < a.n; ++k) {
    if (a.kind[k] == 0) *cnt = *cnt + 1;
    else if (a.kind[k] == 1) __hip_atomic_store(marks + a.slot[k], *cnt * 64 + a.ord[k], __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    else {
      const long long want = *cnt * 64 + a.ord[k];
      while (__hip_atomic_load(marks + a.slot[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {      // (relaxed polls, one acquire at the end)
        if (wall_clock64() - t0 > 200000000LL) { if (fault) *fault = 2.0; break; }      // 2 s of the 100 MHz clock: the sticky flag the host reports as MSEVB_ERR_TIMEOUT
        __builtin_amdgcn_s_sleep(8);
      }
      waited = true;
    }
  }
  if (waited) __atomic_thread_fence(__ATOMIC_ACQUIRE);
}
[[maybe_unused]] static __global__ void k_lg_ops(long long* cnt, long long* marks, LinOps a, double* fault) { if (threadIdx.x == 0) lin_run_ops(a, cnt, marks, fault); }
inline void lin_flush(LinCap& L, int b) {      // what a stream has collected, as ONE kernel in front of whatever it does next
  if (L.pend[b].n == 0) return;
  if (b == 0 && L.head_open) { L.head = L.pend[0]; L.head_open = false; L.pend[0].n = 0; return; }      // (performed by the step-block kernel)
  LinOps a = L.pend[b];
  L.pend[b].n = 0; L.n_op_kernels++;
  if (L.verbose) { fprintf(stderr, "[msevb]   chain %d: ops", b); for (int k = 0; k < a.n; ++k) fprintf(stderr, " %s%d", a.kind[k] == 0 ? "tick" : a.kind[k] == 1 ? "mark e" : "wait e", a.kind[k] ? a.slot[k] : 0); fprintf(stderr, "\n"); }
  k_lg_ops<<<dim3(1), dim3(64), 0, L.st[b]>>>(L.cnt + b, L.marks, a, L.fault);
}
inline void lin_add_op(LinCap& L, int b, int kind, int slot, int ord) {
  if (L.pend[b].n == 8) lin_flush(L, b);
  LinOps& p = L.pend[b];
  p.kind[p.n] = kind; p.slot[p.n] = slot; p.ord[p.n] = ord; p.n++;
}
inline void lin_before_op(hipStream_t s) {      // capture pass: anything issued to a stream comes behind its pending operations
  LinCap* L = t_lin;
  if (!L) return;
  const int b = L->stream_index(s);
  if (b < 0) { L->bad = true; L->why = "an operation on a stream outside the call's four"; return; }
  lin_flush(*L, b);
  if (b == 0) L->head_open = false;
}
template <class K, class... A>
inline void step_launch(K kern, dim3 grid, dim3 block, size_t shm, hipStream_t st, A&&... a) {
  StepRec* r = t_step_rec;
  if (r) {
    const void* fn = (const void*)kern;
    step_hash_one(*r, fn); step_hash_one(*r, grid.x); step_hash_one(*r, grid.y); step_hash_one(*r, grid.z); step_hash_one(*r, block.x); step_hash_one(*r, shm); step_hash_one(*r, st);
    (step_hash_one(*r, a), ...);
    r->n++;
    if (r->trace) r->trace->push_back({fn, r->hash});
    if (t_lin && (r->mode == 1 || r->mode == 3)) { const int b_ = t_lin->stream_index(st); if (b_ >= 0) t_lin->kcount[b_]++; }
    if (r->mode >= 2) return;
    if (r->mode == 1) { lin_before_op(st); if (t_lin && t_lin->verbose) fprintf(stderr, "[msevb]   chain %d: %s\n", t_lin->stream_index(st), hipKernelNameRefByPtr(fn, st)); }
  }
  ApiScope aps(0);
  kern<<<grid, block, shm, st>>>(std::forward<A>(a)...);
}
#endif
inline bool step_skip(int tag, const void* a, const void* b, size_t n) {      // event / fill operations: hashed like launches, skipped on replay (mode 2) and in the learn pass (mode 3)
  StepRec* r = t_step_rec;
  if (!r) return false;
  step_hash_one(*r, tag); step_hash_one(*r, a); step_hash_one(*r, b); step_hash_one(*r, n); r->n++;
  if (r->trace) r->trace->push_back({(const void*)(size_t)tag, r->hash});
  return r->mode >= 2;
}
inline hipError_t step_event_record(hipEvent_t e, hipStream_t s) {
  const bool skip = step_skip(1, e, s, 0);
  LinCap* L = t_lin;
  if (L && t_step_rec) {      // learn / capture pass: the record gets its slot and ordinal; the capture pass emits a mark where the learn pass saw a cross-stream wait
    const int b = L->stream_index(s);
    int k = L->slot_of(e);
    if (k < 0 && L->nev < 60) { k = L->nev++; L->ev[k] = e; L->ev_ord[k] = 0; if (t_step_rec->mode == 3) L->need[k] = 0; }
    if (b < 0 || k < 0 || L->ev_ord[k] >= 62) { L->bad = true; L->why = "event table of the linear capture"; return hipSuccess; }
    L->ev_stream[k] = b; L->ev_ord[k]++; L->ev_kpos[k] = L->kcount[b];
#ifdef __HIPCC__
    if (t_step_rec->mode == 1 && ((L->need[k] >> L->ev_ord[k]) & 1ull)) { L->n_marks++; lin_add_op(*L, b, 1, k, L->ev_ord[k]); }
#endif
    return hipSuccess;
  }
  if (skip) return hipSuccess;
  ApiScope aps(1);
  return (hipEventRecord)(e, s);
}
inline hipError_t step_stream_wait(hipStream_t s, hipEvent_t e, unsigned f) {
  const bool skip = step_skip(2, e, s, f);
  LinCap* L = t_lin;
  if (L && t_step_rec) {
    const int b = L->stream_index(s), k = L->slot_of(e);
    if (b < 0) { L->bad = true; L->why = "a wait on a stream outside the call's four"; return hipSuccess; }
    if (k < 0) { L->bad = true; L->why = "a wait for an event that was not recorded inside the call"; return hipSuccess; }
    if (L->ev_stream[k] == b) return hipSuccess;      // same stream: stream order
    { int& wk = L->waited_kpos[b][L->ev_stream[k]]; if (L->ev_kpos[k] <= wk) return hipSuccess; wk = L->ev_kpos[k]; }      // this stream already waits for a record the producer made at the same point of its chain or later
    if (t_step_rec->mode == 3) { L->need[k] |= 1ull << L->ev_ord[k]; return hipSuccess; }
#ifdef __HIPCC__
    lin_add_op(*L, b, 2, k, L->ev_ord[k]); L->n_waits++;
#endif
    return hipSuccess;
  }
  if (skip) return hipSuccess;
  ApiScope aps(2);
  return (hipStreamWaitEvent)(s, e, f);
}
#ifdef __HIPCC__
inline void lin_count(hipStream_t s) { if (t_lin && t_step_rec && (t_step_rec->mode == 1 || t_step_rec->mode == 3)) { const int b = t_lin->stream_index(s); if (b >= 0) t_lin->kcount[b]++; } }      // (fills and copies are positions of a chain like kernels)
inline hipError_t step_memset_async(void* p, int v, size_t n, hipStream_t s) { lin_count(s); if (step_skip(3 + 16 * (v & 0xff), p, s, n)) return hipSuccess; lin_before_op(s); ApiScope aps(3); return (hipMemsetAsync)(p, v, n, s); }
inline hipError_t step_memset_d32_async(hipDeviceptr_t p, int v, size_t n, hipStream_t s) { lin_count(s); if (step_skip(4 + 16 * v, (const void*)p, s, n)) return hipSuccess; lin_before_op(s); ApiScope aps(3); return (hipMemsetD32Async)(p, v, n, s); }
inline hipError_t step_memcpy_async(void* d, const void* src, size_t n, hipMemcpyKind k, hipStream_t s) { lin_count(s); if (step_skip(5 + 16 * (int)k, d, src, n)) return hipSuccess; lin_before_op(s); ApiScope aps(3); return (hipMemcpyAsync)(d, src, n, k, s); }
#else
inline hipError_t step_memset_async(void* p, int v, size_t n, hipStream_t s) { if (step_skip(3 + 16 * (v & 0xff), p, s, n)) return hipSuccess; return (hipMemsetAsync)(p, v, n, s); }
inline hipError_t step_memset_d32_async(hipDeviceptr_t p, int v, size_t n, hipStream_t s) { if (step_skip(4 + 16 * v, (const void*)p, s, n)) return hipSuccess; return (hipMemsetD32Async)(p, v, n, s); }
inline hipError_t step_memcpy_async(void* d, const void* src, size_t n, hipMemcpyKind k, hipStream_t s) { if (step_skip(5 + 16 * (int)k, d, src, n)) return hipSuccess; return (hipMemcpyAsync)(d, src, n, k, s); }
#endif
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kern, grid, block, shm, st, ...) step_launch(kern, dim3(grid), dim3(block), (size_t)(shm), (st), ##__VA_ARGS__)
#define hipEventRecord(e, s) step_event_record((e), (s))
#define hipStreamWaitEvent(s, e, f) step_stream_wait((s), (e), (f))
#define hipMemsetAsync(p, v, n, s) step_memset_async((p), (v), (n), (s))
#define hipMemsetD32Async(p, v, n, s) step_memset_d32_async((p), (v), (n), (s))
#define hipMemcpyAsync(d, src, n, k, s) step_memcpy_async((d), (src), (n), (k), (s))

// ---- device-side parameter blocks -------------------------------------------------------------------
struct PairParam {  // one (type_i,type_j) entry of atype_vdw_type / atype_vdw_parameter, 64 B
  double p[6];      // LJ: C12,C6 ; SAPT: A,B,C6,C8,C10,C12
  int type;         // 0 LJ, 1 SAPT, -1 none
  int nonzero;      // 0: every coefficient is zero, the vdW term contributes exactly 0
};

struct MolTypeDev {
  int natom, nb, na, nd;
  int atype[MAX_MOL_ATOMS];
  int bond[MAX_BONDED][2], bond_type[MAX_BONDED];
  double bond_par[MAX_BONDED][3];
  int ang[MAX_BONDED][3], ang_type[MAX_BONDED];
  double ang_par[MAX_BONDED][2];
  int dih[MAX_BONDED][4], dih_type[MAX_BONDED];
  double dih_par[MAX_BONDED][6];
  int excl[MAX_MOL_ATOMS][MAX_MOL_ATOMS];
  PairParam nb_par[MAX_MOL_ATOMS][MAX_MOL_ATOMS];  // 1-4 parameters already substituted where excl==2
};

struct BoxDev { double L[3], invL[3]; };

// XCD-aware block order: the dispatcher deals workgroups round-robin over the 8 XCDs (block b -> XCD b % 8), each with its
// own 4 MB L2.  Kernels that walk the cell-sorted atoms map hardware block b to logical block (b % 8) * (nb / 8) + b / 8, so
// one XCD works on one contiguous eighth of the sorted range (a slab of cells) and its L2 holds only that slab's data.
// nb must be a multiple of 8 (launch helpers round up; surplus blocks exit).
#ifdef __HIPCC__
__device__ __forceinline__ int xcd_block(int b, int nb) { return (b & 7) * (nb >> 3) + (b >> 3); }
#endif
inline int round_up8(int n) { return (n + 7) & ~7; }
struct msevb_ctx;

// The (up to 27) distinct neighbour cells of cell cidx, in the canonical order shared by the list build, the LDS-tiled pair
// kernel and the host-side decoder of msevb_get_verlet_list.  Returns the number of slots.
#ifdef __HIPCC__
__host__ __device__
#endif
inline int cell_neighbourhood(int cidx, int ncx, int ncy, int ncz, int* slots) {
  const int cx = cidx % ncx, cy = (cidx / ncx) % ncy, cz = cidx / (ncx * ncy);
  int n = 0;
  for (int oz = -1; oz <= 1; ++oz) {
    if ((ncz < 3 && oz == 1) || (ncz < 2 && oz == -1)) continue;
    const int zz = (cz + oz + ncz) % ncz;
    for (int oy = -1; oy <= 1; ++oy) {
      if ((ncy < 3 && oy == 1) || (ncy < 2 && oy == -1)) continue;
      const int yy = (cy + oy + ncy) % ncy;
      for (int ox = -1; ox <= 1; ++ox) {
        if ((ncx < 3 && ox == 1) || (ncx < 2 && ox == -1)) continue;
        const int xx = (cx + ox + ncx) % ncx;
        slots[n++] = xx + ncx * (yy + ncy * zz);
      }
    }
  }
  return n;
}

// The HALF neighbourhood of cell cidx: slot 0 = the cell itself, slots 1..13 = its 13 "forward" neighbours (offset (ox,oy,oz) with
// oz > 0, or oz == 0 and oy > 0, or oz == oy == 0 and ox > 0), so that every unordered pair of neighbouring cells appears in exactly one
// half neighbourhood.  Needs >= 3 cells per axis (the +1 and -1 neighbours must be different cells); the users ask for >= 4.
// Layout of the LDS tile of the half-list pair kernel and of its tile-local list entries (pair.hip, neighbor.hip, api.hip decoder).
constexpr int HALF_SLOTS = 14;
#ifdef __HIPCC__
__host__ __device__
#endif
inline void cell_half_neighbourhood(int cidx, int ncx, int ncy, int ncz, int* slots) {
  const int cx = cidx % ncx, cy = (cidx / ncx) % ncy, cz = cidx / (ncx * ncy);
  int n = 0;
  slots[n++] = cidx;
  for (int oz = 0; oz <= 1; ++oz)
    for (int oy = (oz ? -1 : 0); oy <= 1; ++oy)
      for (int ox = ((oz || oy) ? -1 : 1); ox <= 1; ++ox) {
        const int xx = (cx + ox + ncx) % ncx, yy = (cy + oy + ncy) % ncy, zz = (cz + oz + ncz) % ncz;
        slots[n++] = xx + ncx * (yy + ncy * zz);
      }
}

// Verlet-list entry: sorted index of the neighbour in the low 24 bits, its atom type in the high 8 (the pair kernel then needs
// no per-neighbour type gather).  Limits: N < 2^24 atoms, MAX_TYPES <= 256; the list is rebuilt whenever the topology changes.
constexpr int NBR_IDX_BITS = 24;
constexpr unsigned NBR_IDX_MASK = (1u << NBR_IDX_BITS) - 1u;
// In tile mode (c->pair_tiles) the list is a HALF list and the index is LOCAL: the position of the neighbour inside the concatenation
// of the cells of the owning atom's half neighbourhood (cell_half_neighbourhood order) - exactly the layout of the LDS tile the pair
// kernel stages per cell (positions 32 B + force accumulators 24 B per tile atom).
constexpr int PAIR_TILE_LDS_BYTES = 156 * 1024;      // of the 160 KiB per CU; the rest holds the static reductions
constexpr int PAIR_TILE_ATOM_BYTES = 56;
constexpr int PAIR_CELL_MAX = 512;                   // most atoms of ONE cell the half-list pair kernel keeps row metadata for (a cell of edge r_v holds ~135 at liquid density)
constexpr int NBR_TILE_CAP = 8000;                   // most atoms of a cell neighbourhood the list build can stage in LDS
constexpr int NBR_TILE_CAP2 = 4032;                  // ... while two workgroups still fit a CU (2 x (4032 x 20 B + 0.9 KB static) <= 160 KB)
constexpr int NBR_TILE_BYTES = 20;                   // per staged atom: float4 {x,y,z,molecule} + the entry word
constexpr int NBR_TILE_MIN_CELLS = 64;               // below this many cells the wave-per-atom build fills the chip better             // fewer cells than this cannot fill the chip with one block per cell

// host mirror of a molecule type (as set through the ABI)
struct MolTypeHost {
  int natom = 0;
  std::vector<int> atype, bonds, angles, dihedrals, excl, reactive_protons, reactive_basic;
};

struct PhaseTimer {   // HIP events on the library's stream around a phase; harvested lazily (no host sync inside a step)
  std::string name;
  std::vector<hipEvent_t> ev;   // pairs (start, stop)
  size_t used = 0;              // events handed out since the last harvest
  double ms = 0;
  long calls = 0;
};

enum Phase { PH_VERLET = 0, PH_SORT, PH_PAIR, PH_MOLECULE, PH_SPREAD, PH_FFT, PH_GATHER, PH_EVB_REAL, PH_EVB_RECIP, PH_EVB_BLEND, PH_INTEGRATE, PH_COUNT };

// ---- guard allocator (diagnostic, MSEVB_GUARD=1 / 2) -------------------------------------------------------------------------------
// Every device buffer gets its own virtual-address reservation whose mapped part ENDS on the buffer's last byte (mode 1: the page
// behind it is reserved but never mapped) or STARTS on its first byte (mode 2: the page in front of it is unmapped).  A kernel that
// reads or writes one element past / in front of an array then faults at once, at an address guard_report() can name, instead of
// landing in whatever the allocator placed there.  MSEVB_GUARD_LOG=1 prints every allocation with the source line that made it.
struct GuardRec { void* base; size_t reserved, mapped_off, mapped; hipMemGenericAllocationHandle_t h; void* user; size_t bytes; const char* file; int line; };
inline std::map<void*, GuardRec>& guard_registry() { static std::map<void*, GuardRec> r; return r; }
inline std::mutex& guard_mutex() { static std::mutex m; return m; }
inline int guard_mode() { static const int m = getenv("MSEVB_GUARD") ? atoi(getenv("MSEVB_GUARD")) : 0; return m; }
inline void* guard_alloc(size_t bytes, const char* file, int line) {
  int dev = 0; if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = dev;
  size_t gran = 0; if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum) != hipSuccess || !gran) return nullptr;
  const size_t mapped = (bytes + gran - 1) / gran * gran, reserved = mapped + 2 * gran;
  GuardRec r = {}; r.reserved = reserved; r.mapped = mapped; r.mapped_off = gran; r.bytes = bytes; r.file = file; r.line = line;
  if (hipMemAddressReserve(&r.base, reserved, gran, nullptr, 0) != hipSuccess) return nullptr;
  if (hipMemCreate(&r.h, mapped, &prop, 0) != hipSuccess) { (void)hipMemAddressFree(r.base, reserved); return nullptr; }
  char* m0 = (char*)r.base + gran;
  hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
  if (hipMemMap(m0, mapped, 0, r.h, 0) != hipSuccess || hipMemSetAccess(m0, mapped, &acc, 1) != hipSuccess) { (void)hipMemRelease(r.h); (void)hipMemAddressFree(r.base, reserved); return nullptr; }
  r.user = guard_mode() == 2 ? (void*)m0 : (void*)(m0 + mapped - bytes);      // mode 1: last byte = last mapped byte
  static const bool log = getenv("MSEVB_GUARD_LOG") != nullptr;
  if (log) fprintf(stderr, "msevb guard: %p..%p (%zu B) mapped %p..%p  %s:%d\n", r.user, (char*)r.user + bytes, bytes, m0, m0 + mapped, file, line);
  std::lock_guard<std::mutex> g(guard_mutex());
  guard_registry()[r.user] = r;
  return r.user;
}
inline bool guard_free(void* user) {
  GuardRec r;
  { std::lock_guard<std::mutex> g(guard_mutex()); auto it = guard_registry().find(user); if (it == guard_registry().end()) return false; r = it->second; guard_registry().erase(it); }
  (void)hipDeviceSynchronize();
  // the physical memory goes back, the ADDRESS RANGE is never handed out again: a kernel that still uses the buffer faults (use after free), and no
  // later mapping can appear at an address a translation cache may still hold (re-mapping freed ranges gave wrong copies in the stage entry points)
  (void)hipMemUnmap((char*)r.base + r.mapped_off, r.mapped); (void)hipMemRelease(r.h);
  return true;
}

template <class T>
struct DBuf {  // device buffer, exactly `count` elements: no kernel of this library may touch a byte outside [p, p + n)
  T* p = nullptr;
  size_t n = 0;
  int alloc(size_t count, const char* file = __builtin_FILE(), int line = __builtin_LINE()) {
    if (count <= n && p) return 0;
    release();
    const size_t bytes = (count ? count : 1) * sizeof(T);
    // MSEVB_DBUF_TAIL=<bytes> (diagnostic): extra mapped bytes behind every buffer - what round 3 shipped with (1024) to hide an overrun
    static const size_t tail = getenv("MSEVB_DBUF_TAIL") ? (size_t)atol(getenv("MSEVB_DBUF_TAIL")) : 0;
    if (guard_mode()) { p = (T*)guard_alloc(bytes, file, line); if (!p) return -1; }
    else if (hipMalloc((void**)&p, bytes + tail) != hipSuccess) { p = nullptr; return -1; }
    n = count; ++g_alloc_epoch;
    // MSEVB_POISON=1 (diagnostic): fresh device memory reads as NaN / -1 instead of whatever the allocator hands back (zeros in a young process,
    // another context's data later), so that a read of something nobody wrote shows in the first test that does it, not in the fortieth
    static const bool poison = getenv("MSEVB_POISON") != nullptr;
    if (poison) { (void)hipMemset(p, 0xFF, bytes); (void)hipDeviceSynchronize(); }      // (hipMemset of device memory may return early; the library's streams do not wait for the null stream)
    return 0;
  }
  // for buffers whose size follows the diabat count: grow with half as much again, so that a tree gaining a diabat or two during a run
  // does not free + allocate (a device synchronisation and milliseconds) every time
  int alloc_room(size_t count, const char* file = __builtin_FILE(), int line = __builtin_LINE()) { return (count <= n && p) ? 0 : alloc(count + count / 2, file, line); }
  void release() { if (p) ++g_alloc_epoch; if (p && !(guard_mode() && guard_free(p))) (void)hipFree(p); p = nullptr; n = 0; }
};

struct EvbHop { int donor_mol, proton_local, heavy_local, acceptor_mol, acceptor_heavy_local; };

struct msevb_ctx {
  std::string err;
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  // internal side stream for small latency-bound kernels that are independent of the main chain (per-molecule terms next to
  // the pair loop / charge spreading, off-diagonal EVB elements next to the diabat reciprocal space); fork/join by events
  // aux : PME front end + principal reciprocal gather, later the off-diagonal EVB elements
  // aux2: per-molecule terms, and for EVB the neighbourhood fetch, descriptor upload, diagonal real-space deltas, energy download
  // (three streams in all: with more, ROCm's 4 hardware queues alias and the intended overlap is lost)
  hipStream_t aux = nullptr, aux2 = nullptr, aux3 = nullptr;      // aux3: off-diagonal EVB chain (4 streams = ROCm's 4 hardware queues)
  hipEvent_t ev_fork = nullptr, ev_fork0 = nullptr /* PME front end leaves before the refresh kernels */, ev_join = nullptr, ev_join2 = nullptr;
  hipStream_t pme_st = nullptr;      // non-null while the PME front end is being enqueued on the side stream
  bool overlap = true;
  bool cells_valid = false;      // cell_start / perm are current (a list build is in progress or done)
  bool nbr_tiles = true;         // LDS-staged list build (workgroup per cell) when there are enough cells
  bool nbr_tiles_used = false;   // ... and whether the last build was one (msevb_get_info)
  bool graphs = true;            // replay the PME front end as a hipGraph on the side stream
  void* pme_graph_exec = nullptr; char pme_graph_key[128] = {0}; int pme_graph_nbk = 0;
  // multi-rank EVB: the principal diabat's pair loop, per-molecule terms and reciprocal-force accumulation are split over the ranks by
  // atom / molecule ranges (the partial principal forces meet in the all-reduce of the blend); set by msevb_comm_init, or by the
  // "shard_principal" option for virtual ranks (tests)
  double cpl_A[MAX_DIABATS] = {0};   // geometric coupling factor A of each owned diabat (off-diagonal stores hold dV, F_pd = -A dV)
  bool evb_element_stores = false;   // keep F_dd / F_pd of every Hamiltonian element (needed by msevb_get_evb_element_force); default: blended-potential path
  bool skip_gather_now = false;      // EVB fast path: the principal reciprocal force comes with the blended gather
  bool shard_principal = false;
  hipEvent_t ev_pos_ready = nullptr;      // set by the MS-EVB call around total_force_enqueue: positions of this step are final
  bool refresh_side_now = false; hipEvent_t ev_refresh = nullptr;      // MS-EVB call with the neighbourhood kernel on the main stream: the refresh kernels run beside it on aux3
  bool pme_split_now = false, pme_split_pending = false, evb_pme_split = true; int pme_split_slot = 0;      // MS-EVB fast path: spread now, solve later (behind the stencil-pair kernels on aux)
  bool skip_side_joins_now = false, evb_single_join = true;      // MS-EVB fast path: the main stream waits ONCE for the side streams (a barrier packet behind the pair loop costs several us each)
  bool defer_fold_now = false, fsorted_pending = false, evb_defer_fold = true;      // MS-EVB fast path: k_pair_fold's work is done by k_evb_blend
  int evb_coupling_blocks = 0;            // option: workgroups per coupling in k_evb_coupling (0: the built-in choice)
  bool evb_region_main = false;           // option: the neighbourhood kernel on the main stream (see evb.hip, step 1)
  bool molecules_after_pair_now = false, evb_molecules_after_pair = false;      // evb_defer_molecules = 2: the per-molecule kernel waits for the pair loop's end and runs beside the blend phase's transform and gather (an otherwise idle GPU)
  bool defer_molecules_now = false, evb_defer_molecules = false;      // MS-EVB call: the per-molecule kernel goes out BEHIND the coupling chain on aux3 (the host's
                                                                     // turn waits for that chain, only the blend for the per-molecule terms)
  bool shard_now = false;        // true while the EVB force call enqueues the principal diabat
  int pair_tiles_opt = -1;       // half-list LDS pair kernel: -1 / 1 = whenever the box has >= 4 cells per axis and the tile fits in LDS (default), 0 = never
  bool pair_tiles = false;       // format of the current list
  int tile_atoms_max = 0;        // largest 27-cell neighbourhood (atoms) the last build saw
  int half_tile_cap_seen = 0, gather_cap_seen = 0;      // largest tile capacity / per-cell staging capacity used so far: launch parameters that follow the list statistics only upwards (a captured force call stays valid across rebuilds)
  int half_tile_max = 0, half_tile_cap = 0;      // largest half neighbourhood the last build saw / tile size (atoms) the current half list was built for
  bool finalized = false;
  bool attr_pair_half[8] = {false, false, false, false, false, false, false, false};
  int pair_threads = 1024;       // workgroup size of the half-list pair kernel (option pair_threads: 512 leaves room on the CU for the side chains)      // hipFuncSetAttribute(MaxDynamicSharedMemorySize) done for this context's device

  // ---- configuration
  int N = 0, M = 0, K = 0, order = 6, spline_grid = 100000, erfc_grid = 100000;
  double box[3] = {0, 0, 0}, rc = 0, rv = 0, alpha = 0;
  double pi = 3.141592654, pi_sqrt = 1.772453851, conv = (double)1389.35465f;
  int nt = 0, nmt = 0;

  // ---- host copies of static data (0-based internally)
  std::vector<int> h_atype, h_mol_type, h_mol_natom, h_mol_atoms, h_mol_of;
  std::vector<double> h_chg, h_mass, h_atype_chg;
  std::vector<int> h_vdw_type, h_bond_type;
  std::vector<double> h_vdw_par, h_vdw_par14, h_bond_par;
  std::vector<int> h_ang_idx, h_dih_idx;
  std::vector<double> h_ang_par, h_dih_par;
  MolTypeHost h_mt[MAX_MOLTYPES];
  // caller-owned host arrays that keep coming back (a Fortran host's atom_data%xyz / %force) can be page-locked on their second use, so
  // that the copies go by DMA instead of through a staging buffer.  Opt-in: an array that is freed and whose address is
  // handed out again (numpy temporaries) would keep a stale registration.
  // frozen atom types (atype_freeze): d_frozen = per-atom flags (nullptr when nothing is frozen), n_free = atoms that are integrated
  std::vector<int> freeze_type; DBuf<int> d_frozen; bool any_frozen = false; double n_free = 0.0, inv_mass_sum_free = 0.0;
  DBuf<int> d_host_perm; DBuf<double> d_stage3; bool host_perm_on = false;      // msevb_set_host_order: caller's atom i = library atom d_host_perm[i]
  bool pin_host_arrays = false;      // opt-in (option pin_host_arrays): only safe when the caller's arrays live as long as the context uses them
  struct PinnedUser { const void* p; size_t bytes; int hits; bool registered; } pinned_user[8] = {};
  // tables
  std::vector<double> t_B6, t_B5, t_erfc, t_scale, t_ewE, t_tt, t_dtt, t_CB;
  double erfc_dx = 0, ewald_self = 0, kk[3][3];

  // ---- EVB host tables
  bool evb_set = false;
  int e_acid[MAX_MOLTYPES], e_base[MAX_MOLTYPES], e_conj_mol[MAX_MOLTYPES], e_proton_type[MAX_MOLTYPES], e_heavy_type[MAX_MOLTYPES];
  int e_conj_atype[MAX_TYPES];
  double e_ref[MAX_MOLTYPES], e_exch_atomic[MAX_TYPES], e_exch_proton[MAX_MOLTYPES][MAX_MOLTYPES];
  int n_da = 0, da_types[MAX_INTERACTION][3]; double da_par[MAX_INTERACTION][6];
  int n_pa = 0, pa_types[MAX_INTERACTION][2]; double pa_par[MAX_INTERACTION][5];
  int n_cp = 0, cp_types[MAX_INTERACTION][3], cp_func[MAX_INTERACTION]; double cp_par[MAX_INTERACTION][10];
  int hydronium = -1, max_chain = 3;

  // ---- device: static
  DBuf<int> d_atype, d_mol_type, d_mol_natom, d_mol_atoms, d_mol_of;
  DBuf<double> d_chg, d_mass;
  DBuf<PairParam> d_pair_par;
  DBuf<MolTypeDev> d_moltypes;
  DBuf<double2> d_ewald_tab;     // {erfc_table, ewaldscale_table} interleaved
  DBuf<double2> d_ewald_tay;     // {erfc_table, E = (ewaldscale - erfc) / (alpha r)}: the one row per pair k_pair_flat fetches (pair_flat.hip)
  DBuf<double> d_B6, d_B5, d_tt, d_dtt, d_CB;
  // ---- device: per step
  DBuf<double> d_xyz, d_vel, d_force, d_force_recip;
  DBuf<double> d_xyz_store, d_disp;       // Verlet displacement bookkeeping
  DBuf<double> d_maxdisp;
  // neighbour structure (sorted order)
  int ncell[3] = {1, 1, 1};
  DBuf<int> d_cell_of, d_perm, d_cell_keys_tmp, d_perm_tmp, d_cell_start, d_inv_perm, d_tile_max;
  DBuf<unsigned> d_nbr_count, d_nbr_start, d_nbr_split;      // split: entries of a half-list row that lie in the first 7 cells of the half neighbourhood (pair_flat.hip, two workgroups per cell)
  DBuf<int> d_nbr;
  bool attr_nbr_tiles[3] = {false, false, false};
  DBuf<char> d_sort_tmp; DBuf<int> d_sort_hist; bool cell_sort_own = true;      // cell sort of a list rebuild: own stable counting sort (three kernels) / hipcub radix sort (option cell_sort_own)
  size_t n_nbr = 0;
  unsigned nbr_cap = 0;          // row stride of the Verlet list (0: not sized yet)
  bool list_valid = false;
  // rolling pruned pair list (pair.hip): rows of the pairs within rc + prune_delta, rewritten by the pair loop itself whenever the two
  // largest displacements since the last pruning add up to prune_delta; d_prune_mode[0]: 1 = this step prunes, 2 = this step uses the rows
  bool prune = true; double prune_delta = 0.6 /* msevb_set_system sets min(0.75, (r_v - r_c) / 2) unless the option prune_delta was given */; bool prune_delta_user = false; int prune_lo = -1, prune_hi = -1;
  DBuf<int> d_nbr2, d_prune_mode; DBuf<unsigned> d_nbr2_count; DBuf<double4> d_posq_prune;
  // flat per-cell pair list of k_pair_flat (pair_flat.hip): the cell's pairs within rc + prune_delta, rows padded to 16 entries, one segment per wave of the
  // cell's workgroup; cell c's list starts at cell_start[c] * (nbr_cap + 96) + 3584 c, d_cell_trips[32 c + 16 h + w] = {offset, length in 64-entry trips} of the segment of wave w of workgroup h
  DBuf<int> d_flat; DBuf<uint2> d_cell_trips;
  int pair_flat_opt = 1;         // option pair_flat: 0 = k_pair_half on per-atom rows (round 2's kernel; still what force fields with a live SAPT term run)
  bool pair_self_mode = true; int pair_self_nb = 0, pair_epoch = 0; hipEvent_t ev_top2 = nullptr;      // option pair_self_mode: k_pair_flat decides itself whether its list still covers r_c (pair_flat.hip); pair_self_nb > 0: for the next launch, over that many per-block partial maxima
  bool prune_freeze = false;     // diagnostics (tools/time_stage.py): stage calls on unchanged positions keep the pair list of the first call instead of rewriting it every time
  bool pair_taylor = true;       // option pair_taylor (default on): one Ewald-table row per pair, the other one from the table's own derivatives (pair_flat.hip); 0 = both rows gathered
  bool attr_pair_flat[20] = {false, false, false, false, false, false, false, false, false, false, false, false, false, false, false, false, false, false, false, false};
  bool pair_pipe = true;         // option pair_pipe: 1 = two trips in flight per wave (128 registers), 0 = one trip at a time (80)
  int pair_split_opt = 0;        // option pair_split (measured slower, off): 1 = two workgroups per cell when the sub-tiles fit (2 x 12 waves per CU), 2 = the same with 2 x 8 waves and the pipelined loop, 0 = one workgroup per cell
  bool pair_split_now = false;
  int sub_tile_max = 0;          // largest sub-tile (own cell + 6 / + 7 forward cells) the last build saw
  int cell_atoms_max = 0;        // most atoms in ONE cell the last build saw (the pair kernels' row metadata holds PAIR_CELL_MAX)
  bool nbr_stats_copy_pending = false;      // in-kernel row statistics of a deferred rebuild still have to be downloaded
  bool nbr_check_pending = false;      // a deferred rebuild's row statistics have not been looked at yet
  int* h_nbr_stats = nullptr;          // pinned: {tile_max, pad, count_max, pad, count_sum lo, hi}
  int flag_verlet_list = 0;
  bool rebuild_zero_force_now = false, stats_zeroed_now = false;      // list rebuild inside a force call: k_gather_sorted zeroes the force array (and clears the flag) / has zeroed the row statistics
  bool step_fuse_refresh = true; int refresh_fused_nb = 0;      // option step_fuse_refresh: k_step_first_half does k_step_refresh's work on steps that keep their list (integrate.hip); > 0: blocks whose displacement partial sums await k_top2_final
  DBuf<double4> d_posq;        // sorted {x,y,z,q}
  DBuf<double> d_fsorted;      // [N][3] force accumulator of the half-list pair kernel in SORTED order (coalesced atomics), folded into d_force and re-zeroed by k_pair_fold
  DBuf<int> d_type_sorted, d_mol_sorted;
  // PME
  DBuf<double2> d_grid;        // complex K^3 (x fastest), in place FFT
  DBuf<double> d_Q;            // real copy of the spread grid (kept for the EVB delta path)
  DBuf<double> d_theta;        // real part of the convolved grid (theta_conv_Q)
  DBuf<double> d_green;        // G = IFFT(CB): real-space kernel of the reciprocal sum (theta = G * Q), K^3
  DBuf<double> d_w, d_dw, d_uscale, d_uscale_s;   // per atom: B6 weights [N][3][6], derivative weights, scaled coords
  DBuf<int> d_base, d_splindex;       // nearpt [N][3], B6 rows [N][3][6]
  DBuf<int2> d_rows_s;      // cell-sorted row seeds of the B-spline tables per (atom, axis), pme_row_seed (pme.hip)
  DBuf<int> d_base_s, d_brick_flag; DBuf<double> d_w_s;      // cell-sorted copies of nearpt / B6 weights for the brick spread; its overflow flag
  rocfft_plan plan_fwd = nullptr, plan_bwd = nullptr;
  // batched plans by batch size (= number of non-principal diabat grids): kept for the life of the context, because the diabat count moves
  // up and down during a run and creating a plan pair costs milliseconds (a step costs 0.6 ms)
  std::map<int, std::pair<rocfft_plan, rocfft_plan>> plans_b;
  bool spread_tiles = true;
  bool pme_weightless = true;       // tiled spread / gather recompute the B-spline weights from (uscale, nearpt) and the tables: no 288 B/atom weight store
  bool weights_needed_now = false;  // this call has another reader of d_w / d_dw (set by the MS-EVB call for its per-diabat-transform paths)
  int attr_gather_wl = 0, attr_gather_wl_b = 0;      // dynamic LDS the weight-free gather (plain / blending instantiation) has been allowed so far
  bool pme_side_weights = false; bool weights_side = false; hipEvent_t ev_weights = nullptr;      // MS-EVB fast path: stored weights computed on a side stream for the blend phase's gather (pme_side_weights)
  bool weights_valid = false;       // d_w / d_dw hold this step's weights (k_pme_prep ran, not k_pme_prep_light)
  int pme_gather_threads = 0;    // option: workgroup size of the weight-free gather alone (multiple of 64, <= 512; 0 = 256 while no cell holds more atoms than that, else pme_tile_threads)
  int pme_tile_threads = 512;    // workgroup size of the LDS-tile spread / gather kernels (option pme_tile_threads: 256 or 512)
  bool spread_bricks = false;    // option: owner-computes charge spreading on the matrix cores (k_pme_spread_bricks): every cell's workgroup writes ITS brick of the grid, no atomics, no zero-fill
                                 // (needs bricks of <= 8 points per axis, K >= 16).  Measured 63 us alone against 55 + 4 + 5 for the LDS-atomic tiles with fill and copy, 2-3 % slower in the overlapped step: off
  bool spread_mfma = false;      // charge spreading as a per-cell contraction on the matrix cores (v_mfma_f64_16x16x4_f64), option spread_mfma
  bool attr_spread_mfma = false;
  bool evb_region_aside = false;     // speculative step: the neighbourhood kernel (record + tree check) runs beside the descriptor chain, not in front of it
  bool evb_late_mol_join = true;     // MS-EVB fast path: the blend phase waits for the per-molecule kernel only in front of k_evb_blend, not at its start
  bool evb_publish_spin = true;      // fetch A's publish kernel waits for the side chains' marks on the device (no cross-stream event hops in front of the host's turn)
  bool evb_pair_first = false; hipEvent_t ev_pair_done = nullptr;      // schedule experiment (option evb_pair_first): diabat chains behind the pair loop
  bool gather_mfma = false;      // force interpolation with the x contraction on the matrix cores (k_pme_gather_mfma), option gather_mfma: measured, not the default
  bool has_sapt = false;     // some (type_i,type_j) entry has non-zero SAPT coefficients
  rocfft_execution_info fft_info = nullptr, fft_info_b = nullptr;      // single-grid plans (their work buffer is baked into the captured PME graph) / batched plans
  DBuf<char> d_fft_work, d_fft_work_b;
  int fft_batch = 1;
  bool fft_own = true;           // 64^3 grids: the library's own three-kernel LDS transform + convolution (fft_lds.hip) instead of rocFFT, option fft_own
  // reductions
  DBuf<double> d_partials;     // [n_slots][MAX_BLOCKS]
  DBuf<double> d_energies;     // [32]
  double* h_energies = nullptr;  // pinned
  int slot_count[32] = {0};      // pending per-slot partial counts (fused reduction)
  double E_recip = 0;

  // EVB state
  int D = 0;
  int plog[MAX_DIABATS][MAX_CHAIN][5];
  int parent[MAX_DIABATS];
  double H[MAX_DIABATS][MAX_DIABATS];
  double eigvec[MAX_DIABATS];
  int principal = 0, new_hydronium = -1;
  DBuf<double> d_Fdiag, d_Foff;     // [D][N][3]
  int evb_cap = 0;
  int rank = 0, n_ranks = 1;
  int comm_timeout_ms = 30000, comm_stall_ms = 0;      // options: wall-clock limit of the host's wait for the per-step all-gather / one-shot artificial stall in front of it (test)
  void* nccl_comm = nullptr;
  // per-context scratch that used to be process-wide (two contexts on different GPUs / host threads must not share it)
  // principal-diabat switches since the log was last read: {EVB force call number (1-based, counted over the context's life), old hydronium, new hydronium}
  // (1-based molecules) - the reference writes every hop at its own step (src/ms_evb.f90:331-335)
  std::vector<int> hop_log; long evb_calls = 0;
  // counters behind msevb_get_info: speculative MS-EVB calls, how many of them found the diabat tree changed (the call is repeated the long way), enumerations (DFS + descriptor build) done by the HOST
  long evb_spec_steps = 0, evb_spec_misses = 0, evb_host_dfs_calls = 0, evb_device_plans = 0;
  bool evb_device_plan = true;      // f2: a call that has to enumerate the diabat tree does so on the device (k_evb_plan); 0 = on the host (the same source, evb_plan.hpp: the checker)
  // step graph (StepRec above): option step_graph; step_sb is non-null while a force call is being captured / replayed (kernels then read their per-call numbers from it);
  // opt_epoch counts msevb_set_option / box / communicator changes (a captured graph bakes the options in)
  // OFF by default: on this runtime (ROCm 7.2) hipGraphLaunch enqueues its nodes one by one on the calling thread - 3.1 us per node against 3.3 us per stream launch
  // (profiles/r05_graph_probe.txt) - so the replay saves no host time, and the graph's branches cannot use the device-side marks (3 % of the step)
  bool top2_join_pending = false;      // k_top2_final ran on aux3 and the main stream did not wait for it: whoever publishes the displacement maxima does
  LinOps lin_tail; long long* lin_cnt = nullptr; long long* lin_marks = nullptr; bool lin_tail_defer = false;      // closing waits of a force call that went out as linear graphs: performed by the integrator's second kernel (MD loop: lin_tail_defer) or by a one-thread kernel at the end of the call
  int evb_step_graph = 0;      // option step_graph (off by default: on one GPU the one-thread kernels that order the chains cost more than the host time saved)
  const long long* step_sb = nullptr; long evb_graph_launches = 0, evb_graph_captures = 0, evb_graph_mismatches = 0;
  void* evb_work = nullptr;            // EvbWork of evb.hip (created by the first EVB force call, freed by evb_release)
  DBuf<double> d_comm_h, d_comm_e;     // staging of the Hamiltonian all-gather / the energy all-reduce (comm.hip)
  bool evb_device_solve_ranks = true;      // option: with more than one rank the Hamiltonian is solved on the device behind the all-gather and blend + all-reduce are pre-enqueued (0: host Jacobi, round 4's path)
  std::vector<double> virt_payload; bool virt_payload_dirty = false;      // virtual ranks: the R payloads a real all-gather would deliver (msevb_comm_set_virtual_payload)
  double* h_comm_e8 = nullptr; bool attr_evb_solve_g = false;      // pinned: the summed principal energy slots (device-side all-reduce)
  double* h_comm = nullptr; size_t h_comm_cap = 0; long long* h_comm_flag = nullptr; long long comm_seq = 0;      // pinned staging of the per-step all-gather (comm.hip)
  DBuf<int> d_bad;                     // 'force too large' flag of the integrator (integrate.hip)
  double* h_ke = nullptr; int h_ke_cap = 0;      // pinned kinetic-energy records of msevb_md_integrate_atomic
  bool attr_evb_solve = false;         // hipFuncSetAttribute done on this context's device

  PhaseTimer ph[PH_COUNT];
  bool timing = true;
  // phase events are SAMPLED: one force call in timing_stride records them (every record costs ~2 us of host time, 25 per step)
  bool timing_now = false; int timing_stride = 8; long timing_tick = 0;
  // NVT: Langevin leap-frog kicks (langevin_integrator, src/md_integration.f90:186-228) instead of the plain half kicks
  int stamps_on = -1; unsigned long long* h_stamps = nullptr; double stamp_sum[16] = {0}; long stamp_n = 0, stamp_step = 0;
  bool evb_recip_linear = true;      // diabats' reciprocal space through the linear response of theta to the charge differences (evb.hip): no transform per diabat
  bool evb_speculate = true;         // MS-EVB step without a host wait for the neighbourhood record while the diabat tree stays the same (evb.hip, SpecPlan); 0 = enumerate every step
  long topo_epoch = 0;               // bumped by everything that can change what a recorded diabat tree means (topology upload, box, options)
  bool evb_fft_patch = true;         // 64^3: the blended charge grid is formed inside the forward transform (fft_lds.hip, FftPatch); 0 = copy of Q_1 + k_evb_blend_patch in front of it
  bool evb_blend_aside = true;       // MS-EVB fast path: blend terms added from a side stream beside the transform and the gather (k_evb_blend_terms), pair accumulator folded by the gather; 0 = k_evb_blend behind the gather
  int evb_device_solve_method = 1;   // k_evb_solve: 1 = ground state by repeated squaring (<= 16 diabats; Jacobi decides when it does not converge), 0 = round-robin Jacobi always
  bool evb_device_solve = true;      // single rank, no element stores: H' assembly + round-robin Jacobi + blend coefficients in a kernel on the main stream, the blend phase behind it (no host turn on the critical path); 0 = host Jacobi
  bool nvt = false; double friction = 0.1 /* 1/ps, the reference's default */, temperature = 300.0; unsigned long long langevin_seed = 0, langevin_tick = 0;
};

// ---- cross-file entry points (all enqueue on c->stream) ----------------------------------------------
int nb_build(msevb_ctx* c);                        // neighbor.hip
int nb_build_begin(msevb_ctx* c);     // cells + sorted arrays
int nb_build_finish(msevb_ctx* c, bool defer);    // rows (one host synchronisation unless deferred)
int nb_check_overflow(msevb_ctx* c);  // after a deferred finish
int build_box_tables(msevb_ctx* c);    // tables.cpp: CB array + reciprocal lattice for the current box
extern "C" void pin_user_buffer(msevb_ctx* c, const void* p, size_t bytes);      // api.hip
int nb_prune_reset(msevb_ctx* c);     // the next pair loop prunes (list rebuilt / positions replaced)
int nb_settle(msevb_ctx* c);          // looks at statistics a failed call left behind
// GPU-side timestamps of a step (MSEVB_STAMPS=1): tiny kernels write the 100 MHz wall clock at points of interest of each stream; the
// averages, relative to stamp 0, are printed when the context is destroyed.  The profiler's per-call host overhead distorts exactly the
// thing this shows: WHEN the host gets the side chains out.
extern "C" void stamp(msevb_ctx* c, int idx, hipStream_t st);
extern "C" void stamps_report(msevb_ctx* c);
constexpr int MSEVB_RETRY = 1000;     // internal: repeat the force call (never leaves the library)
int nb_refresh_sorted(msevb_ctx* c);               // gather posq/type in sorted order
extern "C" int msevb_comm_allreduce_forces(msevb_ctx* c);   // comm.hip
void evb_release(msevb_ctx* c);        // evb.hip: per-context EVB scratch
extern "C" void msevb_comm_release(msevb_ctx* c);   // comm.hip: destroys the RCCL communicator, if any
int flush_one(msevb_ctx* c, int slot, hipStream_t st);      // api.hip
int nb_step_refresh(msevb_ctx* c, hipStream_t st = nullptr, bool with_top2 = true);
int nb_step_top2(msevb_ctx* c, int nb, hipStream_t st = nullptr, bool decide = true);      // k_top2_final alone, over nb per-block partial results (the integrator's first kernel left them)      // refresh + displacement check + zero forces (neighbor.hip)
int nb_check_displacements(msevb_ctx* c, int initialize, int* flag);
int pair_forces(msevb_ctx* c, int slot_elec, int slot_vdw);              // pair.hip
bool pair_flat_usable(const msevb_ctx* c);                                // pair_flat.hip
int pair_flat_launch(msevb_ctx* c, int c_lo, int c_hi, int slot_elec, int slot_vdw, int* nblocks);
int molecule_forces(msevb_ctx* c, int slot0, bool do_nb, bool do_bonded, hipStream_t st = nullptr);  // molecule.hip
int pme_forces(msevb_ctx* c, int slot_recip, bool join_aux = false);
int pme_front(msevb_ctx* c, int slot_recip, bool join_aux);
int pme_tail(msevb_ctx* c, bool join_aux);                           // pme.hip
int pme_front_finish(msevb_ctx* c);                                  // the solve of a split front end (pme_split_now)
int pme_setup(msevb_ctx* c);
int pme_build_green(msevb_ctx* c);   // after the CB array changed (box change)
void pme_teardown(msevb_ctx* c);
int reduce_partials(msevb_ctx* c, int slot, int nblocks);                // deferred: one fused launch in flush_reductions
int flush_reductions(msevb_ctx* c);
int flush_range(msevb_ctx* c, int s0, int s1, hipStream_t st);
int fetch_energies(msevb_ctx* c);
int integrate_half_kick_drift(msevb_ctx* c, double dt);                  // integrate.hip
int integrate_half_kick(msevb_ctx* c, double dt);
int integrate_com_shift(msevb_ctx* c);
int integrate_remove_momentum(msevb_ctx* c);
int integrate_kinetic(msevb_ctx* c, double* ke);
int evb_force_energy(msevb_ctx* c, double* epot);                        // evb.hip
int build_tables(msevb_ctx* c);                                          // tables.cpp
void recip_lattice(const double box[3], double vol, double kk[3][3]);
double volume32(const double box[3]);

// defined inside api.hip's extern "C" block
extern "C" int total_force_enqueue(msevb_ctx* c);
extern "C" int total_force_collect(msevb_ctx* c, double* en);
extern "C" int upload_topology(msevb_ctx* c);
extern "C" int upload_atoms3(msevb_ctx* c, double* dst_dev, const double* src_host);      // api.hip: (3,N) arrays in the caller's atom order (msevb_set_host_order)
extern "C" int download_atoms3(msevb_ctx* c, double* dst_host, const double* src_dev);
int pme_convolve(msevb_ctx* c, double2* grids, int batch, double* part, int part_stride, int* nblocks);
int pme_prepare_atoms(msevb_ctx* c, bool with_indices);
bool pme_weightless_now(const msevb_ctx* c);
bool pme_tiles_ok(const msevb_ctx* c);
bool pme_gather_can_fold(const msevb_ctx* c);      // the next pme_gather call will be the weight-free tiled kernel (which can fold the pair loop's accumulator in)
int pme_side_weights(msevb_ctx* c, hipStream_t st, hipEvent_t after);
bool fft_lds_usable(const msevb_ctx* c);                              // fft_lds.hip
int fft_lds_setup(msevb_ctx* c);
// B-spline stencil of one special atom in one state (MS-EVB reciprocal space by linear response, evb.hip): nearest grid points, charge, tap weights
struct Stencil { int base[3]; int grid, slot, idx; double q; double w[18], dw[18]; };
// charge-grid patch applied by the forward transform itself (64^3, fft_lds.hip): plane += sum over the stencils of coefficient[grid] q S.  The blended
// charge grid of the Hellmann-Feynman force then needs neither a copy of Q_1 nor a patch kernel in front of its transform.
struct FftPatch {
  int nst; const Stencil* st;
  const double* dc;                       // coefficients in device memory: dc[1 + grid] (written by the solve body) ...
  double c[MAX_DIABATS];                  // ... or as kernel arguments (dc == nullptr)
  const long long* wait_mark; long long wait_seq; double* fault;      // the coefficients' kernel runs on another stream: start on its mark
  const long long* wait_sb;                                           // (a captured launch: the mark's number is wait_sb[SB_A])
};
bool fft_lds_can_patch(const msevb_ctx* c);      // the 64^3 transform (its forward kernel applies an FftPatch)
int fft_lds_solve(msevb_ctx* c, const double* Q, double2* work, double* theta, double* part, int* nblocks, hipStream_t st, const FftPatch* patch = nullptr);
int pme_gather(msevb_ctx* c, const double2* grid, const double* chg, double* F_recip, double* F_add, bool atomic = false, const double* theta = nullptr, double* Fs_fold = nullptr, bool* folded = nullptr);      // Fs_fold: the pair loop's sorted-order accumulator joins the atomic add and is re-zeroed (weight-free tiled kernel only; *folded says whether)

constexpr int MAX_BLOCKS = 65536;
enum Slot { S_ELEC_PAIR = 0, S_VDW_PAIR, S_ELEC_INTRA, S_VDW_INTRA, S_BOND, S_ANGLE, S_DIH, S_RECIP, S_KE, S_PX, S_PY, S_PZ, S_VX, S_VY, S_VZ /* sum of velocities: KE after the momentum removal */, S_MAXD1, S_MAXD2, S_PRD1, S_PRD2 /* two largest displacements since the pair list was last pruned */, S_EVB0,
            S_TIMEOUT = 29,    // sticky: a device-side wait (chain marks, blend mark) ran into its wall-clock limit -> MSEVB_ERR_TIMEOUT; a slot of its own, so that a coordinate code
                               // written to S_FAULT afterwards cannot hide it
            S_BRICK = 30,      // sticky: a brick of the owner-computes spread met more atoms than it can take (nb_check_overflow switches to the tile kernel and has the force call repeated)
            S_FAULT = 31,      // sticky: a kernel met a coordinate outside what its index arithmetic covers (non-finite, or more than one box away)
            N_SLOTS = 32 };

inline void phase_begin(msevb_ctx* c, int p, hipStream_t st = nullptr) {
  if (!(c->timing_now || (c->timing && p == PH_VERLET))) return;      // list rebuilds are rare: always timed (and counted)
  if (!st) st = c->stream;
  PhaseTimer& t = c->ph[p];
  if (t.used + 2 > 16384) return;   // pool exhausted: stop sampling until the next harvest
  if (t.used + 2 > t.ev.size()) { hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b); t.ev.push_back(a); t.ev.push_back(b); }
  (void)hipEventRecord(t.ev[t.used], st);
  t.used += 1;
}
inline void phase_end(msevb_ctx* c, int p, hipStream_t st = nullptr) {
  if (!(c->timing_now || (c->timing && p == PH_VERLET))) return;
  if (!st) st = c->stream;
  PhaseTimer& t = c->ph[p];
  if ((t.used & 1) == 0) return;
  (void)hipEventRecord(t.ev[t.used], st);
  t.used += 1;
}
// [lo, hi) of n items owned by this rank while the principal diabat is sharded, everything otherwise
inline void shard_range(const msevb_ctx* c, int n, int& lo, int& hi) {
  if (!c->shard_now || c->n_ranks < 2) { lo = 0; hi = n; return; }
  lo = (int)((long)n * c->rank / c->n_ranks); hi = (int)((long)n * (c->rank + 1) / c->n_ranks);
}
inline hipError_t aux_fork(msevb_ctx* c) {     // both side streams continue from the current end of the main stream
  hipError_t e = hipEventRecord(c->ev_fork, c->stream);
  if (e == hipSuccess) e = hipStreamWaitEvent(c->aux, c->ev_fork, 0);
  if (e == hipSuccess) e = hipStreamWaitEvent(c->aux2, c->ev_fork, 0);
  if (e == hipSuccess) e = hipStreamWaitEvent(c->aux3, c->ev_fork, 0);
  return e;
}
inline hipError_t aux_join(msevb_ctx* c) {     // the main stream waits for what both side streams hold so far
  hipError_t e = hipEventRecord(c->ev_join, c->aux);
  if (e == hipSuccess) e = hipStreamWaitEvent(c->stream, c->ev_join, 0);
  if (e == hipSuccess) e = hipEventRecord(c->ev_join2, c->aux2);
  if (e == hipSuccess) e = hipStreamWaitEvent(c->stream, c->ev_join2, 0);
  return e;
}
inline void phase_harvest(msevb_ctx* c) {
  for (int p = 0; p < PH_COUNT; ++p) {
    PhaseTimer& t = c->ph[p];
    for (size_t k = 0; k + 1 < t.used; k += 2) {
      float v = 0;
      if (hipEventSynchronize(t.ev[k + 1]) == hipSuccess && hipEventElapsedTime(&v, t.ev[k], t.ev[k + 1]) == hipSuccess) { t.ms += v; t.calls++; }
    }
    t.used = 0;
  }
}
