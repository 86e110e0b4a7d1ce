// libmsevb_dfti.so - the reference's ONE true C-ABI edge: the six Intel MKL DFTI symbols its Fortran objects import
//   dfti_create_descriptor_highd_    src/mkl_modules/mkl_module_src/mkl_dfti.f90:334-355   (call sites src/initialize_routines.f90:202, src/ms_evb.f90:1996)
//   dfti_commit_descriptor_external_ :245-255                                               (src/initialize_routines.f90:204-205)
//   dfti_set_value_intval_           :542-549   DFTI_NUMBER_OF_USER_THREADS                 (src/ms_evb.f90:1997)
//   dfti_compute_forward_z_          :681-688                                               (src/pme.f90:85, src/ms_evb.f90:2034)
//   dfti_compute_backward_z_         :741-747                                               (src/pme.f90:113, src/ms_evb.f90:2042)
//   dfti_free_descriptor_external_   (src/ms_evb.f90:2092-2093)
// served by rocFFT on the MI355X: 1..3-D double-complex, in place, unscaled both ways, exactly what every descriptor of the reference
// asks for.  With it the unchanged reference links WITHOUT MKL; in a GPU-backed build the force path never reaches these entry points
// (libmsevb_hip.so keeps its grids on the device), they only serve code that still calls DFTI itself.
// The Fortran dummy `type(DFTI_DESCRIPTOR), POINTER` arrives as the address of a pointer descriptor whose first word is the base
// address; the reference never dereferences it (the type's only component is PRIVATE), so that word carries our handle.
// The compute entry points are thread-safe per descriptor (a mutex), as the reference calls them from an OpenMP region.
#include <hip/hip_runtime.h>
#include <rocfft/rocfft.h>

#include <mutex>
#include <new>

namespace {
struct Handle {
  int dim = 0; size_t len[3] = {1, 1, 1}, n = 0;
  rocfft_plan fwd = nullptr, bwd = nullptr; rocfft_execution_info info = nullptr;
  void* d_data = nullptr; void* d_work = nullptr; hipStream_t st = nullptr; bool committed = false;
  std::mutex mu;
};
struct FPtr { void* base_addr; };
std::once_flag g_setup;
enum { DFTI_COMPLEX_ = 32, DFTI_DOUBLE_ = 36, DFTI_UNIMPLEMENTED_ = 6, DFTI_MEMORY_ERROR_ = 1, DFTI_INVALID_CONFIGURATION_ = 2, DFTI_BAD_DESCRIPTOR_ = 5 };

int compute(Handle* h, void* x, bool forward) {
  if (!h || !h->committed) return DFTI_BAD_DESCRIPTOR_;
  std::lock_guard<std::mutex> lk(h->mu);
  const size_t bytes = h->n * 16;
  if (hipMemcpyAsync(h->d_data, x, bytes, hipMemcpyHostToDevice, h->st) != hipSuccess) return DFTI_MEMORY_ERROR_;
  void* buf[1] = {h->d_data};
  if (rocfft_execute(forward ? h->fwd : h->bwd, buf, nullptr, h->info) != rocfft_status_success) return DFTI_INVALID_CONFIGURATION_;
  if (hipMemcpyAsync(x, h->d_data, bytes, hipMemcpyDeviceToHost, h->st) != hipSuccess) return DFTI_MEMORY_ERROR_;
  return hipStreamSynchronize(h->st) == hipSuccess ? 0 : DFTI_MEMORY_ERROR_;
}
void destroy(Handle* h) {
  if (!h) return;
  if (h->fwd) rocfft_plan_destroy(h->fwd);
  if (h->bwd) rocfft_plan_destroy(h->bwd);
  if (h->info) rocfft_execution_info_destroy(h->info);
  if (h->d_data) (void)hipFree(h->d_data);
  if (h->d_work) (void)hipFree(h->d_work);
  if (h->st) (void)hipStreamDestroy(h->st);
  delete h;
}
}  // namespace

extern "C" {

int dfti_create_descriptor_highd_(FPtr* d, const int* precision, const int* domain, const int* dim, const int* length) {
  if (!d || !precision || !domain || !dim || !length) return DFTI_BAD_DESCRIPTOR_;
  if (*precision != DFTI_DOUBLE_ || *domain != DFTI_COMPLEX_ || *dim < 1 || *dim > 3) return DFTI_UNIMPLEMENTED_;
  Handle* h = new (std::nothrow) Handle();
  if (!h) return DFTI_MEMORY_ERROR_;
  h->dim = *dim; h->n = 1;
  // rocFFT takes the fastest-varying length first, exactly the Fortran extent order X(n1,n2,n3)
  for (int i = 0; i < *dim; ++i) { if (length[i] < 1) { delete h; return DFTI_INVALID_CONFIGURATION_; } h->len[i] = (size_t)length[i]; h->n *= h->len[i]; }
  d->base_addr = h;
  return 0;
}
int dfti_set_value_intval_(FPtr* d, const int* /*config_param*/, const int* /*value*/) { return d && d->base_addr ? 0 : DFTI_BAD_DESCRIPTOR_; }      // thread counts have no meaning here
int dfti_commit_descriptor_external_(FPtr* d) {
  Handle* h = d ? (Handle*)d->base_addr : nullptr;
  if (!h) return DFTI_BAD_DESCRIPTOR_;
  if (h->committed) return 0;
  std::call_once(g_setup, [] { (void)rocfft_setup(); });
  if (hipStreamCreateWithFlags(&h->st, hipStreamNonBlocking) != hipSuccess) return DFTI_MEMORY_ERROR_;
  if (hipMalloc(&h->d_data, h->n * 16) != hipSuccess) return DFTI_MEMORY_ERROR_;
  if (rocfft_plan_create(&h->fwd, rocfft_placement_inplace, rocfft_transform_type_complex_forward, rocfft_precision_double, (size_t)h->dim, h->len, 1, nullptr) != rocfft_status_success ||
      rocfft_plan_create(&h->bwd, rocfft_placement_inplace, rocfft_transform_type_complex_inverse, rocfft_precision_double, (size_t)h->dim, h->len, 1, nullptr) != rocfft_status_success ||
      rocfft_execution_info_create(&h->info) != rocfft_status_success)
    return DFTI_INVALID_CONFIGURATION_;
  size_t wf = 0, wb = 0;
  (void)rocfft_plan_get_work_buffer_size(h->fwd, &wf); (void)rocfft_plan_get_work_buffer_size(h->bwd, &wb);
  if (wb > wf) wf = wb;
  if (wf) {
    if (hipMalloc(&h->d_work, wf) != hipSuccess) return DFTI_MEMORY_ERROR_;
    if (rocfft_execution_info_set_work_buffer(h->info, h->d_work, wf) != rocfft_status_success) return DFTI_INVALID_CONFIGURATION_;
  }
  if (rocfft_execution_info_set_stream(h->info, h->st) != rocfft_status_success) return DFTI_INVALID_CONFIGURATION_;
  h->committed = true;
  return 0;
}
int dfti_compute_forward_z_(FPtr* d, void* x) { return compute(d ? (Handle*)d->base_addr : nullptr, x, true); }
int dfti_compute_backward_z_(FPtr* d, void* x) { return compute(d ? (Handle*)d->base_addr : nullptr, x, false); }
int dfti_free_descriptor_external_(FPtr* d) {
  if (!d) return DFTI_BAD_DESCRIPTOR_;
  destroy((Handle*)d->base_addr);
  d->base_addr = nullptr;
  return 0;
}

}  // extern "C"
