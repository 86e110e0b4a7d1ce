/*
 * TEST INFRASTRUCTURE ONLY (oracle/): never linked into the product library.
 *
 * Calling-convention adapter between the reference's Fortran DFTI interface
 * (reference: src/mkl_modules/mkl_module_src/mkl_dfti.f90:245-255, 334-355,
 * 542-549, 681-688, 741-747) and the REAL Intel MKL DFTI C entry points that
 * this image ships inside PyTorch's libtorch_cpu.so (DftiCreateDescriptor_d_md,
 * DftiCommitDescriptor, DftiSetValue, DftiComputeForward, DftiComputeBackward,
 * DftiFreeDescriptor).  No FFT is implemented here: every transform the
 * reference asks for is executed by MKL itself.  The six symbols below are the
 * names flang emits for the reference's `external` interface functions; MKL's
 * own Fortran wrapper layer (libmkl_intel_lp64) is the only piece of MKL that
 * libtorch_cpu.so does not carry, and this file is that thin layer.
 *
 * The Fortran dummy `TYPE(DFTI_DESCRIPTOR), POINTER :: desc` arrives as the
 * address of a Fortran pointer descriptor whose first word is the base address
 * (ISO_Fortran_binding CFI_cdesc_t layout, which flang follows).  The reference
 * never dereferences the descriptor (its only component is PRIVATE), so the MKL
 * handle is stored directly in that first word.
 */
#include <stddef.h>

typedef void* DFTI_HANDLE;
/* MKL C API (mkl_dfti.h), declared by hand because the image has no MKL headers */
extern long DftiCreateDescriptor_d_md(DFTI_HANDLE*, int domain, long dim, long* lengths);
extern long DftiCommitDescriptor(DFTI_HANDLE);
extern long DftiSetValue(DFTI_HANDLE, int param, ...);
extern long DftiComputeForward(DFTI_HANDLE, void*, ...);
extern long DftiComputeBackward(DFTI_HANDLE, void*, ...);
extern long DftiFreeDescriptor(DFTI_HANDLE*);

typedef struct { void* base_addr; } f_ptr_desc; /* leading member of the Fortran pointer descriptor */

enum { DFTI_COMPLEX_ = 32, DFTI_DOUBLE_ = 36 };

int dfti_create_descriptor_highd_(f_ptr_desc* d, int* precision, int* domain, int* dim, int* length) {
  long len[8];
  if (*precision != DFTI_DOUBLE_ || *dim > 8) return -1;
  /* the Fortran array X(n1,n2,n3) is the row-major array [n3][n2][n1] of MKL's C interface */
  for (int i = 0; i < *dim; ++i) len[i] = length[*dim - 1 - i];
  DFTI_HANDLE h = NULL;
  long st = DftiCreateDescriptor_d_md(&h, *domain, (long)*dim, len);
  d->base_addr = h;
  return (int)st;
}
int dfti_commit_descriptor_external_(f_ptr_desc* d) { return (int)DftiCommitDescriptor(d->base_addr); }
int dfti_set_value_intval_(f_ptr_desc* d, int* opt, int* val) {
  return (int)DftiSetValue(d->base_addr, *opt, (long)*val);
}
int dfti_compute_forward_z_(f_ptr_desc* d, void* x) { return (int)DftiComputeForward(d->base_addr, x); }
int dfti_compute_backward_z_(f_ptr_desc* d, void* x) { return (int)DftiComputeBackward(d->base_addr, x); }
int dfti_free_descriptor_external_(f_ptr_desc* d) {
  DFTI_HANDLE h = d->base_addr;
  long st = DftiFreeDescriptor(&h);
  d->base_addr = NULL;
  return (int)st;
}
