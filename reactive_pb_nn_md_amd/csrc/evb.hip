// placeholder, replaced below in this round
#include "ctx.hpp"
int evb_force_energy(msevb_ctx* c, double* epot) { (void)epot; FAIL(MSEVB_ERR_STATE, "EVB path not built"); }
extern "C" {
int msevb_ms_evb_calculate_total_force_energy(msevb_ctx* c, double* e, double* f) { (void)e; (void)f; FAIL(MSEVB_ERR_STATE, "EVB path not built"); }
int msevb_get_evb_state(msevb_ctx* c, int*, int*, int*, double*, double*, int*, int*) { FAIL(MSEVB_ERR_STATE, "EVB path not built"); }
int msevb_get_evb_element_force(msevb_ctx* c, int, int, double*) { FAIL(MSEVB_ERR_STATE, "EVB path not built"); }
int msevb_comm_unique_id(void*) { return MSEVB_ERR_COMM; }
int msevb_comm_init(msevb_ctx* c, const void*, int, int) { FAIL(MSEVB_ERR_COMM, "not built"); }
int msevb_comm_set_virtual(msevb_ctx* c, int r, int n) { c->rank = r; c->n_ranks = n; return MSEVB_OK; }
}
