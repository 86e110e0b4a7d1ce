import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from reactive_pb_nn_md_amd import system, api
np.set_printoptions(linewidth=200, precision=6, suppress=True)
g = np.load(os.path.join(ROOT, "tests", "golden", "evb_n6_d2_s1.npz"))
spec = system.from_golden(g)
h = api.MsEvbHip(spec)
e, F = h.ms_evb_calculate_total_force_energy()
out = np.zeros(4096)
nu = h.L.msevb_get_evb_debug(h.ctx, out.ctypes.data_as(C.POINTER(C.c_double)), 4096)
print("nu", nu)
print("components [pairs_el, pairs_vdw, intra_nb_D, intra_nb_A, bonded_D, bonded_A, rep]")
print(out[:8*nu].reshape(nu, 8))
print("E_env", out[8*nu:9*nu], "E_spec", out[9*nu:10*nu], "E_grid", out[10*nu:10*nu+1], "E_recip1", h.pme_reciprocal_space_energy_force())
