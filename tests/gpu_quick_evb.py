"""quick GPU EVB sanity script (not a pytest)"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from reactive_pb_nn_md_amd import system, api
np.set_printoptions(linewidth=200, precision=6, suppress=True)
for fx in ("evb_n6_d1_s1", "evb_n6_d2_s1", "evb_n6_d4_s2", "evb_n6_d8_s1", "evb_n8_d8_s1"):
    g = np.load(os.path.join(ROOT, "tests", "golden", fx + ".npz"))
    spec = system.from_golden(g)
    h = api.MsEvbHip(spec)
    h.set_option("evb_element_stores", 1)
    e, F = h.ms_evb_calculate_total_force_energy()
    st = h.evb_state()
    D = int(g["n_diabat"][0])
    print(fx, "D", st["D"], D, flush=True)
    Href = g["evb_hamiltonian"]
    print("  max|H-Href|", np.abs(st["H"] - Href).max(), "rel", np.abs(st["H"] - Href).max() / np.abs(Href).max())
    if np.abs(st["H"] - Href).max() > 1e-6:
        print(st["H"] - Href)
    print("  plog eq", np.array_equal(st["proton_log"][:, :3, :], g["evb_diabat_proton_log"]), "parent eq", np.array_equal(st["coupling_matrix"], g["evb_diabat_coupling_matrix"]))
    print("  E_adiab", e, g["E_adiabatic"][0], e - g["E_adiabatic"][0], "F err", np.abs(F - g["F_adiabatic"]).max(), "principal", st["principle_diabat"], int(g["principle_diabat"][0]))
    if "evb_forces_store" in g:
        lk = g["evb_forces_lookup_index"]; sto = g["evb_forces_store"]
        worst = 0
        for i in range(D):
            for j in range(i, D):
                if lk[i, j] > 0:
                    err = np.abs(h.evb_element_force(i + 1, j + 1) - sto[lk[i, j] - 1]).max()
                    worst = max(worst, err)
                    if err > 1e-8: print("    element", i + 1, j + 1, "force err", err)
        print("  worst element force err", worst)
    t = time.time()
    for _ in range(10): h.ms_evb_calculate_total_force_energy(want_force=False)
    print("  %.3f ms / EVB call" % ((time.time() - t) / 10 * 1e3))
    h.close()
