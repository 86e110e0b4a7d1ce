"""quick GPU sanity script (not a pytest): python tests/gpu_quick.py"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from reactive_pb_nn_md_amd import system, api

def cmp(name, a, b):
    print("%-14s max|ref| %.3e  max abs err %.3e" % (name, np.abs(b).max(), np.abs(a - b).max()), flush=True)

for fx in ("std_n8_s0", "std_n6_s3"):
    g = np.load(os.path.join(ROOT, "tests", "golden", fx + ".npz"))
    spec = system.from_golden(g)
    t = time.time()
    h = api.MsEvbHip(spec)
    print(fx, "init %.2fs" % (time.time() - t), flush=True)
    for nm, key in (("B6_spline",) * 2, ("CB",) * 2, ("erfc_table",) * 2):
        tb = h.table(nm)
        print("  table", nm, "sample max rel err", np.nanmax(np.abs(tb.ravel()[g[key + "_sample_idx"]] - g[key + "_sample"]) / np.maximum(np.abs(g[key + "_sample"]), 1e-300)))
    h.construct_verlet_list()
    vp, nl = h.get_verlet_list()
    print("  verlet", len(nl), int(g["neighbor_count"]), np.array_equal(vp, g["verlet_point"]))
    h.zero_forces(); e = h.real_space_energy_force(); F = h.get_forces()
    cmp("F_real", F, g["F_pair"] + g["F_intra_nb"])
    print("   E", e, g["E_elec_pair"][0], g["E_vdw_pair"][0], g["E_elec_intra_nb"][0], g["E_vdw_intra_nb"][0])
    h.zero_forces(); er = h.pme_reciprocal_space_energy_force(); F = h.get_forces()
    cmp("F_recip", F, g["F_recip"]); print("   E_recip", er, g["E_recip"][0])
    Q, th, u, fr = h.pme_state()
    cmp("Q", Q, g["Q_grid"]); cmp("theta", th, g["theta_conv_Q"]); print("   u bit-equal", np.array_equal(u, g["xyz_scale"]))
    h.zero_forces(); eb = h.intra_molecular_energy_force(); F = h.get_forces()
    cmp("F_bonded", F, g["F_bonded"]); print("   E", eb, g["E_bond"][0], g["E_angle"][0], g["E_dihedral"][0])
    en, F = h.calculate_total_force_energy()
    cmp("F_total", F, g["F_total"]); print("   E_total", en["potential"], g["E_total"][0], en["potential"] - g["E_total"][0])
    t = time.time()
    for _ in range(20):
        en, _f = h.calculate_total_force_energy(want_force=False)
    print("  %.3f ms / force call" % ((time.time() - t) / 20 * 1e3))
    h.close()
