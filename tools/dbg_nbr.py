import sys
sys.path.insert(0, "/root/repo")
import numpy as np
from reactive_pb_nn_md_amd import api, forcefield, synth, system
s = synth.make_system(14, n_diabat=2, seed=3)
p = forcefield.build_params(forcefield.ch3so3h_water_forcefield(), moltype_order=["h2o", "so3h"])
spec = system.from_synth(s, p, rc=9.0, rv=10.5, pme_grid=32)
h = api.MsEvbHip(spec)
h.construct_verlet_list()
vp1, nl1 = h.get_verlet_list()
h.set_option("nbr_tiles", 0)
h.construct_verlet_list()
vp0, nl0 = h.get_verlet_list()
print("N", spec.n_atom, "pairs", len(nl1), len(nl0), "vp equal", np.array_equal(vp1, vp0))
c1, c0 = np.diff(vp1), np.diff(vp0)
bad = np.nonzero(c1 != c0)[0]
print("atoms with different half-row length:", len(bad), bad[:10], c1[bad[:10]], c0[bad[:10]])
x = spec.xyz
L = spec.box
for i in bad[:5]:
    r1 = set(nl1[vp1[i]-1:vp1[i+1]-1]); r0 = set(nl0[vp0[i]-1:vp0[i+1]-1])
    miss = sorted(r0 - r1); extra = sorted(r1 - r0)
    print("atom", i, "pos", x[i], "missing", miss[:5], "extra", extra[:5])
    for j in (miss + extra)[:3]:
        d = x[i] - x[j-1]; d -= L * np.round(d / L)
        print("   j", j, "pos", x[j-1], "r", np.sqrt((d*d).sum()))
