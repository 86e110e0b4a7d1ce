import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from reactive_pb_nn_md_amd import api, system
import test_gpu_parity as T
g = T.golden("evb_n6_d8_s1"); spec = system.from_golden(g)
h = api.MsEvbHip(spec); e1, F1 = h.ms_evb_calculate_total_force_energy(); st1 = h.evb_state(); h.close()
np.set_printoptions(linewidth=250, precision=3, suppress=True)
print("single parent", st1["coupling_matrix"]); print(np.triu(st1["H"], 1)[:, :][np.triu(st1["H"],1) != 0], np.argwhere(np.triu(st1["H"],1) != 0).tolist())
res = T._virtual_allgather_run(spec, 2, 1)
for r, (e, F, st, ns) in enumerate(res):
    print("rank", r, "parent", st["coupling_matrix"]); print(np.triu(st["H"], 1)[np.triu(st["H"],1) != 0], np.argwhere(np.triu(st["H"],1) != 0).tolist()); print("eig", st["eigvec"])
print("eig1", st1["eigvec"])
