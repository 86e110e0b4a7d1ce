#!/usr/bin/env python3
"""Cost of the host-array boundary: one EVB force call per 'step' with positions uploaded and forces downloaded through the C ABI
(what a Fortran host that keeps its own integrator pays), against the same call with everything left on the device."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from reactive_pb_nn_md_amd import api, forcefield, synth, system
s = synth.make_system(32, n_diabat=8, seed=0)
p = forcefield.build_params(forcefield.ch3so3h_water_forcefield(), moltype_order=["h2o", "so3h"])
spec = system.from_synth(s, p, rc=9.0, rv=10.5, alpha=0.3, pme_grid=64)
h = api.MsEvbHip(spec)
x = spec.xyz.copy()
h.ms_evb_calculate_total_force_energy()
import ctypes as C
F = np.zeros((spec.n_atom, 3)); e = C.c_double()
dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
for mode in ("device-resident", "host arrays", "host arrays, page-locked by the library"):
    n = 200
    if mode.endswith("library"):
        h.set_option("pin_host_arrays", 1)
    h.synchronize(); t0 = time.perf_counter()
    for k in range(n):
        if mode != "device-resident":      # the same two caller arrays every call, as a Fortran host has them
            h._chk(h.L.msevb_set_positions(h.ctx, dp(x)))
            h._chk(h.L.msevb_ms_evb_calculate_total_force_energy(h.ctx, C.byref(e), dp(F)))
        else:
            h.ms_evb_calculate_total_force_energy(want_force=False)
    h.synchronize(); dt = time.perf_counter() - t0
    print("%-42s %.3f ms per force call  (%.0f calls/s)" % (mode, dt / n * 1e3, n / dt), flush=True)
h.close()
