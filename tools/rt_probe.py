"""(probe) The engine under the HIP runtime of /opt/rocm (argument rocm: dflo_amd first) and under the one PyTorch bundles (torch: torch first, as bench.py).
usage: python tools/rt_probe.py rocm|torch"""
import sys, time
sys.path.insert(0, "/root/repo")
if sys.argv[1] == "torch":
    import torch
    torch.zeros(1, device="cuda")
import numpy as np
import dflo_amd
from dflo_amd import problems
mesh = dflo_amd.Mesh.cartesian(1024, 1024, -5.0, -5.0, 10.0 / 1024, [-1, -1, -1, -1], 2)
prm = dflo_amd.Parameters(flux="hllc", cfl=0.8)
e = dflo_amd.ConservationLaw(mesh, prm)
e.set_initial_condition(mesh.interpolate(problems.isentropic_vortex))
e.advance(300)
for k in range(3):
    t0 = time.perf_counter()
    e.advance(400)
    dt = time.perf_counter() - t0
    print(sys.argv[1], "%.0f MDoF/s" % (mesh.n_cells * 36 * 400 / dt / 1e6), flush=True)
e.close()
