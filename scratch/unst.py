import sys, time, numpy as np
sys.path.insert(0, ".")
import dflo_amd
from dflo_amd import gmsh, problems
verts, quads, bed, side = gmsh.unstructured_quads(int(sys.argv[3]), Lx=3.0, Ly=3.0, seed=1)
bid = np.array([2, 3, 2, 1], dtype=np.int32)[side]
deg, flux = int(sys.argv[1]), sys.argv[2]
mesh = dflo_amd.Mesh.from_quads(verts, quads, bed, bid, deg)
prm = dflo_amd.Parameters(flux=flux, cfl=0.5, final_time=1e9, boundary={1: "inflow", 2: "slip", 3: "outflow"})
claw = dflo_amd.ConservationLaw(mesh, prm)
cell, face, b, xy = claw.boundary_faces()
bv = np.stack(problems.forward_step_inflow(xy[..., 0], xy[..., 1]), axis=-1)
claw.set_boundary_values(0, bv); claw.set_boundary_values(1, bv)
xyc = mesh.support_points()
u0 = mesh.interpolate(problems.forward_step_inflow).reshape(mesh.n_cells, 4, -1) * (1.0 + 0.1 * np.exp(-20.0 * ((xyc[..., 0] - 1.5) ** 2 + (xyc[..., 1] - 1.5) ** 2)))[:, None, :]
claw.set_initial_condition(u0.reshape(-1))
claw.advance(30); claw.stage_timing(True)
t0 = time.perf_counter(); claw.advance(150); sec = time.perf_counter() - t0
ms, n = claw.stage_timing(False)
nrk = 2 if deg == 1 else 3
print("Q%d %s cells %d: %.0f MDoF/s, stage kernel %.1f us, frac %.3f" % (deg, flux, mesh.n_cells, mesh.n_cells * mesh.ndof * nrk * 150 / sec / 1e6, ms * 1e3, mesh.n_cells * mesh.ndof * 24 / (ms * 1e-3) / 8e12))
