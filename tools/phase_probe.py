"""Developer probe: where a stage-kernel workgroup's time goes.  Needs a library built with tools/variants/phase_timing.patch
applied and -DDFLO_PHASE_TIMING (python tools/build_variant.py phase -DDFLO_PHASE_TIMING -> scratch/variants/phase.so).
usage: python tools/phase_probe.py c5 [k]  |  python tools/phase_probe.py sq <degree> <flux> <nx> [pos]
Prints the mean s_memtime ticks (shader clock) every wave of a workgroup spends in each phase of the LAST stage launch."""
import ctypes as C, os, shutil, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
real = os.path.join(ROOT, "dflo_amd", "libdflo_hip.so")
probe = os.path.join(ROOT, "scratch", "variants", "phase.so")
shutil.copy(real, "/tmp/base_keep.so")
shutil.copy(probe, real)
try:
    import dflo_amd
    from dflo_amd import problems, _lib, gmsh
    if sys.argv[1] == "c5":
        k = int(sys.argv[2]) if len(sys.argv) > 2 else 40
        verts, quads, bed, bid = gmsh.forward_step_quads(cl=0.2 / k, seed=1)
        mesh = dflo_amd.Mesh.from_quads(verts, quads, bed, bid, 3)
        prm = dflo_amd.Parameters(flux="kfvs", pos_lim=True, cfl=0.02, final_time=1e9, boundary={1: "inflow", 2: "slip", 3: "outflow"})
        claw = dflo_amd.ConservationLaw(mesh, prm)
        cell, face, b, xy = claw.boundary_faces()
        bv = np.stack(problems.forward_step_inflow(xy[..., 0], xy[..., 1]), axis=-1)
        claw.set_boundary_values(0, bv)
        claw.set_boundary_values(1, bv)
        claw.set_initial_condition(mesh.interpolate(problems.forward_step_inflow))
        N = 4
    else:
        deg, flux, nx = int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
        mesh = dflo_amd.Mesh.cartesian(nx, nx, -5.0, -5.0, 10.0 / nx, [-1] * 4, deg)
        claw = dflo_amd.ConservationLaw(mesh, dflo_amd.Parameters(flux=flux, pos_lim="pos" in sys.argv[5:]))
        claw.set_initial_condition(mesh.interpolate(problems.isentropic_vortex))
        N = deg + 1
    claw.advance(5)
    f = _lib.lib.dflo_hip_debug_phase_cycles
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    nsh = mesh.n_cells // 64 + 4096
    buf = np.zeros((nsh, 8, 8), dtype=np.uint64)
    g = f(claw._h, buf.ctypes.data_as(C.c_void_p), nsh)
    b = buf[:min(g, nsh)].astype(np.float64)
    b = b[b[:, 0, :].sum(axis=1) > 0]
    names = ["issue loads", "A: wait loads, regs->LDS", "barrier A", "B: fluxes", "barrier B",
             "C: row update (incl. its 2 barriers)", "barrier", "positivity / dt / reduce"]
    print("%d workgroups; mean s_memtime ticks per wave" % len(b))
    for w in range(N):
        tot = b[:, w, :].sum(axis=1).mean()
        print(" wave %d: total %.0f ticks" % (w, tot))
        for i, n in enumerate(names):
            print("    %-40s %8.1f  (%4.1f%%)" % (n, b[:, w, i].mean(), 100 * b[:, w, i].mean() / tot))
finally:
    shutil.copy("/tmp/base_keep.so", real)
