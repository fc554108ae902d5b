"""Developer probe: build the engine with -DDFLO_PHASE_TIMING into a scratch .so and print the mean
cycles each stage-kernel phase takes per shard iteration (wave 0 of every workgroup)."""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
csrc = os.path.join(ROOT, "dflo_amd", "csrc")
real = os.path.join(ROOT, "dflo_amd", "libdflo_hip.so")
probe = os.path.join(ROOT, "gpurun_out", "libdflo_hip_probe.so")
os.makedirs(os.path.dirname(probe), exist_ok=True)
import __graft_entry__ as _ge
_ge.compile_engine(probe, extra_flags=["-DDFLO_PHASE_TIMING"], objdir=os.path.join(os.path.dirname(probe), "probe_obj"))
os.replace(real, real + ".keep")
os.symlink(probe, real)
try:
    import dflo_amd
    from dflo_amd import problems, _lib
    deg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    flux = sys.argv[2] if len(sys.argv) > 2 else "hllc"
    nx = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
    if len(sys.argv) > 4 and sys.argv[4] == "unstructured":   # 6 nx^2 bilinear cells (q1 mapping)
        import numpy as np
        from dflo_amd import gmsh
        verts, quads, bed, side = gmsh.unstructured_quads(nx, Lx=10.0, Ly=10.0, seed=1)
        mesh = dflo_amd.Mesh.from_quads(verts - 5.0, quads, bed, side, deg)
        claw = dflo_amd.ConservationLaw(mesh, dflo_amd.Parameters(flux=flux, pos_lim="pos" in sys.argv[5:], boundary={0: "farfield", 1: "farfield", 2: "farfield", 3: "farfield"}))
        cell, face, bid, xy = claw.boundary_faces()
        bv = np.stack(problems.isentropic_vortex(xy[..., 0], xy[..., 1]), axis=-1)
        claw.set_boundary_values(0, bv)
        claw.set_boundary_values(1, bv)
    else:
        mesh = dflo_amd.Mesh.cartesian(nx, nx, -5.0, -5.0, 10.0 / nx, [-1] * 4, deg)
        claw = dflo_amd.ConservationLaw(mesh, dflo_amd.Parameters(flux=flux, pos_lim="pos" in sys.argv[5:]))
    claw.set_initial_condition(mesh.interpolate(problems.isentropic_vortex))
    claw.advance(3)
    f = _lib.lib.dflo_hip_debug_phase_cycles
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    nsh = mesh.n_cells // 64 + 8
    buf = np.zeros((nsh, 4, 8), dtype=np.uint64)
    g = f(claw._h, buf.ctypes.data_as(C.c_void_p), nsh)
    b = buf[:g].astype(np.float64)
    names = ["issue loads", "A: wait loads, regs->LDS", "barrier A", "B: fluxes", "barrier B",
             "C: row update (incl. its 2 barriers)", "barrier", "reduce / averages"]
    N = deg + 1
    print("one workgroup per shard (%d shards), last stage launch; mean cycles per wave (s_memtime ticks)" % g)
    for w in range(N):
        tot = b[:, w, :].sum(axis=1).mean()
        print(" wave %d: total %.0f" % (w, tot))
        for i, n in enumerate(names):
            print("    %-40s %8.0f  (%4.1f%%)" % (n, b[:, w, i].mean(), 100 * b[:, w, i].mean() / tot))
finally:
    os.remove(real)
    os.replace(real + ".keep", real)
