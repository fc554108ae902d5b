"""Developer probe: wall time per step of dflo_hip_advance with (a) per-stage HIP events, (b) plain launches,
(c) the captured 2-step graph, at a launch-bound size (C1: 64x64 Q1 LxF) and at C2."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def run(nx, deg, flux, steps, graph, timing):
    code = r'''
import sys, time
sys.path.insert(0, %r)
import dflo_amd
from dflo_amd import problems
mesh = dflo_amd.Mesh.cartesian(%d, %d, -5.0, -5.0, 10.0 / %d, [-1] * 4, %d)
claw = dflo_amd.ConservationLaw(mesh, dflo_amd.Parameters(flux=%r, final_time=1e9))
claw.set_initial_condition(mesh.interpolate(problems.isentropic_vortex))
claw.advance(8)
if %d: claw.stage_timing(True)
best = 1e9
for rep in range(3):
    t0 = time.perf_counter(); claw.advance(%d); best = min(best, time.perf_counter() - t0)
print(best / %d * 1e6)
''' % (ROOT, nx, nx, nx, deg, flux, timing, steps, steps)
    env = dict(os.environ, DFLO_GRAPH="1" if graph else "0")
    return float(subprocess.check_output([sys.executable, "-c", code], env=env).decode().split()[-1])

for nx, deg, flux, steps in [(64, 1, "lxf", 400), (128, 2, "hllc", 400), (1024, 2, "hllc", 40)]:
    ev = run(nx, deg, flux, steps, 0, 1)
    plain = run(nx, deg, flux, steps, 0, 0)
    graph = run(nx, deg, flux, steps, 1, 0)
    print("%4d^2 Q%d %-5s  us/step: events %.1f  plain %.1f  graph %.1f" % (nx, deg, flux, ev, plain, graph))
