"""Developer tool: one fuzz case (tools/fuzz_parity.py) stage by stage -- the update and the limiter pass separately, the oracle's limiter
applied to the DEVICE's unlimited state -- to tell a limiter difference from an update difference.
usage: python tools/fuzz_debug_case.py <case> <seed> <max_degree>   (found fuzz case 2312 of seed 4243: see kernels_common.hpp, gll_point)"""
import os, sys
os.environ["FUZZ_NPERT"] = "1"
os.environ["DFLO_FUSE_POS"] = "0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # tools/ -> repo
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
src = open(os.path.join(ROOT, "tools", "fuzz_parity.py")).read().split("\nstats = {}")[0]
CASE, SEED, MAXD = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
sys.argv = ["x", str(CASE + 1), str(SEED), str(MAXD)]
g = {"__name__": "fz", "__file__": os.path.join(ROOT, "tools", "fuzz_parity.py")}
exec(src, g)
g["last"] = {}
for i in range(CASE + 1):
    case = g["make_case"](i)
import dflo_amd, oracle_lib
from dflo_amd._lib import lib
print(case["desc"])
mesh, prm = case["mesh"], case["prm"]
claw = dflo_amd.ConservationLaw(mesh, prm)
ora = g["new_oracle"](case, case["u0"])
bv = g["boundary_values"](case, claw)
if bv is not None:
    for w in (0, 1):
        claw.set_boundary_values(w, bv)
claw.set_initial_condition(case["u0"])
t = 0.0
nc = mesh.n_cells
for it in range(4):
    dt = ora.compute_time_step(t)
    dtc = claw.compute_time_step()
    print("step", it, "dt", dt, dtc)
    for rk in range(ora.n_rk):
        rc = lib.dflo_hip_stage_update(claw._h, rk, dt)
        assert rc == 0, rc
        pre = claw.current_solution.copy()
        # the oracle's limiter on the DEVICE's unlimited state
        o3 = g["new_oracle"](case, case["u0"])
        o3.set_current_only(pre)
        o3.compute_cell_average()
        try:
            o3.apply_positivity_limiter()
            o3_ok = "oracle limiter on device's pre-state: fine"
            lim_o = o3.get_solution()
        except oracle_lib.OracleError as e:
            o3_ok = "oracle limiter on device's pre-state: %s" % e
            lim_o = None
        rc = lib.dflo_hip_stage_limit(claw._h)
        rcc = lib.dflo_hip_check(claw._h)
        post = claw.current_solution.copy()
        try:
            if not case["local"]:
                ora.set_dt(dt)
            ora.stage(rk)
            ost = "ok"
        except oracle_lib.OracleError as e:
            ost = str(e)
        uo = ora.get_solution()
        d = np.abs(post - uo).reshape(nc, -1).max(axis=1)
        print("  stage", rk, "device limit rc", rc, rcc, "| oracle stage:", ost, "|", o3_ok, "| max diff dev-ora", np.nanmax(d), "cell", int(np.nanargmax(d)),
              "| dev-limited vs oracle-limited(dev pre):", (np.abs(post - lim_o).max() if lim_o is not None else None))
        if rcc:
            # which cells did the device change / where is the trouble: compare pre with post
            ch = np.where(np.abs(post - pre).reshape(nc, -1).max(axis=1) > 0)[0]
            print("   cells changed by the device limiter:", ch[:20])
            A = pre.reshape(nc, 4, -1)
            avg = o3.get_cell_average()
            pr = 0.4 * (avg[:, 3] - 0.5 * (avg[:, 0] ** 2 + avg[:, 1] ** 2) / avg[:, 2])
            print("   min avg density / pressure:", avg[:, 2].min(), pr.min(), "at", int(np.argmin(pr)))
            sys.exit(0)
    claw.end_step(); ora.end_step(); t += dt
