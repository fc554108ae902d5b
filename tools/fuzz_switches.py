"""Randomised test of the engine's and the driver's run-time switches (dflo_amd/csrc/tunables.h): every switch selects between two
paths that are meant to give the same bits -- a fused pass against the separate one, a list walk against one wavefront per shard,
streaming against plain stores, graph replay against plain launches, ghost traces against ghost cells, pack kernels against peer
copies, one host thread against one per group.  Random configurations (those of tools/fuzz_multi.py), each run once with the
defaults and once with a random handful of switches thrown, on one engine or cut into 2-4 parts; the two runs must agree bit for
bit on the nodal basis (modal basis: 1e-13 when a switch changes the cut into shards or the kind of ghost, see fuzz_multi.py).
usage: python tools/fuzz_switches.py [n_cases] [seed] [max_degree = 3]     (GPU box)"""
import os, sys, time
import numpy as np
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
max_degree = sys.argv[3] if len(sys.argv) > 3 else "3"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.argv = [sys.argv[0], "0", str(seed), max_degree]
import importlib.util
spec = importlib.util.spec_from_file_location("fm", os.path.join(ROOT, "tools", "fuzz_multi.py"))
fm = importlib.util.module_from_spec(spec)
try:
    spec.loader.exec_module(fm)
except SystemExit:
    pass
import dflo_amd
srng = np.random.default_rng([seed, 77])
ENGINE = [("DFLO_GRAPH", ["1"]), ("DFLO_SWEEP", ["0"]), ("DFLO_STREAM", ["0", "1"]), ("DFLO_FUSE_DTQ", ["0"]), ("DFLO_FUSE_POS", ["0"]),
          ("DFLO_FUSE_FIN", ["0"]), ("DFLO_LAZY_AVG", ["0"]), ("DFLO_LXF_FROM_DOFS", ["0"]), ("DFLO_LIM_LIST", ["0"]),
          ("DFLO_LIM_GRID", ["1", "7", "64", "333", "4096"]), ("DFLO_LIM_MASK", ["0", "1"]), ("DFLO_PLAN_REFINE", ["0", "2"]),
          ("DFLO_PLAN_RIM_FIRST", ["0"])]
MULTI = [("DFLO_HALO_CELLS", ["1"]), ("DFLO_MULTI_GROUP", ["part", "device"]), ("DFLO_MULTI_THREADS", ["0"]), ("DFLO_MULTI_STRICT", ["1"]),
         ("DFLO_MULTI_COPY", ["1"]), ("DFLO_MULTI_PRIORITY", ["0"]), ("DFLO_MULTI_AVG_UNPACK", ["1"]), ("DFLO_PEER_FINEGRAINED", ["1"]),
         ("DFLO_TVB_ONE_EXCHANGE", ["1"])]   # (round 6: one exchange per TVB stage; in one process the default is the reference's two)
# round 5: a third arrangement -- ONE part that is its own neighbour through the IPC transport's kernels (dflo_hip_multi_create_self) --
# with the switches of that transport: the stage kernel delivering its traces itself or the rim launch + pack kernel on a second stream
SELF = [("DFLO_IPC_STRICT", ["1"]), ("DFLO_IPC_FUSED", ["0"]), ("DFLO_IPC_KWAIT", ["0"]), ("DFLO_PEER_FINEGRAINED", ["0"]), ("DFLO_MULTI_PRIORITY", ["0"]), ("DFLO_MULTI_AVG_UNPACK", ["1"]), ("DFLO_HALO_CELLS", ["1"])]
RESHARD = {"DFLO_PLAN_REFINE", "DFLO_HALO_CELLS"}   # (these change which cells share a shard, or how a ghost cell gives its trace)


def build(case, parts):
    d = case["desc"]
    if parts == "self":
        return dflo_amd.MultiConservationLaw.for_self(case["mesh"], case["prm"], 0, transport="ipc", partitioner=d["partitioner"])
    if parts:
        return dflo_amd.MultiConservationLaw(case["mesh"], case["prm"], devices=[0] * d["parts"], partitioner=d["partitioner"])
    return dflo_amd.ConservationLaw(case["mesh"], case["prm"])


def run_with(case, parts, env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        claw = build(case, parts)
        try:
            fm.setup(case, claw)
            return fm.run(case, claw)
        finally:
            claw.close()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


fails, counts = 0, {}
t0 = time.time()
for i in range(n_cases):
    case = fm.make_case(i)
    d = case["desc"]
    parts = [False, True, "self"][int(srng.integers(0, 3))]
    pool = ENGINE + (SELF if parts == "self" else (MULTI if parts else []))
    picks = srng.choice(len(pool), size=int(srng.integers(1, 5)), replace=False)
    env = {pool[k][0]: str(srng.choice(pool[k][1])) for k in picks}
    try:
        a = run_with(case, parts, {})
    except dflo_amd.DfloError as e:
        if e.code in (-7, -1):
            counts["refused"] = counts.get("refused", 0) + 1
            continue
        raise
    b = run_with(case, parts, env)
    bar = 1e-13 if (d["basis"] == "Pk" and (RESHARD & set(env))) else 0.0
    bar_avg = bar
    # Two switches are not bit-neutral:
    #  * DFLO_FUSE_POS=0 selects ANOTHER INSTANTIATION of the stage kernel (POS 0 + the limiter pass instead of POS 1).  hipcc's
    #    default -ffp-contract=fast fuses multiply-adds across statements wherever the optimiser sees them, and it sees them
    #    differently in the two instantiations: states move by 1-5e-16 even when the limiter changes nothing (measured: with
    #    -ffp-contract=on, fusing only what one source expression spells, the switch is bit-neutral -- at -0.5 % (Q3 KFVS) to
    #    -1.8 % (C5), so the default stays; profiles/LAB.md R4.11);
    #  * on bilinear cells an average formed on demand (average_kernel: weights w w det J / |K| per node) and one stored by a stage
    #    epilogue (row partials, 1 / |K| at the end) differ in the last bit; DFLO_LAZY_AVG=0 / DFLO_FUSE_POS=0 change which of the two
    #    the caller is handed (the state does not see it).  On squares the two are the same bits (average_rows_kernel).
    if "DFLO_FUSE_POS" in env and d["pos"] and not d["tvb"]:
        bar = max(bar, 1e-13)
        bar_avg = max(bar_avg, 1e-13)
    if ({"DFLO_LAZY_AVG", "DFLO_FUSE_POS"} & set(env)) and d["geo"] != "cart":
        bar_avg = max(bar_avg, 1e-15)
    why = None
    if a["stop"] != b["stop"]:
        why = ("stop", a["stop"], b["stop"])
    else:
        fa, fb = np.isfinite(a["u"]), np.isfinite(b["u"])
        if (fa != fb).any():
            why = ("nan pattern",)
        else:
            e = fm.rel(a["u"][fa], b["u"][fa]) if fa.any() else 0.0
            ea = fm.rel(a["avg"][np.isfinite(a["avg"])], b["avg"][np.isfinite(a["avg"])]) if np.isfinite(a["avg"]).any() and (np.isfinite(a["avg"]) == np.isfinite(b["avg"])).all() else 0.0
            edt = max([abs(x - y) / max(abs(x), 1e-300) for x, y in zip(a["dt"], b["dt"])] + [0.0]) if len(a["dt"]) == len(b["dt"]) else np.inf
            et = 0.0 if a["t"] == b["t"] else abs(a["t"] - b["t"]) / max(abs(a["t"]), 1e-300)
            if max(e, edt, et) > bar or ea > bar_avg:
                why = ("differs", e, ea, edt, et)
            k = "identical" if max(e, ea, edt, et) == 0.0 else "rounding (a switch that is not bit-neutral)"
    if why:
        fails += 1
        print("CASE %d FAILED: %s  switches %s  parts %s  %s" % (i, why, env, parts, d), flush=True)
        k = "FAIL"
    counts[k] = counts.get(k, 0) + 1
print("%d cases, %d failures, outcomes %s, %.1f s" % (n_cases, fails, counts, time.time() - t0))
sys.exit(1 if fails else 0)
