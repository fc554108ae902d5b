"""Randomised differential test of the ONE-PROCESS-PER-PART schedule of the multi-device driver (what `bench.py --gpus N` and a
dflo on MPI run: dflo_hip_multi_create_rank_*; rim on the comm stream, pack, exchange, unpack, all-reduced time step, agreed error
status) against the single engine: `world` real processes on one device, the ranks moving their bytes through the driver's
bring-your-own-transport entry with the gloo callbacks of dflo_amd/gloo_transport.py (RCCL refuses two ranks on one GPU).  The
processes stay up and walk the random configurations of tools/fuzz_multi.py together.
usage: python tools/fuzz_ranks.py [n_cases] [seed] [world = 2] [max_degree = 3]     (GPU box)
Bars as in fuzz_multi.py: nodal basis bit-identical to the single engine, modal basis 1e-13."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 50
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
world = int(sys.argv[3]) if len(sys.argv) > 3 else 2
max_degree = sys.argv[4] if len(sys.argv) > 4 else "3"


def worker(rank, world, port, ret):
    import numpy as np
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    sys.argv = [sys.argv[0], "0", str(seed), max_degree]
    import importlib.util
    spec = importlib.util.spec_from_file_location("fm", os.path.join(ROOT, "tools", "fuzz_multi.py"))
    fm = importlib.util.module_from_spec(spec)
    try:
        spec.loader.exec_module(fm)
    except SystemExit:
        pass
    import dflo_amd
    from dflo_amd.gloo_transport import make_callbacks
    xf, af = make_callbacks("cuda:0")
    fails, counts = 0, {}
    for i in range(n_cases):
        case = fm.make_case(i)
        d, mesh = case["desc"], case["mesh"]
        if fm.ONLY and i not in fm.ONLY:   # FUZZ_CASES=88,92: only these (the generator still draws the others)
            continue
        ref0 = None
        if os.environ.get("FUZZ_REF_TWICE") and rank == 0:   # the single engine before the driver exists as well
            one0 = dflo_amd.ConservationLaw(mesh, case["prm"])
            try:
                fm.setup(case, one0)
                ref0 = fm.run(case, one0)
            finally:
                one0.close()
        try:
            claw = dflo_amd.MultiConservationLaw.for_rank_custom(mesh, case["prm"], 0, rank, world, xf, af, partitioner=d["partitioner"])
        except dflo_amd.DfloError as e:
            if e.code in (-7, -1):
                counts["refused"] = counts.get("refused", 0) + 1
                continue
            print("rank %d: CASE %d could not be created: %s  %s" % (rank, i, e, d), flush=True)
            raise
        try:
            fm.setup(case, claw)
            got = fm.run(case, claw)
            own = claw.part_cells(0)[0]
        finally:
            claw.close()
        if os.environ.get("FUZZ_REF_TWICE"):   # the single engine, three times, on every rank, right after the driver has gone
            runs = []
            for k in range(3):
                o = dflo_amd.ConservationLaw(mesh, case["prm"])
                try:
                    fm.setup(case, o)
                    runs.append(fm.run(case, o)["u"])
                finally:
                    o.close()
            print("  rank %d: three single engines after the driver: %s%s" % (rank, ["same as the first" if np.array_equal(r, runs[0]) else "DIFFERENT %.2e" % fm.rel(r, runs[0]) for r in runs[1:]],
                  "; first vs the one before: %s" % ("identical" if np.array_equal(runs[0], ref0["u"]) else "DIFFERENT %.2e" % fm.rel(runs[0], ref0["u"])) if ref0 is not None else ""), flush=True)
        parts = [None] * world
        dist.all_gather_object(parts, (own, got["u"].reshape(mesh.n_cells, -1)[own], got["stop"], got["dt"], got["t"]))
        if rank:
            continue
        u = np.empty((mesh.n_cells, mesh.ndof))
        for o, a, *_ in parts:
            u[o] = a
        u = u.reshape(-1)
        why = None
        if any(p[2] != parts[0][2] or p[3] != parts[0][3] or p[4] != parts[0][4] for p in parts):
            why = ("the ranks disagree among themselves", [p[2:] for p in parts])
        one = dflo_amd.ConservationLaw(mesh, case["prm"])
        try:
            fm.setup(case, one)
            ref = fm.run(case, one)
        finally:
            one.close()
        if ref0 is not None:
            print("  single engine before / after the driver: %s; ranks vs the one before: %.3e" % (
                "identical" if np.array_equal(ref0["u"], ref["u"]) else "DIFFERENT (%.3e)" % fm.rel(ref0["u"], ref["u"]), fm.rel(u, ref0["u"])), flush=True)
        k = "identical"
        if why is None:
            if ref["stop"] != got["stop"]:
                limited = d["tvb"] or d["pos"]
                if not (limited and ref["stop"] and got["stop"] and ref["stop"][1] == got["stop"][1] and abs(ref["stop"][0] - got["stop"][0]) <= 1):
                    why = ("stop", ref["stop"], got["stop"])
                k = "stop"
            elif ref["stop"]:
                k = "stop"
            else:
                fa, fb = np.isfinite(ref["u"]), np.isfinite(u)
                bar = 1e-13 if d["basis"] == "Pk" else 0.0
                if (fa != fb).any():
                    why = ("nan pattern", int((~fa).sum()), int((~fb).sum()))
                else:
                    e = fm.rel(u[fa], ref["u"][fa]) if fa.any() else 0.0
                    edt = max([abs(x - y) / max(abs(x), 1e-300) for x, y in zip(ref["dt"], got["dt"])] + [0.0])
                    et = 0.0 if ref["t"] == got["t"] else abs(ref["t"] - got["t"]) / max(abs(ref["t"]), 1e-300)
                    if max(e, edt, et) > bar:
                        why = ("differs from the single engine", e, edt, et)
                    k = "identical" if max(e, edt, et) == 0.0 else "rounding (Pk)"
                    if not fa.all():
                        k = "nan"
        if why and os.environ.get("FUZZ_WHERE"):   # which cells, whose, how far from a cut
            du = np.abs(u.reshape(mesh.n_cells, -1) - ref["u"].reshape(mesh.n_cells, -1)).max(axis=1)
            bad = np.nonzero(du > 0)[0]
            owner = np.full(mesh.n_cells, -1)
            for r, (o, *_rest) in enumerate(parts):
                owner[o] = r
            nb = np.asarray(mesh.neighbors)
            on_cut = np.array([any(n >= 0 and owner[n] != owner[c] for n in nb[c]) for c in range(mesh.n_cells)])
            print("  %d of %d cells differ; by owner %s; on a cut: %d of them (%d cells on a cut in all); largest %.3e at cell %d (owner %d, on cut %s); steps %d"
                  % (len(bad), mesh.n_cells, np.bincount(owner[bad], minlength=world).tolist(), int(on_cut[bad].sum()), int(on_cut.sum()),
                     du.max(), int(du.argmax()), owner[du.argmax()], bool(on_cut[du.argmax()]), len(got["dt"])), flush=True)
        if why:
            fails += 1
            k = "FAIL"
            print("CASE %d FAILED: %s  %s" % (i, why, d), flush=True)
        counts[k] = counts.get(k, 0) + 1
    if rank == 0:
        ret["fails"], ret["counts"] = fails, dict(counts)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    import random
    import torch.multiprocessing as mp
    t0 = time.time()
    mgr = mp.get_context("spawn").Manager()
    ret = mgr.dict()
    mp.spawn(worker, args=(world, 29500 + random.randint(0, 2000), ret), nprocs=world, join=True)
    print("%d cases on %d ranks, %d failures, outcomes %s, %.1f s" % (n_cases, world, ret["fails"], ret["counts"], time.time() - t0))
    sys.exit(1 if ret["fails"] else 0)
