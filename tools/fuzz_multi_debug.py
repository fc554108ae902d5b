"""Developer tool: one case of tools/fuzz_multi.py step by step -- where do the single engine and the parts first differ?
usage: python tools/fuzz_multi_debug.py <seed> <case>"""
import os, sys
import numpy as np
seed, case_no = sys.argv[1], int(sys.argv[2])
sys.argv = [sys.argv[0], "0", seed]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import importlib.util
spec = importlib.util.spec_from_file_location("fm", os.path.join(ROOT, "tools", "fuzz_multi.py"))
fm = importlib.util.module_from_spec(spec)
try:
    spec.loader.exec_module(fm)
except SystemExit:
    pass
import dflo_amd
for i in range(case_no + 1):
    case = fm.make_case(i)
d = case["desc"]
print(d)
mesh = case["mesh"]
single = dflo_amd.ConservationLaw(mesh, case["prm"])
multi = dflo_amd.MultiConservationLaw(mesh, case["prm"], devices=[0] * d["parts"], partitioner=d["partitioner"])
fm.setup(case, single)
fm.setup(case, multi)
owner = np.full(mesh.n_cells, -1)
for p in range(d["parts"]):
    own, ghost = multi.part_cells(p)
    owner[own] = p
nb = mesh.neighbors
cut = np.array([any(n >= 0 and owner[n] != owner[c] for n in nb[c]) for c in range(mesh.n_cells)])
def cmp(name, a, b):
    a, b = np.asarray(a), np.asarray(b)
    if a.shape != b.shape:
        print(name, "shapes", a.shape, b.shape); return
    diff = a != b
    if a.ndim == 1 and a.size == mesh.n_cells * mesh.ndof:
        cells = diff.reshape(mesh.n_cells, -1).any(axis=1)
    elif a.shape[0] == mesh.n_cells:
        cells = diff.reshape(mesh.n_cells, -1).any(axis=1)
    else:
        print(name, "differs" if diff.any() else "equal"); return
    print("%-28s %d of %d cells differ (%d of them on a cut; parts of the differing cells %s), max rel %.2e" % (
        name, cells.sum(), mesh.n_cells, (cells & cut).sum(), np.unique(owner[cells]), np.abs(a - b).max() / max(np.abs(a).max(), 1e-300)))
cmp("averages of the start", single.cell_average, multi.cell_average)
cmp("residual of the start", single.assemble_system(), multi.assemble_system())
dt1, dt2 = single.compute_time_step(), multi.compute_time_step()
print("dt", dt1, dt2, dt1 == dt2)
single.iterate_explicit(dt1); multi.iterate_explicit(dt1)
cmp("state after one step", single.current_solution, multi.current_solution)
if os.environ.get("FM_MORE"):
    single.close(); multi.close()
    single = dflo_amd.ConservationLaw(mesh, case["prm"])
    multi = dflo_amd.MultiConservationLaw(mesh, case["prm"], devices=[0] * d["parts"], partitioner=d["partitioner"])
    fm.setup(case, single)
    fm.setup(case, multi)
    r1, r2 = single.assemble_system().reshape(mesh.n_cells, -1), multi.assemble_system().reshape(mesh.n_cells, -1)
    cells = (r1 != r2).any(axis=1)
    idx = np.nonzero(cells)[0]
    print("first differing cells", idx[:20])
    print("differing DoF columns (count per dof index)", (r1 != r2).sum(axis=0))
    bcell = (nb < 0).any(axis=1)
    print("differing cells on the domain boundary: %d of %d; all boundary cells %d" % ((cells & bcell).sum(), cells.sum(), bcell.sum()))
    cx = mesh.vertices.mean(axis=1)
    for p in range(d["parts"]):
        m = owner == p
        print(" part %d: %d cells, %d differ; x range of differing %s" % (p, m.sum(), (cells & m).sum(), (cx[cells & m, 0].min(), cx[cells & m, 0].max()) if (cells & m).any() else None))
    one = dflo_amd.MultiConservationLaw(mesh, case["prm"], devices=[0], partitioner=d["partitioner"])
    fm.setup(case, one)
    cmp("1 part vs single: residual", single.assemble_system(), one.assemble_system())
    rel = np.abs(r1 - r2) / np.maximum(np.abs(r1), 1e-300)
    print("largest relative difference of a single entry %.2e; ulp-sized (<= 4e-16) entries among the differing: %d of %d" % (rel.max(), ((rel <= 4.5e-16) & (r1 != r2)).sum(), (r1 != r2).sum()))
