"""Gmsh `.msh` (format 2.2, ASCII) writer for the transfinite, recombined multi-block rectangles the example
`.geo` files describe.  The reference ships only the `.geo` sources and expects `gmsh -2` to turn them into the
`.msh` its `mesh file` entry names (README.md:70-72); gmsh is not part of this image, so the structured meshes of
examples/{sod_shock_tube,isentropic_vortex,double_mach_reflection}/*.geo are generated here, in the same file
format the reader (dflo_mesh_read_gmsh) and GridIn::read_msh consume: nodes, 2-node lines carrying the
"Physical Line" id, 4-node quadrangles carrying the "Physical Surface" id.
"""
import numpy as np


def write_structured_msh(path, xs, ys, boundary_id, surface_id=100):
    """Rectangle with grid lines xs, ys.  boundary_id(side, mid) -> physical line id for the boundary edge on
    side "bottom" | "right" | "top" | "left" whose midpoint has the running coordinate `mid`."""
    xs, ys = np.asarray(xs, dtype=np.float64), np.asarray(ys, dtype=np.float64)
    nx, ny = len(xs) - 1, len(ys) - 1
    node = lambda i, j: 1 + i + (nx + 1) * j
    lines = []
    for i in range(nx):
        mid = 0.5 * (xs[i] + xs[i + 1])
        lines.append((boundary_id("bottom", mid), node(i, 0), node(i + 1, 0)))
        lines.append((boundary_id("top", mid), node(i + 1, ny), node(i, ny)))
    for j in range(ny):
        mid = 0.5 * (ys[j] + ys[j + 1])
        lines.append((boundary_id("right", mid), node(nx, j), node(nx, j + 1)))
        lines.append((boundary_id("left", mid), node(0, j + 1), node(0, j)))
    with open(path, "w") as f:
        f.write("$MeshFormat\n2.2 0 8\n$EndMeshFormat\n$Nodes\n%d\n" % ((nx + 1) * (ny + 1)))
        for j in range(ny + 1):
            for i in range(nx + 1):
                f.write("%d %.17g %.17g 0\n" % (node(i, j), xs[i], ys[j]))
        f.write("$EndNodes\n$Elements\n%d\n" % (len(lines) + nx * ny))
        e = 1
        for pid, a, b in lines:
            f.write("%d 1 2 %d %d %d %d\n" % (e, pid, pid, a, b))
            e += 1
        for j in range(ny):
            for i in range(nx):
                f.write("%d 3 2 %d 1 %d %d %d %d\n" % (e, surface_id, node(i, j), node(i + 1, j), node(i + 1, j + 1), node(i, j + 1)))
                e += 1
        f.write("$EndElements\n")


def sod_tube(path, nx=101, ny=11, Lx=1.0):
    """examples/sod_shock_tube/tube.geo: nx x ny points, dx = Lx/(nx-1), Ly = dx (ny-1); lines: 0 walls, 1 outlet, 2 inlet."""
    dx = Lx / (nx - 1)
    write_structured_msh(path, np.linspace(0.0, Lx, nx), dx * np.arange(ny),
                         lambda side, mid: {"bottom": 0, "top": 0, "right": 1, "left": 2}[side])


def vortex_square(path, n=101, L=10.0):
    """examples/isentropic_vortex/grid.geo: [-L/2, L/2]^2 with n points per side; lines 1 bottom, 2 right, 3 top, 4 left."""
    p = np.linspace(-0.5 * L, 0.5 * L, n)
    write_structured_msh(path, p, p, lambda side, mid: {"bottom": 1, "right": 2, "top": 3, "left": 4}[side], surface_id=10)


def double_mach(path, ny=101, Lx=4.0, Ly=1.0, x0=1.0 / 6.0):
    """examples/double_mach_reflection/grid.geo: square cells of size dy = Ly/(ny-1), the grid line x = x0 kept;
    lines 0 bottom (x < x0), 1 bottom (x > x0), 2 right, 3 top, 4 left."""
    dy = Ly / (ny - 1)
    n1 = int(np.ceil(x0 / dy))
    n2 = int(np.ceil((Lx - x0) / dy))
    xs = x0 + dy * np.arange(-n1, n2 + 1)
    ys = dy * np.arange(ny)

    def bid(side, mid):
        if side == "bottom":
            return 0 if mid < x0 else 1
        return {"right": 2, "top": 3, "left": 4}[side]

    write_structured_msh(path, xs, ys, bid)


def unstructured_quads(n, Lx=1.0, Ly=1.0, jitter=0.25, seed=0):
    """A fully unstructured all-quadrilateral mesh of [0,Lx] x [0,Ly] (stand-in for gmsh's recombined meshes of
    forward_step / naca0012, which need gmsh): a Delaunay triangulation of a jittered n x n point lattice, every
    triangle cut into three quadrilaterals (centroid + edge midpoints).  Node valences 3..8, arbitrary cell
    orientation.  -> (vertices [nv][2], quads [nq][4] counter-clockwise, boundary edges [ne][2] with side ids
    0 bottom, 1 right, 2 top, 3 left)."""
    from scipy.spatial import Delaunay
    rng = np.random.default_rng(seed)
    xs, ys = np.meshgrid(np.linspace(0.0, Lx, n + 1), np.linspace(0.0, Ly, n + 1), indexing="xy")
    pts = np.stack([xs.reshape(-1), ys.reshape(-1)], axis=1)
    inner = (pts[:, 0] > 0) & (pts[:, 0] < Lx) & (pts[:, 1] > 0) & (pts[:, 1] < Ly)
    pts[inner] += jitter * np.array([Lx / n, Ly / n]) * rng.uniform(-1.0, 1.0, (inner.sum(), 2))
    tri = Delaunay(pts).simplices
    # orient counter-clockwise
    a, b, c = pts[tri[:, 0]], pts[tri[:, 1]], pts[tri[:, 2]]
    cw = (b[:, 0] - a[:, 0]) * (c[:, 1] - a[:, 1]) - (b[:, 1] - a[:, 1]) * (c[:, 0] - a[:, 0]) < 0
    tri[cw] = tri[cw][:, ::-1]
    npts, nt = len(pts), len(tri)
    e = np.stack([tri[:, [0, 1]], tri[:, [1, 2]], tri[:, [2, 0]]], axis=1)              # [t][k] edge k = (t_k, t_k+1)
    key = np.sort(e.reshape(-1, 2), axis=1)
    uniq, inv, counts = np.unique(key, axis=0, return_inverse=True, return_counts=True)
    mid = (npts + inv.reshape(-1)).reshape(nt, 3)                                       # midpoint node of edge k
    cen = npts + len(uniq) + np.arange(nt)                                              # centroid node
    verts = np.concatenate([pts, 0.5 * (pts[uniq[:, 0]] + pts[uniq[:, 1]]), pts[tri].mean(axis=1)])
    quads = np.stack([np.stack([tri[:, k], mid[:, k], cen, mid[:, (k + 2) % 3]], axis=1) for k in range(3)], axis=1).reshape(-1, 4)
    b = np.nonzero(counts == 1)[0]                                                      # boundary edges of the triangulation
    pm = 0.5 * (pts[uniq[b, 0]] + pts[uniq[b, 1]])
    side = np.where(np.abs(pm[:, 1]) < 1e-12, 0, np.where(np.abs(pm[:, 0] - Lx) < 1e-12, 1, np.where(np.abs(pm[:, 1] - Ly) < 1e-12, 2, 3)))
    bedges = np.stack([np.stack([uniq[b, 0], npts + b], axis=1), np.stack([npts + b, uniq[b, 1]], axis=1)], axis=1).reshape(-1, 2)
    bid = np.repeat(side, 2)
    return verts, quads.astype(np.int32), bedges.astype(np.int32), bid.astype(np.int32)


def forward_step_quads(cl=0.05, jitter=0.25, seed=0):
    """The wind tunnel with a step of examples/forward_step/step.geo ([0,3] x [0,1] minus [0.6,3] x [0,0.2]) meshed with
    unstructured quadrilaterals of size ~ cl/2, the way BASELINE config 5 wants it ("drop Transfinite Surface"): lattice
    points of spacing cl (cl must divide 0.2), the interior ones jittered, Delaunay triangles with the ones inside the
    step removed, every triangle cut into three quadrilaterals.  Boundary ids as in step.geo / input.prm: 1 inflow
    (x = 0), 3 outflow (x = 3), 2 every wall.  -> (vertices, quads, boundary edges, boundary ids)"""
    from scipy.spatial import Delaunay
    L1, L, h1, H = 0.6, 3.0, 0.2, 1.0
    nx, ny = int(round(L / cl)), int(round(H / cl))
    assert abs(nx * cl - L) < 1e-12 and abs(round(h1 / cl) * cl - h1) < 1e-12 and abs(round(L1 / cl) * cl - L1) < 1e-12
    i, j = np.meshgrid(np.arange(nx + 1), np.arange(ny + 1), indexing="xy")
    i, j = i.reshape(-1), j.reshape(-1)
    i1, j1 = int(round(L1 / cl)), int(round(h1 / cl))
    keep = ~((i > i1) & (j < j1))                                   # lattice points strictly inside the step are dropped
    i, j = i[keep], j[keep]
    pts = np.stack([i * cl, j * cl], axis=1).astype(np.float64)
    on_wall = (i == 0) | (i == nx) | (j == ny) | ((j == 0) & (i <= i1)) | ((i == i1) & (j <= j1)) | ((j == j1) & (i >= i1))
    rng = np.random.default_rng(seed)
    pts[~on_wall] += jitter * cl * rng.uniform(-1.0, 1.0, ((~on_wall).sum(), 2))
    tri = Delaunay(pts).simplices
    cen = pts[tri].mean(axis=1)
    tri = tri[~((cen[:, 0] > L1) & (cen[:, 1] < h1))]               # the triangles that bridge the notch
    a, b, c = pts[tri[:, 0]], pts[tri[:, 1]], pts[tri[:, 2]]
    area2 = (b[:, 0] - a[:, 0]) * (c[:, 1] - a[:, 1]) - (b[:, 1] - a[:, 1]) * (c[:, 0] - a[:, 0])
    tri = tri[np.abs(area2) > 1e-12 * cl * cl]                      # (collinear wall points can leave slivers)
    area2 = area2[np.abs(area2) > 1e-12 * cl * cl]
    tri[area2 < 0] = tri[area2 < 0][:, ::-1]
    npts, nt = len(pts), len(tri)
    e = np.stack([tri[:, [0, 1]], tri[:, [1, 2]], tri[:, [2, 0]]], axis=1)
    key = np.sort(e.reshape(-1, 2), axis=1)
    uniq, inv, counts = np.unique(key, axis=0, return_inverse=True, return_counts=True)
    mid = (npts + inv.reshape(-1)).reshape(nt, 3)
    cenid = npts + len(uniq) + np.arange(nt)
    verts = np.concatenate([pts, 0.5 * (pts[uniq[:, 0]] + pts[uniq[:, 1]]), pts[tri].mean(axis=1)])
    quads = np.stack([np.stack([tri[:, k], mid[:, k], cenid, mid[:, (k + 2) % 3]], axis=1) for k in range(3)], axis=1).reshape(-1, 4)
    b = np.nonzero(counts == 1)[0]
    pm = 0.5 * (pts[uniq[b, 0]] + pts[uniq[b, 1]])
    side = np.where(np.abs(pm[:, 0]) < 1e-12, 1, np.where(np.abs(pm[:, 0] - L) < 1e-12, 3, 2))
    bedges = np.stack([np.stack([uniq[b, 0], npts + b], axis=1), np.stack([npts + b, uniq[b, 1]], axis=1)], axis=1).reshape(-1, 2)
    return verts, quads.astype(np.int32), bedges.astype(np.int32), np.repeat(side, 2).astype(np.int32)


def write_quads_msh(path, verts, quads, bedges, bid, surface_id=100):
    """Any quadrilateral mesh as Gmsh 2.2 ASCII: nodes, 2-node lines with their "Physical Line" id, 4-node quadrangles."""
    with open(path, "w") as f:
        f.write("$MeshFormat\n2.2 0 8\n$EndMeshFormat\n$Nodes\n%d\n" % len(verts))
        for n, (x, y) in enumerate(verts):
            f.write("%d %.17g %.17g 0\n" % (n + 1, x, y))
        f.write("$EndNodes\n$Elements\n%d\n" % (len(bedges) + len(quads)))
        e = 1
        for (a, b), pid in zip(bedges, bid):
            f.write("%d 1 2 %d %d %d %d\n" % (e, pid, pid, a + 1, b + 1))
            e += 1
        for q in quads:
            f.write("%d 3 2 %d 1 %d %d %d %d\n" % (e, surface_id, q[0] + 1, q[1] + 1, q[2] + 1, q[3] + 1))
            e += 1
        f.write("$EndElements\n")


def forward_step(path, cl=0.05, jitter=0.25, seed=0):
    """step.msh for examples/forward_step/input.prm with `mapping = q1` (unstructured quadrilaterals, see forward_step_quads)."""
    write_quads_msh(path, *forward_step_quads(cl, jitter, seed))
