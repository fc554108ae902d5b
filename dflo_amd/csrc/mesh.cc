// mesh.cc -- host-side construction of the flat mesh description (dflo_mesh_t).
//
// dflo gets its mesh from deal.II (GridIn::read_msh -> Triangulation -> DoFHandler,
// src/claw.cc:957-967, 271-298).  The engine needs the same information as flat arrays:
// cell vertices in deal.II's lexicographic order, face neighbours, boundary ids.  These
// builders produce it without deal.II: a structured generator for the Cartesian example
// meshes (what "gmsh -2" writes for the transfinite .geo files of the examples), a
// general quad-soup importer, a Gmsh v2 reader and the owned+ghost partition that replaces
// parallel::distributed::Triangulation (src_mpi/claw.h:220).
#include "abi.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <unordered_map>
#include <vector>

namespace {

thread_local std::string g_mesh_error;

struct MeshOwner {
  dflo_mesh_t m;  // must stay the first member
  std::vector<double> vert;
  std::vector<int32_t> nbr, nbrf;
  std::vector<int64_t> gid;
  std::vector<int32_t> send_cells, send_off, recv_off;
  void bind() {
    m.cell_vertices = vert.data();
    m.cell_face_neighbor = nbr.data();
    m.cell_face_neighbor_face = nbrf.data();
    m.cell_global_id = gid.empty() ? nullptr : gid.data();
  }
};

int fail(int code, const std::string &msg) {
  g_mesh_error = msg;
  return code;
}

// local vertex pairs of the four faces, in the direction of increasing free coordinate
const int kFaceVerts[4][2] = {{0, 2}, {1, 3}, {0, 1}, {2, 3}};

}  // namespace

extern "C" {

const char *dflo_mesh_last_error(void) { return g_mesh_error.c_str(); }

void dflo_mesh_free(dflo_mesh_t *mesh) {
  if (mesh) delete reinterpret_cast<MeshOwner *>(mesh);
}

int dflo_mesh_cartesian(int32_t nx, int32_t ny, double x0, double y0, double h, const int32_t side_bc[4],
                        int32_t degree, dflo_mesh_t **out) {
  if (!out || nx < 1 || ny < 1 || !(h > 0) || !side_bc) return fail(DFLO_ERR_BAD_PARAM, "dflo_mesh_cartesian: bad arguments");
  if (degree < 0 || degree > DFLO_MAX_DEGREE) return fail(DFLO_ERR_BAD_PARAM, "degree out of range");
  if ((side_bc[0] < 0) != (side_bc[1] < 0) || (side_bc[2] < 0) != (side_bc[3] < 0))
    return fail(DFLO_ERR_BAD_PARAM, "periodic sides must come in opposite pairs");
  for (int s = 0; s < 4; ++s)
    if (side_bc[s] >= DFLO_MAX_BOUNDARIES) return fail(DFLO_ERR_BAD_PARAM, "boundary id >= max_n_boundaries");
  const int64_t n = (int64_t)nx * ny;
  if (n > 2000000000LL) return fail(DFLO_ERR_BAD_PARAM, "too many cells");
  MeshOwner *o = new MeshOwner;
  o->vert.resize((size_t)n * 8);
  o->nbr.resize((size_t)n * 4);
  o->nbrf.resize((size_t)n * 4);
  for (int j = 0; j < ny; ++j)
    for (int i = 0; i < nx; ++i) {
      const size_t c = (size_t)i + (size_t)nx * j;
      const double xa = x0 + i * h, xb = x0 + (i + 1) * h, ya = y0 + j * h, yb = y0 + (j + 1) * h;
      double *v = &o->vert[c * 8];
      v[0] = xa; v[1] = ya; v[2] = xb; v[3] = ya; v[4] = xa; v[5] = yb; v[6] = xb; v[7] = yb;
      int32_t *nb = &o->nbr[c * 4], *nf = &o->nbrf[c * 4];
      // face 0: x = xa
      if (i > 0) { nb[0] = (int32_t)(c - 1); nf[0] = 1; }
      else if (side_bc[0] < 0) { nb[0] = (int32_t)(c + nx - 1); nf[0] = 1 | 8; }
      else { nb[0] = DFLO_NBR_BOUNDARY(side_bc[0]); nf[0] = 0; }
      if (i < nx - 1) { nb[1] = (int32_t)(c + 1); nf[1] = 0; }
      else if (side_bc[1] < 0) { nb[1] = (int32_t)(c - (nx - 1)); nf[1] = 0 | 8; }
      else { nb[1] = DFLO_NBR_BOUNDARY(side_bc[1]); nf[1] = 0; }
      if (j > 0) { nb[2] = (int32_t)(c - nx); nf[2] = 3; }
      else if (side_bc[2] < 0) { nb[2] = (int32_t)(c + (size_t)nx * (ny - 1)); nf[2] = 3 | 8; }
      else { nb[2] = DFLO_NBR_BOUNDARY(side_bc[2]); nf[2] = 0; }
      if (j < ny - 1) { nb[3] = (int32_t)(c + nx); nf[3] = 2; }
      else if (side_bc[3] < 0) { nb[3] = (int32_t)(c - (size_t)nx * (ny - 1)); nf[3] = 2 | 8; }
      else { nb[3] = DFLO_NBR_BOUNDARY(side_bc[3]); nf[3] = 0; }
    }
  o->m.n_cells = (int32_t)n;
  o->m.n_owned_cells = (int32_t)n;
  o->m.degree = degree;
  o->m.basis = DFLO_BASIS_QK;
  o->m.mapping = DFLO_MAP_CARTESIAN;
  o->bind();
  *out = &o->m;
  return DFLO_OK;
}

int dflo_mesh_from_quads(int32_t n_vertices, const double *vertices, int32_t n_quads, const int32_t *quads,
                         int32_t n_bedges, const int32_t *bedges, const int32_t *bedge_id, int32_t degree,
                         dflo_mesh_t **out) {
  if (!out || !vertices || !quads || n_quads < 1) return fail(DFLO_ERR_BAD_PARAM, "dflo_mesh_from_quads: bad arguments");
  if (degree < 0 || degree > DFLO_MAX_DEGREE) return fail(DFLO_ERR_BAD_PARAM, "degree out of range");
  MeshOwner *o = new MeshOwner;
  o->vert.resize((size_t)n_quads * 8);
  o->nbr.assign((size_t)n_quads * 4, DFLO_NBR_NONE);
  o->nbrf.assign((size_t)n_quads * 4, 0);
  std::vector<int32_t> lex((size_t)n_quads * 4);  // vertex ids in deal.II order
  for (int c = 0; c < n_quads; ++c) {
    int32_t q[4];
    for (int k = 0; k < 4; ++k) {
      q[k] = quads[c * 4 + k];
      if (q[k] < 0 || q[k] >= n_vertices) {
        delete o;
        return fail(DFLO_ERR_BAD_PARAM, "quad vertex index out of range");
      }
    }
    // make the loop counter-clockwise
    double area2 = 0;
    for (int k = 0; k < 4; ++k) {
      const double *a = &vertices[q[k] * 2], *b = &vertices[q[(k + 1) % 4] * 2];
      area2 += a[0] * b[1] - b[0] * a[1];
    }
    if (area2 < 0) std::swap(q[1], q[3]);
    // start the loop at the lower-left-most vertex so axis-aligned cells get xi along +x
    int s = 0;
    for (int k = 1; k < 4; ++k) {
      const double *a = &vertices[q[k] * 2], *b = &vertices[q[s] * 2];
      if (a[0] + a[1] < b[0] + b[1] - 1e-14 * (std::fabs(b[0]) + std::fabs(b[1]) + 1)) s = k;
    }
    int32_t r[4] = {q[s], q[(s + 1) % 4], q[(s + 2) % 4], q[(s + 3) % 4]};
    // loop (r0,r1,r2,r3) -> lexicographic (v0,v1,v2,v3) = (r0,r1,r3,r2)
    int32_t l[4] = {r[0], r[1], r[3], r[2]};
    for (int k = 0; k < 4; ++k) {
      lex[c * 4 + k] = l[k];
      o->vert[(size_t)c * 8 + k * 2 + 0] = vertices[l[k] * 2 + 0];
      o->vert[(size_t)c * 8 + k * 2 + 1] = vertices[l[k] * 2 + 1];
    }
  }
  // edge -> (cell, face) map
  struct Half { int32_t cell, face, start; };
  std::unordered_map<uint64_t, std::vector<Half>> edges;
  edges.reserve((size_t)n_quads * 3);
  auto key = [](int32_t a, int32_t b) { return ((uint64_t)(uint32_t)std::min(a, b) << 32) | (uint32_t)std::max(a, b); };
  for (int c = 0; c < n_quads; ++c)
    for (int f = 0; f < 4; ++f) {
      int32_t a = lex[c * 4 + kFaceVerts[f][0]], b = lex[c * 4 + kFaceVerts[f][1]];
      edges[key(a, b)].push_back({c, f, a});
    }
  std::unordered_map<uint64_t, int32_t> bid;
  for (int e = 0; e < n_bedges; ++e) bid[key(bedges[e * 2], bedges[e * 2 + 1])] = bedge_id ? bedge_id[e] : 0;
  for (auto &kv : edges) {
    auto &hs = kv.second;
    if (hs.size() == 2) {
      for (int s = 0; s < 2; ++s) {
        const Half &me = hs[s], &ot = hs[1 - s];
        o->nbr[(size_t)me.cell * 4 + me.face] = ot.cell;
        o->nbrf[(size_t)me.cell * 4 + me.face] = ot.face | (me.start != ot.start ? 4 : 0);
      }
    } else if (hs.size() == 1) {
      auto it = bid.find(kv.first);
      int32_t id = it == bid.end() ? 0 : it->second;
      if (id < 0 || id >= DFLO_MAX_BOUNDARIES) {
        delete o;
        return fail(DFLO_ERR_BAD_PARAM, "boundary id out of range [0,10)");
      }
      o->nbr[(size_t)hs[0].cell * 4 + hs[0].face] = DFLO_NBR_BOUNDARY(id);
    } else {
      delete o;
      return fail(DFLO_ERR_BAD_PARAM, "non-manifold edge in quad mesh");
    }
  }
  o->m.n_cells = n_quads;
  o->m.n_owned_cells = n_quads;
  o->m.degree = degree;
  o->m.basis = DFLO_BASIS_QK;
  o->m.mapping = DFLO_MAP_Q1;
  o->bind();
  *out = &o->m;
  return DFLO_OK;
}

int dflo_mesh_read_gmsh(const char *path, int32_t degree, int32_t mapping, dflo_mesh_t **out) {
  std::ifstream in(path);
  if (!in) return fail(DFLO_ERR_BAD_PARAM, std::string("cannot open mesh file ") + path);
  std::string line;
  std::vector<double> verts;
  std::vector<int32_t> quads, bedges, bids;
  std::unordered_map<long long, int32_t> node_index;
  while (std::getline(in, line)) {
    if (line.rfind("$MeshFormat", 0) == 0) {
      double ver; int ft, ds;
      in >> ver >> ft >> ds;
      if (ver >= 3.0 || ft != 0) return fail(DFLO_ERR_UNSUPPORTED, "only Gmsh v2 ASCII .msh is supported");
    } else if (line.rfind("$Nodes", 0) == 0) {
      long long nn;
      in >> nn;
      verts.resize((size_t)nn * 2);
      for (long long i = 0; i < nn; ++i) {
        long long id; double x, y, z;
        in >> id >> x >> y >> z;
        node_index[id] = (int32_t)i;
        verts[i * 2] = x;
        verts[i * 2 + 1] = y;
      }
    } else if (line.rfind("$Elements", 0) == 0) {
      long long ne;
      in >> ne;
      for (long long i = 0; i < ne; ++i) {
        long long id; int type, ntags;
        in >> id >> type >> ntags;
        std::vector<long long> tags(ntags);
        for (int t = 0; t < ntags; ++t) in >> tags[t];
        int nn = type == 1 ? 2 : type == 3 ? 4 : type == 15 ? 1 : type == 2 ? 3 : -1;
        if (nn < 0) return fail(DFLO_ERR_UNSUPPORTED, "unsupported gmsh element type");
        long long nodes[4];
        for (int k = 0; k < nn; ++k) in >> nodes[k];
        if (type == 2) return fail(DFLO_ERR_UNSUPPORTED, "triangles in mesh: dflo needs all-quad meshes");
        if (type == 1) {
          bedges.push_back(node_index.at(nodes[0]));
          bedges.push_back(node_index.at(nodes[1]));
          bids.push_back(ntags > 0 ? (int32_t)tags[0] : 0);  // physical id -> boundary_id (Appendix A.11)
        } else if (type == 3) {
          for (int k = 0; k < 4; ++k) quads.push_back(node_index.at(nodes[k]));
        }
      }
    }
  }
  if (quads.empty()) return fail(DFLO_ERR_BAD_PARAM, "no quadrilaterals in mesh file");
  int rc = dflo_mesh_from_quads((int32_t)(verts.size() / 2), verts.data(), (int32_t)(quads.size() / 4), quads.data(),
                                (int32_t)bids.size(), bedges.data(), bids.data(), degree, out);
  if (rc) return rc;
  MeshOwner *o = reinterpret_cast<MeshOwner *>(*out);
  o->m.mapping = mapping;
  return DFLO_OK;
}

// Owner rank of every cell.  method DFLO_PART_SLAB: contiguous chunks of the cells sorted by centroid (x, then y) --
// x-slabs on structured meshes (C4: 4001 columns -> 8 slabs, 1000 faces per cut).  DFLO_PART_RCB: recursive
// coordinate bisection of the centroids (the longer extent is cut, the ranks are split as evenly as they go, the
// cells in proportion) -- compact blocks on unstructured meshes (C5), what p4est's Morton partition gives the MPI
// variant (src_mpi/claw.h:220).  Deterministic: every rank computes the same owners from the same mesh.
static void rcb_split(std::vector<int32_t> &cells, size_t lo, size_t hi, int r0, int r1, const std::vector<double> &cx,
                      const std::vector<double> &cy, std::vector<int32_t> &owner) {
  if (r1 - r0 <= 1 || hi <= lo) {
    for (size_t k = lo; k < hi; ++k) owner[cells[k]] = r0;
    return;
  }
  double x0 = 1e300, x1 = -1e300, y0 = 1e300, y1 = -1e300;
  for (size_t k = lo; k < hi; ++k) {
    const int32_t c = cells[k];
    x0 = std::min(x0, cx[c]); x1 = std::max(x1, cx[c]);
    y0 = std::min(y0, cy[c]); y1 = std::max(y1, cy[c]);
  }
  const bool cut_x = (x1 - x0) >= (y1 - y0);
  const std::vector<double> &a = cut_x ? cx : cy, &b = cut_x ? cy : cx;
  const double a0 = cut_x ? x0 : y0, ea = std::max((cut_x ? x1 - x0 : y1 - y0), 1e-300) * 1e-9;
  const double b0 = cut_x ? y0 : x0, eb = std::max((cut_x ? y1 - y0 : x1 - x0), 1e-300) * 1e-9;
  std::sort(cells.begin() + lo, cells.begin() + hi, [&](int32_t p, int32_t q) {   // quantised keys: a strict weak order
    const long long pa = std::llround((a[p] - a0) / ea), qa = std::llround((a[q] - a0) / ea);
    if (pa != qa) return pa < qa;
    const long long pb = std::llround((b[p] - b0) / eb), qb = std::llround((b[q] - b0) / eb);
    if (pb != qb) return pb < qb;
    return p < q;
  });
  const int rl = (r1 - r0) / 2;
  const size_t mid = lo + (size_t)((long long)(hi - lo) * rl / (r1 - r0));
  rcb_split(cells, lo, mid, r0, r0 + rl, cx, cy, owner);
  rcb_split(cells, mid, hi, r0 + rl, r1, cx, cy, owner);
}

static void partition_owners(const dflo_mesh_t *mesh, int32_t n_ranks, int32_t method, std::vector<int32_t> &owner) {
  const int32_t n = mesh->n_cells;
  std::vector<int32_t> order(n);
  std::vector<double> cx(n), cy(n);
  for (int c = 0; c < n; ++c) {
    order[c] = c;
    const double *v = &mesh->cell_vertices[(size_t)c * 8];
    cx[c] = 0.25 * (v[0] + v[2] + v[4] + v[6]);
    cy[c] = 0.25 * (v[1] + v[3] + v[5] + v[7]);
  }
  owner.assign(n, 0);
  if (method == DFLO_PART_RCB) {
    rcb_split(order, 0, (size_t)n, 0, n_ranks, cx, cy, owner);
    return;
  }
  // order cells by centroid (x, then y): contiguous chunks are x-slabs on structured meshes
  double xmin = 1e300, xmax = -1e300;
  for (int c = 0; c < n; ++c) { xmin = std::min(xmin, cx[c]); xmax = std::max(xmax, cx[c]); }
  const double tol = 1e-9 * (xmax - xmin + 1e-300);
  std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) {
    if (std::fabs(cx[a] - cx[b]) > tol) return cx[a] < cx[b];
    return cy[a] < cy[b];
  });
  for (int64_t k = 0; k < n; ++k) owner[order[k]] = (int32_t)(k * n_ranks / n);
}

int dflo_mesh_partition(const dflo_mesh_t *mesh, int32_t n_ranks, int32_t rank, dflo_mesh_t **out,
                        const int32_t **send_cells, const int32_t **send_offsets, const int32_t **recv_offsets) {
  return dflo_mesh_partition_ex(mesh, n_ranks, rank, DFLO_PART_SLAB, out, send_cells, send_offsets, recv_offsets);
}

int dflo_mesh_partition_owners(const dflo_mesh_t *mesh, int32_t n_ranks, int32_t method, int32_t *owner_out) {
  if (!mesh || !owner_out || n_ranks < 1 || (method != DFLO_PART_SLAB && method != DFLO_PART_RCB))
    return fail(DFLO_ERR_BAD_PARAM, "dflo_mesh_partition_owners: bad arguments");
  std::vector<int32_t> owner;
  partition_owners(mesh, n_ranks, method, owner);
  std::memcpy(owner_out, owner.data(), owner.size() * sizeof(int32_t));
  return DFLO_OK;
}

int dflo_mesh_partition_ex(const dflo_mesh_t *mesh, int32_t n_ranks, int32_t rank, int32_t method, dflo_mesh_t **out,
                           const int32_t **send_cells, const int32_t **send_offsets, const int32_t **recv_offsets) {
  if (!mesh || !out || n_ranks < 1 || rank < 0 || rank >= n_ranks || (method != DFLO_PART_SLAB && method != DFLO_PART_RCB))
    return fail(DFLO_ERR_BAD_PARAM, "dflo_mesh_partition: bad arguments");
  const int32_t n = mesh->n_cells;
  if (mesh->n_owned_cells != n) return fail(DFLO_ERR_BAD_PARAM, "mesh is already partitioned");
  std::vector<int32_t> owner;
  partition_owners(mesh, n_ranks, method, owner);
  // owned cells keep the global order; ghosts sorted by (source rank, global id)
  std::vector<int32_t> local;
  for (int c = 0; c < n; ++c)
    if (owner[c] == rank) local.push_back(c);
  const int32_t n_owned = (int32_t)local.size();
  std::vector<std::vector<int32_t>> ghosts(n_ranks), sends(n_ranks);
  std::vector<char> mark(n, 0);
  for (int c = 0; c < n; ++c) {
    if (owner[c] != rank) continue;
    for (int f = 0; f < 4; ++f) {
      int32_t nb = mesh->cell_face_neighbor[(size_t)c * 4 + f];
      if (nb < 0 || owner[nb] == rank) continue;
      if (!(mark[nb] & 1)) { mark[nb] |= 1; ghosts[owner[nb]].push_back(nb); }
    }
  }
  // cell c (owned here) is sent to rank s iff some face neighbour of c is owned by s
  for (int c = 0; c < n; ++c) {
    if (owner[c] != rank) continue;
    int32_t seen[4]; int ns = 0;
    for (int f = 0; f < 4; ++f) {
      int32_t nb = mesh->cell_face_neighbor[(size_t)c * 4 + f];
      if (nb < 0 || owner[nb] == rank) continue;
      int32_t s = owner[nb];
      bool dup = false;
      for (int k = 0; k < ns; ++k) dup |= seen[k] == s;
      if (!dup) { seen[ns++] = s; sends[s].push_back(c); }
    }
  }
  MeshOwner *o = new MeshOwner;
  o->recv_off.assign(n_ranks + 1, 0);
  o->send_off.assign(n_ranks + 1, 0);
  for (int r = 0; r < n_ranks; ++r) {
    std::sort(ghosts[r].begin(), ghosts[r].end());
    std::sort(sends[r].begin(), sends[r].end());
    o->recv_off[r + 1] = o->recv_off[r] + (int32_t)ghosts[r].size();
    o->send_off[r + 1] = o->send_off[r] + (int32_t)sends[r].size();
    for (int32_t g : ghosts[r]) local.push_back(g);
  }
  const int32_t nl = (int32_t)local.size();
  std::unordered_map<int32_t, int32_t> g2l;
  g2l.reserve((size_t)nl * 2);
  for (int l = 0; l < nl; ++l) g2l[local[l]] = l;
  for (int r = 0; r < n_ranks; ++r)
    for (int32_t c : sends[r]) o->send_cells.push_back(g2l[c]);
  o->vert.resize((size_t)nl * 8);
  o->nbr.resize((size_t)nl * 4);
  o->nbrf.resize((size_t)nl * 4);
  o->gid.resize(nl);
  for (int l = 0; l < nl; ++l) {
    const int32_t c = local[l];
    o->gid[l] = mesh->cell_global_id ? mesh->cell_global_id[c] : c;
    std::memcpy(&o->vert[(size_t)l * 8], &mesh->cell_vertices[(size_t)c * 8], 8 * sizeof(double));
    for (int f = 0; f < 4; ++f) {
      int32_t nb = mesh->cell_face_neighbor[(size_t)c * 4 + f];
      int32_t nf = mesh->cell_face_neighbor_face[(size_t)c * 4 + f];
      if (nb >= 0) {
        auto it = g2l.find(nb);
        nb = it == g2l.end() ? DFLO_NBR_NONE : it->second;
        // two ghost cells never exchange a flux
        if (l >= n_owned && nb != DFLO_NBR_NONE && nb >= n_owned) nb = DFLO_NBR_NONE;
      }
      o->nbr[(size_t)l * 4 + f] = nb;
      o->nbrf[(size_t)l * 4 + f] = nf;
    }
  }
  o->m.n_cells = nl;
  o->m.n_owned_cells = n_owned;
  o->m.degree = mesh->degree;
  o->m.basis = mesh->basis;
  o->m.mapping = mesh->mapping;
  o->bind();
  *out = &o->m;
  if (send_cells) *send_cells = o->send_cells.data();
  if (send_offsets) *send_offsets = o->send_off.data();
  if (recv_offsets) *recv_offsets = o->recv_off.data();
  return DFLO_OK;
}

// Self-halo partition: ONE part that owns every cell and is its own neighbour across a "virtual cut".  It exists to drive one
// full-size part through the complete multi-device stage schedule (rim / interior split, pack, transport, unpack, time-step
// reduction) on a single GPU: the ratio to the plain engine bounds the per-GPU efficiency of a weak-scaling run without the
// compute-unit sharing that several parts on one device suffer.  The cut:
//   n_virtual >= 2: the faces between the cells of different virtual owners (partition_owners with n_virtual parts), periodic
//                   faces left alone -- n_virtual = 2 on a bounded mesh is one cut through the middle, whose two sides send and
//                   receive what an interior rank of an x-slab run exchanges with its two neighbours;
//   n_virtual == 1: the periodic faces in x (faces 0 / 1 with the periodic bit) -- on the periodic square of C2 one seam, again
//                   two sides: exactly the records, rim shards and ghost cells of one rank of the weak-scaling run.
// Every cell with a cut face gets a ghost COPY (local index n_cells + k, sorted by global id, cell_global_id = the original's);
// across a cut face an owned cell sees the copy of its neighbour, a copy sees the owned cells across its cut faces and nothing
// else -- what dflo_mesh_partition_ex gives a rank for the ghost cells owned by a peer.  send_cells = the same cells (the owner of
// every ghost is this part): offsets [0, n] for the single "peer" 0.
int dflo_mesh_partition_self(const dflo_mesh_t *mesh, int32_t n_virtual, int32_t method, dflo_mesh_t **out,
                             const int32_t **send_cells, const int32_t **send_offsets, const int32_t **recv_offsets) {
  if (!mesh || !out || n_virtual < 1 || (method != DFLO_PART_SLAB && method != DFLO_PART_RCB))
    return fail(DFLO_ERR_BAD_PARAM, "dflo_mesh_partition_self: bad arguments");
  const int32_t n = mesh->n_cells;
  if (mesh->n_owned_cells != n) return fail(DFLO_ERR_BAD_PARAM, "mesh is already partitioned");
  std::vector<int32_t> owner;
  partition_owners(mesh, n_virtual, method, owner);
  auto is_cut = [&](int32_t c, int f) {
    const int32_t nb = mesh->cell_face_neighbor[(size_t)c * 4 + f];
    if (nb < 0 || nb == c) return false;
    const bool periodic = (mesh->cell_face_neighbor_face[(size_t)c * 4 + f] & 8) != 0;
    if (n_virtual == 1) return periodic && f < 2;
    return !periodic && owner[nb] != owner[c];
  };
  std::vector<int32_t> copy_of(n, -1), cut_cells;
  for (int32_t c = 0; c < n; ++c) {
    bool any = false;
    for (int f = 0; f < 4; ++f) any |= is_cut(c, f);
    if (any) { copy_of[c] = n + (int32_t)cut_cells.size(); cut_cells.push_back(c); }
  }
  if (cut_cells.empty()) return fail(DFLO_ERR_BAD_PARAM, n_virtual == 1 ? "dflo_mesh_partition_self: the mesh has no periodic faces in x to cut at (use n_virtual >= 2)"
                                                                         : "dflo_mesh_partition_self: the virtual parts share no face");
  const int32_t ng = (int32_t)cut_cells.size(), nl = n + ng;
  MeshOwner *o = new MeshOwner;
  o->send_off = {0, ng};
  o->recv_off = {0, ng};
  o->send_cells = cut_cells;   // owned cells keep their global numbers
  o->vert.resize((size_t)nl * 8);
  o->nbr.resize((size_t)nl * 4);
  o->nbrf.resize((size_t)nl * 4);
  o->gid.resize(nl);
  for (int32_t l = 0; l < nl; ++l) {
    const int32_t c = l < n ? l : cut_cells[l - n];
    o->gid[l] = mesh->cell_global_id ? mesh->cell_global_id[c] : c;
    std::memcpy(&o->vert[(size_t)l * 8], &mesh->cell_vertices[(size_t)c * 8], 8 * sizeof(double));
    for (int f = 0; f < 4; ++f) {
      int32_t nb = mesh->cell_face_neighbor[(size_t)c * 4 + f];
      if (nb >= 0) {
        const bool cut = is_cut(c, f);
        if (l < n) nb = cut ? copy_of[nb] : nb;   // an owned cell: the copy across a cut face
        else nb = cut ? nb : DFLO_NBR_NONE;       // a copy: the owned cells across its cut faces, nothing else
      }
      o->nbr[(size_t)l * 4 + f] = nb;
      o->nbrf[(size_t)l * 4 + f] = mesh->cell_face_neighbor_face[(size_t)c * 4 + f];
    }
  }
  o->m.n_cells = nl;
  o->m.n_owned_cells = n;
  o->m.degree = mesh->degree;
  o->m.basis = mesh->basis;
  o->m.mapping = mesh->mapping;
  o->bind();
  *out = &o->m;
  if (send_cells) *send_cells = o->send_cells.data();
  if (send_offsets) *send_offsets = o->send_off.data();
  if (recv_offsets) *recv_offsets = o->recv_off.data();
  return DFLO_OK;
}

// GridTools::collect_periodic_faces + add_periodicity (src_mpi/claw.cc:156-200): the boundary faces with ids
// id_first / id_second, offset along `direction` (0 = x, 1 = y), are matched by their extent in the other
// coordinate and become each other's (periodic) neighbours.
int dflo_mesh_make_periodic(dflo_mesh_t *mesh, int32_t id_first, int32_t id_second, int32_t direction) {
  if (!mesh || direction < 0 || direction > 1 || id_first == id_second) return fail(DFLO_ERR_BAD_PARAM, "dflo_mesh_make_periodic: bad arguments");
  if (mesh->n_owned_cells != mesh->n_cells) return fail(DFLO_ERR_BAD_PARAM, "make_periodic works on unpartitioned meshes");
  struct PF { double lo, hi, start; int32_t cell, face; };
  std::vector<PF> A, B;
  const int o = 1 - direction;
  int32_t *nbr = const_cast<int32_t *>(mesh->cell_face_neighbor), *nbrf = const_cast<int32_t *>(mesh->cell_face_neighbor_face);
  for (int32_t c = 0; c < mesh->n_cells; ++c)
    for (int f = 0; f < 4; ++f) {
      const int32_t nb = nbr[(size_t)c * 4 + f];
      if (nb != DFLO_NBR_BOUNDARY(id_first) && nb != DFLO_NBR_BOUNDARY(id_second)) continue;
      const double *v = mesh->cell_vertices + (size_t)c * 8;
      const double p0 = v[kFaceVerts[f][0] * 2 + o], p1 = v[kFaceVerts[f][1] * 2 + o];
      (nb == DFLO_NBR_BOUNDARY(id_first) ? A : B).push_back({std::min(p0, p1), std::max(p0, p1), p0, c, f});
    }
  if (A.empty() || A.size() != B.size()) return fail(DFLO_ERR_BAD_PARAM, "periodic boundaries have different numbers of faces");
  auto by_lo = [](const PF &x, const PF &y) { return x.lo < y.lo; };
  std::sort(A.begin(), A.end(), by_lo);
  std::sort(B.begin(), B.end(), by_lo);
  for (size_t k = 0; k < A.size(); ++k) {
    const double tol = 1.0e-10 * std::max(1.0, std::fabs(A[k].hi - A[k].lo));
    if (std::fabs(A[k].lo - B[k].lo) > tol || std::fabs(A[k].hi - B[k].hi) > tol)
      return fail(DFLO_ERR_BAD_PARAM, "faces of the periodic boundaries do not match");
  }
  for (size_t k = 0; k < A.size(); ++k) {
    const bool flip = std::fabs(A[k].start - B[k].start) > 0.5 * (A[k].hi - A[k].lo);
    nbr[(size_t)A[k].cell * 4 + A[k].face] = B[k].cell;
    nbrf[(size_t)A[k].cell * 4 + A[k].face] = B[k].face | (flip ? 4 : 0) | 8;
    nbr[(size_t)B[k].cell * 4 + B[k].face] = A[k].cell;
    nbrf[(size_t)B[k].cell * 4 + B[k].face] = A[k].face | (flip ? 4 : 0) | 8;
  }
  return DFLO_OK;
}

}  // extern "C"

// ---- support points of the Qk DoFs (unit support points mapped to real space; what
// VectorTools::interpolate evaluates the initial condition at, src/ic.cc:104-121)
#include "basis.h"
extern "C" int dflo_mesh_support_points(const dflo_mesh_t *mesh, double *xy) {
  if (!mesh || !xy) return fail(DFLO_ERR_BAD_PARAM, "dflo_mesh_support_points: null argument");
  // for the Pk basis these are the points of QGauss<2>(k+1) the L2 projection of the initial condition uses
  const dflo::BasisTables b = dflo::make_basis(mesh->degree);
  const int N = b.N;
  for (int c = 0; c < mesh->n_cells; ++c) {
    const double *v = &mesh->cell_vertices[(size_t)c * 8];
    for (int j = 0; j < N * N; ++j) {
      const double xi = b.x[j % N], eta = b.x[j / N];
      for (int d = 0; d < 2; ++d)
        xy[((size_t)c * N * N + j) * 2 + d] = (1 - xi) * (1 - eta) * v[d] + xi * (1 - eta) * v[2 + d] +
                                               (1 - xi) * eta * v[4 + d] + xi * eta * v[6 + d];
    }
  }
  return DFLO_OK;
}
