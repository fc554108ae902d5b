// plan.cc -- builds the shard / face index layout (see plan.h).
#include "plan.h"
#include "tunables.h"

#include <algorithm>
#include <climits>
#include <cstdint>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <unordered_map>

namespace dflo {
namespace {

// index of (x, y) along the Hilbert curve of a 2^order x 2^order lattice: consecutive indices are always
// neighbouring lattice points (a Morton curve jumps), so runs of 64 centroids make compact shards with short rims
inline uint64_t hilbert2(uint32_t x, uint32_t y, int order) {
  uint64_t d = 0;
  for (uint32_t s = 1u << (order - 1); s > 0; s >>= 1) {
    const uint32_t rx = (x & s) ? 1 : 0, ry = (y & s) ? 1 : 0;
    d += (uint64_t)s * s * ((3 * rx) ^ ry);
    if (ry == 0) {   // rotate the quadrant
      if (rx == 1) { x = s - 1 - (x & (s - 1)); y = s - 1 - (y & (s - 1)); }
      else { x &= s - 1; y &= s - 1; }
      const uint32_t t = x; x = y; y = t;
    } else { x &= s - 1; y &= s - 1; }
  }
  return d;
}

inline uint64_t morton2(uint32_t x, uint32_t y) {
  auto spread = [](uint64_t v) {
    v &= 0xFFFFFFFFull;
    v = (v | (v << 16)) & 0x0000FFFF0000FFFFull;
    v = (v | (v << 8)) & 0x00FF00FF00FF00FFull;
    v = (v | (v << 4)) & 0x0F0F0F0F0F0F0F0Full;
    v = (v | (v << 2)) & 0x3333333333333333ull;
    v = (v | (v << 1)) & 0x5555555555555555ull;
    return v;
  };
  return spread(x) | (spread(y) << 1);
}

}  // namespace

int build_plan(const dflo_mesh_t &mesh, int shard_ex, int shard_ey, Plan &p, std::string &err, double h_hint) {
  const int n = mesh.n_cells;
  const int n_owned = mesh.n_owned_cells > 0 ? mesh.n_owned_cells : n;
  if (n < 1 || n_owned > n) { err = "bad cell counts"; return DFLO_ERR_BAD_PARAM; }
  if (shard_ex * shard_ey != kShard) { err = "shard shape must hold 64 cells"; return DFLO_ERR_BAD_PARAM; }
  p.n_cells = n;
  p.n_owned = n_owned;
  const double *V = mesh.cell_vertices;
  auto gid = [&](int c) -> int64_t { return mesh.cell_global_id ? mesh.cell_global_id[c] : c; };

  // ---- geometry checks (compute_cartesian_mesh_size, src/claw.cc:197-221)
  std::vector<double> hx(n);
  double hmin = 1e300, hmax = 0, xmin = 1e300, ymin = 1e300;
  for (int c = 0; c < n; ++c) {
    const double *v = &V[(size_t)c * 8];
    if (mesh.mapping == DFLO_MAP_CARTESIAN) {
      const double dx = v[2] - v[0], dy = v[5] - v[1];
      // face-centre extents as the reference measures them
      const double fx = 0.5 * (v[2] + v[6]) - 0.5 * (v[0] + v[4]);
      const double fy = 0.5 * (v[5] + v[7]) - 0.5 * (v[1] + v[3]);
      if (!(std::fabs(fx - fy) < 1.0e-12) || !(dx > 0) || !(dy > 0) || std::fabs(v[3] - v[1]) > 1e-12 * dx ||
          std::fabs(v[4] - v[0]) > 1e-12 * dx) {
        err = "Cell is not square";
        return DFLO_ERR_NONSQUARE_CELL;
      }
      hx[c] = dx;
    } else {
      hx[c] = std::sqrt(std::fabs((v[2] - v[0]) * (v[5] - v[1]) - (v[3] - v[1]) * (v[4] - v[0])));
    }
    hmin = std::min(hmin, hx[c]);
    hmax = std::max(hmax, hx[c]);
    xmin = std::min(xmin, v[0]);
    ymin = std::min(ymin, v[1]);
  }
  p.uniform_h = mesh.mapping == DFLO_MAP_CARTESIAN && (hmax - hmin) <= 1e-10 * hmax;
  p.h = hmin;
  if (p.uniform_h && h_hint > 0.0 && std::fabs(h_hint - hmin) <= 1e-10 * hmax) p.h = h_hint;   // the whole mesh's (plan.h)

  // ---- assign owned cells to shards
  std::vector<int32_t> shard_of(n, -1), local_of(n, -1);
  std::vector<std::vector<int32_t>> shard_cells;
  if (p.uniform_h) {
    // block shards of shard_ex x shard_ey cells on the lattice, ordered along a Morton curve
    // Inside a block the cells on its rim come first -- left column, right column, bottom row, top row, then the
    // interior: the cells a neighbouring shard reads as halo are then contiguous lanes of a DoF row (one or two
    // 128-byte lines instead of a stride-8 walk over the whole 512-byte row).  The kernels never assume a
    // lane <-> position map, faces and neighbours are explicit.
    struct Key { uint64_t m; int32_t w, c; };
    std::vector<Key> keys(n_owned);
    for (int c = 0; c < n_owned; ++c) {
      const double *v = &V[(size_t)c * 8];
      int32_t i = (int32_t)std::llround((v[0] - xmin) / p.h), j = (int32_t)std::llround((v[1] - ymin) / p.h);
      const int li = i % shard_ex, lj = j % shard_ey;
      int w;
      if (li == 0) w = lj;
      else if (li == shard_ex - 1) w = 64 + lj;
      else if (lj == 0) w = 128 + li;
      else if (lj == shard_ey - 1) w = 192 + li;
      else w = 256 + lj * shard_ex + li;
      keys[c] = {morton2((uint32_t)(i / shard_ex), (uint32_t)(j / shard_ey)), w, c};
    }
    std::sort(keys.begin(), keys.end(), [](const Key &a, const Key &b) {
      if (a.m != b.m) return a.m < b.m;
      return a.w < b.w;
    });
    for (int k = 0; k < n_owned; ++k) {
      if (k == 0 || keys[k].m != keys[k - 1].m || (int)shard_cells.back().size() == kShard) shard_cells.emplace_back();
      shard_cells.back().push_back(keys[k].c);
    }
  } else {
    // unstructured: Hilbert order of the centroids, cut into runs of 64
    double xmax = -1e300, ymax = -1e300;
    for (int c = 0; c < n; ++c) {
      const double *v = &V[(size_t)c * 8];
      for (int k = 0; k < 4; ++k) { xmax = std::max(xmax, v[2 * k]); ymax = std::max(ymax, v[2 * k + 1]); }
    }
    const double span = std::max(xmax - xmin, ymax - ymin) + 1e-300;
    std::vector<std::pair<uint64_t, int32_t>> keys(n_owned);
    for (int c = 0; c < n_owned; ++c) {
      const double *v = &V[(size_t)c * 8];
      double cx = 0.25 * (v[0] + v[2] + v[4] + v[6]), cy = 0.25 * (v[1] + v[3] + v[5] + v[7]);
      uint32_t qx = (uint32_t)((cx - xmin) / span * 1048575.0), qy = (uint32_t)((cy - ymin) / span * 1048575.0);
      keys[c] = {hilbert2(qx, qy, 20), c};
    }
    std::sort(keys.begin(), keys.end());
    std::vector<int32_t> sh(n_owned);
    for (int k = 0; k < n_owned; ++k) sh[keys[k].second] = k / kShard;
    // Refinement: swap pairs of cells between neighbouring shards as long as that shortens the cut (every cell keeps
    // 64 partners, the number of faces between different shards -- halo entries, doubly evaluated fluxes -- only falls).
    auto links = [&](int c, int shard) {
      int k = 0;
      for (int f = 0; f < 4; ++f) {
        const int nb = mesh.cell_face_neighbor[(size_t)c * 4 + f];
        k += nb >= 0 && nb < n_owned && sh[nb] == shard;
      }
      return k;
    };
    const int n_pass = read_tunables().plan_refine;
    for (int pass = 0; pass < n_pass; ++pass) {
      std::unordered_map<uint64_t, std::vector<std::pair<int, int32_t>>> want;   // (from, to) -> (gain, cell)
      for (int c = 0; c < n_owned; ++c) {
        const int own = links(c, sh[c]);
        int best = -1, bg = -5;
        for (int f = 0; f < 4; ++f) {
          const int nb = mesh.cell_face_neighbor[(size_t)c * 4 + f];
          if (nb < 0 || nb >= n_owned || sh[nb] == sh[c]) continue;
          const int g = links(c, sh[nb]) - own;
          if (g > bg) { bg = g; best = sh[nb]; }
        }
        if (best >= 0 && bg >= 0) want[((uint64_t)sh[c] << 32) | (uint32_t)best].push_back({bg, c});
      }
      long swaps = 0;
      for (auto &kv : want) {
        const int A = (int)(kv.first >> 32), B = (int)(kv.first & 0xFFFFFFFFu);
        if (A > B) continue;
        auto it = want.find(((uint64_t)B << 32) | (uint32_t)A);
        if (it == want.end()) continue;
        auto &la = kv.second, &lb = it->second;
        std::sort(la.rbegin(), la.rend());
        std::sort(lb.rbegin(), lb.rend());
        for (size_t i = 0; i < std::min(la.size(), lb.size()); ++i) {
          const int c = la[i].second, d = lb[i].second;
          if (sh[c] != A || sh[d] != B) continue;
          bool adj = false;
          for (int f = 0; f < 4; ++f) adj |= mesh.cell_face_neighbor[(size_t)c * 4 + f] == d;
          const int gain = (links(c, B) - links(c, A)) + (links(d, A) - links(d, B)) - (adj ? 2 : 0);
          if (gain <= 0) continue;
          sh[c] = B;
          sh[d] = A;
          ++swaps;
        }
      }
      if (swaps == 0) break;
    }
    shard_cells.assign((n_owned + kShard - 1) / kShard, {});
    for (int k = 0; k < n_owned; ++k) shard_cells[sh[keys[k].second]].push_back(keys[k].second);   // Hilbert order inside a shard
    // ... and then, as on a lattice, the cells on the shard's rim first, grouped by the neighbouring shard they touch: what a
    // neighbouring shard gathers as halo is then a run of consecutive lanes of every DoF row -- one or two 64-byte sectors per
    // row instead of one per cell scattered over the row's four lines (a halo load that misses the L2 costs a sector of HBM
    // traffic for 8 useful bytes; C5 moved 1.15 x its algorithmic bytes).  Stable: cells of a group keep their Hilbert order.
    if (read_tunables().rim_first) {
      for (auto &cells : shard_cells) {
        std::vector<std::pair<int64_t, int32_t>> order(cells.size());
        for (size_t l = 0; l < cells.size(); ++l) {
          const int c = cells[l];
          int64_t foreign = INT64_MAX;   // the neighbouring shard of the cell with the lowest number (none: interior cell, last)
          for (int f = 0; f < 4; ++f) {
            const int nb = mesh.cell_face_neighbor[(size_t)c * 4 + f];
            if (nb < 0) continue;
            const int64_t s2 = nb < n_owned ? sh[nb] : (int64_t)1 << 40;   // ghost cells: another part's, treated as one far shard
            if (s2 != sh[c]) foreign = std::min(foreign, s2);
          }
          order[l] = {foreign, (int32_t)l};
        }
        std::stable_sort(order.begin(), order.end(), [](const std::pair<int64_t, int32_t> &x, const std::pair<int64_t, int32_t> &y) { return x.first < y.first; });
        std::vector<int32_t> sorted(cells.size());
        for (size_t l = 0; l < cells.size(); ++l) sorted[l] = cells[order[l].second];
        cells.swap(sorted);
      }
    }
  }
  p.n_shards = (int)shard_cells.size();
  const int n_ghost = n - n_owned;
  p.n_ghost_shards = (n_ghost + kShard - 1) / kShard;
  p.n_slots = (p.n_shards + p.n_ghost_shards) * kShard;
  p.iid.assign(n, -1);
  p.user_of.assign(p.n_slots, -1);
  p.shard_count.resize(p.n_shards);
  for (int s = 0; s < p.n_shards; ++s) {
    p.shard_count[s] = (int)shard_cells[s].size();
    for (int l = 0; l < (int)shard_cells[s].size(); ++l) {
      const int c = shard_cells[s][l];
      shard_of[c] = s;
      local_of[c] = l;
      p.iid[c] = s * kShard + l;
      p.user_of[s * kShard + l] = c;
    }
  }
  for (int g = 0; g < n_ghost; ++g) {  // ghost cells keep their (source rank, global id) order
    p.iid[n_owned + g] = p.n_shards * kShard + g;
    p.user_of[p.n_shards * kShard + g] = n_owned + g;
  }

  // ---- boundary faces in MeshWorker order
  std::vector<int32_t> bface_of((size_t)n * 4, -1);
  for (int c = 0; c < n_owned; ++c)
    for (int f = 0; f < 4; ++f) {
      const int nb = mesh.cell_face_neighbor[(size_t)c * 4 + f];
      if (nb < 0 && nb != DFLO_NBR_NONE) {
        const int id = DFLO_NBR_BOUNDARY_ID(nb);
        if (id < 0 || id >= DFLO_MAX_BOUNDARIES) { err = "boundary id out of range"; return DFLO_ERR_BAD_PARAM; }
        bface_of[(size_t)c * 4 + f] = (int32_t)p.bface_cell.size();
        p.bface_cell.push_back(c);
        p.bface_face.push_back(f);
        p.bface_id.push_back(id);
      }
    }

  // outward normal (unit) and length of face f of cell c: faces are straight edges, so both are
  // constant along the face (what FEFaceValues::normal_vector / JxW give for MappingQ1 / Cartesian)
  auto push_geom = [&](int c, int f) {
    static const int fv[4][2] = {{0, 2}, {1, 3}, {0, 1}, {2, 3}};
    const double *v = &V[(size_t)c * 8];
    const double tx = v[2 * fv[f][1]] - v[2 * fv[f][0]], ty = v[2 * fv[f][1] + 1] - v[2 * fv[f][0] + 1];
    const double len = std::sqrt(tx * tx + ty * ty);
    // t runs along increasing free coordinate; faces 1 (xi=1) and 2 (eta=0) have the cell on the
    // left of t, faces 0 and 3 on the right (counter-clockwise cells)
    const double nx = (f == 1 || f == 2) ? ty : -ty, ny = (f == 1 || f == 2) ? -tx : tx;
    p.face_geom.push_back(nx / len);
    p.face_geom.push_back(ny / len);
    p.face_geom.push_back(len);
  };

  // ---- per-shard halo and face lists
  p.halo_begin.assign(p.n_shards + 1, 0);
  p.face_begin.assign(p.n_shards + 1, 0);
  p.cell_face.assign((size_t)(p.n_shards + 2) * 4 * kShard, kNoFace);  // +2: the stage kernel reads two shards ahead
  p.lrbt.assign((size_t)p.n_shards * 4 * kShard, -1);
  p.nbr_code.assign((size_t)p.n_shards * 4 * kShard, 0);
  p.max_halo = p.max_faces = p.max_bnd = 0;
  p.shard_bnd.assign(p.n_shards, 0);
  std::unordered_map<int64_t, int32_t> halo_slot;  // (cell, its face) -> halo entry
  for (int s = 0; s < p.n_shards; ++s) {
    halo_slot.clear();
    const auto &cells = shard_cells[s];
    const int face0 = (int)p.faces.size();
    auto slot_of = [&](int c, int cface) -> int {  // cface: face of c on which the trace is needed
      if (shard_of[c] == s) return local_of[c];
      const int64_t key = (int64_t)c * 4 + cface;
      auto it = halo_slot.find(key);
      if (it != halo_slot.end()) return it->second;
      const int sl = kShard + (int)halo_slot.size();
      halo_slot.emplace(key, sl);
      p.halo_cells.push_back(p.iid[c]);
      p.halo_faces.push_back(cface);
      return sl;
    };
    for (int l = 0; l < (int)cells.size(); ++l) {
      const int c = cells[l];
      for (int f = 0; f < 4; ++f) {
        const int nb = mesh.cell_face_neighbor[(size_t)c * 4 + f];
        const int code = mesh.cell_face_neighbor_face[(size_t)c * 4 + f];
        const size_t ref = ((size_t)s * 4 + f) * kShard + l;
        if (nb == DFLO_NBR_NONE) continue;
        if (nb < 0) {
          const int k = (int)p.faces.size() - face0;
          const int bl = p.shard_bnd[s]++;
          if (bl >= 0x3FF) { err = "too many boundary faces in a shard"; return DFLO_ERR_BAD_PARAM; }
          p.faces.push_back({(uint32_t)l | ((uint32_t)f << 16) | (1u << 18) | ((uint32_t)bl << 20), bface_of[(size_t)c * 4 + f]});
          push_geom(c, f);
          p.cell_face[ref] = (uint16_t)k;
          continue;
        }
        if (nb >= n) { err = "neighbour index out of range"; return DFLO_ERR_BAD_PARAM; }
        const int nf = code & 3;
        const bool flip = (code & 4) != 0;
        p.lrbt[ref] = p.iid[nb];  // cartesian meshes: faces 0..3 are the left/right/bottom/top neighbours
        p.nbr_code[ref] = (uint8_t)((code & 7) | ((code & 8) ? 0 : 8));
        const bool integrator = gid(c) < gid(nb) || (gid(c) == gid(nb) && f < nf);
        const bool nb_inside = shard_of[nb] == s;
        if (nb_inside && !integrator) continue;  // the integrating cell creates the record and both references
        const int k = (int)p.faces.size() - face0;
        if (k >= 0x3FFF) { err = "too many faces in a shard"; return DFLO_ERR_BAD_PARAM; }
        if (integrator) {
          const int so = slot_of(nb, nf);
          p.faces.push_back({(uint32_t)l | ((uint32_t)f << 16) | ((uint32_t)flip << 19) | ((uint32_t)nf << 20), so});
          push_geom(c, f);
          // a face with a halo side keeps its flux at the point index of the HALO cell (it overwrites that cell's trace, see
          // the stage kernel), so the reference carries the flip on the integrating side as well
          p.cell_face[ref] = (uint16_t)(k | ((!nb_inside && flip) ? 0x4000 : 0));
          if (nb_inside)
            p.cell_face[((size_t)s * 4 + nf) * kShard + local_of[nb]] = (uint16_t)(k | (flip ? 0x4000 : 0) | 0x8000);
        } else {  // neighbour outside the shard integrates; we still evaluate its flux
          const int so = slot_of(nb, nf);
          p.faces.push_back({(uint32_t)so | ((uint32_t)nf << 16) | ((uint32_t)flip << 19) | ((uint32_t)f << 20), l});
          push_geom(nb, nf);
          p.cell_face[ref] = (uint16_t)(k | (flip ? 0x4000 : 0) | 0x8000);
        }
      }
    }
    {  // order the shard's faces: those with a halo cell on one side, then the interior ones, then the boundary
       // faces -- the flux phase maps consecutive lanes to consecutive faces, and a wavefront whose faces are all
       // of one kind takes one path through the trace code instead of both
      const int nf = (int)p.faces.size() - face0;
      std::vector<int> order(nf), pos(nf);
      for (int k = 0; k < nf; ++k) order[k] = k;
      auto kind = [&](int k) {
        const FaceRec &r = p.faces[face0 + k];
        if ((r.w0 >> 18) & 1) return 2;
        return ((r.w0 & 0xFFFF) >= (uint32_t)kShard || r.w1 >= kShard) ? 0 : 1;
      };
      // inside a kind: by the local face of the integrating cell, then by its slot -- neighbouring lanes of the flux phase
      // then read the same rows of the LDS image at different cells (different banks)
      auto key = [&](int k) {
        const FaceRec &r = p.faces[face0 + k];
        return (kind(k) << 20) | (int)(((r.w0 >> 16) & 3) << 16) | (int)(r.w0 & 0xFFFF);
      };
      std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return key(x) < key(y); });
      std::vector<FaceRec> fr(nf);
      std::vector<double> fg((size_t)nf * 3);
      for (int k = 0; k < nf; ++k) {
        pos[order[k]] = k;
        fr[k] = p.faces[face0 + order[k]];
        for (int j2 = 0; j2 < 3; ++j2) fg[(size_t)k * 3 + j2] = p.face_geom[((size_t)face0 + order[k]) * 3 + j2];
      }
      std::copy(fr.begin(), fr.end(), p.faces.begin() + face0);
      std::copy(fg.begin(), fg.end(), p.face_geom.begin() + (size_t)face0 * 3);
      for (int f = 0; f < 4; ++f)
        for (int l = 0; l < kShard; ++l) {
          uint16_t &ref = p.cell_face[((size_t)s * 4 + f) * kShard + l];
          if (ref != kNoFace) ref = (uint16_t)((ref & 0xC000) | pos[ref & 0x3FFF]);
        }
      // the halo entries in the order of their faces (which come first): entry e belongs to face e, and the flux of that face
      // can take the place of the entry's trace in the LDS table of the stage kernel
      const int h0 = p.halo_begin[s], nh = (int)halo_slot.size();
      std::vector<int32_t> hc(nh), hf(nh);
      for (int k = 0; k < nh; ++k) {
        FaceRec &r = p.faces[face0 + k];
        const bool halo_l = (r.w0 & 0xFFFF) >= (uint32_t)kShard;
        const int e_old = (halo_l ? (int)(r.w0 & 0xFFFF) : (int)r.w1) - kShard;
        if (kind(k) != 0 || e_old < 0 || e_old >= nh) { err = "internal: halo faces and halo entries do not pair up"; return DFLO_ERR_BAD_PARAM; }
        hc[k] = p.halo_cells[h0 + e_old];
        hf[k] = p.halo_faces[h0 + e_old];
        if (halo_l) r.w0 = (r.w0 & ~0xFFFFu) | (uint32_t)(kShard + k);
        else r.w1 = kShard + k;
      }
      if (nh < nf && kind(nh) == 0) { err = "internal: more halo faces than halo entries"; return DFLO_ERR_BAD_PARAM; }
      std::copy(hc.begin(), hc.end(), p.halo_cells.begin() + h0);
      std::copy(hf.begin(), hf.end(), p.halo_faces.begin() + h0);
      p.max_inner = std::max(p.max_inner, nf - nh);
    }
    p.halo_begin[s + 1] = (int)p.halo_cells.size();
    p.face_begin[s + 1] = (int)p.faces.size();
    p.max_halo = std::max(p.max_halo, p.halo_begin[s + 1] - p.halo_begin[s]);
    p.max_faces = std::max(p.max_faces, p.face_begin[s + 1] - p.face_begin[s]);
    p.max_bnd = std::max(p.max_bnd, p.shard_bnd[s]);
    bool reads_ghost = false;
    for (int k = p.halo_begin[s]; k < p.halo_begin[s + 1]; ++k) reads_ghost |= p.halo_cells[k] >= p.n_shards * kShard;
    (reads_ghost ? p.rim_shards : p.interior_shards).push_back(s);
  }

  // ---- the stage kernel keeps traces and fluxes in one LDS table of halo_cols + max_inner columns: a face with a halo side in
  //      the column of its halo entry, the other faces behind them.  The cells refer to their faces by column.
  p.halo_cols = std::max(p.max_halo, 1);
  for (int s = 0; s < p.n_shards; ++s) {
    const int nh = p.halo_begin[s + 1] - p.halo_begin[s];
    for (int f = 0; f < 4; ++f)
      for (int l = 0; l < kShard; ++l) {
        uint16_t &ref = p.cell_face[((size_t)s * 4 + f) * kShard + l];
        if (ref == kNoFace) continue;
        const int k = ref & 0x3FFF, col = k < nh ? k : k - nh + p.halo_cols;
        if (col >= 0x3FFF) { err = "too many faces in a shard"; return DFLO_ERR_BAD_PARAM; }
        ref = (uint16_t)((ref & 0xC000) | col);
      }
  }

  // ---- ghost traces: number the (ghost cell, face) pairs the owned cells look at, ghost cell first, then face
  {
    std::vector<int64_t> keys;
    for (size_t k = 0; k < p.halo_cells.size(); ++k)
      if (p.halo_cells[k] >= p.n_shards * kShard) keys.push_back((int64_t)p.halo_cells[k] * 4 + p.halo_faces[k]);
    std::sort(keys.begin(), keys.end());
    keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
    p.gt_cell.resize(keys.size());
    p.gt_face.resize(keys.size());
    for (size_t t = 0; t < keys.size(); ++t) {
      p.gt_cell[t] = (int32_t)(keys[t] / 4);
      p.gt_face[t] = (int32_t)(keys[t] % 4);
    }
    p.halo_gt.assign(p.halo_cells.size(), -1);
    for (size_t k = 0; k < p.halo_cells.size(); ++k)
      if (p.halo_cells[k] >= p.n_shards * kShard) {
        const int64_t key = (int64_t)p.halo_cells[k] * 4 + p.halo_faces[k];
        p.halo_gt[k] = (int32_t)(std::lower_bound(keys.begin(), keys.end(), key) - keys.begin());
      }
  }

  // ---- the rim widened by one ring of shards (multi-device runs with the TVB limiter): the limiter of a rim cell reads the
  //      new averages of its face neighbours, and those that are not ghosts live in rim shards or in the ring
  {
    std::vector<char> is_rim(std::max(p.n_shards, 1), 0);
    for (int s : p.rim_shards) is_rim[s] = 1;
    p.rim2_shards = p.rim_shards;
    for (int s : p.interior_shards) {
      bool ring = false;
      for (int k = p.halo_begin[s]; k < p.halo_begin[s + 1] && !ring; ++k) {
        const int hs = p.halo_cells[k] / kShard;
        ring = hs < p.n_shards && is_rim[hs];
      }
      (ring ? p.rim2_shards : p.rest2_shards).push_back(s);
    }
  }

  // ---- the ghost cells' neighbours, for the limiter pass over the ghost shards (see plan.h)
  p.ghost_count.assign(p.n_ghost_shards, 0);
  p.ghost_lrbt.assign((size_t)p.n_ghost_shards * 4 * kShard, -1);
  for (int g = 0; g < n_ghost; ++g) {
    const int c = n_owned + g, gs = g / kShard, l = g % kShard;
    ++p.ghost_count[gs];
    for (int f = 0; f < 4; ++f) {
      const int nb = mesh.cell_face_neighbor[(size_t)c * 4 + f];
      int32_t v = -1;                                            // a physical boundary: the limiter takes the cell's own slope
      if (nb == DFLO_NBR_NONE) v = p.n_slots + 4 * g + f;        // a cell of the ghost's owner (or another ghost): from the record
      else if (nb >= 0) v = nb < n_owned ? p.iid[nb] : p.n_slots + 4 * g + f;
      p.ghost_lrbt[((size_t)gs * 4 + f) * kShard + l] = v;
    }
  }

  // ---- geometry in internal order
  if (mesh.mapping == DFLO_MAP_CARTESIAN) {
    p.cell_h.assign(p.n_slots + 2 * kShard, p.h);
    for (int c = 0; c < n; ++c) p.cell_h[p.iid[c]] = hx[c];
  } else {
    p.cell_h.assign(p.n_slots + 2 * kShard, 1.0);
    for (int c = 0; c < n; ++c) {  // h = diameter / sqrt(2) (src/claw.cc:551), used by compute_time_step_q
      const double *v = &V[(size_t)c * 8];
      const double d1 = std::hypot(v[6] - v[0], v[7] - v[1]), d2 = std::hypot(v[4] - v[2], v[5] - v[3]);
      p.cell_h[p.iid[c]] = std::max(d1, d2) / std::sqrt(2.0);
    }
    p.cell_vert.assign((size_t)8 * p.n_slots, 0.0);
    // padding slots get a unit square so that metric terms stay finite
    for (int sl = 0; sl < p.n_slots; ++sl) {
      const double unit[8] = {0, 0, 1, 0, 0, 1, 1, 1};
      for (int k = 0; k < 8; ++k) p.cell_vert[(size_t)k * p.n_slots + sl] = unit[k];
    }
    for (int c = 0; c < n; ++c)
      for (int k = 0; k < 8; ++k) p.cell_vert[(size_t)k * p.n_slots + p.iid[c]] = V[(size_t)c * 8 + k];
  }
  return DFLO_OK;
}

}  // namespace dflo

// Test hook (dflo_hip_diag.h; host only, no device): the shard plan's view of the ghost cells' neighbours -- what the limiter pass over
// the ghost shards of a one-exchange TVB stage reads (Plan::ghost_lrbt) --, in the mesh's own cell numbering.
extern "C" int dflo_hip_plan_ghost_neighbours(const dflo_mesh_t *part_mesh, int32_t *table) {
  if (!part_mesh || !table) return DFLO_ERR_BAD_PARAM;
  dflo::Plan p;
  std::string err;
  const int rc = dflo::build_plan(*part_mesh, 8, 8, p, err);
  if (rc) return rc;
  const int n_ghost = p.n_cells - p.n_owned;
  for (int g = 0; g < n_ghost; ++g)
    for (int f = 0; f < 4; ++f) {
      const int32_t v = p.ghost_lrbt[((size_t)(g / dflo::kShard) * 4 + f) * dflo::kShard + g % dflo::kShard];
      table[(size_t)g * 4 + f] = v < 0 ? -1 : (v >= p.n_slots ? (v == p.n_slots + 4 * g + f ? -2 : -3) : p.user_of[v]);
    }
  return DFLO_OK;
}

