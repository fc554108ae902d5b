// plan.h -- the CSR-like element/face index layout of the device engine.
//
// deal.II's DoFHandler + MeshWorker::loop (src/assemble_explicit.cc:440-451) are flattened once
// into "shards": groups of up to kShard cells stored contiguously and structure-of-arrays on the
// device (one wavefront lane per cell).  For every shard the plan lists
//   - its halo cells (face neighbours living in other shards or in the ghost range),
//   - its faces, each integrated exactly once from the cell with the smaller global id as
//     MeshWorker does; faces on the shard rim are listed by both shards with the SAME
//     integrating side, so both evaluate a bit-identical flux,
//   - for each cell and local face the slot of that face's numerical flux and its sign.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "abi.h"

namespace dflo {

constexpr int kShard = 64;  // cells per shard = lanes per wavefront

// face record, 8 bytes
//   w0: bits 0-15 slot of the integrating cell in the shard's LDS image (own cells first, then halo)
//       bits 16-17 local face number seen from the integrating cell
//       bit  18    1 = boundary face
//       bit  19    1 = face points run opposite on the other side
//       bits 20-21 local face number seen from the other cell (interior faces)
//       bits 20-29 index of the face among the shard's boundary faces (boundary faces)
//   w1: slot of the other cell, or the boundary-face index
struct FaceRec {
  uint32_t w0;
  int32_t w1;
};

// per (cell, local face) reference, 16 bit:
//   bits 0-13 face index inside the shard, bit 14 flip, bit 15: 1 = this cell is the "other" side
//   (receives +flux), 0xFFFF = no face (ghost rim)
constexpr uint16_t kNoFace = 0xFFFF;

struct Plan {
  int n_cells = 0, n_owned = 0;      // user numbering
  int n_shards = 0;                  // shards of owned cells
  int n_ghost_shards = 0;            // shards holding ghost cells (never integrated)
  int n_slots = 0;                   // (n_shards + n_ghost_shards) * kShard internal cell slots
  std::vector<int32_t> iid;          // user cell -> internal slot
  std::vector<int32_t> user_of;      // internal slot -> user cell or -1 (padding)
  std::vector<int32_t> shard_count;  // cells per owned shard
  std::vector<int32_t> halo_begin;   // [n_shards+1]
  std::vector<int32_t> halo_cells;   // internal slot of the halo entry's cell
  std::vector<int32_t> halo_faces;   // its local face shared with the shard (an entry = one (cell, face) pair)
  std::vector<int32_t> halo_gt;      // ghost-trace number of the entry, or -1: the entry's cell is an owned cell
  // Ghost cells are known to the stage kernels only by their traces on the faces they share with owned cells (the halo
  // records of a multi-device run: N*4 doubles per cut face): trace number = position in the list of
  // (ghost cell, face) pairs ordered by ghost cell (source rank, global id), then face -- the order the owners pack them in
  std::vector<int32_t> gt_cell;      // internal slot of the ghost cell of trace number t
  std::vector<int32_t> gt_face;
  std::vector<int32_t> face_begin;   // [n_shards+1]
  std::vector<FaceRec> faces;
  std::vector<double> face_geom;     // [n faces][3]: outward unit normal of the integrating cell, edge length
  std::vector<uint16_t> cell_face;   // [n_shards][4][kShard]: bits 0-13 the COLUMN of the face in the stage kernel's flux table
                                     // (halo entry e -> e, the k-th other face -> halo_cols + k), 14 flip, 15 the other side integrates
  std::vector<int32_t> lrbt;         // [n_shards][4][kShard] internal slot of the left/right/bottom/top
                                     // neighbour or -1 (src/claw.cc:336-380)
  std::vector<uint8_t> nbr_code;     // [n_shards][4][kShard] bits 0-1 the neighbour's local face, bit 2 flip,
                                     // bit 3: interior non-periodic face (the ones the KXRCF indicator visits)
  std::vector<int32_t> shard_bnd;    // boundary faces per shard
  std::vector<int32_t> rim_shards;   // owned shards that read ghost cells (multi-device: computed first, then
  std::vector<int32_t> interior_shards;  // their cells are exchanged while the interior shards are computed)
  std::vector<int32_t> rim2_shards;  // rim shards + the ring of shards next to them, and the rest: the split of the UPDATE when a
  std::vector<int32_t> rest2_shards; // TVB limiter follows (the limiter of the rim cells reads averages from the ring)
  // The ghost shards as a limiter pass sees them (multi-device TVB runs with ONE exchange per stage: the ghost cells arrive
  // unlimited and are limited here as their owners limit them): cells per ghost shard, and for every (ghost cell, face) the
  // slot of the neighbour when it is an owned cell of this part, -1 at a physical boundary, n_slots + 4 g + f where the
  // neighbour lives with the ghost's owner (its average comes with the ghost's record)
  std::vector<int32_t> ghost_count;  // [n_ghost_shards]
  std::vector<int32_t> ghost_lrbt;   // [n_ghost_shards][4][kShard]
  int max_halo = 0, max_faces = 0, max_bnd = 0;
  int max_inner = 0;                 // most faces of a shard without a halo side
  int halo_cols = 1;                 // columns of the stage kernel's trace / flux table that belong to halo entries
  // boundary faces in MeshWorker order (cell ascending, face ascending)
  std::vector<int32_t> bface_cell, bface_face, bface_id;
  bool uniform_h = false;
  double h = 0.0;
  std::vector<double> cell_h;        // [n_slots] (cartesian)
  std::vector<double> cell_vert;     // [8][n_slots] (q1 mapping), component-major
};

// Returns DFLO_OK or an error code with a message.
// h_hint > 0: the cell size to give a lattice of squares (Plan::h).  By itself the plan takes the smallest x-extent of ITS cells;
// the extents of a lattice whose coordinates are not dyadic differ in the last bit from cell to cell, so the plan of a part of a
// mesh may find another value than the plan of the whole mesh.  The multi-device driver hands the whole mesh's value to the plans
// of its parts (create_engine below): a multi-part run then carries the bits of the single engine on any lattice.
int build_plan(const dflo_mesh_t &mesh, int shard_ex, int shard_ey, Plan &plan, std::string &err, double h_hint = 0.0);

}  // namespace dflo

// dflo_hip_create with the cell size of the undivided mesh (engine.hip; the C ABI entry passes 0: the mesh's own)
struct dflo_hip_engine;
int dflo_hip_create_with_cell_size(const dflo_mesh_t *mesh, const dflo_params_t *params, int device_id, dflo_hip_engine **out, double h_hint,
                                    bool pass_takes_exchange = false);   // (the driver means to use dflo_hip_limit_exchange: marks at every degree)
