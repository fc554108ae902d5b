/*
 * dflo_mesh.h -- host-side construction and partition of the flat mesh description dflo_mesh_t (dflo_hip.h).
 * What GridIn::read_msh + Triangulation (+ parallel::distributed::Triangulation) hand to dflo, flattened: used by the stand-alone
 * driver dflo_hip_run, the Python mirror and the tests.  A dflo build fills dflo_mesh_t from its own Triangulation
 * (include/dflo_hip_dealii.hpp) and does not need this header.
 */
#ifndef DFLO_MESH_H
#define DFLO_MESH_H

#include "dflo_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------- host-side mesh construction */
/* What GridIn::read_msh + Triangulation hand to dflo (src/claw.cc:957-967),
 * flattened.  The returned mesh owns its arrays; free with dflo_mesh_free. */

/* nx x ny squares on [x0,x0+nx*h] x [y0,y0+ny*h], cell c = i + nx*j.
 * side_bc[4] = boundary id on the faces x=min, x=max, y=min, y=max, or -1 for
 * a periodic side (src_mpi semantics, src_mpi/assemble_explicit.cc:186-260). */
int dflo_mesh_cartesian(int32_t nx, int32_t ny, double x0, double y0, double h, const int32_t side_bc[4],
                        int32_t degree, dflo_mesh_t **out);
/* General conforming quad mesh: vertices [n_vertices][2], quads [n_quads][4]
 * (any consistent vertex order; re-ordered to deal.II's), boundary edges
 * [n_bedges][2] vertex pairs with ids.  mapping = DFLO_MAP_Q1. */
int dflo_mesh_from_quads(int32_t n_vertices, const double *vertices, int32_t n_quads, const int32_t *quads,
                         int32_t n_bedges, const int32_t *bedges, const int32_t *bedge_id, int32_t degree,
                         dflo_mesh_t **out);
/* Gmsh v2 ASCII .msh with quads + physical lines (what "gmsh -2 file.geo" writes, README.md:70-72). */
int dflo_mesh_read_gmsh(const char *path, int32_t degree, int32_t mapping, dflo_mesh_t **out);
/* Pair the boundary faces with ids id_first / id_second, offset along direction (0 = x, 1 = y), into periodic
 * neighbours in place: GridTools::collect_periodic_faces + add_periodicity for the "type = periodic", "pair",
 * "direction" entries of a boundary subsection (src_mpi/parameters.cc:397-410, src_mpi/claw.cc:156-200). */
int dflo_mesh_make_periodic(dflo_mesh_t *mesh, int32_t id_first, int32_t id_second, int32_t direction);
/* Sub-mesh of rank `rank` of `n_ranks` (contiguous slabs of the cell order after a
 * coordinate sort) with one layer of face-neighbour ghost cells -- the flat
 * equivalent of parallel::distributed::Triangulation's owned+ghost view
 * (src_mpi/claw.h:220).  send_cells/send_offsets (size n_ranks+1) list owned cells
 * to send per destination rank; recv_offsets the ghost ranges per source rank
 * (ghost cells are ordered by source rank). Arrays owned by the mesh. */
int dflo_mesh_partition(const dflo_mesh_t *mesh, int32_t n_ranks, int32_t rank, dflo_mesh_t **out,
                        const int32_t **send_cells, const int32_t **send_offsets, const int32_t **recv_offsets);
/* The same with a choice of partitioner (dflo_partitioner): DFLO_PART_SLAB as above (C4: x-slabs of the 4001 x 1000
 * lattice), DFLO_PART_RCB recursive coordinate bisection of the cell centres (compact blocks on unstructured meshes, C5;
 * the MPI variant gets Morton-order blocks from p4est, src_mpi/claw.h:220).  partition_owners writes the owner rank of
 * every cell ([n_cells]) without building a sub-mesh. */
int dflo_mesh_partition_ex(const dflo_mesh_t *mesh, int32_t n_ranks, int32_t rank, int32_t method, dflo_mesh_t **out,
                           const int32_t **send_cells, const int32_t **send_offsets, const int32_t **recv_offsets);
int dflo_mesh_partition_owners(const dflo_mesh_t *mesh, int32_t n_ranks, int32_t method, int32_t *owner_out);
/* Self-halo partition: ONE part that owns every cell and is its own neighbour across a virtual cut (n_virtual >= 2: the
 * non-periodic faces between the cells of different virtual owners of dflo_mesh_partition_owners; n_virtual == 1: the
 * periodic faces in x).  Every cell on the cut gets a ghost copy; send list = those cells, offsets for the one "peer" 0.
 * A measuring device (dflo_hip_multi_create_self): one full-size part runs the whole schedule that replaces
 * update_ghost_values / Utilities::MPI::min (src_mpi/claw.cc:793, 579) with itself as the neighbour. */
int dflo_mesh_partition_self(const dflo_mesh_t *mesh, int32_t n_virtual, int32_t method, dflo_mesh_t **out,
                             const int32_t **send_cells, const int32_t **send_offsets, const int32_t **recv_offsets);
void dflo_mesh_free(dflo_mesh_t *mesh);
const char *dflo_mesh_last_error(void);

/* Initial condition by nodal interpolation for Qk (VectorTools::interpolate, src/ic.cc:104-121):
 * xy [n_cells][n_s][2] = support point coordinates in dflo's DoF order. */
int dflo_mesh_support_points(const dflo_mesh_t *mesh, double *xy);

#ifdef __cplusplus
}
#endif
#endif /* DFLO_MESH_H */
