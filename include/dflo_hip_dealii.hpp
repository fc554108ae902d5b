// dflo_hip_dealii.hpp -- header-only adaptor between dflo's deal.II objects and the C ABI of include/dflo_hip.h.
//
// STATUS: written against the reference sources, NOT COMPILED in this repository -- the build image has no deal.II (and
// writing stand-in headers for it is not allowed), so nothing here is covered by the test suite.  It is the file a dflo
// maintainer adds next to src/claw.h; INTEGRATION.md walks through the three places it is called from.  Everything it
// does on the other side of the ABI (flat mesh, DoF order, boundary tables, step loop) is what the tests drive through
// dflo_amd/_lib.py and dflo_amd/csrc/dflo_run.cc.
//
// What it replaces in ConservationLaw<2> (paths relative to the dflo repository):
//   setup_system() tail            src/claw.cc:271-386   -> dflo_hip::Adaptor::attach   (Triangulation + DoFHandler -> dflo_mesh_t,
//                                                           Parameters::AllParameters -> dflo_params_t)
//   iterate_explicit()             src/claw.cc:726-772   -> Adaptor::iterate_explicit  (all RK stages of one step)
//   compute_time_step()            src/claw.cc:444-557   -> Adaptor::compute_time_step
//   run(): the explicit branch     src/claw.cc:1026-1110 -> Adaptor::advance / the loop shown in INTEGRATION.md section 2
//   cell_number()                  src/claw.h:331-334    -> used as is (user_index set in setup_system, src/claw.cc:293-297)
//
// Conventions relied upon: reference-cell vertices lexicographic, faces 0: x=0, 1: x=1, 2: y=0,
// 3: y=1; FESystem of one DG element x 4 numbered cell by cell, dof = cell * dofs_per_cell + component * n_s + node
// (no renumbering, src/claw.cc:273); 2-D meshes orientation-consistent (both cells walk a shared edge the same way).
// A build that renumbers DoFs sets Adaptor::use_dof_indices = true: the state then travels through get_dof_indices.
#pragma once

#include <deal.II/base/exceptions.h>
#include <deal.II/base/function_parser.h>
#include <deal.II/base/point.h>
#include <deal.II/dofs/dof_handler.h>
#include <deal.II/fe/fe.h>
#include <deal.II/grid/tria.h>
#include <deal.II/lac/vector.h>

#include <cstdint>
#include <string>
#include <vector>

#include "dflo_hip.h"

namespace dflo_hip {

// Parameters::AllParameters<2> is dflo's own type (src/parameters.h:363-411): the adaptor is a template on it so that this
// header depends on deal.II and dflo_hip.h only.
template <class AllParameters>
inline dflo_params_t make_params(const AllParameters &prm) {
  dflo_params_t p{};
  p.flux_type = static_cast<int32_t>(prm.flux_type);            // Flux::FluxType {lxf, sw, kfvs, roe, hllc}, src/parameters.h:229 == dflo_flux
  p.limiter_type = static_cast<int32_t>(prm.limiter_type);      // Limiter::LimiterType {none, TVB}, :244 == dflo_limiter
  p.shock_indicator = static_cast<int32_t>(prm.shock_indicator_type);   // {limiter, density, energy, u2}, :245 == dflo_shock_indicator
  p.char_lim = prm.char_lim ? 1 : 0;
  p.pos_lim = prm.pos_lim ? 1 : 0;
  p.conserve_angular_momentum = prm.conserve_angular_momentum ? 1 : 0;
  p.M = prm.M;
  p.beta = prm.beta;
  p.global_time_step = prm.time_step_type == "global" ? 1 : 0;  // src/parameters.cc:452
  p.n_rk = 0;                                                   // by degree, src/claw.cc:141-159
  p.gravity = prm.gravity;
  p.cfl = prm.cfl;
  p.time_step = prm.time_step;
  p.final_time = prm.final_time;
  for (unsigned int b = 0; b < AllParameters::max_n_boundaries && b < DFLO_MAX_BOUNDARIES; ++b)
    p.bc_kind[b] = static_cast<int32_t>(prm.boundary_conditions[b].kind);   // EulerEquations::BoundaryKind, src/equation.h:862-869 == dflo_bc_kind
  return p;
}

class Adaptor {
 public:
  bool use_dof_indices = false;   // true: permute the state through cell->get_dof_indices (a build that renumbers DoFs)

  Adaptor() = default;
  Adaptor(const Adaptor &) = delete;
  Adaptor &operator=(const Adaptor &) = delete;
  ~Adaptor() { detach(); }

  // End of setup_system(): flatten Triangulation + DoFHandler (active cells in user_index order) and create the engine(s).
  //   devices.size() == 1: one engine (dflo_hip_create);  > 1: the native multi-device driver, one part per listed device
  //   (dflo_hip_multi_create; src/ is one process, so this is the one-process form -- src_mpi/ uses attach_rank below).
  // The reference's AssertThrow convention is kept: a refused configuration throws with the engine's message.
  template <class AllParameters>
  void attach(const dealii::DoFHandler<2> &dof_handler, const AllParameters &prm, const std::vector<int> &devices = {0},
              int partitioner = DFLO_PART_SLAB) {
    flatten(dof_handler, prm);
    const dflo_params_t p = make_params(prm);
    if (devices.size() <= 1) {
      const int rc = dflo_hip_create(&mesh_, &p, devices.empty() ? 0 : devices[0], &one_);
      AssertThrow(rc == DFLO_OK, dealii::ExcMessage(std::string("dflo_hip_create: ") + dflo_hip_last_error(nullptr)));
    } else {
      const int rc = dflo_hip_multi_create(&mesh_, &p, static_cast<int>(devices.size()), devices.data(), partitioner, &multi_);
      AssertThrow(rc == DFLO_OK, dealii::ExcMessage(std::string("dflo_hip_multi_create: ") + dflo_hip_multi_last_error(nullptr)));
    }
    read_boundary_faces();
  }

  // src_mpi/: one rank per GPU.  Every rank flattens the UNDIVIDED mesh (as GridIn reads it before
  // parallel::distributed::Triangulation partitions it, src_mpi/claw.cc:116-200); unique_id: DFLO_COMM_ID_BYTES bytes from
  // dflo_hip_comm_unique_id() on rank 0, broadcast with MPI_Bcast.
  template <class AllParameters>
  void attach_rank(const dealii::DoFHandler<2> &undivided_dof_handler, const AllParameters &prm, int device, int rank, int n_ranks,
                   const void *unique_id, int partitioner = DFLO_PART_SLAB) {
    flatten(undivided_dof_handler, prm);
    const dflo_params_t p = make_params(prm);
    const int rc = dflo_hip_multi_create_rank(&mesh_, &p, device, rank, n_ranks, unique_id, partitioner, &multi_);
    AssertThrow(rc == DFLO_OK, dealii::ExcMessage(std::string("dflo_hip_multi_create_rank: ") + dflo_hip_multi_last_error(nullptr)));
    read_boundary_faces();
  }

  void detach() {
    if (one_) dflo_hip_destroy(one_);
    if (multi_) dflo_hip_multi_destroy(multi_);
    one_ = nullptr;
    multi_ = nullptr;
  }

  bool attached() const { return one_ || multi_; }

  // current_solution / old_solution -> device (after set_initial_condition(), src/claw.cc:982-1003; after a refinement)
  void set_solution(const dealii::Vector<double> &u) {
    check(multi_ ? dflo_hip_multi_set_solution(multi_, to_abi(u)) : dflo_hip_set_solution(one_, to_abi(u)), "set_solution");
  }
  // device -> current_solution (at the output cadence, src/claw.cc:1093-1099; before anything on the host reads the state)
  void get_solution(dealii::Vector<double> &u) {
    double *dst = use_dof_indices ? (scratch_.resize(u.size()), scratch_.data()) : u.begin();
    check(multi_ ? dflo_hip_multi_get_solution(multi_, dst) : dflo_hip_get_solution(one_, dst), "get_solution");
    if (use_dof_indices)
      for (std::size_t i = 0; i < perm_.size(); ++i) u[perm_[i]] = scratch_[i];
  }
  // cell_average (src/claw.cc:562-597) of the device's current state, [n_cells][4]
  void get_cell_average(std::vector<dealii::Vector<double>> &cell_average) {
    avg_.resize(static_cast<std::size_t>(mesh_.n_cells) * 4);
    check(multi_ ? dflo_hip_multi_get_cell_average(multi_, avg_.data()) : dflo_hip_get_cell_average(one_, avg_.data()), "get_cell_average");
    for (std::size_t c = 0; c < cell_average.size(); ++c)
      for (unsigned int k = 0; k < 4; ++k) cell_average[c][k] = avg_[c * 4 + k];
  }

  // run(): compute_shock_indicator(); apply_limiter();  on the initial condition (src/claw.cc:997-1001)
  void apply_limiter() { check(multi_ ? dflo_hip_multi_apply_limiter(multi_) : dflo_hip_apply_limiter(one_), "apply_limiter"); }
  void apply_positivity_limiter() {
    check(multi_ ? dflo_hip_multi_apply_positivity_limiter(multi_) : dflo_hip_apply_positivity_limiter(one_), "apply_positivity_limiter");
  }

  // Boundary functions.  Either the values at the face quadrature points for bc_time = t (table 0, RK stage 0) and t + dt
  // (table 1, later stages), as integrate_boundary_term_explicit evaluates them (src/assemble_explicit.cc:161-165,
  // src/claw.cc:733-745) ...
  template <class AllParameters>
  void set_boundary_values(AllParameters &prm, double time, int table) {
    const int N = mesh_.degree + 1;
    bval_.resize(bid_.size() * N * 4);
    dealii::Vector<double> v(4);
    for (std::size_t b = 0; b < bid_.size(); ++b) {
      dealii::FunctionParser<2> &fp = prm.boundary_conditions[bid_[b]].values;
      fp.set_time(time);
      for (int q = 0; q < N; ++q) {
        fp.vector_value(dealii::Point<2>(bxy_[(b * N + q) * 2], bxy_[(b * N + q) * 2 + 1]), v);
        for (int c = 0; c < 4; ++c) bval_[(b * N + q) * 4 + c] = v[c];
      }
    }
    if (bid_.empty()) return;
    check(multi_ ? dflo_hip_multi_set_boundary_values(multi_, table, bval_.data()) : dflo_hip_set_boundary_values(one_, table, bval_.data()),
          "set_boundary_values");
  }
  // ... or, for functions of time (double Mach reflection), the expression itself as a postfix program the device evaluates at
  // t and t + dt of every step (dflo_expr_op; dflo_amd/csrc/frontend.cc holds a compiler for the FunctionParser syntax).
  void set_boundary_program(int boundary_id, int component, const std::vector<int32_t> &ops, const std::vector<double> &consts) {
    const int32_t n_ops = static_cast<int32_t>(ops.size() / 2), n_c = static_cast<int32_t>(consts.size());
    check(multi_ ? dflo_hip_multi_set_boundary_program(multi_, boundary_id, component, n_ops, ops.data(), n_c, consts.data())
                 : dflo_hip_set_boundary_program(one_, boundary_id, component, n_ops, ops.data(), n_c, consts.data()),
          "set_boundary_program");
  }

  // compute_time_step() (src/claw.cc:444-557): global_dt of the device's current state, rules of :468-476 applied
  double compute_time_step(double elapsed_time) {
    double dt = 0.0;
    check(multi_ ? dflo_hip_multi_compute_dt(multi_, elapsed_time, &dt) : dflo_hip_compute_dt(one_, elapsed_time, &dt), "compute_time_step");
    return dt;
  }

  // iterate_explicit() (src/claw.cc:726-772): all RK stages -- assemble_system, solve, the two vector updates,
  // compute_cell_average, compute_shock_indicator, apply_limiter, apply_positivity_limiter -- and old_solution =
  // current_solution (:1110).  With time-dependent boundary functions call set_boundary_values(prm, t, 0) and
  // (prm, t + dt, 1) first, or hand the functions over once with set_boundary_program.  Returns like the reference
  // stops: AssertThrow on "Fatal: Negative states" (src/positivity.cc:28-38) and on the positivity root failure
  // (exit(0) there, :160-169; an error here).
  void iterate_explicit(double global_dt, double &res_norm0, double &res_norm) {
    check(multi_ ? dflo_hip_multi_step(multi_, global_dt, &res_norm0, &res_norm) : dflo_hip_step(one_, global_dt, &res_norm0, &res_norm),
          "iterate_explicit");
  }

  // n_steps x { compute_time_step; iterate_explicit; elapsed_time += global_dt } with the time step and the clock resident
  // on the device (no host round trip per step): what run() does between two outputs when nothing on the host looks at
  // the state (src/claw.cc:1026-1110).
  void advance(int n_steps, double &elapsed_time) {
    check(multi_ ? dflo_hip_multi_advance(multi_, n_steps, &elapsed_time) : dflo_hip_advance(one_, n_steps, &elapsed_time), "advance");
  }

  // parity hook: right_hand_side of the device's current state (assemble_system, src/assemble_explicit.cc:433-452)
  void assemble_system(dealii::Vector<double> &right_hand_side, int boundary_table = 0) {
    double *dst = use_dof_indices ? (scratch_.resize(right_hand_side.size()), scratch_.data()) : right_hand_side.begin();
    check(multi_ ? dflo_hip_multi_residual(multi_, boundary_table, dst) : dflo_hip_residual(one_, boundary_table, dst), "assemble_system");
    if (use_dof_indices)
      for (std::size_t i = 0; i < perm_.size(); ++i) right_hand_side[perm_[i]] = scratch_[i];
  }

  const dflo_mesh_t &flat_mesh() const { return mesh_; }
  dflo_hip_handle engine() const { return one_; }
  dflo_hip_multi_handle driver() const { return multi_; }

 private:
  template <class AllParameters>
  void flatten(const dealii::DoFHandler<2> &dof_handler, const AllParameters &prm) {
    detach();
    const auto &tria = dof_handler.get_triangulation();
    const unsigned int nc = tria.n_active_cells();
    const unsigned int dpc = dof_handler.get_fe().dofs_per_cell;
    vertices_.assign(static_cast<std::size_t>(nc) * 8, 0.0);
    nbr_.assign(static_cast<std::size_t>(nc) * 4, DFLO_NBR_NONE);
    nbr_face_.assign(static_cast<std::size_t>(nc) * 4, 0);
    perm_.clear();
    if (use_dof_indices) perm_.resize(static_cast<std::size_t>(nc) * dpc);
    std::vector<dealii::types::global_dof_index> dof_indices(dpc);
    for (auto cell = dof_handler.begin_active(); cell != dof_handler.end(); ++cell) {
      const unsigned int c = cell->user_index();                 // cell_number(), src/claw.h:331-334
      AssertThrow(c < nc, dealii::ExcMessage("dflo_hip: user_index is not the active cell number (set in setup_system, src/claw.cc:293-297)"));
      for (unsigned int v = 0; v < 4; ++v)
        for (unsigned int d = 0; d < 2; ++d) vertices_[c * 8 + v * 2 + d] = cell->vertex(v)[d];
      for (unsigned int f = 0; f < 4; ++f) {
        if (cell->at_boundary(f)) {
          nbr_[c * 4 + f] = DFLO_NBR_BOUNDARY(cell->face(f)->boundary_id());
        } else {
          // (no hanging nodes on this path: the engine takes conforming meshes; do_refine runs stay on the host)
          AssertThrow(cell->neighbor(f)->is_active() && !cell->face(f)->has_children(), dealii::ExcMessage("dflo_hip: the mesh must be conforming (no hanging nodes)"));
          nbr_[c * 4 + f] = static_cast<int32_t>(cell->neighbor(f)->user_index());
          nbr_face_[c * 4 + f] = static_cast<int32_t>(cell->neighbor_of_neighbor(f));
        }
      }
      if (use_dof_indices) {
        cell->get_dof_indices(dof_indices);
        // FESystem of one DG element x 4: system_to_component_index(i) = (component, node) as used in src/limiter.cc:361,414-415
        for (unsigned int i = 0; i < dpc; ++i) {
          const auto ci = dof_handler.get_fe().system_to_component_index(i);
          perm_[static_cast<std::size_t>(c) * dpc + ci.first * (dpc / 4) + ci.second] = dof_indices[i];
        }
      }
    }
    mesh_ = dflo_mesh_t{};
    mesh_.n_cells = mesh_.n_owned_cells = static_cast<int32_t>(nc);
    mesh_.degree = static_cast<int32_t>(dof_handler.get_fe().degree);
    mesh_.basis = static_cast<int32_t>(prm.basis) == 0 ? DFLO_BASIS_QK : DFLO_BASIS_PK;        // BasisType {Qk, Pk}, src/parameters.h:390-391
    // MappingType {q1, q2, cartesian}, src/parameters.h:392-393 == dflo_mapping.  q2 is taken as q1 by the engine: the flat mesh
    // carries four vertices per cell, i.e. straight edges, on which MappingQ(2) IS the bilinear map -- valid only while no
    // manifold / curved boundary description is attached to the triangulation (the reference's is commented out, src/claw.cc:976-979)
    mesh_.mapping = static_cast<int32_t>(prm.mapping_type);
    mesh_.cell_vertices = vertices_.data();
    mesh_.cell_face_neighbor = nbr_.data();
    mesh_.cell_face_neighbor_face = nbr_face_.data();
    mesh_.cell_global_id = nullptr;
  }

  void read_boundary_faces() {
    const int nb = multi_ ? dflo_hip_multi_n_boundary_faces(multi_) : dflo_hip_n_boundary_faces(one_);
    const int N = mesh_.degree + 1;
    bid_.assign(nb, 0);
    bxy_.assign(static_cast<std::size_t>(nb) * N * 2, 0.0);
    if (nb == 0) return;
    check(multi_ ? dflo_hip_multi_boundary_faces(multi_, nullptr, nullptr, bid_.data(), bxy_.data())
                 : dflo_hip_boundary_faces(one_, nullptr, nullptr, bid_.data(), bxy_.data()),
          "boundary_faces");
  }

  const double *to_abi(const dealii::Vector<double> &u) {
    if (!use_dof_indices) return u.begin();
    scratch_.resize(u.size());
    for (std::size_t i = 0; i < perm_.size(); ++i) scratch_[i] = u[perm_[i]];
    return scratch_.data();
  }

  void check(int rc, const char *what) {
    if (rc == DFLO_OK) return;
    const char *msg = multi_ ? dflo_hip_multi_last_error(multi_) : dflo_hip_last_error(one_);
    AssertThrow(false, dealii::ExcMessage(std::string("dflo_hip ") + what + ": " + msg));
  }

  dflo_hip_handle one_ = nullptr;
  dflo_hip_multi_handle multi_ = nullptr;
  dflo_mesh_t mesh_{};
  std::vector<double> vertices_, bxy_, bval_, avg_, scratch_;
  std::vector<int32_t> nbr_, nbr_face_, bid_;
  std::vector<dealii::types::global_dof_index> perm_;
};

}  // namespace dflo_hip

// ---------------------------------------------------------------------------------------------------------------------
// The three edits in dflo (src/), with the adaptor as a member `dflo_hip::Adaptor hip;` of ConservationLaw<dim> (dim == 2):
//
//   setup_system(), at its end (src/claw.cc:386):
//       if (!parameters.implicit) hip.attach(dof_handler, parameters);              // or {0,1,..,7} for the 8 GPUs of a node
//
//   run(), after the initial condition has been set and limited (src/claw.cc:982-1003):
//       hip.set_boundary_values(parameters, elapsed_time, 0);  hip.set_boundary_values(parameters, elapsed_time, 1);
//       hip.set_solution(current_solution);
//       // the host versions of compute_shock_indicator / apply_limiter on the initial condition may stay, or:
//       // hip.apply_limiter();
//
//   run(), the rk3 branch of the time loop (src/claw.cc:1029, :1047-1052, :1093-1110):
//       global_dt = hip.compute_time_step(elapsed_time);                            // compute_time_step()
//       hip.set_boundary_values(parameters, elapsed_time, 0);                       // only if the functions depend on t
//       hip.set_boundary_values(parameters, elapsed_time + global_dt, 1);
//       hip.iterate_explicit(global_dt, res_norm0, res_norm);                       // IntegratorExplicit + iterate_explicit
//       ...
//       if (output is due) { hip.get_solution(current_solution); hip.get_cell_average(cell_average); output_results(); }
//       // `old_solution = current_solution` (:1110) has happened on the device; `predictor` is unused on the explicit path
//
// test_parity (what a deal.II build should run first): with the reference's own assemble_system(integrator_explicit) on the
// host and hip.assemble_system(rhs) on the device for the same current_solution, max |rhs - right_hand_side| / max |rhs|
// is expected <= 1e-12 (the bar tests/test_gpu_parity.py holds the engine to against oracle/dflo_oracle.cc).
