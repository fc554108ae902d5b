// dflo_hip_run -- stand-alone driver: ConservationLaw<2>::run() (src/claw.cc:955-1129) for the explicit rk3 path on the
// device engine, with dflo's own command line (src/main.cc:22-27):
//
//     dflo_hip_run input.prm [n_threads] [--outdir DIR] [--max-steps N] [--fast] [--quiet] [--device K]
//     dflo_hip_run --parse input.prm          print what was read (no device needed)
//     dflo_hip_run --eval "expression" x y t  evaluate a FunctionParser expression on the host
//
// It reads the same input.prm keys as dflo and the Gmsh .msh the file names, hands every boundary function to the
// device as a postfix program, limits the initial condition, runs the time loop with the reference's output cadence
// and writes solution-NNN.vtu / shock.vtu.  Host code around the C ABI of include/dflo_hip.h only.
#include <sys/stat.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "frontend.h"

using namespace dflo_fe;

namespace {

std::string read_file(const std::string &path) {
  std::ifstream f(path);
  if (!f) throw std::runtime_error("cannot open " + path);
  std::stringstream ss;
  ss << f.rdbuf();
  return ss.str();
}
std::string dirname_of(const std::string &path) {
  const size_t s = path.find_last_of('/');
  return s == std::string::npos ? "." : path.substr(0, s);
}
void chk(int rc, dflo_hip_handle h, const char *what) {
  if (rc) throw std::runtime_error(std::string(what) + ": " + dflo_hip_last_error(h));
}
void chk_mesh(int rc, const char *what) {
  if (rc) throw std::runtime_error(std::string(what) + ": " + dflo_mesh_last_error());
}

struct ExprIC {
  Program w[4];
};
void expr_state(double x, double y, double *w, const void *ctx) {
  const ExprIC *e = (const ExprIC *)ctx;
  for (int c = 0; c < 4; ++c) w[c] = e->w[c].eval(x, y, 0.0);
}

int parse_only(const std::string &path) {
  const Deck d = make_deck(parse_prm(read_file(path)), dirname_of(path));
  std::printf("mesh file = %s\ndegree = %d\nbasis = %s\nmapping = %s\n", d.mesh_file.c_str(), d.degree, d.basis.c_str(), d.mapping.c_str());
  const dflo_params_t &p = d.params;
  std::printf("flux = %d\nlimiter = %d\nchar_lim = %d\npos_lim = %d\nglobal = %d\ncfl = %.17g\ntime_step = %.17g\nfinal_time = %.17g\n"
              "M = %.17g\nbeta = %.17g\ngravity = %.17g\nshock_indicator = %d\n",
              p.flux_type, p.limiter_type, p.char_lim, p.pos_lim, p.global_time_step, p.cfl, p.time_step, p.final_time, p.M, p.beta,
              p.gravity, p.shock_indicator);
  for (int b = 0; b < DFLO_MAX_BOUNDARIES; ++b) std::printf("bc_kind[%d] = %d\n", b, p.bc_kind[b]);
  for (auto &q : d.periodic) std::printf("periodic = %d %d %d\n", q.first, q.second, q.direction);
  std::printf("ic_function = %s\nschlieren = %d\noutput_iter_step = %ld\noutput_time_step = %.17g\n", d.ic_function.c_str(), (int)d.schlieren,
              d.output_iter_step, d.output_time_step);
  for (int b = 0; b < DFLO_MAX_BOUNDARIES; ++b)
    for (int c = 0; c < 4; ++c) {
      const Program pr = compile_expression(d.boundary_expr[b][c], true);
      std::printf("program[%d][%d] = %zu ops, %zu consts, uses_t %d\n", b, c, pr.ops.size() / 2, pr.consts.size(), (int)pr.uses_t);
    }
  return 0;
}

int run(int argc, char **argv) {
  std::string input, outdir = ".";
  long max_steps = -1;
  bool fast = false, quiet = false;
  int device = 0;
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    if (a == "--outdir" && i + 1 < argc) outdir = argv[++i];
    else if (a == "--max-steps" && i + 1 < argc) max_steps = std::atol(argv[++i]);
    else if (a == "--device" && i + 1 < argc) device = std::atoi(argv[++i]);
    else if (a == "--fast") fast = true;
    else if (a == "--quiet") quiet = true;
    else if (input.empty()) input = a;
    // a second positional argument is dflo's thread count: accepted, unused
  }
  if (input.empty()) throw std::runtime_error("usage: dflo_hip_run input.prm [n_threads] [--outdir DIR] [--max-steps N] [--fast] [--quiet]");
  mkdir(outdir.c_str(), 0777);
  const Deck d = make_deck(parse_prm(read_file(input)), dirname_of(input));

  // ---- mesh (src/claw.cc:957-967) and engine
  const std::string mpath = d.mesh_file[0] == '/' ? d.mesh_file : d.directory + "/" + d.mesh_file;
  dflo_mesh_t *mesh = nullptr;
  chk_mesh(dflo_mesh_read_gmsh(mpath.c_str(), d.degree, d.mapping == "cartesian" ? DFLO_MAP_CARTESIAN : DFLO_MAP_Q1, &mesh), "mesh");
  mesh->basis = d.basis == "Pk" ? DFLO_BASIS_PK : DFLO_BASIS_QK;
  for (auto &q : d.periodic) chk_mesh(dflo_mesh_make_periodic(mesh, q.first, q.second, q.direction), "periodic boundaries");
  dflo_hip_handle h = nullptr;
  chk(dflo_hip_create(mesh, &d.params, device, &h), nullptr, "dflo_hip_create");
  const long long n_dofs = dflo_hip_n_dofs(h);
  if (!quiet) std::printf("Number of active cells:       %d\nNumber of degrees of freedom: %lld\n", mesh->n_cells, n_dofs);

  // ---- boundary functions: every boundary in use is handed to the device as postfix programs
  const int nb = dflo_hip_n_boundary_faces(h);
  if (nb > 0) {
    std::vector<int32_t> bid(nb);
    chk(dflo_hip_boundary_faces(h, nullptr, nullptr, bid.data(), nullptr), h, "boundary faces");
    bool used[DFLO_MAX_BOUNDARIES] = {};
    for (int b : bid) used[b] = true;
    for (int b = 0; b < DFLO_MAX_BOUNDARIES; ++b)
      if (used[b])
        for (int c = 0; c < 4; ++c) {
          const Program p = compile_expression(d.boundary_expr[b][c], true);
          chk(dflo_hip_set_boundary_program(h, b, c, (int32_t)(p.ops.size() / 2), p.ops.data(), (int32_t)p.consts.size(), p.consts.data()), h,
              "boundary program");
        }
    std::vector<double> zero((size_t)nb * (d.degree + 1) * 4, 0.0);
    chk(dflo_hip_set_boundary_values(h, 0, zero.data()), h, "boundary values");
    chk(dflo_hip_set_boundary_values(h, 1, zero.data()), h, "boundary values");
  }

  // ---- initial condition (src/ic.cc:104-181), cell averages, limited once (src/claw.cc:997-1002)
  std::vector<double> u;
  if (d.ic_function == "isenvort") u = initial_state(mesh, isentropic_vortex, nullptr);
  else if (d.ic_function == "vortsys") u = initial_state(mesh, vortex_system, nullptr);
  else if (d.ic_function == "rt") u = initial_state(mesh, rayleigh_taylor, &d.params.gravity);
  else {
    ExprIC e;
    for (int c = 0; c < 4; ++c) e.w[c] = compile_expression(d.ic_expr[c], false);
    u = initial_state(mesh, expr_state, &e);
  }
  chk(dflo_hip_set_solution(h, u.data()), h, "set_solution");
  chk(dflo_hip_apply_limiter(h), h, "apply_limiter");
  double elapsed = 0.0;
  long time_iter = 0;
  int file_number = 0;
  std::vector<double> shock(mesh->n_cells);
  auto output = [&]() {
    char name[64];
    std::snprintf(name, sizeof name, "solution-%03d.vtu", file_number);
    if (!quiet) std::printf("Writing file %s\n", name);
    chk(dflo_hip_get_solution(h, u.data()), h, "get_solution");
    write_vtu(outdir + "/" + name, mesh, u, elapsed, file_number, d.schlieren);
    chk(dflo_hip_get_shock_indicator(h, shock.data()), h, "shock indicator");
    write_shock_vtu(outdir + "/shock.vtu", mesh, shock);
    ++file_number;
  };
  output();
  double next_output_time = elapsed + d.output_time_step;
  long next_output_iter = time_iter + d.output_iter_step;
  const double final_time = d.params.final_time;
  while (elapsed < final_time && (max_steps < 0 || time_iter < max_steps)) {
    long chunk = 1;
    if (fast && d.output_time_step >= 1e19) {
      chunk = std::min<long>(next_output_iter - time_iter, 64);
      if (max_steps >= 0) chunk = std::min(chunk, max_steps - time_iter);
      chunk = std::max<long>(chunk, 1);
      if (chunk > 1 && final_time < 1e19) {
        // a chunk must not run past final_time (the loop of src/claw.cc:1026 stops there): size it by the current time
        // step with a margin for its growth; the last steps are taken one by one
        double dt_now = 0.0;
        chk(dflo_hip_compute_dt(h, elapsed, &dt_now), h, "compute_dt");
        chunk = dt_now > 0 ? std::max<long>(1, std::min<long>(chunk, (long)(0.8 * (final_time - elapsed) / dt_now))) : 1;
      }
    }
    if (chunk > 1) {   // dt and time stay on the device
      chk(dflo_hip_advance(h, (int)chunk, &elapsed), h, "advance");
      time_iter += chunk;
      if (!quiet) std::printf("It=%ld, T=%.12g\n", time_iter, elapsed);
    } else {
      double dt, r0, r1;
      chk(dflo_hip_compute_dt(h, elapsed, &dt), h, "compute_dt");                 // :1029
      if (!quiet) std::printf("\nIt=%ld, T=%.12g, dt=%.12g, cfl=%g\n", time_iter + 1, elapsed + dt, dt, d.params.cfl);
      chk(dflo_hip_step(h, dt, &r0, &r1), h, "step");                              // iterate_explicit, :1051
      if (!quiet) std::printf("   %-16.3e %-16.3e\n", r0, r1);
      elapsed += dt;
      ++time_iter;
    }
    if (elapsed >= next_output_time || time_iter >= next_output_iter || std::fabs(elapsed - final_time) < 1.0e-13) {   // :1091-1099
      output();
      next_output_time = elapsed + d.output_time_step;
      next_output_iter = time_iter + d.output_iter_step;
    }
  }
  dflo_hip_destroy(h);
  dflo_mesh_free(mesh);
  return 0;
}

}  // namespace

int main(int argc, char **argv) {
  try {
    if (argc >= 3 && std::strcmp(argv[1], "--parse") == 0) return parse_only(argv[2]);
    if (argc >= 3 && std::strcmp(argv[1], "--eval") == 0) {
      const Program p = compile_expression(argv[2], true);
      const double x = argc > 3 ? std::atof(argv[3]) : 0.0, y = argc > 4 ? std::atof(argv[4]) : 0.0, t = argc > 5 ? std::atof(argv[5]) : 0.0;
      std::printf("%.17g\n", p.eval(x, y, t));
      return 0;
    }
    return run(argc, argv);
  } catch (const std::exception &e) {   // src/main.cc:56-78
    std::fprintf(stderr, "\n----------------------------------------------------\nException on processing:\n%s\nAborting!\n"
                         "----------------------------------------------------\n", e.what());
    return 1;
  }
}
