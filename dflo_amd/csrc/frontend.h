// frontend.h -- host-side front end of the stand-alone driver (dflo_hip_run): dflo's input.prm, the
// FunctionParser expressions in it, initial data, and the VTU output.  Everything here is set-up and I/O
// around the C ABI of include/dflo_hip.h; none of it is on the time-stepping path.
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <vector>

#include "../../include/dflo_hip.h"
#include "../../include/dflo_mesh.h"

namespace dflo_fe {

// ---- expressions (what deal.II's FunctionParser evaluates for the "w_i value" entries)
struct Program {
  std::vector<int32_t> ops;    // [n][2] (dflo_expr_op, constant index) -- the form dflo_hip_set_boundary_program takes
  std::vector<double> consts;
  bool uses_t = false;
  double eval(double x, double y, double t) const;   // host interpreter (initial data, tests)
};
// throws std::runtime_error on a syntax error / unknown symbol; variables: x, y and (if allow_t) t
Program compile_expression(const std::string &text, bool allow_t);

// ---- input.prm (ParameterHandler text format; schema of src/parameters.cc:10-551 + the periodic entries of
//      src_mpi/parameters.cc:397-410)
struct Prm {
  std::map<std::string, std::string> top;
  std::map<std::string, std::map<std::string, std::string>> sub;
  const std::string &get(const std::string &section, const std::string &key) const;
  double get_double(const std::string &section, const std::string &key) const;
  long get_int(const std::string &section, const std::string &key) const;
  bool get_bool(const std::string &section, const std::string &key) const;
};
Prm parse_prm(const std::string &text);   // defaults filled in; throws on undeclared keys / bad values

struct Deck {   // Parameters::AllParameters<2> for the explicit path
  std::string mesh_file, basis, mapping, ic_function, directory;
  int degree = 1;
  dflo_params_t params{};
  std::string boundary_expr[DFLO_MAX_BOUNDARIES][4], ic_expr[4];
  struct Periodic { int first, second, direction; };
  std::vector<Periodic> periodic;
  bool schlieren = false;
  double output_time_step = 1e20;
  long output_iter_step = 1000000;
  std::string output_format;
};
Deck make_deck(const Prm &prm, const std::string &directory);   // the consistency checks of parse_parameters; throws

// ---- initial data: interpolation (Qk) or L2 projection (Pk) of fn(x, y) -> w[4], src/ic.cc:104-181
typedef void (*StateFn)(double x, double y, double *w, const void *ctx);
std::vector<double> initial_state(const dflo_mesh_t *mesh, StateFn fn, const void *ctx);
void isentropic_vortex(double x, double y, double *w, const void *);   // src/ic.cc:44-61
void vortex_system(double x, double y, double *w, const void *);       // src/ic.cc:68-94
void rayleigh_taylor(double x, double y, double *w, const void *gravity);   // src/ic.cc:12-37

// ---- output (src/output.cc:33-107): solution-NNN.vtu and shock.vtu
void write_vtu(const std::string &path, const dflo_mesh_t *mesh, const std::vector<double> &u, double time, int cycle, bool schlieren);
void write_shock_vtu(const std::string &path, const dflo_mesh_t *mesh, const std::vector<double> &shock_indicator);

}  // namespace dflo_fe
