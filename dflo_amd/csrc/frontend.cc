// frontend.cc -- see frontend.h
#include "frontend.h"

#include <zlib.h>

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <memory>
#include <sstream>
#include <stdexcept>

#include "basis.h"

namespace dflo_fe {

// =====================================================================================================
// expressions
// =====================================================================================================
namespace {

struct Tok { int kind; double num; std::string s; };   // kind: 0 number, 1 name, 2 operator, 3 end

std::vector<Tok> tokenize(const std::string &t) {
  std::vector<Tok> out;
  size_t i = 0;
  while (i < t.size()) {
    if (std::isspace((unsigned char)t[i])) { ++i; continue; }
    if (std::isdigit((unsigned char)t[i]) || (t[i] == '.' && i + 1 < t.size() && std::isdigit((unsigned char)t[i + 1]))) {
      size_t j = i;
      while (j < t.size() && (std::isdigit((unsigned char)t[j]) || t[j] == '.')) ++j;
      if (j < t.size() && (t[j] == 'e' || t[j] == 'E')) {
        size_t k = j + 1;
        if (k < t.size() && (t[k] == '+' || t[k] == '-')) ++k;
        if (k < t.size() && std::isdigit((unsigned char)t[k])) {
          while (k < t.size() && std::isdigit((unsigned char)t[k])) ++k;
          j = k;
        }
      }
      out.push_back({0, std::stod(t.substr(i, j - i)), ""});
      i = j;
    } else if (std::isalpha((unsigned char)t[i]) || t[i] == '_') {
      size_t j = i;
      while (j < t.size() && (std::isalnum((unsigned char)t[j]) || t[j] == '_')) ++j;
      out.push_back({1, 0.0, t.substr(i, j - i)});
      i = j;
    } else {
      static const char *two[] = {"<=", ">=", "==", "!=", "&&", "||"};
      std::string op;
      for (const char *w : two)
        if (t.compare(i, 2, w) == 0) op = w;
      if (op.empty()) {
        if (std::strchr("+-*/^()<>,?:", t[i])) op = std::string(1, t[i]);
        else throw std::runtime_error("cannot parse '" + t + "' at '" + t.substr(i) + "'");
      }
      out.push_back({2, 0.0, op});
      i += op.size();
    }
  }
  out.push_back({3, 0.0, ""});
  return out;
}

struct Node {
  int op = 0;           // dflo_expr_op; DFLO_OP_CONST carries `value`
  double value = 0.0;
  std::vector<std::unique_ptr<Node>> kids;
};
typedef std::unique_ptr<Node> NodeP;

NodeP make(int op, NodeP a = nullptr, NodeP b = nullptr, NodeP c = nullptr) {
  NodeP n(new Node);
  n->op = op;
  if (a) n->kids.push_back(std::move(a));
  if (b) n->kids.push_back(std::move(b));
  if (c) n->kids.push_back(std::move(c));
  return n;
}
NodeP num(double v) {
  NodeP n(new Node);
  n->op = DFLO_OP_CONST;
  n->value = v;
  return n;
}

const std::map<std::string, int> &functions1() {
  static const std::map<std::string, int> f = {
      {"sin", DFLO_OP_SIN}, {"cos", DFLO_OP_COS}, {"tan", DFLO_OP_TAN}, {"exp", DFLO_OP_EXP}, {"log", DFLO_OP_LOG},
      {"ln", DFLO_OP_LOG}, {"sqrt", DFLO_OP_SQRT}, {"abs", DFLO_OP_ABS}, {"tanh", DFLO_OP_TANH}, {"sinh", DFLO_OP_SINH},
      {"cosh", DFLO_OP_COSH}, {"asin", DFLO_OP_ASIN}, {"acos", DFLO_OP_ACOS}, {"atan", DFLO_OP_ATAN},
      {"floor", DFLO_OP_FLOOR}, {"ceil", DFLO_OP_CEIL}, {"sign", DFLO_OP_SIGN}, {"log10", DFLO_OP_LOG10},
      {"erf", DFLO_OP_ERF}, {"erfc", DFLO_OP_ERFC}};
  return f;
}
const std::map<std::string, int> &functions2() {
  static const std::map<std::string, int> f = {{"min", DFLO_OP_MIN}, {"max", DFLO_OP_MAX}, {"pow", DFLO_OP_POW}, {"atan2", DFLO_OP_ATAN2}};
  return f;
}

// precedence (low to high): ?:  ||  &&  == !=  < <= > >=  + -  * /  unary -  ^
struct Parser {
  std::vector<Tok> toks;
  size_t i = 0;
  bool allow_t;
  std::string text;
  const Tok &peek() const { return toks[i]; }
  bool is_op(const char *o) const { return toks[i].kind == 2 && toks[i].s == o; }
  void expect(const char *o) {
    if (!is_op(o)) throw std::runtime_error("expected '" + std::string(o) + "' in '" + text + "'");
    ++i;
  }
  NodeP ternary() {
    NodeP c = logic_or();
    if (is_op("?")) {
      ++i;
      NodeP a = ternary();
      expect(":");
      NodeP b = ternary();
      return make(DFLO_OP_SEL, std::move(c), std::move(a), std::move(b));
    }
    return c;
  }
  NodeP binary(NodeP (Parser::*sub)(), std::initializer_list<std::pair<const char *, int>> ops) {
    NodeP f = (this->*sub)();
    for (;;) {
      int code = -1;
      for (auto &o : ops)
        if (is_op(o.first)) code = o.second;
      if (code < 0) return f;
      ++i;
      f = make(code, std::move(f), (this->*sub)());
    }
  }
  NodeP logic_or() { return binary(&Parser::logic_and, {{"||", DFLO_OP_OR}}); }
  NodeP logic_and() { return binary(&Parser::equality, {{"&&", DFLO_OP_AND}}); }
  NodeP equality() { return binary(&Parser::relational, {{"==", DFLO_OP_EQ}, {"!=", DFLO_OP_NE}}); }
  NodeP relational() {
    return binary(&Parser::additive, {{"<=", DFLO_OP_LE}, {">=", DFLO_OP_GE}, {"<", DFLO_OP_LT}, {">", DFLO_OP_GT}});
  }
  NodeP additive() { return binary(&Parser::multiplicative, {{"+", DFLO_OP_ADD}, {"-", DFLO_OP_SUB}}); }
  NodeP multiplicative() { return binary(&Parser::unary, {{"*", DFLO_OP_MUL}, {"/", DFLO_OP_DIV}}); }
  NodeP unary() {
    if (is_op("-")) { ++i; return make(DFLO_OP_NEG, unary()); }
    if (is_op("+")) { ++i; return unary(); }
    return power();
  }
  NodeP power() {
    NodeP base = atom();
    if (is_op("^")) { ++i; return make(DFLO_OP_POW, std::move(base), unary()); }   // right associative
    return base;
  }
  NodeP atom() {
    const Tok t = peek();
    if (t.kind == 0) { ++i; return num(t.num); }
    if (is_op("(")) {
      ++i;
      NodeP f = ternary();
      expect(")");
      return f;
    }
    if (t.kind == 1) {
      ++i;
      if (is_op("(")) {
        ++i;
        std::vector<NodeP> args;
        args.push_back(ternary());
        while (is_op(",")) { ++i; args.push_back(ternary()); }
        expect(")");
        if (t.s == "if" && args.size() == 3) return make(DFLO_OP_SEL, std::move(args[0]), std::move(args[1]), std::move(args[2]));
        if (args.size() == 1 && functions1().count(t.s)) return make(functions1().at(t.s), std::move(args[0]));
        if (args.size() == 2 && functions2().count(t.s)) return make(functions2().at(t.s), std::move(args[0]), std::move(args[1]));
        throw std::runtime_error("unknown function " + t.s + " in '" + text + "'");
      }
      if (t.s == "x") return make(DFLO_OP_X);
      if (t.s == "y") return make(DFLO_OP_Y);
      if (t.s == "t" && allow_t) return make(DFLO_OP_T);
      if (t.s == "pi" || t.s == "Pi" || t.s == "PI" || t.s == "_pi") return num(3.14159265358979323846);
      if (t.s == "e" || t.s == "_e") return num(2.71828182845904523536);
      throw std::runtime_error("unknown symbol '" + t.s + "' in '" + text + "'");
    }
    throw std::runtime_error("unexpected '" + t.s + "' in '" + text + "'");
  }
};

double apply(int op, double a, double b, double c) {
  switch (op) {
    case DFLO_OP_NEG: return -a;
    case DFLO_OP_ADD: return a + b;
    case DFLO_OP_SUB: return a - b;
    case DFLO_OP_MUL: return a * b;
    case DFLO_OP_DIV: return a / b;
    case DFLO_OP_POW: return std::pow(a, b);
    case DFLO_OP_LT: return a < b;
    case DFLO_OP_LE: return a <= b;
    case DFLO_OP_GT: return a > b;
    case DFLO_OP_GE: return a >= b;
    case DFLO_OP_EQ: return a == b;
    case DFLO_OP_NE: return a != b;
    case DFLO_OP_AND: return a != 0.0 && b != 0.0;
    case DFLO_OP_OR: return a != 0.0 || b != 0.0;
    case DFLO_OP_SEL: return a != 0.0 ? b : c;
    case DFLO_OP_SIN: return std::sin(a);
    case DFLO_OP_COS: return std::cos(a);
    case DFLO_OP_TAN: return std::tan(a);
    case DFLO_OP_EXP: return std::exp(a);
    case DFLO_OP_LOG: return std::log(a);
    case DFLO_OP_SQRT: return std::sqrt(a);
    case DFLO_OP_ABS: return std::fabs(a);
    case DFLO_OP_MIN: return std::fmin(a, b);
    case DFLO_OP_MAX: return std::fmax(a, b);
    case DFLO_OP_ATAN2: return std::atan2(a, b);
    case DFLO_OP_TANH: return std::tanh(a);
    case DFLO_OP_SINH: return std::sinh(a);
    case DFLO_OP_COSH: return std::cosh(a);
    case DFLO_OP_ASIN: return std::asin(a);
    case DFLO_OP_ACOS: return std::acos(a);
    case DFLO_OP_ATAN: return std::atan(a);
    case DFLO_OP_FLOOR: return std::floor(a);
    case DFLO_OP_CEIL: return std::ceil(a);
    case DFLO_OP_SIGN: return a > 0.0 ? 1.0 : (a < 0.0 ? -1.0 : 0.0);
    case DFLO_OP_LOG10: return std::log10(a);
    case DFLO_OP_ERF: return std::erf(a);
    case DFLO_OP_ERFC: return std::erfc(a);
  }
  return std::nan("");
}

void fold(NodeP &n) {   // evaluate variable-free subtrees once
  bool all_const = !n->kids.empty();
  for (auto &k : n->kids) {
    fold(k);
    all_const &= k->op == DFLO_OP_CONST;
  }
  if (all_const) {
    const double a = n->kids[0]->value, b = n->kids.size() > 1 ? n->kids[1]->value : 0.0, c = n->kids.size() > 2 ? n->kids[2]->value : 0.0;
    const double v = apply(n->op, a, b, c);
    n->kids.clear();
    n->op = DFLO_OP_CONST;
    n->value = v;
  }
}

int emit(const Node &n, Program &p) {   // -> stack depth needed
  if (n.op == DFLO_OP_CONST) {
    p.ops.push_back(DFLO_OP_CONST);
    p.ops.push_back((int32_t)p.consts.size());
    p.consts.push_back(n.value);
    return 1;
  }
  if (n.kids.empty()) {
    p.ops.push_back(n.op);
    p.ops.push_back(0);
    if (n.op == DFLO_OP_T) p.uses_t = true;
    return 1;
  }
  int depth = 0, k = 0;
  for (auto &c : n.kids) depth = std::max(depth, (k++) + emit(*c, p));
  p.ops.push_back(n.op);
  p.ops.push_back(0);
  return depth;
}

}  // namespace

Program compile_expression(const std::string &text, bool allow_t) {
  Parser ps;
  ps.toks = tokenize(text);
  ps.allow_t = allow_t;
  ps.text = text;
  NodeP root = ps.ternary();
  if (ps.peek().kind != 3) throw std::runtime_error("unexpected '" + ps.peek().s + "' in '" + text + "'");
  fold(root);
  Program p;
  if (emit(*root, p) > 16) throw std::runtime_error("expression '" + text + "' needs an evaluation stack deeper than 16");
  return p;
}

double Program::eval(double x, double y, double t) const {
  double st[32];
  int sp = 0;
  for (size_t i = 0; i < ops.size(); i += 2) {
    const int op = ops[i];
    if (op == DFLO_OP_CONST) st[sp++] = consts[ops[i + 1]];
    else if (op == DFLO_OP_X) st[sp++] = x;
    else if (op == DFLO_OP_Y) st[sp++] = y;
    else if (op == DFLO_OP_T) st[sp++] = t;
    else if (op == DFLO_OP_SEL) { sp -= 2; st[sp - 1] = apply(op, st[sp - 1], st[sp], st[sp + 1]); }
    else {
      const bool binary = (op >= DFLO_OP_ADD && op <= DFLO_OP_OR) || op == DFLO_OP_MIN || op == DFLO_OP_MAX || op == DFLO_OP_ATAN2;
      if (binary) { --sp; st[sp - 1] = apply(op, st[sp - 1], st[sp], 0.0); }
      else st[sp - 1] = apply(op, st[sp - 1], 0.0, 0.0);
    }
  }
  return st[0];
}

// =====================================================================================================
// input.prm
// =====================================================================================================
namespace {

struct Entry { const char *section, *key, *def, *pattern; };   // pattern: "d" double, "i" int, "b" bool, "*" anything, or a|b|c

const std::vector<Entry> &schema() {
  static std::vector<Entry> s;
  if (!s.empty()) return s;
  static std::vector<std::string> keep;   // storage for the generated names
  keep.reserve(4096);
  auto add = [&](const std::string &sec, const std::string &key, const char *def, const char *pat) {
    keep.push_back(sec);
    const char *a = keep.back().c_str();
    keep.push_back(key);
    s.push_back({a, keep.back().c_str(), def, pat});
  };
  add("", "mesh type", "gmsh", "ucd|gmsh"); add("", "mesh file", "grid.msh", "*"); add("", "degree", "1", "i");
  add("", "basis", "Qk", "Qk|Pk"); add("", "mapping", "q1", "q1|q2|cartesian"); add("", "diffusion power", "2.0", "d");
  add("", "diffusion coefficient", "0.0", "d"); add("", "gravity", "0.0", "d");
  const char *ts = "time stepping";
  add(ts, "stationary", "false", "b"); add(ts, "cfl", "0.0", "d"); add(ts, "time step type", "global", "global|local");
  add(ts, "time step", "-1.0", "d"); add(ts, "final time", "1.0e20", "d"); add(ts, "theta scheme value", "1.0", "d");
  add(ts, "nonlinear iterations", "1", "i");
  const char *ic = "initial condition";
  add(ic, "function", "none", "none|rt|isenvort|vortsys");
  for (int c = 0; c < 4; ++c) add(ic, "w_" + std::to_string(c) + " value", "0.0", "*");
  const char *ls = "linear solver";
  add(ls, "output", "quiet", "quiet|verbose"); add(ls, "method", "rk3", "gmres|direct|umfpack|rk3|mood"); add(ls, "residual", "1e-10", "d");
  add(ls, "max iters", "300", "i"); add(ls, "ilut fill", "2", "d"); add(ls, "ilut absolute tolerance", "1e-9", "d");
  add(ls, "ilut relative tolerance", "1.1", "d"); add(ls, "ilut drop tolerance", "1e-10", "d");
  const char *rf = "refinement";
  add(rf, "refinement", "true", "b"); add(rf, "time step", "1.0e20", "d"); add(rf, "iter step", "100000000", "i");
  add(rf, "refinement fraction", "0.1", "d"); add(rf, "unrefinement fraction", "0.1", "d"); add(rf, "max elements", "1000000", "d");
  add(rf, "shock value", "4.0", "d"); add(rf, "shock levels", "3.0", "d");
  add("flux", "flux", "lxf", "lxf|sw|kfvs|roe|hllc"); add("flux", "stab", "mesh", "constant|mesh"); add("flux", "stab value", "1", "d");
  const char *lm = "limiter";
  add(lm, "shock indicator", "limiter", "limiter|density|energy|u2"); add(lm, "type", "none", "none|TVB");
  add(lm, "characteristic limiter", "false", "b"); add(lm, "positivity limiter", "false", "b"); add(lm, "M", "0", "d");
  add(lm, "beta", "1.0", "d"); add(lm, "conserve angular momentum", "false", "b");
  const char *ou = "output";
  add(ou, "schlieren plot", "false", "b"); add(ou, "time step", "1e20", "d"); add(ou, "iter step", "1000000", "d");
  add(ou, "format", "vtk", "vtk|tecplot"); add(ou, "compute angular momentum", "10000000", "d");
  for (int b = 0; b < DFLO_MAX_BOUNDARIES; ++b) {
    const std::string sec = "boundary_" + std::to_string(b);
    add(sec, "type", "outflow", "slip|inflow|outflow|pressure|farfield|periodic");
    add(sec, "pair", "0", "i");
    add(sec, "direction", "x", "x|y");
    for (int c = 0; c < 4; ++c) add(sec, "w_" + std::to_string(c) + " value", "0.0", "*");
  }
  return s;
}

std::string trim(const std::string &s) {
  size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
  return a == std::string::npos ? "" : s.substr(a, b - a + 1);
}
std::string squeeze(const std::string &s) {   // collapse runs of blanks
  std::istringstream in(s);
  std::string w, out;
  while (in >> w) out += (out.empty() ? "" : " ") + w;
  return out;
}

void check_value(const Entry &e, const std::string &v) {
  const std::string pat = e.pattern;
  if (pat == "*") return;
  if (pat == "b") {
    if (v != "true" && v != "false") throw std::runtime_error("entry <" + std::string(e.key) + ">: '" + v + "' is not a bool");
    return;
  }
  if (pat == "d" || pat == "i") {
    char *end = nullptr;
    std::strtod(v.c_str(), &end);
    if (v.empty() || *end) throw std::runtime_error("entry <" + std::string(e.key) + ">: '" + v + "' is not a number");
    return;
  }
  std::istringstream in(pat);
  std::string alt;
  while (std::getline(in, alt, '|'))
    if (alt == v) return;
  throw std::runtime_error("entry <" + std::string(e.key) + ">: '" + v + "' is not one of " + pat);
}

}  // namespace

const std::string &Prm::get(const std::string &section, const std::string &key) const {
  return section.empty() ? top.at(key) : sub.at(section).at(key);
}
double Prm::get_double(const std::string &s, const std::string &k) const { return std::strtod(get(s, k).c_str(), nullptr); }
long Prm::get_int(const std::string &s, const std::string &k) const { return (long)std::strtod(get(s, k).c_str(), nullptr); }
bool Prm::get_bool(const std::string &s, const std::string &k) const { return get(s, k) == "true"; }

Prm parse_prm(const std::string &text) {
  Prm p;
  for (const Entry &e : schema()) {
    if (!*e.section) p.top[e.key] = e.def;
    else p.sub[e.section][e.key] = e.def;
  }
  std::istringstream in(text);
  std::string raw, current;
  bool in_sub = false;
  int ln = 0;
  while (std::getline(in, raw)) {
    ++ln;
    const std::string line = trim(raw.substr(0, raw.find('#')));
    if (line.empty()) continue;
    const std::string where = "line " + std::to_string(ln) + ": ";
    if (line.compare(0, 10, "subsection") == 0 && (line.size() == 10 || std::isspace((unsigned char)line[10]))) {
      if (in_sub) throw std::runtime_error(where + "nested subsection");
      current = squeeze(line.substr(10));
      if (!p.sub.count(current)) throw std::runtime_error(where + "no subsection <" + current + "> was declared");
      in_sub = true;
    } else if (line == "end") {
      if (!in_sub) throw std::runtime_error(where + "'end' outside a subsection");
      in_sub = false;
    } else if (line.compare(0, 3, "set") == 0 && line.size() > 3 && std::isspace((unsigned char)line[3])) {
      const size_t eq = line.find('=');
      if (eq == std::string::npos) throw std::runtime_error(where + "expected 'set key = value'");
      const std::string key = squeeze(line.substr(3, eq - 3)), value = trim(line.substr(eq + 1));
      auto &target = in_sub ? p.sub[current] : p.top;
      if (!target.count(key))
        throw std::runtime_error(where + "no entry with name <" + key + "> was declared" + (in_sub ? " in subsection <" + current + ">" : ""));
      target[key] = value;
    } else {
      throw std::runtime_error(where + "cannot interpret '" + trim(raw) + "'");
    }
  }
  if (in_sub) throw std::runtime_error("subsection <" + current + "> is not closed");
  for (const Entry &e : schema()) check_value(e, p.get(e.section, e.key));
  return p;
}

Deck make_deck(const Prm &prm, const std::string &directory) {
  Deck d;
  d.directory = directory;
  if (prm.get("", "mesh type") != "gmsh") throw std::runtime_error("mesh type = ucd is not provided (gmsh)");
  d.mesh_file = prm.get("", "mesh file");
  d.degree = (int)prm.get_int("", "degree");
  d.basis = prm.get("", "basis");
  d.mapping = prm.get("", "mapping");
  const char *ts = "time stepping";
  double cfl = prm.get_double(ts, "cfl"), time_step = prm.get_double(ts, "time step"), final_time = prm.get_double(ts, "final time");
  // stationary: src/parameters.cc:425-429 sets dt = 1, final time = 1e20 and compute_time_step returns at once
  // (src/claw.cc:449-450) -- the steady-state mode of the implicit solver, not part of the explicit path
  if (prm.get_bool(ts, "stationary")) throw std::runtime_error("stationary = true: steady-state runs belong to the implicit solver, which is not provided");
  if (!(cfl > 0 || time_step > 0)) throw std::runtime_error("cfl and time_step zero");
  if (prm.get("linear solver", "method") != "rk3")
    throw std::runtime_error("linear solver method = " + prm.get("linear solver", "method") + ": only the explicit rk3 path is provided");
  if (prm.get_bool("refinement", "refinement") && d.basis == "Pk") throw std::runtime_error("Refinement does not work for Pk basis");
  if (prm.get_bool("refinement", "refinement"))
    throw std::runtime_error("refinement = true: grid adaptation is not part of the explicit device path (set refinement = false)");
  const std::string lim = prm.get("limiter", "type");
  if (lim == "TVB" && d.mapping != "cartesian") throw std::runtime_error("TVB limiter works on cartesian grids only");
  if (d.basis == "Pk" && d.mapping != "cartesian") throw std::runtime_error("Pk basis can only be used with Cartesian grids");
  // (mapping = q2: MappingQ(2) on the straight-edged cells of a .msh file is the bilinear map; the engine takes it as q1)
  if (prm.get_double("", "diffusion coefficient") != 0.0)
    throw std::runtime_error("diffusion coefficient != 0: the shock-capturing term belongs to the implicit path");
  dflo_params_t &p = d.params;
  static const char *fluxes[] = {"lxf", "sw", "kfvs", "roe", "hllc"};
  for (int f = 0; f < 5; ++f)
    if (prm.get("flux", "flux") == fluxes[f]) p.flux_type = f;
  p.limiter_type = lim == "TVB" ? DFLO_LIMITER_TVB : DFLO_LIMITER_NONE;
  p.char_lim = prm.get_bool("limiter", "characteristic limiter");
  p.pos_lim = prm.get_bool("limiter", "positivity limiter");
  p.global_time_step = prm.get(ts, "time step type") == "global";
  p.n_rk = 0;
  p.gravity = prm.get_double("", "gravity");
  p.cfl = cfl;
  p.time_step = time_step;
  p.final_time = final_time;
  p.M = prm.get_double("limiter", "M");
  p.beta = prm.get_double("limiter", "beta");
  p.conserve_angular_momentum = prm.get_bool("limiter", "conserve angular momentum");   // src/limiter.cc:496-500
  static const char *inds[] = {"limiter", "density", "energy", "u2"};
  for (int k = 0; k < 4; ++k)
    if (prm.get("limiter", "shock indicator") == inds[k]) p.shock_indicator = k;
  static const char *kinds[] = {"inflow", "outflow", "slip", "pressure", "farfield"};
  for (int b = 0; b < DFLO_MAX_BOUNDARIES; ++b) {
    const std::string sec = "boundary_" + std::to_string(b), type = prm.get(sec, "type");
    p.bc_kind[b] = DFLO_BC_OUTFLOW;
    for (int k = 0; k < 5; ++k)
      if (type == kinds[k]) p.bc_kind[b] = k;
    if (type == "periodic") {   // each pair once, src_mpi/parameters.cc:524-560
      const int other = (int)prm.get_int(sec, "pair");
      bool seen = false;
      for (auto &q : d.periodic) seen |= (q.first == other && q.second == b) || (q.first == b && q.second == other);
      if (!seen) d.periodic.push_back({b, other, prm.get(sec, "direction") == "y" ? 1 : 0});
    }
    for (int c = 0; c < 4; ++c) d.boundary_expr[b][c] = prm.get(sec, "w_" + std::to_string(c) + " value");
  }
  d.ic_function = prm.get("initial condition", "function");
  for (int c = 0; c < 4; ++c) d.ic_expr[c] = prm.get("initial condition", "w_" + std::to_string(c) + " value");
  d.schlieren = prm.get_bool("output", "schlieren plot");
  d.output_time_step = prm.get_double("output", "time step");
  d.output_iter_step = prm.get_int("output", "iter step");
  d.output_format = prm.get("output", "format");
  if (d.output_format != "vtk") throw std::runtime_error("output format = tecplot: dflo_hip_run writes vtk (python -m dflo_amd writes both)");
  return d;
}

// =====================================================================================================
// initial data
// =====================================================================================================
namespace {
constexpr double kGamma = 1.4;

struct Tables {   // Gauss points / weights and the modal table of the scalar element
  int N, ns;
  double x[4], w[4];
  std::vector<double> T;   // [N*N][ns] psi_m at the Gauss node (Pk), identity not stored for Qk
};
Tables tables(int degree, int basis) {
  Tables t;
  t.N = degree + 1;
  dflo::BasisTables b = dflo::make_basis(degree);
  for (int a = 0; a < t.N; ++a) { t.x[a] = b.x[a]; t.w[a] = b.w[a]; }
  t.ns = basis == DFLO_BASIS_PK ? t.N * (t.N + 1) / 2 : t.N * t.N;
  if (basis == DFLO_BASIS_PK) {
    t.T.assign((size_t)t.N * t.N * t.ns, 0.0);
    int m = 0;
    for (int j = 0; j < t.N; ++j)
      for (int i = 0; i < t.N - j; ++i, ++m)
        for (int bb = 0; bb < t.N; ++bb)
          for (int a = 0; a < t.N; ++a) t.T[(size_t)(a + t.N * bb) * t.ns + m] = dflo::legendre01(i, t.x[a]) * dflo::legendre01(j, t.x[bb]);
  }
  return t;
}
void map_point(const double *v, double xi, double eta, double &x, double &y) {   // bilinear map, lexicographic vertices
  const double s0 = (1 - xi) * (1 - eta), s1 = xi * (1 - eta), s2 = (1 - xi) * eta, s3 = xi * eta;
  x = s0 * v[0] + s1 * v[2] + s2 * v[4] + s3 * v[6];
  y = s0 * v[1] + s1 * v[3] + s2 * v[5] + s3 * v[7];
}
}  // namespace

std::vector<double> initial_state(const dflo_mesh_t *mesh, StateFn fn, const void *ctx) {
  const Tables t = tables(mesh->degree, mesh->basis);
  const int N = t.N, nq = N * N, ns = t.ns;
  std::vector<double> u((size_t)mesh->n_cells * 4 * ns, 0.0), f((size_t)4 * nq);
  for (int c = 0; c < mesh->n_cells; ++c) {
    const double *v = mesh->cell_vertices + (size_t)c * 8;
    for (int b = 0; b < N; ++b)
      for (int a = 0; a < N; ++a) {
        double x, y, w[4];
        map_point(v, t.x[a], t.x[b], x, y);
        fn(x, y, w, ctx);
        for (int k = 0; k < 4; ++k) f[(size_t)k * nq + a + N * b] = w[k];
      }
    double *uc = &u[(size_t)c * 4 * ns];
    if (mesh->basis == DFLO_BASIS_QK) {   // VectorTools::interpolate: the support points are the Gauss points
      std::copy(f.begin(), f.end(), uc);
    } else {   // u_m = sum_q f(x_q) psi_m(x_q) w_q: the mass matrix of the orthonormal modes is |K| I (src/ic.cc:128-164)
      for (int k = 0; k < 4; ++k)
        for (int m = 0; m < ns; ++m) {
          double s = 0;
          for (int q = 0; q < nq; ++q) s += f[(size_t)k * nq + q] * t.T[(size_t)q * ns + m] * t.w[q % N] * t.w[q / N];
          uc[k * ns + m] = s;
        }
    }
  }
  return u;
}

void isentropic_vortex(double x, double y, double *w, const void *) {
  const double beta = 5.0, a1 = 0.5 * beta / M_PI, a2 = (kGamma - 1.0) * a1 * a1 / 2.0;
  const double r2 = x * x + y * y;
  const double rho = std::pow(1.0 - a2 * std::exp(1.0 - r2), 1.0 / (kGamma - 1.0));
  const double vex = -a1 * y * std::exp(0.5 * (1.0 - r2)), vey = a1 * x * std::exp(0.5 * (1.0 - r2));
  const double pre = std::pow(rho, kGamma);
  w[0] = rho * vex; w[1] = rho * vey; w[2] = rho;
  w[3] = pre / (kGamma - 1.0) + 0.5 * rho * (vex * vex + vey * vey);
}
void vortex_system(double x, double y, double *w, const void *) {
  const double beta = 5.0, Rc = 4.0, a1 = 0.5 * beta / M_PI, a2 = (kGamma - 1.0) * a1 * a1 / 2.0;
  const double xs[3] = {0.0, Rc * std::cos(M_PI / 6.0), -Rc * std::cos(M_PI / 6.0)};
  const double ys[3] = {-Rc, Rc * std::sin(M_PI / 6.0), Rc * std::sin(M_PI / 6.0)};
  double rho = 0, vex = 0, vey = 0;
  for (int i = 0; i < 3; ++i) {
    const double r2 = (x - xs[i]) * (x - xs[i]) + (y - ys[i]) * (y - ys[i]);
    rho += std::pow(1.0 - a2 * std::exp(1.0 - r2), 1.0 / (kGamma - 1.0));
    vex += -a1 * (y - ys[i]) * std::exp(0.5 * (1.0 - r2));
    vey += a1 * (x - xs[i]) * std::exp(0.5 * (1.0 - r2));
  }
  rho -= 2.0; vex /= 3.0; vey /= 3.0;
  double pre = std::pow(rho, kGamma);
  if (std::fabs(x) < 0.1 && std::fabs(y) < 0.1) pre = 50.0;
  w[0] = rho * vex; w[1] = rho * vey; w[2] = rho;
  w[3] = pre / (kGamma - 1.0) + 0.5 * rho * (vex * vex + vey * vey);
}
void rayleigh_taylor(double x, double y, double *w, const void *gravity) {
  const double g = *(const double *)gravity, Lx = 0.5, Ly = 1.5, A = 0.01, P0 = 2.5;
  const double rho = y < 0.0 ? 1.0 : 2.0;
  const double vel = A * (1.0 + std::cos(2.0 * M_PI * x / Lx)) / 2.0 * (1.0 + std::cos(2.0 * M_PI * y / Ly)) / 2.0;
  const double pre = P0 - g * rho * y;
  w[0] = 0.0; w[1] = rho * vel; w[2] = rho;
  w[3] = pre / (kGamma - 1.0) + 0.5 * rho * vel * vel;
}

// =====================================================================================================
// VTU (layout of deal.II's DataOut::build_patches(mapping, degree) + write_vtu; see dflo_amd/vtu.py)
// =====================================================================================================
namespace {

std::string base64(const unsigned char *d, size_t n) {
  static const char *tbl = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
  std::string out;
  out.reserve((n + 2) / 3 * 4);
  for (size_t i = 0; i < n; i += 3) {
    const unsigned v = (d[i] << 16) | ((i + 1 < n ? d[i + 1] : 0) << 8) | (i + 2 < n ? d[i + 2] : 0);
    out += tbl[(v >> 18) & 63];
    out += tbl[(v >> 12) & 63];
    out += i + 1 < n ? tbl[(v >> 6) & 63] : '=';
    out += i + 2 < n ? tbl[v & 63] : '=';
  }
  return out;
}
template <class T>
void data_array(std::ostream &f, const char *name, const std::vector<T> &a, const char *type, int ncomp) {
  const uLong raw = (uLong)(a.size() * sizeof(T));
  uLongf clen = compressBound(raw);
  std::vector<unsigned char> comp(clen);
  if (compress(comp.data(), &clen, (const Bytef *)a.data(), raw) != Z_OK) throw std::runtime_error("zlib compress failed");
  const uint32_t head[4] = {1u, (uint32_t)raw, (uint32_t)raw, (uint32_t)clen};
  f << "    <DataArray type=\"" << type << "\"";
  if (name) f << " Name=\"" << name << "\"";
  if (ncomp > 0) f << " NumberOfComponents=\"" << ncomp << "\"";
  f << " format=\"binary\">\n" << base64((const unsigned char *)head, 16) << base64(comp.data(), clen) << "\n    </DataArray>\n";
}

}  // namespace

void write_vtu(const std::string &path, const dflo_mesh_t *mesh, const std::vector<double> &u, double time, int cycle, bool schlieren) {
  const Tables t = tables(mesh->degree, mesh->basis);
  const int N = t.N, k = mesh->degree, np = N * N, ns = t.ns, nc = mesh->n_owned_cells;
  // value / reference-derivative tables of the scalar element at the equidistant patch points
  std::vector<double> V((size_t)np * ns), Vx((size_t)np * ns), Vy((size_t)np * ns);
  {
    long double gx[dflo::kMaxN], gw[dflo::kMaxN];
    dflo::gauss01(N, gx, gw);
    for (int q = 0; q < N; ++q)
      for (int p = 0; p < N; ++p) {
        const long double xi = (long double)p / k, eta = (long double)q / k;
        if (mesh->basis == DFLO_BASIS_QK) {
          for (int b = 0; b < N; ++b)
            for (int a = 0; a < N; ++a) {
              const long double la = dflo::lagrange_ld(N, gx, a, xi), lb = dflo::lagrange_ld(N, gx, b, eta);
              V[(size_t)(p + N * q) * ns + a + N * b] = (double)(la * lb);
              Vx[(size_t)(p + N * q) * ns + a + N * b] = (double)(dflo::dlagrange_ld(N, gx, a, xi) * lb);
              Vy[(size_t)(p + N * q) * ns + a + N * b] = (double)(la * dflo::dlagrange_ld(N, gx, b, eta));
            }
        } else {
          auto dleg = [](int n, double x) {   // d/dx of sqrt(2n+1) P_n(2x-1) by the three-term recurrence of the derivative
            const double tt = 2.0 * x - 1.0;
            double p0 = 1.0, p1 = tt, d0 = 0.0, d1 = 1.0;
            if (n == 0) return 0.0;
            for (int j = 2; j <= n; ++j) {
              const double pj = ((2.0 * j - 1.0) * tt * p1 - (j - 1.0) * p0) / j;
              const double dj = ((2.0 * j - 1.0) * (p1 + tt * d1) - (j - 1.0) * d0) / j;
              p0 = p1; p1 = pj; d0 = d1; d1 = dj;
            }
            return 2.0 * dflo::kSqrtOdd[n] * d1;
          };
          int m = 0;
          for (int j = 0; j < N; ++j)
            for (int i = 0; i < N - j; ++i, ++m) {
              const double pi = dflo::legendre01(i, (double)xi), pj = dflo::legendre01(j, (double)eta);
              V[(size_t)(p + N * q) * ns + m] = pi * pj;
              Vx[(size_t)(p + N * q) * ns + m] = dleg(i, (double)xi) * pj;
              Vy[(size_t)(p + N * q) * ns + m] = pi * dleg(j, (double)eta);
            }
        }
      }
  }
  const size_t npts = (size_t)nc * np;
  std::vector<double> pts(npts * 3, 0.0), mom(npts * 3, 0.0), vel(npts * 3, 0.0), rho(npts), en(npts), pre(npts), sch;
  if (schlieren) sch.resize(npts);
  std::vector<int32_t> conn, offs;
  std::vector<uint8_t> types;
  for (int c = 0; c < nc; ++c) {
    const double *v = mesh->cell_vertices + (size_t)c * 8;
    const double *uc = &u[(size_t)c * 4 * ns];
    for (int p = 0; p < np; ++p) {
      const size_t g = (size_t)c * np + p;
      const double xi = (double)(p % N) / k, eta = (double)(p / N) / k;
      map_point(v, xi, eta, pts[3 * g], pts[3 * g + 1]);
      double w[4] = {0, 0, 0, 0}, rxi = 0, reta = 0;
      for (int j = 0; j < ns; ++j) {
        for (int cc = 0; cc < 4; ++cc) w[cc] += V[(size_t)p * ns + j] * uc[cc * ns + j];
        rxi += Vx[(size_t)p * ns + j] * uc[2 * ns + j];
        reta += Vy[(size_t)p * ns + j] * uc[2 * ns + j];
      }
      mom[3 * g] = w[0]; mom[3 * g + 1] = w[1];
      vel[3 * g] = w[0] / w[2]; vel[3 * g + 1] = w[1] / w[2];
      rho[g] = w[2]; en[g] = w[3];
      pre[g] = (kGamma - 1.0) * (w[3] - 0.5 * (w[0] * w[0] + w[1] * w[1]) / w[2]);
      if (schlieren) {   // |grad rho|^2 through the bilinear map, src/equation.cc:127-129
        const double a = (1 - eta) * (v[2] - v[0]) + eta * (v[6] - v[4]), cY = (1 - eta) * (v[3] - v[1]) + eta * (v[7] - v[5]);
        const double b = (1 - xi) * (v[4] - v[0]) + xi * (v[6] - v[2]), dY = (1 - xi) * (v[5] - v[1]) + xi * (v[7] - v[3]);
        const double det = a * dY - b * cY, gxr = (dY * rxi - cY * reta) / det, gyr = (-b * rxi + a * reta) / det;
        sch[g] = gxr * gxr + gyr * gyr;
      }
    }
    for (int j = 0; j < k; ++j)
      for (int i = 0; i < k; ++i) {
        const int32_t p0 = (int32_t)((size_t)c * np + i + N * j);
        conn.insert(conn.end(), {p0, p0 + 1, p0 + N + 1, p0 + N});
        offs.push_back((int32_t)conn.size());
        types.push_back(9);
      }
  }
  std::ofstream f(path);
  if (!f) throw std::runtime_error("cannot write " + path);
  f << "<?xml version=\"1.0\" ?>\n<!--\n# vtk DataFile Version 3.0\n#This file was generated by dflo_hip_run (layout of deal.II DataOut::write_vtu)\n-->\n"
    << "<VTKFile type=\"UnstructuredGrid\" version=\"0.1\" compressor=\"vtkZLibDataCompressor\" byte_order=\"LittleEndian\">\n"
    << "<UnstructuredGrid>\n<FieldData>\n";
  char buf[64];
  std::snprintf(buf, sizeof buf, "%.9g", time);
  f << "<DataArray type=\"Float32\" Name=\"TIME\" NumberOfTuples=\"1\" format=\"ascii\">" << buf << "</DataArray>\n"
    << "<DataArray type=\"Float32\" Name=\"CYCLE\" NumberOfTuples=\"1\" format=\"ascii\">" << cycle << "</DataArray>\n"
    << "</FieldData>\n<Piece NumberOfPoints=\"" << npts << "\" NumberOfCells=\"" << types.size() << "\" >\n  <Points>\n";
  data_array(f, nullptr, pts, "Float64", 3);
  f << "  </Points>\n\n  <Cells>\n";
  data_array(f, "connectivity", conn, "Int32", 0);
  data_array(f, "offsets", offs, "Int32", 0);
  data_array(f, "types", types, "UInt8", 0);
  f << "  </Cells>\n  <PointData Scalars=\"scalars\">\n";
  data_array(f, "XMomentum__YMomentum", mom, "Float64", 3);
  data_array(f, "Density", rho, "Float64", 0);
  data_array(f, "Energy", en, "Float64", 0);
  data_array(f, "XVelocity__YVelocity", vel, "Float64", 3);
  data_array(f, "Pressure", pre, "Float64", 0);
  if (schlieren) data_array(f, "schlieren_plot", sch, "Float64", 0);
  f << "  </PointData>\n </Piece>\n </UnstructuredGrid>\n</VTKFile>\n";
}

void write_shock_vtu(const std::string &path, const dflo_mesh_t *mesh, const std::vector<double> &shock) {
  const int nc = mesh->n_owned_cells;
  std::vector<double> pts((size_t)nc * 12, 0.0), mu((size_t)nc, 0.0), ind(shock.begin(), shock.begin() + nc);
  std::vector<int32_t> conn, offs;
  std::vector<uint8_t> types((size_t)nc, 9);
  for (int c = 0; c < nc; ++c) {
    for (int v = 0; v < 4; ++v) {
      pts[(size_t)(4 * c + v) * 3] = mesh->cell_vertices[(size_t)c * 8 + 2 * v];
      pts[(size_t)(4 * c + v) * 3 + 1] = mesh->cell_vertices[(size_t)c * 8 + 2 * v + 1];
    }
    conn.insert(conn.end(), {4 * c, 4 * c + 1, 4 * c + 3, 4 * c + 2});   // lexicographic vertices -> VTK_QUAD order
    offs.push_back(4 * (c + 1));
  }
  std::ofstream f(path);
  if (!f) throw std::runtime_error("cannot write " + path);
  f << "<?xml version=\"1.0\" ?>\n<VTKFile type=\"UnstructuredGrid\" version=\"0.1\" compressor=\"vtkZLibDataCompressor\" byte_order=\"LittleEndian\">\n"
    << "<UnstructuredGrid>\n<Piece NumberOfPoints=\"" << 4 * nc << "\" NumberOfCells=\"" << nc << "\" >\n  <Points>\n";
  data_array(f, nullptr, pts, "Float64", 3);
  f << "  </Points>\n\n  <Cells>\n";
  data_array(f, "connectivity", conn, "Int32", 0);
  data_array(f, "offsets", offs, "Int32", 0);
  data_array(f, "types", types, "UInt8", 0);
  f << "  </Cells>\n  <CellData Scalars=\"scalars\">\n";
  data_array(f, "mu_shock", mu, "Float64", 0);
  data_array(f, "shock_indicator", ind, "Float64", 0);
  f << "  </CellData>\n </Piece>\n </UnstructuredGrid>\n</VTKFile>\n";
}

}  // namespace dflo_fe
