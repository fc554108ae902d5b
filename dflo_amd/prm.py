"""Reader of dflo's `input.prm` (deal.II ParameterHandler text format) for the keys the explicit path uses.

Schema and defaults follow Parameters::*::declare_parameters / parse_parameters (src/parameters.cc:10-296,
316-551): every key dflo declares is accepted (an undeclared key is an error, as in ParameterHandler); keys that
belong to the implicit solver, refinement or MOOD are parsed and kept but select nothing here, and a value that
asks for one of those subsystems is rejected with the reference's wording where it has one.
"""
import os

from . import _lib
from .expr import VectorFunction
from .params import Parameters

MAX_BOUNDARIES = _lib.MAX_BOUNDARIES
N_COMPONENTS = 4


class PrmError(ValueError):
    pass


def _schema():
    top = {
        "mesh type": ("gmsh", ("ucd", "gmsh")), "mesh file": ("grid.msh", None), "degree": ("1", int),
        "basis": ("Qk", ("Qk", "Pk")), "mapping": ("q1", ("q1", "q2", "cartesian")),
        "diffusion power": ("2.0", float), "diffusion coefficient": ("0.0", float), "gravity": ("0.0", float),
    }
    wvals = {"w_%d value" % c: ("0.0", None) for c in range(N_COMPONENTS)}
    sub = {
        "time stepping": {
            "stationary": ("false", bool), "cfl": ("0.0", float), "time step type": ("global", ("global", "local")),
            "time step": ("-1.0", float), "final time": ("1.0e20", float), "theta scheme value": ("1.0", float),
            "nonlinear iterations": ("1", int),
        },
        "initial condition": dict({"function": ("none", ("none", "rt", "isenvort", "vortsys"))}, **wvals),
        "linear solver": {
            "output": ("quiet", ("quiet", "verbose")), "method": ("rk3", ("gmres", "direct", "umfpack", "rk3", "mood")),
            "residual": ("1e-10", float), "max iters": ("300", int), "ilut fill": ("2", float),
            "ilut absolute tolerance": ("1e-9", float), "ilut relative tolerance": ("1.1", float),
            "ilut drop tolerance": ("1e-10", float),
        },
        "refinement": {
            "refinement": ("true", bool), "time step": ("1.0e20", float), "iter step": ("100000000", int),
            "refinement fraction": ("0.1", float), "unrefinement fraction": ("0.1", float),
            "max elements": ("1000000", float), "shock value": ("4.0", float), "shock levels": ("3.0", float),
        },
        "flux": {"flux": ("lxf", ("lxf", "sw", "kfvs", "roe", "hllc")), "stab": ("mesh", ("constant", "mesh")),
                 "stab value": ("1", float)},
        "limiter": {
            "shock indicator": ("limiter", ("limiter", "density", "energy", "u2")), "type": ("none", ("none", "TVB")),
            "characteristic limiter": ("false", bool), "positivity limiter": ("false", bool), "M": ("0", float),
            "beta": ("1.0", float), "conserve angular momentum": ("false", bool),
        },
        "output": {
            "schlieren plot": ("false", bool), "time step": ("1e20", float), "iter step": ("1000000", float),
            "format": ("vtk", ("vtk", "tecplot")), "compute angular momentum": ("10000000", float),
        },
    }
    for b in range(MAX_BOUNDARIES):
        # "periodic" with its "pair" / "direction" entries is the MPI variant's schema (src_mpi/parameters.cc:397-410),
        # which the shipped isentropic_vortex / vortex_system_pbc input files use
        sub["boundary_%d" % b] = dict({"type": ("outflow", ("slip", "inflow", "outflow", "pressure", "farfield", "periodic")),
                                       "pair": ("0", int), "direction": ("x", ("x", "y"))}, **wvals)
    return top, sub


def _convert(key, text, kind):
    text = text.strip()
    if kind is None:
        return text
    if kind is bool:
        if text not in ("true", "false"):
            raise PrmError("entry <%s>: %r is not a bool" % (key, text))
        return text == "true"
    if kind in (int, float):
        try:
            return kind(float(text)) if kind is int else float(text)
        except ValueError:
            raise PrmError("entry <%s>: %r is not a number" % (key, text))
    if text not in kind:
        raise PrmError("entry <%s>: %r is not one of %s" % (key, text, "|".join(kind)))
    return text


def parse_prm_text(text):
    """-> (top: dict, subsections: dict of dict) with defaults filled in and values converted."""
    top_s, sub_s = _schema()
    top = {k: v[0] for k, v in top_s.items()}
    sub = {s: {k: v[0] for k, v in d.items()} for s, d in sub_s.items()}
    current = None
    for ln, raw in enumerate(text.splitlines(), 1):
        line = raw.split("#", 1)[0].strip()
        if not line:
            continue
        words = line.split(None, 1)
        if words[0] == "subsection":
            if current is not None:
                raise PrmError("line %d: nested subsection" % ln)
            current = words[1].strip() if len(words) > 1 else ""
            if current not in sub:
                raise PrmError("line %d: no subsection <%s> was declared" % (ln, current))
        elif words[0] == "end":
            if current is None:
                raise PrmError("line %d: 'end' outside a subsection" % ln)
            current = None
        elif words[0] == "set":
            if len(words) < 2 or "=" not in words[1]:
                raise PrmError("line %d: expected 'set key = value'" % ln)
            key, value = words[1].split("=", 1)
            key = " ".join(key.split())
            target = top if current is None else sub[current]
            if key not in target:
                raise PrmError("line %d: no entry with name <%s> was declared%s" % (
                    ln, key, "" if current is None else " in subsection <%s>" % current))
            target[key] = value.strip()
        else:
            raise PrmError("line %d: cannot interpret %r" % (ln, raw.strip()))
    if current is not None:
        raise PrmError("subsection <%s> is not closed" % current)
    top = {k: _convert(k, v, top_s[k][1]) for k, v in top.items()}
    sub = {s: {k: _convert(k, v, sub_s[s][k][1]) for k, v in d.items()} for s, d in sub.items()}
    return top, sub


class InputDeck:
    """Parameters::AllParameters<2> for the explicit path: mesh description, Parameters, the boundary and
    initial-condition functions, the output cadence."""

    def __init__(self, text, directory="."):
        top, sub = parse_prm_text(text)
        self.directory = directory
        self.mesh_type = top["mesh type"]
        self.mesh_file = top["mesh file"]
        self.degree = top["degree"]
        self.basis = top["basis"]
        self.mapping = top["mapping"]
        ts, lim, out, sol = sub["time stepping"], sub["limiter"], sub["output"], sub["linear solver"]
        self.is_stationary = ts["stationary"]
        cfl, time_step, final_time = ts["cfl"], ts["time step"], ts["final time"]
        if self.is_stationary:   # src/parameters.cc:425-429 sets dt = 1, final time = 1e20 and compute_time_step returns at
            # once (src/claw.cc:449-450): the steady-state mode of the implicit solver, not part of the explicit path
            raise PrmError("stationary = true: steady-state runs belong to the implicit solver, which is not provided")
        elif not (cfl > 0 or time_step > 0):
            raise PrmError("cfl and time_step zero")
        if sol["method"] != "rk3":
            raise PrmError("linear solver method = %s: only the explicit rk3 path is provided" % sol["method"])
        if sub["refinement"]["refinement"] and self.basis == "Pk":
            raise PrmError("Refinement does not work for Pk basis")
        if sub["refinement"]["refinement"]:
            raise PrmError("refinement = true: grid adaptation is not part of the explicit device path (set refinement = false)")
        if lim["type"] == "TVB" and self.mapping != "cartesian":
            raise PrmError("TVB limiter works on cartesian grids only")
        if self.basis == "Pk" and self.mapping != "cartesian":
            raise PrmError("Pk basis can only be used with Cartesian grids")
        # (mapping = q2: MappingQ(2) on the straight-edged cells of a .msh file is the bilinear map; the engine takes it as q1)
        if top["diffusion coefficient"] != 0.0:
            raise PrmError("diffusion coefficient != 0: the shock-capturing term belongs to the implicit path")
        self.boundary_kind, self.boundary_values = {}, {}
        self.periodic_pairs = []   # (first id, second id, direction), each pair once (src_mpi/parameters.cc:524-560)
        for b in range(MAX_BOUNDARIES):
            s = sub["boundary_%d" % b]
            self.boundary_kind[b] = s["type"]
            if s["type"] == "periodic":
                self.boundary_kind[b] = "outflow"   # no face keeps this id once the pair is connected
                if not any((p[0], p[1]) in ((b, s["pair"]), (s["pair"], b)) for p in self.periodic_pairs):
                    self.periodic_pairs.append((b, s["pair"], s["direction"]))
            self.boundary_values[b] = VectorFunction([s["w_%d value" % c] for c in range(N_COMPONENTS)], ("x", "y", "t"))
        ic = sub["initial condition"]
        self.ic_function = ic["function"]
        self.initial_conditions = VectorFunction([ic["w_%d value" % c] for c in range(N_COMPONENTS)], ("x", "y"))
        self.parameters = Parameters(
            flux=sub["flux"]["flux"], limiter=lim["type"], char_lim=lim["characteristic limiter"],
            pos_lim=lim["positivity limiter"], cfl=cfl, time_step=time_step, final_time=final_time, M=lim["M"],
            beta=lim["beta"], gravity=top["gravity"], time_step_type=ts["time step type"],
            boundary=self.boundary_kind, shock_indicator=lim["shock indicator"],
            conserve_angular_momentum=lim["conserve angular momentum"])
        self.schlieren_plot = out["schlieren plot"]
        self.output_time_step = out["time step"]
        self.output_iter_step = int(out["iter step"])
        self.output_format = out["format"]
        self.ang_mom_step = int(out["compute angular momentum"])
        self.sections = sub
        self.top = top

    @staticmethod
    def read(path):
        with open(path) as f:
            return InputDeck(f.read(), os.path.dirname(os.path.abspath(path)))

    @property
    def mesh_path(self):
        return self.mesh_file if os.path.isabs(self.mesh_file) else os.path.join(self.directory, self.mesh_file)
