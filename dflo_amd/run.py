"""Mini-driver: ConservationLaw<2>::run() (src/claw.cc:955-1129) for the explicit rk3 path on the device engine.

    python -m dflo_amd input.prm [--outdir DIR] [--max-steps N] [--fast] [--quiet]

reads the same `input.prm` keys as dflo (dflo_amd/prm.py), the Gmsh `.msh` the file names, sets and limits the
initial condition, runs the time loop with the reference's output cadence and writes solution-NNN.vtu /
shock.vtu (dflo_amd/vtu.py).  Everything here is host-side set-up and I/O; the time loop body is the C ABI.
"""
import argparse
import os
import sys

import numpy as np

from . import problems
from .claw import ConservationLaw
from .expr import ExpressionError
from .mesh import Mesh
from .prm import InputDeck
from .vtu import write_shock_tecplot, write_shock_vtu, write_tecplot, write_vtu


class Run:
    def __init__(self, deck, outdir=".", quiet=False, device=0, host_bc=False):
        self.deck, self.outdir, self.quiet = deck, outdir, quiet
        os.makedirs(outdir, exist_ok=True)
        if deck.mesh_type != "gmsh":
            raise ValueError("mesh type = ucd is not provided (gmsh)")
        mesh = Mesh.read_gmsh(deck.mesh_path, deck.degree, deck.mapping)   # src/claw.cc:957-967
        mesh.set_basis(deck.basis)
        for first, second, direction in deck.periodic_pairs:
            mesh.make_periodic(first, second, direction)
        self.mesh = mesh
        self.claw = ConservationLaw(mesh, deck.parameters, device=device)
        self.cell, self.face, self.bid, self.bxy = self.claw.boundary_faces()
        self.bc_time_dependent = any(deck.boundary_values[int(b)].time_dependent for b in np.unique(self.bid))
        # time-dependent boundary functions are handed to the device as postfix programs (evaluated at t / t + dt of
        # every step from the device clock); an expression the device evaluator lacks falls back to host evaluation
        self.bc_on_device = False
        if self.bc_time_dependent and not host_bc:
            try:
                for b in np.unique(self.bid):
                    if deck.boundary_values[int(b)].time_dependent:
                        self.claw.set_boundary_function(int(b), deck.boundary_values[int(b)].expressions)
                self.bc_on_device = True
            except ExpressionError:
                for b in np.unique(self.bid):
                    self.claw.set_boundary_function(int(b), [None] * 4)
        self.output_file_number = 0
        self.time_iter = 0

    def log(self, *a):
        if not self.quiet:
            print(*a)
            sys.stdout.flush()

    # ---- set_initial_condition (src/ic.cc:104-181)
    def initial_condition(self):
        d = self.deck
        fn = {"none": d.initial_conditions,
              "rt": lambda x, y: problems.rayleigh_taylor(x, y, gravity=d.parameters.gravity),
              "isenvort": problems.isentropic_vortex, "vortsys": problems.vortex_system}[d.ic_function]
        return self.mesh.interpolate(fn)   # interpolation for Qk, L2 projection for Pk

    # ---- boundary values at the face quadrature points (integrate_boundary_term_explicit, src/assemble_explicit.cc:161-165)
    def set_boundary_values(self, t, which):
        bv = np.zeros(self.bxy.shape[:2] + (4,))
        for b in np.unique(self.bid):
            sel = self.bid == b
            w = self.deck.boundary_values[int(b)](self.bxy[sel, :, 0], self.bxy[sel, :, 1], t)
            bv[sel] = np.stack(w, axis=-1)
        self.claw.set_boundary_values(which, bv)

    # ---- output_results (src/output.cc:33-87)
    def output_results(self):
        tec = self.deck.output_format == "tecplot"
        name = "solution-%03d.%s" % (self.output_file_number, "plt" if tec else "vtu")
        self.log("Writing file " + name)
        u, ind = self.claw.current_solution, self.claw.shock_indicator
        if tec:
            write_tecplot(os.path.join(self.outdir, name), self.mesh, u, schlieren=self.deck.schlieren_plot)
            write_shock_tecplot(os.path.join(self.outdir, "shock.plt"), self.mesh, ind)
        else:
            write_vtu(os.path.join(self.outdir, name), self.mesh, u, time=self.claw.elapsed_time,
                      cycle=self.output_file_number, schlieren=self.deck.schlieren_plot)
            write_shock_vtu(os.path.join(self.outdir, "shock.vtu"), self.mesh, ind)
        self.output_file_number += 1

    def run(self, max_steps=None, fast=False):
        d, claw = self.deck, self.claw
        self.log("Number of active cells:       %d" % self.mesh.n_cells)
        self.log("Number of degrees of freedom: %d" % (self.mesh.n_cells * self.mesh.ndof))
        if len(self.bid):
            self.set_boundary_values(0.0, 0)
            self.set_boundary_values(0.0, 1)
        claw.set_initial_condition(self.initial_condition())   # + compute_cell_average, src/claw.cc:997
        claw.apply_limiter()                                   # compute_shock_indicator(); apply_limiter(); :1000-1002
        claw.elapsed_time = 0.0
        self.output_results()
        next_output_time = claw.elapsed_time + d.output_time_step
        next_output_iter = self.time_iter + d.output_iter_step
        final_time = d.parameters.final_time
        res_norm0 = res_norm = 1.0
        while claw.elapsed_time < final_time and (max_steps is None or self.time_iter < max_steps):
            chunk = 1
            if fast and (self.bc_on_device or not self.bc_time_dependent) and d.output_time_step >= 1e19:
                chunk = max(1, min(next_output_iter - self.time_iter, 64,
                                   (max_steps - self.time_iter) if max_steps is not None else 64))
                if chunk > 1 and final_time < 1e19:
                    # a chunk must not run past final_time (the loop of src/claw.cc:1026 stops there): size it by the
                    # current time step with a margin for its growth; the last steps are taken one by one
                    dt_now = claw.compute_time_step()
                    chunk = max(1, min(chunk, int(0.8 * (final_time - claw.elapsed_time) / dt_now))) if dt_now > 0 else 1
            if chunk > 1:
                claw.advance(chunk)                            # dt and time stay on the device
                self.time_iter += chunk
                self.log("It=%d, T=%.12g" % (self.time_iter, claw.elapsed_time))
            else:
                dt = claw.compute_time_step()                  # :1029
                self.log("\nIt=%d, T=%.12g, dt=%.12g, cfl=%g" % (self.time_iter + 1, claw.elapsed_time + dt, dt, d.parameters.cfl))
                if self.bc_time_dependent and not self.bc_on_device:   # bc_time of stage 0 / later stages, :736-745
                    self.set_boundary_values(claw.elapsed_time, 0)
                    self.set_boundary_values(claw.elapsed_time + dt, 1)
                res_norm0, res_norm = claw.iterate_explicit(dt)    # :1051, advances elapsed_time
                self.log("   %-16.3e %-16.3e" % (res_norm0, res_norm))
                self.time_iter += 1
            if chunk == 1 and self.time_iter % d.ang_mom_step == 0:    # :1074-1075
                self.log("Total angular momentum: %18.8e %24.14e" % (claw.elapsed_time, self.mesh.angular_momentum(claw.current_solution)))
            if (claw.elapsed_time >= next_output_time or self.time_iter >= next_output_iter or
                    abs(claw.elapsed_time - final_time) < 1.0e-13):   # :1091-1099
                self.output_results()
                next_output_time = claw.elapsed_time + d.output_time_step
                next_output_iter = self.time_iter + d.output_iter_step
        return res_norm0, res_norm


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m dflo_amd", description="explicit DG Euler solver (dflo input files) on MI355X")
    ap.add_argument("input", help="input.prm")
    ap.add_argument("n_threads", nargs="?", default=None, help="accepted for command-line compatibility with dflo (src/main.cc:22-27); unused")
    ap.add_argument("--outdir", default=".")
    ap.add_argument("--max-steps", type=int, default=None)
    ap.add_argument("--fast", action="store_true", help="advance in chunks without a host round trip per step")
    ap.add_argument("--quiet", action="store_true")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--host-bc", action="store_true", help="evaluate time-dependent boundary functions on the host every step")
    a = ap.parse_args(argv)
    try:
        Run(InputDeck.read(a.input), a.outdir, a.quiet, a.device, a.host_bc).run(a.max_steps, a.fast)
    except Exception as e:   # src/main.cc:56-78: report and exit code 1
        sys.stderr.write("\n----------------------------------------------------\nException on processing:\n%s\nAborting!\n"
                         "----------------------------------------------------\n" % e)
        return 1
    return 0
