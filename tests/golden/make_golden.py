"""Writes the golden fixtures under tests/golden/.

flux_reference.json -- outputs of the REFERENCE's own pointwise flux functions (EulerEquations<2>::
  lxf_flux / steger_warming_flux / kfvs_flux / roe_flux / hllc_flux, src/equation.h:326-782) recorded
  in SURVEY.md section 8c (the survey compiled src/equation.h in place and printed these values at
  %.17g).  This script does not and cannot rebuild them: every reference source needs deal.II, which
  is absent, and writing stand-in headers is not allowed in this repo.  The numbers are data.
states.json -- closed-form left/right states the reference's example scripts print
  (examples/double_mach_reflection/state.py, examples/forward_step/state.py,
  examples/sod_shock_tube/state.m) and the constants the shipped .prm files hold; recomputed here from
  the same formulas (gamma = 1.4).
"""
import json
import math
import os

HERE = os.path.dirname(os.path.abspath(__file__))

flux = {
    "source": "SURVEY.md section 8c: reference src/equation.h compiled in place during the survey",
    "n": [0.6, 0.8],
    "W_l": [0.3, -0.1, 1.0, 2.5],
    "W_r": [0.1, 0.2, 0.8, 2.0],
    "note": "lxf uses the two states themselves as the cell averages (A = W)",
    "fluxes": {
        "hllc": [0.56356554474609344, 0.63384971213108432, 0.23514202839408069, 0.77569928971407975],
        "roe": [0.55198265321226647, 0.65560112468243237, 0.23525136198037883, 0.78991152103509832],
        "kfvs": [0.60906148254334247, 0.57038093312043603, 0.22820923366527962, 0.75065727460083909],
        "lxf": [0.70389356881873888, 0.51215964677189152, 0.30489356881873891, 0.9195151720468473],
        "sw": [0.61268699061203558, 0.55420489303520981, 0.22577695949921217, 0.82267930345993101],
    },
}
json.dump(flux, open(os.path.join(HERE, "flux_reference.json"), "w"), indent=1)

g = 1.4
th = 30.0 * math.pi / 180.0
dmr = {  # examples/double_mach_reflection/state.py
    "left": [8.0 * 8.25 * math.cos(th), -8.0 * 8.25 * math.sin(th), 8.0,
             116.5 / (g - 1) + 0.5 * 8.0 * (8.25 ** 2)],
    "right": [0.0, 0.0, 1.4, 1.0 / (g - 1)],
    "prm_left": [57.1576766498, -33.0, 8.0, 563.5],   # examples/double_mach_reflection/input.prm:35-62
    "prm_right": [0.0, 0.0, 1.4, 2.5],
}
fstep = {  # examples/forward_step/state.py: Mach 3, rho = gamma, p = 1
    "inflow": [g * 3.0, 0.0, g, 1.0 / (g - 1) + 0.5 * g * 9.0],
    "prm_inflow": [4.2, 0.0, 1.4, 8.8],               # examples/forward_step/input.prm:19-47
}
sod = {  # examples/sod_shock_tube/input.prm:39-44: rho 1|0.125, p 1|0.1 -> E = p/(g-1)
    "left": [0.0, 0.0, 1.0, 1.0 / (g - 1)], "right": [0.0, 0.0, 0.125, 0.1 / (g - 1)],
    "prm_left": [0.0, 0.0, 1.0, 2.5], "prm_right": [0.0, 0.0, 0.125, 0.25],
}
json.dump({"double_mach_reflection": dmr, "forward_step": fstep, "sod_shock_tube": sod},
          open(os.path.join(HERE, "states.json"), "w"), indent=1)
print("golden fixtures written")
