"""Derivation of tests/golden/closed_forms.json: known answers for the pointwise layer of the hot path, computed HERE,
independently of oracle/ and of the HIP kernels, in 60-digit decimal arithmetic from the formulas of the reference
(src/equation.h; each function below cites the lines it restates in mathematical form) and, wherever the answer has one,
from a closed form a reader can check by hand:

  consistent     F^(W, W, n) = F(W).n                                       every flux, Sod / double-Mach / step states
  upwind         both states supersonic along n:  F^ = F(W_l).n (or F(W_r).n)   HLLC, Roe, Steger-Warming exactly
  contact        stationary contact (u = 0, p equal, rho jumps): HLLC and Roe mass flux = 0, momentum flux = p n
  wall           slip wall with zero normal velocity: F^ = (p n_x, p n_y, 0, 0)
  ghost states   compute_Wminus for inflow / outflow / slip / pressure / farfield (src/equation.h:942-1033)
  L R = I        the eigenvector matrices of the characteristic limiter at the Sod states

The reference itself cannot be run here (every source needs deal.II), so these vectors pin the
formulas, not the reference binary: `parity` stays "partial" until a deal.II build exists.
Usage: python tests/golden/make_closed_forms.py   (rewrites closed_forms.json; deterministic)
"""
import json
import os
from decimal import Decimal as D, getcontext

getcontext().prec = 60
HERE = os.path.dirname(os.path.abspath(__file__))
G = D("1.4")        # gas_gamma, src/equation.cc:33
PI = D("3.14159265358979323846264338327950288419716939937510582097494")   # M_PI is a double: see kfvs()


def sqrt(x):
    return x.sqrt()


def dexp(x):
    return x.exp()


def prim(W):
    mx, my, rho, E = W
    u, v = mx / rho, my / rho
    p = (G - 1) * (E - (mx * mx + my * my) / (2 * rho))          # src/equation.h:87-92
    return rho, u, v, p


def cons(rho, u, v, p):
    return [rho * u, rho * v, rho, p / (G - 1) + rho * (u * u + v * v) / 2]


def physical_flux(W, n):
    """F(W).n: (m u_n + p n, rho u_n, (E + p) u_n), src/equation.h:170-220"""
    rho, u, v, p = prim(W)
    un = u * n[0] + v * n[1]
    return [W[0] * un + p * n[0], W[1] * un + p * n[1], rho * un, (W[3] + p) * un]


def lxf(n, Wl, Wr, Al, Ar):
    """src/equation.h:326-377: central flux + lambda/2 (W+ - W-), lambda = max over the two CELL AVERAGES of |u_n| + c"""
    def lam(A):
        rho, u, v, p = prim(A)
        return abs(u * n[0] + v * n[1]) + sqrt(G * p / rho)
    lm = max(lam(Al), lam(Ar))
    fl, fr = physical_flux(Wl, n), physical_flux(Wr, n)
    return [(fl[c] + fr[c]) / 2 + lm * (Wl[c] - Wr[c]) / 2 for c in range(4)]


def steger_warming(n, Wl, Wr):
    """src/equation.h:383-460: F+ (W_l) + F- (W_r) with the eigenvalues split by max(., 0) / min(., 0)"""
    def half(W, pick):
        rho, u, v, p = prim(W)
        un, q2, c = u * n[0] + v * n[1], u * u + v * v, sqrt(G * p / rho)
        l1, l2, l3 = pick(un), pick(un + c), pick(un - c)
        a = 2 * (G - 1) * l1 + l2 + l3
        f = rho / (2 * G)
        return [f * (a * u + c * (l2 - l3) * n[0]), f * (a * v + c * (l2 - l3) * n[1]), f * a,
                f * (a * q2 / 2 + c * un * (l2 - l3) + c * c * (l2 + l3) / (G - 1))]
    fp, fm = half(Wl, lambda x: max(x, D(0))), half(Wr, lambda x: min(x, D(0)))
    return [fp[c] + fm[c] for c in range(4)]


def erf_as(x):
    """A&S 7.1.26 with the reference's constants (src/equation.h:688-709) -- the reference does NOT call erf()"""
    a1, a2, a3, a4, a5, p = D("0.254829592"), D("-0.284496736"), D("1.421413741"), D("-1.453152027"), D("1.061405429"), D("0.3275911")
    s = -1 if x < 0 else 1
    x = abs(x)
    t = 1 / (1 + p * x)
    return s * (1 - (((((a5 * t + a4) * t) + a3) * t + a2) * t + a1) * t * dexp(-x * x))


def kfvs(n, Wl, Wr):
    """src/equation.h:714-782: kinetic split fluxes of Deshpande and Mandal"""
    def split(sign, W):
        rho, u, v, p = prim(W)
        un = u * n[0] + v * n[1]
        beta = rho / (2 * p)
        s = un * sqrt(beta)
        A = (1 + sign * erf_as(s)) / 2
        B = sign * dexp(-s * s) / (2 * sqrt(PI * beta))
        uf = un * A + B
        return [p * n[0] * A + W[0] * uf, p * n[1] * A + W[1] * uf, rho * uf, (W[3] + p) * un * A + (W[3] + p / 2) * B]
    a, b = split(+1, Wl), split(-1, Wr)
    return [a[c] + b[c] for c in range(4)]


def roe_average(Wl, Wr, n):
    rl, ul, vl, pl = prim(Wl)
    rr, ur, vr, pr = prim(Wr)
    fl = sqrt(rl) / (sqrt(rl) + sqrt(rr))
    fr = 1 - fl
    u, v = ul * fl + ur * fr, vl * fl + vr * fr
    hl = G * pl / rl / (G - 1) + (ul * ul + vl * vl) / 2
    hr = G * pr / rr / (G - 1) + (ur * ur + vr * vr) / 2
    h = hl * fl + hr * fr
    c = sqrt((G - 1) * (h - (u * u + v * v) / 2))
    return dict(rl=rl, ul=ul, vl=vl, pl=pl, rr=rr, ur=ur, vr=vr, pr=pr, u=u, v=v, h=h, c=c, hl=hl, hr=hr,
                un=u * n[0] + v * n[1], unl=ul * n[0] + vl * n[1], unr=ur * n[0] + vr * n[1], rho=sqrt(rl) * sqrt(rr))


def roe(n, Wl, Wr):
    """src/equation.h:466-560: Roe's flux difference splitting, Harten's entropy fix with delta = 0.1 c on the acoustic waves"""
    a = roe_average(Wl, Wr, n)
    c, un = a["c"], a["un"]
    drho, dp, dvn = a["rr"] - a["rl"], a["pr"] - a["pl"], a["unr"] - a["unl"]
    a1 = (dp - a["rho"] * c * dvn) / (2 * c * c)
    a2 = drho - dp / (c * c)
    a3 = (dp + a["rho"] * c * dvn) / (2 * c * c)
    l1, l2, l3 = abs(un - c), abs(un), abs(un + c)
    delta = c / 10
    fixed = [l1 < delta, l3 < delta]
    if l1 < delta:
        l1 = (l1 * l1 / delta + delta) / 2
    if l3 < delta:
        l3 = (l3 * l3 / delta + delta) / 2
    vel, dv = [a["u"], a["v"]], [a["ur"] - a["ul"], a["vr"] - a["vl"]]
    v2 = a["u"] ** 2 + a["v"] ** 2
    vdv = vel[0] * dv[0] + vel[1] * dv[1]
    Dm = [(vel[d] - n[d] * c) * l1 * a1 + vel[d] * l2 * a2 + (dv[d] - n[d] * dvn) * l2 * a["rho"] + (vel[d] + n[d] * c) * l3 * a3
          for d in range(2)]
    Drho = l1 * a1 + l2 * a2 + l3 * a3
    DE = l1 * a1 * (a["h"] - c * un) + l2 * a2 * v2 / 2 + l2 * a["rho"] * (vdv - un * dvn) + l3 * a3 * (a["h"] + c * un)
    pavg = (a["pl"] + a["pr"]) / 2
    F = [n[d] * pavg + (Wl[d] * a["unl"] + Wr[d] * a["unr"]) / 2 - Dm[d] / 2 for d in range(2)]
    F.append((a["rl"] * a["unl"] + a["rr"] * a["unr"] - Drho) / 2)
    F.append((a["rl"] * a["hl"] * a["unl"] + a["rr"] * a["hr"] * a["unr"] - DE) / 2)
    return F, fixed


def hllc(n, Wl, Wr):
    """src/equation.h:566-681 (SU2's HLLC): S_l = min(u~_n - c~, u_l - c_l), S_r = max(u~_n + c~, u_r + c_r), S_m, p*"""
    a = roe_average(Wl, Wr, n)
    cl, cr = sqrt(G * a["pl"] / a["rl"]), sqrt(G * a["pr"] / a["rr"])
    sl, sr = min(a["un"] - a["c"], a["unl"] - cl), max(a["un"] + a["c"], a["unr"] + cr)
    sm = (a["pl"] - a["pr"] - a["rl"] * a["unl"] * (sl - a["unl"]) + a["rr"] * a["unr"] * (sr - a["unr"])) / (
        a["rr"] * (sr - a["unr"]) - a["rl"] * (sl - a["unl"]))
    ps = a["rr"] * (a["unr"] - sr) * (a["unr"] - sm) + a["pr"]

    def star(W, rho, vel, un, p, s):
        inv = 1 / (s - sm)
        rs = rho * (s - un) * inv
        m = [(rho * vel[d] * (s - un) + (ps - p) * n[d]) * inv for d in range(2)]
        e = ((s - un) * W[3] - p * un + ps * sm) * inv
        return [m[0] * sm + ps * n[0], m[1] * sm + ps * n[1], rs * sm, (e + ps) * sm]
    if sm >= 0:
        if sl > 0:
            return physical_flux(Wl, n), "left"
        return star(Wl, a["rl"], [a["ul"], a["vl"]], a["unl"], a["pl"], sl), "star-left"
    if sr >= 0:
        return star(Wr, a["rr"], [a["ur"], a["vr"]], a["unr"], a["pr"], sr), "star-right"
    return physical_flux(Wr, n), "right"


def compute_Wminus(kind, n, Wp, bv):
    """src/equation.h:942-1033"""
    if kind in ("inflow", "farfield"):
        return list(bv)
    if kind == "outflow":
        return list(Wp)
    if kind == "pressure":   # w_3 of the boundary data is read as a PRESSURE (src/equation.h:992)
        ke = (Wp[0] * Wp[0] + Wp[1] * Wp[1]) / (2 * Wp[2])
        return [Wp[0], Wp[1], Wp[2], bv[3] / (G - 1) + ke]
    if kind == "slip":
        mn = Wp[0] * n[0] + Wp[1] * n[1]
        return [Wp[0] - 2 * mn * n[0], Wp[1] - 2 * mn * n[1], Wp[2], Wp[3]]
    raise KeyError(kind)


def eigen(W):
    """compute_eigen_matrix (src/equation.h:232-271): right / left eigenvectors of the x and y flux Jacobians in the
    ordering of the reference (rows = characteristic variables)"""
    rho, u, v, p = prim(W)
    c = sqrt(G * p / rho)
    h = c * c / (G - 1) + (u * u + v * v) / 2
    q2, g1 = (u * u + v * v) / 2, G - 1
    # conserved ordering inside the matrices: (rho, mx, my, E) -- transform_to_char/_con permute, src/equation.h:273-305
    Rx = [[1, 1, 0, 1], [u - c, u, 0, u + c], [v, v, -1, v], [h - c * u, q2, -v, h + c * u]]
    Lx = [[(g1 * q2 + c * u) / (2 * c * c), (-g1 * u - c) / (2 * c * c), -g1 * v / (2 * c * c), g1 / (2 * c * c)],
          [1 - g1 * q2 / (c * c), g1 * u / (c * c), g1 * v / (c * c), -g1 / (c * c)],
          [v, 0, -1, 0],
          [(g1 * q2 - c * u) / (2 * c * c), (-g1 * u + c) / (2 * c * c), -g1 * v / (2 * c * c), g1 / (2 * c * c)]]
    return Rx, Lx


def S(x):
    return [format(v, ".25e") for v in x]


def main():
    n10, n01, nob = [D(1), D(0)], [D(0), D(1)], [D("0.6"), D("0.8")]
    neg = lambda n: [-n[0], -n[1]]
    sodL, sodR = cons(D(1), D(0), D(0), D(1)), cons(D("0.125"), D(0), D(0), D("0.1"))      # examples/sod_shock_tube/input.prm
    th = PI / 6
    # cos(30 deg) = sqrt(3)/2, sin = 1/2 exactly (examples/double_mach_reflection/state.py)
    dmrL = cons(D(8), D("8.25") * sqrt(D(3)) / 2, -D("8.25") / 2, D("116.5"))
    dmrR = cons(D("1.4"), D(0), D(0), D(1))
    step = cons(D("1.4"), D(3), D(0), D(1))                                                   # examples/forward_step/state.py
    states = {"sod_left": sodL, "sod_right": sodR, "dmr_post": dmrL, "dmr_pre": dmrR, "step_inflow": step}
    out = {"gamma": "1.4", "states": {k: S(v) for k, v in states.items()}, "flux_cases": [], "consistency": [], "wminus": [],
           "wall": [], "eigen": []}
    # ---- two-state cases: all five fluxes (lxf with A = W)
    cases = [
        ("sod, x normal", n10, sodL, sodR),
        ("sod reversed, -x normal", neg(n10), sodR, sodL),
        ("sod, oblique normal", nob, sodL, sodR),
        ("double Mach: post-shock | pre-shock, x normal", n10, dmrL, dmrR),
        ("double Mach: pre-shock | post-shock, oblique normal", nob, dmrR, dmrL),
        ("both supersonic to the right (Mach 3 | Mach 2.5)", n10, step, cons(D("1.1"), D("2.8"), D("0.3"), D("0.9"))),
        ("both supersonic to the left", n10, cons(D("1.1"), D("-2.8"), D("0.3"), D("0.9")), cons(D("1.4"), D(-3), D(0), D(1))),
        ("both supersonic along an oblique normal", nob, cons(D(1), D("2.0"), D("2.2"), D(1)), cons(D("0.9"), D("1.7"), D("1.9"), D("0.8"))),
        ("transonic: u_n - c inside the entropy fix of the left-running wave", n10, cons(D(1), D("1.15"), D("0.2"), D(1)), cons(D("0.9"), D("1.25"), D("0.1"), D("0.9"))),
        ("transonic: u_n + c inside the entropy fix of the right-running wave", n10, cons(D(1), D("-1.15"), D("0.2"), D(1)), cons(D("0.9"), D("-1.1"), D("0.1"), D("0.95"))),
        ("stationary contact (u = 0, equal pressure)", n10, cons(D(1), D(0), D(0), D(1)), cons(D("0.25"), D(0), D(0), D(1))),
        ("moving contact, subsonic to the right (HLLC star-left)", nob, cons(D(1), D("0.3"), D("0.4"), D(1)), cons(D("0.5"), D("0.3"), D("0.4"), D(1))),
        ("subsonic to the left (HLLC star-right)", n10, cons(D(1), D("-0.5"), D("0.1"), D(1)), cons(D("1.2"), D("-0.4"), D("-0.2"), D("1.3"))),
        ("strong jump: pressure ratio 1e5", n10, cons(D(1), D(0), D(0), D(1000)), cons(D(1), D(0), D(0), D("0.01"))),
        ("near vacuum on the right", n01, cons(D(1), D("0.1"), D("0.5"), D(1)), cons(D("1e-6"), D("0.1"), D("0.5"), D("1e-6"))),
        ("the survey's tuple", nob, [D("0.3"), D("-0.1"), D(1), D("2.5")], [D("0.1"), D("0.2"), D("0.8"), D(2)]),
    ]
    for name, n, Wl, Wr in cases:
        r, fixed = roe(n, Wl, Wr)
        hf, branch = hllc(n, Wl, Wr)
        rec = {"name": name, "n": S(n), "W_l": S(Wl), "W_r": S(Wr), "hllc_branch": branch, "roe_entropy_fix": fixed,
               "lxf": S(lxf(n, Wl, Wr, Wl, Wr)), "sw": S(steger_warming(n, Wl, Wr)), "kfvs": S(kfvs(n, Wl, Wr)), "roe": S(r), "hllc": S(hf)}
        # hand-checkable closed forms
        a = roe_average(Wl, Wr, n)
        cl, cr = sqrt(G * a["pl"] / a["rl"]), sqrt(G * a["pr"] / a["rr"])
        if a["unl"] > cl and a["unr"] > cr and a["un"] > a["c"]:
            rec["upwind"] = S(physical_flux(Wl, n))     # HLLC, Roe, Steger-Warming give exactly this
        if a["unl"] < -cl and a["unr"] < -cr and a["un"] < -a["c"]:
            rec["upwind"] = S(physical_flux(Wr, n))
        out["flux_cases"].append(rec)
    branches = {c["hllc_branch"] for c in out["flux_cases"]}
    assert branches == {"left", "star-left", "star-right", "right"}, branches
    assert any(c["roe_entropy_fix"][0] for c in out["flux_cases"]) and any(c["roe_entropy_fix"][1] for c in out["flux_cases"])
    # the stationary contact by hand: mass flux 0, momentum flux p n, energy flux 0 for HLLC and Roe
    sc = [c for c in out["flux_cases"] if c["name"].startswith("stationary contact")][0]
    for f in ("hllc", "roe"):
        v = [D(x) for x in sc[f]]
        assert abs(v[0] - 1) < D("1e-50") and abs(v[1]) < D("1e-50") and abs(v[2]) < D("1e-50") and abs(v[3]) < D("1e-50"), (f, v)
    # ---- consistency: F^(W, W, n) = F(W).n
    for sname, W in states.items():
        for n in (n10, neg(n01), nob):
            out["consistency"].append({"state": sname, "n": S(n), "W": S(W), "flux": S(physical_flux(W, n))})
            for f in (lxf(n, W, W, W, W), steger_warming(n, W, W), roe(n, W, W)[0], hllc(n, W, W)[0]):
                assert max(abs(f[c] - physical_flux(W, n)[c]) for c in range(4)) < D("1e-45")
            k = kfvs(n, W, W)   # consistent up to the error of the A&S polynomial (|erf error| <= 1.5e-7): A+ + A- = 1, B+ + B- = 0 hold exactly
            assert max(abs(k[c] - physical_flux(W, n)[c]) for c in range(4)) < D("1e-45")
    # ---- ghost states and the flux they give (HLLC; the boundary LxF uses the interior average on both sides)
    Wp = cons(D("1.1"), D("0.7"), D("-0.3"), D("1.2"))
    bv = [D("0.4"), D("0.1"), D("0.9"), D("2.2")]
    for kind in ("inflow", "outflow", "slip", "pressure", "farfield"):
        for n in (n10, neg(n01), nob):
            Wm = compute_Wminus(kind, n, Wp, bv)
            out["wminus"].append({"kind": kind, "n": S(n), "W_plus": S(Wp), "boundary_values": S(bv), "W_minus": S(Wm),
                                  "hllc": S(hllc(n, Wp, Wm)[0]), "roe": S(roe(n, Wp, Wm)[0]), "lxf_interior_average": S(lxf(n, Wp, Wm, Wp, Wp))})
    # ---- slip wall, zero normal velocity: only the pressure acts, whatever the flux
    for n in (n10, n01, nob):
        t = [-n[1], n[0]]
        W = cons(D("1.3"), D("0.8") * t[0], D("0.8") * t[1], D("0.7"))
        Wm = compute_Wminus("slip", n, W, W)
        assert Wm == W
        out["wall"].append({"n": S(n), "W": S(W), "flux": S([D("0.7") * n[0], D("0.7") * n[1], D(0), D(0)])})
        for f in (lxf(n, W, Wm, W, W), steger_warming(n, W, Wm), roe(n, W, Wm)[0], hllc(n, W, Wm)[0], kfvs(n, W, Wm)):
            assert max(abs(f[c] - [D("0.7") * n[0], D("0.7") * n[1], D(0), D(0)][c]) for c in range(4)) < D("1e-45")
    # ---- eigenvector matrices at the Sod and post-shock states: L R = I
    for sname in ("sod_left", "sod_right", "dmr_post"):
        Rx, Lx = eigen(states[sname])
        for i in range(4):
            for j in range(4):
                s = sum(D(Lx[i][k]) * D(Rx[k][j]) for k in range(4))
                assert abs(s - (1 if i == j else 0)) < D("1e-45"), (sname, i, j, s)
        out["eigen"].append({"state": sname, "W": S(states[sname])})
    json.dump(out, open(os.path.join(HERE, "closed_forms.json"), "w"), indent=1)
    print("closed_forms.json: %d two-state cases, %d consistency, %d ghost-state, %d wall records" % (
        len(out["flux_cases"]), len(out["consistency"]), len(out["wminus"]), len(out["wall"])))


if __name__ == "__main__":
    main()
