"""CPU: the inequality the stage kernel's limiter marks rest on (dflo_amd/csrc/stage_kernels.hpp: limiter_marks_from_box).

A cell is left unmarked -- the limiter pass never looks at it -- when a bound on its characteristic slopes, formed from a box
around its nodal values alone, is below the thresholds of the TVB limiter (src/limiter.cc:15-30, 283-289, 347).  The bound:
  sum_i |(L D)_i|  <=  kappa * K * sum_c (hi_c - lo_c)      in either direction,
with D = "dx * gradient of the cell average" of the nodal values (conserved variables), L the left eigenvectors at the cell
average (src/limiter.cc:307-340 via equation.h), K = 1/2 sum_m |l_m(1) - l_m(0)| and kappa a bound of ||L||_1 over the box.
Checked here by brute force: random admissible states, random nodal values inside a random box, degrees 1-5, both directions.
A restatement of the device formulas in numpy (test infrastructure; nothing here runs on the product path)."""
import numpy as np
import pytest

GAMMA, G1 = 1.4, 0.4
MX, MY, RHO, EN = 0, 1, 2, 3   # component order of dflo (src/equation.h)


def gauss01(n):
    x, w = np.polynomial.legendre.leggauss(n)
    return 0.5 * (x + 1.0), 0.5 * w


def lagrange_at(xn, x):
    out = np.ones(len(xn))
    for m in range(len(xn)):
        for j in range(len(xn)):
            if j != m:
                out[m] *= (x - xn[j]) / (xn[m] - xn[j])
    return out


def left_eigenvectors(A, direction):
    """rows of L as physics.hpp: to_char applies them (characteristic variables in the reference's order), columns in the
    order (rho, mx, my, E)"""
    rho = A[RHO]
    u, v = A[MX] / rho, A[MY] / rho
    q2 = u * u + v * v
    p = G1 * (A[EN] - 0.5 * rho * q2)
    c2 = GAMMA * p / rho
    c = np.sqrt(c2)
    ic2, beta, phi2 = 1.0 / c2, 0.5 / c2, 0.5 * G1 * q2
    un = u if direction == 0 else v
    r0 = [1 - phi2 * ic2, G1 * u * ic2, G1 * v * ic2, -G1 * ic2]
    if direction == 0:
        r1 = [v, 0.0, -1.0, 0.0]
        r2 = [beta * (phi2 - c * un), beta * (c - G1 * u), -beta * G1 * v, beta * G1]
        r3 = [beta * (phi2 + c * un), -beta * (c + G1 * u), -beta * G1 * v, beta * G1]
    else:
        r1 = [-u, 1.0, 0.0, 0.0]
        r2 = [beta * (phi2 - c * un), -beta * G1 * u, beta * (c - G1 * v), beta * G1]
        r3 = [beta * (phi2 + c * un), -beta * G1 * u, -beta * (c + G1 * v), beta * G1]
    return np.array([r0, r1, r2, r3])


def kappa_from_box(lo, hi, sn):
    """the device's bound of ||L||_1 (limiter_marks_from_box), or None when the box is not 'settled' (no bound then: the
    kernel forms the slopes themselves)"""
    rho_lo = lo[RHO] - (hi[RHO] - lo[RHO]) * sn
    e_lo = lo[EN] - (hi[EN] - lo[EN]) * sn
    dmx, dmy = (hi[MX] - lo[MX]) * sn, (hi[MY] - lo[MY]) * sn
    mxa = max(abs(lo[MX] - dmx), abs(hi[MX] + dmx))
    mya = max(abs(lo[MY] - dmy), abs(hi[MY] + dmy))
    if not rho_lo > 0:
        return None
    p_lo = G1 * (e_lo - 0.5 * (mxa * mxa + mya * mya) / rho_lo)
    if not (rho_lo >= 1e-10 + 1e-8 * hi[RHO] and p_lo >= 1e-10 + 1e-8 * abs(hi[EN])):
        return None
    q = (mxa + mya) / rho_lo
    ic2 = hi[RHO] / (GAMMA * p_lo)
    return 1.0 + q + ic2 * (0.4 * q * q + 0.8 * q + 0.8) + 0.5 * (q + 1.0) * (1.0 + ic2)


@pytest.mark.parametrize("N", [2, 3, 4, 5, 6])
def test_characteristic_slopes_stay_below_the_box_bound(N):
    rng = np.random.default_rng(1000 + N)
    xn, w = gauss01(N)
    L0, L1 = lagrange_at(xn, 0.0), lagrange_at(xn, 1.0)
    g = L1 - L0
    assert abs(g.sum()) < 1e-12          # the weights of the slope sum to zero
    K = 0.5 * np.abs(g).sum()
    worst = 0.0
    n_bound = 0
    for trial in range(4000):
        rho = 10.0 ** rng.uniform(-1.3, 1.0)
        c = 10.0 ** rng.uniform(-1.5, 1.0)
        mach = rng.uniform(0.0, 8.0) if trial % 3 else 0.0
        ang = rng.uniform(0.0, 2 * np.pi)
        u, v = mach * c * np.cos(ang), mach * c * np.sin(ang)
        p = rho * c * c / GAMMA
        centre = np.array([rho * u, rho * v, rho, p / G1 + 0.5 * rho * (u * u + v * v)])
        half = np.abs(centre).max() * 10.0 ** rng.uniform(-9, -0.7) * rng.uniform(0.0, 1.0, 4)
        # nodal values anywhere in the box, corners included
        U = centre[:, None, None] + half[:, None, None] * rng.choice([-1.0, 1.0, rng.uniform(-1, 1)], size=(4, N, N))
        lo, hi = U.min(axis=(1, 2)), U.max(axis=(1, 2))
        kappa = kappa_from_box(lo, hi, sn=rng.uniform(0.0, 0.6))
        if kappa is None:
            continue
        A = np.einsum("a,b,cab->c", w, w, U)      # cell average: a convex combination of the nodal values
        assert (A >= lo - 1e-12 * np.abs(lo)).all() and (A <= hi + 1e-12 * np.abs(hi)).all()
        bound = kappa * K * (hi - lo).sum()
        for direction in (0, 1):
            # sum_b w_b sum_a (l_a(1) - l_a(0)) U(a, b), the nodes paired as the pass pairs them (l_a(1) - l_a(0) is antisymmetric
            # in a: differences of neighbouring values first, so that the rounding is relative to the spread, not to the state)
            V = U if direction == 0 else U.transpose(0, 2, 1)
            D = np.zeros(4)
            for m in range(N // 2):
                D += g[m] * np.einsum("b,cb->c", w, V[:, m, :] - V[:, N - 1 - m, :])
            assert (np.abs(D) <= K * (hi - lo) * (1 + 1e-12) + 1e-300).all()
            Dc = left_eigenvectors(A, direction) @ D[[RHO, MX, MY, EN]]
            s = np.abs(Dc).sum()
            assert s <= bound * (1 + 1e-10), (trial, direction, s, bound)
            worst = max(worst, s / bound if bound > 0 else 0.0)
        n_bound += 1
    assert n_bound > 2000          # most boxes are admissible: the test did look
    assert worst > 0.02            # and the bound is not vacuous (it is reached within a factor of 50)


def test_q1_nodal_spread_from_row_means_and_row_differences():
    """k = 1: the kernel has no row extremes; it bounds (largest - smallest) of the four nodal values of a component by
    |mean_1 - mean_0| + max |difference along a row| (stage_kernels.hpp, last wave's block)"""
    rng = np.random.default_rng(7)
    for _ in range(20000):
        u = rng.normal(size=(2, 2)) * 10.0 ** rng.uniform(-8, 2)     # u[a, b]: node a of row b
        means = u.mean(axis=0)
        diffs = u[1] - u[0]
        d = abs(means[1] - means[0]) + np.abs(diffs).max()
        assert u.max() - u.min() <= d * (1 + 1e-14)
        assert np.abs(u - u.mean()).max() <= d * (1 + 1e-14)
