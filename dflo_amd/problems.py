"""Initial / boundary data of the example problems, evaluated on the host (the reference does this
with FunctionParser or hard-coded Function classes; it is outside the device path)."""
import numpy as np

GAMMA = 1.4  # src/equation.cc:33


def isentropic_vortex(x, y, beta=5.0, x0=0.0, y0=0.0):
    """IsentropicVortex::vector_value, src/ic.cc:44-61 with (beta, x0, y0) = (5,0,0) (src/ic.cc:110)."""
    a1 = 0.5 * beta / np.pi
    a2 = (GAMMA - 1.0) * a1 ** 2 / 2.0
    r2 = (x - x0) ** 2 + (y - y0) ** 2
    rho = (1.0 - a2 * np.exp(1.0 - r2)) ** (1.0 / (GAMMA - 1.0))
    vex = -a1 * (y - y0) * np.exp(0.5 * (1.0 - r2))
    vey = +a1 * (x - x0) * np.exp(0.5 * (1.0 - r2))
    pre = rho ** GAMMA
    return rho * vex, rho * vey, rho, pre / (GAMMA - 1.0) + 0.5 * rho * (vex * vex + vey * vey)


def isentropic_vortex_exact(x, y, t=0.0, beta=5.0, x0=0.0, y0=0.0, u0=0.0, v0=0.0):
    """The MPI variant's vortex (src_mpi/ic.cc:44-60): p = rho^gamma / gamma, optionally advected with
    (u0, v0).  This one IS an exact Euler solution (radial balance dp/dr = rho v_theta^2 / r holds);
    the serial tree's p = rho^gamma (isentropic_vortex above) is off by the factor gamma and slowly
    evolves -- it is kept as the benchmark initial state because BASELINE names src/, not as a known
    answer."""
    a1 = 0.5 * beta / np.pi
    a2 = (GAMMA - 1.0) * a1 ** 2 / 2.0
    xr, yr = x - x0 - u0 * t, y - y0 - v0 * t
    r2 = xr ** 2 + yr ** 2
    rho = (1.0 - a2 * np.exp(1.0 - r2)) ** (1.0 / (GAMMA - 1.0))
    vex = u0 - a1 * yr * np.exp(0.5 * (1.0 - r2))
    vey = v0 + a1 * xr * np.exp(0.5 * (1.0 - r2))
    pre = rho ** GAMMA / GAMMA
    return rho * vex, rho * vey, rho, pre / (GAMMA - 1.0) + 0.5 * rho * (vex * vex + vey * vey)


def sod(x, y):
    """examples/sod_shock_tube/input.prm:39-44"""
    left = x <= 0.5
    z = np.zeros_like(x)
    return z, z, np.where(left, 1.0, 0.125), np.where(left, 2.5, 0.25)


def double_mach(x, y, t=0.0, top=False):
    """examples/double_mach_reflection/input.prm:35-62 (IC: shock x < 1/6 + y/sqrt(3); top BC uses (1+20t))."""
    s = (1.0 / 6.0 + (1.0 + 20.0 * t) / np.sqrt(3.0)) if top else (1.0 / 6.0 + y / np.sqrt(3.0))
    post = x < s
    return (np.where(post, 57.1576766498, 0.0), np.where(post, -33.0, 0.0), np.where(post, 8.0, 1.4),
            np.where(post, 563.5, 2.5))


def forward_step_inflow(x, y):
    """examples/forward_step/input.prm:19-47"""
    o = np.ones_like(x)
    return 4.2 * o, 0.0 * o, 1.4 * o, 8.8 * o


def smooth_perturbation(x, y, L=10.0):
    """Seedless smooth state for residual-parity tests: uniform flow + sines."""
    rho = 1.0 + 0.2 * np.sin(2 * np.pi * x / L + 0.3) * np.cos(2 * np.pi * y / L)
    u = 0.5 + 0.3 * np.cos(2 * np.pi * x / L) * np.sin(2 * np.pi * y / L + 0.1)
    v = -0.2 + 0.25 * np.sin(2 * np.pi * (x + y) / L)
    p = 1.0 + 0.3 * np.cos(2 * np.pi * x / L + 0.7) * np.cos(2 * np.pi * y / L - 0.2)
    return rho * u, rho * v, rho, p / (GAMMA - 1.0) + 0.5 * rho * (u * u + v * v)


def rayleigh_taylor(x, y, gravity=1.0, Lx=0.5, Ly=1.5, A=0.01, P0=2.5):
    """RayleighTaylor::vector_value, src/ic.cc:12-37 (constants src/ic.h:23-26)."""
    rho = np.where(y < 0.0, 1.0, 2.0)
    vel = A * (1.0 + np.cos(2.0 * np.pi * x / Lx)) / 2.0 * (1.0 + np.cos(2.0 * np.pi * y / Ly)) / 2.0
    pre = P0 - gravity * rho * y
    return 0.0 * x, rho * vel, rho, pre / (GAMMA - 1.0) + 0.5 * rho * vel * vel


def vortex_system(x, y, beta=5.0, Rc=4.0):
    """VortexSystem::vector_value, src/ic.cc:68-94 (three vortices on a circle of radius Rc, src/ic.h:63-77)."""
    a1 = 0.5 * beta / np.pi
    a2 = (GAMMA - 1.0) * a1 ** 2 / 2.0
    xs = [0.0, Rc * np.cos(np.pi / 6.0), -Rc * np.cos(np.pi / 6.0)]
    ys = [-Rc, Rc * np.sin(np.pi / 6.0), Rc * np.sin(np.pi / 6.0)]
    rho, vex, vey = 0.0, 0.0, 0.0
    for xi, yi in zip(xs, ys):
        r2 = (x - xi) ** 2 + (y - yi) ** 2
        rho = rho + (1.0 - a2 * np.exp(1.0 - r2)) ** (1.0 / (GAMMA - 1.0))
        vex = vex - a1 * (y - yi) * np.exp(0.5 * (1.0 - r2))
        vey = vey + a1 * (x - xi) * np.exp(0.5 * (1.0 - r2))
    rho = rho - 2.0
    vex, vey = vex / 3.0, vey / 3.0
    pre = np.where((np.abs(x) < 0.1) & (np.abs(y) < 0.1), 50.0, rho ** GAMMA)
    return rho * vex, rho * vey, rho, pre / (GAMMA - 1.0) + 0.5 * rho * (vex * vex + vey * vey)
