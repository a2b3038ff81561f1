"""SURVEY.md 8(c), pins (3) and (4) of the unpinned oracle: the fvsc face gradients converge at second order on smooth
fields (method of manufactured solutions), and the explicit QGDFoam step reproduces Sod's shock tube against the exact
Riemann solution.  CPU: the oracle; GPU: the device path on the same problems."""
import numpy as np
import pytest

import qgdsolver_amd as q
from qgdsolver_amd import fvsc
from qgdsolver_amd import _lib as L
from oracle import OracleCase
from util import oracle_mesh_of

G, E = L.PATCH_GENERIC, L.PATCH_EMPTY


# ---- (3) manufactured smooth field: order of the face-gradient error ------------------------------------------------
def smooth(x):
    return np.sin(2 * np.pi * x[:, 0]) * np.cos(2 * np.pi * x[:, 1]) + 0.5 * np.cos(2 * np.pi * x[:, 2])


def smooth_grad(x):
    return np.stack([2 * np.pi * np.cos(2 * np.pi * x[:, 0]) * np.cos(2 * np.pi * x[:, 1]),
                     -2 * np.pi * np.sin(2 * np.pi * x[:, 0]) * np.sin(2 * np.pi * x[:, 1]),
                     -np.pi * np.sin(2 * np.pi * x[:, 2])], axis=1)


def gradient_error(mesh, scheme, grad_fn):
    C = mesh.array("C").reshape(-1, 3)
    Cf = mesh.array("Cf").reshape(-1, 3)
    nIF = mesh.nInternalFaces
    out = grad_fn(scheme, smooth(C), smooth(Cf[nIF:]))
    fc = Cf[:nIF]
    dims = [d for d in range(3) if np.ptp(C[:, d]) > 0]
    sel = np.ones(nIF, bool)
    for d in dims:  # away from the patches (vertex values there are one-sided averages)
        sel &= (fc[:, d] > 0.2) & (fc[:, d] < 0.8)
    exact = smooth_grad(fc)
    err = np.abs(out[:nIF] - exact)[sel][:, dims]
    return float(np.sqrt((err ** 2).mean()))


def mesh_2d(n):
    return q.PolyMesh.box(n, n, 1, hi=(1.0, 1.0, 1.0 / n), patch_types=[G, G, G, G, E, E])


def observed_orders(make, sizes, scheme, grad_fn_of):
    errs = []
    for n in sizes:
        mesh = make(n)
        errs.append(gradient_error(mesh, scheme, grad_fn_of(mesh)))
    return [np.log2(errs[i] / errs[i + 1]) for i in range(len(errs) - 1)], errs


def oracle_grad(mesh):
    om = oracle_mesh_of(mesh)

    def fn(scheme, cell, bnd):
        rc, out = om.fvsc(scheme, "grad_s", cell, bnd)
        assert rc == 0
        return out
    return fn


@pytest.mark.parametrize("dim,scheme", [(2, "leastSquares"), (2, "GaussVolPoint"), (3, "GaussVolPoint")])
def test_oracle_gradient_is_second_order(dim, scheme):
    make = mesh_2d if dim == 2 else (lambda n: q.PolyMesh.box(n, n, n))
    sizes = (16, 32, 64) if dim == 2 else (8, 16, 32)
    orders, errs = observed_orders(make, sizes, scheme, oracle_grad)
    assert all(o > 1.8 for o in orders), (scheme, orders, errs)


def test_reduced_is_only_consistent_in_the_normal_direction():
    """nf*snGrad has no tangential part: its error against the full gradient does not converge (why QGD needs fvsc)"""
    orders, errs = observed_orders(mesh_2d, (16, 32, 64), "reduced", oracle_grad)
    assert all(abs(o) < 0.3 for o in orders) and errs[-1] > 1.0


# ---- (4) Sod's shock tube ---------------------------------------------------------------------------------------------
GAMMA = 1.4


def sod_exact(x, t, x0=0.5, left=(1.0, 0.0, 1.0), right=(0.125, 0.0, 0.1)):
    """exact Riemann solution (Toro): density, velocity, pressure at positions x and time t"""
    g = GAMMA
    rl, ul, pl = left
    rr, ur, pr = right
    cl, cr = np.sqrt(g * pl / rl), np.sqrt(g * pr / rr)

    def f(p, rk, pk, ck):
        if p > pk:
            A, B = 2 / ((g + 1) * rk), (g - 1) / (g + 1) * pk
            return (p - pk) * np.sqrt(A / (p + B))
        return 2 * ck / (g - 1) * ((p / pk) ** ((g - 1) / (2 * g)) - 1)

    lo, hi = 1e-8, 10.0
    for _ in range(200):  # bisection on f_L + f_R + du = 0
        mid = 0.5 * (lo + hi)
        if f(mid, rl, pl, cl) + f(mid, rr, pr, cr) + (ur - ul) > 0:
            hi = mid
        else:
            lo = mid
    ps = 0.5 * (lo + hi)
    us = 0.5 * (ul + ur) + 0.5 * (f(ps, rr, pr, cr) - f(ps, rl, pl, cl))
    rsl = rl * (ps / pl) ** (1 / g)                                   # left rarefaction
    csl = cl * (ps / pl) ** ((g - 1) / (2 * g))
    rsr = rr * ((ps / pr + (g - 1) / (g + 1)) / ((g - 1) / (g + 1) * ps / pr + 1))   # right shock
    S = ur + cr * np.sqrt((g + 1) / (2 * g) * ps / pr + (g - 1) / (2 * g))
    xi = (x - x0) / t
    rho, u, p = np.empty_like(x), np.empty_like(x), np.empty_like(x)
    for i, s in enumerate(xi):
        if s < ul - cl:
            rho[i], u[i], p[i] = rl, ul, pl
        elif s < us - csl:
            u[i] = 2 / (g + 1) * (cl + (g - 1) / 2 * ul + s)
            c = cl - (g - 1) / 2 * (u[i] - ul)
            rho[i] = rl * (c / cl) ** (2 / (g - 1))
            p[i] = pl * (c / cl) ** (2 * g / (g - 1))
        elif s < us:
            rho[i], u[i], p[i] = rsl, us, ps
        elif s < S:
            rho[i], u[i], p[i] = rsr, us, ps
        else:
            rho[i], u[i], p[i] = rr, ur, pr
    return rho, u, p


def run_sod(n, case_factory, t_end=0.2, co=0.1, consistent=1):
    mesh = q.PolyMesh.box(n, 1, 1, hi=(1.0, 1.0 / n, 1.0 / n), patch_types=[G, G, E, E, E, E])
    x = mesh.array("C").reshape(-1, 3)[:, 0]
    R = 1.0 / GAMMA                      # c = sqrt(gamma R T): c = 1 at T = 1
    rho0 = np.where(x < 0.5, 1.0, 0.125)
    p0 = np.where(x < 0.5, 1.0, 0.1)
    T0 = p0 / (rho0 * R)
    dt = co * (1.0 / n) / 2.2            # |u| + c stays below 2.2
    steps = int(round(t_end / dt))
    dt = t_end / steps
    case = case_factory(mesh, q.default_options(stencil="GaussVolPoint", deltaT=dt, R=R, Cv=R / (GAMMA - 1), mu=0.0, alphaQGD=0.5,
                                                consistentEnergy=consistent))
    for patch in (2, 3, 4, 5):
        case.set_bc(patch, U=("none", None), T=("none", None), p=("none", None))
    case.set_fields(np.zeros((n, 3)), T0, p0)
    case.step(steps)
    return x, case.field("rho"), case.field("U")[:, 0], case.field("p")


def l1(a, b):
    return float(np.abs(a - b).mean())


def check_sod(factory):
    """consistentEnergy = 1: the exact Riemann solution is approached as the mesh is refined"""
    errs = []
    for n in (200, 400):
        x, rho, u, p = run_sod(n, factory)
        re, ue, pe = sod_exact(x, 0.2)
        errs.append((l1(rho, re), l1(u, ue), l1(p, pe)))
        # plateau between contact and shock, away from the smeared fronts
        star = (x > 0.71) & (x < 0.78)
        assert np.abs(p[star] - pe[star]).max() < 0.01 and np.abs(u[star] - ue[star]).max() < 0.02, (n, p[star].mean(), u[star].mean())
        assert rho.min() > 0.12 and rho.max() < 1.001
    assert errs[0][0] < 0.022 and errs[0][1] < 0.042 and errs[0][2] < 0.02, errs
    # a regularised first-order-in-tau scheme: the L1 error shrinks with the mesh (discontinuities: order < 1)
    assert all(errs[1][k] < 0.75 * errs[0][k] for k in range(3)), errs
    return errs


def check_sod_as_listed(factory):
    """consistentEnergy = 0, the explicit energy re-solve exactly as the listing has it [QGDEEqn.H:67-72]: rho*e advances by
    the increment of rhoE, so the pressure is formed from the total instead of the internal energy wherever the gas was set in
    motion.  The tube then settles on p* = 0.331, u* = 0.964 instead of 0.3031, 0.9275 -- the same on every mesh and with
    every time step: a property of the listing, replicated on purpose (needed to agree with the real QGDFoam), not a
    discretisation error."""
    plate = []
    for n, co in ((200, 0.1), (200, 0.03), (400, 0.1)):
        x, rho, u, p = run_sod(n, factory, co=co, consistent=0)
        star = (x > 0.71) & (x < 0.78)
        plate.append((p[star].mean(), u[star].mean()))
    for ps, us in plate:
        assert abs(ps - 0.331) < 0.002 and abs(us - 0.964) < 0.003, plate
    return plate


def test_oracle_sod_shock_tube():
    factory = lambda mesh, opt: OracleCase(oracle_mesh_of(mesh), opt)  # noqa: E731
    check_sod(factory)
    check_sod_as_listed(factory)


@pytest.mark.gpu
def test_device_sod_shock_tube_and_gradient_order():
    devs = []

    def factory(mesh, opt):
        dev = q.Device(mesh)
        devs.append(dev)
        return q.QGDFoamCase(dev, opt)

    check_sod(factory)
    check_sod_as_listed(factory)
    # and the device follows the oracle in both forms of the energy update
    for consistent in (0, 1):
        xo, ro, uo, po = run_sod(200, lambda mesh, opt: OracleCase(oracle_mesh_of(mesh), opt), consistent=consistent)
        xd, rd, ud, pd = run_sod(200, factory, consistent=consistent)
        assert np.abs(rd - ro).max() <= 1e-10 and np.abs(pd - po).max() <= 1e-10 and np.abs(ud - uo).max() <= 1e-10

    def device_grad(mesh):
        dev = q.Device(mesh)
        devs.append(dev)

        def fn(scheme, cell, bnd):
            dev.fvSchemes = {"fvsc": {"default": scheme}}
            return fvsc.grad(dev, q.volField("phi", cell, bnd))
        return fn

    orders, errs = observed_orders(lambda n: q.PolyMesh.box(n, n, n), (16, 32, 64), "GaussVolPoint", device_grad)
    assert all(o > 1.9 for o in orders), (orders, errs)
    orders, errs = observed_orders(mesh_2d, (32, 64, 128), "leastSquares", device_grad)
    assert all(o > 1.9 for o in orders), (orders, errs)
    for d in devs:
        d.close()
