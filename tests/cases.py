"""Shared synthetic cases for the parity tests, the smoke test and the bench."""
import numpy as np


def box_initial_fields(C, seed=12345, noise=1e-3):
    """C3/C4 initial state on cell centres C (n,3): p = 1 + 0.1 exp(-|x-xc|^2/0.01), T = 1 (+ seeded noise so that
    rho carries uniform(-noise, noise) perturbations), U = 0.1 (sin2pi x cos2pi y, -cos2pi x sin2pi y, 0)."""
    C = np.asarray(C).reshape(-1, 3)
    x, y = C[:, 0], C[:, 1]
    xc = np.array([0.5, 0.5, 0.5])
    r2 = ((C - xc) ** 2).sum(axis=1)
    p = 1.0 + 0.1 * np.exp(-r2 / 0.01)
    rng = np.random.Generator(np.random.MT19937(seed))
    T = 1.0 + rng.uniform(-noise, noise, size=C.shape[0])
    U = np.zeros_like(C)
    U[:, 0] = 0.1 * np.sin(2 * np.pi * x) * np.cos(2 * np.pi * y)
    U[:, 1] = -0.1 * np.cos(2 * np.pi * x) * np.sin(2 * np.pi * y)
    return U, T, p


def random_fields(n, nb, ncomp, seed):
    rng = np.random.default_rng(seed)
    shape_c = (n, ncomp) if ncomp > 1 else (n,)
    shape_b = (nb, ncomp) if ncomp > 1 else (nb,)
    return rng.standard_normal(shape_c), rng.standard_normal(shape_b)


def forward_step_bcs(case, mesh_patch_names=("inlet", "outlet", "bottom", "top", "step", "frontAndBack")):
    """C2 boundary conditions: inlet fixedValue (U=(3,0,0), T=1, p=1), outlet zeroGradient, walls slip U +
    qgdFlux p + zeroGradient T, frontAndBack empty."""
    case.set_bc(0, U=("fixedValue", (3.0, 0.0, 0.0)), T=("fixedValue", 1.0), p=("fixedValue", 1.0))
    case.set_bc(1)
    for wall in (2, 3, 4):
        case.set_bc(wall, U=("slip", None), T=("zeroGradient", None), p=("qgdFlux", None))
    case.set_bc(5, U=("none", None), T=("none", None), p=("none", None))
