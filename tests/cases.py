"""Shared synthetic cases for the parity tests, the smoke test and the bench."""
import numpy as np


from qgdsolver_amd.synthetic import box_initial_fields  # noqa: F401,E402  (the benchmark's synthetic input lives in the package)


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
