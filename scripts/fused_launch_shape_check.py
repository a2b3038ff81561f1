"""A launch-shape switch of the plain explicit step (default QGD_FU_PAIR = a workgroup takes two consecutive blocks, the second one's lists fetched
during the first one's cell update; QGD_FU_PERSIST with scripts/probes/fused_persistent_probe_r06.patch applied) against one workgroup per block: the same states bit for bit after a few steps on an n^3 box and on a box whose edges
the bricks do not divide.    python scripts/fused_launch_shape_check.py [n=128] [steps=5] [VARIABLE=QGD_FU_PAIR] [value=1]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import qgdsolver_amd as q  # noqa: E402
from qgdsolver_amd.synthetic import box_initial_fields  # noqa: E402


VAR = sys.argv[3] if len(sys.argv) > 3 else "QGD_FU_PAIR"
ON = int(sys.argv[4]) if len(sys.argv) > 4 else 1


def run(dims, persist, steps):
    os.environ[VAR] = str(persist)
    mesh = q.PolyMesh.box(*dims)
    dev = q.Device(mesh)
    case = q.QGDFoamCase(dev, q.default_options(stencil="GaussVolPoint", deltaT=0.1 / max(dims) / 1.3))
    U, T, p = box_initial_fields(mesh.array("C").reshape(-1, 3))
    T = T + 1e-3 * np.random.default_rng(1).standard_normal(T.shape)
    case.set_fields(U, T, p)
    fi = case.fused_info()
    for _ in range(steps):
        case.step_phase(3)
    case.sync()
    out = {k: np.array(case.field(k)) for k in ("rho", "U", "p", "e")}
    info = case.info()
    case.close(); dev.close(); mesh.close()
    return out, fi, info


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    for dims in ((n, n, n), (n + 3, n - 5, n + 1)):
        a, fa, ia = run(dims, 0, steps)
        b, fb, ib = run(dims, ON, steps)
        same = all(np.array_equal(a[k], b[k]) for k in a)
        worst = max(float(np.abs(a[k] - b[k]).max()) for k in a)
        print(dims, "blocks", fa["blocks"], "fused", fa["fused"], fb["fused"], "bit-identical", same, "max diff", worst, "minRho", ia["minRho"], ib["minRho"], flush=True)
        assert fa["fused"] and fb["fused"] and same


if __name__ == "__main__":
    main()
