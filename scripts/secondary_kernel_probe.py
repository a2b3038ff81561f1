"""The two kernels the secondary bench lines put in their `roofline` object, launched through the library's own measurement entries so
that a rocprofv3 --pmc pass over this script sees them with their real operands:
    secondary_kernel_probe.py qhd [n]        30 level-0 sweeps of the pressure preconditioner (mgSmoothKernel<float>, qgd_qhd_case_sweep_time)
    secondary_kernel_probe.py qhd_c5 [n]     the same sweeps on BASELINE config 5's stand-in mesh (tests/test_config5_gpu.c5_mesh, n = 252: 16 M cells)
    secondary_kernel_probe.py implicit [n]   30 matrix products of the U system (iApplyKernel<3,1>, qgd_case_implicit_apply_time)
(scripts/collect_secondary_pmc.sh folds the counters into profiles/<tag>_pmc_secondary.json)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import qgdsolver_amd as q
which = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else (252 if which == "qhd_c5" else 200)
if which == "qhd_c5":
    from test_config5_gpu import c5_mesh
    mesh = c5_mesh(n, 64 ** 3)
else:
    mesh = q.PolyMesh.box(n, n, n)
dev = q.Device(mesh)
if which in ("qhd", "qhd_c5"):
    from qgdsolver_amd import qhdfoam
    from test_qhd_case import cavity_bcs, options, initial
    c = qhdfoam.QHDFoamCase(dev, options(deltaT=0.2 / n, pTol=1e-8, pMaxIter=400))
    cavity_bcs(c, mesh)
    c.set_fields(*initial(mesh))
    c.step(2)
    print("PROBE", which, n, c.sweep_time(30), flush=True)
else:
    import cases
    c = q.QGDFoamCase(dev, q.default_options(stencil="GaussVolPoint", deltaT=0.1 / n / 1.3, implicitDiffusion=1, mu=1e-3))
    c.set_fields(*cases.box_initial_fields(mesh.array("C").reshape(-1, 3)))
    c.step(2)
    print("PROBE", which, n, c.implicit_apply_time(30), flush=True)
c.close(); dev.close()
