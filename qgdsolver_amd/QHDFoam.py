"""``python -m qgdsolver_amd.QHDFoam -case <dir>``: the QHDFoam application (QHDFoam.C L63-139) run from an OpenFOAM case
directory on one MI355X.

What the reference's ``main`` does per step -- updateFields.H, updateFluxes.H, QHDpEqn.H, QHDUEqn.H, QHDTEqn.H (either branch of
implicitDiffusion; the reference's default is true), the reference level of p, runTime.write() -- is ``QHDFoamCase.step`` plus
``foamfile.write_qhd_time`` here.  Read from system/controlDict: startFrom/startTime, endTime, deltaT, writeControl (timeStep, or
runTime/adjustableRunTime), writeInterval, timePrecision; adjustTimeStep is refused (the path's matrices are built once for the
fixed deltaT).  One rank; a decomposed run goes through ``bench.py --workload qhd --gpus N`` / ``qgd_qhd_case_step_sharded``.
"""
import argparse
import os
import sys
import time as _time

import numpy as np

from . import foamfile as ff
from .QGDFoam import find_start_time, time_name


def run(case_dir, n_steps=None, device_id=0, write=True, log=print):
    cd = ff.read_dict(os.path.join(case_dir, "system", "controlDict"))
    t0, t0_name = find_start_time(case_dir, cd)
    from .fvsc import Device
    from .qhdfoam import QHDFoamCase, qhd_options

    mesh, opt, fields, bcs = ff.read_qhd_case_setup(case_dir, t0_name)
    # QGD_QHD_FUSED=1: the cell blocks of QGDFoam's one-launch step carry QHDFoam's explicit U and T equations on 3-D GaussVolPoint cases (off
    # by default: no gain measured); otherwise QHDFoam has no use for their tables
    blocks = (os.environ.get("QGD_QHD_FUSED", "0") == "1" and opt["stencil"] == "GaussVolPoint" and not opt.get("implicitDiffusion")
              and mesh.nGeometricD == 3)
    dev = Device(mesh, device_id, fv_schemes={"fvsc": {"default": opt["stencil"]}}, fused_tables=True if blocks else False)
    case = QHDFoamCase(dev, qhd_options(**opt))
    for i, bc in enumerate(bcs):
        case.set_bc(i, U=bc["U"], T=bc["T"], p=bc["p"])
    case.set_fields(fields["U"], fields["T"], fields["p"])
    dt = float(cd["deltaT"])
    end_time = float(cd["endTime"])
    control = str(cd.get("writeControl", "timeStep"))
    interval = float(cd.get("writeInterval", 1))
    precision = int(cd.get("timePrecision", 6))
    if control == "timeStep":
        chunk = max(1, int(round(interval)))
    elif control in ("runTime", "adjustableRunTime"):
        chunk = max(1, int(round(interval / dt)))
    else:
        raise ff.FoamFileError(f"writeControl '{control}' is not supported (timeStep, runTime, adjustableRunTime)")
    total = n_steps if n_steps is not None else int(round((end_time - t0) / dt))
    branch = "implicitDiffusion true" if opt["implicitDiffusion"] else "implicitDiffusion false"
    log(f"QHDFoam (qgdsolver_amd, {branch}): {mesh.nCells} cells, fvsc {opt['stencil']}, QGDCoeffs {opt['tauModel']}, deltaT {dt:g}, "
        f"start {t0_name}")
    px, xx = os.environ.get("QGD_QHD_PEXTRAP", "4"), os.environ.get("QGD_IMPL_XEXTRAP", "3")
    if px != "0" or (opt.get("implicitDiffusion") and xx != "0"):
        log(f"  NOTE: the pressure solve starts from p extrapolated in time over the last steps (QGD_QHD_PEXTRAP={px}, limited per value)"
            + (f", the U / T solves from the extrapolated fields (QGD_IMPL_XEXTRAP={xx})" if opt.get("implicitDiffusion") else "")
            + ", not from the old field as OpenFOAM does [QHDpEqn.H L45]: same systems, same tolerances, same answer to those tolerances -- but "
            "'Initial residual' and 'No Iterations' below are NOT comparable with a reference QHDFoam log.  QGD_QHD_PEXTRAP=0 "
            "QGD_IMPL_XEXTRAP=0 restore OpenFOAM's start values.")
    done = 0
    wall0 = _time.perf_counter()
    written = []
    while done < total:
        n = min(chunk, total - done)
        case.step(n)
        done += n
        info = case.info()
        t = t0 + info["time"]
        T = case.field("T")
        log(f"Time = {time_name(t, precision)}  steps {done}")
        log(f"  p: Initial residual = {info['pInitialResidual']:.3g}, Final residual = {info['pFinalResidual']:.3g}, No Iterations {info['pIterations']}")
        if opt["implicitDiffusion"]:
            ii = case.implicit_info()
            log("  " + "  ".join(f"{k}: {v['initial']:.3g} -> {v['final']:.3g} in {v['iterations']}" for k, v in ii["solves"].items()
                                  if v["iterations"] or v["initial"]))
            if ii["unconverged_steps"]:
                log(f"  WARNING: {ii['unconverged_steps']} step(s) so far in which an implicit solve stopped above its tolerance "
                    f"(implicitTol {case.options.implicitTol:g}, maxIter {case.options.implicitMaxIter})")
            if ii["stalled_steps"]:
                log(f"  NOTE: {ii['stalled_steps']} step(s) so far in which a Chebyshev solve ended at the rounding floor of its residual, above "
                    f"implicitTol {case.options.implicitTol:g} (OpenFOAM would have iterated on to maxIter)")
        log(f"  max/min of T: {T.max():.9g}/{T.min():.9g}  ClockTime {_time.perf_counter() - wall0:.2f} s")   # QHDTEqn.H L94
        if not np.isfinite(T).all():
            raise FloatingPointError(f"T is not finite at time {t:g}")
        if write:
            name = time_name(t, precision)
            ff.write_qhd_time(case, case_dir, name, bcs)
            written.append(name)
    log("End")
    return dev, case, written


def main(argv=None):
    ap = argparse.ArgumentParser(prog="QHDFoam", description=__doc__.split("\n\n")[0])
    ap.add_argument("-case", dest="case", default=".")
    ap.add_argument("-nSteps", dest="n_steps", type=int, default=None, help="run this many steps instead of up to endTime")
    ap.add_argument("-device", dest="device", type=int, default=0)
    ap.add_argument("-noWrite", dest="no_write", action="store_true")
    a = ap.parse_args(argv)
    dev, case, _ = run(a.case, a.n_steps, a.device, not a.no_write)
    case.close()
    dev.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
