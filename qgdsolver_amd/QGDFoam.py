"""``python -m qgdsolver_amd.QGDFoam -case <dir>``: the explicit branch of the QGDFoam application
(QGDFoam.C L63-163) run from an OpenFOAM case directory on one MI355X.

What the reference's ``main`` does per step -- updateFields.H, updateFluxes.H, QGDCourantNo.H,
setDeltaT-QGDQHD.H, the three explicit equations, thermo.correct(), runTime.write() -- is
``QGDFoamCase.step`` plus ``foamfile.write_time`` here.  Read from system/controlDict:
startFrom/startTime, endTime, deltaT, writeControl (timeStep, or runTime/adjustableRunTime with a
fixed deltaT), writeInterval, adjustTimeStep/maxCo/maxDeltaT, timePrecision.
"""
import argparse
import os
import sys
import time as _time

import numpy as np

from . import foamfile as ff


def time_name(t, precision=6):
    """Time::timeName(): general format with timePrecision significant digits (L0)"""
    s = f"{t:.{precision}g}"
    if "e" in s:
        m, e = s.split("e")
        s = f"{m}e{int(e):+03d}"
    return s


def find_start_time(case_dir, cd):
    start_from = str(cd.get("startFrom", "startTime"))
    times = []
    for d in os.listdir(case_dir):
        try:
            times.append((float(d), d))
        except ValueError:
            pass
    if not times:
        raise ff.FoamFileError(f"{case_dir}: no time directory")
    if start_from == "latestTime":
        return max(times)
    if start_from == "firstTime":
        return min(times)
    want = float(cd.get("startTime", 0))
    for t, d in times:
        if abs(t - want) <= 1e-12 * max(1.0, abs(want)):
            return t, d
    raise ff.FoamFileError(f"{case_dir}: no time directory for startTime {want}")


def run(case_dir, n_steps=None, device_id=0, write=True, log=print):
    cd = ff.read_dict(os.path.join(case_dir, "system", "controlDict"))
    t0, t0_name = find_start_time(case_dir, cd)
    dev, case = ff.load_case(case_dir, t0_name, device_id)
    _, _, _, bcs = ff.read_case_setup(case_dir, t0_name)
    dt = float(cd["deltaT"])
    end_time = float(cd["endTime"])
    adjust = bool(case.options.adjustTimeStep)
    control = str(cd.get("writeControl", "timeStep"))
    interval = float(cd.get("writeInterval", 1))
    precision = int(cd.get("timePrecision", 6))
    if control == "timeStep":
        chunk = max(1, int(round(interval)))
    elif control in ("runTime", "adjustableRunTime") and not adjust:
        chunk = max(1, int(round(interval / dt)))
    else:
        raise ff.FoamFileError(f"writeControl '{control}' with adjustTimeStep is not supported (use writeControl timeStep)")
    total = n_steps if n_steps is not None else (None if adjust else int(round((end_time - t0) / dt)))
    log(f"QGDFoam (qgdsolver_amd, explicit branch): {case.mesh.nCells} cells, fvsc {case.dev.fvSchemes['fvsc']['default']}, "
        f"deltaT {dt:g}, start {t0_name}")
    done = 0
    wall0 = _time.perf_counter()
    written = []
    while True:
        if total is not None and done >= total:
            break
        n = chunk if total is None else min(chunk, total - done)
        case.step(n)
        done += n
        info = case.info()
        t = t0 + info["time"]
        log(f"Time = {time_name(t, precision)}  steps {done}  deltaT {info['deltaT']:.6g}  Courant max {info['CoNum']:.6g}  "
            f"min rho {info['minRho']:.6g}  min e {info['minE']:.6g}  ClockTime {_time.perf_counter() - wall0:.2f} s")
        if not np.isfinite(info["minRho"]) or info["minRho"] <= 0:
            raise FloatingPointError(f"density lost positivity at time {t:g}")
        if write:
            name = time_name(t, precision)
            ff.write_time(case, case_dir, name, bcs)
            written.append(name)
        if total is None and t >= end_time - 1e-12 * max(1.0, abs(end_time)):
            break
    log("End")
    return dev, case, written


def main(argv=None):
    ap = argparse.ArgumentParser(prog="QGDFoam", description=__doc__.split("\n\n")[0])
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
