"""Discharge "parity unpinned" on a machine that has OpenFOAM v2312 + QGDsolver.

    1. run the reference there:   QGDFoam -case <case>      (controlDict: writeFormat ascii; writePrecision 17;
                                                            fixed deltaT; implicitDiffusion false; constScPrModel1)
    2. run this script:           python scripts/compare_with_reference_run.py <case> <startTimeName> <endTimeName>

It loads <case>/<startTimeName> through qgdsolver_amd.foamfile.load_case, advances the device path to <endTimeName>
with the case's deltaT, and prints the largest relative difference of rho, U, p, T against the fields the reference
wrote into <case>/<endTimeName>.  Exit code 0 when every field is within --tol (default 1e-10, the north-star bar).
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("case")
    ap.add_argument("start")
    ap.add_argument("end")
    ap.add_argument("--tol", type=float, default=1e-10)
    a = ap.parse_args(argv)
    from qgdsolver_amd import foamfile as ff

    dev, case = ff.load_case(a.case, a.start)
    dt = case.options.deltaT
    n = int(round((float(a.end) - float(a.start)) / dt))
    if n <= 0 or abs(float(a.start) + n * dt - float(a.end)) > 1e-9 * max(1.0, abs(float(a.end))):
        print(f"end time {a.end} is not a whole number of deltaT = {dt:g} steps after {a.start}", file=sys.stderr)
        return 2
    case.step(n)
    worst = 0.0
    for name in ("rho", "U", "p", "T"):
        path = os.path.join(a.case, a.end, name)
        if not os.path.exists(path):
            print(f"{name:4s} (not written by the reference run)")
            continue
        ref, _ = ff.read_field(path, case.mesh)
        got = case.field(name).reshape(ref.shape)
        err = float(np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-300))
        worst = max(worst, err)
        print(f"{name:4s} max rel. difference after {n} steps: {err:.3e}")
    ok = worst <= a.tol
    print("PARITY", "OK" if ok else "FAILED", f"(tolerance {a.tol:g})")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
