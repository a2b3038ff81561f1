"""Worker of tests/test_halo_gpu.py::test_overlapped_choreography: two k-slab shards on GPU 0 in one process,
driven with the stream choreography of SlabHalo.step_overlapped (compute stream + halo stream, torch events),
the RCCL transfer replaced by the shared device buffers.  torch is imported first so both libraries share one HIP
runtime.  Prints the max relative deviation from the unsharded run."""
import os
import sys

import torch  # noqa: F401  (before the HIP library)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

import qgdsolver_amd as q  # noqa: E402
from qgdsolver_amd.halo import slab_range  # noqa: E402
import cases  # noqa: E402


def main():
    nx, ny, n, steps = 24, 20, 30, 12
    opt = q.default_options(stencil="GaussVolPoint", deltaT=1e-3, mu=1e-3)
    gmesh = q.PolyMesh.box(nx, ny, n)
    U, T, p = cases.box_initial_fields(gmesh.array("C").reshape(-1, 3))
    gdev = q.Device(gmesh)
    gcase = q.QGDFoamCase(gdev, opt)
    gcase.set_fields(U, T, p)
    gcase.step(steps)
    plane = nx * ny
    S0 = torch.cuda.current_stream()
    S1 = torch.cuda.Stream()
    shards = []
    for rank in range(2):
        lo, hi, k_lo, k_hi = slab_range(n, rank, 2)
        mesh = q.PolyMesh.box(nx, ny, n, k_range=(k_lo, k_hi))
        dev = q.Device(mesh)
        case = q.QGDFoamCase(dev, opt)
        sl = slice(plane * k_lo, plane * k_hi)
        case.set_fields(U[sl], T[sl], p[sl])
        case.set_stream(S0.cuda_stream)
        case.set_halo_stream(S1.cuda_stream)
        shards.append((lo, hi, k_lo, k_hi, case, dev))
    c0, c1 = shards[0][4], shards[1][4]
    b01 = torch.zeros(c0.halo_count(1), dtype=torch.float64, device="cuda")
    b10 = torch.zeros(c1.halo_count(0), dtype=torch.float64, device="cuda")

    def exchange():
        c0.halo_pack(1, b01.data_ptr()); c1.halo_pack(0, b10.data_ptr())
        c1.halo_unpack(0, b01.data_ptr()); c0.halo_unpack(1, b10.data_ptr())

    S1.wait_stream(S0)
    with torch.cuda.stream(S1):
        exchange()
    S0.wait_stream(S1)
    for _ in range(steps):
        c0.step_phase(0); c1.step_phase(0)
        c0.step_phase(10); c1.step_phase(10)
        S1.wait_stream(S0)
        with torch.cuda.stream(S1):
            exchange()
        c0.step_phase(11); c1.step_phase(11)
        S0.wait_stream(S1)
    torch.cuda.synchronize()
    worst = 0.0
    for lo, hi, k_lo, k_hi, case, dev in shards:
        own = slice(plane * (lo - k_lo), plane * (hi - k_lo))
        for f in ("rho", "U", "p", "e"):
            a = case.field(f)[own]
            b = gcase.field(f)[plane * lo: plane * hi]
            worst = max(worst, float(np.abs(a - b).max() / np.abs(b).max()))
    print("OVERLAP_WORST", worst)


if __name__ == "__main__":
    main()
