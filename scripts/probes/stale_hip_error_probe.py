"""Which library call leaves an error behind in hipGetLastError()?  Replays the sequence of tests/test_halo_gpu.py and asks the HIP
runtime after every call (asking clears it).  usage: python scripts/probes/stale_hip_error_probe.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import qgdsolver_amd as q
from qgdsolver_amd.halo import slab_range
import cases
hip = ctypes.CDLL("libamdhip64.so")
hip.hipGetErrorString.restype = ctypes.c_char_p


def ask(where):
    e = hip.hipGetLastError()
    if e:
        print(f"stale error {e} ({hip.hipGetErrorString(e).decode()}) after: {where}", flush=True)


nx, ny, n = 10, 9, 12
opt = q.default_options(stencil="GaussVolPoint", deltaT=2e-3, mu=1e-3)
gmesh = q.PolyMesh.box(nx, ny, n); ask("PolyMesh.box")
U, T, p = cases.box_initial_fields(gmesh.array("C").reshape(-1, 3))
plane = nx * ny
shards = []
for rank in range(2):
    lo, hi, k_lo, k_hi = slab_range(n, rank, 2)
    mesh = q.PolyMesh.box(nx, ny, n, k_range=(k_lo, k_hi)); ask("shard mesh")
    dev = q.Device(mesh); ask("Device")
    case = q.QGDFoamCase(dev, opt); ask("QGDFoamCase")
    sl = slice(plane * k_lo, plane * k_hi)
    case.set_fields(U[sl], T[sl], p[sl]); ask("set_fields")
    shards.append((mesh, dev, case))
c0, c1 = shards[0][2], shards[1][2]
try:
    c0.step(1)
except q.QgdError:
    pass
ask("refused step")
b01 = shards[0][1].alloc(8 * c0.halo_count(1)); ask("alloc")
b10 = shards[1][1].alloc(8 * c1.halo_count(0)); ask("alloc")
c0.halo_pack(1, b01); ask("halo_pack"); c1.halo_pack(0, b10)
c0.sync(); ask("sync"); c1.sync()
c1.halo_unpack(0, b01); ask("halo_unpack"); c0.halo_unpack(1, b10)
for ph in (0, 1, 10, 11):
    c0.step_phase(ph); ask(f"step_phase {ph}")
c0.sync(); c0.field("rho"); ask("field")
c0.info(); ask("info")
shards[0][1].release(b01); ask("release")
shards[1][1].release(b10)
for mesh, dev, case in shards:
    case.close(); ask("case.close")
    dev.close(); ask("dev.close")
print("done", flush=True)
