"""Worker of tests/test_halo_gloo.py: one rank of a world_size-N gloo run of the SHARDED ORACLE.
Writes its owned cells' state to <outdir>/rank<r>.npz."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    outdir, n, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    adjust = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    import torch
    import torch.distributed as dist

    import qgdsolver_amd as q
    from qgdsolver_amd.halo import SlabHalo, slab_range
    import cases
    from oracle import OracleCase, OracleMesh

    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    nx, ny = n, n - 1
    lo, hi, k_lo, k_hi = slab_range(n, rank, world)
    mesh = q.PolyMesh.box(nx, ny, n, k_range=(k_lo, k_hi))
    om = OracleMesh(mesh.primitives())
    for side in (0, 1):
        om.set_halo(side, mesh.array(f"haloGhost{side}"), mesh.array(f"haloSend{side}"))
    opt = q.default_options(stencil="GaussVolPoint", deltaT=2e-3, mu=1e-3, adjustTimeStep=adjust, maxCo=0.3, maxDeltaT=1.0, cTau=0.75)
    case = OracleCase(om, opt)
    # initial fields are functions of the GLOBAL mesh: build them there and cut the window
    gmesh = q.PolyMesh.box(nx, ny, n)
    U, T, p = cases.box_initial_fields(gmesh.array("C").reshape(-1, 3))
    plane = nx * ny
    sl = slice(plane * k_lo, plane * k_hi)
    case.set_fields(U[sl], T[sl], p[sl])
    halo = SlabHalo(case, rank, world, dist, alloc=lambda c: torch.zeros(c, dtype=torch.float64),
                    arg=lambda t: t.numpy())
    halo.exchange()
    case.step_phase(2)

    def allreduce_max(c):
        t = torch.from_numpy(c.reduction())
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        c.reduction(t.numpy())

    for _ in range(steps):
        halo.step(allreduce_max if adjust else None)
    own = slice(plane * (lo - k_lo), plane * (hi - k_lo))
    info = case.info()
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), lo=lo, hi=hi, time=info["time"], deltaT=info["deltaT"],
             **{f: case.field(f)[own] for f in ("rho", "U", "p", "e")})
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
