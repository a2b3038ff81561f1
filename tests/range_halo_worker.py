"""Worker of tests/test_halo_gloo.py::test_range_sharded_*: one rank of a gloo run of the SHARDED ORACLE on a
cell-range shard of an arbitrary (renumbered, polygonal-face) mesh, exchanged with qgdsolver_amd.halo.RangeHalo."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


class OracleShardCase:
    """the oracle case + the mesh attribute RangeHalo reads the peers from"""

    def __init__(self, case, mesh):
        self._c, self.mesh = case, mesh

    def __getattr__(self, name):
        return getattr(self._c, name)


def main():
    outdir, kind, stencil, steps = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
    import torch
    import torch.distributed as dist

    import qgdsolver_amd as q
    from qgdsolver_amd.halo import RangeHalo
    from oracle import OracleCase, OracleMesh
    from test_partition import case_setup

    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    g, bc_fn, (U, T, p), opt = case_setup(kind)
    mesh = g.shard(world, rank)
    om = OracleMesh(mesh.primitives())
    for k in range(mesh.halo_slots):
        om.set_halo(k, mesh.array(f"haloGhost{k}"), mesh.array(f"haloSend{k}"))
    case = OracleCase(om, q.default_options(stencil=stencil, **opt))
    if bc_fn:
        bc_fn(case)
    cg = mesh.array("cellGlobal")
    case.set_fields(U[cg], T[cg], p[cg])
    halo = RangeHalo(OracleShardCase(case, mesh), rank, world, dist, alloc=lambda c: torch.zeros(c, dtype=torch.float64),
                     arg=lambda t: t.numpy())
    halo.exchange()
    case.step_phase(2)
    for _ in range(steps):
        halo.step()
    lo, hi = (g.nCells * rank) // world, (g.nCells * (rank + 1)) // world
    own = (cg >= lo) & (cg < hi)
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), cells=cg[own], peers=mesh.array("haloPeer"),
             **{f: case.field(f)[own] for f in ("rho", "U", "p", "e")})
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
