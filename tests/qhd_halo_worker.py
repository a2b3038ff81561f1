"""Worker of tests/test_qhd_sharded.py::test_sharded_oracle_over_gloo: one rank of a gloo run of the sharded QHD ORACLE on a
cell-range shard of a renumbered polygonal mesh, driven by qgdsolver_amd.halo.QhdStepper over DistWorld."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    outdir, steps = sys.argv[1], int(sys.argv[2])
    import torch
    import torch.distributed as dist

    from qgdsolver_amd.halo import DistWorld, QhdStepper
    from qhd_shards import make_oracle_shard_case, range_shards
    from test_qhd_case import cavity_bcs, options
    from test_qhd_sharded import perturbed
    from util import make_mesh

    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    g = make_mesh("box654_poly")
    g.renumber(g.rcm_order())
    sh = range_shards(g, world)[rank]
    opt = options("GaussVolPoint", deltaT=1e-3, precond=0, pRefCell=17, pRefValue=0.0)
    case = make_oracle_shard_case(sh, opt, cavity_bcs, perturbed(g), 17, True)
    to_t = lambda buf, n: torch.from_numpy(buf[:n])           # noqa: E731  (views: the receive lands in the case's buffer)
    from_t = lambda t, buf: None                              # noqa: E731
    QhdStepper(DistWorld(case, dist, torch, sh["peers"], to_t, from_t)).step(steps)
    own = sh["owned"]
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), cells=sh["cell_global"][own], iterations=case.info()["pIterations"],
             **{f: case.field(f)[own] for f in ("U", "T", "p")})
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
