"""Worker of tests/test_implicit_sharded.py::test_sharded_oracle_over_gloo: one rank of a gloo run of the sharded ORACLE with
implicitDiffusion true on a cell-range shard, driven by qgdsolver_amd.halo.ImplicitStepper over DistWorld."""
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

    from qgdsolver_amd.halo import DistWorld, ImplicitShard, ImplicitStepper
    import cases
    from test_implicit_sharded import cut_mesh, oracle_shard_case, mixed_bcs

    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    g, shards = cut_mesh("ranges", world)
    sh = shards[rank]
    case = oracle_shard_case(sh, mixed_bcs, cases.box_initial_fields(g.array("C").reshape(-1, 3)))
    to_t = lambda buf, n: torch.from_numpy(buf[:n])           # noqa: E731
    from_t = lambda t, buf: None                              # noqa: E731
    ImplicitStepper(DistWorld(ImplicitShard(case), dist, torch, sh["peers"], to_t, from_t, kinds=range(5))).step(steps)
    own = sh["owned"]
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), cells=sh["cell_global"][own], **{f: case.field(f)[own] for f in ("rho", "U", "p", "e")})
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
