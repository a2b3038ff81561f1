"""Worker of tests/test_qhd_sharded.py::test_device_shards_over_gloo: one rank of a gloo run of the sharded QHD case ON THE DEVICE
(every rank on GPU 0 of the box, messages and reductions staged through host tensors), driven by QhdStepper over DistWorld -- the
protocol a real multi-GPU run follows, incl. the comm points of the multigrid hierarchy that spans the ranks (qgd_qhd_case_pending)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def spanning_test_mesh():
    import qgdsolver_amd as q
    g = q.PolyMesh.box(14, 12, 16).jitter(0.2, seed=3)
    g.split_quads(5)
    g.renumber(np.random.default_rng(8).permutation(g.nCells).astype(np.int32))
    g.renumber(g.morton_order())
    return g


def main():
    outdir, steps = sys.argv[1], int(sys.argv[2])
    import torch
    import torch.distributed as dist

    import qgdsolver_amd as q
    from qgdsolver_amd import _lib as L, qhdfoam
    from qgdsolver_amd.halo import DistWorld, QhdStepper
    from qhd_shards import range_shards
    from test_qhd_case import cavity_bcs, options
    from test_qhd_sharded import perturbed

    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    g = spanning_test_mesh()
    sh = range_shards(g, world)[rank]
    opt = options("GaussVolPoint", deltaT=1e-3, pRefCell=g.nCells // 2, pRefValue=0.1, pTol=1e-11)
    fields = perturbed(g)
    dev = q.Device(sh["mesh"])
    case = qhdfoam.QHDFoamCase(dev, opt)
    cavity_bcs(case, sh["mesh"])
    cg = sh["cell_global"]
    case.set_fields(fields[0][cg], fields[1][cg], fields[2][cg])

    def to_t(ptr, n):                     # device pointer -> host tensor (a copy)
        return torch.from_numpy(dev.to_host(ptr, (int(n),)))

    def from_t(t, ptr):                   # ... and back
        a = np.ascontiguousarray(t.numpy())
        if a.nbytes:
            L.check(L.lib.qgd_device_copy(dev._h, C.c_void_p(ptr), a.ctypes.data_as(C.c_void_p), a.nbytes, 1), "qgd_device_copy")

    QhdStepper(DistWorld(case, dist, torch, sh["peers"], to_t, from_t)).step(steps)
    own = sh["owned"]
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), cells=sh["cell_global"][own], iterations=case.info()["pIterations"],
             levels=case.info()["mgLevels"], **{f: case.field(f)[own] for f in ("U", "T", "p")})
    dist.barrier()
    case.close(); dev.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
