"""Helpers shared by the sharded-QHD tests (CPU oracle in process / over gloo, HIP shards on one GPU): building the shards of a
mesh, the per-shard cases with the global reference cell, and gathering owned cells back into the unsharded order."""
import numpy as np

import qgdsolver_amd as q
from qgdsolver_amd import qhdfoam
from qgdsolver_amd.halo import LocalWorld, QhdStepper

from oracle import OracleMesh, OracleQhdCase
from test_qhd_case import cavity_bcs, initial, options


def oracle_shard_mesh(shard):
    om = OracleMesh(shard.primitives())
    for k in range(shard.halo_slots):
        om.set_halo(k, shard.array(f"haloGhost{k}"), shard.array(f"haloSend{k}"))
    om.set_halo_face_h(shard.array("haloFaceH"))
    return om


def box_slabs(nx, ny, nz, world):
    """k-slab shards of a box (qgd_mesh_box with a k range): (mesh, owned local cells, their global labels, peers)"""
    from qgdsolver_amd.halo import slab_range
    out = []
    plane = nx * ny
    for r in range(world):
        lo, hi, k_lo, k_hi = slab_range(nz, r, world)
        m = q.PolyMesh.box(nx, ny, nz, k_range=(k_lo, k_hi))
        local = np.arange(plane * (lo - k_lo), plane * (hi - k_lo))
        glob_all = np.arange(plane * k_lo, plane * k_hi)
        peers = [r - 1 if r > 0 else -1, r + 1 if r < world - 1 else -1]
        out.append(dict(mesh=m, owned=local, cell_global=glob_all, peers=peers))
    return out


def range_shards(gmesh, world):
    out = []
    for r in range(world):
        m = gmesh.shard(world, r)
        cg = m.array("cellGlobal")
        lo, hi = (gmesh.nCells * r) // world, (gmesh.nCells * (r + 1)) // world
        owned = np.nonzero((cg >= lo) & (cg < hi))[0]
        out.append(dict(mesh=m, owned=owned, cell_global=cg, peers=[int(p) for p in m.array("haloPeer")]))
    return out


def needs_reference(bc_setup_kinds):
    return not any(k == "fixedValue" for k in bc_setup_kinds)


def make_oracle_shard_case(sh, opt, bc_fn, fields, ref_cell, need_ref):
    om = oracle_shard_mesh(sh["mesh"])
    c = OracleQhdCase(om, opt)
    bc_fn(c, sh["mesh"])
    cg = sh["cell_global"]
    c.set_fields(fields[0][cg], fields[1][cg], fields[2][cg])
    hit = np.nonzero(cg[sh["owned"]] == ref_cell)[0]
    c.set_reference(need_ref, int(sh["owned"][hit[0]]) if hit.size else -1)
    return c


def gather(shards, cases, name, n_global, ncomp=1):
    out = np.full((n_global, ncomp) if ncomp > 1 else (n_global,), np.nan)
    for sh, c in zip(shards, cases):
        out[sh["cell_global"][sh["owned"]]] = c.field(name)[sh["owned"]]
    assert not np.isnan(out).any()
    return out
