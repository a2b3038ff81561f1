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


def _distributed():
    """(rank, world, local_rank) from the torch.distributed.run environment"""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def run(case_dir, n_steps=None, device_id=0, write=True, log=print, renumber="none", backend="nccl"):
    """One rank of the run.  With WORLD_SIZE > 1 (``python -m torch.distributed.run ... -m qgdsolver_amd.QGDFoam``) the
    undecomposed case is read by every rank, relabelled (``renumber``: none | rcm | morton), cut into contiguous cell
    ranges (``PolyMesh.shard``) and advanced with one halo message per neighbouring rank per step (``halo.RangeHalo``
    over RCCL; backend "gloo" stages the messages through host memory and lets all ranks share GPU 0: a debugging
    aid).  Rank 0 gathers the owned cells and writes the time directories in the case's own cell order."""
    rank, world, local_rank = _distributed()
    if rank != 0:
        log = lambda *a, **k: None  # noqa: E731
    cd = ff.read_dict(os.path.join(case_dir, "system", "controlDict"))
    t0, t0_name = find_start_time(case_dir, cd)
    from .fvsc import Device
    from .qgdfoam import QGDFoamCase, default_options

    gmesh, opt, fields, bcs = ff.read_case_setup(case_dir, t0_name)
    n_global = gmesh.nCells
    new_of_old = np.arange(n_global, dtype=np.int32)
    if renumber != "none":
        new_of_old = gmesh.rcm_order() if renumber == "rcm" else gmesh.morton_order()
        gmesh.renumber(new_of_old)
    old_of_new = np.argsort(new_of_old)
    dist = halo = torch = None
    if world > 1:
        import torch
        import torch.distributed as dist
        from .halo import HostStaged, RangeHalo

        staged = backend == "gloo"
        device_id = 0 if staged else local_rank
        torch.cuda.set_device(device_id)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if staged:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", device_id))
        mesh = gmesh.shard(world, rank)
        cells = old_of_new[mesh.array("cellGlobal")]  # labels in the case files of the shard's cells
        lo, hi = (n_global * rank) // world, (n_global * (rank + 1)) // world
        owned = (mesh.array("cellGlobal") >= lo) & (mesh.array("cellGlobal") < hi)
    else:
        mesh, cells, owned = gmesh, old_of_new, np.ones(n_global, dtype=bool)
    if getattr(gmesh, "cyclic_pairs", None):
        # translational cyclic pairs: the halves are glued, translated copies of the cells behind either half are refreshed by the library
        # after every step (PolyMesh.unroll_cyclic, DESIGN 6 "cyclic patches"); real cells keep their labels, the copies follow them
        if world > 1 or opt.get("implicitDiffusion") or renumber != "none":
            raise ff.FoamFileError(f"{case_dir}: a case with cyclic patches runs on one rank, explicit branch (QGD {{ implicitDiffusion false; }}), "
                                   "in the case's own cell order")
        mesh = gmesh.unroll_cyclic(gmesh.cyclic_pairs)
        cells = mesh.array("cellGlobal")
        owned = np.arange(mesh.nCells) < n_global
    # only a fixed-deltaT case with one GaussVolPoint stencil uses the cell blocks: no block tables otherwise
    # (the blocks serve the explicit branch's one-launch step, shards included, and the implicitDiffusion branch's assembly of the U systems on one rank)
    eligible = (not (opt.get("adjustTimeStep") or opt.get("termStencils")) and opt["stencil"] == "GaussVolPoint"
                and not (opt.get("implicitDiffusion") and world > 1))
    dev = Device(mesh, device_id, fv_schemes={"fvsc": {"default": opt["stencil"]}}, fused_tables=eligible)
    case = QGDFoamCase(dev, default_options(**opt))
    for i, bc in enumerate(bcs):
        case.set_bc(i, U=bc["U"], T=bc["T"], p=bc["p"])
    if "alphaQGD" in fields or "ScQGD" in fields:
        # non-uniform alphaQGD / ScQGD files: cell values follow the relabelling; boundary faces keep their labels under
        # renumbering, a shard's real patch faces are found through faceGlobal (cut faces carry no patch value)
        nif_g = gmesh.nInternalFaces
        if world > 1:
            fg = mesh.array("faceGlobal")[mesh.nInternalFaces:]
            gb = np.where(fg >= nif_g, fg - nif_g, -1)
        else:
            gb = np.arange(gmesh.nBoundaryFaces)

        def local(pair):
            if pair is None:
                return None
            cellv, bndv = pair
            b = np.where(gb >= 0, bndv[np.maximum(gb, 0)], cellv[cells][mesh.array("owner")[mesh.nInternalFaces:]])
            return cellv[cells], b
        case.set_qgd_coeffs(alphaQGD=local(fields.get("alphaQGD")), ScQGD=local(fields.get("ScQGD")))
    case.set_fields(fields["U"][cells], fields["T"][cells], fields["p"][cells])
    adjust = bool(case.options.adjustTimeStep)
    if world > 1:
        case.set_stream(torch.cuda.current_stream().cuda_stream)
        if staged:
            class Halo(HostStaged, RangeHalo):
                pass
            Halo.torch = torch
            halo = Halo(case, rank, world, dist, alloc=lambda c: torch.empty(c, dtype=torch.float64), arg=None)
        else:
            halo = RangeHalo(case, rank, world, dist, alloc=lambda c: torch.empty(c, dtype=torch.float64, device="cuda"),
                             arg=lambda t: t.data_ptr())
        from .halo import allreduce_max_of
        allreduce_max = allreduce_max_of(torch, dist, case, staged)
        stepper = None
        if case.options.implicitDiffusion:
            # the reference's default branch on shards: phases 20..35 with their messages and all-reduced solver scalars
            # (include/qgd_amd.h "the implicitDiffusion branch on a cell-range shard")
            import ctypes as C
            from . import _lib as L
            from .halo import DistWorld, ImplicitShard, ImplicitStepper, device_tensor
            peers = [int(p) for p in mesh.array("haloPeer")]
            if staged:
                def to_t(ptr, cnt):
                    return torch.from_numpy(dev.to_host(ptr, (int(cnt),)))

                def from_t(t, ptr):
                    a = np.ascontiguousarray(t.numpy())
                    if a.nbytes:
                        L.check(L.lib.qgd_device_copy(dev._h, C.c_void_p(ptr), a.ctypes.data_as(C.c_void_p), a.nbytes, 1), "qgd_device_copy")
            else:
                to_t = lambda ptr, cnt: device_tensor(torch, ptr, cnt)     # noqa: E731  (views of the library's buffers)
                from_t = lambda t, ptr: None                               # noqa: E731
            stepper = ImplicitStepper(DistWorld(ImplicitShard(case), dist, torch, peers, to_t, from_t, kinds=range(5), device_reduce=not staged))
        else:
            halo.exchange()

    def advance(n):
        if world == 1:
            case.step(n)
            return
        if stepper is not None:
            stepper.step(n, (lambda: allreduce_max(case)) if adjust else None)
        else:
            for _ in range(n):
                halo.step(allreduce_max if adjust else None)
        case.sync()

    def gather(name):
        """owned cells of every rank -> the field in the case's own cell order (rank 0; None elsewhere)"""
        local = case.field(name)
        if world == 1:
            out = np.empty((n_global,) + local.shape[1:])
            out[cells[owned]] = local[owned]     # (a mesh with cyclic patches carries copies of its cells behind the real ones)
            return out
        parts = [None] * world if rank == 0 else None
        dist.gather_object((cells[owned], local[owned]), parts, dst=0)
        if rank != 0:
            return None
        out = np.empty((n_global,) + local.shape[1:])
        for idx, vals in parts:
            out[idx] = vals
        return out

    dt = float(cd["deltaT"])
    end_time = float(cd["endTime"])
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
    branch = "implicitDiffusion true" if opt.get("implicitDiffusion") else "explicit branch"
    log(f"QGDFoam (qgdsolver_amd, {branch}): {n_global} cells on {world} rank(s), fvsc {opt['stencil']}, "
        f"deltaT {dt:g}, start {t0_name}, cell order {renumber}")
    if opt.get("implicitDiffusion"):
        x = os.environ.get("QGD_IMPL_XEXTRAP", "3")
        if x != "0":
            log(f"  NOTE: the U and e solves start from the predictor + the correction of the last steps extrapolated in time (QGD_IMPL_XEXTRAP={x}; "
                + ("Lagrange weights from the deltaT ratios under adjustTimeStep; " if adjust else "")
                + "limited per value), not from the predictor alone as OpenFOAM does: same systems, same tolerance, same answer to that "
                "tolerance -- but the 'initial' residuals and iteration counts below are NOT comparable with a reference QGDFoam log.  "
                "QGD_IMPL_XEXTRAP=0 restores OpenFOAM's start values.")
    done = 0
    wall0 = _time.perf_counter()
    written = []
    while True:
        if total is not None and done >= total:
            break
        n = chunk if total is None else min(chunk, total - done)
        if total is None:
            # adjustTimeStep: runTime.run() is asked after every step [QGDFoam.C L90]; the write cadence stays `chunk`
            n_done = 0
            while n_done < n:
                advance(1)
                n_done += 1
                if t0 + case.info()["time"] >= end_time - 1e-12 * max(1.0, abs(end_time)):
                    break
            n = n_done
        else:
            advance(n)
        done += n
        info = case.info()
        if world > 1:
            mins = [None] * world
            dist.all_gather_object(mins, (info["minRho"], info["minE"], info["CoNum"]))
            info["minRho"], info["minE"] = min(m[0] for m in mins), min(m[1] for m in mins)
            info["CoNum"] = max(m[2] for m in mins)
        t = t0 + info["time"]
        log(f"Time = {time_name(t, precision)}  steps {done}  deltaT {info['deltaT']:.6g}  Courant max {info['CoNum']:.6g}  "
            f"min rho {info['minRho']:.6g}  min e {info['minE']:.6g}  ClockTime {_time.perf_counter() - wall0:.2f} s")
        if opt.get("implicitDiffusion"):
            ii = case.implicit_info()
            log("  " + "  ".join(f"{k}: {v['initial']:.3g} -> {v['final']:.3g} in {v['iterations']}" for k, v in ii["solves"].items()))
            if ii["unconverged_steps"]:
                log(f"  WARNING: {ii['unconverged_steps']} step(s) so far in which an implicit solve stopped above its tolerance "
                    f"(implicitTol {case.options.implicitTol:g}, maxIter {case.options.implicitMaxIter})")
            if ii["stalled_steps"]:
                log(f"  NOTE: {ii['stalled_steps']} step(s) so far in which a Chebyshev solve ended at the rounding floor of its residual, above "
                    f"implicitTol {case.options.implicitTol:g} (OpenFOAM would have iterated on to maxIter)")
        if not np.isfinite(info["minRho"]) or info["minRho"] <= 0:
            raise FloatingPointError(f"density lost positivity at time {t:g}")
        if write:
            name = time_name(t, precision)
            data = {f: gather(f) for f in ("U", "T", "p", "rho")}
            if rank == 0:
                _write_cell_fields(case_dir, name, data, bcs, gmesh)
            written.append(name)
        if total is None and t >= end_time - 1e-12 * max(1.0, abs(end_time)):
            break
    log("End")
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return dev, case, written


def _write_cell_fields(case_dir, name, data, bcs, file_mesh):
    """Time directory from gathered cell fields.  Patch entries carry the BC type; values of fixedValue patches are
    the prescribed ones, the others are written without a value (the solver evaluates them at start-up)."""
    dims = {"U": "[0 1 -1 0 0 0 0]", "T": "[0 0 0 1 0 0 0]", "p": "[1 -1 -2 0 0 0 0]", "rho": "[1 -3 0 0 0 0 0]"}
    names = file_mesh.patch_names
    pt = file_mesh.array("patchType")
    for fname, values in data.items():
        patches = {}
        for i, pn in enumerate(names):
            word = ff.PATCH_WORDS.get(int(pt[i]), "patch")
            if word in ff._CONSTRAINT_BCS:
                patches[pn] = (word, None)
                continue
            kind, val = bcs[i].get(fname, ("calculated", None)) if fname != "rho" else ("calculated", None)
            if kind == "none":
                kind = "calculated"
            patches[pn] = (kind, np.asarray(val, dtype=np.float64) if (kind == "fixedValue" and val is not None) else None)
        ff.write_field(os.path.join(case_dir, str(name), fname), file_mesh, fname, values, patches, dims[fname])


def main(argv=None):
    ap = argparse.ArgumentParser(prog="QGDFoam", description=__doc__.split("\n\n")[0])
    ap.add_argument("-case", dest="case", default=".")
    ap.add_argument("-nSteps", dest="n_steps", type=int, default=None, help="run this many steps instead of up to endTime")
    ap.add_argument("-device", dest="device", type=int, default=0)
    ap.add_argument("-noWrite", dest="no_write", action="store_true")
    ap.add_argument("-renumber", dest="renumber", default="none", choices=["none", "rcm", "morton"],
                    help="relabel the cells on the device side (results are written in the case's own order)")
    ap.add_argument("-backend", dest="backend", default="nccl", choices=["nccl", "gloo"],
                    help="multi-rank transport; gloo = host-staged messages, all ranks on GPU 0 (debugging)")
    a = ap.parse_args(argv)
    dev, case, _ = run(a.case, a.n_steps, a.device, not a.no_write, renumber=a.renumber, backend=a.backend)
    case.close()
    dev.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
