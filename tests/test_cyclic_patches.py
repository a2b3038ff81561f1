"""Translational cyclic patch pairs served by ghost cells (qgd_mesh_unroll_cyclic): the two halves of a pair are glued -- behind
each half sit translated copies of the cells that touch the other half, refreshed from their originals once per step by the halo pack / unpack of
the same rank.  What is checked, with properties a shared mistake of oracle and device could not satisfy:
  * the extended mesh is consistent (every real cell / point / patch face keeps its label, the cyclic patches are empty, the copies of the two
    sides of a pair mirror each other);
  * a box periodic in x reproduces, cell for cell, the middle third of the SAME case on a three times longer box without any cyclic patch
    (information travels a few cells per step: for the first steps the middle third cannot know the long box ends);
  * a triply periodic box conserves mass, momentum and total energy to rounding (edge and corner copies included);
  * a pulse carried by a uniform flow leaves through one half and re-enters through the other;
  * rotational or unmatched halves are refused by name;
and, on the GPU, the device case -- plain qgd_case_step, the library refreshes the copies itself -- against the oracle driven by hand, through the
fused one-launch step and through the separate kernels."""
import numpy as np
import pytest

import qgdsolver_amd as q
from qgdsolver_amd import _lib as L

import cases
from oracle import OracleCase, OracleMesh

G, CYC = L.PATCH_GENERIC, L.PATCH_CYCLIC


def periodic_box(n, periodic=(True, False, False), lo=(0.0, 0.0, 0.0), hi=(1.0, 1.0, 1.0), jitter=0.0):
    pt = []
    for d in range(3):
        pt += [CYC, CYC] if periodic[d] else [G, G]
    mesh = q.PolyMesh.box(n[0], n[1], n[2], lo=lo, hi=hi, patch_types=pt)
    return mesh


class OraclePeriodic:
    """the oracle on an unrolled mesh, its copies refreshed by hand: pack every slot, unpack what the partner slot packed"""

    def __init__(self, ext, options, bc_fn=None):
        self.ext = ext
        om = OracleMesh(ext.primitives())
        self.self_slot = [int(x) for x in ext.array("haloSelf")]
        for k in range(ext.halo_slots):
            om.set_halo(k, ext.array(f"haloGhost{k}"), ext.array(f"haloSend{k}"))
        om.set_halo_face_h(ext.array("haloFaceH"))
        self.case = OracleCase(om, options)
        if bc_fn:
            bc_fn(self.case)

    def exchange(self, mid=False):
        c = self.case
        bufs = []
        for k in range(len(self.self_slot)):
            if mid:
                b = np.zeros(c.mid_halo_count(k)[0]); c.mid_halo_pack(k, b)
            else:
                b = np.zeros(c.halo_count(k)); c.halo_pack(k, b)
            bufs.append(b)
        for k, src in enumerate(self.self_slot):
            if mid:
                assert bufs[src].size == c.mid_halo_count(k)[1]
                c.mid_halo_unpack(k, bufs[src])
            else:
                assert bufs[src].size == c.halo_recv_count(k)
                c.halo_unpack(k, bufs[src])
        if not mid:
            c.step_phase(2)

    def set_fields(self, U, T, p):
        cg = self.ext.array("cellGlobal")
        self.case.set_fields(U[cg], T[cg], p[cg])
        self.exchange()

    def step(self, n):
        c = self.case
        for _ in range(n):
            if c.needs_mid_exchange():
                c.step_phase(5); self.exchange(mid=True); c.step_phase(6)
            else:
                c.step_phase(0)
            c.step_phase(1)
            self.exchange()

    def field(self, name, n_real):
        return self.case.field(name)[:n_real]


def test_unrolled_mesh_keeps_the_real_mesh_and_mirrors_its_sides():
    g = periodic_box((6, 5, 4), (True, True, False))
    ext = g.unroll_cyclic()
    n = g.nCells
    assert ext.nCells > n and np.array_equal(ext.array("cellGlobal")[:n], np.arange(n))
    assert np.array_equal(ext.array("points")[:3 * g.nPoints], g.array("points"))
    ps, pz, pty = ext.array("patchStart"), ext.array("patchSize"), ext.array("patchType")
    assert ext.nPatches == g.nPatches + 1 and pty[-1] == L.PATCH_HALO and pz[-1] > 0
    assert all(pz[i] == 0 for i in range(g.nPatches) if pty[i] == CYC)
    # a box periodic in x and y: 8 tiles have copies (4 sides + 4 edges), each slot's ghosts are its partner's send list shifted
    self_slot = ext.array("haloSelf")
    assert ext.halo_slots == 8 and sorted(self_slot) == list(range(8)) and all(self_slot[self_slot[k]] == k for k in range(8))
    cg = ext.array("cellGlobal")
    C = ext.array("C").reshape(-1, 3)
    for k in range(8):
        ghost, send = ext.array(f"haloGhost{k}"), ext.array(f"haloSend{int(self_slot[k])}")
        assert np.array_equal(cg[ghost], send)               # the copies are exactly the cells the partner slot sends, in its order
        shift = C[ghost] - C[send]
        assert np.abs(shift - shift[0]).max() < 1e-12 and np.abs(shift[0]).max() > 0.5       # ... all shifted by ONE lattice vector
    # every cyclic face of the real mesh became an internal face: nothing of the real cells is left on the halo patch
    own = ext.array("owner")
    assert (own[ps[-1]: ps[-1] + pz[-1]] >= n).all()
    # the zMin / zMax patches got the copies' faces behind the real ones, which kept their order
    fg = ext.array("faceGlobal")
    for i in (4, 5):
        real = fg[ps[i]: ps[i] + g.array("patchSize")[i]]
        assert np.array_equal(real, np.arange(g.array("patchStart")[i], g.array("patchStart")[i] + g.array("patchSize")[i]))
        assert pz[i] > g.array("patchSize")[i]


def test_rotational_or_unmatched_halves_are_refused_by_name():
    g = periodic_box((4, 4, 4), (True, False, False))
    with pytest.raises(q.QgdError, match="not both cyclic"):
        g.unroll_cyclic([(0, 2)])
    h = q.PolyMesh.box(4, 4, 4, patch_types=[CYC, G, CYC, G, G, G])       # xMin paired with yMin: not translates of each other
    with pytest.raises(q.QgdError, match="not translates"):
        h.unroll_cyclic([(0, 2)])
    plain = q.PolyMesh.box(4, 4, 4)
    with pytest.raises(q.QgdError, match="no cyclic patches"):
        plain.unroll_cyclic()


def tiled_fields(fields_fn, C, period):
    """fields of the periodic case evaluated on any box: x taken modulo the period"""
    Cm = C.copy()
    Cm[:, 0] = np.mod(Cm[:, 0], period)
    return fields_fn(Cm)


def smooth_fields(C):
    x, y, z = C[:, 0], C[:, 1], C[:, 2]
    U = np.stack([0.3 + 0.1 * np.sin(2 * np.pi * x) * np.cos(np.pi * y), 0.05 * np.cos(2 * np.pi * x) * np.sin(np.pi * y),
                  0.02 * np.sin(4 * np.pi * x) * np.sin(np.pi * z)], axis=1)
    T = 1.0 + 0.05 * np.cos(2 * np.pi * x) * np.cos(np.pi * y)
    p = 1.0 + 0.1 * np.sin(2 * np.pi * x + 0.3) * np.cos(np.pi * z)
    return U, T, p


def wall_bcs(case):
    """slip walls with the qgdFlux pressure condition on y, zeroGradient on z (patches 2..5; 0, 1 are the cyclic halves)"""
    for patch in (2, 3):
        case.set_bc(patch, U=("slip", None), T=("zeroGradient", None), p=("qgdFlux", None))


@pytest.mark.parametrize("stencil,bc_fn", [("GaussVolPoint", None), ("GaussVolPoint", wall_bcs), ("reduced", None)])
def test_periodic_box_is_the_middle_of_a_three_times_longer_box(stencil, bc_fn):
    nx, ny, nz, steps = 8, 5, 4, 3
    opt = q.default_options(stencil=stencil, deltaT=2e-3, mu=1e-3)
    g = periodic_box((nx, ny, nz), (True, False, False))
    ext = g.unroll_cyclic()
    per = OraclePeriodic(ext, opt, bc_fn)
    per.set_fields(*smooth_fields(g.array("C").reshape(-1, 3)))
    per.step(steps)
    # the same case on [-1, 2) x [0, 1)^2 with ordinary patches at its far ends
    long = q.PolyMesh.box(3 * nx, ny, nz, lo=(-1.0, 0.0, 0.0), hi=(2.0, 1.0, 1.0))
    oc = OracleCase(OracleMesh(long.primitives()), opt)
    if bc_fn:
        bc_fn(oc)
    oc.set_fields(*tiled_fields(smooth_fields, long.array("C").reshape(-1, 3), 1.0))
    oc.step(steps)
    idx = np.arange(long.nCells).reshape(nz, ny, 3 * nx)[:, :, nx: 2 * nx].reshape(-1)
    for f in ("rho", "U", "p", "e"):
        a, b = per.field(f, g.nCells), oc.field(f)[idx]
        assert np.abs(a - b).max() <= 1e-12 * np.abs(b).max(), (stencil, f, np.abs(a - b).max())


def conserved(case, V, n):
    rho, U, rhoE = case.field("rho")[:n], case.field("U")[:n], case.field("rhoE")[:n]
    return np.array([(rho * V).sum(), *((rho[:, None] * U) * V[:, None]).sum(axis=0), (rhoE * V).sum()])


def test_triply_periodic_box_conserves_mass_momentum_and_energy():
    g = periodic_box((6, 5, 4), (True, True, True))
    ext = g.unroll_cyclic()
    assert ext.halo_slots == 26                                   # 6 sides, 12 edges, 8 corners
    per = OraclePeriodic(ext, q.default_options(stencil="GaussVolPoint", deltaT=2e-3, mu=1e-3))
    C = g.array("C").reshape(-1, 3)
    U, T, p = smooth_fields(C)
    U[:, 2] += 0.04 * np.sin(2 * np.pi * C[:, 1])
    T = T + 0.03 * np.sin(2 * np.pi * C[:, 2])
    per.set_fields(U, T, p)
    V = g.array("V")
    before = conserved(per.case, V, g.nCells)
    per.step(12)
    after = conserved(per.case, V, g.nCells)
    scale = np.array([before[0], before[0], before[0], before[0], before[4]])
    assert (np.abs(after - before) <= 1e-13 * scale).all(), (after - before) / scale
    assert np.abs(per.case.field("rho")[:g.nCells] - 1.0).max() > 1e-3      # (something happened)


def test_a_pulse_leaves_through_one_half_and_re_enters_through_the_other():
    nx = 40
    g = periodic_box((nx, 3, 3), (True, True, True), hi=(1.0, 0.075, 0.075))
    ext = g.unroll_cyclic()
    C = g.array("C").reshape(-1, 3)
    U = np.zeros_like(C); U[:, 0] = 2.5                            # supersonic carrier (c = 1): everything moves to the right
    T = np.ones(g.nCells)
    p = 1.0 + 0.05 * np.exp(-((C[:, 0] - 0.85) / 0.05) ** 2)
    per = OraclePeriodic(ext, q.default_options(stencil="GaussVolPoint", deltaT=1e-3))
    per.set_fields(U, T, p)
    x0 = C[np.argmax(per.field("p", g.nCells)), 0]
    per.step(200)     # t = 0.2: the pressure pulse has split into its two acoustic halves, at u + c = 3.5 -> 0.85 + 0.70 = 1.55 -> 0.55 and at
    pn = per.field("p", g.nCells)                                  # u - c = 1.5 -> 0.85 + 0.30 = 1.15 -> 0.15: both have left through xMax and re-entered at xMin
    x1 = C[np.argmax(pn), 0]
    assert x0 > 0.8 and 0.05 < x1 < 0.65, (x0, x1)
    assert pn.max() - 1.0 > 5e-3                                   # still a pulse, not noise
    # the same pulse in a box whose ends are ordinary (zeroGradient) patches just leaves: nothing like it is found in x < 0.65 then
    pt = [G, G, CYC, CYC, CYC, CYC]
    h = q.PolyMesh.box(nx, 3, 3, hi=(1.0, 0.075, 0.075), patch_types=pt)
    he = h.unroll_cyclic()
    out = OraclePeriodic(he, q.default_options(stencil="GaussVolPoint", deltaT=1e-3))
    out.set_fields(U, T, p)
    out.step(200)
    po = out.field("p", h.nCells)
    assert po[C[:, 0] < 0.65].max() - 1.0 < 0.2 * (pn.max() - 1.0), (po.max(), pn.max())


# ---- the device ----------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("arm", ["fused", "kernels"])
@pytest.mark.parametrize("periodic,bc_fn,stencil", [((True, False, False), wall_bcs, "GaussVolPoint"), ((True, True, True), None, "GaussVolPoint"),
                                                    ((True, True, False), None, "reduced")])
def test_device_steps_a_periodic_case_like_the_oracle(periodic, bc_fn, stencil, arm):
    if arm == "fused" and stencil != "GaussVolPoint":
        pytest.skip("the fused step serves GaussVolPoint")
    g = periodic_box((10, 8, 6), periodic)
    ext = g.unroll_cyclic()
    opt = q.default_options(stencil=stencil, deltaT=1e-3, mu=1e-3)
    U, T, p = smooth_fields(g.array("C").reshape(-1, 3))
    per = OraclePeriodic(ext, opt, bc_fn)
    per.set_fields(U, T, p)
    dev = q.Device(ext, fused_tables="any" if arm == "fused" else False)
    gc = q.QGDFoamCase(dev, opt)
    assert gc.fused_info()["fused"] == (arm == "fused")
    if bc_fn:
        bc_fn(gc)
    cg = ext.array("cellGlobal")
    gc.set_fields(U[cg], T[cg], p[cg])
    for chunk in (1, 9):
        gc.step(chunk)                                             # plain step: the library refreshes the copies itself
        per.step(chunk)
        for f in ("rho", "U", "p", "e"):
            a, b = gc.field(f)[:g.nCells], per.field(f, g.nCells)
            assert np.abs(a - b).max() <= 1e-10 * np.abs(b).max(), (periodic, stencil, arm, f)
        # the copies hold their originals' records
        assert np.array_equal(gc.field("rho")[g.nCells:], gc.field("rho")[cg[g.nCells:]])
    with pytest.raises(q.QgdError, match="implicitDiffusion"):
        ic = q.QGDFoamCase(dev, q.default_options(stencil=stencil, deltaT=1e-3, mu=1e-3, implicitDiffusion=1))
        ic.set_fields(U[cg], T[cg], p[cg])
        ic.step(1)
    from qgdsolver_amd import qhdfoam
    with pytest.raises(q.QgdError, match="unrolled"):
        qhdfoam.QHDFoamCase(dev, qhdfoam.qhd_options(stencil=stencil, deltaT=1e-3))
    gc.close(); dev.close()


def write_periodic_case(case_dir, n=(10, 6, 4)):
    """a channel periodic in x between slip walls (qgdFlux pressure) in y, zeroGradient in z, written the way a QGDFoam user would: the
    dictionaries of the forwardStep test case, this mesh (`neighbourPatch` entries in constant/polyMesh/boundary) and non-uniform fields"""
    import os
    import shutil
    from qgdsolver_amd import foamfile as ff
    from test_foamfile import write_step_case

    write_step_case(case_dir, "GaussVolPoint")
    shutil.rmtree(os.path.join(case_dir, "constant", "polyMesh"))
    mesh = periodic_box(n, (True, False, False))
    mesh.patch_names = ["left", "right", "bottom", "top", "back", "front"]
    mesh.cyclic_pairs = [(0, 1)]
    ff.write_polymesh(mesh, os.path.join(case_dir, "constant", "polyMesh"))
    U, T, p = smooth_fields(mesh.array("C").reshape(-1, 3))
    cyc = ("cyclic", None)
    ff.write_field(os.path.join(case_dir, "0", "U"), mesh, "U", U,
                   {"left": cyc, "right": cyc, "bottom": ("slip", None), "top": ("slip", None), "back": ("zeroGradient", None), "front": ("zeroGradient", None)})
    ff.write_field(os.path.join(case_dir, "0", "T"), mesh, "T", T,
                   {"left": cyc, "right": cyc, **{k: ("zeroGradient", None) for k in ("bottom", "top", "back", "front")}})
    ff.write_field(os.path.join(case_dir, "0", "p"), mesh, "p", p,
                   {"left": cyc, "right": cyc, "bottom": ("qgdFlux", None), "top": ("qgdFlux", None), "back": ("zeroGradient", None), "front": ("zeroGradient", None)})
    return mesh, (U, T, p)


def test_reader_pairs_the_halves_and_refuses_what_is_not_served(tmp_path):
    from qgdsolver_amd import foamfile as ff

    mesh, _ = write_periodic_case(str(tmp_path / "ok"))
    m2, opt, fields, bcs = ff.read_case_setup(str(tmp_path / "ok"))
    assert m2.cyclic_pairs == [(0, 1)] and bcs[0]["U"] == ("none", None) and bcs[2]["p"] == ("qgdFlux", None)
    ext = m2.unroll_cyclic(m2.cyclic_pairs)
    assert ext.nCells == mesh.nCells + 2 * 6 * 4 and np.array_equal(ext.array("haloSelf"), [1, 0])
    # a half that does not name its partner, a rotational pair, a field that is not `cyclic` on a cyclic patch
    bpath = str(tmp_path / "ok" / "constant" / "polyMesh" / "boundary")
    text = open(bpath).read()
    assert "neighbourPatch  right;" in text
    for bad, match in ((text.replace("neighbourPatch  right;", ""), "neighbourPatch"),
                       (text.replace("neighbourPatch  right;", "neighbourPatch right; transform rotational;"), "rotational")):
        open(bpath, "w").write(bad)
        with pytest.raises(ff.FoamFileError, match=match):
            ff.read_case_setup(str(tmp_path / "ok"))
    open(bpath, "w").write(text)
    upath = str(tmp_path / "ok" / "0" / "U")
    utext = open(upath).read()
    open(upath, "w").write(utext.replace("cyclic", "zeroGradient", 1))
    with pytest.raises(ff.FoamFileError, match="inconsistent patch and patchField types"):
        ff.read_case_setup(str(tmp_path / "ok"))


@pytest.mark.gpu
def test_application_runs_a_periodic_case_directory(tmp_path):
    """python -m qgdsolver_amd.QGDFoam on a case with a cyclic pair: the written fields are the oracle's (copies refreshed by hand), cell for cell"""
    import os
    from qgdsolver_amd import QGDFoam, foamfile as ff

    case_dir = str(tmp_path)
    mesh, (U, T, p) = write_periodic_case(case_dir)
    cd = os.path.join(case_dir, "system", "controlDict")
    text = open(cd).read()
    open(cd, "w").write(text.replace("endTime 1;", "endTime 0.01;") + "writeControl timeStep;\nwriteInterval 10;\n")
    dev, case, written = QGDFoam.run(case_dir, n_steps=20, log=lambda *a, **k: None)
    assert written == ["0.005", "0.01"]
    opt = q.default_options(stencil="GaussVolPoint", deltaT=5e-4, R=1 / 1.4, Cv=1 / 1.4 / 0.4, mu=0.0, Pr=1.0, ScQGD=1.0, PrQGD=1.0, alphaQGD=0.5)
    per = OraclePeriodic(mesh.unroll_cyclic([(0, 1)]), opt, wall_bcs)
    per.set_fields(U, T, p)
    per.step(20)
    for name in ("U", "T", "p", "rho"):
        vals, patches = ff.read_field(os.path.join(case_dir, "0.01", name), mesh)
        want = per.field(name, mesh.nCells)
        assert np.abs(vals.reshape(want.shape) - want).max() <= 1e-10 * np.abs(want).max(), name
        assert patches["left"]["type"] == "cyclic"
    case.close(); dev.close()
