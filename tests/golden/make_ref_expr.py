#!/usr/bin/env python3
"""Golden vectors evaluated MECHANICALLY from the text of the reference's own source listings.

The reference snapshot (/root/reference) holds only Doxygen listings; nothing of it compiles here (OpenFOAM is absent), so
the oracle cannot be checked against a reference build.  What CAN be done is to take the scalar/vector/tensor C++
expressions of the densest arithmetic on the hot path straight out of the listing text, transliterate the C++ statement
syntax (declarations, forAll, if/else, `::`, `!`) into Python WITHOUT touching the expressions themselves, and evaluate
them with small Python classes that emulate OpenFOAM's Vector/Tensor algebra (operators `&` inner product, `^` cross
product, `*` scaling / outer product, `.x()`, `tr`, `T`, `I`, `symmTensor`, `inv`, `det`; C++ and Python give `* / + -
& ^` the same relative precedence).  Indices, signs, operand order and slot assignments therefore come from the reference
text and not from anybody's reading of it.  No reference text is stored: only numbers (inputs + results) go to
tests/golden/ref_expr_*.npz.

Runs only where /root/reference exists (the build container):   python tests/golden/make_ref_expr.py

Snippets evaluated (listing lines of /root/reference/docs/html/<file>_source.html):
  gvp3d   GaussVolPointBase3D.C   L186-229 (triangle coefficients), L346-389 (quad coefficients),
                                  L490-512 (macro dfdxif), L750-757 / L831-854 (its invocations for scalar / vector fields)
  gvp3d_bnd  the same file's boundary-face text: mirror point and |vO - vN| L129-154, boundary triangle / quad coefficients
             L232-317 / L391-475, macro dfdxbf L515-539, its invocations with psin = patch value + snGrad*bmvON/2 (calcGradfBF
             L780-806 scalar, L877-919 vector; calcDivfBF L583-609 vector, L683-725 tensor)
  gvp2d   GaussVolPointBase2D.C   L154-168 (c1..c4), L317-328 (apply)
  gvp2d_bnd  the same file's boundary faces: v42 = 2 (Cf - C) L235-239, vertex choice and coefficients L244-288, psi2 = patch value +
             snGrad*|v42|/2 L343-346, apply L352-359
  reduced    reducedFaceNormalStencil.C L71, L85, L92, L105 (nf * snGrad, nf & snGrad: operand order and tensor layout)
  lsq_bnd    extendedFaceStencilScalarGrad.C L86-109: boundary faces of the leastSquares gradient (nf * snGrad on ordinary patches,
             the zero of L55 left on empty / wedge / coupled / symmetry / symmetryPlane patches)
  lsq     extendedFaceStencilCalculateWeights.C L64-153, extendedFaceStencilScalarGrad.C L66-72
  qhdface    QHDFoam/updateFields.H L36-73, QHDFoam/updateFluxes.H L33-38, QHDUEqn.H L36-43, QHDTEqn.H L65-66 (the face
             expressions qgd_qhd_fluxes returns), with the three fvsc::grad evaluated by the gvp3d text
  species    reactingLagrangianQGDFoam/updateFields.H L38 (Yf), updateFluxes.H L122-127 (one species of the forAll): what
             qgd_species_flux returns
  case2cell  QGDFoam/updateFields.H L45-80, QGDFoam/updateFluxes.H L41-139 (explicit branch), constScPrModel1.C L103-114,
             QGDCoeffs.C L305-307, with the four fvsc::grad evaluated by the gvp3d text; QGDRhoEqn.H L40-47, QGDUEqn.H L36-89,
             QGDEEqn.H L37-76 on the two-cell mesh (explicit branch)
  implicit2cell  the same cases with implicitDiffusion true: updateFluxes.H L95-111 / L131-135 (Pif, qf without the Navier-Stokes /
             Fourier parts, tauMC, phiTauMC), QGDUEqn.H L36-75 and QGDEEqn.H L37-64 incl. the implicit U and e solves and phiSigmaDotU
  gvp2d_vec  GaussVolPointBase.C L90-100 (re-pack of the per-component 2-D gradients: gradf component 3i+j = d_i U_j),
             GaussVolPointBase2D.C L386-396 (vector divergence), L447-450 + L464-484 (tensor divergence)
  gvp_other  GaussVolPointBase3D.C L945-948 / L976-979 (dfdn = nf * snGrad), L759-768 / L856-865 (faces with more than four vertices),
             fvscStencil.C L126-129 (nf)
  qgdlength  QGDCoeffs.C L195-199, L298-376 (updateQGDLength over whole small meshes: hQGDf, the area-weighted hQGD, patches)
  courant    QGDCourantNo.H L36-53, setDeltaT-QGDQHD.H L41-61, hePsiQGDThermo.C L123-124 (speed of sound)
  qhdclosure constTau.C L71-74, HbyUQHD.C L80-83, T0byGr.C L84-87, H2bynuQHD.C L78-82 (tauQGD of the four QHD closures)
  casebnd    the QGDFoam flux assembly on ONE BOUNDARY face: updateFields.H L45-80 with patch values, the boundary-face text of the 3-D
             stencil, updateFluxes.H L41-139, GaussVolPointStencil.C L73 -> qgdFluxFvPatchScalarField.C L184-192 (updateCoeffs with the
             fresh phiwStar), constScPrModel1.C L103-104, L121-128 (patch loop)
  specieseqn QGDYEqn.H L40-45, L69-92 (the species loop, explicit branch, three species with an inert one) on the two-cell mesh
  thermo2cell  hePsiQGDThermo.C L48-64 + L123-124 and QGDFoam.C L152-154 (thermo.correct(), p = rho / psi after the explicit step)
  qhdflux    qhdFluxFvPatchScalarField.C L193-203 (updateCoeffs with the registered flux) on the walls of a small cavity
  lsqorder   extendedFaceStencilFindNeighbours.C L48-84 (the stencil search: which cells, in which order) on whole small 2-D meshes
  specieseqn_implicit  QGDYEqn.H L47-66: the species loop through fvm::laplacian and YEqn.flux()
  qhdeqn_implicit  the same step through the listing's implicitDiffusion branch (QHDUEqn.H L46-65, QHDTEqn.H L69-80: fvm::laplacian)
  qhdeqn     one whole QHDFoam step on the two-cell mesh: updateFields.H L36-73, updateFluxes.H L33-38, QHDpEqn.H L35-47, QHDUEqn.H
             L36-43 + L46-85, QHDTEqn.H L65-66 + L69-92, QHDFoam.C L123-131 (reference level)

Every configuration is ONE internal face between cells whose centres are prescribed (qgd_mesh_set_geometry /
orc_mesh_set_geometry), so the public operators (fvsc grad, the QGDFoam case) can be run on it as they are; vertex values
follow the inverse-distance rule of volPointInterpolation (L0) from those cells.
"""
import html
import os
import re
import sys

import numpy as np

REF = "/root/reference/docs/html"
HERE = os.path.dirname(os.path.abspath(__file__))


# ----------------------------------------------------------------------------------------------------------------------
# listing text
# ----------------------------------------------------------------------------------------------------------------------
def listing(name):
    out = {}
    for line in open(os.path.join(REF, name), encoding="utf-8", errors="replace"):
        m = re.search(r'<a (?:id|name)="l(\d+)"', line)
        if m:
            t = html.unescape(re.sub(r"<[^>]+>", "", line))
            out[int(m.group(1))] = re.sub(r"^\s*\d+", "", t, count=1).rstrip("\n")
    return out


def lines(name, a, b):
    L = listing(name)
    return [L.get(i, "") for i in range(a, b + 1)]


# ----------------------------------------------------------------------------------------------------------------------
# C++ statement syntax -> Python statement syntax (expressions are left alone)
# ----------------------------------------------------------------------------------------------------------------------
TYPES = r"(?:const\s+)?(?:scalar|label|vector|tensor|symmTensor|bool|face|fvPatch|cell|surfaceScalarField|surfaceVectorField|volScalarField)\b\s*&?"


def strip_comments(src):
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    return "\n".join(re.sub(r"//.*$", "", ln) for ln in src.split("\n") if not ln.lstrip().startswith("#"))


def split_top(s, sep=","):
    parts, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == sep and depth == 0:
            parts.append(cur)
            cur = ""
        else:
            cur += ch
    parts.append(cur)
    return parts


def expr(e):
    """the only rewrites inside expressions: scope operator, logical operators, conditional operator, whitespace"""
    e = " ".join(e.split())
    q = split_top(e, "?")
    if len(q) == 2:                                   # cond ? a : b
        a, b = split_top(q[1], ":")
        return f"({expr(a)}) if ({expr(q[0])}) else ({expr(b)})"
    e = re.sub(r"refCast\s*<[^(]*\(", "refCast(", e)   # template arguments of a cast are not part of any arithmetic
    e = re.sub(r"isA<\s*(\w+)\s*>\s*\(", r'isA("\1", ', e)   # isA<emptyFvPatch>(fvp) -> isA("emptyFvPatch", fvp)
    e = e.replace("::", ".").replace("->", ".")
    e = re.sub(r"!(?!=)", " not ", e)
    e = e.replace("&&", " and ").replace("||", " or ")
    return e


def statement(st):
    st = " ".join(st.split())
    if not st:
        return []
    m = re.match(r"^List<\s*vector\s*>\s+(\w+)\((.*)\)$", st) or re.match(r"^scalarList\s+(\w+)\((.*)\)$", st)
    if m:
        return [f"{m.group(1)} = RList([None]*({expr(m.group(2))}))"]
    m = re.match(r"^symmTensor\s+(\w+)\((.*)\)$", st)
    if m:
        return [f"{m.group(1)} = symmTensor({expr(m.group(2))})"]
    m = re.match(r"^fv(?:Scalar|Vector)Matrix\s+(\w+)\s*\((.*)\)$", st)                # fvScalarMatrix EEqn(expr)
    if m:
        return [f"{m.group(1)} = {expr(m.group(2))}"]
    m = re.match(r"^(\w+)\.ref\(\)\s*=(?!=)\s*(.*)$", st)                              # U.ref() = expr: values into the existing field
    if m:
        return [f"{m.group(1)}.assign({expr(m.group(2))})"]
    m = re.match(r"^surface(?:Scalar|Vector|Tensor)Field\s+(\w+)\s*\((.*)\)$", st)        # surfaceScalarField Cof("Cof", expr) | dfdn(expr)
    if m:
        args = split_top(m.group(2))
        if len(args) == 2 and re.match(r'^\s*"\w+"\s*$', args[0]):
            return [f"{m.group(1)} = {expr(args[1])}"]
        return [f"{m.group(1)} = {expr(m.group(2))}"]
    m = re.match(r"^tmp<\s*\w+\s*>\s+(\w+)\s*\((.*)\)$", st)                          # tmp<surfaceVectorField> t(expr)
    if m:
        return [f"{m.group(1)} = {expr(m.group(2))}"]
    m = re.match(r"^List<\s*List<\s*\w+\s*>\s*>\s*(\w+)\s*\((.*)\)$", st)          # List<List<scalar> > psin (n)
    if m:
        return [f"{m.group(1)} = RList([None]*({expr(m.group(2))}))"]
    m = re.match(r"^(?:scalar|vector|tensor)Field\s+(\w+)\s*\((.*)\)$", st)          # vectorField v(n, vector::zero) | scalarField psio (expr)
    if m:
        args = split_top(m.group(2))
        if len(args) == 2:
            return [f"{m.group(1)} = Fld([None]*({expr(args[0])}))"]
        return [f"{m.group(1)} = {expr(m.group(2))}"]
    m = re.match(r"^labelList\s+(\w+)\s*(?:=\s*(.*))?$", st)                         # labelList a; | labelList a = expr;  (a copy)
    if m:
        return [f"{m.group(1)} = LabelList({expr(m.group(2)) if m.group(2) else ''})"]
    m = re.match(rf"^{TYPES}\s*(.*)$", st)
    if m and re.match(r"^\w+\s*(=|,|$)", m.group(1)):
        out = []
        for decl in split_top(m.group(1)):
            if "=" in decl:
                out += statement(decl)
        return out
    m = re.match(r"^(\w+)(\+\+|--)$", st)
    if m:
        return [f"{m.group(1)} {'+' if m.group(2) == '++' else '-'}= 1"]
    m = re.match(r"^(.*?\))\s*(\+=|-=|\*=|=)(?!=)\s*(.*)$", st)
    if m and re.match(r"^(\w+)\.(xx|yy|zz)\(\)$", m.group(1)):   # G0.xx() = 1
        obj, cmpt = re.match(r"^(\w+)\.(xx|yy|zz)\(\)$", m.group(1)).groups()
        return [f"{obj}.set('{cmpt}', {expr(m.group(3))})"]
    if m and re.match(r"^oop\((.*)\)$", m.group(1)):                      # oop(field[facei],ocmpt) += value
        tgt, cm = split_top(re.match(r"^oop\((.*)\)$", m.group(1)).group(1))
        return [f"oop_add({expr(tgt)}, {expr(cm)}, {expr(m.group(3))})"]
    if m and re.match(r"^\w+\(\)$", m.group(1)):                         # tauMCPtr() = ...
        return [f"{m.group(1)}.assign({expr(m.group(3))})"]
    m = re.match(r"^(.*?)\.resize\((.*)\)$", st)
    if m:
        return [f"{expr(m.group(1))}.resize({expr(m.group(2))})"]
    # plain assignment: the conditional operator, if any, belongs to the right-hand side
    depth = 0
    for k, ch in enumerate(st):
        depth += ch in "([{"
        depth -= ch in ")]}"
        if ch == "=" and depth == 0 and st[k + 1:k + 2] != "=" and st[k - 1] not in "=!<>":
            op = st[k - 1] + "=" if st[k - 1] in "+-*/" else "="
            lhs = st[:k - (len(op) - 1)]
            return [f"{expr(lhs)} {op} {expr(st[k + 1:])}"]
    return [expr(st)]


def transpile(src_lines, continuation=False):
    """statements, forAll, if/else blocks -> indented Python source"""
    src = "\n".join(ln.rstrip().rstrip("\\") if continuation else ln for ln in src_lines)
    src = strip_comments(src) + "\n"
    out, ind, i, n = [], 0, 0, len(src)
    pending_block = []

    def emit(s):
        out.append("    " * ind + s)

    buf = ""
    while i < n:
        ch = src[i]
        if ch == "{":
            head = " ".join(buf.split())
            buf = ""
            if head:
                m = re.match(r"^forAll\s*\((.*)\)$", head)
                if m:
                    lst, var = split_top(m.group(1))
                    emit(f"for {var.strip()} in range(len({expr(lst)})):")
                elif re.match(r"^if\s*\(", head):
                    emit(f"if {expr(head[2:].strip())}:")
                elif head == "else":
                    emit("else:")
                else:
                    raise SyntaxError("block head: " + head)
                pending_block.append(True)
            else:
                emit("if True:")
                pending_block.append(True)
            ind += 1
            emit("pass")
        elif ch == "}":
            if buf.strip():
                raise SyntaxError("unterminated statement: " + buf)
            ind -= 1
            pending_block.pop()
        elif ch == ";":
            for s in statement(buf):
                emit(s)
            buf = ""
        else:
            buf += ch
            # macro invocations written without a semicolon, one per line
            if ch == "\n" and re.match(r"^\s*dfdx[ib]f\(.*\)\s*$", buf):
                emit(expr(buf))
                buf = ""
        i += 1
    if buf.strip():
        raise SyntaxError("trailing text: " + buf)
    return "\n".join(out) + "\n"


# ----------------------------------------------------------------------------------------------------------------------
# OpenFOAM algebra (L0 semantics: Vector/Tensor/SymmTensor operators of OpenFOAM's primitives library)
# ----------------------------------------------------------------------------------------------------------------------
def _num(x):
    return isinstance(x, (int, float, np.floating, np.integer))


class Vec:
    def __init__(self, *c):
        self.c = np.array(c if len(c) == 3 else c[0], dtype=float)

    def x(self): return float(self.c[0])
    def y(self): return float(self.c[1])
    def z(self): return float(self.c[2])
    def component(self, i): return float(self.c[i])
    def __getitem__(self, i): return float(self.c[i])
    def __setitem__(self, i, v): self.c[i] = v
    def __add__(self, o): return Vec(self.c + o.c)
    def __sub__(self, o): return Vec(self.c - o.c)
    def __neg__(self): return Vec(-self.c)

    def __mul__(self, o):
        if _num(o):
            return Vec(self.c * o)
        if isinstance(o, Vec):                      # outer product: (a*b)_ij = a_i b_j
            return Tensor(np.outer(self.c, o.c))
        return NotImplemented

    def __rmul__(self, o):
        return Vec(o * self.c) if _num(o) else NotImplemented

    def __truediv__(self, o): return Vec(self.c / o)

    def __and__(self, o):
        if isinstance(o, Vec):
            return float(self.c[0] * o.c[0] + self.c[1] * o.c[1] + self.c[2] * o.c[2])
        if isinstance(o, Tensor):                   # (v & T)_j = v_i T_ij
            return Vec([sum(self.c[i] * o.m[i, j] for i in range(3)) for j in range(3)])
        return NotImplemented

    def __xor__(self, o):
        a, b = self.c, o.c
        return Vec(a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0])


class Sph:
    """sphericalTensor: I and its scalar multiples"""
    def __init__(self, s=1.0): self.s = float(s)
    def __mul__(self, o): return Sph(self.s * o) if _num(o) else NotImplemented
    def __rmul__(self, o): return Sph(self.s * o) if _num(o) else NotImplemented


class Tensor:
    def __init__(self, m): self.m = np.array(m, dtype=float).reshape(3, 3)

    def __add__(self, o):
        if isinstance(o, Sph):
            return Tensor(self.m + o.s * np.eye(3))
        return Tensor(self.m + o.m)

    def __radd__(self, o): return self.__add__(o)

    def __sub__(self, o):
        if isinstance(o, Sph):
            return Tensor(self.m - o.s * np.eye(3))
        return Tensor(self.m - o.m)

    def __mul__(self, o): return Tensor(self.m * o) if _num(o) else NotImplemented
    def __rmul__(self, o): return Tensor(self.m * o) if _num(o) else NotImplemented

    def __and__(self, o):
        if isinstance(o, Vec):                      # (T & v)_i = T_ij v_j
            return Vec([sum(self.m[i, j] * o.c[j] for j in range(3)) for i in range(3)])
        if isinstance(o, Tensor):
            return Tensor([[sum(self.m[i, k] * o.m[k, j] for k in range(3)) for j in range(3)] for i in range(3)])
        return NotImplemented

    def T(self): return Tensor(self.m.T.copy())


class symmTensor:
    """xx xy xz yy yz zz"""
    def __init__(self, *c):
        self.c = np.zeros(6) if (len(c) == 1 and c[0] == 0) else np.array(c, dtype=float)

    def xx(self): return float(self.c[0])
    def yy(self): return float(self.c[3])
    def zz(self): return float(self.c[5])
    def set(self, name, v): self.c[{"xx": 0, "yy": 3, "zz": 5}[name]] = v
    def __add__(self, o): return symmTensor(*(self.c + o.c))
    def __sub__(self, o): return symmTensor(*(self.c - o.c))
    def __iadd__(self, o): return symmTensor(*(self.c + o.c))
    def __mul__(self, o): return symmTensor(*(self.c * o)) if _num(o) else NotImplemented

    def __and__(self, v):
        xx, xy, xz, yy, yz, zz = self.c
        return Vec(xx * v.c[0] + xy * v.c[1] + xz * v.c[2], xy * v.c[0] + yy * v.c[1] + yz * v.c[2], xz * v.c[0] + yz * v.c[1] + zz * v.c[2])


def sqr(v): return symmTensor(v.c[0] * v.c[0], v.c[0] * v.c[1], v.c[0] * v.c[2], v.c[1] * v.c[1], v.c[1] * v.c[2], v.c[2] * v.c[2])
def magSqr(v):
    if isinstance(v, Nil):
        return v
    if isinstance(v, list):
        return type(v)([magSqr(a) for a in v])
    return float(v.c[0] * v.c[0] + v.c[1] * v.c[1] + v.c[2] * v.c[2])
def mag(v):
    if isinstance(v, Fld):
        return Fld([mag(a) for a in v])
    return abs(v) if _num(v) else float(np.sqrt(magSqr(v)))


def det(s):  # SymmTensorI.H
    xx, xy, xz, yy, yz, zz = s.c
    return float(xx * yy * zz + xy * yz * xz + xz * xy * yz - xx * yz * yz - xy * xy * zz - xz * yy * xz)


def inv(s):  # SymmTensorI.H: inv(st, det(st))
    xx, xy, xz, yy, yz, zz = s.c
    d = det(s)
    return symmTensor((yy * zz - yz * yz) / d, (xz * yz - xy * zz) / d, (xy * yz - xz * yy) / d, (xx * zz - xz * xz) / d,
                      (xy * xz - xx * yz) / d, (xx * yy - xy * xy) / d)


def tr(t): return float(t.m[0, 0] + t.m[1, 1] + t.m[2, 2])


class RList(list):
    def resize(self, n):
        del self[n:]
        self.extend([0.0] * (n - len(self)))

    def size(self): return len(self)


class Fld(list):
    """a field over the faces of one patch: elementwise arithmetic, scalars broadcast (scalarField / vectorField / tensorField)"""
    def _zip(self, o, f):
        return Fld([f(a, b) for a, b in zip(self, o)]) if isinstance(o, list) else Fld([f(a, o) for a in self])
    def __add__(self, o): return self._zip(o, lambda a, b: a + b)
    def __sub__(self, o): return self._zip(o, lambda a, b: a - b)
    def __mul__(self, o): return self._zip(o, lambda a, b: a * b)
    def __rmul__(self, o): return Fld([o * a for a in self])
    def __truediv__(self, o): return self._zip(o, lambda a, b: a / b)
    def __neg__(self): return Fld([-a for a in self])


class FieldWithPatches(list):
    """a volScalarField as the 2-D listing uses it: f[celli] and f.boundaryField()[patchi]"""
    def __init__(self, cells, patches):
        super().__init__(cells)
        self._patches = patches
    def boundaryField(self): return self._patches


class Obj:
    def __init__(self, **kw): self.__dict__.update(kw)


def call(v):
    return lambda: v


# ----------------------------------------------------------------------------------------------------------------------
# snippets
# ----------------------------------------------------------------------------------------------------------------------
def rnd_vec(rng, scale=1.0): return Vec(*(scale * rng.standard_normal(3)))


def skew_face(rng, nv):
    """a skewed, non-planar face near the unit square / triangle in the plane x = 0, with cell centres on either side"""
    base = [(0, 0, 0), (0, 1, 0), (0, 1, 1), (0, 0, 1)] if nv == 4 else [(0, 0, 0), (0, 1, 0), (0, 0.3, 1)]
    pts = [Vec(*(np.array(b, float) + 0.15 * rng.standard_normal(3))) for b in base]
    own = Vec(*(np.array([-0.5, 0.5, 0.5]) + 0.15 * rng.standard_normal(3)))
    nei = Vec(*(np.array([0.5, 0.5, 0.5]) + 0.15 * rng.standard_normal(3)))
    return pts, own, nei


def inv_dist(x, centres, vals):
    """L0 volPointInterpolation on a mesh whose every point belongs to the same cells: weights 1/|x - C_c|, normalised"""
    w = [1.0 / float(np.sqrt(((x.c - c.c) ** 2).sum())) for c in centres]
    sw = sum(w)
    w = [wi / sw for wi in w]
    acc = None
    for wi, v in zip(w, vals):
        acc = wi * v if acc is None else acc + wi * v
    return acc


def face_area_centre(pts):
    """some area vector and centre for the face (inputs of set_geometry, not part of what is checked)"""
    P = np.array([p.c for p in pts])
    c = P.mean(axis=0)
    S = np.zeros(3)
    for i in range(len(P)):
        S += 0.5 * np.cross(P[i] - c, P[(i + 1) % len(P)] - c)
    return S, c


class Cell0:
    """a scalar face value that `oop(field[facei], ocmpt) += x` with oop = SCA_CMPT can accumulate into"""
    def __init__(self): self.v = 0.0
    def __iadd__(self, x):
        self.v += x
        return self


class Gvp3dText:
    """the 3-D GaussVolPoint listing text, ready to be evaluated on one face"""
    def __init__(self):
        f3 = "GaussVolPointBase3D_8C_source.html"
        self.tri_src = transpile(lines(f3, 186, 229))
        self.qua_src = transpile(lines(f3, 346, 389))
        self.macro = transpile(lines(f3, 489, 513), continuation=True)
        self.inv_s = transpile(lines(f3, 749, 757))
        self.inv_v = transpile(lines(f3, 830, 854))
        self.inv_div_v = transpile(lines(f3, 552, 560))     # calcDivfIF(volVectorField): scalar out
        self.inv_div_t = transpile(lines(f3, 633, 659))     # calcDivfIF(volTensorField): vector out
        # boundary faces: mirror point and |vO - vN| [L129-154], triangle / quad coefficients [L232-317 / L391-475], the macro
        # dfdxbf [L515-539] and its invocations with psin = patch value + snGrad*bmvON/2 [calcGradfBF L780-806 (scalar),
        # L860-ff (vector), calcDivfBF L583-609 (vector), L683-725 (tensor)]
        self.bmv_src = transpile(lines(f3, 129, 154))
        self.btri_src = transpile(lines(f3, 232, 317))
        self.bqua_src = transpile(lines(f3, 391, 475))
        self.bmacro = transpile(lines(f3, 516, 539), continuation=True)
        self.binv = {"grad_s": transpile(lines(f3, 780, 806)), "grad_v": transpile(lines(f3, 877, 919)), "div_v": transpile(lines(f3, 583, 609)),
                     "div_t": transpile(lines(f3, 683, 725))}

    def coeffs(self, pts, own, nei):
        nv = len(pts)
        mesh = Obj(C=call([own, nei]), owner=call([0]), neighbour=call([1]))
        env = dict(points=pts, faces=[list(range(nv))], mesh=mesh, OneBySix=(1.0 / 6.0), own=0, nei=1, i=0, facei=0)
        if nv == 3:
            env.update(p1=0, p2=1, p3=2, vt_=[0.0], atx_=RList([RList()]), aty_=RList([RList()]), atz_=RList([RList()]))
            exec(self.tri_src, env)
            return env["atx_"], env["aty_"], env["atz_"], env["vt_"], mesh
        env.update(p1=0, p2=1, p3=2, p4=3, vq_=[0.0], aqx_=RList([RList()]), aqy_=RList([RList()]), aqz_=RList([RList()]))
        exec(self.qua_src, env)
        return env["aqx_"], env["aqy_"], env["aqz_"], env["vq_"], mesh

    def grad(self, pts, own, nei, cell_vals, pt_vals, vector, op="grad"):
        """the macro body becomes a function, its invocations [L750-757 / L831-854; div: L552-560 / L633-659] are executed as written"""
        nv = len(pts)
        ax, ay, az, vol, mesh = self.coeffs(pts, own, nei)
        faces = [list(range(nv))]
        grad = [np.zeros(9 if vector else 3)]
        if op == "div_v":
            grad = [Cell0()]        # a scalar the SCA_CMPT output operator can add into
        elif op == "div_t":
            grad = [np.zeros(3)]
        fld = Obj(mesh=call(mesh), primitiveField=call(cell_vals))
        out = Obj(primitiveFieldRef=call(grad))
        macro = self.macro

        def dfdxif(vf, pf, dfdxfield, fi, vi, ai, icmpt, ocmpt, iop, oop):
            def oop_add(target, cm, value):
                if isinstance(target, Cell0):   # oop = SCA_CMPT: the scalar itself
                    target.v += value
                else:
                    target[cm] += value
            exec(macro, dict(vf=vf, pf=pf, dfdxfield=dfdxfield, fi=fi, vi=vi, ai=ai, icmpt=icmpt, ocmpt=ocmpt, iop=iop, oop=oop,
                             faces=faces, oop_add=oop_add))

        e = dict(dfdxif=dfdxif, sf=fld, vf=fld, tf=fld, pf=pt_vals, gradf=out, divf=out, SCA_CMPT=lambda V, c: V, VEC_CMPT=lambda V, c: V[c],
                 qf_=[0] if nv == 4 else [], tf_=[0] if nv == 3 else [], vq_=vol if nv == 4 else [], vt_=vol if nv == 3 else [],
                 aqx_=ax if nv == 4 else RList(), aqy_=ay if nv == 4 else RList(), aqz_=az if nv == 4 else RList(),
                 atx_=ax if nv == 3 else RList(), aty_=ay if nv == 3 else RList(), atz_=az if nv == 3 else RList())
        exec({"grad": self.inv_v if vector else self.inv_s, "div_v": self.inv_div_v, "div_t": self.inv_div_t}[op], e)
        return (grad[0].v if op == "div_v" else grad[0]), (ax, ay, az, vol)

    def boundary(self, pts, own, Cf, cell_val, bnd_val, sn_grad, pt_vals, op):
        """fvsc gradient / divergence on ONE boundary face (patch 0, face 0) owned by a cell centred at `own`: everything from the
        text.  cell_val / bnd_val / sn_grad: the owner's value, the patch value and fvPatchField::snGrad() of the face (L0)."""
        nv = len(pts)
        patch = Obj(size=lambda: 1, Cn=lambda: Fld([own]), Cf=lambda: Fld([Cf]))
        mesh = Obj(boundary=call([patch]))
        base = dict(mesh=mesh, processorPatch_=[False], bgfid_=[RList([0])], Fld=Fld, RList=RList, mag=mag, refCast=None,
                    vector=Obj(zero=Vec(0, 0, 0)))
        env = dict(base, bmvON_=[None])
        exec(self.bmv_src, env)
        bmvON = env["bmvON_"]
        co = dict(base, points=pts, faces=[list(range(nv))], OneBySix=(1.0 / 6.0), facei=-1, p1=-1, p2=-1, p3=-1, p4=-1)
        empty = lambda: [RList()]   # noqa: E731  no faces of the other kind on the patch
        one = lambda: [RList([RList()])]   # noqa: E731
        if nv == 3:
            co.update(btf_=[RList([0])], batx_=one(), baty_=one(), batz_=one(), bvt_=[RList([0.0])])
            exec(self.btri_src, co)
            coef = dict(btf_=co["btf_"], batx_=co["batx_"], baty_=co["baty_"], batz_=co["batz_"], bvt_=co["bvt_"],
                        bqf_=[RList()], baqx_=empty(), baqy_=empty(), baqz_=empty(), bvq_=[RList()])
            ax, ay, az, vol = co["batx_"][0][0], co["baty_"][0][0], co["batz_"][0][0], co["bvt_"][0][0]
        else:
            co.update(bqf_=[RList([0])], baqx_=one(), baqy_=one(), baqz_=one(), bvq_=[RList([0.0])])
            exec(self.bqua_src, co)
            coef = dict(bqf_=co["bqf_"], baqx_=co["baqx_"], baqy_=co["baqy_"], baqz_=co["baqz_"], bvq_=co["bvq_"],
                        btf_=[RList()], batx_=empty(), baty_=empty(), batz_=empty(), bvt_=[RList()])
            ax, ay, az, vol = co["baqx_"][0][0], co["baqy_"][0][0], co["baqz_"][0][0], co["bvq_"][0][0]

        class PatchField(Fld):
            def snGrad(self): return Fld([sn_grad])
            def patchInternalField(self): return Fld([cell_val])
        fld = Obj(boundaryField=call(RList([PatchField([bnd_val])])))
        res = [Cell0()] if op == "div_v" else [np.zeros(9 if op == "grad_v" else 3)]
        out = Obj(boundaryFieldRef=call([res]))
        faces = [list(range(nv))]
        macro = self.bmacro
        e = dict(coef, sf=fld, vf=fld, tf=fld, pf=pt_vals, gradf=out, divf=out, processorPatch_=[False], bmvON_=bmvON, RList=RList, refCast=None,
                 SCA_CMPT=lambda V, c: V, VEC_CMPT=lambda V, c: V[c], bof_=[RList()], dfdn=None)

        def dfdxbf(vf, pf, patchi, dfdxfield, bfi, bvfi, bai, icmpt, ocmpt, iop, oop):
            def oop_add(target, cm, value):
                if isinstance(target, Cell0):
                    target.v += value
                else:
                    target[cm] += value
            exec(macro, dict(vf=vf, pf=pf, patchi=patchi, dfdxfield=dfdxfield, bfi=bfi, bvfi=bvfi, bai=bai, icmpt=icmpt, ocmpt=ocmpt, iop=iop,
                             oop=oop, faces=faces, bgfid_=[RList([0])], psin=e["psin"], psio=e["psio"], oop_add=oop_add))
        e["dfdxbf"] = dfdxbf
        exec(self.binv[op], e)
        return (res[0].v if op == "div_v" else res[0]), (ax, ay, az, vol, bmvON[0][0])


def gvp3d(nfaces=48, seed=11):
    """one internal face between two cells with prescribed centres; vertex values by inverse distance from the two cells"""
    text = Gvp3dText()
    rng = np.random.default_rng(seed)
    rec = {k: [] for k in ("nv", "pts", "Sf", "Cf", "C", "cell_s", "cell_v", "pt_s", "pt_v", "coef_x", "coef_y", "coef_z", "vol",
                           "grad_s", "grad_v", "cell_t", "div_v", "div_t")}
    for n in range(nfaces):
        nv = 4 if n % 2 == 0 else 3
        pts, own, nei = skew_face(rng, nv)
        cell_s = [float(rng.uniform(0.8, 1.3)), float(rng.uniform(0.8, 1.3))]
        cell_v = [rnd_vec(rng), rnd_vec(rng)]
        pt_s = [inv_dist(x, [own, nei], cell_s) for x in pts]
        pt_v = [inv_dist(x, [own, nei], cell_v) for x in pts]
        gs, (ax, ay, az, vol) = text.grad(pts, own, nei, cell_s, pt_s, False)
        gv, _ = text.grad(pts, own, nei, cell_v, pt_v, True)
        # the two divergences [calcDivfIF L552-560, L633-659]; the tensor field as 9 components, T[3*i + j] as the listing indexes it
        cell_t = [rng.standard_normal(9), rng.standard_normal(9)]
        pt_t = [inv_dist(x, [own, nei], cell_t) for x in pts]
        dv, _ = text.grad(pts, own, nei, cell_v, pt_v, True, op="div_v")
        dt, _ = text.grad(pts, own, nei, cell_t, pt_t, True, op="div_t")
        S, cf = face_area_centre(pts)
        pad3, pad = ([[0, 0, 0]] if nv == 3 else []), ([0.0] if nv == 3 else [])
        rec["nv"].append(nv); rec["pts"].append(np.array([p.c for p in pts] + pad3)); rec["Sf"].append(S); rec["Cf"].append(cf)
        rec["C"].append(np.array([own.c, nei.c])); rec["cell_s"].append(cell_s); rec["cell_v"].append(np.array([v.c for v in cell_v]))
        rec["pt_s"].append(np.array(pt_s + pad)); rec["pt_v"].append(np.array([v.c for v in pt_v] + pad3))
        rec["coef_x"].append(np.array(list(ax[0]) + pad)); rec["coef_y"].append(np.array(list(ay[0]) + pad))
        rec["coef_z"].append(np.array(list(az[0]) + pad)); rec["vol"].append(vol[0])
        rec["grad_s"].append(gs); rec["grad_v"].append(gv)
        rec["cell_t"].append(np.array(cell_t)); rec["div_v"].append(dv); rec["div_t"].append(dt)
    return {k: np.array(v) for k, v in rec.items()}


def gvp3d_bnd(nfaces=40, seed=17):
    """one BOUNDARY face (generic patch) of a cell with prescribed centre; its vertices are boundary points and carry the patch value
    (L0: inverse-distance mean over the adjacent real-patch faces, here one); fvPatchField::snGrad = deltaCoeffs (patch value - cell
    value) with the patch-normal delta of OpenFOAM v2312 (L0)"""
    text = Gvp3dText()
    rng = np.random.default_rng(seed)
    names = ("nv", "pts", "Sf", "Cf", "C", "cell_s", "bnd_s", "cell_v", "bnd_v", "cell_t", "bnd_t", "bmvON", "vol", "coef_x", "coef_y", "coef_z",
             "grad_s", "grad_v", "div_v", "div_t")
    rec = {k: [] for k in names}
    for n in range(nfaces):
        nv = 4 if n % 2 == 0 else 3
        pts, own, _ = skew_face(rng, nv)
        S, cf = face_area_centre(pts)
        Cf = Vec(*cf)
        nf = S / np.sqrt((S * S).sum())
        dc = 1.0 / abs(float(nf @ (cf - own.c)))           # fvPatch::deltaCoeffs, delta = nf (nf & (Cf - Cn))  (L0)
        cs, bs = float(rng.uniform(0.8, 1.3)), float(rng.uniform(0.8, 1.3))
        cv, bv = rnd_vec(rng), rnd_vec(rng)
        ct, bt = rng.standard_normal(9), rng.standard_normal(9)
        gs, (ax, ay, az, vol, bmv) = text.boundary(pts, own, Cf, cs, bs, dc * (bs - cs), [bs] * nv, "grad_s")
        gv, _ = text.boundary(pts, own, Cf, cv, bv, dc * (bv - cv), [bv] * nv, "grad_v")
        dv, _ = text.boundary(pts, own, Cf, cv, bv, dc * (bv - cv), [bv] * nv, "div_v")
        dt, _ = text.boundary(pts, own, Cf, ct, bt, dc * (bt - ct), [bt] * nv, "div_t")
        pad3, pad = ([[0, 0, 0]] if nv == 3 else []), ([0.0] if nv == 3 else [])
        out = dict(nv=nv, pts=np.array([p.c for p in pts] + pad3), Sf=S, Cf=cf, C=own.c, cell_s=cs, bnd_s=bs, cell_v=cv.c, bnd_v=bv.c, cell_t=ct,
                   bnd_t=bt, bmvON=bmv, vol=vol, coef_x=np.array(list(ax) + pad), coef_y=np.array(list(ay) + pad), coef_z=np.array(list(az) + pad),
                   grad_s=gs, grad_v=gv, div_v=dv, div_t=dt)
        for k in names:
            rec[k].append(np.array(out[k], dtype=float))
    return {k: np.array(v) for k, v in rec.items()}


def gvp2d(nfaces=30, seed=12):
    """one internal quad between two cells of a one-cell-thick mesh (empty direction ie3)"""
    f2 = "GaussVolPointBase2D_8C_source.html"
    coef_src = transpile(lines(f2, 154, 168))
    apply_src = transpile(lines(f2, 317, 328))
    rng = np.random.default_rng(seed)
    rec = {k: [] for k in ("ie3", "pts", "Sf", "Cf", "C", "f", "pF", "c", "mv", "grad")}
    for n in range(nfaces):
        ie3 = n % 3
        ie1, ie2 = (1, 2) if ie3 == 0 else ((0, 2) if ie3 == 1 else (0, 1))
        e1, e2 = Vec(*np.eye(3)[ie1]), Vec(*np.eye(3)[ie2])

        def place(a, b, h):
            v = np.zeros(3); v[ie1], v[ie2], v[ie3] = a, b, h
            return Vec(*v)
        C4 = place(*(0.2 * rng.standard_normal(2)), 0.05)                                      # owner centre
        C2 = place(1.0 + 0.2 * rng.standard_normal(), 0.2 * rng.standard_normal(), 0.05)       # neighbour centre
        a0, b0 = 0.5 + 0.2 * rng.standard_normal(), -0.5 + 0.2 * rng.standard_normal()
        a1, b1 = 0.5 + 0.2 * rng.standard_normal(), 0.5 + 0.2 * rng.standard_normal()
        # face vertices in face order: lower edge, then the two upper vertices (the ones the listing picks: first and second
        # vertex with coordinate >= the neighbour centre's along the empty direction [L129-147])
        order = [(a0, b0, 0.0), (a0, b0, 0.1), (a1, b1, 0.1), (a1, b1, 0.0)]
        shift = n % 4
        order = order[shift:] + order[:shift]
        pts = [place(*o) for o in order]
        upper = [k for k, o in enumerate(order) if o[2] >= 0.05]
        ip1, ip3 = upper[0], upper[1]
        mesh = Obj(C=call([C4, C2]), points=call(pts))
        one = lambda: [None]  # noqa: E731
        env = dict(mesh=mesh, iFace=0, ic2=1, ic4=0, ip1=ip1, ip3=ip3, e1_=e1, e2_=e2, mag=mag, v42=one(), v13=one(), mv42_=one(),
                   mv13_=one(), cosa1=one(), cosa2=one(), sina1=one(), sina2=one(), den=one(), c1_=one(), c2_=one(), c3_=one(), c4_=one())
        exec(coef_src, env)
        fvals = [float(rng.standard_normal()), float(rng.standard_normal())]                    # f[ic4], f[ic2]
        pvals = [inv_dist(x, [C4, C2], fvals) for x in pts]
        g = [[0.0, 0.0, 0.0]]
        env.update(f=fvals, pF=pvals, gradf=g, ic2_=[1], ic4_=[0], ip1_=[ip1], ip3_=[ip3], ie1_=ie1, ie2_=ie2, ie3_=ie3, dfdn=0.0, dfdt=0.0)
        exec(apply_src, env)
        S, cf = face_area_centre(pts)
        rec["ie3"].append(ie3); rec["pts"].append(np.array([p.c for p in pts])); rec["Sf"].append(S); rec["Cf"].append(cf)
        rec["C"].append(np.array([C4.c, C2.c])); rec["f"].append(fvals); rec["pF"].append(pvals)
        rec["c"].append([env["c1_"][0], env["c2_"][0], env["c3_"][0], env["c4_"][0]])
        rec["mv"].append([env["mv42_"][0], env["mv13_"][0]]); rec["grad"].append(g[0])
    return {k: np.array(v) for k, v in rec.items()}


def gvp2d_bnd(nfaces=24, seed=18):
    """one BOUNDARY quad (generic patch) of a cell of a one-cell-thick mesh (empty direction ie3); its vertices carry the patch value
    (L0: boundary points average the adjacent real-patch faces, here one); snGrad = deltaCoeffs (patch value - cell value), patch-normal
    delta (L0)"""
    f2 = "GaussVolPointBase2D_8C_source.html"
    v42_src = transpile(lines(f2, 235, 239))
    coef_src = transpile(lines(f2, 244, 288))
    psi_src = transpile(lines(f2, 343, 346))
    apply_src = transpile(lines(f2, 352, 359))
    rng = np.random.default_rng(seed)
    rec = {k: [] for k in ("ie3", "pts", "Sf", "Cf", "C", "f", "fb", "c", "mv", "ip", "grad")}
    for n in range(nfaces):
        ie3 = n % 3
        ie1, ie2 = (1, 2) if ie3 == 0 else ((0, 2) if ie3 == 1 else (0, 1))
        e1, e2 = Vec(*np.eye(3)[ie1]), Vec(*np.eye(3)[ie2])

        def place(a, b, h):
            v = np.zeros(3); v[ie1], v[ie2], v[ie3] = a, b, h
            return Vec(*v)
        C4 = place(*(0.2 * rng.standard_normal(2)), 0.05)
        a0, b0 = 0.5 + 0.2 * rng.standard_normal(), -0.5 + 0.2 * rng.standard_normal()
        a1, b1 = 0.5 + 0.2 * rng.standard_normal(), 0.5 + 0.2 * rng.standard_normal()
        order = [(a0, b0, 0.0), (a0, b0, 0.1), (a1, b1, 0.1), (a1, b1, 0.0)]
        shift = n % 4
        order = order[shift:] + order[:shift]
        pts = [place(*o) for o in order]
        S, cf = face_area_centre(pts)
        if S @ (cf - C4.c) < 0:          # a boundary face points away from its cell
            pts = pts[::-1]
            S, cf = face_area_centre(pts)
        Cf = Vec(*cf)
        nf = S / np.sqrt((S * S).sum())
        dc = 1.0 / abs(float(nf @ (cf - C4.c)))
        fc, fb = float(rng.standard_normal()), float(rng.standard_normal())
        patch = Obj(Cf=call(Fld([Cf])), start=lambda: 0)
        mesh = Obj(C=call([C4]), points=call(pts), boundary=call([patch]), faces=call([list(range(4))]))
        one = lambda: [[None]]  # noqa: E731
        env = dict(mesh=mesh, patchId=0, iFace=0, ie3_=ie3, e1_=e1, e2_=e2, mag=mag, v42=[None], v13=[None], ic4e_=[[0]], ip3e_=one(),
                   ip1e_=one(), mv42e_=one(), mv13e_=one(), cosa1e=one(), cosa2e=one(), sina1e=one(), sina2e=one(), dene=one(), c1e_=one(),
                   c2e_=one(), c3e_=one(), c4e_=one())
        exec(v42_src, env)
        exec(coef_src, env)

        class PatchField(Fld):
            def snGrad(self): return Fld([dc * (fb - fc)])
        g = [[[0.0, 0.0, 0.0]]]
        env.update(f=FieldWithPatches([fc], [PatchField([fb])]), pF=[fb] * 4, psi2=[None], gradf=Obj(boundaryFieldRef=call(g)), ie1_=ie1,
                   ie2_=ie2, dfdn=0.0, dfdt=0.0, mv42e_=[Fld(env["mv42e_"][0])])
        exec(psi_src, env)
        exec(apply_src, env)
        rec["ie3"].append(ie3); rec["pts"].append(np.array([p.c for p in pts])); rec["Sf"].append(S); rec["Cf"].append(cf); rec["C"].append(C4.c)
        rec["f"].append(fc); rec["fb"].append(fb)
        rec["c"].append([env["c1e_"][0][0], env["c2e_"][0][0], env["c3e_"][0][0], env["c4e_"][0][0]])
        rec["mv"].append([env["mv42e_"][0][0], env["mv13e_"][0][0]]); rec["ip"].append([env["ip1e_"][0][0], env["ip3e_"][0][0]])
        rec["grad"].append(g[0][0])
    return {k: np.array(v) for k, v in rec.items()}


def reduced(nfaces=24, seed=19):
    """the four operators of the reduced stencil on one internal face: nf (x) snGrad and nf . snGrad as the listing writes them;
    fvc::snGrad is OpenFOAM's uncorrected (psi_N - psi_O) nonOrthDeltaCoeffs (L0), nf = Sf/|Sf| [fvscStencil.C]"""
    f = "reducedFaceNormalStencil_8C_source.html"
    src = {"grad_s": (transpile(lines(f, 71, 71)), "vF", "tgradIF"), "grad_v": (transpile(lines(f, 85, 85)), "iVF", "tgradIVF"),
           "div_v": (transpile(lines(f, 92, 92)), "iVF", "tdivIVF"), "div_t": (transpile(lines(f, 105, 105)), "iTF", "tdivITF")}
    rng = np.random.default_rng(seed)
    rec = {k: [] for k in ("nv", "pts", "Sf", "Cf", "C", "cell_s", "cell_v", "cell_t", "grad_s", "grad_v", "div_v", "div_t")}
    for n in range(nfaces):
        nv = 4 if n % 2 == 0 else 3
        pts, own, nei = skew_face(rng, nv)
        S, cf = face_area_centre(pts)
        nf = Vec(*(S / np.sqrt((S * S).sum())))
        d = nei - own
        dcoef = 1.0 / max(nf & d, 0.05 * mag(d))            # surfaceInterpolation::nonOrthDeltaCoeffs (L0)
        cs = [float(rng.standard_normal()) for _ in range(2)]
        cv = [rnd_vec(rng), rnd_vec(rng)]
        ct = [Tensor(rng.standard_normal(9)), Tensor(rng.standard_normal(9))]
        vals = {"vF": cs, "iVF": cv, "iTF": ct}
        out = {}
        for op, (code, arg, res) in src.items():
            fld = vals[arg]
            env = dict(nf_=nf, fvc=Obj(snGrad=lambda x: (x[1] - x[0]) * dcoef), **{arg: fld})
            exec(code, env)
            r = env[res]
            out[op] = r.c if isinstance(r, Vec) else (r.m.reshape(9) if isinstance(r, Tensor) else r)
        rec["nv"].append(nv); rec["pts"].append(np.array([p.c for p in pts] + ([[0, 0, 0]] if nv == 3 else []))); rec["Sf"].append(S)
        rec["Cf"].append(cf); rec["C"].append(np.array([own.c, nei.c])); rec["cell_s"].append(cs); rec["cell_v"].append(np.array([v.c for v in cv]))
        rec["cell_t"].append(np.array([t.m.reshape(9) for t in ct]))
        for op in src:
            rec[op].append(np.array(out[op], dtype=float))
    return {k: np.array(v) for k, v in rec.items()}


def lsq_bnd(nfaces=16, seed=20):
    """one boundary quad of a cell of a one-cell-thick mesh under the leastSquares gradient: ordinary patch -> nf * snGrad, symmetryPlane
    patch -> the initial zero [ScalarGrad.C L55, L86-109]"""
    src = transpile(lines("extendedFaceStencilScalarGrad_8C_source.html", 86, 109))
    rng = np.random.default_rng(seed)
    rec = {k: [] for k in ("ie3", "pts", "Sf", "Cf", "C", "f", "fb", "symmetry", "grad")}
    for n in range(nfaces):
        ie3 = n % 3
        ie1, ie2 = (1, 2) if ie3 == 0 else ((0, 2) if ie3 == 1 else (0, 1))

        def place(a, b, h):
            v = np.zeros(3); v[ie1], v[ie2], v[ie3] = a, b, h
            return Vec(*v)
        C4 = place(*(0.2 * rng.standard_normal(2)), 0.05)
        a0, b0 = 0.5 + 0.2 * rng.standard_normal(), -0.5 + 0.2 * rng.standard_normal()
        a1, b1 = 0.5 + 0.2 * rng.standard_normal(), 0.5 + 0.2 * rng.standard_normal()
        pts = [place(a0, b0, 0.0), place(a0, b0, 0.1), place(a1, b1, 0.1), place(a1, b1, 0.0)]
        S, cf = face_area_centre(pts)
        if S @ (cf - C4.c) < 0:
            pts = pts[::-1]
            S, cf = face_area_centre(pts)
        nf = S / np.sqrt((S * S).sum())
        dc = 1.0 / abs(float(nf @ (cf - C4.c)))
        fc, fb = float(rng.standard_normal()), float(rng.standard_normal())
        symmetry = n % 4 == 3
        kinds = {"symmetryPlaneFvPatch"} if symmetry else set()

        class PatchField(Fld):
            def snGrad(self): return Fld([dc * (fb - fc)])
        out = [Fld([Vec(0, 0, 0)])]          # gradIF starts from zero [L55]
        env = dict(mesh_=Obj(boundaryMesh=call([0]), boundary=call([kinds])), isA=lambda name, patch: name in patch, true=True, false=False,
                   gradIF=Obj(boundaryFieldRef=call(out)), nf_=Obj(boundaryField=call([Fld([Vec(*nf)])])),
                   iF=Obj(boundaryField=call([PatchField([fb])])))
        exec(src, env)
        rec["ie3"].append(ie3); rec["pts"].append(np.array([p.c for p in pts])); rec["Sf"].append(S); rec["Cf"].append(cf); rec["C"].append(C4.c)
        rec["f"].append(fc); rec["fb"].append(fb); rec["symmetry"].append(int(symmetry)); rec["grad"].append(out[0][0].c)
    return {k: np.array(v) for k, v in rec.items()}


def lsq(nfaces=30, seed=13):
    """one internal face with a stencil of n cells (cells 0 and 1 are its owner and neighbour, placed symmetrically about the
    face centre so that the linear weight is 1/2 and sF = (iF[0] + iF[1])/2 needs no further L0 rule)"""
    w_src = transpile(lines("extendedFaceStencilCalculateWeights_8C_source.html", 64, 153))
    a_src = transpile(lines("extendedFaceStencilScalarGrad_8C_source.html", 66, 72))
    rng = np.random.default_rng(seed)
    rec = {k: [] for k in ("n", "Cf", "centres", "deg", "Gdf", "wf2", "iF", "sF", "grad")}
    nmax = 8
    for n in range(nfaces):
        ns = [6, 6, 4, 8, 5, 2][n % 6]
        scale = [1.0, 0.3, 3.0][n % 3]   # det(G) < 1 marks a face degenerate: it depends on the length unit (quirk B3)
        Cf = Vec(*(scale * np.array([0.5, 0.5, 0.05])))
        d0 = scale * np.array([-0.5 + 0.1 * rng.standard_normal(), 0.1 * rng.standard_normal(), 0.0])
        cents = [Vec(*(Cf.c + d0)), Vec(*(Cf.c - d0))]
        if ns == 2:                       # a 1-D stencil: the two cells on the x-axis
            cents = [Vec(*(Cf.c + scale * np.array([-0.5, 0, 0]))), Vec(*(Cf.c + scale * np.array([0.5, 0, 0])))]
        for _ in range(ns - 2):           # 2-D meshes: every centre in the plane of the face centre (G.zz == 0)
            cents.append(Vec(*(Cf.c + scale * np.array([rng.uniform(-1.5, 1.5), rng.uniform(-1.0, 1.0), 0.0]))))
        cMesh = Obj(faceCentres=call([Cf]), cellCentres=call(cents), nGeometricD=call(1 if ns == 2 else 2))

        class Sink:
            def __lshift__(self, o): return self
        deg = RList()
        env = dict(neighbourCells_=[RList(range(len(cents)))], facei=0, cMesh_=cMesh, symmTensor=symmTensor, RList=RList, sqr=sqr,
                   magSqr=magSqr, mag=mag, det=det, inv=inv, SMALL=1e-15, GREAT=1e15, cellDim=3, minDet=1e15, detG=0.0,
                   nDegFaces=0, internalDegFaces_=deg, GdfAll_=[None], wf2All_=[None], WarningInFunction=Sink(), Pout=Sink(),
                   nl="\n", endl="\n")
        exec(w_src, env)
        Gdf, wf2 = env["GdfAll_"][0], env["wf2All_"][0]
        iF = [float(rng.standard_normal()) for _ in cents]
        sF = [0.5 * (iF[0] - iF[1]) + iF[1]]
        gradIF = [None]
        e2 = dict(GdfAll_=[Gdf], wf2All_=[wf2], neighbourCells_=[RList(range(len(cents)))], iF=iF, sF=sF, gradIF=gradIF, facei=0,
                  vector=Obj(zero=Vec(0, 0, 0)), gf=None)
        exec(a_src, e2)
        pad = nmax - len(cents)
        rec["n"].append(len(cents)); rec["Cf"].append(Cf.c)
        rec["centres"].append(np.array([c.c for c in cents] + [[0, 0, 0]] * pad))
        rec["deg"].append(len(deg)); rec["Gdf"].append(np.array([g.c for g in Gdf] + [[0, 0, 0]] * pad))
        rec["wf2"].append(np.array(list(wf2) + [0.0] * pad)); rec["iF"].append(np.array(iF + [0.0] * pad)); rec["sF"].append(sF[0])
        rec["grad"].append(gradIF[0].c)
    return {k: np.array(v) for k, v in rec.items()}


def dev2(t):
    """OpenFOAM's dev2(T) = T - (2/3) tr(T) I (L0)"""
    return Tensor(t.m - (2.0 / 3.0) * np.trace(t.m) * np.eye(3))


class MulPair:
    """a pair of scalars that scales a pair of tensors from the left: muEff * dev2(...)"""
    def __init__(self, o, n): self.o, self.n = o, n
    def __mul__(self, t): return type(t)(self.o * t.o, self.n * t.n)


class Pair:
    """a cell field seen from one face: (owner value, neighbour value)"""
    def __init__(self, o, n): self.o, self.n = o, n
    def __mul__(self, other): return Pair(self.o * other.o, self.n * other.n)
    def __add__(self, other): return Pair(self.o + other.o, self.n + other.n)
    def __truediv__(self, other): return Pair(self.o / other.o, self.n / other.n)


class CF(list):
    """a cell field with its old-time values: elementwise arithmetic; assignment keeps the object (OpenFOAM assigns values)"""
    def __init__(self, vals, old=None):
        super().__init__(vals)
        self.old = list(old) if old is not None else list(vals)
    def _z(self, o, f): return CF([f(a, b) for a, b in zip(self, o)]) if isinstance(o, list) else CF([f(a, o) for a in self])
    def __add__(self, o): return self._z(o, lambda a, b: a + b)
    def __sub__(self, o): return self._z(o, lambda a, b: a - b)
    def __mul__(self, o): return self._z(o, lambda a, b: a * b)
    def __rmul__(self, o): return CF([o * a for a in self])
    def __truediv__(self, o): return self._z(o, lambda a, b: a / b)
    def __rsub__(self, o): return CF([o - a for a in self])
    def __iadd__(self, o):                       # Yt += Yi: elementwise (a plain list would be extended)
        self[:] = list(self + o)
        return self
    def __call__(self): return self
    def oldTime(self): return CF(self.old)
    def max(self, v): self[:] = [max(a, v) for a in self]
    def assign(self, o): self[:] = list(o)
    def correctBoundaryConditions(self): pass
    def boundaryFieldRef(self): return Nil()
    def boundaryField(self): return Nil()


class Nil:
    """the (absent) boundary of the two-cell mesh: absorbs whatever the listing does to patch fields"""
    def __eq__(self, o): return None
    def __mul__(self, o): return self
    __rmul__ = __add__ = __radd__ = __mul__
    __hash__ = None


class Mat:
    """a diagonal fvMatrix: a psi = b per cell (Euler ddt terms only -- the explicit branch has no other implicit operator)"""
    def __init__(self, psi, a, b): self.psi, self.a, self.b = psi, list(a), list(b)
    def __add__(self, f): return Mat(self.psi, self.a, [b - x for b, x in zip(self.b, f)])      # M + F = 0  ->  a psi = b - F
    def __sub__(self, f): return Mat(self.psi, self.a, [b + x for b, x in zip(self.b, f)])
    def __eq__(self, s): return Mat(self.psi, self.a, [b + x for b, x in zip(self.b, s)])       # M == S   ->  a psi = b + S
    __hash__ = None
    def solve(self): self.psi.assign([b / a for a, b in zip(self.a, self.b)])


def fv_emulation(dt, V):
    """fvm::ddt / fvc::ddt (Euler) and fvc::div (= surfaceIntegrate of a face flux) on the two-cell mesh: face 0 points from cell 0
    to cell 1 (L0 semantics; the equations themselves come from the listing text)"""
    fvm = Obj(ddt=lambda *a: (Mat(a[0], [1.0 / dt] * 2, [o / dt for o in a[0].old]) if len(a) == 1 else
                              Mat(a[1], [r / dt for r in a[0]], [ro * o / dt for ro, o in zip(a[0].old, a[1].old)])))
    fvc = Obj(ddt=lambda *a: (CF([(x - o) / dt for x, o in zip(a[0], a[0].old)]) if len(a) == 1 else
                              CF([(r * x - ro * o) / dt for r, x, ro, o in zip(a[0], a[1], a[0].old, a[1].old)])),
              div=lambda phi: CF([phi / V[0], -phi / V[1]]))
    return fvm, fvc


class Mat2:
    """L(psi) = b on the two-cell mesh with L(psi)_i = d_i psi_i - a psi_nb: what fvm::ddt(rho, psi) (a = 0) and fvm::laplacian(gamma,
    psi) (d = a = -gamma |Sf| delta, unit volumes) assemble; explicit fields move to the right-hand side (L0 fvMatrix algebra)"""
    def __init__(self, psi, d, a, b): self.psi, self.d, self.a, self.b = psi, list(d), a, list(b)
    def __sub__(self, o):
        if isinstance(o, Mat2):
            return Mat2(self.psi, [x - y for x, y in zip(self.d, o.d)], self.a - o.a, [x - y for x, y in zip(self.b, o.b)])
        return Mat2(self.psi, self.d, self.a, [b + x for b, x in zip(self.b, o)])       # L - F = 0  ->  L = b + F
    def __add__(self, o): return Mat2(self.psi, self.d, self.a, [b - x for b, x in zip(self.b, o)])
    def __eq__(self, s): return Mat2(self.psi, self.d, self.a, [b + x for b, x in zip(self.b, s)])
    def solve(self):
        d0, d1, a = self.d[0], self.d[1], self.a
        det = d0 * d1 - a * a
        b0, b1 = self.b
        self.psi.assign([(b0 * d1 + b1 * a) * (1.0 / det), (b1 * d0 + b0 * a) * (1.0 / det)])
    def flux(self):              # fvMatrix::flux on the internal face (L0): upper psi_N - lower psi_O, upper = lower = -a of L
        return -self.a * self.psi[1] + self.a * self.psi[0]


def fv_emulation_implicit(dt, magS, delta):
    """the same verbs for the implicitDiffusion branch [QGDUEqn.H L56-68, QGDEEqn.H L55-61]: fvm::ddt(rho, psi), fvc::ddt(rho, psi),
    fvm::laplacian(gamma, psi) with the uncorrected surface-normal gradient, fvc::div (unit volumes)"""
    fvm = Obj(ddt=lambda rho, psi: Mat2(psi, [r / dt for r in rho], 0.0, [ro * o / dt for ro, o in zip(rho.old, psi.old)]),
              laplacian=lambda gam, psi: Mat2(psi, [-(gam * magS * delta)] * 2, -(gam * magS * delta), [psi[0] * 0.0, psi[1] * 0.0]))
    fvc = Obj(ddt=lambda rho, psi: CF([(r * x - ro * o) / dt for r, x, ro, o in zip(rho, psi, rho.old, psi.old)]),
              div=lambda phi: CF([phi / 1.0, -phi / 1.0]))
    return fvm, fvc


def field_assignments(src, names):
    """`e = expr;` on a field assigns values into the existing object (its old-time level stays): rewrite those statements"""
    return re.sub(r"^(\s*)(%s)\s*=(?!=)\s*(.*)$" % "|".join(names), r"\1\2.assign(\3)", src, flags=re.M)


def case2cell(nfaces=40, seed=14):
    """A whole QGDFoam flux assembly on one internal face between two cells (3-D, GaussVolPoint): updateFields.H,
    the four fvsc::grad (coefficients + dfdxif from the 3-D listing), updateFluxes.H, constScPrModel1, hQGDf -- every line from
    the listing text.  Inputs the case entries take: mesh arrays + geometry, U, T, p, the gas model."""
    text = Gvp3dText()
    fields_src = transpile(lines("QGDFoam_2updateFields_8H_source.html", 45, 80))
    flux_src = transpile(lines("QGDFoam_2updateFluxes_8H_source.html", 41, 139))
    cs = listing("constScPrModel1_8C_source.html")
    tau_src = transpile([cs[103].replace("this->", ""), cs[104].replace("this->", "")])
    mu_src = transpile([cs[i] for i in range(108, 115)])
    h_src = transpile(lines("QGDCoeffs_8C_source.html", 305, 307))
    # the three equations of the explicit step [QGDRhoEqn.H L40-47, QGDUEqn.H L36-89, QGDEEqn.H L37-76]
    eq_src = [transpile(lines("QGDRhoEqn_8H_source.html", 40, 47)),
              field_assignments(transpile(lines("QGDUEqn_8H_source.html", 36, 89)), ("rhoU", "U")),
              field_assignments(transpile(lines("QGDEEqn_8H_source.html", 37, 76)), ("rhoE", "e"))]
    dt = 1e-3
    rng = np.random.default_rng(seed)
    names = ("nv", "pts", "Sf", "Cf", "C", "V", "U", "T", "p", "R", "Cv", "mu", "Pr", "ScQGD", "PrQGD", "alphaQGD", "deltaT", "rho1", "U1", "e1",
             "rhoU1", "rhoE1", "phiPi_impl", "phiQ_impl", "phiTauMC", "tauMC",
             "w", "hQGDf", "rhof", "Uf", "rhoUf", "UrhoUf", "pf", "cf", "gammaf", "Hf", "alphauf", "muf", "tauQGDf",
             "gradUf", "gradef", "gradRhof", "gradPf", "phiwStar", "phiJm", "phi", "phiJmU", "phiP", "phiPi", "phiJmH", "phiQ", "phiPiU",
             "muQGD", "alphauQGD", "tauQGD", "hQGD")
    rec = {k: [] for k in names}
    rec2 = case2cell.implicit_step = {}
    for n in range(nfaces):
        nv = 4 if n % 3 != 2 else 3
        pts, own, nei = skew_face(rng, nv)
        S, cf = face_area_centre(pts)
        S, Cf = Vec(*S), Vec(*cf)
        R = 1 / 1.4
        Cv = R / 0.4
        gam = (Cv + R) / Cv
        mu0, Pr = float(rng.uniform(0, 2e-3)), float(rng.uniform(0.6, 1.2))
        Sc, PrQ, aQ = float(rng.uniform(0.5, 1.5)), float(rng.uniform(0.5, 1.5)), float(rng.uniform(0.3, 0.7))
        U = [rnd_vec(rng, 0.7), rnd_vec(rng, 0.7)]
        T = [float(rng.uniform(0.8, 1.3)) for _ in range(2)]
        p = [float(rng.uniform(0.7, 1.4)) for _ in range(2)]
        # L0 thermo (perfectGas + eConst + constTransport): what createFields.H / thermo.correct() hand to the face code
        e = [Cv * t for t in T]
        psi = [1.0 / (R * t) for t in T]
        rho = [ps * pp for ps, pp in zip(psi, p)]
        c = [float(np.sqrt(gam / ps)) for ps in psi]
        Cp = Cv + R
        alphah0 = (Cp * mu0 * (1.0 / Pr)) / Cp
        # linear weight (L0: surfaceInterpolation::makeWeights)
        sfo, sfn = abs(S & (Cf - own)), abs(S & (nei - Cf))
        w = sfn / (sfo + sfn)
        lin = lambda q: w * (q.o - q.n) + q.n  # noqa: E731   surfaceInterpolationScheme::interpolate (L0)
        # hQGDf [QGDCoeffs.C L305-307]; hQGD of a cell with this one face [L336-361]
        hq = [0.0]
        magS = mag(S)
        ev = dict(mag=mag, min=min, mesh=Obj(C=call([own, nei]), Cf=call([Cf]), owner=call([0]), neighbour=call([1])), iFace=0,
                  hQGDf_=Obj(primitiveFieldRef=call(hq)), hown=0.0, hnei=0.0)
        exec(h_src, ev)
        hf = hq[0]
        hcell = (hf * magS) / magS
        muQGD, alphauQGD = [], []
        for k in range(2):
            m_, a_ = [0.0], [0.0]
            ev = dict(p=Obj(primitiveField=call([p[k]])), ScQGD_=Obj(primitiveField=call([Sc])), PrQGD_=Obj(primitiveField=call([PrQ])),
                      tauQGD_=Obj(primitiveField=call([aQ * hcell / c[k]])), celli=0,
                      muQGD_=Obj(primitiveFieldRef=call(m_), primitiveField=call(m_)), alphauQGD_=Obj(primitiveFieldRef=call(a_)))
            exec(mu_src, ev)
            muQGD.append(m_[0]); alphauQGD.append(a_[0])
        ev = dict(aQGD_=Pair(aQ, aQ), cSound=Pair(*c), hQGDf_=hf, hQGD_=Pair(hcell, hcell), linearInterpolate=lin)
        exec(tau_src, ev)
        tauf, tauc = ev["tauQGDf_"], ev["tauQGD_"]
        # QGDThermo::correctQGD adds muQGD / alphauQGD into mu / alpha [QGDThermo.C L91-98]; laminar muEff = mu,
        # alphaEff = gamma*alpha for an internal-energy thermo (L0)
        muEff = Pair(0.0 + (mu0 + muQGD[0]), 0.0 + (mu0 + muQGD[1]))
        alphaEff = Pair(gam * ((alphah0 + alphauQGD[0]) + 0.0), gam * ((alphah0 + alphauQGD[1]) + 0.0))
        rhoU = Pair(rho[0] * U[0], rho[1] * U[1])
        rhoE = Pair(rho[0] * e[0] + rho[0] * 0.5 * (U[0] & U[0]), rho[1] * e[1] + rho[1] * 0.5 * (U[1] & U[1]))
        env = dict(qgdInterpolate=lin, rho=Pair(*rho), U=Pair(*U), rhoU=rhoU, p=Pair(*p), gamma=Pair(gam, gam), rhoE=rhoE,
                   thermo=Obj(c=call(Pair(*c)), Cp=call(Pair(Cp, Cp))), turbulence=Obj(alphaEff=call(alphaEff), muEff=call(muEff)))
        exec(fields_src, env)
        # the four fvsc::grad: vertex values by inverse distance, then coefficients + dfdxif as listed
        cen = [own, nei]
        gU, _ = text.grad(pts, own, nei, U, [inv_dist(x, cen, U) for x in pts], True)
        gE, _ = text.grad(pts, own, nei, e, [inv_dist(x, cen, e) for x in pts], False)
        gR, _ = text.grad(pts, own, nei, rho, [inv_dist(x, cen, rho) for x in pts], False)
        gP, _ = text.grad(pts, own, nei, p, [inv_dist(x, cen, p) for x in pts], False)
        grads = dict(U=Tensor(gU), e=Vec(*gE), rho=Vec(*gR), p=Vec(*gP))
        env2 = {k: env[k] for k in ("rhof", "Uf", "rhoUf", "UrhoUf", "pf", "gammaf", "Hf", "alphauf", "muf")}
        env2.update(tr=tr, tauQGDf=tauf, mesh=Obj(Sf=call(S)), fvsc=Obj(grad=lambda fld: grads[fld]), U="U", e="e", rho="rho", p="p",
                    I=Sph(1.0), implicitDiffusion=False, Foam=Obj(T=lambda t: t.T()), qgdFlux=lambda flux, psi, psif: flux * psif, H="H")
        exec(flux_src, env2)
        g = env2
        # the same listing with implicitDiffusion true: Pif / qf without the Navier-Stokes / Fourier parts and
        # tauMC = qgdInterpolate(muEff * dev2(T(fvc::grad(U)))), phiTauMC = Sf & tauMC [updateFluxes.H L95-111, L131-135].  fvc::grad(U) of
        # the two one-face cells is Gauss linear: +Sf (x) Uf / V in the owner, -Sf (x) Uf / V in the neighbour (L0)
        Uf_lin = lin(Pair(*U))
        gradUc = Pair(Tensor(np.outer(S.c, Uf_lin.c) / 1.0), Tensor(-np.outer(S.c, Uf_lin.c) / 1.0))

        class Holder:
            v = None
            def __call__(self): return self if self.v is None else self.v
            def assign(self, x): self.v = x
        tauMC = Holder()

        class PT(Pair):   # pairs of tensors scaled by pairs of scalars
            def __rmul__(self, sc): return PT(sc.o * self.o, sc.n * self.n) if isinstance(sc, Pair) else PT(sc * self.o, sc * self.n)
        env3 = {k: env[k] for k in ("rhof", "Uf", "rhoUf", "UrhoUf", "pf", "gammaf", "Hf", "alphauf", "muf")}
        env3.update(tr=tr, tauQGDf=tauf, mesh=Obj(Sf=call(S)), fvsc=Obj(grad=lambda fld: grads[fld]), U="U", e="e", rho="rho", p="p",
                    I=Sph(1.0), implicitDiffusion=True, Foam=Obj(T=lambda t: PT(t.o.T(), t.n.T()) if isinstance(t, Pair) else t.T()),
                    qgdFlux=lambda flux, psi, psif: flux * psif, H="H", tauMCPtr=tauMC, qgdInterpolate=lin,
                    turbulence=Obj(muEff=call(MulPair(muEff.o, muEff.n))), fvc=Obj(grad=lambda fld: PT(gradUc.o, gradUc.n)),
                    dev2=lambda t: PT(dev2(t.o), dev2(t.n)))
        exec(flux_src, env3)
        gi = env3

        # one explicit step of the three equations with those fluxes (zero sources)
        fvm, fvc = fv_emulation(dt, [1.0, 1.0])
        zero_s, zero_v = CF([0.0, 0.0]), CF([Vec(0, 0, 0), Vec(0, 0, 0)])
        eq = dict(fvm=fvm, fvc=fvc, solve=lambda M: M.solve(), implicitDiffusion=False, magSqr=magSqr,
                  rho=CF(rho), U=CF(U), e=CF(e), rhoU=CF([rhoU.o, rhoU.n]), rhoE=CF([rhoE.o, rhoE.n]),
                  rhoSu=zero_s, rhoUSu=zero_v, rhoESu=zero_s, phiSigmaDotU=0.0,
                  **{k: g[k] for k in ("phiJm", "phiJmU", "phiP", "phiPi", "phiJmH", "phiQ", "phiPiU")})
        for code in eq_src:
            exec(code, eq)

        # the same step with implicitDiffusion true [QGDUEqn.H L36-75, QGDEEqn.H L37-64]: explicit part with the fluxes of the implicit
        # branch (gi), then the two implicit solves (2 x 2 systems here), phiSigmaDotU from fvc::grad of the NEW velocity
        dvec = nei - own
        delta = 1.0 / max((S / mag(S)) & dvec, 0.05 * mag(dvec))                       # nonOrthDeltaCoeffs (L0)
        fvm_e, fvc_e = fv_emulation(dt, [1.0, 1.0])
        fvm_i, fvc_i = fv_emulation_implicit(dt, mag(S), delta)

        class Both:      # fvm::ddt / fvc::ddt with one argument belong to the explicit part, with two to the implicit part
            def __init__(self, a, b): self.a, self.b = a, b
            def ddt(self, *x): return (self.a if len(x) == 1 else self.b).ddt(*x)
            def __getattr__(self, k): return getattr(self.b, k)
        sig = Holder()
        Ui = CF(U)

        def grad_lin(fld):   # linearInterpolate(fvc::grad(U)): Gauss linear gradient of the two one-face cells, then the face value
            uf = lin(Pair(fld[0], fld[1]))
            gc = Pair(Tensor(np.outer(S.c, uf.c)), Tensor(-np.outer(S.c, uf.c)))
            return lin(gc)
        eqi = dict(fvm=Both(fvm_e, fvm_i), fvc=Both(fvc_e, Obj(ddt=fvc_i.ddt, div=fvc_i.div, grad=lambda fld: fld)), solve=lambda M: M.solve(),
                   implicitDiffusion=True, magSqr=magSqr, rho=CF(rho), U=Ui, e=CF(e), rhoU=CF([rhoU.o, rhoU.n]), rhoE=CF([rhoE.o, rhoE.n]),
                   rhoSu=zero_s, rhoUSu=zero_v, rhoESu=zero_s, phiSigmaDotU=0.0, muf=gi["muf"], alphauf=gi["alphauf"], Uf=gi["Uf"],
                   phiTauMC=gi["phiTauMC"], tauMCPtr=tauMC, sigmaDotUPtr=sig, linearInterpolate=grad_lin, mesh=Obj(Sf=call(S)),
                   **{k: gi[k] for k in ("phiJm", "phiJmU", "phiP", "phiPi", "phiJmH", "phiQ", "phiPiU")})
        eq_impl = [eq_src[0], field_assignments(eq_src[1], ("rhoU",)), eq_src[2]]
        for code in eq_impl:
            exec(code, eqi)
        extra = dict(rho1=list(eqi["rho"]), U1=np.array([u.c for u in eqi["U"]]), e1=list(eqi["e"]), rhoE1=list(eqi["rhoE"]),
                     phiSigmaDotU=eqi["phiSigmaDotU"], delta=delta)
        for k, v_ in extra.items():
            rec2.setdefault(k, []).append(np.array(v_, dtype=float))

        def val(x):
            return x.c if isinstance(x, Vec) else (x.m.reshape(9) if isinstance(x, Tensor) else x)
        out = dict(nv=nv, pts=np.array([q.c for q in pts] + ([[0, 0, 0]] if nv == 3 else [])), Sf=S.c, Cf=Cf.c, C=np.array([own.c, nei.c]),
                   V=[1.0, 1.0], U=np.array([u.c for u in U]), T=T, p=p, R=R, Cv=Cv, mu=mu0, Pr=Pr, ScQGD=Sc, PrQGD=PrQ, alphaQGD=aQ, deltaT=dt,
                   rho1=list(eq["rho"]), U1=np.array([u.c for u in eq["U"]]), e1=list(eq["e"]), rhoU1=np.array([u.c for u in eq["rhoU"]]),
                   rhoE1=list(eq["rhoE"]), phiPi_impl=val(gi["phiPi"]), phiQ_impl=val(gi["phiQ"]), phiTauMC=val(gi["phiTauMC"]),
                   tauMC=val(tauMC()),
                   w=w, hQGDf=hf, cf=env["cf"], tauQGDf=tauf, phiwStar=g["phiw"], muQGD=muQGD, alphauQGD=alphauQGD,
                   tauQGD=[tauc.o, tauc.n], hQGD=[hcell, hcell])
        for k in ("rhof", "Uf", "rhoUf", "UrhoUf", "pf", "gammaf", "Hf", "alphauf", "muf", "gradUf", "gradef", "gradRhof", "gradPf",
                  "phiJm", "phi", "phiJmU", "phiP", "phiPi", "phiJmH", "phiQ", "phiPiU"):
            out[k] = val(g[k])
        for k in names:
            rec[k].append(np.array(out[k], dtype=float))
    return {k: np.array(v) for k, v in rec.items()}


def qhdface(nfaces=36, seed=15):
    """The QHDFoam face expressions on one internal face between two cells (3-D, GaussVolPoint): updateFields.H (interpolations, body
    force), updateFluxes.H (phiu, phiwo, taubyrhof), the flux parts of QHDUEqn.H (Wf, phiUfWf, phiUf) and QHDTEqn.H (phiTf,
    phiTauTReg) -- every line from the listing text, the gradients through the 3-D GaussVolPoint text.  tauQGDf, rho, phi are inputs
    (what thermo.tauQGDf(), the rhoConst thermo and the pressure equation hand over)."""
    text = Gvp3dText()
    fields_src = transpile(lines("QHDFoam_2updateFields_8H_source.html", 36, 73))
    flux_src = transpile(lines("QHDFoam_2updateFluxes_8H_source.html", 33, 38))
    ueqn_src = transpile(lines("QHDUEqn_8H_source.html", 36, 43))
    teqn_src = transpile(lines("QHDTEqn_8H_source.html", 65, 66))
    rng = np.random.default_rng(seed)
    names = ("nv", "pts", "Sf", "Cf", "C", "U", "T", "p", "rho", "tauQGDf", "phi", "beta", "g", "w",
             "gradUf", "gradTf", "gradPf", "phiu", "phiwo", "taubyrhof", "Wf", "phiUf", "phiTf", "phiTauTReg")
    rec = {k: [] for k in names}
    for n in range(nfaces):
        nv = 4 if n % 3 != 2 else 3
        pts, own, nei = skew_face(rng, nv)
        S, cf = face_area_centre(pts)
        S, Cf = Vec(*S), Vec(*cf)
        U = [rnd_vec(rng, 0.3), rnd_vec(rng, 0.3)]
        T = [float(rng.uniform(290.0, 310.0)) for _ in range(2)]
        p = [float(rng.uniform(-0.1, 0.1)) for _ in range(2)]
        rho = [float(rng.uniform(0.9, 1.2)) for _ in range(2)]
        tauf = float(rng.uniform(1e-4, 1e-2))
        phi = float(rng.standard_normal() * 1e-2)
        beta = float(rng.uniform(1e-3, 5e-3))
        g = rnd_vec(rng, 9.81)
        sfo, sfn = abs(S & (Cf - own)), abs(S & (nei - Cf))
        w = sfn / (sfo + sfn)
        lin = lambda q: w * (q.o - q.n) + q.n  # noqa: E731   surfaceInterpolationScheme::interpolate (L0)
        cen = [own, nei]
        gU, _ = text.grad(pts, own, nei, U, [inv_dist(x, cen, U) for x in pts], True)
        gT, _ = text.grad(pts, own, nei, T, [inv_dist(x, cen, T) for x in pts], False)
        gP, _ = text.grad(pts, own, nei, p, [inv_dist(x, cen, p) for x in pts], False)
        grads = dict(U=Tensor(gU), T=Vec(*gT), p=Vec(*gP), W=Tensor(np.zeros(9)))
        one = Pair(1.0, 1.0)

        class PS(Pair):   # a pair of scalars that can be scaled and multiplied by a vector (BdFrc = beta*T*g)
            def __rmul__(self, s): return PS(s * self.o, s * self.n)
            def __mul__(self, v): return Pair(self.o * v, self.n * v) if isinstance(v, Vec) else Pair.__mul__(self, v)
        fU, fT, fW = Pair(*U), PS(*T), Pair(Vec(0, 0, 0), Vec(0, 0, 0))
        # the listing passes fields by name: fvsc.grad looks the gradient up by the field's identity, qgdInterpolate takes its
        # (owner, neighbour) pair
        env = dict(qgdInterpolate=lin, fvsc=Obj(grad=lambda fld: grads["U" if fld is fU else ("T" if fld is fT else "W")]),
                   U=fU, T=fT, W=fW, rho=Pair(*rho), beta=beta, g=g,
                   turbulence=Obj(muEff=call(one)), thermo=Obj(alpha=call(one), Cp=call(one)))
        exec(fields_src, env)
        env2 = dict(mesh=Obj(Sf=call(S)), Uf=env["Uf"], gradUf=env["gradUf"], BdFrcf=env["BdFrcf"], tauQGDf=tauf, rhof=env["rhof"])
        src2 = "\n".join(l for l in flux_src.split("\n") if "setOriented" not in l)
        exec(src2, env2)
        env3 = dict(env2, fvsc=Obj(grad=lambda fld: grads["p"]), p="p", phi=phi, U="U", qgdFlux=lambda flux, psi, psif: flux * psif)
        src3 = "\n".join(l for l in ueqn_src.split("\n") if "setOriented" not in l)
        exec(src3, env3)
        env4 = dict(env3, T="T", Tf=env["Tf"], gradTf=env["gradTf"])
        exec(teqn_src, env4)

        def val(x):
            return x.c if isinstance(x, Vec) else (x.m.reshape(9) if isinstance(x, Tensor) else x)
        out = dict(nv=nv, pts=np.array([q.c for q in pts] + ([[0, 0, 0]] if nv == 3 else [])), Sf=S.c, Cf=Cf.c, C=np.array([own.c, nei.c]),
                   U=np.array([u.c for u in U]), T=T, p=p, rho=rho, tauQGDf=tauf, phi=phi, beta=beta, g=g.c, w=w,
                   gradUf=val(env["gradUf"]), gradTf=val(env["gradTf"]), gradPf=val(env3["gradPf"]), phiu=env2["phiu"], phiwo=env2["phiwo"],
                   taubyrhof=env2["taubyrhof"], Wf=val(env3["Wf"]), phiUf=val(env3["phiUf"]), phiTf=env4["phiTf"], phiTauTReg=env4["phiTauTReg"])
        for k in names:
            rec[k].append(np.array(out[k], dtype=float))
    return {k: np.array(v) for k, v in rec.items()}


def species(nfaces=30, seed=16):
    """One species of reactingLagrangianQGDFoam/updateFluxes.H L117-132 on one internal face: gradYf, phiJmY = qgdFlux(phiJm, Y, Yf) +
    dydtflux, diffusiveFlux = dydtflux = -phi tauQGDf (Uf & gradYf); Yf from updateFields.H L38; Uf = qgdInterpolate(U)."""
    text = Gvp3dText()
    yf_src = transpile(lines("reactingLagrangianQGDFoam_2updateFields_8H_source.html", 38, 38))
    src = transpile(lines("reactingLagrangianQGDFoam_2updateFluxes_8H_source.html", 122, 127))
    rng = np.random.default_rng(seed)
    names = ("nv", "pts", "Sf", "Cf", "C", "U", "Y", "phiJm", "phi", "tauQGDf", "gradYf", "phiJmY", "diffusiveFlux")
    rec = {k: [] for k in names}
    for n in range(nfaces):
        nv = 4 if n % 3 != 2 else 3
        pts, own, nei = skew_face(rng, nv)
        S, cf = face_area_centre(pts)
        S, Cf = Vec(*S), Vec(*cf)
        U = [rnd_vec(rng, 0.7), rnd_vec(rng, 0.7)]
        Y = [float(rng.uniform(0.0, 1.0)) for _ in range(2)]
        phiJm, phi, tauf = float(rng.standard_normal()), float(rng.standard_normal()), float(rng.uniform(1e-4, 1e-2))
        sfo, sfn = abs(S & (Cf - own)), abs(S & (nei - Cf))
        w = sfn / (sfo + sfn)
        lin = lambda q: w * (q.o - q.n) + q.n  # noqa: E731
        gY, _ = text.grad(pts, own, nei, Y, [inv_dist(x, [own, nei], Y) for x in pts], False)
        fY = Pair(*Y)
        env = dict(qgdInterpolate=lin, Y=[fY], Yf=[None], i=0)
        exec(yf_src, env)
        env2 = dict(fvsc=Obj(grad=lambda fld: Vec(*gY)), Y=[fY], Yf=env["Yf"], i=0, qgdFlux=lambda flux, psi, psif: flux * psif,
                    phiJm=phiJm, phi=phi, tauQGDf=tauf, Uf=lin(Pair(*U)), phiJmY=[None], diffusiveFlux=[None])
        exec(src, env2)
        out = dict(nv=nv, pts=np.array([q.c for q in pts] + ([[0, 0, 0]] if nv == 3 else [])), Sf=S.c, Cf=Cf.c, C=np.array([own.c, nei.c]),
                   U=np.array([u.c for u in U]), Y=Y, phiJm=phiJm, phi=phi, tauQGDf=tauf, gradYf=env2["gradYf"].c, phiJmY=env2["phiJmY"][0],
                   diffusiveFlux=env2["diffusiveFlux"][0])
        for k in names:
            rec[k].append(np.array(out[k], dtype=float))
    return {k: np.array(v) for k, v in rec.items()}


# ----------------------------------------------------------------------------------------------------------------------
# round 3: the listing sections that were still outside the mechanical pin
# ----------------------------------------------------------------------------------------------------------------------
class Stream:
    """Info << ... << endl"""
    def __lshift__(self, o): return self


def gvp2d_vec(nfaces=30, seed=21):
    """The 2-D GaussVolPoint operators on VECTOR fields, one internal quad between two cells of a one-cell-thick mesh: the gradient
    through the per-component calls and the re-pack of GaussVolPointBase.C L90-100 (gradf component 3*i+j = d_i U_j), the
    divergence through GaussVolPointBase2D.C L386-396 (vector) and L447-450, L464-484 (tensor)."""
    f2 = "GaussVolPointBase2D_8C_source.html"
    coef_src = transpile(lines(f2, 154, 168))
    apply_src = transpile(lines(f2, 317, 328))
    repack_src = transpile(lines("GaussVolPointBase_8C_source.html", 90, 100))
    divv_src = transpile(lines(f2, 386, 396))
    idx_src = transpile(lines(f2, 447, 450))
    divt_src = transpile(lines(f2, 464, 484))
    rng = np.random.default_rng(seed)
    rec = {k: [] for k in ("ie3", "pts", "Sf", "Cf", "C", "U", "Tn", "grad_v", "div_v", "div_t")}
    for n in range(nfaces):
        ie3 = n % 3
        ie1, ie2 = (1, 2) if ie3 == 0 else ((0, 2) if ie3 == 1 else (0, 1))
        e1, e2 = Vec(*np.eye(3)[ie1]), Vec(*np.eye(3)[ie2])

        def place(a, b, h):
            v = np.zeros(3); v[ie1], v[ie2], v[ie3] = a, b, h
            return Vec(*v)
        C4 = place(*(0.2 * rng.standard_normal(2)), 0.05)
        C2 = place(1.0 + 0.2 * rng.standard_normal(), 0.2 * rng.standard_normal(), 0.05)
        a0, b0 = 0.5 + 0.2 * rng.standard_normal(), -0.5 + 0.2 * rng.standard_normal()
        a1, b1 = 0.5 + 0.2 * rng.standard_normal(), 0.5 + 0.2 * rng.standard_normal()
        order = [(a0, b0, 0.0), (a0, b0, 0.1), (a1, b1, 0.1), (a1, b1, 0.0)]
        shift = n % 4
        order = order[shift:] + order[:shift]
        pts = [place(*o) for o in order]
        upper = [k for k, o in enumerate(order) if o[2] >= 0.05]
        ip1, ip3 = upper[0], upper[1]
        mesh = Obj(C=call([C4, C2]), points=call(pts))
        one = lambda: [None]  # noqa: E731
        env = dict(mesh=mesh, iFace=0, ic2=1, ic4=0, ip1=ip1, ip3=ip3, e1_=e1, e2_=e2, mag=mag, v42=one(), v13=one(), mv42_=one(),
                   mv13_=one(), cosa1=one(), cosa2=one(), sina1=one(), sina2=one(), den=one(), c1_=one(), c2_=one(), c3_=one(), c4_=one())
        exec(coef_src, env)
        U = [rng.standard_normal(3), rng.standard_normal(3)]            # owner (ic4), neighbour (ic2)
        comp_grads = []
        for k in range(3):
            fvals = [float(U[0][k]), float(U[1][k])]
            pvals = [inv_dist(x, [C4, C2], fvals) for x in pts]
            g = [[0.0, 0.0, 0.0]]
            e = dict(env)
            e.update(f=fvals, pF=pvals, gradf=g, ic2_=[1], ic4_=[0], ip1_=[ip1], ip3_=[ip3], ie1_=ie1, ie2_=ie2, ie3_=ie3, dfdn=0.0, dfdt=0.0)
            exec(apply_src, e)
            comp_grads.append(g)

        class VFld(list):      # a surfaceVectorField's internal part: .primitiveField().component(c)
            def primitiveField(self): return self
            def component(self, c): return [v[c] for v in self]

        class TFld(list):      # a surfaceTensorField's internal part: .primitiveFieldRef().replace(k, values)
            def primitiveFieldRef(self): return self
            def replace(self, k, vals):
                for row, v in zip(self, vals):
                    row[k] = v
        gradf = TFld([[None] * 9])
        exec(repack_src, dict(gradf=gradf, gradU=VFld(comp_grads[0]), gradV=VFld(comp_grads[1]), gradW=VFld(comp_grads[2])))
        # divergence of the vector field
        fU = [list(map(float, U[0])), list(map(float, U[1]))]
        pU = [list(inv_dist(x, [C4, C2], [Vec(*U[0]), Vec(*U[1])]).c) for x in pts]
        dv = [None]
        e = dict(env)
        e.update(f=fU, pF=pU, divf=dv, ic2_=[1], ic4_=[0], ip1_=[ip1], ip3_=[ip3], ie1_=ie1, ie2_=ie2, ie3_=ie3, df1dn=0.0, df2dn=0.0, df1dt=0.0, df2dt=0.0)
        exec(divv_src, e)
        # divergence of a tensor field
        Tn = [rng.standard_normal(9), rng.standard_normal(9)]
        fT = [list(map(float, Tn[0])), list(map(float, Tn[1]))]
        w = [1.0 / float(np.sqrt(((x.c - c.c) ** 2).sum())) for x in pts for c in (C4, C2)]
        pT = []
        for k, x in enumerate(pts):
            w0, w1 = w[2 * k], w[2 * k + 1]
            pT.append(list((w0 / (w0 + w1)) * Tn[0] + (w1 / (w0 + w1)) * Tn[1]))
        dt_ = [[0.0, 0.0, 0.0]]
        e = dict(env)
        e.update(f=fT, pF=pT, divf=dt_, ic2_=[1], ic4_=[0], ip1_=[ip1], ip3_=[ip3], ie1_=ie1, ie2_=ie2, ie3_=ie3)
        for nm in ("df11dn", "df11dt", "df21dn", "df21dt", "df22dn", "df22dt", "df12dn", "df12dt"):
            e[nm] = 0.0
        exec(idx_src, e)
        exec(divt_src, e)
        S, cf = face_area_centre(pts)
        rec["ie3"].append(ie3); rec["pts"].append(np.array([p.c for p in pts])); rec["Sf"].append(S); rec["Cf"].append(cf)
        rec["C"].append(np.array([C4.c, C2.c])); rec["U"].append(np.array(U)); rec["Tn"].append(np.array(Tn))
        rec["grad_v"].append(gradf[0]); rec["div_v"].append(dv[0]); rec["div_t"].append(dt_[0])
    return {k: np.array(v, dtype=float) for k, v in rec.items()}


def gvp_other(nfaces=16, seed=22):
    """Internal faces with MORE THAN FOUR vertices: the 3-D GaussVolPoint gradient falls back to dfdn = nf * snGrad
    [GaussVolPointBase3D.C L945-948 / L976-979, L759-768 / L856-865]; nf = Sf / |Sf| [fvscStencil.C L126-129]; fvc::snGrad is the
    uncorrected one, nonOrthDeltaCoeffs (phi_N - phi_O) (L0)."""
    f3 = "GaussVolPointBase3D_8C_source.html"
    nf_l = listing("fvscStencil_8C_source.html")
    nf_src = transpile(["nf_ = " + nf_l[128].strip() + ";"])
    dfdn_s = transpile(lines(f3, 945, 948))
    dfdn_v = transpile(lines(f3, 976, 979))
    other_s = transpile(lines(f3, 760, 768))
    other_v = transpile(lines(f3, 857, 865))
    rng = np.random.default_rng(seed)
    rec = {k: [] for k in ("nv", "pts", "Sf", "Cf", "C", "cell_s", "cell_v", "grad_s", "grad_v")}
    for n in range(nfaces):
        nv = 5 + n % 2
        ang = np.sort(rng.uniform(0, 2 * np.pi, nv))
        pts = [Vec(0.05 * rng.standard_normal(), 0.5 + 0.5 * np.cos(a), 0.5 + 0.5 * np.sin(a)) for a in ang]
        own = Vec(*(np.array([-0.5, 0.5, 0.5]) + 0.15 * rng.standard_normal(3)))
        nei = Vec(*(np.array([0.5, 0.5, 0.5]) + 0.15 * rng.standard_normal(3)))
        S, cf = face_area_centre(pts)
        if S[0] < 0:
            pts = pts[::-1]
            S, cf = face_area_centre(pts)
        Sv = Vec(*S)
        d = nei - own
        nhat = Sv / mag(Sv)
        delta = 1.0 / max(nhat & d, 0.05 * mag(d))                       # nonOrthDeltaCoeffs (L0)
        env = dict(mesh_=Obj(Sf=call(Fld([Sv])), magSf=call(Fld([mag(Sv)]))))

        class DivFld(Fld):
            def __truediv__(self, o): return Fld([a / b for a, b in zip(self, o)])
        env["mesh_"] = Obj(Sf=call(DivFld([Sv])), magSf=call(Fld([mag(Sv)])))
        exec(nf_src, env)
        nf = env["nf_"]
        fs = [float(rng.standard_normal()), float(rng.standard_normal())]
        fv = [rnd_vec(rng), rnd_vec(rng)]

        class Fld2(Fld):      # nf * snGrad: vector x scalar, vector x vector (outer product), face by face
            def __mul__(self, o): return Fld([a * b for a, b in zip(self, o)])
        e = dict(nf=Fld2(nf), fvc=Obj(snGrad=lambda fld: Fld([delta * (fld[1] - fld[0])])), sf=fs, vf=fv)
        exec(dfdn_s, e)
        gs = [None]
        exec(other_s, dict(of_=[0], gradf=Obj(primitiveFieldRef=call(gs)), dfdn=Obj(primitiveField=call(e["dfdn"]))))
        exec(dfdn_v, e)
        gv = [None]
        exec(other_v, dict(of_=[0], gradf=Obj(primitiveFieldRef=call(gv)), dfdn=Obj(primitiveField=call(e["dfdn"]))))
        rec["nv"].append(nv); rec["pts"].append(np.array([p.c for p in pts] + [[0, 0, 0]] * (6 - nv))); rec["Sf"].append(S); rec["Cf"].append(cf)
        rec["C"].append(np.array([own.c, nei.c])); rec["cell_s"].append(fs); rec["cell_v"].append(np.array([v.c for v in fv]))
        rec["grad_s"].append(gs[0].c); rec["grad_v"].append(gv[0].m.reshape(9))
    return {k: np.array(v, dtype=float) for k, v in rec.items()}


def qgdlength():
    """QGDCoeffs::updateQGDLength [QGDCoeffs.C L298-376] executed as listed over WHOLE small meshes (hQGDf of internal and patch
    faces, the area-weighted hQGD of every cell, hQGD on the patches), starting from hQGDf = 1 / |deltaCoeffs| [L195-199].  Mesh
    geometry (C, Cf, |Sf|, deltaCoeffs) and addressing (cells() = owned faces in ascending order, then neighbour faces) are the
    L0 inputs; three meshes: jittered hexahedra, the same with split quads (prism-like cells), a one-cell-thick plane whose empty
    patches must be skipped."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    import qgdsolver_amd as q
    G, E = 0, 1
    src = transpile(lines("QGDCoeffs_8C_source.html", 300, 375))
    init_l = listing("QGDCoeffs_8C_source.html")[198].strip()
    init_src = transpile(["hinit = " + init_l.replace("mesh.surfaceInterpolation::deltaCoeffs()", "deltaCoeffs") + ";"])
    meshes = [q.PolyMesh.box(3, 2, 2).jitter(0.15, seed=31), q.PolyMesh.box(3, 2, 2).jitter(0.1, seed=32).split_quads(3),
              q.PolyMesh.box(4, 3, 1, hi=(1.0, 0.75, 0.1), patch_types=[G, G, G, G, E, E]).jitter(0.12, seed=33)]
    out = {"nMeshes": np.array(len(meshes))}
    for mi, m in enumerate(meshes):
        prim = m.primitives()
        nIF, nF, nC = m.nInternalFaces, m.nFaces, m.nCells
        Cc, Cf = m.array("C").reshape(-1, 3), m.array("Cf").reshape(-1, 3)
        magSf, dc = m.array("magSf"), m.array("deltaCoeffs")
        own, nei = prim["owner"], prim["neighbour"]
        ps, pz, pt = prim["patchStart"], prim["patchSize"], prim["patchType"]
        cells = [[] for _ in range(nC)]
        for f in range(nF):
            cells[own[f]].append(f)
        for f in range(nIF):
            cells[nei[f]].append(f)

        class PF(Fld):
            def __imul__(self, o): return PF([a * o for a in self])
            def __mul__(self, o): return PF([a * o for a in self])

        class AbsFld(list):
            pass
        e0 = dict(mag=lambda fl: [abs(x) for x in fl], deltaCoeffs=list(dc))

        class Recip(float):
            def __truediv__(self, lst): return [float(self) / x for x in lst]
        # `1.0 / mag(...)`: a scalar divided by a field
        init_py = init_src.replace("1.0 /", "Recip(1.0) /")
        e0["Recip"] = Recip
        exec(init_py, e0)
        hinit = e0["hinit"]
        fv_size = [0 if t == E else int(z) for t, z in zip(pt, pz)]          # emptyFvPatch::size() == 0 (L0)
        hb = [PF([hinit[int(st) + k] for k in range(n)]) for st, n in zip(ps, fv_size)]
        hint = list(hinit[:nIF])

        class FaceField:
            def primitiveField(self): return hint
            def primitiveFieldRef(self): return hint
            def boundaryFieldRef(self): return hb
            def boundaryField(self): return hb
            def __getitem__(self, f): return hint[f]

        class MagSf:
            def __getitem__(self, f): return float(magSf[f])
            def boundaryField(self): return [[float(magSf[int(st) + k]) for k in range(n)] for st, n in zip(ps, fv_size)]

        class BMesh:
            def whichPatch(self, f):
                for k, (st, n) in enumerate(zip(ps, pz)):
                    if st <= f < st + n:
                        return k
                return -1
            def __getitem__(self, k): return Obj(whichFace=lambda f, st=int(ps[k]): f - st)
        kinds = {G: "fvPatch", E: "emptyFvPatch"}
        patches = [Obj(kind=kinds[int(t)], coupled=lambda: False) for t in pt]
        hc, hcb = [0.0] * nC, [None] * len(pt)

        class CellField:
            def __len__(self): return nC
            def primitiveFieldRef(self): return hc
            def boundaryFieldRef(self): return hcb
        mesh = Obj(C=call([Vec(*c) for c in Cc]), Cf=call([Vec(*c) for c in Cf]), owner=call(list(own)), neighbour=call(list(nei)),
                   boundary=call(patches), cells=call(cells), isInternalFace=lambda f: f < nIF, magSf=call(MagSf()), boundaryMesh=call(BMesh()))
        env = dict(mesh=mesh, hQGDf_=FaceField(), hQGD_=CellField(), mag=mag, min=min, isA=lambda kind, o: o.kind == kind)
        exec(src, env)
        hf = np.zeros(nF)
        hf[:nIF] = hint
        hbnd = np.zeros(nF - nIF)
        for k, (st, n) in enumerate(zip(ps, fv_size)):
            for j in range(n):
                hf[int(st) + j] = hb[k][j]
                hbnd[int(st) - nIF + j] = hcb[k][j] if hcb[k] is not None else 0.0
        for key, val in prim.items():
            out[f"m{mi}_{key}"] = np.asarray(val)
        out[f"m{mi}_hQGDf"], out[f"m{mi}_hQGD"], out[f"m{mi}_hQGDb"] = hf, np.array(hc), hbnd
    return out


def courant(case):
    """QGDCourantNo.H L36-53 and setDeltaT-QGDQHD.H L41-61 on the one-face cases of case2cell (Uf, cf, hQGDf, tauQGDf, Sf as that
    fixture's listing-text evaluation left them), and the speed of sound of hePsiQGDThermo.C L123-124 from psi = 1/(R T), gamma =
    Cp/Cv of the cells.  Inputs beyond case2cell's: maxCo, maxDeltaT, cTau."""
    co_src = transpile(lines("QGDCourantNo_8H_source.html", 36, 53))
    dt_src = transpile(lines("setDeltaT-QGDQHD_8H_source.html", 41, 61))
    th = listing("hePsiQGDThermo_8C_source.html")
    c_src = transpile([th[123].replace("this->", "").replace("==", "="), th[124].replace("this->", "")])
    rng = np.random.default_rng(23)
    n = len(case["nv"])
    rec = {k: [] for k in ("maxCo", "maxDeltaT", "cTau", "CoNum", "deltaT1", "c_cells")}
    for i in range(n):
        S = Vec(*case["Sf"][i])
        Uf, cf, hf, tauf, dt = Vec(*case["Uf"][i]), float(case["cf"][i]), float(case["hQGDf"][i]), float(case["tauQGDf"][i]), float(case["deltaT"][i])
        maxCo, maxDeltaT, cTau = float(rng.uniform(0.05, 0.6)), float(10.0 ** rng.uniform(-4, 0)), float(rng.uniform(0.3, 0.9))
        ctrl = {"adjustTimeStep": True, "cTau": cTau}
        holder = {"dt": dt}

        class M(float):       # max(Cof) / min(tauQGDf) of a one-face field
            def value(self): return float(self)

        def fmax(*a): return M(a[0]) if len(a) == 1 else max(a)
        def fmin(*a): return M(a[0]) if len(a) == 1 else min(a)

        class VecDiv(Vec):    # mesh.Sf() / mesh.magSf()
            pass
        runTime = Obj(controlDict=call(Obj(lookupOrDefault=lambda key, dflt: ctrl.get(key, dflt))), deltaT=call(dt), deltaTValue=lambda: holder["dt"],
                      setDeltaT=lambda v: holder.__setitem__("dt", float(v)))
        env = dict(runTime=runTime, Uf=Uf, cf=cf, hQGDf=hf, mesh=Obj(Sf=call(S), magSf=call(mag(S))), max=fmax, mag=mag, Info=Stream(), endl=None,
                   CoNum=0.0, bool=bool, false=False, true=True)
        src = co_src.replace("lookupOrDefault<bool>", "lookupOrDefault")
        exec(src, env)
        env2 = dict(adjustTimeStep=True, maxCo=maxCo, CoNum=env["CoNum"], SMALL=1e-15, min=fmin, runTime=runTime, maxDeltaT=maxDeltaT,
                    thermo=Obj(tauQGDf=call(call(tauf))), Info=Stream(), endl=None)
        exec(dt_src.replace("lookupOrDefault<scalar>", "lookupOrDefault"), env2)
        R, Cv = float(case["R"][i]), float(case["Cv"][i])
        cc = []
        for T in case["T"][i]:
            e3 = dict(Cp=call(Cv + R), Cv=call(Cv), psi=call(1.0 / (R * float(T))), sqrt=lambda x: float(np.sqrt(x)))
            exec(c_src, e3)
            cc.append(e3["c_"])
        for k, v in (("maxCo", maxCo), ("maxDeltaT", maxDeltaT), ("cTau", cTau), ("CoNum", float(env["CoNum"])), ("deltaT1", holder["dt"]),
                     ("c_cells", cc)):
            rec[k].append(v)
    out = {k: np.array(v, dtype=float) for k, v in rec.items()}
    for k in ("nv", "pts", "Sf", "Cf", "C", "V", "U", "T", "p", "R", "Cv", "mu", "Pr", "ScQGD", "PrQGD", "alphaQGD", "deltaT"):
        out[k] = case[k]
    return out


def thermo2cell(case):
    """thermo.correct() and the pressure update after the explicit step, executed as listed on the two cells of every case2cell case:
    hePsiQGDThermo.C L48-64 (T from e, psi, mu, alpha per cell), L123-124 (gamma, c) and QGDFoam.C L152-154 (p = rho / psi), with the
    new rho and e of that fixture.  `mixture_` is the L0 part: perfectGas (psi = 1 / (R T)), eConst with Tref = 0, Hf = 0 (THE: T = e /
    Cv -- the value OpenFOAM's Newton iteration converges to), constTransport (mu, alphah = mu / Pr)."""
    th = listing("hePsiQGDThermo_8C_source.html")
    cell_txt = [th[i] for i in range(48, 65) if i not in (50, 51)]          # without the declaration of the reference mixture_
    cell_src = transpile(cell_txt)
    c_src = transpile([th[123].replace("this->", "").replace("==", "="), th[124].replace("this->", "")])
    qf = listing("QGDFoam_8C_source.html")
    p_expr = " ".join(qf[i].strip() for i in (152, 153, 154)).rstrip(";")
    assert p_expr.replace(" ", "") == "p.ref()=rho()/psi()", p_expr
    rec = {k: [] for k in ("T1", "psi1", "c1", "p1", "mu1", "alpha1")}
    for i in range(len(case["nv"])):
        R, Cv, mu0, Pr = float(case["R"][i]), float(case["Cv"][i]), float(case["mu"][i]), float(case["Pr"][i])
        mix = Obj(THE=lambda he, p, T0: he / Cv, psi=lambda p, T: 1.0 / (R * T), mu=lambda p, T: mu0, alphah=lambda p, T: mu0 / Pr)
        env = dict(TCells=[float(t) for t in case["T"][i]], hCells=[float(e) for e in case["e1"][i]], pCells=[float(x) for x in case["p"][i]],
                   psiCells=[0.0, 0.0], muCells=[0.0, 0.0], alphaCells=[0.0, 0.0], mixture_=mix)
        exec(cell_src, env)
        cc, pp = [], []
        for k in range(2):
            e3 = dict(Cp=call(Cv + R), Cv=call(Cv), psi=call(env["psiCells"][k]), sqrt=lambda x: float(np.sqrt(x)))
            exec(c_src, e3)
            cc.append(e3["c_"])
            pp.append(eval(p_expr.split("=", 1)[1].replace("rho()", "rho_").replace("psi()", "psi_"), dict(rho_=float(case["rho1"][i][k]), psi_=env["psiCells"][k])))
        for k, v in (("T1", env["TCells"]), ("psi1", env["psiCells"]), ("c1", cc), ("p1", pp), ("mu1", env["muCells"]), ("alpha1", env["alphaCells"])):
            rec[k].append(v)
    return {k: np.array(v, dtype=float) for k, v in rec.items()}


def qhdclosure(nfaces=24, seed=24):
    """tauQGD of the four QHD closures as listed [constTau.C L71-74, HbyUQHD.C L80-83, T0byGr.C L84-87, H2bynuQHD.C L78-82] on the two
    cells of a one-face mesh (hQGDf from QGDCoeffs.C L305-307; a cell with one face has hQGD = hQGDf), tauQGDf = linearInterpolate."""
    h_src = transpile(lines("QGDCoeffs_8C_source.html", 305, 307))
    def body(fname, a, b): return transpile([ln.replace("this->", "") for ln in lines(fname, a, b)])
    texts = [body("constTau_8C_source.html", 73, 74), body("HbyUQHD_8C_source.html", 82, 83), body("T0byGr_8C_source.html", 86, 87),
             body("H2bynuQHD_8C_source.html", 80, 82)]
    rng = np.random.default_rng(seed)
    names = ("nv", "pts", "Sf", "Cf", "C", "model", "Tau", "aQGD", "UQHD", "T0", "Gr", "mu", "rho0", "tauQGDf", "tauQGD")
    rec = {k: [] for k in names}

    class PS(Pair):
        def _b(self, o): return o if isinstance(o, Pair) else Pair(o, o)
        def __mul__(self, o): o = self._b(o); return PS(self.o * o.o, self.n * o.n)
        def __rmul__(self, o): return PS(o * self.o, o * self.n)
        def __truediv__(self, o): o = self._b(o); return PS(self.o / o.o, self.n / o.n)
    for n in range(nfaces):
        nv = 4 if n % 3 != 2 else 3
        model = n % 4
        pts, own, nei = skew_face(rng, nv)
        S, cf = face_area_centre(pts)
        Sv, Cf = Vec(*S), Vec(*cf)
        hq = [0.0]
        ev = dict(mag=mag, min=min, mesh=Obj(C=call([own, nei]), Cf=call([Cf]), owner=call([0]), neighbour=call([1])), iFace=0,
                  hQGDf_=Obj(primitiveFieldRef=call(hq)), hown=0.0, hnei=0.0)
        exec(h_src, ev)
        hf = hq[0]
        sfo, sfn = abs(Sv & (Cf - own)), abs(Sv & (nei - Cf))
        w = sfn / (sfo + sfn)
        lin = lambda qq: w * (qq.o - qq.n) + qq.n  # noqa: E731
        Tau, aQ, UQ, T0, Gr = float(10 ** rng.uniform(-4, -2)), float(rng.uniform(0.2, 0.8)), float(rng.uniform(0.1, 2.0)), float(rng.uniform(0.5, 2)), float(10 ** rng.uniform(2, 5))
        mu, rho0 = float(10 ** rng.uniform(-4, -2)), float(rng.uniform(0.8, 1.3))
        env = dict(tau_=Tau, aQGD_=aQ, hQGD_=PS(hf, hf), UQHD_=UQ, T0_=T0, Gr_=Gr, linearInterpolate=lin, sqr=lambda x: x * x,
                   dimensionedScalar=lambda nm, dims, v: PS(v, v), dimTime=1.0, dimLength=1.0,     # a uniform value assigned to a field
                   qgdThermo=Obj(mu=call(PS(mu, mu)), rho=call(PS(rho0, rho0))))
        exec(texts[model], env)
        tq, tf = env["tauQGD_"], env["tauQGDf_"]
        vals = dict(nv=nv, pts=np.array([q_.c for q_ in pts] + ([[0, 0, 0]] if nv == 3 else [])), Sf=S, Cf=cf, C=np.array([own.c, nei.c]), model=model,
                    Tau=Tau, aQGD=aQ, UQHD=UQ, T0=T0, Gr=Gr, mu=mu, rho0=rho0, tauQGDf=float(tf), tauQGD=[float(tq.o), float(tq.n)])
        for k in names:
            rec[k].append(np.array(vals[k], dtype=float))
    return {k: np.array(v) for k, v in rec.items()}


def casebnd(nfaces=30, seed=25):
    """The QGDFoam flux assembly on ONE BOUNDARY FACE (3-D, GaussVolPoint): updateFields.H L45-80 with the patch values, the three
    fvsc::grad of U, e, rho through the boundary-face text of GaussVolPointBase3D.C, updateFluxes.H L41-139, and -- at the place
    where the listing calls fvsc::grad(p) -- what GaussVolPoint does there first [GaussVolPointStencil.C L73]: p's boundary
    conditions again, i.e. qgdFluxFvPatchScalarField::updateCoeffs [qgdFluxFvPatchScalarField.C L184-192] with the phiwStar just
    formed, then fixedGradient's evaluate (patch value = cell value + gradient / deltaCoeffs, L0), then the vertex values of p.
    constScPrModel1.C L103-104 and its patch loop L121-128 give tauQGDf, muQGD, alphauQGD of the patch face; hQGDf of a patch face is
    2 / deltaCoeffs [QGDCoeffs.C L195-199, L310-317].  Patch: U fixedValue, T zeroGradient, p qgdFlux (gradient 0 at start-up)."""
    text = Gvp3dText()
    fields_src = transpile(lines("QGDFoam_2updateFields_8H_source.html", 45, 80))
    flux_src = transpile(lines("QGDFoam_2updateFluxes_8H_source.html", 41, 139))
    cs = listing("constScPrModel1_8C_source.html")
    tauf_src = transpile([cs[103].replace("this->", "")])
    taub_src = transpile([cs[104].replace("this->", "")])
    mub_src = transpile([cs[i] for i in range(121, 129)])
    bc_l = listing("qgdFluxFvPatchScalarField_8C_source.html")
    bc_src = transpile([bc_l[i].replace("this->gradient()", "gradient_") for i in range(184, 193)])
    rng = np.random.default_rng(seed)
    names = ("nv", "pts", "Sf", "Cf", "C", "U", "T", "p", "Ub", "R", "Cv", "mu", "Pr", "ScQGD", "PrQGD", "alphaQGD", "gradient", "pMid",
             "tauQGDf", "gradUf", "gradef", "gradRhof", "gradPf", "phiwStar", "phiJm", "phi", "phiJmU", "phiP", "phiPi", "phiJmH", "phiQ", "phiPiU")
    rec = {k: [] for k in names}

    class BF(Fld):
        def __truediv__(self, o): return BF([a / b for a, b in zip(self, o)]) if isinstance(o, list) else BF([a / o for a in self])
        def __mul__(self, o): return BF([a * b for a, b in zip(self, o)]) if isinstance(o, list) else BF([a * o for a in self])
        def __neg__(self): return BF([-a for a in self])
    for n in range(nfaces):
        nv = 4 if n % 3 != 2 else 3
        pts, own, _ = skew_face(rng, nv)
        S, cf = face_area_centre(pts)
        Sv, Cf = Vec(*S), Vec(*cf)
        magS = mag(Sv)
        nhat = S / np.sqrt((S * S).sum())
        dc = 1.0 / abs(float(nhat @ (cf - own.c)))           # fvPatch::deltaCoeffs, patch-normal delta (L0)
        R = 1 / 1.4
        Cv = R / 0.4
        Cp = Cv + R
        gam = Cp / Cv
        mu0, Pr = float(rng.uniform(0, 2e-3)), float(rng.uniform(0.6, 1.2))
        Sc, PrQ, aQ = float(rng.uniform(0.5, 1.5)), float(rng.uniform(0.5, 1.5)), float(rng.uniform(0.3, 0.7))
        Uo, Ub = rnd_vec(rng, 0.7), rnd_vec(rng, 0.7)
        To, po = float(rng.uniform(0.8, 1.3)), float(rng.uniform(0.7, 1.4))
        # createFields.H on the patch (L0 boundary conditions): T zeroGradient, p qgdFlux with gradient 0, U fixedValue
        Tb, pb = To, po
        eo, eb = Cv * To, Cv * Tb
        psio, psib = 1.0 / (R * To), 1.0 / (R * Tb)
        rhoo, rhob = psio * po, psib * pb
        cb = float(np.sqrt(gam / psib))
        alphah0 = (Cp * mu0 * (1.0 / Pr)) / Cp
        hfb = (1.0 / abs(dc)) * 2.0                                                 # QGDCoeffs.C L195-199 (1/|deltaCoeffs|), L315 (*= 2)
        hb = hfb * 1.0                                                              # L373
        B = lambda v: Pair(v, v)  # noqa: E731   a patch value where the listing interpolates: qgdInterpolate returns it
        lin = lambda qq: qq.n  # noqa: E731
        ev = dict(aQGD_=B(aQ), cSound=B(cb), hQGDf_=hfb, linearInterpolate=lin)
        exec(tauf_src, ev)
        tauf = ev["tauQGDf_"]
        ev = dict(aQGD_=aQ, hQGD_=hb, cSound=cb)
        exec(taub_src, ev)
        taub = ev["tauQGD_"]
        m_, a_ = [[0.0]], [[0.0]]
        ev = dict(p=Obj(boundaryField=call([[pb]])), ScQGD_=Obj(boundaryField=call([[Sc]])), PrQGD_=Obj(boundaryField=call([[PrQ]])),
                  tauQGD_=Obj(boundaryField=call([[taub]])), patchi=0, facei=0, muQGD_=Obj(boundaryFieldRef=call(m_)),
                  alphauQGD_=Obj(boundaryFieldRef=call(a_)))
        exec(mub_src, ev)
        muQb, alQb = m_[0][0], a_[0][0]
        muEff = B(0.0 + (mu0 + muQb))
        alphaEff = B(gam * ((alphah0 + alQb) + 0.0))
        rhoUb = rhob * Ub
        rhoEb = rhob * eb + rhob * 0.5 * (Ub & Ub)
        env = dict(qgdInterpolate=lin, rho=B(rhob), U=B(Ub), rhoU=B(rhoUb), p=B(pb), gamma=B(gam), rhoE=B(rhoEb),
                   thermo=Obj(c=call(B(cb)), Cp=call(B(Cp))), turbulence=Obj(alphaEff=call(alphaEff), muEff=call(muEff)))
        exec(fields_src, env)
        # the three gradients that do not involve p's boundary condition: boundary-face text, vertex values = patch value (L0)
        gU, _ = text.boundary(pts, own, Cf, Uo, Ub, dc * (Ub - Uo), [Ub] * nv, "grad_v")
        gE, _ = text.boundary(pts, own, Cf, eo, eb, 0.0, [eb] * nv, "grad_s")                         # gradientEnergy with zero gradient
        gR, _ = text.boundary(pts, own, Cf, rhoo, rhob, dc * (rhob - rhoo), [rhob] * nv, "grad_s")   # calculated patch: generic snGrad
        mid = {}

        def grad_of(fld):
            if fld != "p":
                return {"U": Tensor(gU), "e": Vec(*gE), "rho": Vec(*gR)}[fld]
            # correctBoundaryConditions() of p inside fvsc::grad(p): qgdFlux::updateCoeffs with the registered phiwStar, tauQGDf
            # (the listing's `phiw` is the surfaceScalarField registered under the name "phiwStar" [createFaceFluxes.H])
            e2 = dict(phiws=Obj(boundaryField=call([BF([env2["phiw"]])])), tauQGDf=Obj(boundaryField=call([BF([tauf])])),
                      patch=call(Obj(index=call(0), magSf=call(BF([magS])))))
            exec(bc_src, e2)
            mid["gradient"] = e2["gradient_"][0]
            mid["pMid"] = po + mid["gradient"] / dc                                                   # fixedGradient::evaluate (L0)
            gP, _ = text.boundary(pts, own, Cf, po, mid["pMid"], mid["gradient"], [mid["pMid"]] * nv, "grad_s")
            return Vec(*gP)
        env2 = {k: env[k] for k in ("rhof", "Uf", "rhoUf", "UrhoUf", "pf", "gammaf", "Hf", "alphauf", "muf")}
        env2.update(tr=tr, tauQGDf=tauf, mesh=Obj(Sf=call(Sv)), fvsc=Obj(grad=grad_of), U="U", e="e", rho="rho", p="p",
                    I=Sph(1.0), implicitDiffusion=False, Foam=Obj(T=lambda t: t.T()), qgdFlux=lambda flux, psi, psif: flux * psif, H="H")
        exec(flux_src, env2)
        g = env2

        def val(x):
            return x.c if isinstance(x, Vec) else (x.m.reshape(9) if isinstance(x, Tensor) else x)
        out = dict(nv=nv, pts=np.array([q_.c for q_ in pts] + ([[0, 0, 0]] if nv == 3 else [])), Sf=S, Cf=cf, C=own.c, U=Uo.c, T=To, p=po, Ub=Ub.c,
                   R=R, Cv=Cv, mu=mu0, Pr=Pr, ScQGD=Sc, PrQGD=PrQ, alphaQGD=aQ, gradient=mid["gradient"], pMid=mid["pMid"], tauQGDf=tauf,
                   phiwStar=g["phiw"])
        for k in ("gradUf", "gradef", "gradRhof", "gradPf", "phiJm", "phi", "phiJmU", "phiP", "phiPi", "phiJmH", "phiQ", "phiPiU"):
            out[k] = val(g[k])
        for k in names:
            rec[k].append(np.array(out[k], dtype=float))
    return {k: np.array(v) for k, v in rec.items()}


def qhdeqn(nfaces=30, seed=26, implicit=False):
    """One whole QHDFoam step on one internal face between two cells (3-D, GaussVolPoint; implicit: the implicitDiffusion branch
    of QHDUEqn.H L46-65 / QHDTEqn.H L69-80, i.e. fvm::laplacian, selected by the listing's own `if (implicitDiffusion)`): updateFields.H L36-73,
    updateFluxes.H L33-38, QHDpEqn.H L35-47, QHDUEqn.H L36-43 + L68-84, QHDTEqn.H L65-66 + L83-91 and the reference level of
    QHDFoam.C L123-130 -- every line from the listing text, the fvsc gradients through the GaussVolPoint text.  fvm / fvc are
    emulated on the two-cell mesh with OpenFOAM's conventions (L0): fvc::div = surfaceIntegrate, fvc::grad Gauss linear,
    fvc::laplacian = div(Gamma |Sf| snGrad) with the uncorrected snGrad, fvm::laplacian the same as a matrix, volume-weighted
    sources, setReference doubling the diagonal, flux() = -a (p_N - p_O); Euler ddt."""
    text = Gvp3dText()
    fields_src = transpile(lines("QHDFoam_2updateFields_8H_source.html", 36, 73))
    flux_src = "\n".join(l for l in transpile(lines("QHDFoam_2updateFluxes_8H_source.html", 33, 38)).split("\n") if "setOriented" not in l)
    peqn_src = transpile(lines("QHDpEqn_8H_source.html", 35, 47))
    ueqn_a = "\n".join(l for l in transpile(lines("QHDUEqn_8H_source.html", 36, 43)).split("\n") if "setOriented" not in l)
    ueqn_b = transpile(lines("QHDUEqn_8H_source.html", 46, 85))
    teqn_a = transpile(lines("QHDTEqn_8H_source.html", 65, 66))
    teqn_b = transpile(lines("QHDTEqn_8H_source.html", 69, 92))
    ref_src = transpile(lines("QHDFoam_8C_source.html", 123, 131))
    rng = np.random.default_rng(seed)
    names = ("nv", "pts", "Sf", "Cf", "C", "U", "T", "p", "rho0", "mu", "Pr", "beta", "g", "Tau", "deltaT", "pRefCell", "pRefValue", "delta",
             "p1", "phi1", "U1", "T1", "phiu", "phiwo")
    rec = {k: [] for k in names}

    class QF(list):
        """a vol field on the two cells: elementwise arithmetic, the old-time level, and the verbs the listing uses"""
        def __init__(self, vals, old=None):
            super().__init__(vals)
            self.old = list(old if old is not None else vals)
        def _z(self, o, f): return QF([f(a, b) for a, b in zip(self, o)]) if isinstance(o, list) else QF([f(a, o) for a in self])
        def __add__(self, o): return o.__radd__(self) if isinstance(o, Eq) else self._z(o, lambda a, b: a + b)
        def __sub__(self, o): return o.__rsub__(self) if isinstance(o, Eq) else self._z(o, lambda a, b: a - b)
        def __mul__(self, o): return self._z(o, lambda a, b: a * b)
        def __rmul__(self, o): return QF([o * a for a in self])
        def __truediv__(self, o): return self._z(o, lambda a, b: a * (1.0 / b))
        def __neg__(self): return QF([a * -1.0 for a in self])
        def __iadd__(self, o): self[:] = [a + o for a in self]; return self
        def correctBoundaryConditions(self): pass
        def oldTime(self): return self
        def needReference(self): return True
        def dimensions(self): return None
        def assign(self, vals): self[:] = list(vals)

    class Eq:
        """M psi = b on the two cells; M = diag(d) - a (offdiagonal), b per cell.  Built from the terms as the listing writes them:
        every explicit vol-field term F enters b as -(V F) when added on the left, +(V F) when on the right of =="""
        def __init__(self, psi, d, a, b): self.psi, self.d, self.a, self.b = psi, list(d), a, list(b)
        def copy(self): return Eq(self.psi, self.d, self.a, self.b)
        def __add__(self, F): r = self.copy(); r.b = [b - V[i] * f for i, (b, f) in enumerate(zip(r.b, F))]; return r
        def __sub__(self, F):
            if isinstance(F, Eq):       # M - L: matrices subtract coefficient by coefficient (fvm::ddt(U) + ... - fvm::laplacian(gamma, U))
                return Eq(self.psi, [x - y for x, y in zip(self.d, F.d)], self.a - F.a, [x - y for x, y in zip(self.b, F.b)])
            r = self.copy(); r.b = [b + V[i] * f for i, (b, f) in enumerate(zip(r.b, F))]; return r
        def __radd__(self, F): return self.__add__(F)
        def __rsub__(self, F):      # F - M  ==  -(M) + F: the matrix changes sign, F goes to the left-hand side
            r = Eq(self.psi, [-x for x in self.d], -self.a, [b * -1.0 for b in self.b])
            return r.__add__(F)
        def __neg__(self): return Eq(self.psi, [-x for x in self.d], -self.a, [b * -1.0 for b in self.b])
        def __eq__(self, F): r = self.copy(); r.b = [b + V[i] * f for i, (b, f) in enumerate(zip(r.b, F))]; return r
        def setReference(self, cell, value):
            self.b[cell] = self.b[cell] + self.d[cell] * value
            self.d[cell] = self.d[cell] + self.d[cell]
        def solve(self):
            d0, d1, a = self.d[0], self.d[1], self.a
            det = d0 * d1 - a * a
            b0, b1 = self.b
            self.psi.assign([(b0 * d1 + b1 * a) * (1.0 / det), (b1 * d0 + b0 * a) * (1.0 / det)])
        def flux(self):          # lduMatrix face flux: upper psi_N - lower psi_O with upper = lower = -a of M
            return -self.a * self.psi[1] + self.a * self.psi[0]
    for n in range(nfaces):
        nv = 4 if n % 3 != 2 else 3
        pts, own, nei = skew_face(rng, nv)
        S, cf = face_area_centre(pts)
        Sv, Cf = Vec(*S), Vec(*cf)
        magS = mag(Sv)
        V = [1.0, 1.0]
        dvec = nei - own
        nhat = Sv / magS
        delta = 1.0 / max(nhat & dvec, 0.05 * mag(dvec))          # nonOrthDeltaCoeffs (L0)
        U0 = [rnd_vec(rng, 0.3), rnd_vec(rng, 0.3)]
        T0 = [float(rng.uniform(290.0, 310.0)) for _ in range(2)]
        p0 = [float(rng.uniform(-0.1, 0.1)) for _ in range(2)]
        rho0, mu, Pr = float(rng.uniform(0.9, 1.2)), float(10 ** rng.uniform(-3, -1.5)), float(rng.uniform(0.6, 1.0))
        beta, gv = float(rng.uniform(1e-3, 5e-3)), rnd_vec(rng, 9.81)
        Tau, dt = float(10 ** rng.uniform(-3.5, -2)), float(10 ** rng.uniform(-3.5, -2.5))
        ref_cell, ref_val = int(n % 2), float(rng.uniform(-0.5, 0.5))
        sfo, sfn = abs(Sv & (Cf - own)), abs(Sv & (nei - Cf))
        w = sfn / (sfo + sfn)

        def lin(qf):
            a, b = (qf.o, qf.n) if isinstance(qf, Pair) else (qf[0], qf[1])
            return w * (a - b) + b
        cen = [own, nei]
        gU, _ = text.grad(pts, own, nei, U0, [inv_dist(x, cen, U0) for x in pts], True)
        gT, _ = text.grad(pts, own, nei, T0, [inv_dist(x, cen, T0) for x in pts], False)
        grads = dict(U=Tensor(gU), T=Vec(*gT), W=Tensor(np.zeros(9)))

        class TF(QF):       # T*g: a scalar field times a uniform vector
            def __rmul__(self, sc): return TF([sc * a for a in self])
            def __mul__(self, v): return QF([a * v for a in self]) if isinstance(v, Vec) else QF.__mul__(self, v)
        U, T, pfld = QF(U0), TF(T0), QF(p0)
        W = QF([Vec(0, 0, 0), Vec(0, 0, 0)])
        one = QF([1.0, 1.0])
        env = dict(qgdInterpolate=lin, fvsc=Obj(grad=lambda fld: grads["U" if fld is U else ("T" if fld is T else "W")]),
                   U=U, T=T, W=W, rho=QF([rho0, rho0]), beta=beta, g=gv,
                   turbulence=Obj(muEff=call(mu * one)), thermo=Obj(alpha=call((mu / Pr) * one), Cp=call(one)))
        exec(fields_src, env)
        env2 = dict(mesh=Obj(Sf=call(Sv)), Uf=env["Uf"], gradUf=env["gradUf"], BdFrcf=env["BdFrcf"], tauQGDf=Tau, rhof=env["rhof"])
        exec(flux_src, env2)
        # the equations: fvm / fvc on the two-cell mesh (L0 semantics, see the docstring)
        def surf_int(flux): return QF([flux * (1.0 / V[0]), flux * (-1.0 / V[1])])

        def fvc_grad(fld):
            ff = lin(fld)
            return surf_int(Sv * ff)

        def fvc_lap(gam, fld): return surf_int((gam * magS * delta) * (fld[1] - fld[0]))
        fvm = Obj(ddt=lambda fld: Eq(fld, [V[0] / dt, V[1] / dt], 0.0, [V[0] * o * (1.0 / dt) for o in fld.old][:1] + [V[1] * fld.old[1] * (1.0 / dt)]),
                  laplacian=lambda gam, fld: Eq(fld, [-(gam * magS * delta)] * 2, -(gam * magS * delta), [fld[0] * 0.0, fld[1] * 0.0]))
        fvc = Obj(div=surf_int, grad=fvc_grad, laplacian=fvc_lap)
        pe = dict(p=pfld, fvc=fvc, fvm=fvm, phiu=env2["phiu"], phiwo=env2["phiwo"], taubyrhof=env2["taubyrhof"], pRefCell=ref_cell,
                  getRefCellValue=lambda fld, c: fld[c])
        exec(peqn_src, pe)
        phi = pe["phi"]
        gP, _ = text.grad(pts, own, nei, list(pfld), [inv_dist(x, cen, list(pfld)) for x in pts], False)
        ue = dict(env2, fvsc=Obj(grad=lambda fld: Vec(*gP)), p=pfld, phi=phi, U=U, qgdFlux=lambda flux, psi, psif: flux * psif, implicitDiffusion=bool(implicit),
                  fvm=fvm, fvc=fvc, solve=lambda M: M.solve(), muf=env["muf"], rho=QF([rho0, rho0]), BdFrc=env["BdFrc"],
                  USu=QF([Vec(0, 0, 0), Vec(0, 0, 0)]), qgdInterpolate=lin, Foam=Obj(T=lambda fld: QF([t.T() for t in fld])))
        exec(ueqn_a, ue)
        Uold = list(U)
        exec(ueqn_b, ue)
        te = dict(ue, T=T, Tf=env["Tf"], gradTf=env["gradTf"], Hif=env["Hif"], TSu=QF([0.0, 0.0]), max=lambda f: Obj(value=call(max(f))),
                  min=lambda f: Obj(value=call(min(f))), Info=Stream(), endl=None, U=QF(Uold))
        exec(teqn_a, te)
        exec(teqn_b, te)
        exec(ref_src, dict(p=pfld, dimensionedScalar=lambda nm, dims, v: v, pRefValue=ref_val, pRefCell=ref_cell, getRefCellValue=lambda fld, c: fld[c]))
        out = dict(nv=nv, pts=np.array([q_.c for q_ in pts] + ([[0, 0, 0]] if nv == 3 else [])), Sf=S, Cf=cf, C=np.array([own.c, nei.c]),
                   U=np.array([u.c for u in U0]), T=T0, p=p0, rho0=rho0, mu=mu, Pr=Pr, beta=beta, g=gv.c, Tau=Tau, deltaT=dt, pRefCell=ref_cell,
                   pRefValue=ref_val, delta=delta, p1=list(pfld), phi1=phi, U1=np.array([u.c for u in U]), T1=list(T), phiu=env2["phiu"],
                   phiwo=env2["phiwo"])
        for k in names:
            rec[k].append(np.array(out[k], dtype=float))
    return {k: np.array(v) for k, v in rec.items()}


def specieseqn(nfaces=24, seed=27, implicit=False):
    """implicit: QGDYEqn.H L40-66, L86-92 (the implicitDiffusion branch: fvScalarMatrix YEqn(fvm::ddt(rho,Yi) + fvc::div(phiJmYi) -
    fvm::laplacian(muf/ScNumbers[i],Yi) == ...), YEqn.solve(), diffusiveFlux[i] += YEqn.flux()) with fv_emulation_implicit; otherwise
    QGDYEqn.H L40-45, L69-92 (the explicit branch of the species loop) executed as listed on the two-cell mesh for three species, the
    last one inert: solve(fvm::ddt(rho,Yi) + fvc::div(phiJmYi) - fvc::laplacian(muf/ScNumbers[i],Yi) == combustion->R(Yi) + parcels.SYi(i,Yi)),
    diffusiveFlux[i] += (muf/ScNumbers[i]) fvc::snGrad(Yi.oldTime()) magSf, diffusiveFlux[inert] -= diffusiveFlux[i], Yi.max(0), Yt,
    Y[inert] = 1 - Yt.  fvm / fvc are the two-cell emulations (L0: Euler ddt, surfaceIntegrate, Gauss laplacian with the uncorrected
    snGrad); combustion->R and parcels.SYi return explicit source fields (their sum is what qgd_species_step takes as Su)."""
    ye = listing("QGDYEqn_8H_source.html")
    txt = [ye[i] for i in list(range(40, 46)) + (list(range(47, 67)) if implicit else list(range(69, 84))) + list(range(86, 93))]
    src = transpile(txt)
    rng = np.random.default_rng(seed)
    names = ("nv", "pts", "Sf", "Cf", "C", "delta", "rhoOld", "rho", "Y", "phiJmY", "muf", "Sc", "Su", "deltaT", "inertIndex", "diffusiveFlux0",
             "Ynew", "diffusiveFlux1")
    rec = {k: [] for k in names}
    for n in range(nfaces):
        nv = 4 if n % 3 != 2 else 3
        pts, own, nei = skew_face(rng, nv)
        S, cf = face_area_centre(pts)
        S, Cf = Vec(*S), Vec(*cf)
        dvec = nei - own
        delta = 1.0 / max((S / mag(S)) & dvec, 0.05 * mag(dvec))                       # nonOrthDeltaCoeffs (L0)
        dt = float(10.0 ** rng.uniform(-3, -2))
        V = [1.0, 1.0]
        rho_old = [float(rng.uniform(0.8, 1.2)) for _ in range(2)]
        rho_new = [float(r * rng.uniform(0.97, 1.03)) for r in rho_old]
        ns, inert = 3, 2
        Y0 = [[float(rng.uniform(0.0, 0.5)) for _ in range(2)] for _ in range(ns)]
        if n % 4 == 1:
            Y0[0][0] = 1e-4                                                             # a value the step drives below zero: Yi.max(0.0)
        jm = [float(0.3 * rng.standard_normal()) for _ in range(ns)]
        if n % 4 == 1:
            jm[0] = 5.0 * abs(jm[0]) + 1.0
        muf = float(rng.uniform(1e-3, 1e-1))
        Sc = [float(rng.uniform(0.5, 1.5)) for _ in range(ns)]
        Su = [[float(0.2 * rng.standard_normal()) for _ in range(2)] for _ in range(ns)]
        df0 = [float(0.1 * rng.standard_normal()) for _ in range(ns)]
        fvm, fvc = fv_emulation_implicit(dt, mag(S), delta) if implicit else fv_emulation(dt, V)
        fvc.laplacian = lambda gam, psi: CF([gam * mag(S) * delta * (psi[1] - psi[0]) / V[0], -(gam * mag(S) * delta * (psi[1] - psi[0])) / V[1]])
        fvc.snGrad = lambda psi: delta * (psi[1] - psi[0])
        Y = [CF(list(y)) for y in Y0]
        rho = CF(rho_new, old=rho_old)
        half = [[0.5 * x for x in su] for su in Su]                                      # R and SYi: two halves of the explicit source
        env = dict(Y=Y, phiJmY=list(jm), inertIndex=inert, composition=Obj(active=lambda i: True), fvm=fvm, fvc=fvc, rho=rho, muf=muf,
                   ScNumbers=Sc, combustion=Obj(R=lambda Yi: CF(half[[id(y) for y in Y].index(id(Yi))])),
                   parcels=Obj(SYi=lambda i, Yi: CF(half[i])), diffusiveFlux=list(df0), mesh=Obj(magSf=call(mag(S))), Yt=CF([0.0, 0.0]),
                   solve=lambda M: M.solve(), scalar=float, implicitDiffusion=bool(implicit))
        exec(src, env)
        out = dict(nv=nv, pts=np.array([q.c for q in pts] + ([[0, 0, 0]] if nv == 3 else [])), Sf=S.c, Cf=Cf.c, C=np.array([own.c, nei.c]), delta=delta,
                   rhoOld=rho_old, rho=rho_new, Y=Y0, phiJmY=jm, muf=muf, Sc=Sc, Su=Su, deltaT=dt, inertIndex=inert, diffusiveFlux0=df0,
                   Ynew=[list(y) for y in env["Y"]], diffusiveFlux1=env["diffusiveFlux"])
        for k in names:
            rec[k].append(np.array(out[k], dtype=float))
    return {k: np.array(v) for k, v in rec.items()}


class LabelList(list):
    """OpenFOAM's labelList as the stencil search uses it: append(label) and append(list) (List::append(const UList&))"""
    def append(self, x):
        if isinstance(x, list):
            self.extend(x)
        else:
            list.append(self, x)


def lsqorder():
    """The stencil search of the leastSquares scheme executed as listed [extendedFaceStencilFindNeighbours.C L48-84] on whole small 2-D
    meshes: for every internal face the cells around its points, the face's points in order, each point's cells in pointCells() order
    (L0: ascending cell label, primitiveMesh::calcPointCells), the first occurrence kept.  The order is the summation order of the weights
    [CalcW.C L64-153] and of the gradient [ScalarGrad.C L66-72]."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    sys.path.insert(0, os.path.dirname(HERE))
    from util import make_mesh
    src = transpile(lines("extendedFaceStencilFindNeighbours_8C_source.html", 48, 84))
    out = {}
    for kind in ("plane2d_jitter", "step2d", "plane2d"):
        mesh = make_mesh(kind)
        fo, fp = mesh.array("faceOffsets"), mesh.array("facePoints")
        own, nei, nif = mesh.array("owner"), mesh.array("neighbour"), mesh.nInternalFaces
        faces = [LabelList(int(p) for p in fp[fo[f]:fo[f + 1]]) for f in range(mesh.nFaces)]
        cell_faces = [[] for _ in range(mesh.nCells)]
        for f in range(mesh.nFaces):
            cell_faces[own[f]].append(f)
            if f < nif:
                cell_faces[nei[f]].append(f)
        point_cells = [LabelList() for _ in range(mesh.nPoints)]
        for c in range(mesh.nCells):                       # L0 calcPointCells: cells ascending, a cell once per point
            for f in cell_faces[c]:
                for p in faces[f]:
                    if c not in point_cells[p]:
                        point_cells[p].append(c)

        class CMesh:
            def isInternalFace(self, f): return f < nif
            def pointCells(self): return point_cells
        env = dict(faces=faces, cMesh_=CMesh(), neighbourCellsForFace=[LabelList() for _ in range(nif)], LabelList=LabelList, true=True, false=False)
        exec(src, env)
        lists = env["neighbourCellsForFace"]
        out[kind + "_off"] = np.cumsum([0] + [len(x) for x in lists]).astype(np.int64)
        out[kind + "_cells"] = np.array([c for x in lists for c in x], dtype=np.int64)
    return out


def qhdflux():
    """qhdFluxFvPatchScalarField::updateCoeffs L193-203 executed as listed on the wall faces of a small buoyant cavity: the fixed gradient
    of p from the registered flux, gradient = -(phiwStar_b / tauQGDf_b * rhof_b / |Sf|), then fixedGradient's evaluate (L0: patch value =
    cell value + gradient / deltaCoeffs).  The state is the CPU restatement's after three steps (mulesQHDFoam-style walls fed by the
    registered flux, QGD_BC_QHDFLUX); what is pinned is that its patch pressure IS this formula of its own flux, and the device's."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    sys.path.insert(0, os.path.dirname(HERE))
    from oracle import OracleQhdCase
    from test_qhd_case import cavity_bcs, initial, options
    from util import make_mesh, oracle_mesh_of
    bc_l = listing("qhdFluxFvPatchScalarField_8C_source.html")
    txt = [bc_l[i] for i in range(193, 204)]
    txt = [t.replace("phiws.boundaryField()[patch().index()]", "phiws_b").replace("tauQGDf.boundaryField()[patch().index()]", "tauQGDf_b")
            .replace("rhof.boundaryField()[patch().index()]", "rhof_b").replace("patch().magSf()", "magSf_b").replace("this->gradient()", "gradient_")
           for t in txt]
    src = transpile(txt)
    mesh = make_mesh("box654_jitter")
    oc = OracleQhdCase(oracle_mesh_of(mesh), options(deltaT=1e-3))
    cavity_bcs(oc, mesh)
    U, T, p = initial(mesh)
    U = U + 1e-2 * np.random.default_rng(3).standard_normal(U.shape)
    oc.set_fields(U, T, p)
    oc.step(3)
    nif = mesh.nInternalFaces
    own = mesh.array("owner")[nif:]
    env = dict(phiws_b=Fld(list(oc.field("phiwo")[nif:])), tauQGDf_b=Fld(list(oc.field("tauQGDf")[nif:])),
               rhof_b=Fld([1.0] * mesh.nBoundaryFaces), magSf_b=Fld(list(mesh.array("magSf")[nif:])), Fld=Fld)
    exec(src, env)
    grad = np.array(list(env["gradient_"]), dtype=float)
    delta = mesh.array("deltaCoeffs")[nif:]
    pc = oc.field("p")[own]
    return dict(steps=np.array(3), gradient=grad, pb=pc + grad / delta, phiwo_b=oc.field("phiwo")[nif:], tau_b=oc.field("tauQGDf")[nif:])


def main():
    if not os.path.isdir(REF):
        sys.exit("make_ref_expr.py needs the reference listings under /root/reference (build container only)")
    for name, fn in (("gvp3d", gvp3d), ("gvp3d_bnd", gvp3d_bnd), ("gvp2d", gvp2d), ("gvp2d_bnd", gvp2d_bnd), ("reduced", reduced), ("lsq_bnd", lsq_bnd), ("lsq", lsq), ("case2cell", case2cell), ("qhdface", qhdface), ("species", species)):
        data = fn()
        path = os.path.join(HERE, f"ref_expr_{name}.npz")
        np.savez_compressed(path, **data)
        print(name, {k: v.shape for k, v in data.items()})
        if name == "case2cell":
            case = data
            impl = {k: np.array(v) for k, v in case2cell.implicit_step.items()}
            for k in ("nv", "pts", "Sf", "Cf", "C", "U", "T", "p", "R", "Cv", "mu", "Pr", "ScQGD", "PrQGD", "alphaQGD", "deltaT"):
                impl[k] = data[k]
            np.savez_compressed(os.path.join(HERE, "ref_expr_implicit2cell.npz"), **impl)
            print("implicit2cell", {k: v.shape for k, v in impl.items()})
    for name, data in (("gvp2d_vec", gvp2d_vec()), ("gvp_other", gvp_other()), ("qgdlength", qgdlength()), ("courant", courant(case)),
                       ("thermo2cell", thermo2cell(case)),
                       ("qhdclosure", qhdclosure()), ("casebnd", casebnd()), ("qhdeqn", qhdeqn()), ("qhdeqn_implicit", qhdeqn(seed=29, implicit=True)), ("lsqorder", lsqorder()), ("qhdflux", qhdflux()), ("specieseqn", specieseqn()),
                       ("specieseqn_implicit", specieseqn(seed=31, implicit=True))):
        np.savez_compressed(os.path.join(HERE, f"ref_expr_{name}.npz"), **data)
        print(name, {k: getattr(v, "shape", None) for k, v in data.items()})


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--show":
        f, a, b = sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
        print(transpile(lines(f, a, b), continuation="--macro" in sys.argv))
    else:
        main()
