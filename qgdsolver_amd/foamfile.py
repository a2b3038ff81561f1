"""OpenFOAM ASCII case files <-> the flat arrays of the C-ABI.

Readers and writers for ``constant/polyMesh/{points,faces,owner,neighbour,boundary}``,
``vol{Scalar,Vector}Field`` files of a time directory, and the handful of dictionary
entries QGDFoam reads (QGDFoam/createFields.H, thermophysicalProperties, fvSchemes
``fvsc`` [fvsc.C L47-58], controlDict), so that a case prepared for the reference
solver can be loaded into a ``QGDFoamCase`` and its result written back.

Only the ASCII stream format is handled (``format binary`` is refused loudly).
The file grammar is OpenFOAM's (L0: OpenFOAM itself is not part of the reference
tree); nothing in here is on the timed path.
"""
import os
import re

import numpy as np

from . import _lib as L
from .mesh import PolyMesh

# OpenFOAM v2312 universal gas constant [J/(kmol K)] (L0 assumption: 1000 * physicoChemical::R)
RR = 8314.46261815324

PATCH_TYPES = {
    "patch": L.PATCH_GENERIC, "wall": L.PATCH_GENERIC, "empty": L.PATCH_EMPTY, "symmetryPlane": L.PATCH_SYMMETRYPLANE,
    "symmetry": L.PATCH_SYMMETRY, "wedge": L.PATCH_WEDGE, "cyclic": L.PATCH_CYCLIC,
}
PATCH_WORDS = {L.PATCH_GENERIC: "patch", L.PATCH_EMPTY: "empty", L.PATCH_SYMMETRYPLANE: "symmetryPlane",
               L.PATCH_SYMMETRY: "symmetry", L.PATCH_WEDGE: "wedge", L.PATCH_CYCLIC: "cyclic"}
_CONSTRAINT_BCS = {"empty", "symmetryPlane", "symmetry", "wedge", "cyclic"}


class FoamFileError(ValueError):
    pass


# ---------------------------------------------------------------------------------------------------------------------
# tokenizer / dictionary parser
# ---------------------------------------------------------------------------------------------------------------------
_COMMENT = re.compile(r"//[^\n]*|/\*.*?\*/", re.S)
_TOKEN = re.compile(r'\s*("(?:[^"\\]|\\.)*"|[{}()\[\];]|[^\s{}()\[\];"]+)')
_LIST_HEAD = re.compile(r"(?:List<\s*(\w+)\s*>\s*)?(?<![\w.+-])(\d+)\s*\(")


def _strip_comments(text):
    return _COMMENT.sub(" ", text)


def _number(tok):
    try:
        return int(tok)
    except ValueError:
        try:
            return float(tok)
        except ValueError:
            return tok


_WORD_START = re.compile(r"[A-Za-z_]")
_WORD_END = set(" \t\r\n;{}[]\"")


def _tokenize(text):
    """OpenFOAM's token stream as far as dictionaries need it.  A word that starts with a letter runs on through BALANCED parentheses
    (ISstream::read(word&), L0): `grad(U)`, `div(phiJm,U)`, `interpolate((U*rhoU))` are single keywords; a '(' at the start of a
    token, or after a number (`4(a b c d)`), is punctuation."""
    toks = []
    pos, n = 0, len(text)
    while pos < n:
        m = _TOKEN.match(text, pos)
        if m is None:
            pos += 1
            continue
        tok = m.group(1)
        start, end = m.start(1), m.end()
        if _WORD_START.match(tok) and end < n and text[end] == "(":
            depth, k = 0, end
            while k < n:
                ch = text[k]
                if ch == "(":
                    depth += 1
                elif ch == ")":
                    if depth == 0:
                        break
                    depth -= 1
                elif ch in _WORD_END:
                    break
                k += 1
            if depth == 0:   # balanced: the parentheses belong to the word
                tok, end = text[start:k], k
        toks.append(tok)
        pos = end
    return toks


class _Parser:
    def __init__(self, text, lists):
        self.toks = _tokenize(text)
        self.i = 0
        self.lists = lists

    def peek(self):
        return self.toks[self.i] if self.i < len(self.toks) else None

    def next(self):
        t = self.peek()
        if t is None:
            raise FoamFileError("unexpected end of file")
        self.i += 1
        return t

    def parse_dict_body(self, top=False):
        d = {}
        while True:
            t = self.peek()
            if t is None:
                if top:
                    return d
                raise FoamFileError("missing '}'")
            if t == "}":
                if top:
                    raise FoamFileError("unbalanced '}'")
                self.i += 1
                return d
            key = self.next()
            if key == ";":
                continue
            if key.startswith('"'):
                key = key[1:-1]
            if self.peek() == "{":
                self.i += 1
                d[key] = self.parse_dict_body()
                continue
            vals = []
            while self.peek() != ";":
                if self.peek() is None or self.peek() == "}":
                    raise FoamFileError(f"entry '{key}' is not terminated by ';'")
                vals.append(self.parse_value())
            self.i += 1
            d[key] = vals[0] if len(vals) == 1 else vals
        return d

    def parse_value(self):
        t = self.next()
        if t == "(":
            return self.parse_list()
        if t == "[":
            out = []
            while self.peek() != "]":
                out.append(_number(self.next()))
            self.i += 1
            return ("dimensions", out)
        if t == "{":
            return self.parse_dict_body()
        if t.startswith("@LIST"):
            return self.lists[int(t[5:])]
        if t.startswith('"'):
            return t[1:-1]
        # "N(" and "List<T> N (" prefixes of short in-line lists
        if re.fullmatch(r"\d+", t) and self.peek() == "(":
            self.i += 1
            return self.parse_list()
        if re.fullmatch(r"List<\w+>", t):
            return self.parse_value()
        return _number(t)

    def parse_list(self):
        out = []
        while self.peek() != ")":
            if self.peek() is None:
                raise FoamFileError("missing ')'")
            out.append(self.parse_value())
        self.i += 1
        return out


def _extract_big_lists(text, min_items=64):
    """Replace long numeric lists ``N ( ... )`` by @LISTk placeholders, parsed with numpy."""
    lists = []
    out = []
    pos = 0
    for m in _LIST_HEAD.finditer(text):
        if m.start() < pos:
            continue
        n = int(m.group(2))
        if n < min_items:
            continue
        start = m.end()
        # depth of the items: scalar list "1 2 3", or nested "(x y z)" / "4(a b c d)"
        k = start
        while k < len(text) and text[k].isspace():
            k += 1
        nested = k < len(text) and (text[k] == "(" or re.match(r"\d+\s*\(", text[k:k + 24]) is not None)
        if nested:
            e = re.compile(r"\)\s*\)").search(text, start)
            if e is None:
                raise FoamFileError("unterminated list")
            end = e.end() - 1
        else:
            end = text.find(")", start)
            if end < 0:
                raise FoamFileError("unterminated list")
        body = text[start:end]
        ragged = nested and text[k] != "("
        try:
            flat = np.array(body.replace("(", " ").replace(")", " ").split(), dtype=np.float64)
        except ValueError:
            continue  # not a numeric list (e.g. a boundary file with 64 or more patches): left to the token parser
        if ragged:
            lists.append(("ragged", n, flat))
        elif nested:
            if flat.size % n:
                raise FoamFileError("list length does not match its header")
            lists.append(flat.reshape(n, flat.size // n))
        else:
            if flat.size != n:
                raise FoamFileError(f"list of {flat.size} items announced as {n}")
            lists.append(flat)
        out.append(text[pos:m.start()])
        out.append(f" @LIST{len(lists) - 1} ")
        pos = end + 1
    out.append(text[pos:])
    return "".join(out), lists


def parse_foam_text(text):
    """Parse an OpenFOAM dictionary-style file into nested dicts (lists -> python lists / numpy arrays)."""
    text = _strip_comments(text)
    text, lists = _extract_big_lists(text)
    p = _Parser(text, lists)
    d = p.parse_dict_body(top=True)
    hdr = d.get("FoamFile", {})
    if isinstance(hdr, dict) and str(hdr.get("format", "ascii")) != "ascii":
        raise FoamFileError("only 'format ascii' files are supported")
    return d


def _read_text(path):
    """text of a case file; OpenFOAM's writeCompression leaves <name>.gz instead of <name>"""
    if not os.path.exists(path) and os.path.exists(path + ".gz"):
        import gzip
        with gzip.open(path + ".gz", "rt") as f:
            return f.read()
    with open(path) as f:
        return f.read()


def _exists(path):
    return os.path.exists(path) or os.path.exists(path + ".gz")


def read_dict(path):
    return parse_foam_text(_read_text(path))


def _split_header(text):
    """(FoamFile dict, rest of the text) of a list-style file (points, faces, owner, ...)"""
    text = _strip_comments(text)
    m = re.search(r"FoamFile\s*\{", text)
    hdr = {}
    if m:
        end = text.index("}", m.end())
        hdr = _Parser(text[m.end():end], []).parse_dict_body(top=True)
        text = text[end + 1:]
    if str(hdr.get("format", "ascii")) != "ascii":
        raise FoamFileError("only 'format ascii' files are supported")
    return hdr, text


def _read_list_file(path):
    hdr, text = _split_header(_read_text(path))
    m = _LIST_HEAD.search(text)
    if m is None:
        raise FoamFileError(f"{path}: no list found")
    n = int(m.group(2))
    body = text[m.end():text.rindex(")")]
    flat = np.array(body.replace("(", " ").replace(")", " ").split(), dtype=np.float64)
    return hdr, n, flat


# ---------------------------------------------------------------------------------------------------------------------
# polyMesh
# ---------------------------------------------------------------------------------------------------------------------
def read_polymesh(poly_dir):
    """constant/polyMesh -> PolyMesh (with ``patch_names``).  processor patches are refused: a decomposed case is
    loaded whole and sharded by cell ranges instead (multi-GPU)."""
    _, npnt, flat = _read_list_file(os.path.join(poly_dir, "points"))
    if flat.size != 3 * npnt:
        raise FoamFileError("points: size mismatch")
    points = flat
    _, nf, flat = _read_list_file(os.path.join(poly_dir, "faces"))
    ints = flat.astype(np.int64)
    if ints.size == 5 * nf and np.all(ints[0::5] == 4):
        sizes = np.full(nf, 4, dtype=np.int64)
        fp = ints.reshape(nf, 5)[:, 1:].reshape(-1)
    else:
        sizes = np.empty(nf, dtype=np.int64)
        keep = np.ones(ints.size, dtype=bool)
        k = 0
        for f in range(nf):
            s = ints[k]
            sizes[f] = s
            keep[k] = False
            k += s + 1
        if k != ints.size:
            raise FoamFileError("faces: size mismatch")
        fp = ints[keep]
    fo = np.zeros(nf + 1, dtype=np.int64)
    np.cumsum(sizes, out=fo[1:])
    hdr_o, no, owner = _read_list_file(os.path.join(poly_dir, "owner"))
    _, nn, neighbour = _read_list_file(os.path.join(poly_dir, "neighbour"))
    if no != nf or owner.size != nf or neighbour.size != nn:
        raise FoamFileError("owner/neighbour: size mismatch")
    owner = owner.astype(np.int32)
    neighbour = neighbour.astype(np.int32)
    n_cells = int(owner.max()) + 1 if nf else 0
    note = str(hdr_o.get("note", ""))
    m = re.search(r"nCells:\s*(\d+)", note)
    if m:
        n_cells = int(m.group(1))

    _, text = _split_header(_read_text(os.path.join(poly_dir, "boundary")))
    m = re.search(r"(\d+)\s*\(", text)
    if m is None:
        raise FoamFileError("boundary: no patch list")
    body = text[m.end():text.rindex(")")]
    pd = _Parser(body, []).parse_dict_body(top=True)
    if len(pd) != int(m.group(1)):
        raise FoamFileError("boundary: patch count mismatch")
    names, start, size, ptype, partner = [], [], [], [], {}
    for name, e in pd.items():
        t = str(e.get("type", "patch"))
        if t not in PATCH_TYPES:
            raise FoamFileError(f"boundary: patch '{name}' of type '{t}' is not supported")
        names.append(name)
        start.append(int(e["startFace"]))
        size.append(int(e["nFaces"]))
        ptype.append(PATCH_TYPES[t])
        if t == "cyclic" and "neighbourPatch" in e:
            partner[name] = str(e["neighbourPatch"])
            if str(e.get("transform", "unknown")) == "rotational":
                raise FoamFileError(f"boundary: cyclic patch '{name}' is rotational: only translational cyclic pairs are served")
    mesh = PolyMesh.from_arrays(points, fo, fp, owner, neighbour, n_cells, start, size, ptype)
    mesh.patch_names = names
    # cyclic pairs (patch index of a half, of its neighbourPatch), each pair once, in patch order: what PolyMesh.unroll_cyclic takes
    mesh.cyclic_pairs = []
    for i, name in enumerate(names):
        if name in partner:
            other = partner[name]
            if other not in names or partner.get(other) != name:
                raise FoamFileError(f"boundary: cyclic patch '{name}' names neighbourPatch '{other}', which does not name it back")
            j = names.index(other)
            if i < j:
                mesh.cyclic_pairs.append((i, j))
    # the faceSet the leastSquares stencil reads if present [leastSquaresStencil.C L63-70]
    dsf = os.path.join(poly_dir, "sets", "degenerateStencilFaces")
    if _exists(dsf):
        _, nset, labels = _read_list_file(dsf)
        if labels.size != nset:
            raise FoamFileError("sets/degenerateStencilFaces: size mismatch")
        mesh.set_degenerate_faces(labels.astype(np.int32))
    return mesh


def _header(cls, obj, location, note=None):
    lines = ["FoamFile", "{", "    version     2.0;", "    format      ascii;", f"    class       {cls};"]
    if note:
        lines.append(f'    note        "{note}";')
    lines += [f'    location    "{location}";', f"    object      {obj};", "}", ""]
    return "\n".join(lines) + "\n"


def _fmt(x):
    return repr(float(x))


def write_polymesh(mesh, poly_dir, patch_names=None):
    """PolyMesh -> constant/polyMesh (ASCII, full-precision reals so a round trip is bit-exact)."""
    os.makedirs(poly_dir, exist_ok=True)
    names = patch_names or getattr(mesh, "patch_names", None) or [f"patch{i}" for i in range(mesh.nPatches)]
    pts = mesh.array("points").reshape(-1, 3)
    fo = mesh.array("faceOffsets")
    fp = mesh.array("facePoints")
    owner = mesh.array("owner")
    neighbour = mesh.array("neighbour")
    note = f"nPoints:{mesh.nPoints}  nCells:{mesh.nCells}  nFaces:{mesh.nFaces}  nInternalFaces:{mesh.nInternalFaces}"
    with open(os.path.join(poly_dir, "points"), "w") as f:
        f.write(_header("vectorField", "points", "constant/polyMesh"))
        f.write(f"{mesh.nPoints}\n(\n")
        f.write("\n".join(f"({_fmt(p[0])} {_fmt(p[1])} {_fmt(p[2])})" for p in pts))
        f.write("\n)\n")
    with open(os.path.join(poly_dir, "faces"), "w") as f:
        f.write(_header("faceList", "faces", "constant/polyMesh"))
        f.write(f"{mesh.nFaces}\n(\n")
        f.write("\n".join(f"{fo[i + 1] - fo[i]}({' '.join(map(str, fp[fo[i]:fo[i + 1]]))})" for i in range(mesh.nFaces)))
        f.write("\n)\n")
    for name, arr in (("owner", owner), ("neighbour", neighbour)):
        with open(os.path.join(poly_dir, name), "w") as f:
            f.write(_header("labelList", name, "constant/polyMesh", note))
            f.write(f"{arr.size}\n(\n")
            f.write("\n".join(map(str, arr)))
            f.write("\n)\n")
    ps, pz, pt = mesh.array("patchStart"), mesh.array("patchSize"), mesh.array("patchType")
    with open(os.path.join(poly_dir, "boundary"), "w") as f:
        f.write(_header("polyBoundaryMesh", "boundary", "constant/polyMesh"))
        f.write(f"{mesh.nPatches}\n(\n")
        for i in range(mesh.nPatches):
            if int(pt[i]) not in PATCH_WORDS:
                raise FoamFileError("halo (cell-range shard) patches have no polyMesh spelling")
            nbr = ""
            for pa, pb in getattr(mesh, "cyclic_pairs", []):
                if i in (pa, pb):
                    nbr = f"        neighbourPatch  {names[pb if i == pa else pa]};\n"
            f.write(f"    {names[i]}\n    {{\n        type            {PATCH_WORDS[int(pt[i])]};\n{nbr}"
                    f"        nFaces          {int(pz[i])};\n        startFace       {int(ps[i])};\n    }}\n")
        f.write(")\n")


# ---------------------------------------------------------------------------------------------------------------------
# fields
# ---------------------------------------------------------------------------------------------------------------------
def _field_values(v, n, ncomp, what):
    """'uniform x' / 'nonuniform List<T> N (...)' entry -> (n, ncomp) array"""
    if isinstance(v, list) and len(v) == 2 and v[0] == "uniform":
        val = np.asarray(v[1], dtype=np.float64).reshape(-1)
        if val.size != ncomp:
            raise FoamFileError(f"{what}: uniform value has {val.size} components, expected {ncomp}")
        return np.tile(val, (n, 1))
    if isinstance(v, list) and len(v) >= 2 and v[0] == "nonuniform":
        arr = np.asarray(v[-1], dtype=np.float64)
        if arr.size != n * ncomp:
            raise FoamFileError(f"{what}: {arr.size} values for {n} x {ncomp}")
        return arr.reshape(n, ncomp)
    raise FoamFileError(f"{what}: cannot read value '{v}'")


def read_field(path, mesh):
    """vol<Type>Field file -> (internal (nCells, ncomp) array, {patch name: {'type': word, 'value': array|None, ...}})"""
    d = read_dict(path)
    cls = str(d.get("FoamFile", {}).get("class", ""))
    ncomp = {"volScalarField": 1, "volVectorField": 3}.get(cls)
    if ncomp is None:
        raise FoamFileError(f"{path}: class '{cls}' is not a vol scalar/vector field")
    internal = _field_values(d["internalField"], mesh.nCells, ncomp, f"{path}: internalField")
    names = getattr(mesh, "patch_names", None) or [f"patch{i}" for i in range(mesh.nPatches)]
    sizes = mesh.array("patchSize")
    bf = d.get("boundaryField", {})
    patches = {}
    for i, name in enumerate(names):
        e = bf.get(name)
        if e is None:  # OpenFOAM also accepts regular-expression keys
            for k, v in bf.items():
                try:
                    if re.fullmatch(k, name):
                        e = v
                        break
                except re.error:
                    pass
        if e is None:
            raise FoamFileError(f"{path}: no boundaryField entry for patch '{name}'")
        rec = dict(e)
        rec["type"] = str(e["type"])
        rec["value"] = _field_values(e["value"], int(sizes[i]), ncomp, f"{path}: {name}.value") if "value" in e else None
        if "gradient" in e:   # fixedGradient / qhdFlux patches
            rec["gradient"] = _field_values(e["gradient"], int(sizes[i]), ncomp, f"{path}: {name}.gradient")
        patches[name] = rec
    return internal, patches


def write_field(path, mesh, name, internal, patches, dimensions="[0 0 0 0 0 0 0]"):
    """(nCells[, 3]) array + {patch: (type word, value or None[, {extra entry: scalar}])} -> vol<Type>Field file (values at full
    precision; the optional third member writes further uniform entries of the patch, e.g. the gradient of a fixedGradient patch)"""
    a = np.asarray(internal, dtype=np.float64)
    vec = a.ndim == 2 and a.shape[1] == 3
    cls, typ = ("volVectorField", "vector") if vec else ("volScalarField", "scalar")

    def one(x):
        return f"({_fmt(x[0])} {_fmt(x[1])} {_fmt(x[2])})" if vec else _fmt(x)

    def values(arr):
        arr = np.asarray(arr, dtype=np.float64)
        if arr.ndim == (1 if vec else 0):
            return f"uniform {one(arr)}"
        return f"nonuniform List<{typ}> {len(arr)}\n(\n" + "\n".join(one(x) for x in arr) + "\n)"

    names = getattr(mesh, "patch_names", None) or [f"patch{i}" for i in range(mesh.nPatches)]
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        f.write(_header(cls, name, os.path.basename(os.path.dirname(path))))
        f.write(f"dimensions      {dimensions};\n\ninternalField   {values(a)};\n\nboundaryField\n{{\n")
        for pn in names:
            t, v = patches[pn][0], patches[pn][1]
            f.write(f"    {pn}\n    {{\n        type            {t};\n")
            for key, x in (patches[pn][2].items() if len(patches[pn]) > 2 else ()):
                f.write(f"        {key:<15} uniform {_fmt(float(x))};\n")
            if v is not None:
                f.write(f"        value           {values(v)};\n")
            f.write("    }\n")
        f.write("}\n")


# ---------------------------------------------------------------------------------------------------------------------
# case set-up (what QGDFoam's createFields.H reads)
# ---------------------------------------------------------------------------------------------------------------------
def _bc(rec, vector, patch_type_word, what):
    """boundaryField entry -> the (kind, value) pair QGDFoamCase.set_bc takes.

    Constraint patches (L0: fvPatchField<Type>::New(p, iF, dict)): the entry's type must be the patch's own constraint type --
    OpenFOAM stops with "inconsistent patch and patchField types" otherwise -- and the field then IS that constraint field:
    symmetryPlane / symmetry reflect U (basicSymmetry, the library's ``slip`` arithmetic) and leave scalars zero-gradient; empty
    carries nothing.  cyclic / wedge fields are not served by the resident cases (read_case_setup refuses such meshes)."""
    t = rec["type"]
    if patch_type_word in _CONSTRAINT_BCS:
        if t != patch_type_word:
            raise FoamFileError(f"{what}: inconsistent patch and patchField types: the patch is '{patch_type_word}', the field says '{t}'")
        if patch_type_word == "empty":
            return ("none", None)
        if patch_type_word in ("symmetryPlane", "symmetry"):
            return ("slip", None) if vector else ("zeroGradient", None)
        if patch_type_word == "cyclic":
            return ("none", None)   # the pair's faces become internal faces (PolyMesh.unroll_cyclic): no patch field is left to evaluate
        raise FoamFileError(f"{what}: '{patch_type_word}' patch fields are not supported by the resident cases")
    if t in _CONSTRAINT_BCS:
        raise FoamFileError(f"{what}: boundary condition '{t}' needs a patch of that type (the patch is '{patch_type_word}')")
    if t in ("zeroGradient", "slip", "qgdFlux"):
        if t == "slip" and not vector:
            return ("zeroGradient", None)  # slip on a scalar is zeroGradient (basicSymmetry, L0)
        return (t, None)
    if t == "fixedValue":
        v = rec["value"]
        if v is None:
            raise FoamFileError(f"{what}: fixedValue needs a value")
        if np.any(v != v[0]):
            raise FoamFileError(f"{what}: only uniform fixedValue patches are supported")
        return ("fixedValue", v[0] if vector else float(v[0, 0]))
    raise FoamFileError(f"{what}: boundary condition '{t}' is not supported")


def _refuse_unserved_patches(mesh, case_dir, cyclic_ok=False):
    """wedge patches with faces: their rotated patch fields are not served by the resident cases (the fvsc operators of a Device do serve
    such meshes).  cyclic patches: QGDFoam's explicit branch serves translational pairs named by `neighbourPatch` (cyclic_ok; the
    application unrolls them into ghost cells, PolyMesh.unroll_cyclic); QHDFoam does not."""
    pt, pz = mesh.array("patchType"), mesh.array("patchSize")
    paired = {i for pr in getattr(mesh, "cyclic_pairs", []) for i in pr}
    for i, name in enumerate(mesh.patch_names):
        word = PATCH_WORDS.get(int(pt[i]), "patch")
        if word == "cyclic" and int(pz[i]) > 0 and cyclic_ok:
            if i not in paired:
                raise FoamFileError(f"{case_dir}: cyclic patch '{name}' has no neighbourPatch entry in constant/polyMesh/boundary")
            continue
        if word in ("cyclic", "wedge") and int(pz[i]) > 0:
            raise FoamFileError(f"{case_dir}: patch '{name}' is a {word} patch; the resident QGDFoam / QHDFoam cases do not serve "
                                f"{word} patch fields")


def _scheme_words(v):
    """a scheme entry as the list of its words: `linear;` -> ['linear'], `Gauss linear corrected;` -> ['Gauss','linear','corrected']"""
    return [str(x) for x in v] if isinstance(v, list) else [str(v)]


def _find_entry(d, key):
    """dictionary::found / lookup of OpenFOAM (L0): the literal key first, then the quoted (regular-expression) keys, last one first"""
    if not isinstance(d, dict):
        return None
    if key in d:
        return d[key]
    for k in reversed(list(d.keys())):
        if k == key or not any(ch in k for ch in ".*+?[]|^$\\"):
            continue
        try:
            if re.fullmatch(k, key):
                return d[k]
        except re.error:
            pass
    return None


def read_face_schemes(fs, flux_terms, grad_terms, what="system/fvSchemes"):
    """The fvSchemes sub-dictionaries the face-flux path consults, checked against what the library computes.

    * ``fvsc`` [fvsc.C L47-58]: the word of every term in ``grad_terms`` (its own entry, else ``default``), returned per term.  The
      QGDFoam case serves up to two distinct stencils (``qgd_case_options::termStencil``); what a caller cannot serve it refuses itself.
    * ``interpolationSchemes`` [QGDInterpolate.H L42-66]: a field with an ``interpolate(<name>)`` entry, or any field under a
      ``default`` other than ``none``, goes to ``fvc::interpolate``: with ``linear`` that is linearInterpolate = the numbers the
      library produces; anything else is refused, naming the entry.  The sub-dictionary must exist (``subDict`` is fatal otherwise).
    * ``divSchemes`` [QGDInterpolate.H L86-104]: a flux with an entry of its own name (``div(phiJm,U)`` ...) goes to ``fvc::flux``;
      ``Gauss linear`` is flux * linear(psi) = the default branch, ``Gauss upwind`` the in-register upwind branch of the face kernels
      (options ``fluxScheme*``); limited schemes are refused.  ``default`` is NOT consulted by qgdFlux (``found(fluxName)``).

    Returns ({gradient term: stencil word}, {flux term: 'linear' | 'upwind'})."""
    fvsc = fs.get("fvsc")
    if not isinstance(fvsc, dict):
        raise FoamFileError(f"{what}: sub-dictionary 'fvsc' is missing [fvsc.C L51]")
    words = {}
    for term in grad_terms:
        v = _find_entry(fvsc, term)
        if v is None:
            v = _find_entry(fvsc, "default")
            if v is None:
                raise FoamFileError(f"{what}: fvsc has neither '{term}' nor 'default' [fvsc.C L57]")
        words[term] = _scheme_words(v)[0]
    interp = fs.get("interpolationSchemes")
    if not isinstance(interp, dict):
        raise FoamFileError(f"{what}: sub-dictionary 'interpolationSchemes' is missing (qgdInterpolate looks it up, QGDInterpolate.H L44)")
    for key, v in interp.items():
        w = _scheme_words(v)
        if key == "default":
            if w not in (["none"], ["linear"]):
                raise FoamFileError(f"{what}: interpolationSchemes.default '{' '.join(w)}' is not supported: the face-flux path interpolates "
                                    f"linearly (accepted: none, linear)")
        elif w != ["linear"]:
            raise FoamFileError(f"{what}: interpolationSchemes.{key} '{' '.join(w)}' is not supported: the face-flux path interpolates "
                                f"linearly (accepted: linear)")
    div = fs.get("divSchemes")
    if not isinstance(div, dict):
        raise FoamFileError(f"{what}: sub-dictionary 'divSchemes' is missing (qgdFlux looks it up, QGDInterpolate.H L86)")
    flux = {}
    for term in flux_terms:
        v = _find_entry(div, term)
        if v is None:
            flux[term] = "linear"      # flux*psif [QGDInterpolate.H L104]
            continue
        w = _scheme_words(v)
        if w == ["Gauss", "linear"]:
            flux[term] = "linear"
        elif w == ["Gauss", "upwind"]:
            flux[term] = "upwind"
        else:
            raise FoamFileError(f"{what}: divSchemes.{term} '{' '.join(w)}' is not supported (accepted: Gauss linear, Gauss upwind)")
    return words, flux


def _check_fv_schemes(fs, mesh, need_laplacian, need_grad, what="system/fvSchemes"):
    """ddtSchemes / laplacianSchemes / gradSchemes as far as the path's fvm:: / fvc:: operators read them (L0): Euler; the uncorrected
    Gauss linear laplacian (a corrected one is the same operator on an orthogonal mesh and is accepted there); Gauss linear gradients"""
    ddt = fs.get("ddtSchemes", {})
    w = _scheme_words(ddt.get("default", "Euler")) if isinstance(ddt, dict) else ["Euler"]
    if w != ["Euler"]:
        raise FoamFileError(f"{what}: ddtSchemes.default '{' '.join(w)}' is not supported (the path is the explicit Euler step of the reference)")
    if need_laplacian:
        lap = fs.get("laplacianSchemes", {})
        for key, v in (lap.items() if isinstance(lap, dict) else ()):
            w = _scheme_words(v)
            ok = w[:2] == ["Gauss", "linear"] and len(w) == 3 and w[2] in ("uncorrected", "orthogonal", "corrected")
            if ok and w[2] == "corrected" and not _orthogonal(mesh):
                raise FoamFileError(f"{what}: laplacianSchemes.{key} 'Gauss linear corrected' on a non-orthogonal mesh: the explicit "
                                    f"non-orthogonal correction is not part of the path (accepted: Gauss linear uncorrected)")
            if not ok:
                raise FoamFileError(f"{what}: laplacianSchemes.{key} '{' '.join(w)}' is not supported (accepted: Gauss linear uncorrected)")
    if need_grad:
        gs = fs.get("gradSchemes", {})
        for key, v in (gs.items() if isinstance(gs, dict) else ()):
            w = _scheme_words(v)
            if w != ["Gauss", "linear"]:
                raise FoamFileError(f"{what}: gradSchemes.{key} '{' '.join(w)}' is not supported (accepted: Gauss linear)")


def _orthogonal(mesh, tol=1e-12):
    """every internal face normal parallel to the line of its two cell centres (nonOrthCorrectionVectors = 0 to rounding)"""
    nif = mesh.nInternalFaces
    if nif == 0:
        return True
    Sf = mesh.array("Sf").reshape(-1, 3)[:nif]
    C = mesh.array("C").reshape(-1, 3)
    d = C[mesh.array("neighbour")] - C[mesh.array("owner")[:nif]]
    n = Sf / np.linalg.norm(Sf, axis=1)[:, None]
    corr = n - d / (n * d).sum(axis=1)[:, None]
    return bool(np.abs(corr).max() <= tol)


def read_case_setup(case_dir, time="0"):
    """Read an OpenFOAM QGDFoam case directory.

    Returns (mesh, options dict for ``default_options``, {'U','T','p'} internal arrays, per-patch BC triples).
    Entries read: constant/polyMesh; constant/thermophysicalProperties (mixture.specie.molWeight,
    thermodynamics.Cv|Cp, transport.mu/Pr; QGD{implicitDiffusion, QGDCoeffs, <model>Dict{ScQGD,PrQGD}} as in
    QGDThermo.C L48-82, QGDCoeffs.C L57-160, constScPrModel1.C L48-90); system/fvSchemes fvsc.default (fvsc.C L47-58);
    system/controlDict deltaT/adjustTimeStep/maxCo/maxDeltaT (setDeltaT-QGDQHD.H); system/fvSolution solvers.{U,e}
    tolerance/maxIter (implicitDiffusion only); <time>/{U,T,p}.
    """
    mesh = read_polymesh(os.path.join(case_dir, "constant", "polyMesh"))
    _refuse_unserved_patches(mesh, case_dir, cyclic_ok=True)
    opt = {}
    tp = read_dict(os.path.join(case_dir, "constant", "thermophysicalProperties"))
    tt = tp.get("thermoType", {})
    for key, want in (("equationOfState", "perfectGas"), ("transport", "const")):
        if isinstance(tt, dict) and key in tt and str(tt[key]) != want:
            raise FoamFileError(f"thermoType.{key} '{tt[key]}' is not supported (only {want})")
    mix = tp["mixture"]
    R = RR / float(mix["specie"]["molWeight"])
    th = mix["thermodynamics"]
    opt["R"] = R
    opt["Cv"] = float(th["Cv"]) if "Cv" in th else float(th["Cp"]) - R
    opt["mu"] = float(mix["transport"]["mu"])
    opt["Pr"] = float(mix["transport"]["Pr"])
    qgd = tp["QGD"]
    # QGDThermo::read(): implicitDiffusion defaults to true when absent [QGDThermo.C L70-82]
    opt["implicitDiffusion"] = 1 if str(qgd.get("implicitDiffusion", "true")) in ("true", "on", "yes", "1") else 0
    # not a reference entry: selects the physically consistent explicit energy update (qgd_case_options::consistentEnergy)
    if "consistentEnergy" in qgd:
        opt["consistentEnergy"] = 1 if str(qgd["consistentEnergy"]) in ("true", "on", "yes", "1") else 0
    model = str(qgd["QGDCoeffs"])
    if model != "constScPrModel1":
        raise FoamFileError(f"QGDCoeffs '{model}' is not supported (only constScPrModel1)")
    md = qgd.get(model + "Dict", qgd)  # QGDCoeffs::New: <type>Dict when present, else the QGD dict [QGDCoeffs.C L81-116]
    for k in ("ScQGD", "PrQGD"):       # both default to 1 [constScPrModel1.C L58-89]
        opt[k] = float(md[k]) if k in md else 1.0
    tdir = os.path.join(case_dir, str(time))
    # alphaQGD and ScQGD are READ_IF_PRESENT fields of the time directory [QGDCoeffs.C L119-160, constScPrModel1.C
    # L66-79]: a uniform one becomes the scalar of the options, a non-uniform one travels as (cell values, patch values)
    opt["alphaQGD"] = 0.5
    coeff_fields = {}
    for fname in ("alphaQGD", "ScQGD"):
        fpath = os.path.join(tdir, fname)
        if _exists(fpath):
            vals, bvals = read_field(fpath, mesh)
            patch_vals = _patch_values(mesh, vals[:, 0], bvals, fpath)
            if np.all(vals == vals[0]) and np.all(patch_vals == vals[0, 0]):
                opt[fname] = float(vals[0, 0])
            else:
                coeff_fields[fname] = (vals[:, 0].copy(), patch_vals)
    fs_path = os.path.join(case_dir, "system", "fvSchemes")
    fs = read_dict(fs_path)
    # the four fvsc::grad calls [QGDFoam/updateFluxes.H L41-65] and the two qgdFlux calls [L78, L119]
    words, flux = read_face_schemes(fs, ("div(phiJm,U)", "div(phiJm,H)"), ("grad(U)", "grad(e)", "grad(rho)", "grad(p)"), fs_path)
    # the case's `stencil` = the default word when there is one, else grad(U)'s; terms with another word travel as termStencils
    dflt = _find_entry(fs["fvsc"], "default")
    opt["stencil"] = _scheme_words(dflt)[0] if dflt is not None else words["grad(U)"]
    per_term = {t: w for t, w in words.items() if w != opt["stencil"]}
    if len(set(words.values())) > 2:
        raise FoamFileError(f"{fs_path}: fvsc gives the four face gradients more than two distinct stencils ({words}); the resident case "
                            f"serves at most two")
    if per_term:
        opt["termStencils"] = per_term
    opt["fluxSchemeU"] = 1 if flux["div(phiJm,U)"] == "upwind" else 0
    opt["fluxSchemeH"] = 1 if flux["div(phiJm,H)"] == "upwind" else 0
    _check_fv_schemes(fs, mesh, need_laplacian=bool(opt["implicitDiffusion"]), need_grad=bool(opt["implicitDiffusion"]), what=fs_path)
    cd = read_dict(os.path.join(case_dir, "system", "controlDict"))
    opt["deltaT"] = float(cd["deltaT"])
    opt["adjustTimeStep"] = 1 if str(cd.get("adjustTimeStep", "no")) in ("yes", "on", "true", "1") else 0
    # createTimeControls.H (L0): maxCo defaults to 1, maxDeltaT to GREAT; setDeltaT-QGDQHD.H L45: cTau defaults to 0.75
    opt["maxCo"] = float(cd.get("maxCo", 1.0))
    opt["maxDeltaT"] = float(cd.get("maxDeltaT", 1e300))
    opt["cTau"] = float(cd.get("cTau", 0.75))

    # the linear solves of the implicitDiffusion branch take fvSolution's controls of U and e [QGDUEqn.H L66, QGDEEqn.H
    # L61: solve(...) looks up solvers.<field>]; the tighter of the two tolerances and the larger maxIter serve both
    # (qgd_case_options has one pair).  OpenFOAM's defaults when an entry is absent: tolerance 1e-6, maxIter 1000.
    fsol_path = os.path.join(case_dir, "system", "fvSolution")
    if opt["implicitDiffusion"] and _exists(fsol_path):
        solvers = read_dict(fsol_path).get("solvers", {})
        tols, iters = [], []
        for key, entry in solvers.items() if isinstance(solvers, dict) else ():
            names = str(key).strip('"()').replace("|", " ").split()
            if isinstance(entry, dict) and any(n in ("U", "e", "Ux", "Uy", "Uz") or n.startswith(("U.", "e.")) for n in names):
                tols.append(float(entry.get("tolerance", 1e-6)))
                iters.append(int(float(entry.get("maxIter", 1000))))
        if tols:
            opt["implicitTol"] = min(tols)
            opt["implicitMaxIter"] = max(iters)

    U, bU = read_field(os.path.join(tdir, "U"), mesh)
    T, bT = read_field(os.path.join(tdir, "T"), mesh)
    p, bP = read_field(os.path.join(tdir, "p"), mesh)
    ptw = [PATCH_WORDS.get(int(t), "patch") for t in mesh.array("patchType")]
    bcs = []
    for i, name in enumerate(mesh.patch_names):
        bcs.append({"U": _bc(bU[name], True, ptw[i], f"U.{name}"), "T": _bc(bT[name], False, ptw[i], f"T.{name}"),
                    "p": _bc(bP[name], False, ptw[i], f"p.{name}")})
    fields = {"U": U, "T": T[:, 0], "p": p[:, 0]}
    fields.update(coeff_fields)
    return mesh, opt, fields, bcs


def _truthy(v):
    return str(v) in ("true", "on", "yes", "1")


def _bc_p_qhd(rec, what):
    """p of a QHDFoam case: zeroGradient | fixedValue (uniform) | fixedGradient (uniform gradient) | qhdFlux.  Inside QHDFoam a
    qhdFlux patch keeps the gradient of its file -- its registry lookup of "phiwStar" finds nothing, the solver registers its flux
    as "phiwo" [qhdFluxFvPatchScalarField.C L166-168] -- so it is read as the fixedGradient patch it behaves like (gradient
    entry, default 0)."""
    t = rec["type"]
    if t in _CONSTRAINT_BCS:
        return ("none", None)
    if t == "zeroGradient":
        return ("zeroGradient", None)
    if t == "fixedValue":
        v = rec["value"]
        if v is None or np.any(v != v[0]):
            raise FoamFileError(f"{what}: fixedValue needs a uniform value")
        return ("fixedValue", float(v[0, 0]))
    if t in ("fixedGradient", "qhdFlux"):
        g = rec.get("gradient")
        if g is None:
            return ("fixedGradient", 0.0)
        g = np.asarray(g, dtype=np.float64).reshape(-1)
        if np.any(g != g[0]):
            raise FoamFileError(f"{what}: only uniform gradients are supported")
        return ("fixedGradient", float(g[0]))
    raise FoamFileError(f"{what}: boundary condition '{t}' is not supported")


def read_qhd_case_setup(case_dir, time="0"):
    """Read an OpenFOAM QHDFoam case directory [QHDFoam.C L63-72, QHDFoam/createFields.H L32-167].

    Returns (mesh, options dict for ``qhdfoam.qhd_options``, {'U','T','p'} internal arrays, per-patch BC triples).
    Entries read: constant/polyMesh; constant/thermophysicalProperties -- thermoType (equationOfState rhoConst, transport const),
    mixture.equationOfState.rho, mixture.transport.{mu, Pr, beta} [createFields.H L110-115], QGD{implicitDiffusion (default true,
    QGDThermo.C L70-82), QGDCoeffs constTau | HbyUQHD | T0byGr | H2bynuQHD with the keys of <model>Dict or of QGD itself
    [QGDCoeffs.C L81-116; constTau.C L71, HbyUQHD.C L61, T0byGr.C L74-75], pRefCell, pRefValue [createFields.H L162-165]};
    constant/gravitationalProperties g [createFields.H L96-108]; <time>/alphaQGD when present and uniform [QGDCoeffs.C L119-160,
    default 0.5]; system/fvSchemes fvsc.default; system/controlDict deltaT; system/fvSolution solvers.p {tolerance, relTol,
    maxIter} and, with implicitDiffusion, solvers.(U|T) {tolerance, maxIter}; <time>/{U,T,p}."""
    mesh = read_polymesh(os.path.join(case_dir, "constant", "polyMesh"))
    _refuse_unserved_patches(mesh, case_dir)
    opt = {}
    tp = read_dict(os.path.join(case_dir, "constant", "thermophysicalProperties"))
    tt = tp.get("thermoType", {})
    for key, want in (("equationOfState", "rhoConst"), ("transport", "const")):
        if isinstance(tt, dict) and key in tt and str(tt[key]) != want:
            raise FoamFileError(f"thermoType.{key} '{tt[key]}' is not supported by the QHDFoam path (only {want})")
    mix = tp["mixture"]
    opt["rho0"] = float(mix["equationOfState"]["rho"])
    tr = mix["transport"]
    opt["mu"], opt["Pr"], opt["beta"] = float(tr["mu"]), float(tr["Pr"]), float(tr["beta"])
    qgd = tp["QGD"]
    opt["implicitDiffusion"] = 1 if _truthy(qgd.get("implicitDiffusion", "true")) else 0
    model = str(qgd["QGDCoeffs"])
    if model not in ("constTau", "HbyUQHD", "T0byGr", "H2bynuQHD"):
        raise FoamFileError(f"QGDCoeffs '{model}' is not a closure of the QHDFoam path (constTau, HbyUQHD, T0byGr, H2bynuQHD)")
    opt["tauModel"] = model
    md = qgd.get(model + "Dict", qgd)
    for key, needed in (("Tau", model == "constTau"), ("UQHD", model == "HbyUQHD"), ("T0", model == "T0byGr"), ("Gr", model == "T0byGr")):
        if needed:
            if key not in md:
                raise FoamFileError(f"QGD.{model}: entry '{key}' is missing")
            opt[key] = float(md[key])
    opt["pRefCell"] = int(float(qgd.get("pRefCell", 0)))      # setRefCell(p, thermo.subDict("QGD"), ...) [createFields.H L165]
    opt["pRefValue"] = float(qgd.get("pRefValue", 0.0))
    gp = read_dict(os.path.join(case_dir, "constant", "gravitationalProperties"))
    g = gp["g"]
    g = [x for x in (g if isinstance(g, (list, tuple, np.ndarray)) else [g])]
    # `g g [0 1 -2 0 0 0 0] (0 -9.81 0);`, `g [0 1 -2 0 0 0 0] (0 -9.81 0);` or `g (0 -9.81 0);`: the vector is the last list of 3
    vec = None
    for item in reversed(g):
        if isinstance(item, (list, tuple, np.ndarray)) and len(item) == 3:
            vec = [float(x) for x in item]
            break
    if vec is None and len(g) == 3 and all(isinstance(x, (int, float)) for x in g):
        vec = [float(x) for x in g]
    if vec is None:
        raise FoamFileError(f"{case_dir}/constant/gravitationalProperties: cannot read the vector g")
    opt["g"] = tuple(vec)
    tdir = os.path.join(case_dir, str(time))
    opt["aQGD"] = 0.5
    apath = os.path.join(tdir, "alphaQGD")
    if _exists(apath):
        vals, _ = read_field(apath, mesh)
        if not np.all(vals == vals[0]):
            raise FoamFileError(f"{apath}: a non-uniform alphaQGD is not supported by the QHDFoam path")
        opt["aQGD"] = float(vals[0, 0])
    fs_path = os.path.join(case_dir, "system", "fvSchemes")
    fs = read_dict(fs_path)
    # fvsc::grad of U, W, T [QHDFoam/updateFields.H L36-40] and p [QHDUEqn.H L36]; qgdFlux(phi,U,Uf) [QHDUEqn.H L41], qgdFlux(phi,T,Tf) [QHDTEqn.H L65]
    words, flux = read_face_schemes(fs, ("div(phi,U)", "div(phi,T)"), ("grad(U)", "grad(W)", "grad(T)", "grad(p)"), fs_path)
    used = {t: w for t, w in words.items() if t != "grad(W)"}     # gradWf [QHDFoam/updateFields.H L38] is formed by the reference and never consumed
    if len(set(used.values())) > 1:
        raise FoamFileError(f"{fs_path}: fvsc gives different stencils to different terms ({used}); the resident QHDFoam case runs ONE stencil "
                            f"for its face gradients -- use the fvsc operators of a Device for mixed stencils")
    opt["stencil"] = used["grad(U)"]
    opt["fluxSchemeU"] = 1 if flux["div(phi,U)"] == "upwind" else 0
    opt["fluxSchemeT"] = 1 if flux["div(phi,T)"] == "upwind" else 0
    _check_fv_schemes(fs, mesh, need_laplacian=True, need_grad=True, what=fs_path)
    cd = read_dict(os.path.join(case_dir, "system", "controlDict"))
    opt["deltaT"] = float(cd["deltaT"])
    if _truthy(cd.get("adjustTimeStep", "no")):
        raise FoamFileError("adjustTimeStep yes: the QHDFoam path runs with the fixed deltaT of controlDict (its matrices are built once)")
    fsol_path = os.path.join(case_dir, "system", "fvSolution")
    if _exists(fsol_path):
        solvers = read_dict(fsol_path).get("solvers", {})
        tols, iters = [], []
        for key, entry in solvers.items() if isinstance(solvers, dict) else ():
            names = str(key).strip('"()').replace("|", " ").split()
            if not isinstance(entry, dict):
                continue
            if "p" in names:
                opt["pTol"] = float(entry.get("tolerance", 1e-6))
                opt["pRelTol"] = float(entry.get("relTol", 0.0))
                opt["pMaxIter"] = int(float(entry.get("maxIter", 1000)))
            if any(n in ("U", "T", "Ux", "Uy", "Uz") for n in names):
                tols.append(float(entry.get("tolerance", 1e-6)))
                iters.append(int(float(entry.get("maxIter", 1000))))
        if tols and opt["implicitDiffusion"]:
            opt["implicitTol"] = min(tols)
            opt["implicitMaxIter"] = max(iters)
    U, bU = read_field(os.path.join(tdir, "U"), mesh)
    T, bT = read_field(os.path.join(tdir, "T"), mesh)
    p, bP = read_field(os.path.join(tdir, "p"), mesh)
    ptw = [PATCH_WORDS.get(int(t), "patch") for t in mesh.array("patchType")]
    bcs = []
    for i, name in enumerate(mesh.patch_names):
        bu, bt = _bc(bU[name], True, ptw[i], f"U.{name}"), _bc(bT[name], False, ptw[i], f"T.{name}")
        if bu[0] == "qgdFlux" or bt[0] in ("qgdFlux", "slip"):
            raise FoamFileError(f"patch {name}: boundary condition not supported for U / T of a QHDFoam case")
        bcs.append({"U": bu, "T": bt, "p": _bc_p_qhd(bP[name], f"p.{name}")})
    return mesh, opt, {"U": U, "T": T[:, 0], "p": p[:, 0]}, bcs


def load_qhd_case(case_dir, time="0", device_id=0):
    """Case directory -> (Device, QHDFoamCase) ready to ``step()``: the createFields.H sequence of QHDFoam over the C-ABI."""
    from .fvsc import Device
    from .qhdfoam import QHDFoamCase, qhd_options

    mesh, opt, fields, bcs = read_qhd_case_setup(case_dir, time)
    dev = Device(mesh, device_id, fv_schemes={"fvsc": {"default": opt["stencil"]}})
    case = QHDFoamCase(dev, qhd_options(**opt))
    for i, bc in enumerate(bcs):
        case.set_bc(i, U=bc["U"], T=bc["T"], p=bc["p"])
    case.set_fields(fields["U"], fields["T"], fields["p"])
    return dev, case


def write_qhd_time(case, case_dir, time_name, bcs=None):
    """U, T, p of a QHDFoamCase into <case_dir>/<time_name>/ (the AUTO_WRITE fields of QHDFoam that this path carries); patch
    entries carry the patch values as ``value`` (fixedGradient patches also their ``gradient``)."""
    mesh = case.mesh
    if hasattr(mesh, "file_mesh"):
        raise FoamFileError("write_time: a case with cyclic patches carries copies of its cells; python -m qgdsolver_amd.QGDFoam writes its time directories")
    names = getattr(mesh, "patch_names", None) or [f"patch{i}" for i in range(mesh.nPatches)]
    ps, pz, pt = mesh.array("patchStart"), mesh.array("patchSize"), mesh.array("patchType")
    nIF = mesh.nInternalFaces
    dims = {"U": "[0 1 -1 0 0 0 0]", "T": "[0 0 0 1 0 0 0]", "p": "[0 2 -2 0 0 0 0]"}
    for fname in ("U", "T", "p"):
        internal = case.field(fname)
        bvals = case.field(fname + ".boundary")
        patches = {}
        for i, pn in enumerate(names):
            word = PATCH_WORDS.get(int(pt[i]), "patch")
            if word in _CONSTRAINT_BCS:
                patches[pn] = (word, None)
                continue
            kind, extra = "calculated", None
            if bcs is not None and fname in bcs[i]:
                kind = bcs[i][fname][0]
                if kind == "none":
                    kind = "calculated"
                if kind == "fixedGradient":
                    extra = {"gradient": float(bcs[i][fname][1] or 0.0)}
            b0 = int(ps[i]) - nIF
            patches[pn] = (kind, bvals[b0:b0 + int(pz[i])]) if extra is None else (kind, bvals[b0:b0 + int(pz[i])], extra)
        write_field(os.path.join(case_dir, str(time_name), fname), mesh, fname, internal, patches, dims[fname])


def _patch_values(mesh, internal, patches, what):
    """patch values (nBoundaryFaces) of a scalar field read by read_field: `value` where the file gives one, otherwise
    the owner cell's value (zeroGradient / calculated without value; also on constraint patches, which carry no field)"""
    nif = mesh.nInternalFaces
    own = mesh.array("owner")
    out = np.zeros(mesh.nBoundaryFaces)
    ps, pz, pt = mesh.array("patchStart"), mesh.array("patchSize"), mesh.array("patchType")
    names = getattr(mesh, "patch_names", None) or [f"patch{i}" for i in range(mesh.nPatches)]
    for i, name in enumerate(names):
        sl = slice(int(ps[i]) - nif, int(ps[i]) - nif + int(pz[i]))
        if PATCH_WORDS.get(int(pt[i]), "patch") in _CONSTRAINT_BCS:
            out[sl] = internal[own[int(ps[i]): int(ps[i]) + int(pz[i])]]   # carries no field; never read
            continue
        rec = patches[name]
        if rec["value"] is not None:
            out[sl] = rec["value"][:, 0]
        elif rec["type"] in ("zeroGradient", "calculated"):
            out[sl] = internal[own[int(ps[i]): int(ps[i]) + int(pz[i])]]
        else:
            raise FoamFileError(f"{what}: patch '{name}' of type '{rec['type']}' needs a value")
    return out


def load_case(case_dir, time="0", device_id=0):
    """Case directory -> (Device, QGDFoamCase) ready to ``step()``: the createFields.H sequence of QGDFoam over the
    C-ABI.  Needs the HIP device (there is no CPU fallback)."""
    from .fvsc import Device
    from .qgdfoam import QGDFoamCase, default_options

    mesh, opt, fields, bcs = read_case_setup(case_dir, time)
    if getattr(mesh, "cyclic_pairs", None):
        # translational cyclic pairs: glued, with translated copies of the cells behind either half (PolyMesh.unroll_cyclic); the real cells
        # keep their labels, case.field(...)[:case.mesh.file_mesh.nCells] are theirs
        if opt.get("implicitDiffusion") or "alphaQGD" in fields or "ScQGD" in fields:
            raise FoamFileError(f"{case_dir}: a case with cyclic patches runs the explicit branch (QGD {{ implicitDiffusion false; }}) with uniform alphaQGD / ScQGD")
        file_mesh = mesh
        mesh = file_mesh.unroll_cyclic(file_mesh.cyclic_pairs)
        mesh.file_mesh = file_mesh
        cg = mesh.array("cellGlobal")
        fields = {k: v[cg] for k, v in fields.items()}
    dev = Device(mesh, device_id, fv_schemes={"fvsc": {"default": opt["stencil"]}})
    case = QGDFoamCase(dev, default_options(**opt))
    for i, bc in enumerate(bcs):
        case.set_bc(i, U=bc["U"], T=bc["T"], p=bc["p"])
    if "alphaQGD" in fields or "ScQGD" in fields:
        case.set_qgd_coeffs(alphaQGD=fields.get("alphaQGD"), ScQGD=fields.get("ScQGD"))
    case.set_fields(fields["U"], fields["T"], fields["p"])
    return dev, case


def write_time(case, case_dir, time_name, bcs=None):
    """Write U, T, p, rho of a QGDFoamCase into <case_dir>/<time_name>/ (what runTime.write() leaves for QGDFoam's
    AUTO_WRITE fields); patch entries carry the patch values as ``value``."""
    mesh = case.mesh
    if hasattr(mesh, "file_mesh"):
        raise FoamFileError("write_time: a case with cyclic patches carries copies of its cells; python -m qgdsolver_amd.QGDFoam writes its time directories")
    names = getattr(mesh, "patch_names", None) or [f"patch{i}" for i in range(mesh.nPatches)]
    ps, pz, pt = mesh.array("patchStart"), mesh.array("patchSize"), mesh.array("patchType")
    nIF = mesh.nInternalFaces
    dims = {"U": "[0 1 -1 0 0 0 0]", "T": "[0 0 0 1 0 0 0]", "p": "[1 -1 -2 0 0 0 0]", "rho": "[1 -3 0 0 0 0 0]"}
    for fname in ("U", "T", "p", "rho"):
        internal = case.field(fname)
        bvals = case.field(fname + ".boundary")
        patches = {}
        for i, pn in enumerate(names):
            word = PATCH_WORDS.get(int(pt[i]), "patch")
            if word in _CONSTRAINT_BCS:
                patches[pn] = (word, None)
                continue
            kind = "calculated"
            if bcs is not None and fname in bcs[i]:
                kind = bcs[i][fname][0]
                if kind == "none":
                    kind = "calculated"
            b0 = int(ps[i]) - nIF
            patches[pn] = (kind, bvals[b0:b0 + int(pz[i])])
        write_field(os.path.join(case_dir, str(time_name), fname), mesh, fname, internal, patches, dims[fname])
