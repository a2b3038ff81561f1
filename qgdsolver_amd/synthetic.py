"""Synthetic input of the benchmark configurations (SURVEY 8(d) "C3 / C4 synthetic input"): what bench.py, the probes under scripts/ and the
tests start their boxes from.  Lives in the package so that the measurement harness does not import the test tree."""
import numpy as np


def box_initial_fields(C, seed=12345, noise=1e-3):
    """C3/C4 initial state on cell centres C (n,3): p = 1 + 0.1 exp(-|x-xc|^2/0.01), T = 1 (+ seeded noise so that
    rho carries uniform(-noise, noise) perturbations), U = 0.1 (sin2pi x cos2pi y, -cos2pi x sin2pi y, 0)."""
    C = np.asarray(C).reshape(-1, 3)
    x, y = C[:, 0], C[:, 1]
    xc = np.array([0.5, 0.5, 0.5])
    r2 = ((C - xc) ** 2).sum(axis=1)
    p = 1.0 + 0.1 * np.exp(-r2 / 0.01)
    rng = np.random.Generator(np.random.MT19937(seed))
    T = 1.0 + rng.uniform(-noise, noise, size=C.shape[0])
    U = np.zeros_like(C)
    U[:, 0] = 0.1 * np.sin(2 * np.pi * x) * np.cos(2 * np.pi * y)
    U[:, 1] = -0.1 * np.cos(2 * np.pi * x) * np.sin(2 * np.pi * y)
    return U, T, p


def c5_mesh(n, chunk, poly=False):
    """BASELINE config 5's stand-in (SURVEY 8(d) "C5 synthetic input"; no polyhedral mesher without OpenFOAM): an n^3 box of hexahedra with
    the vertices jittered by 0.2 h (seed 2024), every 7th quadrilateral split into two triangles (cells of 6-8 faces; `poly` also splits every
    11th edge: polygon faces), the cell labels scrambled within chunks of `chunk` labels and put back into Morton order by the library's own
    renumbering -- the "irregular stencil stress" of the north star at n = 252 (16 M cells)."""
    from .mesh import PolyMesh

    mesh = PolyMesh.box(n, n, n)
    mesh.jitter(0.2, seed=2024)
    mesh.split_quads(7)
    if poly:
        mesh.split_edges(11)
    rng = np.random.default_rng(7)
    perm = np.arange(mesh.nCells, dtype=np.int32)
    for a in range(0, mesh.nCells, chunk):
        b = min(a + chunk, mesh.nCells)
        perm[a:b] = a + rng.permutation(b - a)
    mesh.renumber(perm)
    mesh.renumber(mesh.morton_order())
    return mesh
