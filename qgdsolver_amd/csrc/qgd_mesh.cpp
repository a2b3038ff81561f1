// qgd_mesh.cpp -- host polyMesh: generators, geometry, adjacency.
#include "qgd_mesh.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <random>
#include <sstream>
#include <stdexcept>

#include "../../include/qgd_amd.h"

namespace qgd {

int32_t HostMesh::patchOfFace(int32_t f) const {
    if (f < nInternalFaces) return -1;
    for (size_t p = 0; p < patches.size(); ++p)
        if (f >= patches[p].start && f < patches[p].start + patches[p].size) return (int32_t)p;
    return -1;
}

// ---------------------------------------------------------------------------
// Geometry.  L0 assumption: OpenFOAM primitiveMeshTools::faceCentresAndAreas /
// cellCentresAndVols (triangle fan about the vertex average; face pyramids
// about the average of face centres).
// ---------------------------------------------------------------------------
void HostMesh::computeGeometry() {
    Sf.resize(3 * (size_t)nFaces);   // (every entry of these is written by the parallel loops below)
    Cf.resize(3 * (size_t)nFaces);
    magSf.resize((size_t)nFaces);
    C.resize(3 * (size_t)nCells);
    V.resize((size_t)nCells);
    const double* p = points.data();

#pragma omp parallel for schedule(static)
    for (int32_t f = 0; f < nFaces; ++f) {
        const int32_t* fp = &facePoints[faceOffsets[f]];
        const int np = faceSize(f);
        double* S = &Sf[3 * (size_t)f];
        double* c = &Cf[3 * (size_t)f];
        if (np == 3) {
            const double *a = p + 3 * (size_t)fp[0], *b = p + 3 * (size_t)fp[1],
                         *d = p + 3 * (size_t)fp[2];
            for (int k = 0; k < 3; ++k) c[k] = (1.0 / 3.0) * (a[k] + b[k] + d[k]);
            const double u[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]};
            const double v[3] = {d[0] - a[0], d[1] - a[1], d[2] - a[2]};
            S[0] = 0.5 * (u[1] * v[2] - u[2] * v[1]);
            S[1] = 0.5 * (u[2] * v[0] - u[0] * v[2]);
            S[2] = 0.5 * (u[0] * v[1] - u[1] * v[0]);
        } else {
            double fc[3] = {p[3 * (size_t)fp[0]], p[3 * (size_t)fp[0] + 1], p[3 * (size_t)fp[0] + 2]};
            for (int i = 1; i < np; ++i)
                for (int k = 0; k < 3; ++k) fc[k] += p[3 * (size_t)fp[i] + k];
            for (int k = 0; k < 3; ++k) fc[k] /= np;
            double sumN[3] = {0, 0, 0}, sumA = 0, sumAc[3] = {0, 0, 0};
            for (int i = 0; i < np; ++i) {
                const double* a = p + 3 * (size_t)fp[i];
                const double* b = p + 3 * (size_t)fp[(i + 1) % np];
                const double cc[3] = {a[0] + b[0] + fc[0], a[1] + b[1] + fc[1], a[2] + b[2] + fc[2]};
                const double u[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]};
                const double v[3] = {fc[0] - a[0], fc[1] - a[1], fc[2] - a[2]};
                const double n[3] = {u[1] * v[2] - u[2] * v[1], u[2] * v[0] - u[0] * v[2],
                                     u[0] * v[1] - u[1] * v[0]};
                const double an = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
                for (int k = 0; k < 3; ++k) {
                    sumN[k] += n[k];
                    sumAc[k] += an * cc[k];
                }
                sumA += an;
            }
            if (sumA < 1e-150) {
                for (int k = 0; k < 3; ++k) { c[k] = fc[k]; S[k] = 0.0; }
            } else {
                for (int k = 0; k < 3; ++k) { c[k] = (1.0 / 3.0) * sumAc[k] / sumA; S[k] = 0.5 * sumN[k]; }
            }
        }
        magSf[f] = std::sqrt(S[0] * S[0] + S[1] * S[1] + S[2] * S[2]);
    }

    // cell centres / volumes: per cell, its faces in the order the two sequential face walks of primitiveMesh::makeCellCentresAndVols add them
    // (the faces it owns in ascending label, then the faces it is the neighbour of, ascending) -- the same sums bit for bit, one cell per thread
    {
        const Csr cfo = buildCellFacesFoamOrder(*this);
#pragma omp parallel for schedule(static)
        for (int32_t c = 0; c < nCells; ++c) {
            double e[3] = {0, 0, 0}, cc[3] = {0, 0, 0}, vol = 0;
            const int32_t k0 = cfo.offsets[c], k1 = cfo.offsets[c + 1];
            for (int32_t k = k0; k < k1; ++k)
                for (int d = 0; d < 3; ++d) e[d] += Cf[3 * (size_t)cfo.items[k] + d];
            for (int d = 0; d < 3; ++d) e[d] /= (k1 - k0);
            for (int32_t k = k0; k < k1; ++k) {
                const int32_t f = cfo.items[k];
                const double* S = &Sf[3 * (size_t)f];
                const double* fc = &Cf[3 * (size_t)f];
                const double pyr3 = owner[f] == c ? S[0] * (fc[0] - e[0]) + S[1] * (fc[1] - e[1]) + S[2] * (fc[2] - e[2])
                                                  : S[0] * (e[0] - fc[0]) + S[1] * (e[1] - fc[1]) + S[2] * (e[2] - fc[2]);
                for (int d = 0; d < 3; ++d) cc[d] += pyr3 * (0.75 * fc[d] + 0.25 * e[d]);
                vol += pyr3;
            }
            if (std::fabs(vol) > 1e-300)
                for (int d = 0; d < 3; ++d) C[3 * (size_t)c + d] = cc[d] / vol;
            else
                for (int d = 0; d < 3; ++d) C[3 * (size_t)c + d] = e[d];
            V[c] = vol * (1.0 / 3.0);
        }
    }
    computeDerived();
}

// L0 assumption: surfaceInterpolation::makeWeights / makeDeltaCoeffs /
// makeNonOrthDeltaCoeffs and fvPatch::delta() (patch-normal on non-coupled
// patches), polyMesh::calcDirections for geometricD.
void HostMesh::computeDerived() {
    weights.resize((size_t)nFaces);
    deltaCoeffs.resize((size_t)nFaces);
    nonOrthDeltaCoeffs.resize((size_t)nFaces);
#pragma omp parallel for schedule(static)
    for (int32_t f = 0; f < nFaces; ++f) {
        weights[f] = 1.0; deltaCoeffs[f] = 0.0; nonOrthDeltaCoeffs[f] = 0.0;
        const double* S = &Sf[3 * (size_t)f];
        const double* cf = &Cf[3 * (size_t)f];
        const double* co = &C[3 * (size_t)owner[f]];
        if (f < nInternalFaces) {
            const double* cn = &C[3 * (size_t)neighbour[f]];
            const double sfdOwn = std::fabs(S[0] * (cf[0] - co[0]) + S[1] * (cf[1] - co[1]) + S[2] * (cf[2] - co[2]));
            const double sfdNei = std::fabs(S[0] * (cn[0] - cf[0]) + S[1] * (cn[1] - cf[1]) + S[2] * (cn[2] - cf[2]));
            weights[f] = (std::fabs(sfdOwn + sfdNei) > 1e-300) ? sfdNei / (sfdOwn + sfdNei) : 0.5;
            const double d[3] = {cn[0] - co[0], cn[1] - co[1], cn[2] - co[2]};
            const double magd = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
            deltaCoeffs[f] = 1.0 / magd;
            const double nd = (S[0] * d[0] + S[1] * d[1] + S[2] * d[2]) / magSf[f];
            nonOrthDeltaCoeffs[f] = 1.0 / std::max(nd, 0.05 * magd);
        } else {
            const double ms = magSf[f];
            if (ms > 0) {
                const double n[3] = {S[0] / ms, S[1] / ms, S[2] / ms};
                const double nd = n[0] * (cf[0] - co[0]) + n[1] * (cf[1] - co[1]) + n[2] * (cf[2] - co[2]);
                const double dv[3] = {n[0] * nd, n[1] * nd, n[2] * nd};
                const double magd = std::sqrt(dv[0] * dv[0] + dv[1] * dv[1] + dv[2] * dv[2]);
                deltaCoeffs[f] = 1.0 / magd;
                const double ndd = n[0] * dv[0] + n[1] * dv[1] + n[2] * dv[2];
                nonOrthDeltaCoeffs[f] = 1.0 / std::max(ndd, 0.05 * magd);
            }
        }
    }
    // the patch normal of symmetry planes (see Patch::nHat)
    for (Patch& pt : patches) {
        if (pt.type != QGD_PATCH_SYMMETRYPLANE || pt.nHatInherited || pt.size <= 0) continue;
        double sum[3] = {0, 0, 0};
        for (int32_t f = pt.start; f < pt.start + pt.size; ++f)
            for (int k = 0; k < 3; ++k) sum[k] += Sf[3 * (size_t)f + k] / magSf[f];
        for (int k = 0; k < 3; ++k) pt.nHat[k] = sum[k] / (double)pt.size;
    }
    // geometric directions from empty patches
    double dirVec[3] = {0, 0, 0};
    bool hasEmpty = false;
    for (const Patch& pt : patches) {
        if (pt.type != QGD_PATCH_EMPTY) continue;
        hasEmpty = hasEmpty || pt.size > 0;
        for (int32_t f = pt.start; f < pt.start + pt.size; ++f)
            for (int k = 0; k < 3; ++k) dirVec[k] += std::fabs(Sf[3 * (size_t)f + k] / magSf[f]);
    }
    nGeometricD = 0;
    const double mag = std::sqrt(dirVec[0] * dirVec[0] + dirVec[1] * dirVec[1] + dirVec[2] * dirVec[2]);
    for (int k = 0; k < 3; ++k) {
        geometricD[k] = 1;
        if (hasEmpty && mag > 0 && dirVec[k] / mag > 1e-6) geometricD[k] = -1;
        if (geometricD[k] == 1) nGeometricD++;
    }
}

std::string HostMesh::check() const {
    std::ostringstream e;
    if (nPoints <= 0 || nFaces <= 0 || nCells <= 0 || nInternalFaces < 0 || nInternalFaces > nFaces)
        return "bad sizes";
    if ((int32_t)faceOffsets.size() != nFaces + 1) return "faceOffsets size";
    if ((int32_t)owner.size() != nFaces) return "owner size";
    if ((int32_t)neighbour.size() != nInternalFaces) return "neighbour size";
    if (faceOffsets[0] != 0 || faceOffsets[nFaces] != (int32_t)facePoints.size()) return "faceOffsets range";
    for (int32_t f = 0; f < nFaces; ++f) {
        if (faceSize(f) < 3) return "face with < 3 points";
        if (owner[f] < 0 || owner[f] >= nCells) return "owner out of range";
        if (f < nInternalFaces) {
            if (neighbour[f] <= owner[f] || neighbour[f] >= nCells) return "neighbour must exceed owner";
            if (f > 0 && (owner[f] < owner[f - 1])) return "faces not in upper-triangular order";
        }
    }
    for (int32_t v : facePoints)
        if (v < 0 || v >= nPoints) return "face point out of range";
    int32_t expect = nInternalFaces;
    for (const Patch& p : patches) {
        if (p.start != expect) return "patches not contiguous";
        expect += p.size;
    }
    if (expect != nFaces) return "patches do not cover the boundary";
    return "";
}

// ---------------------------------------------------------------------------
// adjacency
// ---------------------------------------------------------------------------
// (parallel: counts and cursors by atomic increments, then every row sorted -- the rows come out as the serial walks left them: ascending
// face label per cell, ascending cell label per point; 64 M cells took 6 s of one thread here)
namespace {
inline void prefixSum(std::vector<int32_t>& off) {
    for (size_t i = 1; i < off.size(); ++i) off[i] += off[i - 1];
}
}  // namespace
Csr buildCellFaces(const HostMesh& m) {
    Csr c;
    c.offsets.assign((size_t)m.nCells + 1, 0);
#pragma omp parallel for schedule(static)
    for (int32_t f = 0; f < m.nFaces; ++f) {
#pragma omp atomic
        c.offsets[m.owner[f] + 1]++;
        if (f < m.nInternalFaces) {
#pragma omp atomic
            c.offsets[m.neighbour[f] + 1]++;
        }
    }
    prefixSum(c.offsets);
    c.items.resize((size_t)c.offsets[m.nCells]);
    std::vector<int32_t> fill(c.offsets.begin(), c.offsets.end() - 1);
#pragma omp parallel for schedule(static)
    for (int32_t f = 0; f < m.nFaces; ++f) {
        int32_t pos;
#pragma omp atomic capture
        pos = fill[m.owner[f]]++;
        c.items[pos] = f;
        if (f < m.nInternalFaces) {
#pragma omp atomic capture
            pos = fill[m.neighbour[f]]++;
            c.items[pos] = f;
        }
    }
#pragma omp parallel for schedule(static)
    for (int32_t i = 0; i < m.nCells; ++i) std::sort(c.items.begin() + c.offsets[i], c.items.begin() + c.offsets[i + 1]);   // ascending face label
    return c;
}

Csr buildCellFacesFoamOrder(const HostMesh& m) {
    // the faces a cell owns (ascending), then the faces it is the neighbour of (ascending)
    Csr c;
    c.offsets.assign((size_t)m.nCells + 1, 0);
    std::vector<int32_t> nOwned((size_t)m.nCells, 0);
#pragma omp parallel for schedule(static)
    for (int32_t f = 0; f < m.nFaces; ++f) {
#pragma omp atomic
        c.offsets[m.owner[f] + 1]++;
#pragma omp atomic
        nOwned[m.owner[f]]++;
        if (f < m.nInternalFaces) {
#pragma omp atomic
            c.offsets[m.neighbour[f] + 1]++;
        }
    }
    prefixSum(c.offsets);
    c.items.resize((size_t)c.offsets[m.nCells]);
    std::vector<int32_t> fillO(c.offsets.begin(), c.offsets.end() - 1), fillN((size_t)m.nCells);
#pragma omp parallel for schedule(static)
    for (int32_t i = 0; i < m.nCells; ++i) fillN[i] = c.offsets[i] + nOwned[i];
#pragma omp parallel for schedule(static)
    for (int32_t f = 0; f < m.nFaces; ++f) {
        int32_t pos;
#pragma omp atomic capture
        pos = fillO[m.owner[f]]++;
        c.items[pos] = f;
        if (f < m.nInternalFaces) {
#pragma omp atomic capture
            pos = fillN[m.neighbour[f]]++;
            c.items[pos] = f;
        }
    }
#pragma omp parallel for schedule(static)
    for (int32_t i = 0; i < m.nCells; ++i) {
        std::sort(c.items.begin() + c.offsets[i], c.items.begin() + c.offsets[i] + nOwned[i]);
        std::sort(c.items.begin() + c.offsets[i] + nOwned[i], c.items.begin() + c.offsets[i + 1]);
    }
    return c;
}

Csr buildPointCells(const HostMesh& m) { return buildPointCells(m, buildCellFaces(m)); }
Csr buildPointCells(const HostMesh& m, const Csr& cf) {
    // per point the distinct cells around it, ascending (a point is reached through several faces of the same cell)
    Csr pc;
    pc.offsets.assign((size_t)m.nPoints + 1, 0);
    auto pointsOf = [&](int32_t c, std::vector<int32_t>& pts) {
        pts.clear();
        for (int32_t k = cf.offsets[c]; k < cf.offsets[c + 1]; ++k) {
            const int32_t f = cf.items[k];
            for (int32_t q = m.faceOffsets[f]; q < m.faceOffsets[f + 1]; ++q) pts.push_back(m.facePoints[q]);
        }
        std::sort(pts.begin(), pts.end());
        pts.erase(std::unique(pts.begin(), pts.end()), pts.end());
    };
#pragma omp parallel
    {
        std::vector<int32_t> pts;
#pragma omp for schedule(static)
        for (int32_t c = 0; c < m.nCells; ++c) {
            pointsOf(c, pts);
            for (int32_t pt : pts) {
#pragma omp atomic
                pc.offsets[pt + 1]++;
            }
        }
    }
    prefixSum(pc.offsets);
    pc.items.resize((size_t)pc.offsets[m.nPoints]);
    std::vector<int32_t> fill(pc.offsets.begin(), pc.offsets.end() - 1);
#pragma omp parallel
    {
        std::vector<int32_t> pts;
#pragma omp for schedule(static)
        for (int32_t c = 0; c < m.nCells; ++c) {
            pointsOf(c, pts);
            for (int32_t pt : pts) {
                int32_t pos;
#pragma omp atomic capture
                pos = fill[pt]++;
                pc.items[pos] = c;
            }
        }
    }
#pragma omp parallel for schedule(static)
    for (int32_t i = 0; i < m.nPoints; ++i) std::sort(pc.items.begin() + pc.offsets[i], pc.items.begin() + pc.offsets[i + 1]);   // ascending cell label
    return pc;
}

// ---------------------------------------------------------------------------
// generators
// ---------------------------------------------------------------------------
// Hex cell-model faces of OpenFOAM (outward normals), local vertices
//   0:(0,0,0) 1:(1,0,0) 2:(1,1,0) 3:(0,1,0) 4:(0,0,1) 5:(1,0,1) 6:(1,1,1) 7:(0,1,1)
static const int kHexFace[6][4] = {
    {0, 4, 7, 3},  // x-min
    {1, 2, 6, 5},  // x-max
    {0, 1, 5, 4},  // y-min
    {3, 7, 6, 2},  // y-max
    {0, 3, 2, 1},  // z-min
    {4, 5, 6, 7}   // z-max
};

HostMesh makeBox(int32_t nx, int32_t ny, int32_t nzGlobal, int32_t kLo, int32_t kHi,
                 const double lo[3], const double hi[3], const int32_t patchTypes[6]) {
    if (nx < 1 || ny < 1 || nzGlobal < 1 || kLo < 0 || kHi > nzGlobal || kLo >= kHi)
        throw std::invalid_argument("makeBox: bad extents");
    const int64_t nz = kHi - kLo;
    const int64_t nC = (int64_t)nx * ny * nz;
    const int64_t nP = (int64_t)(nx + 1) * (ny + 1) * (nz + 1);
    const int64_t nIF = (int64_t)(nx - 1) * ny * nz + (int64_t)nx * (ny - 1) * nz + (int64_t)nx * ny * (nz - 1);
    const int64_t nBF = 2 * ((int64_t)ny * nz + (int64_t)nx * nz + (int64_t)nx * ny);
    const int64_t nF = nIF + nBF;
    if (4 * nF > INT32_MAX) throw std::invalid_argument("makeBox: mesh exceeds int32 face-point labels");

    HostMesh m;
    m.nPoints = (int32_t)nP; m.nFaces = (int32_t)nF; m.nInternalFaces = (int32_t)nIF; m.nCells = (int32_t)nC;
    m.points.resize(3 * (size_t)nP);
    const int64_t px = nx + 1, py = ny + 1;
#pragma omp parallel for schedule(static)
    for (int64_t k = 0; k <= nz; ++k)
        for (int64_t j = 0; j <= ny; ++j)
            for (int64_t i = 0; i <= nx; ++i) {
                double* q = &m.points[3 * (size_t)(i + px * (j + py * k))];
                q[0] = lo[0] + (hi[0] - lo[0]) * ((double)i / (double)nx);
                q[1] = lo[1] + (hi[1] - lo[1]) * ((double)j / (double)ny);
                q[2] = lo[2] + (hi[2] - lo[2]) * ((double)(k + kLo) / (double)nzGlobal);
            }
    m.faceOffsets.resize((size_t)nF + 1);
#pragma omp parallel for schedule(static)
    for (int64_t f = 0; f <= nF; ++f) m.faceOffsets[f] = (int32_t)(4 * f);
    m.facePoints.resize(4 * (size_t)nF);
    m.owner.resize((size_t)nF);
    m.neighbour.resize((size_t)nIF);

    auto P = [&](int64_t i, int64_t j, int64_t k) { return (int32_t)(i + px * (j + py * k)); };
    auto cellVerts = [&](int64_t i, int64_t j, int64_t k, int32_t v[8]) {
        v[0] = P(i, j, k); v[1] = P(i + 1, j, k); v[2] = P(i + 1, j + 1, k); v[3] = P(i, j + 1, k);
        v[4] = P(i, j, k + 1); v[5] = P(i + 1, j, k + 1); v[6] = P(i + 1, j + 1, k + 1); v[7] = P(i, j + 1, k + 1);
    };
    auto cellId = [&](int64_t i, int64_t j, int64_t k) { return (int32_t)(i + (int64_t)nx * (j + (int64_t)ny * k)); };

    // internal faces: per owner cell, neighbours i+1, j+1, k+1 (ascending label).
    // The first face label of each k-plane is known in closed form, so planes
    // are filled in parallel.
    std::vector<int64_t> planeStart((size_t)nz + 1, 0);
    for (int64_t k = 0; k < nz; ++k) {
        const int64_t inPlane = (int64_t)(nx - 1) * ny + (int64_t)nx * (ny - 1) + ((k < nz - 1) ? (int64_t)nx * ny : 0);
        planeStart[k + 1] = planeStart[k] + inPlane;
    }
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t k = 0; k < nz; ++k) {
        int64_t f = planeStart[k];
        int32_t v[8];
        for (int64_t j = 0; j < ny; ++j)
            for (int64_t i = 0; i < nx; ++i) {
                const int32_t c = cellId(i, j, k);
                cellVerts(i, j, k, v);
                if (i < nx - 1) {
                    for (int q = 0; q < 4; ++q) m.facePoints[4 * f + q] = v[kHexFace[1][q]];
                    m.owner[f] = c; m.neighbour[f] = c + 1; ++f;
                }
                if (j < ny - 1) {
                    for (int q = 0; q < 4; ++q) m.facePoints[4 * f + q] = v[kHexFace[3][q]];
                    m.owner[f] = c; m.neighbour[f] = c + nx; ++f;
                }
                if (k < nz - 1) {
                    for (int q = 0; q < 4; ++q) m.facePoints[4 * f + q] = v[kHexFace[5][q]];
                    m.owner[f] = c; m.neighbour[f] = c + nx * ny; ++f;
                }
            }
    }
    // boundary patches in blockMesh order; patch-face order as blockMesh's
    // block boundary loops (x: k outer, j inner; y: i outer, k inner; z: i outer, j inner)
    static const char* names[6] = {"xMin", "xMax", "yMin", "yMax", "zMin", "zMax"};
    int64_t f = nIF;
    int32_t v[8];
    for (int side = 0; side < 6; ++side) {
        Patch pt;
        pt.name = names[side];
        pt.type = patchTypes ? patchTypes[side] : QGD_PATCH_GENERIC;
        if (side == 4 && kLo > 0) pt.type = QGD_PATCH_HALO;
        if (side == 5 && kHi < nzGlobal) pt.type = QGD_PATCH_HALO;
        pt.start = (int32_t)f;
        auto emit = [&](int64_t i, int64_t j, int64_t k) {
            cellVerts(i, j, k, v);
            for (int q = 0; q < 4; ++q) m.facePoints[4 * f + q] = v[kHexFace[side][q]];
            m.owner[f] = cellId(i, j, k);
            ++f;
        };
        if (side < 2) {
            const int64_t i = (side == 0) ? 0 : nx - 1;
            for (int64_t k = 0; k < nz; ++k) for (int64_t j = 0; j < ny; ++j) emit(i, j, k);
        } else if (side < 4) {
            const int64_t j = (side == 2) ? 0 : ny - 1;
            for (int64_t i = 0; i < nx; ++i) for (int64_t k = 0; k < nz; ++k) emit(i, j, k);
        } else {
            const int64_t k = (side == 4) ? 0 : nz - 1;
            for (int64_t i = 0; i < nx; ++i) for (int64_t j = 0; j < ny; ++j) emit(i, j, k);
        }
        pt.size = (int32_t)(f - pt.start);
        // a slab knows the patch sizes of the whole box (zMin / zMax are cut planes on the inner slabs, real patches of the box)
        if (kLo > 0 || kHi < nzGlobal) pt.globalSize = side < 2 ? ny * nzGlobal : (side < 4 ? nx * nzGlobal : nx * ny);
        m.patches.push_back(pt);
    }
    // slab halo lists
    const bool cutLo = kLo > 0, cutHi = kHi < nzGlobal;
    m.ownedBegin = 0; m.ownedEnd = (int32_t)nC;
    if (cutLo || cutHi) {
        m.cellGlobalOffset = (int64_t)nx * ny * kLo;
        m.ownedBegin = cutLo ? (int32_t)((int64_t)nx * ny) : 0;
        m.ownedEnd = (int32_t)(nC - (cutHi ? (int64_t)nx * ny : 0));
        m.cellIsGhost.assign((size_t)nC, 0);
        m.haloGhost.assign(2, {});
        m.haloSend.assign(2, {});
        m.haloPeer.assign(2, -1);
        const int64_t plane = (int64_t)nx * ny;
        if (cutLo) {
            if (nz < 2) throw std::invalid_argument("makeBox: slab too thin for a ghost layer");
            for (int64_t c = 0; c < plane; ++c) {
                m.haloGhost[0].push_back((int32_t)c); m.cellIsGhost[c] = 1;
                m.haloSend[0].push_back((int32_t)(c + plane));
            }
        }
        if (cutHi) {
            if (nz < 2 + (cutLo ? 1 : 0)) throw std::invalid_argument("makeBox: slab too thin for a ghost layer");
            for (int64_t c = 0; c < plane; ++c) {
                m.haloGhost[1].push_back((int32_t)(c + plane * (nz - 1))); m.cellIsGhost[c + plane * (nz - 1)] = 1;
                m.haloSend[1].push_back((int32_t)(c + plane * (nz - 2)));
            }
        }
    }
    m.computeGeometry();
    // the faces of a cut plane are internal faces of the box; its z spacing is uniform, so their hQGDf there is 2 |C_O - C_f|
    for (const Patch& pt : m.patches)
        if (pt.type == QGD_PATCH_HALO)
            for (int32_t fc = pt.start; fc < pt.start + pt.size; ++fc) {
                double d2 = 0;
                for (int k = 0; k < 3; ++k) { const double x = m.C[3 * (size_t)m.owner[fc] + k] - m.Cf[3 * (size_t)fc + k]; d2 += x * x; }
                m.haloFaceH.push_back(2.0 * std::sqrt(d2));
            }
    return m;
}

// Masked one-cell-thick grid: used for the forward-step planform.
HostMesh makeForwardStep(int32_t nx, int32_t ny, int32_t ixStep, int32_t iyStep,
                         double lx, double ly, double lz) {
    if (nx < 2 || ny < 2 || ixStep < 1 || ixStep >= nx || iyStep < 1 || iyStep >= ny)
        throw std::invalid_argument("makeForwardStep: bad extents");
    auto solid = [&](int i, int j) { return i >= ixStep && j < iyStep; };
    std::vector<int32_t> cid((size_t)nx * ny, -1);
    int32_t nC = 0;
    for (int j = 0; j < ny; ++j) for (int i = 0; i < nx; ++i) if (!solid(i, j)) cid[i + (size_t)nx * j] = nC++;
    // points: compact numbering of used grid points, two z-layers
    const int px = nx + 1, py = ny + 1;
    std::vector<int32_t> pid((size_t)px * py * 2, -1);
    HostMesh m;
    auto P = [&](int i, int j, int k) -> int32_t {
        int32_t& id = pid[i + (size_t)px * (j + (size_t)py * k)];
        if (id < 0) {
            id = m.nPoints++;
            m.points.push_back(lx * ((double)i / nx));
            m.points.push_back(ly * ((double)j / ny));
            m.points.push_back(lz * (double)k);
        }
        return id;
    };
    // number points in (k, j, i) order over used cells for a deterministic layout
    for (int k = 0; k < 2; ++k) for (int j = 0; j <= ny; ++j) for (int i = 0; i <= nx; ++i) {
        bool used = false;
        for (int dj = -1; dj <= 0 && !used; ++dj) for (int di = -1; di <= 0 && !used; ++di) {
            const int ci = i + di, cj = j + dj;
            if (ci >= 0 && ci < nx && cj >= 0 && cj < ny && !solid(ci, cj)) used = true;
        }
        if (used) P(i, j, k);
    }
    auto cellVerts = [&](int i, int j, int32_t v[8]) {
        v[0] = P(i, j, 0); v[1] = P(i + 1, j, 0); v[2] = P(i + 1, j + 1, 0); v[3] = P(i, j + 1, 0);
        v[4] = P(i, j, 1); v[5] = P(i + 1, j, 1); v[6] = P(i + 1, j + 1, 1); v[7] = P(i, j + 1, 1);
    };
    auto addFace = [&](const int32_t v[8], int hexFace, int32_t own, int32_t nei) {
        for (int q = 0; q < 4; ++q) m.facePoints.push_back(v[kHexFace[hexFace][q]]);
        m.faceOffsets.push_back((int32_t)m.facePoints.size());
        m.owner.push_back(own);
        if (nei >= 0) m.neighbour.push_back(nei);
        m.nFaces++;
    };
    m.faceOffsets.push_back(0);
    int32_t v[8];
    for (int j = 0; j < ny; ++j) for (int i = 0; i < nx; ++i) {
        if (solid(i, j)) continue;
        const int32_t c = cid[i + (size_t)nx * j];
        cellVerts(i, j, v);
        if (i + 1 < nx && !solid(i + 1, j)) addFace(v, 1, c, cid[i + 1 + (size_t)nx * j]);
        if (j + 1 < ny && !solid(i, j + 1)) addFace(v, 3, c, cid[i + (size_t)nx * (j + 1)]);
    }
    m.nInternalFaces = m.nFaces;
    m.nCells = nC;
    auto beginPatch = [&](const char* name, int type) {
        Patch p; p.name = name; p.type = type; p.start = m.nFaces; m.patches.push_back(p);
    };
    auto endPatch = [&]() { m.patches.back().size = m.nFaces - m.patches.back().start; };
    beginPatch("inlet", QGD_PATCH_GENERIC);
    for (int j = 0; j < ny; ++j) if (!solid(0, j)) { cellVerts(0, j, v); addFace(v, 0, cid[(size_t)nx * j], -1); }
    endPatch();
    beginPatch("outlet", QGD_PATCH_GENERIC);
    for (int j = 0; j < ny; ++j) if (!solid(nx - 1, j)) { cellVerts(nx - 1, j, v); addFace(v, 1, cid[nx - 1 + (size_t)nx * j], -1); }
    endPatch();
    beginPatch("bottom", QGD_PATCH_GENERIC);
    for (int i = 0; i < nx; ++i) if (!solid(i, 0)) { cellVerts(i, 0, v); addFace(v, 2, cid[i], -1); }
    endPatch();
    beginPatch("top", QGD_PATCH_GENERIC);
    for (int i = 0; i < nx; ++i) if (!solid(i, ny - 1)) { cellVerts(i, ny - 1, v); addFace(v, 3, cid[i + (size_t)nx * (ny - 1)], -1); }
    endPatch();
    beginPatch("step", QGD_PATCH_GENERIC);
    for (int j = 0; j < iyStep; ++j) { cellVerts(ixStep - 1, j, v); addFace(v, 1, cid[ixStep - 1 + (size_t)nx * j], -1); }
    for (int i = ixStep; i < nx; ++i) { cellVerts(i, iyStep, v); addFace(v, 2, cid[i + (size_t)nx * iyStep], -1); }
    endPatch();
    beginPatch("frontAndBack", QGD_PATCH_EMPTY);
    for (int j = 0; j < ny; ++j) for (int i = 0; i < nx; ++i) if (!solid(i, j)) { cellVerts(i, j, v); addFace(v, 4, cid[i + (size_t)nx * j], -1); }
    for (int j = 0; j < ny; ++j) for (int i = 0; i < nx; ++i) if (!solid(i, j)) { cellVerts(i, j, v); addFace(v, 5, cid[i + (size_t)nx * j], -1); }
    endPatch();
    m.computeGeometry();
    return m;
}

void jitterPoints(HostMesh& m, double amplitude, uint64_t seed) {
    // boundary points stay where they are so patches remain planar
    std::vector<uint8_t> onBoundary((size_t)m.nPoints, 0);
    for (int32_t f = m.nInternalFaces; f < m.nFaces; ++f)
        for (int32_t q = m.faceOffsets[f]; q < m.faceOffsets[f + 1]; ++q) onBoundary[m.facePoints[q]] = 1;
    double vmin = 1e300;
    for (double v : m.V) vmin = std::min(vmin, v);
    const double h = std::cbrt(vmin);
    std::mt19937_64 gen(seed);
    std::uniform_real_distribution<double> u(-1.0, 1.0);
    for (int32_t p = 0; p < m.nPoints; ++p) {
        const double d[3] = {u(gen), u(gen), u(gen)};
        if (onBoundary[p]) continue;
        for (int k = 0; k < 3; ++k)
            if (m.geometricD[k] == 1) m.points[3 * (size_t)p + k] += amplitude * h * d[k];
    }
    m.computeGeometry();
}

void splitQuads(HostMesh& m, int32_t stride) {
    if (stride < 1) return;
    HostMesh o = m;
    m.faceOffsets.assign(1, 0);
    m.facePoints.clear(); m.owner.clear(); m.neighbour.clear();
    m.nFaces = 0;
    int32_t quadCount = 0;
    auto emit = [&](std::initializer_list<int32_t> v, int32_t own, int32_t nei) {
        for (int32_t x : v) m.facePoints.push_back(x);
        m.faceOffsets.push_back((int32_t)m.facePoints.size());
        m.owner.push_back(own);
        if (nei >= 0) m.neighbour.push_back(nei);
        m.nFaces++;
    };
    auto copyFace = [&](int32_t f) {
        const int32_t* fp = &o.facePoints[o.faceOffsets[f]];
        const int np = o.faceSize(f);
        const int32_t nei = f < o.nInternalFaces ? o.neighbour[f] : -1;
        // faces on empty patches are left alone (2-D stencils assume quads)
        const int32_t pi = o.patchOfFace(f);
        const bool skip = pi >= 0 && o.patches[pi].type == QGD_PATCH_EMPTY;
        if (np == 4 && !skip && (quadCount++ % stride) == 0) {
            emit({fp[0], fp[1], fp[2]}, o.owner[f], nei);
            emit({fp[0], fp[2], fp[3]}, o.owner[f], nei);
        } else {
            for (int q = 0; q < np; ++q) m.facePoints.push_back(fp[q]);
            m.faceOffsets.push_back((int32_t)m.facePoints.size());
            m.owner.push_back(o.owner[f]);
            if (nei >= 0) m.neighbour.push_back(nei);
            m.nFaces++;
        }
    };
    for (int32_t f = 0; f < o.nInternalFaces; ++f) copyFace(f);
    m.nInternalFaces = m.nFaces;
    for (size_t p = 0; p < o.patches.size(); ++p) {
        m.patches[p].start = m.nFaces;
        for (int32_t f = o.patches[p].start; f < o.patches[p].start + o.patches[p].size; ++f) copyFace(f);
        m.patches[p].size = m.nFaces - m.patches[p].start;
    }
    m.computeGeometry();
}

void splitEdges(HostMesh& m, int32_t stride) {
    if (stride < 1) return;
    // pick edges: edge (v0,v1) of every stride-th internal quad, unless one of its vertices already lies on a picked edge
    std::vector<std::pair<int32_t, int32_t>> edges;
    std::vector<uint8_t> used((size_t)m.nPoints, 0);
    int32_t count = 0;
    for (int32_t f = 0; f < m.nInternalFaces; ++f) {
        if (m.faceSize(f) != 4) continue;
        if ((count++ % stride) != 0) continue;
        const int32_t a = m.facePoints[m.faceOffsets[f]], b = m.facePoints[m.faceOffsets[f] + 1];
        if (used[a] || used[b]) continue;
        used[a] = used[b] = 1;
        edges.push_back({std::min(a, b), std::max(a, b)});
    }
    std::sort(edges.begin(), edges.end());
    std::vector<int32_t> mid(edges.size());
    for (size_t e = 0; e < edges.size(); ++e) {
        mid[e] = m.nPoints++;
        for (int k = 0; k < 3; ++k)
            m.points.push_back(0.5 * (m.points[3 * (size_t)edges[e].first + k] + m.points[3 * (size_t)edges[e].second + k]));
    }
    auto midOf = [&](int32_t a, int32_t b) -> int32_t {
        const std::pair<int32_t, int32_t> key{std::min(a, b), std::max(a, b)};
        auto it = std::lower_bound(edges.begin(), edges.end(), key);
        return (it != edges.end() && *it == key) ? mid[it - edges.begin()] : -1;
    };
    std::vector<int32_t> newOff(1, 0), newPts;
    for (int32_t f = 0; f < m.nFaces; ++f) {
        const int n = m.faceSize(f);
        const int32_t* fp = &m.facePoints[m.faceOffsets[f]];
        // faces of empty patches stay quads (the 2-D stencils assume them)
        const int32_t pi = m.patchOfFace(f);
        const bool frozen = pi >= 0 && m.patches[pi].type == QGD_PATCH_EMPTY;
        for (int q = 0; q < n; ++q) {
            newPts.push_back(fp[q]);
            const int32_t mm = frozen ? -1 : midOf(fp[q], fp[(q + 1) % n]);
            if (mm >= 0) newPts.push_back(mm);
        }
        newOff.push_back((int32_t)newPts.size());
    }
    m.faceOffsets.swap(newOff);
    m.facePoints.swap(newPts);
    m.computeGeometry();
}

}  // namespace qgd
