// qgd_setup.cpp -- per-mesh static stencil tables (host side, built once).
//
// What each table restates (listing lines under /root/reference/docs/html/):
//   GaussVolPoint 3-D coefficients   GaussVolPointBase3D_8C_source.html L161-476
//   GaussVolPoint 2-D coefficients   GaussVolPointBase2D_8C_source.html L122-291
//   leastSquares stencil + weights   extendedFaceStencilFindNeighbours_8C_source.html L41-86,
//                                    extendedFaceStencilCalculateWeights_8C_source.html L43-155
//   hQGDf / hQGD                     QGDCoeffs_8C_source.html L195-199, L298-376
//   vertex interpolation weights     OpenFOAM volPointInterpolation (L0 assumption)
#include "qgd_setup.hpp"

#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <unordered_map>

#include "../../include/qgd_amd.h"

namespace qgd {

namespace {
inline double dot(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
inline double norm(const double* a) { return std::sqrt(dot(a, a)); }
inline void cross(const double* a, const double* b, double* c) {
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}
}  // namespace

// CSR (off, items[, weights]) -> sliced ELL with 64-row slices
template <class VI, class VW>
static void toSlicedEll(const std::vector<int32_t>& off, int64_t nRows, const std::vector<int32_t>& items,
                        const std::vector<double>* weights, std::vector<int32_t>& slice, std::vector<uint8_t>& count,
                        VI& ellItems, VW* ellW, int32_t padItem) {
    const int64_t nSlices = (nRows + 63) / 64;
    slice.assign((size_t)nSlices + 1, 0);
    count.resize((size_t)nRows);
    std::vector<int32_t> width((size_t)nSlices, 0);
    bool tooLong = false;
#pragma omp parallel for schedule(static) reduction(|| : tooLong)
    for (int64_t s = 0; s < nSlices; ++s) {
        int32_t w = 0;
        for (int64_t r = 64 * s; r < std::min<int64_t>(nRows, 64 * s + 64); ++r) {
            const int32_t n = off[r + 1] - off[r];
            tooLong = tooLong || n > 255;
            count[r] = (uint8_t)std::min(n, 255);
            w = std::max(w, n);
        }
        width[s] = w;
    }
    if (tooLong) throw std::invalid_argument("gather row longer than 255 entries");
    for (int64_t s = 0; s < nSlices; ++s) slice[s + 1] = slice[s] + width[s];
    // (the padded tables are GBs at 64 M cells: every slice -- entries and padding -- is written, and so first touched, by the thread that owns it)
    ellItems.resize((size_t)slice[nSlices] * 64);
    if (ellW) ellW->resize((size_t)slice[nSlices] * 64);
#pragma omp parallel for schedule(static)
    for (int64_t s = 0; s < nSlices; ++s) {
        const int64_t r0 = 64 * s, r1 = std::min<int64_t>(nRows, r0 + 64);
        for (int32_t i = 0; i < width[s]; ++i) {
            const size_t row = ((size_t)slice[s] + i) * 64;
            for (int64_t lane = 0; lane < 64; ++lane) {
                const int64_t r = r0 + lane;
                const bool on = r < r1 && i < off[r + 1] - off[r];
                ellItems[row + lane] = on ? items[off[r] + i] : padItem;
                if (ellW) (*ellW)[row + lane] = on ? (*weights)[off[r] + i] : 0.0;
            }
        }
    }
}

int64_t StaticData::bytes() const {
    auto sz = [](auto& v) { return (int64_t)(v.size() * sizeof(v[0])); };
    int64_t b = sz(own) + sz(nei) + sz(verts) + sz(fkind) + sz(magSf) + sz(w) + sz(hf) + sz(dn) + sz(X) + sz(Cc) + sz(bN) +
                sz(bmvON) + sz(ip13) + sz(c2d) + sz(lsqSlice) + sz(lsqCnt) + sz(lsqCell) + sz(lsqGx) + sz(lsqGy) + sz(lsqGz) + sz(lsqDeg) + sz(lsqBndZero) + sz(bSymm) +
                sz(pcSlice) + sz(pcCount) + sz(pcCell) + sz(pcW) + sz(bpPoint) + sz(bpOff) + sz(bpFace) + sz(bpW) + sz(cpOff) + sz(cpKind) + sz(cpT) + sz(cfSlice) + sz(cfCount) + sz(cfItem) + sz(cfNbr) + sz(fpos) + sz(cfPos) +
                sz(V) + sz(hQGD) + sz(ghost) + sz(bPatch) + sz(hQGDb);
    for (int k = 0; k < 3; ++k) b += sz(Sf[k]);
    return b;
}

// ---- point constraints (StaticData::cpOff / cpKind / cpT) ------------------------------------------------------------------------------
// L0 assumptions restated from OpenFOAM v2312: symmetryPlanePointPatchField / symmetryPointPatchField / wedgePointPatchField::evaluate,
// PrimitivePatch::calcPointNormals (sum of the unit normals of the point's patch faces, divided by its magnitude + VSMALL),
// pointConstraints::makePatchPatchAddressing (every patch applies its constraint at the points of its rim = the end points of
// patch edges with one patch face), pointConstraint::applyConstraint / constraintTransformation.
namespace {
struct PointConstraint {
    int first = 0;
    double second[3] = {0, 0, 0};
    void apply(const double cd[3]) {
        if (first == 0) { first = 1; for (int k = 0; k < 3; ++k) second[k] = cd[k]; }
        else if (first == 1) {
            const double pn[3] = {cd[1] * second[2] - cd[2] * second[1], cd[2] * second[0] - cd[0] * second[2], cd[0] * second[1] - cd[1] * second[0]};
            const double mg = std::sqrt(pn[0] * pn[0] + pn[1] * pn[1] + pn[2] * pn[2]);
            if (mg > 1e-3) { first = 2; for (int k = 0; k < 3; ++k) second[k] = pn[k] / mg; }
        } else if (first == 2) {
            if (std::fabs(cd[0] * second[0] + cd[1] * second[1] + cd[2] * second[2]) > 1e-3) { first = 3; second[0] = second[1] = second[2] = 0.0; }
        }
    }
    void transformation(double T[9]) const {
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                const double ss = second[i] * second[j];
                T[3 * i + j] = first == 1 ? (i == j ? 1.0 : 0.0) - ss : (first == 2 ? ss : 0.0);
            }
    }
};
}  // namespace

static void buildPointConstraints(const HostMesh& m, const std::vector<int32_t>& bpSlot, StaticData& s) {
    bool any = false;
    for (const Patch& pt : m.patches)
        if (pt.size > 0 && (pt.type == QGD_PATCH_SYMMETRYPLANE || pt.type == QGD_PATCH_SYMMETRY || pt.type == QGD_PATCH_WEDGE)) any = true;
    if (!any) return;
    const size_t nBP = s.bpPoint.size();
    std::vector<std::vector<uint8_t>> kinds(nBP);
    std::vector<std::vector<double>> tens(nBP);
    std::vector<PointConstraint> corner(nBP);
    auto pushOp = [&](int32_t slot, uint8_t kind, const double T[9]) {
        kinds[(size_t)slot].push_back(kind);
        tens[(size_t)slot].insert(tens[(size_t)slot].end(), T, T + 9);
    };
    for (const Patch& pt : m.patches) {
        if (pt.size <= 0 || !(pt.type == QGD_PATCH_SYMMETRYPLANE || pt.type == QGD_PATCH_SYMMETRY || pt.type == QGD_PATCH_WEDGE)) continue;
        // meshPoints of the patch in order of appearance, their point normals, the rim of the patch
        std::vector<int32_t> meshPoints;
        std::unordered_map<int32_t, int32_t> local;
        std::vector<double> pn;   // 3 per patch point: sum of the unit normals of its patch faces (ascending face label)
        std::unordered_map<uint64_t, int32_t> edgeFaces;
        for (int32_t f = pt.start; f < pt.start + pt.size; ++f) {
            const double nf[3] = {m.Sf[3 * (size_t)f] / m.magSf[f], m.Sf[3 * (size_t)f + 1] / m.magSf[f], m.Sf[3 * (size_t)f + 2] / m.magSf[f]};
            const int32_t b = m.faceOffsets[f], e = m.faceOffsets[f + 1];
            for (int32_t q = b; q < e; ++q) {
                const int32_t p = m.facePoints[q];
                auto it = local.find(p);
                int32_t lp;
                if (it == local.end()) { lp = (int32_t)meshPoints.size(); local.emplace(p, lp); meshPoints.push_back(p); pn.insert(pn.end(), {0.0, 0.0, 0.0}); }
                else lp = it->second;
                for (int k = 0; k < 3; ++k) pn[3 * (size_t)lp + k] += nf[k];
                const int32_t p2 = m.facePoints[q + 1 < e ? q + 1 : b];
                const uint64_t key = ((uint64_t)(uint32_t)std::min(p, p2) << 32) | (uint32_t)std::max(p, p2);
                edgeFaces[key]++;
            }
        }
        for (size_t lp = 0; lp < meshPoints.size(); ++lp) {
            double* v = &pn[3 * lp];
            const double mg = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]) + 1e-300;
            for (int k = 0; k < 3; ++k) v[k] /= mg;
        }
        auto normalAt = [&](int32_t lp, double n[3]) {
            if (pt.type == QGD_PATCH_SYMMETRYPLANE) { for (int k = 0; k < 3; ++k) n[k] = pt.nHat[k]; }
            else if (pt.type == QGD_PATCH_WEDGE) { for (int k = 0; k < 3; ++k) n[k] = pn[k]; }   // wedgePointPatchField: pointNormals()[0]
            else { for (int k = 0; k < 3; ++k) n[k] = pn[3 * (size_t)lp + k]; }
        };
        for (size_t lp = 0; lp < meshPoints.size(); ++lp) {
            const int32_t slot = bpSlot[(size_t)meshPoints[lp]];
            if (slot < 0) continue;
            double n[3], T[9];
            normalAt((int32_t)lp, n);
            const double two = pt.type == QGD_PATCH_WEDGE ? 1.0 : 2.0;
            for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) T[3 * i + j] = (i == j ? 1.0 : 0.0) - two * (n[i] * n[j]);
            pushOp(slot, pt.type == QGD_PATCH_WEDGE ? 1 : 0, T);
        }
        std::vector<uint8_t> onRim(meshPoints.size(), 0);
        for (const auto& ef : edgeFaces)
            if (ef.second == 1) { onRim[(size_t)local[(int32_t)(ef.first >> 32)]] = 1; onRim[(size_t)local[(int32_t)(ef.first & 0xffffffffu)]] = 1; }
        for (size_t lp = 0; lp < meshPoints.size(); ++lp) {
            if (!onRim[lp]) continue;
            const int32_t slot = bpSlot[(size_t)meshPoints[lp]];
            if (slot < 0) continue;
            double n[3];
            // the constraint direction of the patch at this point: symmetryPlanePointPatch / wedgePointPatch::applyConstraint use the
            // patch normal (wedgePolyPatch::n() = the average face normal; for a planar wedge = the first point normal), symmetry its point normal
            normalAt((int32_t)lp, n);
            corner[(size_t)slot].apply(n);
        }
    }
    s.cpOff.assign(nBP + 1, 0);
    for (size_t i = 0; i < nBP; ++i) {
        if (corner[i].first != 0) { double T[9]; corner[i].transformation(T); pushOp((int32_t)i, 1, T); }
        s.cpOff[i + 1] = s.cpOff[i] + (int32_t)kinds[i].size();
    }
    s.cpKind.reserve((size_t)s.cpOff[nBP]); s.cpT.reserve(9 * (size_t)s.cpOff[nBP]);
    for (size_t i = 0; i < nBP; ++i) {
        s.cpKind.insert(s.cpKind.end(), kinds[i].begin(), kinds[i].end());
        s.cpT.insert(s.cpT.end(), tens[i].begin(), tens[i].end());
    }
}

StaticData buildStaticData(const HostMesh& m) {
    StaticData s;
    // QGD_SETUP_TIMING=1: this function's sections on stderr (qgd_device_create prints its own stages)
    const bool stageTiming = std::getenv("QGD_SETUP_TIMING") && std::atoi(std::getenv("QGD_SETUP_TIMING")) != 0;
    auto stageClock = std::chrono::steady_clock::now();
    auto stage = [&](const char* what) {
        if (!stageTiming) return;
        const auto t = std::chrono::steady_clock::now();
        std::fprintf(stderr, "  buildStaticData: %-40s %8.2f s\n", what, std::chrono::duration<double>(t - stageClock).count());
        stageClock = t;
    };
    s.nP = m.nPoints; s.nF = m.nFaces; s.nIF = m.nInternalFaces; s.nC = m.nCells; s.nBF = m.nBoundaryFaces();
    s.nGeomD = m.nGeometricD;
    const int64_t nF = s.nF, nIF = s.nIF, nC = s.nC, nBF = s.nBF;

    std::vector<int32_t> patchOf((size_t)nBF, -1);
    std::vector<uint8_t> patchType((size_t)nBF, 0);
    for (size_t p = 0; p < m.patches.size(); ++p)
        for (int32_t f = m.patches[p].start; f < m.patches[p].start + m.patches[p].size; ++f) {
            patchOf[f - nIF] = (int32_t)p;
            patchType[f - nIF] = (uint8_t)m.patches[p].type;
        }
    s.bPatch = patchOf;
    auto isRealPatchFace = [&](int64_t b) {
        const int t = patchType[b];
        return t != QGD_PATCH_EMPTY && t != QGD_PATCH_CYCLIC && t != QGD_PATCH_HALO;
    };
    auto hasFields = [&](int64_t b) { return patchType[b] != QGD_PATCH_EMPTY; };

    // ---- faces: topology + streamed geometry --------------------------------
    parallelCopy(s.own, m.owner);
    parallelCopy(s.nei, m.neighbour);
    s.verts.resize(4 * (size_t)nF);   // (every entry of these is written by the parallel loop below)
    s.fkind.resize((size_t)nF);
    for (int k = 0; k < 3; ++k) s.Sf[k].resize((size_t)nF);
    parallelCopy(s.magSf, m.magSf);
    parallelCopy(s.w, m.weights);
    s.hf.resize((size_t)nF);
    s.dn.resize((size_t)nF);
    bool hasTri = false;
#pragma omp parallel for schedule(static) reduction(|| : hasTri)
    for (int64_t f = 0; f < nF; ++f) hasTri = hasTri || m.faceSize((int32_t)f) == 3;
    s.hasTri = hasTri;
    const bool want3D = (m.nGeometricD == 3);
    if (want3D) {
        s.bmvON.assign((size_t)nBF, 0.0);
        s.bN.assign(4 * (size_t)nBF, 0.0);
        parallelCopy(s.X, m.points);
        parallelCopy(s.Cc, m.C);
    }

    // hQGDf of cut-plane faces as the unsharded mesh has it (HostMesh::haloFaceH), by boundary-face index; -1: not given
    std::vector<double> haloH((size_t)nBF, -1.0);
    size_t nHaloFaces = 0;
    for (const Patch& pt : m.patches) if (pt.type == QGD_PATCH_HALO) nHaloFaces += (size_t)pt.size;
    if (m.haloFaceH.size() == nHaloFaces) {   // a list of another length belongs to other face lists: ignored (local rule)
        size_t k = 0;
        for (const Patch& pt : m.patches)
            if (pt.type == QGD_PATCH_HALO)
                for (int32_t fc = pt.start; fc < pt.start + pt.size && k < m.haloFaceH.size(); ++fc) haloH[(size_t)(fc - nIF)] = m.haloFaceH[k++];
    }
#pragma omp parallel for schedule(static)
    for (int64_t f = 0; f < nF; ++f) {
        const int32_t* fp = &m.facePoints[m.faceOffsets[f]];
        const int n = m.faceSize((int32_t)f);
        for (int k = 0; k < 3; ++k) s.Sf[k][f] = m.Sf[3 * f + k];
        const int32_t o = m.owner[f];
        const double* O = &m.C[3 * (size_t)o];
        const double* cf = &m.Cf[3 * f];
        double Nbuf[3];
        const double* N;
        bool skip = false;
        if (f < nIF) {
            N = &m.C[3 * (size_t)m.neighbour[f]];
            double a[3], b[3];
            for (int k = 0; k < 3; ++k) { a[k] = O[k] - cf[k]; b[k] = N[k] - cf[k]; }
            s.hf[f] = 2.0 * std::min(norm(a), norm(b));
            s.dn[f] = m.nonOrthDeltaCoeffs[f];
        } else {
            const int64_t b = f - nIF;
            // mirror point C_O + 2 (C_f - C_O)   [GaussVolPointBase3D.C L142-147]
            for (int k = 0; k < 3; ++k) Nbuf[k] = O[k] + 2.0 * (cf[k] - O[k]);
            N = Nbuf;
            skip = !hasFields(b);
            const double hb = (m.deltaCoeffs[f] != 0.0) ? 1.0 / std::fabs(m.deltaCoeffs[f]) : 0.0;
            const bool coupled = patchType[b] == QGD_PATCH_CYCLIC || patchType[b] == QGD_PATCH_HALO;
            s.hf[f] = skip ? 0.0 : (coupled ? hb : hb * 2.0);  // no field entries on empty patches
            if (patchType[b] == QGD_PATCH_HALO && haloH[b] >= 0.0) s.hf[f] = haloH[b];   // the value of the unsharded mesh
            s.dn[f] = m.deltaCoeffs[f];
            if (want3D) {
                double d[3];
                for (int k = 0; k < 3; ++k) { d[k] = O[k] - N[k]; s.bN[4 * b + k] = N[k]; }
                s.bmvON[b] = norm(d);
            }
        }
        uint8_t kind = (n == 4) ? FK_QUAD : (n == 3 ? FK_TRI : FK_OTHER);
        if (skip) kind = FK_SKIP;
        s.fkind[f] = kind;
        for (int q = 0; q < 4; ++q) s.verts[4 * f + q] = q < n ? fp[q] : -1;
    }

    stage("faces: topology + streamed geometry");
    // ---- GaussVolPoint 2-D -----------------------------------------------------
    if (m.nGeometricD == 2) {
        int ie3 = 2;
        for (int d = 0; d < 3; ++d) if (m.geometricD[d] < 1) ie3 = d;
        const int ie1 = (ie3 == 0) ? 1 : 0;
        const int ie2 = (ie3 == 2) ? 1 : 2;
        s.ie1 = ie1; s.ie2 = ie2; s.ie3 = ie3;
        s.ip13.assign(2 * (size_t)nF, -1);
        s.c2d.assign(6 * (size_t)nF, 0.0);
        for (int64_t f = 0; f < nF; ++f) {
            const int32_t o = m.owner[f];
            const int32_t* fp = &m.facePoints[m.faceOffsets[f]];
            const int n = m.faceSize((int32_t)f);
            double v42[3];
            int32_t refCell;
            if (f < nIF) {
                const int32_t nb = m.neighbour[f];
                for (int k = 0; k < 3; ++k) v42[k] = m.C[3 * (size_t)nb + k] - m.C[3 * (size_t)o + k];
                refCell = nb;  // vertices are picked against the NEIGHBOUR centre [2D.C L129-147]
            } else {
                const int t = patchType[f - nIF];
                if (t == QGD_PATCH_EMPTY || t == QGD_PATCH_WEDGE || t == QGD_PATCH_CYCLIC) { s.fkind[f] = FK_SKIP; continue; }
                for (int k = 0; k < 3; ++k) v42[k] = 2.0 * (m.Cf[3 * f + k] - m.C[3 * (size_t)o + k]);
                refCell = o;
            }
            int32_t q1 = -1, q3 = -1;
            const double zref = m.C[3 * (size_t)refCell + ie3];
            for (int k = 0; k < n; ++k) if (m.points[3 * (size_t)fp[k] + ie3] >= zref) { q1 = fp[k]; break; }
            for (int k = 0; k < n; ++k) if (m.points[3 * (size_t)fp[k] + ie3] >= zref && fp[k] != q1) { q3 = fp[k]; break; }
            if (q1 < 0 || q3 < 0) { s.fkind[f] = FK_SKIP; continue; }
            s.ip13[2 * f] = q1; s.ip13[2 * f + 1] = q3;
            double v13[3];
            for (int k = 0; k < 3; ++k) v13[k] = m.points[3 * (size_t)q3 + k] - m.points[3 * (size_t)q1 + k];
            const double m42 = norm(v42), m13 = norm(v13);
            // e1/e2 are coordinate axes, so the dot products are single components
            const double cosa1 = v42[ie1] / m42, cosa2 = v13[ie1] / m13;
            const double sina1 = v42[ie2] / m42, sina2 = v13[ie2] / m13;
            const double den = sina2 * cosa1 - sina1 * cosa2;
            s.c2d[0 * nF + f] = sina2 / den;
            s.c2d[1 * nF + f] = sina1 / den;
            s.c2d[2 * nF + f] = cosa1 / den;
            s.c2d[3 * nF + f] = cosa2 / den;
            s.c2d[4 * nF + f] = m42;
            s.c2d[5 * nF + f] = m13;
        }
    }

    stage("GaussVolPoint 2-D");
    // ---- adjacency ---------------------------------------------------------------
    Csr cf = buildCellFaces(m);   // (once: the point-cell lists and the cells' gather lists below both walk it)
    Csr pc = buildPointCells(m, cf);

    stage("adjacency");
    // ---- leastSquares (allowed on 1-D/2-D meshes only [fvsc.C L60-63]) -------
    s.lsqBndZero.assign((size_t)nBF, 0);
    for (int64_t b = 0; b < nBF; ++b) {
        const int t = patchType[b];
        s.lsqBndZero[b] = (t == QGD_PATCH_EMPTY || t == QGD_PATCH_WEDGE || t == QGD_PATCH_CYCLIC || t == QGD_PATCH_HALO ||
                           t == QGD_PATCH_SYMMETRY || t == QGD_PATCH_SYMMETRYPLANE) ? 1 : 0;
    }
    {
        bool anySymm = false;
        for (int64_t b = 0; b < nBF; ++b) anySymm = anySymm || patchType[b] == QGD_PATCH_SYMMETRYPLANE || patchType[b] == QGD_PATCH_SYMMETRY || patchType[b] == QGD_PATCH_WEDGE;
        if (anySymm) {
            s.bSymm.assign((size_t)nBF, 0);
            for (int64_t b = 0; b < nBF; ++b)
                s.bSymm[b] = (patchType[b] == QGD_PATCH_SYMMETRYPLANE || patchType[b] == QGD_PATCH_SYMMETRY || patchType[b] == QGD_PATCH_WEDGE) ? 1 : 0;
        }
    }
    if (m.nGeometricD < 3) {
        std::vector<int32_t> lsqOff((size_t)nIF + 1, 0), lsqCellCsr;
        std::vector<double> gwCsr[3];
        s.lsqDeg.assign((size_t)nIF, 0);
        std::vector<uint8_t> userDeg((size_t)nIF, 0);
        for (int32_t f : m.degenerateFaces) if (f >= 0 && f < nIF) userDeg[f] = 1;
        std::vector<int32_t> nb;
        for (int64_t f = 0; f < nIF; ++f) {
            nb.clear();
            for (int32_t q = m.faceOffsets[f]; q < m.faceOffsets[f + 1]; ++q) {
                const int32_t pt = m.facePoints[q];
                for (int32_t k = pc.offsets[pt]; k < pc.offsets[pt + 1]; ++k) {
                    const int32_t c = pc.items[k];
                    if (std::find(nb.begin(), nb.end(), c) == nb.end()) nb.push_back(c);
                }
            }
            // weights
            double G[6] = {0, 0, 0, 0, 0, 0};
            std::vector<double> d(3 * nb.size()), w2(nb.size());
            for (size_t i = 0; i < nb.size(); ++i) {
                for (int k = 0; k < 3; ++k) d[3 * i + k] = m.C[3 * (size_t)nb[i] + k] - m.Cf[3 * f + k];
                const double* x = &d[3 * i];
                w2[i] = 1 / (x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
                G[0] += (x[0] * x[0]) * w2[i]; G[1] += (x[0] * x[1]) * w2[i]; G[2] += (x[0] * x[2]) * w2[i];
                G[3] += (x[1] * x[1]) * w2[i]; G[4] += (x[1] * x[2]) * w2[i]; G[5] += (x[2] * x[2]) * w2[i];
            }
            double G0[6] = {0, 0, 0, 0, 0, 0};
            if (std::fabs(G[0]) < 1e-15) G0[0] = 1;
            if (std::fabs(G[3]) < 1e-15) G0[3] = 1;
            if (std::fabs(G[5]) < 1e-15) G0[5] = 1;
            for (int k = 0; k < 6; ++k) G[k] = G[k] + G0[k];
            const double det = G[0] * G[3] * G[5] + G[1] * G[4] * G[2] + G[2] * G[1] * G[4] - G[0] * G[4] * G[4] -
                               G[1] * G[1] * G[5] - G[2] * G[3] * G[2];
            if (det < 1 || userDeg[f]) {
                // G stays un-inverted; the face falls back to nf*snGrad [ScalarGrad.C L76-83]: the weights' own degeneracy test
                // [CalcW.C L136-145] or the user's faceSet degenerateStencilFaces [leastSquaresStencil.C L63-128]
                s.lsqDeg[f] = 1;
            } else {
                const double I[6] = {(G[3] * G[5] - G[4] * G[4]) / det, (G[2] * G[4] - G[1] * G[5]) / det,
                                     (G[1] * G[4] - G[2] * G[3]) / det, (G[0] * G[5] - G[2] * G[2]) / det,
                                     (G[1] * G[2] - G[0] * G[4]) / det, (G[0] * G[3] - G[1] * G[1]) / det};
                for (int k = 0; k < 6; ++k) G[k] = I[k] - G0[k];
            }
            for (size_t i = 0; i < nb.size(); ++i) {
                const double* x = &d[3 * i];
                const double g[3] = {G[0] * x[0] + G[1] * x[1] + G[2] * x[2], G[1] * x[0] + G[3] * x[1] + G[4] * x[2],
                                     G[2] * x[0] + G[4] * x[1] + G[5] * x[2]};
                lsqCellCsr.push_back(nb[i]);
                for (int k = 0; k < 3; ++k) gwCsr[k].push_back(w2[i] * g[k]);
            }
            lsqOff[f + 1] = (int32_t)lsqCellCsr.size();
        }
        std::vector<int32_t> dupCells;
        toSlicedEll(lsqOff, nIF, lsqCellCsr, &gwCsr[0], s.lsqSlice, s.lsqCnt, s.lsqCell, &s.lsqGx, 0);
        toSlicedEll(lsqOff, nIF, lsqCellCsr, &gwCsr[1], s.lsqSlice, s.lsqCnt, dupCells, &s.lsqGy, 0);
        toSlicedEll(lsqOff, nIF, lsqCellCsr, &gwCsr[2], s.lsqSlice, s.lsqCnt, dupCells, &s.lsqGz, 0);
    }

    stage("leastSquares");
    // ---- vertex interpolation ------------------------------------------------------
    std::vector<uint8_t> isPatchPoint((size_t)m.nPoints, 0);
    for (int64_t b = 0; b < nBF; ++b) {
        if (!isRealPatchFace(b)) continue;
        const int64_t f = nIF + b;
        for (int32_t q = m.faceOffsets[f]; q < m.faceOffsets[f + 1]; ++q) isPatchPoint[m.facePoints[q]] = 1;
    }
    // 2-D: GaussVolPoint only ever reads the two face vertices on the far side of the empty direction [2D.C L129-147];
    // the other half of the vertices need no value (volPointInterpolation computes them, nothing observes them)
    std::vector<uint8_t> pointUsed((size_t)m.nPoints, 1);
    if (m.nGeometricD == 2) {
        std::fill(pointUsed.begin(), pointUsed.end(), 0);
        for (int32_t v : s.ip13) if (v >= 0) pointUsed[v] = 1;
    }
    std::vector<int32_t> pcOff((size_t)m.nPoints + 1, 0), pcCellCsr;
    std::vector<double> pcWCsr;
    for (int32_t p = 0; p < m.nPoints; ++p)
        pcOff[p + 1] = pcOff[p] + ((isPatchPoint[p] || !pointUsed[p]) ? 0 : pc.rowSize(p));
    pcCellCsr.resize((size_t)pcOff[m.nPoints]);
    pcWCsr.resize((size_t)pcOff[m.nPoints]);
#pragma omp parallel for schedule(static)
    for (int32_t p = 0; p < m.nPoints; ++p) {
        if (isPatchPoint[p] || !pointUsed[p]) continue;
        const int32_t n = pc.rowSize(p);
        int32_t* cells = &pcCellCsr[pcOff[p]];
        double* w = &pcWCsr[pcOff[p]];
        double sum = 0;
        for (int32_t i = 0; i < n; ++i) {
            const int32_t c = pc.items[pc.offsets[p] + i];
            double d[3];
            for (int k = 0; k < 3; ++k) d[k] = m.points[3 * (size_t)p + k] - m.C[3 * (size_t)c + k];
            cells[i] = c;
            w[i] = 1.0 / norm(d);
            sum += w[i];
        }
        for (int32_t i = 0; i < n; ++i) w[i] /= sum;
    }
    toSlicedEll(pcOff, m.nPoints, pcCellCsr, &pcWCsr, s.pcSlice, s.pcCount, s.pcCell, &s.pcW, -1);
    { std::vector<int32_t>().swap(pcCellCsr); std::vector<double>().swap(pcWCsr); std::vector<int32_t>().swap(pcOff); }
    {
        // patch points: boundary faces around each, ascending boundary-face label
        std::vector<int32_t> cnt((size_t)m.nPoints, 0);
        for (int64_t b = 0; b < nBF; ++b) {
            if (!isRealPatchFace(b)) continue;
            const int64_t f = nIF + b;
            for (int32_t q = m.faceOffsets[f]; q < m.faceOffsets[f + 1]; ++q) cnt[m.facePoints[q]]++;
        }
        std::vector<int32_t> slot((size_t)m.nPoints, -1);
        s.bpOff.push_back(0);
        for (int32_t p = 0; p < m.nPoints; ++p)
            if (isPatchPoint[p]) {
                slot[p] = (int32_t)s.bpPoint.size();
                s.bpPoint.push_back(p);
                s.bpOff.push_back(s.bpOff.back() + cnt[p]);
            }
        s.bpFace.resize((size_t)s.bpOff.back());
        s.bpW.resize((size_t)s.bpOff.back());
        std::vector<int32_t> fill(s.bpOff.begin(), s.bpOff.end() - 1);
        for (int64_t b = 0; b < nBF; ++b) {
            if (!isRealPatchFace(b)) continue;
            const int64_t f = nIF + b;
            for (int32_t q = m.faceOffsets[f]; q < m.faceOffsets[f + 1]; ++q) {
                const int32_t p = m.facePoints[q];
                const int32_t k = fill[slot[p]]++;
                s.bpFace[k] = (int32_t)b;
                double d[3];
                for (int j = 0; j < 3; ++j) d[j] = m.points[3 * (size_t)p + j] - m.Cf[3 * f + j];
                s.bpW[k] = 1.0 / norm(d);
            }
        }
        for (size_t i = 0; i < s.bpPoint.size(); ++i) {
            double sum = 0;
            for (int32_t k = s.bpOff[i]; k < s.bpOff[i + 1]; ++k) sum += s.bpW[k];
            for (int32_t k = s.bpOff[i]; k < s.bpOff[i + 1]; ++k) s.bpW[k] /= sum;
        }
        buildPointConstraints(m, slot, s);
    }

    stage("vertex interpolation");
    // ---- cells ---------------------------------------------------------------------
    s.V.assign(m.V.begin(), m.V.end());
    {
        // flux gather list: ascending face label == summation order of
        // fvc::surfaceIntegrate for that cell (upper-triangular face order)
        std::vector<int32_t> cfOff((size_t)nC + 1, 0), cfItemCsr;
        auto keep = [&](int32_t f) {
            if (f < nIF) return true;
            const int t = patchType[f - nIF];
            return t != QGD_PATCH_EMPTY && t != QGD_PATCH_HALO;
        };
#pragma omp parallel for schedule(static)
        for (int64_t c = 0; c < nC; ++c) {
            int32_t n = 0;
            for (int32_t k = cf.offsets[c]; k < cf.offsets[c + 1]; ++k) if (keep(cf.items[k])) ++n;
            cfOff[c + 1] = n;
        }
        for (int64_t c = 0; c < nC; ++c) cfOff[c + 1] += cfOff[c];
        cfItemCsr.resize((size_t)cfOff[nC]);
#pragma omp parallel for schedule(static)
        for (int64_t c = 0; c < nC; ++c) {
            int32_t o = cfOff[c];
            for (int32_t k = cf.offsets[c]; k < cf.offsets[c + 1]; ++k) {
                const int32_t f = cf.items[k];
                if (!keep(f)) continue;
                cfItemCsr[o++] = (m.owner[f] == c) ? f : ~f;
            }
        }
        toSlicedEll(cfOff, nC, cfItemCsr, nullptr, s.cfSlice, s.cfCount, s.cfItem, (RawVec<double>*)nullptr, 0);
        {
            // the cell on the other side of each entry (-1: a boundary face): the matrix products of the implicit solves and of the
            // pressure equation gather x[cfNbr] and a[face] side by side instead of following face -> owner/neighbour -> x
            std::vector<int32_t> nbrCsr(cfItemCsr.size());
#pragma omp parallel for schedule(static)
            for (int64_t k = 0; k < (int64_t)cfItemCsr.size(); ++k) {
                const int32_t it = cfItemCsr[k], f = it >= 0 ? it : ~it;
                nbrCsr[k] = f >= nIF ? -1 : (it >= 0 ? m.neighbour[f] : m.owner[f]);
            }
            std::vector<int32_t> slice2;
            std::vector<uint8_t> count2;
            toSlicedEll(cfOff, nC, nbrCsr, nullptr, slice2, count2, s.cfNbr, (RawVec<double>*)nullptr, -1);
        }
        // slot-major storage positions of the internal-face fluxes (see qgd_setup.hpp)
        {
            // rank of an internal face among the faces its owner owns (they are consecutive: upper-triangular order), then a stable partition
            // of the faces by rank -- a counting sort, in parallel: contiguous chunks, per-chunk counts per rank, positions from their prefix sums
            s.fpos.assign((size_t)nIF, 0);
            std::vector<int32_t> rank((size_t)nIF);
            int32_t maxRank = 0;
#pragma omp parallel for schedule(static) reduction(max : maxRank)
            for (int64_t f = 0; f < nIF; ++f) {
                int32_t r = 0;
                while (f - 1 - r >= 0 && m.owner[f - 1 - r] == m.owner[f]) ++r;
                rank[f] = r;
                maxRank = std::max(maxRank, r);
            }
            const bool labelOrder = std::getenv("QGD_FLUX_LABEL_ORDER") != nullptr;  // experiment switch
            const int64_t nChunks = std::max<int64_t>(1, std::min<int64_t>(1024, nIF / 4096));
            const size_t nR = (size_t)maxRank + 1;
            std::vector<int64_t> cnt((size_t)nChunks * nR, 0);
            auto chunkLo = [&](int64_t ch) { return nIF * ch / nChunks; };
#pragma omp parallel for schedule(static)
            for (int64_t ch = 0; ch < nChunks; ++ch)
                for (int64_t f = chunkLo(ch); f < chunkLo(ch + 1); ++f) cnt[(size_t)ch * nR + rank[f]]++;
            {   // exclusive prefix in (rank, chunk) order: where chunk ch's first face of rank r goes
                int64_t run = 0;
                for (size_t r = 0; r < nR; ++r)
                    for (int64_t ch = 0; ch < nChunks; ++ch) { const int64_t n = cnt[(size_t)ch * nR + r]; cnt[(size_t)ch * nR + r] = run; run += n; }
            }
#pragma omp parallel for schedule(static)
            for (int64_t ch = 0; ch < nChunks; ++ch)
                for (int64_t f = chunkLo(ch); f < chunkLo(ch + 1); ++f) s.fpos[f] = labelOrder ? (int32_t)f : (int32_t)(cnt[(size_t)ch * nR + rank[f]]++);
#pragma omp parallel for schedule(static)
            for (int64_t k = 0; k < (int64_t)cfItemCsr.size(); ++k) {
                const int32_t it = cfItemCsr[k], f = it >= 0 ? it : ~it;
                const int32_t pos = f < nIF ? s.fpos[f] : f;
                cfItemCsr[k] = it >= 0 ? pos : ~pos;
            }
            std::vector<int32_t> slice2;
            std::vector<uint8_t> count2;
            toSlicedEll(cfOff, nC, cfItemCsr, nullptr, slice2, count2, s.cfPos, (RawVec<double>*)nullptr, 0);
        }
        // hQGD: area-weighted mean of hQGDf over the cell's faces, OpenFOAM
        // cells() order, skipping empty/wedge patches [QGDCoeffs.C L323-362]
        Csr cfo = buildCellFacesFoamOrder(m);
        s.hQGD.assign((size_t)nC, 0.0);
#pragma omp parallel for schedule(static)
        for (int64_t c = 0; c < nC; ++c) {
            double hint = 0, surf = 0;
            for (int32_t k = cfo.offsets[c]; k < cfo.offsets[c + 1]; ++k) {
                const int32_t f = cfo.items[k];
                if (f >= nIF) {
                    const int t = patchType[f - nIF];
                    if (t == QGD_PATCH_EMPTY || t == QGD_PATCH_WEDGE) continue;
                }
                hint += s.hf[f] * m.magSf[f];
                surf += m.magSf[f];
            }
            s.hQGD[c] = hint / surf;
        }
    }
    s.hQGDb.assign((size_t)nBF, 0.0);
    for (int64_t b = 0; b < nBF; ++b) s.hQGDb[b] = s.hf[nIF + b] * 1.0;
    // cell roles of a shard: 0 = ordinary owned cell, 1 = ghost (refreshed by the halo exchange, never updated here),
    // 2 = owned cell whose records a neighbour needs (updated first so the exchange can overlap the rest)
    s.ghost = m.cellIsGhost;
    const size_t nSlots = m.haloGhost.size();
    if (!s.ghost.empty())
        for (size_t slot = 0; slot < nSlots; ++slot)
            for (int32_t c : m.haloSend[slot]) s.ghost[c] = 2;

    stage("cells (gather lists, positions, hQGD)");
    // ---- halo lists (cells + their real-patch boundary faces, ascending) -----
    s.haloGhost = m.haloGhost;
    s.haloSend = m.haloSend;
    s.haloGhostBF.assign(nSlots, {});
    s.haloSendBF.assign(nSlots, {});
    for (size_t slot = 0; slot < nSlots; ++slot) {
        std::vector<uint8_t> isG((size_t)nC, 0), isS((size_t)nC, 0);
        for (int32_t c : m.haloGhost[slot]) isG[c] = 1;
        for (int32_t c : m.haloSend[slot]) isS[c] = 1;
        for (int64_t b = 0; b < nBF; ++b) {
            if (patchType[b] == QGD_PATCH_HALO) continue;
            const int32_t o = m.owner[nIF + b];
            if (isG[o]) s.haloGhostBF[slot].push_back((int32_t)b);
            if (isS[o]) s.haloSendBF[slot].push_back((int32_t)b);
        }
    }
    stage("halo lists");
    return s;
}

FaceTiles buildFaceTiles(const StaticData& s, int32_t fb) {
    FaceTiles t;
    const int64_t nIF = s.nIF;
    if (nIF == 0 || s.nGeomD != 3 || (fb != 64 && fb != 128 && fb != 256)) return t;
    if (3 * (int64_t)s.nC > INT32_MAX || 3 * (int64_t)s.nP > INT32_MAX) return t;   // the kernel indexes 16-B pieces with 32-bit integers
    const int64_t nTiles = (nIF + fb - 1) / fb;
    const int32_t capC = faceTileCapCells(fb), capV = faceTileCapVerts(fb);
    std::vector<int32_t> cnt(2 * (size_t)nTiles, 0);
    auto collect = [&](int64_t tile, std::vector<int32_t>& uc, std::vector<int32_t>& uv) {
        uc.clear(); uv.clear();
        const int64_t f0 = tile * fb, f1 = std::min<int64_t>(nIF, f0 + fb);
        for (int64_t f = f0; f < f1; ++f) {
            uc.push_back(s.own[f]); uc.push_back(s.nei[f]);
            for (int q = 0; q < 4; ++q) if (s.verts[4 * f + q] >= 0) uv.push_back(s.verts[4 * f + q]);
        }
        std::sort(uc.begin(), uc.end()); uc.erase(std::unique(uc.begin(), uc.end()), uc.end());
        std::sort(uv.begin(), uv.end()); uv.erase(std::unique(uv.begin(), uv.end()), uv.end());
    };
    bool fits = true;
#pragma omp parallel
    {
        std::vector<int32_t> uc, uv;
#pragma omp for schedule(static)
        for (int64_t tile = 0; tile < nTiles; ++tile) {
            collect(tile, uc, uv);
            const bool over = (int32_t)uc.size() > capC || (int32_t)uv.size() > capV || uv.empty();
            cnt[2 * tile] = over ? 0 : (int32_t)uc.size();
            cnt[2 * tile + 1] = over ? 0 : (int32_t)uv.size();
        }
    }
    t.off.assign(2 * (size_t)(nTiles + 1), 0);
    int64_t totC = 0, totV = 0;
    for (int64_t tile = 0; tile < nTiles; ++tile) {
        t.maxCells = std::max(t.maxCells, cnt[2 * tile]);
        t.maxVerts = std::max(t.maxVerts, cnt[2 * tile + 1]);
        totC += cnt[2 * tile]; totV += cnt[2 * tile + 1];
        if (cnt[2 * tile] == 0) t.spill.push_back((int32_t)tile);
        if (totC > INT32_MAX || totV > INT32_MAX) { fits = false; break; }
        t.off[2 * (tile + 1)] = (int32_t)totC;
        t.off[2 * (tile + 1) + 1] = (int32_t)totV;
    }
    if (!fits || t.maxVerts == 0) return FaceTiles();
    t.cells.resize((size_t)totC); t.verts.resize((size_t)totV);
    t.locC.resize((size_t)nIF); t.locV.resize(2 * (size_t)nIF);
#pragma omp parallel
    {
        std::vector<int32_t> uc, uv;
#pragma omp for schedule(static)
        for (int64_t tile = 0; tile < nTiles; ++tile) {
            if (cnt[2 * tile] == 0) continue;
            collect(tile, uc, uv);
            std::copy(uc.begin(), uc.end(), t.cells.begin() + t.off[2 * tile]);
            std::copy(uv.begin(), uv.end(), t.verts.begin() + t.off[2 * tile + 1]);
            const int64_t f0 = tile * fb, f1 = std::min<int64_t>(nIF, f0 + fb);
            auto posC = [&](int32_t id) { return (uint32_t)(std::lower_bound(uc.begin(), uc.end(), id) - uc.begin()); };
            auto posV = [&](int32_t id) { return id < 0 ? 0u : (uint32_t)(std::lower_bound(uv.begin(), uv.end(), id) - uv.begin()); };
            for (int64_t f = f0; f < f1; ++f) {
                t.locC[f] = posC(s.own[f]) | (posC(s.nei[f]) << 16);
                t.locV[2 * f] = posV(s.verts[4 * f]) | (posV(s.verts[4 * f + 1]) << 16);
                t.locV[2 * f + 1] = posV(s.verts[4 * f + 2]) | (posV(s.verts[4 * f + 3]) << 16);
            }
        }
    }
    t.fb = fb;
    return t;
}

// ---- cell blocks of the fused face + cell kernel (qgd_setup.hpp FusedBlocks) ---------------------------------------------------------
namespace {
inline uint64_t spread21(uint64_t x) {   // 21 bits -> every third bit
    x &= 0x1fffffull;
    x = (x | x << 32) & 0x1f00000000ffffull;
    x = (x | x << 16) & 0x1f0000ff0000ffull;
    x = (x | x << 8) & 0x100f00f00f00f00full;
    x = (x | x << 4) & 0x10c30c30c30c30c3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}
// label -> position among the few hundred labels of one block: open addressing on a stamped table (no allocation, no clearing per block)
struct SmallMap {
    static constexpr uint32_t N = 4096;
    int32_t key[N], val[N];
    uint32_t stamp[N];
    uint32_t gen = 0;
    SmallMap() { for (uint32_t i = 0; i < N; ++i) stamp[i] = 0; }
    void clear() { ++gen; }
    uint32_t slot(int32_t k) const {
        uint32_t h = ((uint32_t)k * 2654435761u) >> 20;
        while (stamp[h] == gen && key[h] != k) h = (h + 1) & (N - 1);
        return h;
    }
    bool has(int32_t k) const { return stamp[slot(k)] == gen; }
    void put(int32_t k, int32_t v) { const uint32_t h = slot(k); stamp[h] = gen; key[h] = k; val[h] = v; }
    int32_t at(int32_t k) const { return val[slot(k)]; }
};
struct OneBlock {
    std::vector<int32_t> cells, verts, face, entry;   // entry: row-major here (own cell x capE), transposed at the end
    std::vector<int32_t> faces, others, vs;           // scratch of tryBlock (kept between blocks: no allocation per block)
    std::vector<uint8_t> vCount;                      // per staged vertex
    std::vector<uint16_t> vPos;                       // row-major here (vertex x maxPE)
    std::vector<double> vW;
    int32_t nAll = 0, maxPE = 0, lds = 0, ldsImpl = 0;   // own + across-a-face cells (cells.size() counts the extras too); LDS bytes of its records (explicit / implicit layout)
    std::vector<uint8_t> nEntry;
    int32_t nOwn = 0, maxE = 0;
    // 128-bit fingerprint of the block's local topology (counts, per-face positions, face entries, per-vertex cell positions): what two
    // blocks must share to share a template
    std::array<uint64_t, 2> topoHash() const {
        uint64_t h1 = 0x9e3779b97f4a7c15ull, h2 = 0xc2b2ae3d27d4eb4full;
        auto mix = [&](uint64_t x) {
            h1 = (h1 ^ x) * 0x100000001b3ull; h1 ^= h1 >> 29;
            h2 = (h2 + x) * 0xff51afd7ed558ccdull; h2 ^= h2 >> 32;
        };
        mix((uint64_t)nOwn); mix((uint64_t)nAll); mix((uint64_t)cells.size()); mix((uint64_t)verts.size()); mix((uint64_t)face.size()); mix((uint64_t)maxE); mix((uint64_t)maxPE);
        for (size_t lf = 0; lf < face.size() / 4; ++lf) { mix((uint32_t)face[4 * lf + 1]); mix((uint32_t)face[4 * lf + 2]); mix((uint32_t)face[4 * lf + 3]); }
        for (size_t j = 0; j < (size_t)nOwn; ++j) { mix(nEntry[j]); for (int k = 0; k < nEntry[j]; ++k) mix((uint32_t)entry[j * (size_t)maxE + k]); }
        for (size_t lv = 0; lv < verts.size(); ++lv) { mix(vCount[lv]); for (int k = 0; k < vCount[lv]; ++k) mix(vPos[lv * (size_t)maxPE + k]); }
        return {h1, h2};
    }
};
}  // namespace

FusedBlocks buildFusedBlocks(const StaticData& s) {
    FusedBlocks B;
    const auto tStart = std::chrono::steady_clock::now();
    const int64_t nC = s.nC, nIF = s.nIF;
    if (nC == 0 || nIF == 0 || s.nGeomD != 3) return B;
    if (3 * (int64_t)s.nC > INT32_MAX || 3 * (int64_t)s.nP > INT32_MAX) return B;
    // lattice spacing per axis: the mean distance of the two cell centres across the faces that look along that axis (dx, dy, dz on a box,
    // whatever its cells' aspect ratio); the mean cell size where an axis has no such faces
    double vol = 0.0;
    for (int64_t c = 0; c < nC; ++c) vol += s.V[c];
    double h[3] = {std::cbrt(vol / (double)nC), 0.0, 0.0};
    h[1] = h[2] = h[0];
    {
        double sum[3] = {0, 0, 0};
        int64_t cnt[3] = {0, 0, 0};
        for (int64_t f = 0; f < nIF; ++f) {
            const double a[3] = {std::fabs(s.Sf[0][f]), std::fabs(s.Sf[1][f]), std::fabs(s.Sf[2][f])};
            const int d = a[0] >= a[1] ? (a[0] >= a[2] ? 0 : 2) : (a[1] >= a[2] ? 1 : 2);
            sum[d] += std::fabs(s.Cc[3 * (size_t)s.nei[f] + d] - s.Cc[3 * (size_t)s.own[f] + d]);
            ++cnt[d];
        }
        for (int d = 0; d < 3; ++d) if (cnt[d] > 0 && sum[d] > 0.0) h[d] = sum[d] / (double)cnt[d];
    }
    // a shard: its ghost cells belong to no block (the halo exchange writes them).  The lattice is anchored at the OWNED cells -- a ghost
    // plane in front of them must not shift every brick off the shard's own planes (round 5 took the minimum over all cells: a slab between
    // two cuts started at lattice index 1 and averaged 106 cells per block instead of 128).
    std::vector<int32_t> ownedCells;
    ownedCells.reserve((size_t)nC);
    for (int64_t c = 0; c < nC; ++c) {
        const int role = s.ghost.empty() ? 0 : s.ghost[c];
        if (role == 1) continue;
        ownedCells.push_back((int32_t)c);
    }
    const int64_t nOwned = (int64_t)ownedCells.size();
    if (nOwned == 0) return B;
    double lo[3] = {1e300, 1e300, 1e300};
    for (int64_t i = 0; i < nOwned; ++i)
        for (int d = 0; d < 3; ++d) lo[d] = std::min(lo[d], s.Cc[3 * (size_t)ownedCells[i] + d]);
    auto latticeOf = [&](int32_t c, int d) {
        return (int64_t)std::min(2097151.0, std::max(0.0, std::floor((s.Cc[3 * (size_t)c + d] - lo[d]) / h[d] + 0.25)));
    };
    // Bricks of AT MOST 8 x 4 x 4 lattice cells whose extents divide the owned lattice evenly: an axis of n lattice cells is cut into
    // ceil(n / b) segments of floor / ceil(n / segments) cells (400 -> 50 x 8; a 50-plane slab -> 13 segments of 3 or 4 planes: 123 cells per
    // brick instead of twelve full layers and a flat two-plane rest).  Key = Morton code of the brick coordinates above the position inside
    // the brick, so that bricks which are neighbours in space are neighbours in the launch order, as before.
    int64_t nLat[3] = {1, 1, 1};
    for (int64_t i = 0; i < nOwned; ++i)
        for (int d = 0; d < 3; ++d) nLat[d] = std::max(nLat[d], latticeOf(ownedCells[i], d) + 1);
    // The brick is 8 x 4 x 4 unless another shape fills its blocks markedly better on this lattice (a 50^3 box: 7 x 13 x 13 bricks of 106 cells,
    // or 10^3 cubes of 5^3 = 125): among the shapes of at most 128 cells whose full brick fits the LDS budget of three blocks per CU, the one
    // with the most cells per brick wins if it beats 8 x 4 x 4 by more than 5 %.  QGD_FUSED_BRICK=bx,by,bz forces one (probes).
    int64_t kBrick[3] = {8, 4, 4};
    {
        auto fits = [](int64_t bx, int64_t by, int64_t bz) {
            const int64_t own = bx * by * bz, surf = bx * by + by * bz + bx * bz;
            const int64_t nAll = own + 2 * surf, tot = nAll + 4 * (bx + by + bz) + 8, nV = (bx + 1) * (by + 1) * (bz + 1), nF = 3 * own + surf;
            return own <= kFusedCells && nAll <= kFusedCapC && tot <= kFusedCapTot && nV <= kFusedCapV && nF <= kFusedCapF &&
                   48 * tot + 32 * nAll + std::max(72 * nV + 24 * nAll, 40 * nF) <= (int64_t)kFusedLdsTarget;
        };
        auto meanCells = [&](int64_t bx, int64_t by, int64_t bz) {
            const int64_t b[3] = {bx, by, bz};
            double m = 1.0;
            for (int d = 0; d < 3; ++d) m *= (double)nLat[d] / (double)((nLat[d] + b[d] - 1) / b[d]);
            return m;
        };
        int forced[3] = {0, 0, 0};
        const char* e = std::getenv("QGD_FUSED_BRICK");
        if (e && std::sscanf(e, "%d,%d,%d", &forced[0], &forced[1], &forced[2]) == 3 && forced[0] >= 1 && forced[0] <= 16 && forced[1] >= 1 &&
            forced[1] <= 16 && forced[2] >= 1 && forced[2] <= 16 && fits(forced[0], forced[1], forced[2])) {
            for (int d = 0; d < 3; ++d) kBrick[d] = forced[d];
        } else {
            double best = meanCells(8, 4, 4) * 1.05;
            for (int64_t bx = 3; bx <= 8; ++bx)
                for (int64_t by = 3; by <= 8; ++by)
                    for (int64_t bz = 3; bz <= 8; ++bz) {
                        if (!fits(bx, by, bz)) continue;
                        const double m = meanCells(bx, by, bz);
                        if (m > best) { best = m; kBrick[0] = bx; kBrick[1] = by; kBrick[2] = bz; }
                    }
        }
    }
    int lbits[3];   // bits of the position inside a brick, per axis
    for (int d = 0; d < 3; ++d) { lbits[d] = 0; while ((1ll << lbits[d]) < kBrick[d]) ++lbits[d]; }
    const int lshift = lbits[0] + lbits[1] + lbits[2];
    std::vector<int32_t> segOf[3], segLo[3];
    for (int d = 0; d < 3; ++d) {
        const int64_t n = nLat[d], ns = (n + kBrick[d] - 1) / kBrick[d];
        segOf[d].resize((size_t)n); segLo[d].resize((size_t)ns + 1);
        for (int64_t i = 0; i <= ns; ++i) segLo[d][(size_t)i] = (int32_t)(i * n / ns);
        for (int64_t i = 0; i < ns; ++i)
            for (int32_t q = segLo[d][(size_t)i]; q < segLo[d][(size_t)i + 1]; ++q) segOf[d][(size_t)q] = (int32_t)i;
    }
    std::vector<std::pair<uint64_t, int32_t>> key((size_t)nOwned);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < nOwned; ++i) {
        const int32_t c = ownedCells[i];
        uint64_t b[3], l[3];
        for (int d = 0; d < 3; ++d) {
            const int64_t q = latticeOf(c, d);
            b[d] = (uint64_t)segOf[d][(size_t)q];
            l[d] = (uint64_t)(q - segLo[d][b[d]]);
        }
        key[i] = {(spread21(b[0]) | spread21(b[1]) << 1 | spread21(b[2]) << 2) << lshift | l[0] | l[1] << lbits[0] | l[2] << (lbits[0] + lbits[1]), c};
    }
    std::vector<int32_t>().swap(ownedCells);
    // (sorted in runs, then merged: keeps the host peak at one copy and uses the cores)
    {
        const int nRuns = 16;
        std::vector<int64_t> cut(nRuns + 1);
        for (int r = 0; r <= nRuns; ++r) cut[r] = nOwned * r / nRuns;
#pragma omp parallel for schedule(dynamic, 1)
        for (int r = 0; r < nRuns; ++r) std::sort(key.begin() + cut[r], key.begin() + cut[r + 1]);
        for (int width = 1; width < nRuns; width *= 2) {
#pragma omp parallel for schedule(dynamic, 1)
            for (int r = 0; r < nRuns; r += 2 * width)
                if (r + width < nRuns)
                    std::inplace_merge(key.begin() + cut[r], key.begin() + cut[r + width], key.begin() + cut[std::min(nRuns, r + 2 * width)]);
        }
    }
    auto entriesOf = [&](int32_t c, int32_t* out) {   // ascending face label, ~f when the cell is the neighbour
        const int n = s.cfCount[c];
        const size_t base = (size_t)s.cfSlice[c >> 6] * 64 + (c & 63);
        for (int i = 0; i < n; ++i) out[i] = s.cfItem[base + (size_t)i * 64];
        return n;
    };
    // one range of the sorted cells -> one block, or false when it does not fit the caps
    auto tryBlock = [&](int64_t b0, int64_t b1, OneBlock& o, SmallMap& posC, SmallMap& posV, SmallMap& posF) {
        o.cells.clear(); o.verts.clear(); o.face.clear(); o.entry.clear(); o.nEntry.clear(); o.maxE = 0;
        o.nOwn = (int32_t)(b1 - b0);
        std::vector<int32_t>& faces = o.faces;
        faces.clear();
        int32_t ent[256];
        for (int64_t i = b0; i < b1; ++i) {
            const int32_t c = key[i].second;
            if (s.cfCount[c] == 255) return false;   // (a saturated count: more faces than the table holds)
            const int n = entriesOf(c, ent);
            o.maxE = std::max(o.maxE, n);
            for (int k = 0; k < n; ++k) { const int32_t f = ent[k] >= 0 ? ent[k] : ~ent[k]; if (f < nIF) faces.push_back(f); }
        }
        std::sort(faces.begin(), faces.end());
        faces.erase(std::unique(faces.begin(), faces.end()), faces.end());
        if ((int32_t)faces.size() > kFusedCapF) return false;
        // staged cells: own cells in block order, then the others in ascending label
        posC.clear(); posV.clear(); posF.clear();
        // (own cells in ascending label inside the block: in LDS a record is 12 banks wide, and neighbours 16, 32 or 64 positions apart -- what
        // the Morton order makes of the y and z neighbours of a brick -- sit in the same banks)
        for (int64_t i = b0; i < b1; ++i) o.cells.push_back(key[i].second);
        std::sort(o.cells.begin(), o.cells.end());
        for (size_t i = 0; i < o.cells.size(); ++i) posC.put(o.cells[i], (int32_t)i);
        std::vector<int32_t>&others = o.others, &vs = o.vs;
        others.clear(); vs.clear();
        for (int32_t f : faces) {
            if (!posC.has(s.own[f])) others.push_back(s.own[f]);
            if (!posC.has(s.nei[f])) others.push_back(s.nei[f]);
            for (int q = 0; q < 4; ++q) if (s.verts[4 * (size_t)f + q] >= 0) vs.push_back(s.verts[4 * (size_t)f + q]);
        }
        std::sort(others.begin(), others.end()); others.erase(std::unique(others.begin(), others.end()), others.end());
        std::sort(vs.begin(), vs.end()); vs.erase(std::unique(vs.begin(), vs.end()), vs.end());
        if ((int64_t)o.cells.size() + (int64_t)others.size() > kFusedCapC || (int32_t)vs.size() > kFusedCapV) return false;
        for (int32_t c : others) { posC.put(c, (int32_t)o.cells.size()); o.cells.push_back(c); }
        for (int32_t v : vs) { posV.put(v, (int32_t)o.verts.size()); o.verts.push_back(v); }
        o.nAll = (int32_t)o.cells.size();
        // the cells around every vertex (pointCells order = the summation order of volPointInterpolation); those not staged yet follow
        o.maxPE = 0;
        for (int32_t v : vs) o.maxPE = std::max<int32_t>(o.maxPE, s.pcCount[v]);
        o.vCount.assign(vs.size(), 0);
        o.vPos.assign(vs.size() * (size_t)std::max(o.maxPE, 1), 0);
        o.vW.assign(vs.size() * (size_t)std::max(o.maxPE, 1), 0.0);
        for (size_t lv = 0; lv < vs.size(); ++lv) {
            const int32_t v = vs[lv];
            const int n = s.pcCount[v];
            o.vCount[lv] = (uint8_t)n;
            const size_t base = (size_t)s.pcSlice[v >> 6] * 64 + (v & 63);
            for (int i = 0; i < n; ++i) {
                const int32_t c = s.pcCell[base + (size_t)i * 64];
                if (!posC.has(c)) {
                    if ((int32_t)o.cells.size() >= kFusedCapTot) return false;
                    posC.put(c, (int32_t)o.cells.size());
                    o.cells.push_back(c);
                }
                o.vPos[lv * o.maxPE + i] = (uint16_t)posC.at(c);
                o.vW[lv * o.maxPE + i] = s.pcW[base + (size_t)i * 64];
            }
        }
        // what the block needs of the kernel's LDS (fusedFaceCellKernel: RecA of every staged cell, RecB of the own + across-a-face ones, then
        // vertex records + coordinates + centres, which the fluxes overwrite): a block over the budget that lets three blocks share a CU is
        // cut like one over the caps, unless it is small already
        o.lds = 48 * (int32_t)o.cells.size() + 32 * o.nAll + std::max(72 * (int32_t)vs.size() + 24 * o.nAll, 40 * (int32_t)faces.size());
        // (the implicitDiffusion branch's assembly reuses the vertex region for four flux planes, then the gradients of the own + across-a-face
        // cells, then four planes again: qgd_kernels.hip fusedFaceCellKernel<..., IMPL>)
        o.ldsImpl = 48 * (int32_t)o.cells.size() + 32 * o.nAll + std::max({72 * (int32_t)vs.size() + 24 * o.nAll, 32 * (int32_t)faces.size(), 72 * o.nAll});
        if (o.lds > kFusedLdsTarget && o.nOwn > 32) return false;
        o.face.resize(4 * faces.size());
        for (size_t lf = 0; lf < faces.size(); ++lf) {
            const int32_t f = faces[lf];
            posF.put(f, (int32_t)lf);
            uint32_t pv[4] = {0, 0, 0, 0};
            for (int q = 0; q < 4; ++q) { const int32_t v = s.verts[4 * (size_t)f + q]; pv[q] = v >= 0 ? (uint32_t)posV.at(v) : 0u; }
            o.face[4 * lf] = f;
            o.face[4 * lf + 1] = (int32_t)((uint32_t)posC.at(s.own[f]) | (uint32_t)posC.at(s.nei[f]) << 16);
            o.face[4 * lf + 2] = (int32_t)(pv[0] | pv[1] << 16);
            o.face[4 * lf + 3] = (int32_t)(pv[2] | pv[3] << 16);
        }
        o.nEntry.assign((size_t)o.nOwn, 0);
        o.entry.assign((size_t)o.nOwn * (size_t)std::max(o.maxE, 1), 0);
        for (int32_t j = 0; j < o.nOwn; ++j) {
            const int n = entriesOf(o.cells[j], ent);
            o.nEntry[j] = (uint8_t)n;
            for (int k = 0; k < n; ++k) {
                const int32_t f = ent[k] >= 0 ? ent[k] : ~ent[k];
                o.entry[(size_t)j * o.maxE + k] = f < nIF ? (posF.at(f) << 1 | (ent[k] < 0 ? 1 : 0)) : ~f;
            }
        }
        return true;
    };
    // ranges of the sorted cells.  A range that does not fit the caps
    // is halved -- and a mesh whose 128-cell ranges mostly do not fit (cells with more than six faces: triangles, polyhedra) would end up with
    // 64-cell blocks whose second face pass runs nearly empty; so the ranges get shorter in steps until few of them need the cut.
    // First choice: the lattice BRICKS themselves -- the cells whose Morton keys share everything above the position inside the brick (8x4x4 lattice
    // cells) --, whatever the mesh's extents and wherever a shard's ghost planes sit; bricks of one 16x8x8 parent that are short of cells (the
    // rim of the mesh, a shard's one-plane boundary layer) are joined up to 128 cells, a brick with more is cut evenly.  Runs of a fixed
    // count instead (second choice, lengths 112 ... 64) drift across the bricks as soon as one brick is short.
    std::vector<int64_t> rangeStart;   // nRanges + 1 positions in the sorted cells
    int64_t nRanges = 0;
    auto brickRanges = [&]() {
        rangeStart.clear();
        auto part = [&](int64_t p0, int64_t p1) {   // [p0, p1) of the sorted cells, all of one role
            int64_t curStart = -1, curEnd = -1;
            uint64_t curParent = 0;
            auto flush = [&]() { if (curStart >= 0) rangeStart.push_back(curStart); curStart = -1; };
            int64_t i = p0;
            while (i < p1) {
                const uint64_t brick = key[i].first >> lshift;
                int64_t j = i + 1;
                while (j < p1 && (key[j].first >> lshift) == brick) ++j;
                const int64_t sz = j - i;
                if (sz > kFusedCells) {
                    flush();
                    const int64_t k = (sz + kFusedCells - 1) / kFusedCells;
                    for (int64_t q = 0; q < k; ++q) rangeStart.push_back(i + sz * q / k);
                } else if (curStart >= 0 && curEnd == i && (curEnd - curStart) + sz <= kFusedCells && (brick >> 3) == curParent) {
                    curEnd = j;
                } else {
                    flush();
                    curStart = i; curEnd = j; curParent = brick >> 3;
                }
                i = j;
            }
            flush();
        };
        part(0, nOwned);
        nRanges = (int64_t)rangeStart.size();
        rangeStart.push_back(nOwned);
    };
    auto runRanges = [&](int64_t len) {
        rangeStart.clear();
        for (int64_t p = 0; p < nOwned; p += len) rangeStart.push_back(p);
        nRanges = (int64_t)rangeStart.size();
        rangeStart.push_back(nOwned);
    };
    auto rangeOf = [&](int64_t r) {
        return std::pair<int64_t, int64_t>{rangeStart[r], rangeStart[r + 1]};
    };
    // Pass 1: where each range is cut (nearly always: not at all) and what the largest block needs.  Pass 2 builds every block again,
    // straight into the padded tables: twice the arithmetic instead of half a million small vectors kept between the passes.
    std::vector<int32_t> nOf;
    std::vector<std::vector<std::pair<int64_t, int64_t>>> cuts;   // only for the ranges that were cut
    using Hash = std::array<uint64_t, 2>;
    std::vector<Hash> hashOne;                  // the topology fingerprint of a range that is one block
    std::vector<std::vector<Hash>> hashCuts;    // ... of the blocks of a range that was cut
    bool failed = false;
    int64_t facesDone = 0, cellsTot = 0, cellsAll = 0, vertsTot = 0;
    int32_t maxC = 0, maxV = 0, maxF = 0, maxE = 1, maxAll = 0, maxPE = 1, maxLds = 0, maxLdsImpl = 0;
    // keep the first way of cutting ranges whose blocks average 104 cells or more, else the one with the largest average
    struct Kept { std::vector<int64_t> rangeStart; std::vector<int32_t> nOf; std::vector<std::vector<std::pair<int64_t, int64_t>>> cuts;
                  std::vector<Hash> hashOne; std::vector<std::vector<Hash>> hashCuts; bool bricks = false;
                  int64_t nRanges = 0, facesDone = 0, cellsTot = 0, cellsAll = 0, vertsTot = 0, blocks = 0;
                  int32_t maxC = 0, maxV = 0, maxF = 0, maxE = 1, maxAll = 0, maxPE = 1, maxLds = 0, maxLdsImpl = 0; } best;
    for (const int64_t len : {(int64_t)0, (int64_t)112, (int64_t)96, (int64_t)80, (int64_t)64}) {
        if (len == 0) brickRanges(); else runRanges(len);
        nOf.assign((size_t)nRanges, 1);
        cuts.assign((size_t)nRanges, {});
        hashOne.assign((size_t)nRanges, Hash{0, 0});
        hashCuts.assign((size_t)nRanges, {});
        failed = false;
        facesDone = cellsTot = cellsAll = vertsTot = 0;
        maxC = maxV = maxF = maxAll = maxLds = maxLdsImpl = 0; maxE = maxPE = 1;
#pragma omp parallel reduction(+ : facesDone, cellsTot, cellsAll, vertsTot) reduction(max : maxC, maxV, maxF, maxE, maxAll, maxPE, maxLds, maxLdsImpl)
        {
            std::vector<SmallMap> maps(3);
            OneBlock o;
#pragma omp for schedule(dynamic, 64)
            for (int64_t r = 0; r < nRanges; ++r) {
                bool stop;
#pragma omp atomic read
                stop = failed;
                if (stop) continue;
                std::vector<std::pair<int64_t, int64_t>> work{rangeOf(r)}, done;
                std::vector<Hash> doneHash;
                while (!work.empty()) {
                    const auto [b0, b1] = work.back();
                    work.pop_back();
                    if (tryBlock(b0, b1, o, maps[0], maps[1], maps[2])) {
                        done.push_back({b0, b1});
                        doneHash.push_back(o.topoHash());
                        maxC = std::max<int32_t>(maxC, (int32_t)o.cells.size());
                        maxV = std::max<int32_t>(maxV, (int32_t)o.verts.size());
                        maxF = std::max<int32_t>(maxF, (int32_t)o.face.size() / 4);
                        maxE = std::max(maxE, o.maxE);
                        maxAll = std::max(maxAll, o.nAll);
                        maxPE = std::max(maxPE, o.maxPE);
                        maxLds = std::max(maxLds, o.lds);
                        maxLdsImpl = std::max(maxLdsImpl, o.ldsImpl);
                        facesDone += (int64_t)o.face.size() / 4;
                        cellsTot += (int64_t)o.cells.size(); cellsAll += o.nAll; vertsTot += (int64_t)o.verts.size();
                        continue;
                    }
                    if (b1 - b0 == 1) {   // one cell with more faces / vertices than a block holds
#pragma omp atomic write
                        failed = true;
                        break;
                    }
                    const int64_t mid = (b0 + b1) / 2;
                    work.push_back({mid, b1});
                    work.push_back({b0, mid});
                }
                nOf[r] = (int32_t)done.size();
                if (done.size() != 1) { cuts[r] = std::move(done); hashCuts[r] = std::move(doneHash); }
                else hashOne[r] = doneHash[0];
            }
        }
        if (failed) break;
        int64_t blocks = 0;
        for (int64_t r = 0; r < nRanges; ++r) blocks += nOf[r];
        if (best.blocks == 0 || blocks < best.blocks) {
            best.bricks = (len == 0);
            best.rangeStart = rangeStart; best.nOf = nOf; best.cuts = cuts; best.nRanges = nRanges;
            best.hashOne = hashOne; best.hashCuts = hashCuts;
            best.facesDone = facesDone; best.cellsTot = cellsTot; best.cellsAll = cellsAll; best.vertsTot = vertsTot; best.blocks = blocks;
            best.maxC = maxC; best.maxV = maxV; best.maxF = maxF; best.maxE = maxE; best.maxAll = maxAll; best.maxPE = maxPE; best.maxLds = maxLds; best.maxLdsImpl = maxLdsImpl;
        }
        if (blocks * 104 <= nOwned) break;
    }
    if (!failed) {
        rangeStart.swap(best.rangeStart); nOf.swap(best.nOf); cuts.swap(best.cuts); nRanges = best.nRanges;
        hashOne.swap(best.hashOne); hashCuts.swap(best.hashCuts);
        facesDone = best.facesDone; cellsTot = best.cellsTot; cellsAll = best.cellsAll; vertsTot = best.vertsTot;
        maxC = best.maxC; maxV = best.maxV; maxF = best.maxF; maxE = best.maxE; maxAll = best.maxAll; maxPE = best.maxPE; maxLds = best.maxLds; maxLdsImpl = best.maxLdsImpl;
    }
    if (failed) return B;
    // a shard: the blocks that hold a cell a neighbour waits for (role 2) come first, so that the step can advance them before the others and
    // overlap the exchange with the rest (qgd_capi.cpp stepAdvance).  They are whole bricks like every other block -- round 5 gave the
    // one-plane boundary layer blocks of its own, flat 8 x 8 x 1 ones of 64 cells that staged three records per cell.
    std::vector<std::pair<int64_t, int64_t>> blk;   // every block's range of the sorted cells, in range order
    std::vector<Hash> blkHash;
    for (int64_t r = 0; r < nRanges; ++r) {
        if (nOf[r] == 1) { blk.push_back(rangeOf(r)); blkHash.push_back(hashOne[r]); }
        else { blk.insert(blk.end(), cuts[r].begin(), cuts[r].end()); blkHash.insert(blkHash.end(), hashCuts[r].begin(), hashCuts[r].end()); }
    }
    const int64_t nBlocks = (int64_t)blk.size();
    std::vector<uint8_t> layerBlock((size_t)nBlocks, 0);
    if (!s.ghost.empty()) {
#pragma omp parallel for schedule(static)
        for (int64_t b = 0; b < nBlocks; ++b)
            for (int64_t i = blk[b].first; i < blk[b].second; ++i) if (s.ghost[key[i].second] == 2) { layerBlock[b] = 1; break; }
    }
    std::vector<int64_t> first((size_t)nBlocks, 0);   // where block b of the range order goes
    int64_t nLayerBlocks = 0;
    {
        int64_t pos = 0;
        for (int pass = 1; pass >= 0; --pass) {
            for (int64_t b = 0; b < nBlocks; ++b) if (layerBlock[b] == pass) first[b] = pos++;
            if (pass == 1) nLayerBlocks = pos;
        }
    }
    B.maxC = maxC; B.maxV = maxV; B.maxF = maxF;
    B.capC = (B.maxC + 7) / 8 * 8; B.capV = (B.maxV + 7) / 8 * 8; B.capF = (B.maxF + 7) / 8 * 8;
    B.capE = maxE;
    B.capPE = maxPE; B.maxTot = maxC; B.maxAll = maxAll; B.maxLds = maxLds; B.maxLdsImpl = maxLdsImpl;
    // templates: the distinct topology fingerprints in ascending order (deterministic); a block's template = its fingerprint's rank; the
    // template's tables are written by the first block (in launch order) that has it.  QGD_FUSED_TEMPLATES=0: one template per block.
    std::vector<int32_t> tplOf((size_t)nBlocks, 0);      // by launch position
    std::vector<int64_t> tplWriter;                       // range-order index of the block that writes template t
    int64_t nTemplates = nBlocks;
    bool templated = false;
    {
        std::vector<Hash> uniq(blkHash);
        std::sort(uniq.begin(), uniq.end());
        uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
        const char* e = std::getenv("QGD_FUSED_TEMPLATES");
        templated = !(e && std::atoi(e) == 0) && (int64_t)uniq.size() < nBlocks;   // any two blocks alike
        if (templated) {
            nTemplates = (int64_t)uniq.size();
            tplWriter.assign((size_t)nTemplates, -1);
            std::vector<int64_t> writerPos((size_t)nTemplates, INT64_MAX);
            for (int64_t ib = 0; ib < nBlocks; ++ib) {
                const int64_t t = std::lower_bound(uniq.begin(), uniq.end(), blkHash[ib]) - uniq.begin();
                tplOf[(size_t)first[ib]] = (int32_t)t;
                if (first[ib] < writerPos[t]) { writerPos[t] = first[ib]; tplWriter[t] = ib; }
            }
        } else {
            for (int64_t b = 0; b < nBlocks; ++b) tplOf[b] = (int32_t)b;
        }
    }
    if (std::max(nBlocks * (int64_t)std::max({B.capC, B.capV * B.capPE, B.capF}), nTemplates * (int64_t)std::max({B.capV * B.capPE, 3 * B.capF, B.capE * kFusedCells})) >
        (int64_t)INT32_MAX) return B;
    B.nBlocks = (int32_t)nBlocks;
    B.nLayerBlocks = (int32_t)nLayerBlocks;
    B.nTemplates = (int32_t)nTemplates; B.templated = templated ? 1 : 0;
    for (int d = 0; d < 3; ++d) B.brick[d] = best.bricks ? (int32_t)kBrick[d] : 0;
    B.facesComputed = facesDone; B.cellsStaged = cellsTot; B.cellsStagedFull = cellsAll; B.vertsStaged = vertsTot;
    B.hdr.resize(4 * (size_t)nBlocks);
    B.hdr2.resize(4 * (size_t)nBlocks);
    B.vCount.resize((size_t)nBlocks * B.capV);
    B.vW.resize((size_t)nBlocks * B.capPE * B.capV);
    B.cells.resize((size_t)nBlocks * B.capC);
    B.verts.resize((size_t)nBlocks * B.capV);
    B.faceLabel.resize((size_t)nBlocks * B.capF);
    B.nEntry.resize((size_t)nBlocks * kFusedCells);
    B.vPos.resize((size_t)nTemplates * B.capPE * B.capV);
    B.facePos.resize((size_t)nTemplates * B.capF * 3);
    B.entry.resize((size_t)nTemplates * B.capE * kFusedCells);
#pragma omp parallel
    {
        std::vector<SmallMap> maps(3);
        OneBlock o;
#pragma omp for schedule(dynamic, 64)
        for (int64_t ib = 0; ib < nBlocks; ++ib) {
            const auto [b0, b1] = blk[ib];
            tryBlock(b0, b1, o, maps[0], maps[1], maps[2]);
            const size_t b = (size_t)first[ib];
            const int32_t nTot = (int32_t)o.cells.size(), nV = (int32_t)o.verts.size(), nF = (int32_t)o.face.size() / 4;
            const size_t t = (size_t)tplOf[b];
            B.hdr[4 * b] = o.nOwn; B.hdr[4 * b + 1] = o.nAll; B.hdr[4 * b + 2] = nV; B.hdr[4 * b + 3] = nF;
            B.hdr2[4 * b] = nTot; B.hdr2[4 * b + 1] = (int32_t)t; B.hdr2[4 * b + 2] = B.hdr2[4 * b + 3] = 0;
            for (int32_t i = 0; i < B.capC; ++i) B.cells[b * B.capC + i] = o.cells[std::min(i, nTot - 1)];
            for (int32_t i = 0; i < B.capV; ++i) {
                const int32_t n = i < nV ? o.vCount[i] : 0;
                B.vCount[b * B.capV + i] = (uint8_t)n;
                for (int32_t e = 0; e < B.capPE; ++e) B.vW[(b * B.capPE + e) * B.capV + i] = e < n ? o.vW[(size_t)i * o.maxPE + e] : 0.0;
            }
            for (int32_t i = 0; i < B.capV; ++i) B.verts[b * B.capV + i] = nV ? o.verts[std::min(i, nV - 1)] : 0;
            for (int32_t i = 0; i < B.capF; ++i) B.faceLabel[b * B.capF + i] = nF ? o.face[4 * (size_t)std::min(i, nF - 1)] : 0;
            for (int32_t j = 0; j < kFusedCells; ++j) B.nEntry[b * kFusedCells + j] = (uint8_t)(j < o.nOwn ? o.nEntry[j] : 0);
            if (templated && tplWriter[t] != ib) continue;   // the template's tables are another block's to write
            for (int32_t i = 0; i < B.capV; ++i) {
                const int32_t n = i < nV ? o.vCount[i] : 0;
                for (int32_t e = 0; e < B.capPE; ++e) B.vPos[(t * B.capPE + e) * B.capV + i] = e < n ? o.vPos[(size_t)i * o.maxPE + e] : (uint16_t)0;
            }
            for (int32_t i = 0; i < B.capF; ++i)
                for (int q = 0; q < 3; ++q) B.facePos[(t * B.capF + i) * 3 + q] = nF ? (uint32_t)o.face[4 * (size_t)std::min(i, nF - 1) + 1 + q] : 0u;
            for (int32_t j = 0; j < kFusedCells; ++j) {
                const int32_t nE = j < o.nOwn ? o.nEntry[j] : 0;
                for (int32_t e = 0; e < B.capE; ++e) B.entry[(t * B.capE + e) * kFusedCells + j] = e < nE ? o.entry[(size_t)j * o.maxE + e] : 0;
            }
        }
    }
    B.buildSeconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - tStart).count();
    return B;
}

}  // namespace qgd
