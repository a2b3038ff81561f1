// qgd_oracle.cpp -- CPU restatement of the QGDsolver face-flux hot path.
//
// TEST INFRASTRUCTURE ONLY (see qgd_oracle.h).  PARITY UNPINNED: the reference
// cannot be built or run here and ships no tests or golden vectors; this file
// follows the Doxygen source listings under /root/reference/docs/html/
// operation by operation, in the reference's own evaluation order, as
// single-threaded field-at-a-time passes (the structure of the OpenFOAM
// expressions it restates).  Citations: [file:lines] = listing lines of
// /root/reference/docs/html/<file with . -> _8, / -> _2>_source.html
// (physical HTML line = listing line + 101).
//
// Everything tagged "L0" restates OpenFOAM v2312 behaviour (the release pinned
// by /root/reference/README.md:32-37) that the listings call into but that is
// not itself part of the reference tree.
#include "qgd_oracle.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <set>
#include <string>
#include <vector>

namespace {

// codes shared with include/qgd_amd.h
enum { PATCH_GENERIC = 0, PATCH_EMPTY = 1, PATCH_SYMMETRYPLANE = 2, PATCH_SYMMETRY = 3,
       PATCH_WEDGE = 4, PATCH_CYCLIC = 5, PATCH_HALO = 6 };
enum { BC_ZEROGRADIENT = 0, BC_FIXEDVALUE = 1, BC_SLIP = 2, BC_QGDFLUX = 3, BC_NONE = 4, BC_QHDFLUX = 5 };
enum { FVSC_REDUCED = 0, FVSC_LEASTSQUARES = 1, FVSC_GAUSSVOLPOINT = 2 };

const double SMALL = 1e-15;   // OpenFOAM doubleScalarSMALL (L0)
const double GREAT = 1e15;
const double VSMALL = 1e-300;

typedef std::vector<double> dvec;
typedef std::vector<int> ivec;

inline double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
inline double mag3(const double* a) { return std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); }
inline void cross3(const double* a, const double* b, double* c) {
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}

struct PatchInfo {
    int type, start, size;
    // symmetryPlane: the one normal of the patch = average of its faces' unit normals (L0 assumption:
    // symmetryPlanePolyPatch::calcGeometry, n_ = gAverage(faceNormals())); symmetryPlaneFvPatchField reflects about it on every face
    double nHat[3] = {0, 0, 0};
};

// ---------------------------------------------------------------------------
// Mesh + L0 geometry
// ---------------------------------------------------------------------------
struct Mesh {
    int nP = 0, nF = 0, nIF = 0, nC = 0;
    dvec pts;
    ivec fOff, fPts, own, nei;
    std::vector<PatchInfo> patches;
    // geometry (L0)
    dvec Sf, Cf, magSf, C, V, w, delta, nonOrthDelta;
    int geomD[3] = {1, 1, 1};
    int nGeomD = 3;
    // adjacency (L0 orders)
    std::vector<ivec> pointCells;     // ascending cell label
    std::vector<ivec> cells;          // owned faces ascending, then neighbour faces ascending
    // volPointInterpolation addressing (L0)
    std::vector<char> isPatchFace;    // per boundary face
    std::vector<char> isPatchPoint;   // per point
    ivec bndMeshPoints;               // boundary.meshPoints()
    std::vector<ivec> bndPointFaces;  // per boundary point: boundary-face indices (ascending)
    std::vector<dvec> pointWeights;   // per point (non patch points)
    std::vector<dvec> bndPointWeights;// per boundary point
    // point constraints of vector / tensor point fields (L0 assumption: pointConstraints::constrain at the end of
    // volPointInterpolation::interpolateBoundaryField): per mesh point the ordered list of operations
    struct PointOp { int kind; double T[9]; };   // 0: x = (x + transform(T, x))/2 (symmetry evaluate), 1: x = transform(T, x)
    std::map<int, std::vector<PointOp>> pointOps;
    void pointConstraintOps();
    // halo
    std::vector<ivec> haloGhost, haloSend;      // one entry per halo slot (neighbouring shard)
    std::vector<ivec> haloGhostBF, haloSendBF;  // boundary-face indices (global label - nIF)
    dvec haloFaceH;                             // hQGDf of the halo-patch faces as the unsharded mesh has it, in patch order
    ivec userDegenerateFaces;                   // faceSet degenerateStencilFaces [leastSquaresStencil.C L63-128]
    bool sharded() const { for (const ivec& g : haloGhost) if (!g.empty()) return true; return false; }

    int nBF() const { return nF - nIF; }
    int fsize(int f) const { return fOff[f + 1] - fOff[f]; }
    const int* fp(int f) const { return &fPts[fOff[f]]; }
    bool patchHasFields(int p) const { return patches[p].type != PATCH_EMPTY; }  // emptyFvPatch::size()==0
    bool coupled(int p) const { return patches[p].type == PATCH_CYCLIC || patches[p].type == PATCH_HALO; }
    // the normal a reflecting patch field uses on face gf of patch p: patch().nf() (basicSymmetry: slip, symmetry) or the
    // patch's own normal (symmetryPlane)
    void symmNormal(int p, int gf, double n[3]) const {
        if (patches[p].type == PATCH_SYMMETRYPLANE) { for (int k = 0; k < 3; ++k) n[k] = patches[p].nHat[k]; return; }
        for (int k = 0; k < 3; ++k) n[k] = Sf[3 * (size_t)gf + k] / magSf[gf];
    }

    void geometry();
    void derivedGeometry();
    void addressing();
    void pointInterpolationWeights();
    void haloFaces();
};

// L0: primitiveMeshTools::faceCentresAndAreas, cellCentresAndVols
void Mesh::geometry() {
    Sf.assign(3 * (size_t)nF, 0); Cf.assign(3 * (size_t)nF, 0); magSf.assign(nF, 0);
    C.assign(3 * (size_t)nC, 0); V.assign(nC, 0);
    for (int f = 0; f < nF; ++f) {
        const int* q = fp(f);
        const int n = fsize(f);
        double* S = &Sf[3 * (size_t)f];
        double* c = &Cf[3 * (size_t)f];
        if (n == 3) {
            const double *a = &pts[3 * (size_t)q[0]], *b = &pts[3 * (size_t)q[1]], *d = &pts[3 * (size_t)q[2]];
            for (int k = 0; k < 3; ++k) c[k] = (1.0 / 3.0) * (a[k] + b[k] + d[k]);
            double u[3], v[3], x[3];
            for (int k = 0; k < 3; ++k) { u[k] = b[k] - a[k]; v[k] = d[k] - a[k]; }
            cross3(u, v, x);
            for (int k = 0; k < 3; ++k) S[k] = 0.5 * x[k];
        } else {
            double fc[3] = {pts[3 * (size_t)q[0]], pts[3 * (size_t)q[0] + 1], pts[3 * (size_t)q[0] + 2]};
            for (int i = 1; i < n; ++i) for (int k = 0; k < 3; ++k) fc[k] += pts[3 * (size_t)q[i] + k];
            for (int k = 0; k < 3; ++k) fc[k] /= n;
            double sumN[3] = {0, 0, 0}, sumA = 0, sumAc[3] = {0, 0, 0};
            for (int i = 0; i < n; ++i) {
                const double* a = &pts[3 * (size_t)q[i]];
                const double* b = &pts[3 * (size_t)q[(i + 1) % n]];
                double cc[3], u[3], v[3], nn[3];
                for (int k = 0; k < 3; ++k) { cc[k] = a[k] + b[k] + fc[k]; u[k] = b[k] - a[k]; v[k] = fc[k] - a[k]; }
                cross3(u, v, nn);
                const double an = mag3(nn);
                for (int k = 0; k < 3; ++k) { sumN[k] += nn[k]; sumAc[k] += an * cc[k]; }
                sumA += an;
            }
            if (sumA < 1e-150) { for (int k = 0; k < 3; ++k) { c[k] = fc[k]; S[k] = 0; } }
            else { for (int k = 0; k < 3; ++k) { c[k] = (1.0 / 3.0) * sumAc[k] / sumA; S[k] = 0.5 * sumN[k]; } }
        }
        magSf[f] = mag3(S);
    }
    dvec cEst(3 * (size_t)nC, 0);
    ivec nFc(nC, 0);
    for (int f = 0; f < nF; ++f) { for (int k = 0; k < 3; ++k) cEst[3 * (size_t)own[f] + k] += Cf[3 * (size_t)f + k]; nFc[own[f]]++; }
    for (int f = 0; f < nIF; ++f) { for (int k = 0; k < 3; ++k) cEst[3 * (size_t)nei[f] + k] += Cf[3 * (size_t)f + k]; nFc[nei[f]]++; }
    for (int c = 0; c < nC; ++c) for (int k = 0; k < 3; ++k) cEst[3 * (size_t)c + k] /= nFc[c];
    for (int f = 0; f < nF; ++f) {
        const int o = own[f];
        double d[3];
        for (int k = 0; k < 3; ++k) d[k] = Cf[3 * (size_t)f + k] - cEst[3 * (size_t)o + k];
        const double pyr3 = dot3(&Sf[3 * (size_t)f], d);
        for (int k = 0; k < 3; ++k) C[3 * (size_t)o + k] += pyr3 * (0.75 * Cf[3 * (size_t)f + k] + 0.25 * cEst[3 * (size_t)o + k]);
        V[o] += pyr3;
    }
    for (int f = 0; f < nIF; ++f) {
        const int n = nei[f];
        double d[3];
        for (int k = 0; k < 3; ++k) d[k] = cEst[3 * (size_t)n + k] - Cf[3 * (size_t)f + k];
        const double pyr3 = dot3(&Sf[3 * (size_t)f], d);
        for (int k = 0; k < 3; ++k) C[3 * (size_t)n + k] += pyr3 * (0.75 * Cf[3 * (size_t)f + k] + 0.25 * cEst[3 * (size_t)n + k]);
        V[n] += pyr3;
    }
    for (int c = 0; c < nC; ++c) {
        if (std::fabs(V[c]) > VSMALL) for (int k = 0; k < 3; ++k) C[3 * (size_t)c + k] /= V[c];
        else for (int k = 0; k < 3; ++k) C[3 * (size_t)c + k] = cEst[3 * (size_t)c + k];
        V[c] *= (1.0 / 3.0);
    }
    derivedGeometry();
}

void Mesh::derivedGeometry() {
    // L0: surfaceInterpolation weights / deltaCoeffs / nonOrthDeltaCoeffs,
    // fvPatch::delta() patch-normal on non-coupled patches
    w.assign(nF, 1.0); delta.assign(nF, 0.0); nonOrthDelta.assign(nF, 0.0);
    for (PatchInfo& pt : patches) {
        if (pt.type != PATCH_SYMMETRYPLANE || pt.size <= 0) continue;
        double sum[3] = {0, 0, 0};
        for (int f = pt.start; f < pt.start + pt.size; ++f) for (int k = 0; k < 3; ++k) sum[k] += Sf[3 * (size_t)f + k] / magSf[f];
        for (int k = 0; k < 3; ++k) pt.nHat[k] = sum[k] / (double)pt.size;
    }
    for (int f = 0; f < nF; ++f) {
        const double* S = &Sf[3 * (size_t)f];
        const double* cf = &Cf[3 * (size_t)f];
        const double* co = &C[3 * (size_t)own[f]];
        if (f < nIF) {
            const double* cn = &C[3 * (size_t)nei[f]];
            double a[3], b[3], d[3];
            for (int k = 0; k < 3; ++k) { a[k] = cf[k] - co[k]; b[k] = cn[k] - cf[k]; d[k] = cn[k] - co[k]; }
            const double sfdOwn = std::fabs(dot3(S, a)), sfdNei = std::fabs(dot3(S, b));
            w[f] = (std::fabs(sfdOwn + sfdNei) > VSMALL) ? sfdNei / (sfdOwn + sfdNei) : 0.5;
            const double magd = mag3(d);
            delta[f] = 1.0 / magd;
            nonOrthDelta[f] = 1.0 / std::max(dot3(S, d) / magSf[f], 0.05 * magd);
        } else if (magSf[f] > 0) {
            double n[3], a[3], dv[3];
            for (int k = 0; k < 3; ++k) { n[k] = S[k] / magSf[f]; a[k] = cf[k] - co[k]; }
            const double nd = dot3(n, a);
            for (int k = 0; k < 3; ++k) dv[k] = n[k] * nd;
            const double magd = mag3(dv);
            delta[f] = 1.0 / magd;
            nonOrthDelta[f] = 1.0 / std::max(dot3(n, dv), 0.05 * magd);
        }
    }
    // L0: polyMesh::calcDirections
    double dir[3] = {0, 0, 0};
    bool hasEmpty = false;
    for (const PatchInfo& p : patches) if (p.type == PATCH_EMPTY) {
        hasEmpty = hasEmpty || p.size > 0;
        for (int f = p.start; f < p.start + p.size; ++f)
            for (int k = 0; k < 3; ++k) dir[k] += std::fabs(Sf[3 * (size_t)f + k] / magSf[f]);
    }
    const double md = mag3(dir);
    nGeomD = 0;
    for (int k = 0; k < 3; ++k) {
        geomD[k] = (hasEmpty && md > 0 && dir[k] / md > 1e-6) ? -1 : 1;
        if (geomD[k] == 1) nGeomD++;
    }
}

// L0: primitiveMesh::calcCells / calcPointCells orders
void Mesh::addressing() {
    cells.assign(nC, ivec());
    for (int f = 0; f < nF; ++f) cells[own[f]].push_back(f);
    for (int f = 0; f < nIF; ++f) cells[nei[f]].push_back(f);
    pointCells.assign(nP, ivec());
    for (int c = 0; c < nC; ++c)
        for (int f : cells[c])
            for (int q = fOff[f]; q < fOff[f + 1]; ++q) {
                ivec& pc = pointCells[fPts[q]];
                if (pc.empty() || pc.back() != c) {
                    if (std::find(pc.begin(), pc.end(), c) == pc.end()) pc.push_back(c);
                }
            }
}

// L0: volPointInterpolation::calcBoundaryAddressing / makeInternalWeights /
// makeBoundaryWeights
void Mesh::pointInterpolationWeights() {
    const int nb = nBF();
    isPatchFace.assign(nb, 0);
    isPatchPoint.assign(nP, 0);
    for (const PatchInfo& p : patches) {
        if (p.type == PATCH_EMPTY || p.type == PATCH_CYCLIC || p.type == PATCH_HALO) continue;
        for (int f = p.start; f < p.start + p.size; ++f) {
            isPatchFace[f - nIF] = 1;
            for (int q = fOff[f]; q < fOff[f + 1]; ++q) isPatchPoint[fPts[q]] = 1;
        }
    }
    // boundary primitivePatch addressing: meshPoints in order of first
    // appearance, pointFaces ascending
    ivec bIndex(nP, -1);
    bndMeshPoints.clear(); bndPointFaces.clear();
    for (int f = nIF; f < nF; ++f)
        for (int q = fOff[f]; q < fOff[f + 1]; ++q) {
            const int pt = fPts[q];
            if (bIndex[pt] < 0) { bIndex[pt] = (int)bndMeshPoints.size(); bndMeshPoints.push_back(pt); bndPointFaces.push_back(ivec()); }
            bndPointFaces[bIndex[pt]].push_back(f - nIF);
        }
    pointWeights.assign(nP, dvec());
    for (int pt = 0; pt < nP; ++pt) {
        if (isPatchPoint[pt]) continue;
        const ivec& pc = pointCells[pt];
        dvec& pw = pointWeights[pt];
        pw.resize(pc.size());
        double sum = 0;
        for (size_t i = 0; i < pc.size(); ++i) {
            double d[3];
            for (int k = 0; k < 3; ++k) d[k] = pts[3 * (size_t)pt + k] - C[3 * (size_t)pc[i] + k];
            pw[i] = 1.0 / mag3(d);
            sum += pw[i];
        }
        for (size_t i = 0; i < pw.size(); ++i) pw[i] /= sum;
    }
    bndPointWeights.assign(bndMeshPoints.size(), dvec());
    for (size_t i = 0; i < bndMeshPoints.size(); ++i) {
        const int pt = bndMeshPoints[i];
        if (!isPatchPoint[pt]) continue;
        const ivec& pf = bndPointFaces[i];
        dvec& pw = bndPointWeights[i];
        pw.resize(pf.size());
        double sum = 0;
        for (size_t j = 0; j < pf.size(); ++j) {
            if (isPatchFace[pf[j]]) {
                const int f = nIF + pf[j];
                double d[3];
                for (int k = 0; k < 3; ++k) d[k] = pts[3 * (size_t)pt + k] - Cf[3 * (size_t)f + k];
                pw[j] = 1.0 / mag3(d);
                sum += pw[j];
            } else pw[j] = 0.0;
        }
        for (size_t j = 0; j < pw.size(); ++j) pw[j] /= sum;
    }
    pointConstraintOps();
}

// L0 (OpenFOAM v2312, restated from knowledge -- none of it is in the reference tree):
//  * pointPatchField::New gives the point field of a constraint patch the patch's own type; pointConstraints::constrain first calls
//    correctBoundaryConditions(): symmetryPlanePointPatchField / symmetryPointPatchField::evaluate set, patch by patch in patch order,
//    x = (x + transform(I - 2 nn, x))/2 at every point of the patch (n = symmetryPlanePolyPatch::n() | pointNormals()[point]),
//    wedgePointPatchField x = transform(I - nn, x) with n = pointNormals()[0];
//  * then constrainCorners(): the points on the rim of a patch (end points of patch edges with a single patch face) collect the
//    constraint directions of the patches they rim (pointConstraint::applyConstraint) and get x = transform(constraintTransformation(), x):
//    I - nn for one direction, the squared edge direction for two, 0 for three;
//  * transform() is the identity on scalars.
void Mesh::pointConstraintOps() {
    pointOps.clear();
    struct Corner { int first = 0; double second[3] = {0, 0, 0}; };
    std::map<int, Corner> corners;
    for (const PatchInfo& pt : patches) {
        const bool plane = pt.type == PATCH_SYMMETRYPLANE, symm = pt.type == PATCH_SYMMETRY, wedge = pt.type == PATCH_WEDGE;
        if (!(plane || symm || wedge) || pt.size <= 0) continue;
        // PrimitivePatch addressing of this patch: meshPoints in order of appearance, pointFaces ascending, edge -> number of faces
        ivec meshPoints;
        std::map<int, int> localOf;
        std::vector<ivec> pointFaces;
        std::map<std::pair<int, int>, int> nEdgeFaces;
        for (int f = pt.start; f < pt.start + pt.size; ++f) {
            const int n = fsize(f);
            const int* q = fp(f);
            for (int i = 0; i < n; ++i) {
                if (!localOf.count(q[i])) { localOf[q[i]] = (int)meshPoints.size(); meshPoints.push_back(q[i]); pointFaces.push_back(ivec()); }
                pointFaces[localOf[q[i]]].push_back(f);
                const int a = q[i], b = q[(i + 1) % n];
                nEdgeFaces[std::make_pair(std::min(a, b), std::max(a, b))] += 1;
            }
        }
        // PrimitivePatch::calcPointNormals
        dvec pointNormals(3 * meshPoints.size(), 0.0);
        for (size_t lp = 0; lp < meshPoints.size(); ++lp) {
            double* cur = &pointNormals[3 * lp];
            for (int f : pointFaces[lp]) for (int k = 0; k < 3; ++k) cur[k] += Sf[3 * (size_t)f + k] / magSf[f];
            const double mg = mag3(cur) + VSMALL;
            for (int k = 0; k < 3; ++k) cur[k] /= mg;
        }
        std::set<int> rim;
        for (const auto& e : nEdgeFaces) if (e.second == 1) { rim.insert(e.first.first); rim.insert(e.first.second); }
        for (size_t lp = 0; lp < meshPoints.size(); ++lp) {
            const double* n = plane ? pt.nHat : (wedge ? &pointNormals[0] : &pointNormals[3 * lp]);
            PointOp op;
            op.kind = wedge ? 1 : 0;
            for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) op.T[3 * i + j] = (i == j ? 1.0 : 0.0) - (wedge ? 1.0 : 2.0) * (n[i] * n[j]);
            pointOps[meshPoints[lp]].push_back(op);
            if (rim.count(meshPoints[lp])) {   // pointConstraint::applyConstraint(n)
                Corner& c = corners[meshPoints[lp]];
                if (c.first == 0) { c.first = 1; for (int k = 0; k < 3; ++k) c.second[k] = n[k]; }
                else if (c.first == 1) {
                    double planeNormal[3];
                    cross3(n, c.second, planeNormal);
                    const double mg = mag3(planeNormal);
                    if (mg > 1e-3) { c.first = 2; for (int k = 0; k < 3; ++k) c.second[k] = planeNormal[k] / mg; }
                } else if (c.first == 2) {
                    if (std::fabs(dot3(n, c.second)) > 1e-3) { c.first = 3; c.second[0] = c.second[1] = c.second[2] = 0.0; }
                }
            }
        }
    }
    for (const auto& pc : corners) {   // pointConstraint::constraintTransformation
        const Corner& c = pc.second;
        PointOp op;
        op.kind = 1;
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
            const double ss = c.second[i] * c.second[j];
            op.T[3 * i + j] = c.first == 1 ? (i == j ? 1.0 : 0.0) - ss : (c.first == 2 ? ss : 0.0);
        }
        pointOps[pc.first].push_back(op);
    }
}

void Mesh::haloFaces() {
    haloGhostBF.resize(haloGhost.size());
    haloSendBF.resize(haloGhost.size());
    for (size_t side = 0; side < haloGhost.size(); ++side) {
        std::vector<char> isG(nC, 0), isS(nC, 0);
        for (int c : haloGhost[side]) isG[c] = 1;
        for (int c : haloSend[side]) isS[c] = 1;
        haloGhostBF[side].clear(); haloSendBF[side].clear();
        for (size_t p = 0; p < patches.size(); ++p) {
            if (patches[p].type == PATCH_HALO) continue;
            for (int f = patches[p].start; f < patches[p].start + patches[p].size; ++f) {
                if (isG[own[f]]) haloGhostBF[side].push_back(f - nIF);
                if (isS[own[f]]) haloSendBF[side].push_back(f - nIF);
            }
        }
    }
}

// ---------------------------------------------------------------------------
// fields
// ---------------------------------------------------------------------------
enum SnKind { SN_GENERIC = 0, SN_ZERO = 1, SN_GRADIENT = 2, SN_SYMM = 3 };

struct VolField {
    int nc = 1;
    dvec in, bf;    // nC*nc, nBF*nc
    ivec snKind;    // per patch (empty => generic)
    dvec grad;      // nBF*nc, gradient() of fixedGradient patches
    VolField() {}
    VolField(const Mesh& m, int ncomp) : nc(ncomp), in((size_t)m.nC * ncomp, 0.0), bf((size_t)m.nBF() * ncomp, 0.0) {}
};
struct SurfField {
    int nc = 1;
    dvec v;  // nF*nc
    SurfField() {}
    SurfField(const Mesh& m, int ncomp) : nc(ncomp), v((size_t)m.nF * ncomp, 0.0) {}
};

// L0: fvPatchField::snGrad() = deltaCoeffs*(*this - patchInternalField()),
// zeroGradient -> 0, fixedGradient -> gradient(), basicSymmetry ->
// (transform(I - 2 nn, pif) - pif)*(deltaCoeffs/2)
void patchSnGrad(const Mesh& m, const VolField& f, int patch, dvec& sn /*nBF*nc*/) {
    const PatchInfo& p = m.patches[patch];
    if (!m.patchHasFields(patch)) return;
    int kind = f.snKind.empty() ? SN_GENERIC : f.snKind[patch];
    // a SCALAR field on a symmetryPlane / symmetry / wedge patch carries the patch's own field type whatever it was created with
    // (L0: fvPatchField::New lets the constraint type win; basicSymmetry / wedge are specialised for scalars): snGrad = 0.  This is
    // what vF.component(d) of the 2-D GaussVolPoint vector gradient [GaussVolPointBase.C L79-87] and thermo.rho() see there.
    if (f.nc == 1 && (p.type == PATCH_SYMMETRYPLANE || p.type == PATCH_SYMMETRY || p.type == PATCH_WEDGE)) kind = SN_ZERO;
    for (int gf = p.start; gf < p.start + p.size; ++gf) {
        const int b = gf - m.nIF, o = m.own[gf];
        if (kind == SN_ZERO) {
            for (int k = 0; k < f.nc; ++k) sn[(size_t)b * f.nc + k] = 0.0;
        } else if (kind == SN_GRADIENT) {
            for (int k = 0; k < f.nc; ++k) sn[(size_t)b * f.nc + k] = f.grad[(size_t)b * f.nc + k];
        } else if (kind == SN_SYMM && f.nc == 3) {
            double n[3];
            m.symmNormal(patch, gf, n);
            // symmTensor T = I - 2.0*sqr(nHat); transform(T, v) = T & v
            double T[9];
            for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) T[3 * i + j] = (i == j ? 1.0 : 0.0) - 2.0 * (n[i] * n[j]);
            const double* v = &f.in[3 * (size_t)o];
            for (int i = 0; i < 3; ++i) {
                const double tv = T[3 * i] * v[0] + T[3 * i + 1] * v[1] + T[3 * i + 2] * v[2];
                sn[(size_t)b * 3 + i] = (tv - v[i]) * (m.delta[gf] / 2.0);
            }
        } else {
            for (int k = 0; k < f.nc; ++k)
                sn[(size_t)b * f.nc + k] = m.delta[gf] * (f.bf[(size_t)b * f.nc + k] - f.in[(size_t)o * f.nc + k]);
        }
    }
}

dvec allPatchSnGrad(const Mesh& m, const VolField& f) {
    dvec sn((size_t)m.nBF() * f.nc, 0.0);
    for (size_t p = 0; p < m.patches.size(); ++p) patchSnGrad(m, f, (int)p, sn);
    return sn;
}

// L0: linearInterpolate = surfaceInterpolationScheme::interpolate(vf, weights):
// sf = lambda*(vf[P] - vf[N]) + vf[N]; boundary = patch value
SurfField linearInterpolate(const Mesh& m, const VolField& f) {
    SurfField s(m, f.nc);
    for (int fc = 0; fc < m.nIF; ++fc)
        for (int k = 0; k < f.nc; ++k) {
            const double a = f.in[(size_t)m.own[fc] * f.nc + k], b = f.in[(size_t)m.nei[fc] * f.nc + k];
            s.v[(size_t)fc * f.nc + k] = m.w[fc] * (a - b) + b;
        }
    for (size_t p = 0; p < m.patches.size(); ++p) {
        if (!m.patchHasFields((int)p)) continue;
        for (int fc = m.patches[p].start; fc < m.patches[p].start + m.patches[p].size; ++fc)
            for (int k = 0; k < f.nc; ++k) s.v[(size_t)fc * f.nc + k] = f.bf[(size_t)(fc - m.nIF) * f.nc + k];
    }
    return s;
}

// qgdFlux's fvc::flux(flux, psi, name) branch [QGDInterpolate.H L86-104] with the entry `Gauss upwind` (L0, restated from knowledge):
// gaussConvectionScheme::flux = faceFlux * tinterpScheme().interpolate(vf); upwind is a limitedSurfaceInterpolationScheme whose
// weights() are pos0(faceFlux); surfaceInterpolationScheme::interpolate(vf, lambdas) forms lambda*(vf[P] - vf[N]) + vf[N] on internal
// faces and takes the patch value on (non-coupled) patch faces.  Returns the interpolated field psi_f (the caller multiplies by the flux).
SurfField upwindInterpolate(const Mesh& m, const SurfField& faceFlux, const VolField& f) {
    SurfField s(m, f.nc);
    for (int fc = 0; fc < m.nIF; ++fc) {
        const double lambda = faceFlux.v[fc] >= 0.0 ? 1.0 : 0.0;   // pos0
        for (int k = 0; k < f.nc; ++k)
            s.v[(size_t)fc * f.nc + k] = lambda * (f.in[(size_t)m.own[fc] * f.nc + k] - f.in[(size_t)m.nei[fc] * f.nc + k]) + f.in[(size_t)m.nei[fc] * f.nc + k];
    }
    for (size_t p = 0; p < m.patches.size(); ++p) {
        if (!m.patchHasFields((int)p)) continue;
        for (int fc = m.patches[p].start; fc < m.patches[p].start + m.patches[p].size; ++fc)
            for (int k = 0; k < f.nc; ++k) s.v[(size_t)fc * f.nc + k] = f.bf[(size_t)(fc - m.nIF) * f.nc + k];
    }
    return s;
}

// L0: fvc::snGrad (uncorrected): nonOrthDeltaCoeffs*(vf[N]-vf[P]); boundary = patch snGrad
SurfField fvcSnGrad(const Mesh& m, const VolField& f) {
    SurfField s(m, f.nc);
    for (int fc = 0; fc < m.nIF; ++fc)
        for (int k = 0; k < f.nc; ++k)
            s.v[(size_t)fc * f.nc + k] = m.nonOrthDelta[fc] * (f.in[(size_t)m.nei[fc] * f.nc + k] - f.in[(size_t)m.own[fc] * f.nc + k]);
    dvec sn = allPatchSnGrad(m, f);
    for (int fc = m.nIF; fc < m.nF; ++fc)
        for (int k = 0; k < f.nc; ++k) s.v[(size_t)fc * f.nc + k] = sn[(size_t)(fc - m.nIF) * f.nc + k];
    return s;
}

VolField component(const Mesh& m, const VolField& f, int d) {
    VolField c(m, 1);
    for (int i = 0; i < m.nC; ++i) c.in[i] = f.in[(size_t)i * f.nc + d];
    for (int i = 0; i < m.nBF(); ++i) c.bf[i] = f.bf[(size_t)i * f.nc + d];
    return c;  // calculated patches: generic snGrad
}

// L0: volPointInterpolation::interpolate (internal + boundary override; point
// constraints are no-ops for the calculated point patches used here)
dvec volPointInterpolate(const Mesh& m, const VolField& f) {
    const int nc = f.nc;
    dvec pf((size_t)m.nP * nc, 0.0);
    for (int pt = 0; pt < m.nP; ++pt) {
        if (m.isPatchPoint[pt]) continue;
        const ivec& pc = m.pointCells[pt];
        const dvec& pw = m.pointWeights[pt];
        for (size_t i = 0; i < pc.size(); ++i)
            for (int k = 0; k < nc; ++k) pf[(size_t)pt * nc + k] += pw[i] * f.in[(size_t)pc[i] * nc + k];
    }
    // flatBoundaryField: zero on empty and coupled patches
    dvec bv((size_t)m.nBF() * nc, 0.0);
    for (size_t p = 0; p < m.patches.size(); ++p) {
        if (m.patches[p].type == PATCH_EMPTY || m.coupled((int)p)) continue;
        for (int fc = m.patches[p].start; fc < m.patches[p].start + m.patches[p].size; ++fc)
            for (int k = 0; k < nc; ++k) bv[(size_t)(fc - m.nIF) * nc + k] = f.bf[(size_t)(fc - m.nIF) * nc + k];
    }
    for (size_t i = 0; i < m.bndMeshPoints.size(); ++i) {
        const int pt = m.bndMeshPoints[i];
        if (!m.isPatchPoint[pt]) continue;
        const ivec& pfc = m.bndPointFaces[i];
        const dvec& pw = m.bndPointWeights[i];
        for (int k = 0; k < nc; ++k) pf[(size_t)pt * nc + k] = 0.0;
        for (size_t j = 0; j < pfc.size(); ++j) {
            if (!m.isPatchFace[pfc[j]]) continue;
            for (int k = 0; k < nc; ++k) pf[(size_t)pt * nc + k] += pw[j] * bv[(size_t)pfc[j] * nc + k];
        }
    }
    // pointConstraints::constrain (see Mesh::pointConstraintOps): vectors and tensors only
    if (nc == 3 || nc == 9) {
        for (const auto& po : m.pointOps) {
            if (!m.isPatchPoint[po.first]) continue;
            double* x = &pf[(size_t)po.first * nc];
            for (const Mesh::PointOp& op : po.second) {
                double tx[9];
                if (nc == 3) { for (int i = 0; i < 3; ++i) tx[i] = op.T[3 * i] * x[0] + op.T[3 * i + 1] * x[1] + op.T[3 * i + 2] * x[2]; }
                else {   // transform(T, A) = T & A & T^T (L0 transform.H)
                    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
                        double acc = 0.0;
                        for (int l = 0; l < 3; ++l) acc += (op.T[3 * i] * x[l] + op.T[3 * i + 1] * x[3 + l] + op.T[3 * i + 2] * x[6 + l]) * op.T[3 * j + l];
                        tx[3 * i + j] = acc;
                    }
                }
                for (int k = 0; k < nc; ++k) x[k] = op.kind == 0 ? (x[k] + tx[k]) / 2.0 : tx[k];
            }
        }
    }
    return pf;
}

// ---------------------------------------------------------------------------
// fvsc stencils
// ---------------------------------------------------------------------------
struct Stencil {
    const Mesh& m;
    dvec nf;  // fvscStencil::nf_ = Sf/magSf [fvscStencil.C:126-129]
    explicit Stencil(const Mesh& mesh) : m(mesh), nf(3 * (size_t)mesh.nF, 0.0) {
        for (int f = 0; f < m.nF; ++f)
            if (m.magSf[f] > 0) for (int k = 0; k < 3; ++k) nf[3 * (size_t)f + k] = m.Sf[3 * (size_t)f + k] / m.magSf[f];
    }
    virtual ~Stencil() {}
    virtual SurfField gradS(const VolField& f) = 0;
    virtual SurfField gradV(const VolField& f) = 0;
    virtual SurfField divV(const VolField& f) = 0;
    virtual SurfField divT(const VolField& f) = 0;

    // nf * snGrad (outer product) and nf & snGrad
    SurfField nfOuterSnGrad(const VolField& f) const {
        SurfField sn = fvcSnGrad(m, f);
        SurfField r(m, 3 * f.nc);
        for (int fc = 0; fc < m.nF; ++fc)
            for (int i = 0; i < 3; ++i)
                for (int k = 0; k < f.nc; ++k)
                    r.v[(size_t)fc * 3 * f.nc + i * f.nc + k] = nf[3 * (size_t)fc + i] * sn.v[(size_t)fc * f.nc + k];
        return r;
    }
    SurfField nfDotSnGrad(const VolField& f) const {
        SurfField sn = fvcSnGrad(m, f);
        const int no = f.nc / 3;  // vector -> scalar, tensor -> vector
        SurfField r(m, no);
        for (int fc = 0; fc < m.nF; ++fc)
            for (int j = 0; j < no; ++j)
                r.v[(size_t)fc * no + j] = nf[3 * (size_t)fc] * sn.v[(size_t)fc * f.nc + j]
                                         + nf[3 * (size_t)fc + 1] * sn.v[(size_t)fc * f.nc + no + j]
                                         + nf[3 * (size_t)fc + 2] * sn.v[(size_t)fc * f.nc + 2 * no + j];
        return r;
    }
};

// [reducedFaceNormalStencil.C:69-108]
struct Reduced : Stencil {
    explicit Reduced(const Mesh& mesh) : Stencil(mesh) {}
    SurfField gradS(const VolField& f) override { return nfOuterSnGrad(f); }
    SurfField gradV(const VolField& f) override { return nfOuterSnGrad(f); }
    SurfField divV(const VolField& f) override { return nfDotSnGrad(f); }
    SurfField divT(const VolField& f) override { return nfDotSnGrad(f); }
};

// leastSquares, serial branch
struct LeastSquares : Stencil {
    std::vector<ivec> neighbourCells;   // [FindNb.C:48-86]
    std::vector<std::vector<double> > GdfAll, wf2All;  // [CalcW.C:152-153]
    ivec internalDegFaces;
    explicit LeastSquares(const Mesh& mesh) : Stencil(mesh) { findNeighbours(); calculateWeights(); }

    void findNeighbours() {  // [extendedFaceStencilFindNeighbours.C:41-86]
        neighbourCells.assign(m.nIF, ivec());
        for (int facei = 0; facei < m.nIF; ++facei) {
            ivec nb;
            for (int q = m.fOff[facei]; q < m.fOff[facei + 1]; ++q) {
                const ivec& pc = m.pointCells[m.fPts[q]];
                for (int celli : pc) {
                    bool contained = false;
                    for (int x : nb) if (x == celli) contained = true;
                    if (!contained) nb.push_back(celli);
                }
            }
            neighbourCells[facei] = nb;
        }
    }
    void calculateWeights() {  // [extendedFaceStencilCalculateWeights.C:43-155]
        GdfAll.assign(m.nIF, dvec()); wf2All.assign(m.nIF, dvec());
        for (int facei = 0; facei < m.nIF; ++facei) {
            const ivec& nb = neighbourCells[facei];
            dvec df(3 * nb.size()), wf2(nb.size());
            double G[6] = {0, 0, 0, 0, 0, 0};  // symmTensor xx xy xz yy yz zz
            const double* Cf = &m.Cf[3 * (size_t)facei];
            for (size_t i = 0; i < nb.size(); ++i) {
                double* d = &df[3 * i];
                for (int k = 0; k < 3; ++k) d[k] = m.C[3 * (size_t)nb[i] + k] - Cf[k];
                wf2[i] = 1 / (d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
                const double a[6] = {d[0] * d[0], d[0] * d[1], d[0] * d[2], d[1] * d[1], d[1] * d[2], d[2] * d[2]};
                for (int k = 0; k < 6; ++k) G[k] += a[k] * wf2[i];
            }
            double G0[6] = {0, 0, 0, 0, 0, 0};
            if (std::fabs(G[0]) < SMALL) G0[0] = 1;
            if (std::fabs(G[3]) < SMALL) G0[3] = 1;
            if (std::fabs(G[5]) < SMALL) G0[5] = 1;
            for (int k = 0; k < 6; ++k) G[k] = G[k] + G0[k];
            // det(symmTensor) (L0 formula)
            const double detG = G[0] * G[3] * G[5] + G[1] * G[4] * G[2] + G[2] * G[1] * G[4]
                              - G[0] * G[4] * G[4] - G[1] * G[1] * G[5] - G[2] * G[3] * G[2];
            // [CalcW.C L136-145]; the user's faceSet joins the same list [leastSquaresStencil.C L84-117]
            if (detG < 1 || std::find(m.userDegenerateFaces.begin(), m.userDegenerateFaces.end(), facei) != m.userDegenerateFaces.end()) {
                internalDegFaces.push_back(facei);
            } else {
                // inv(symmTensor) = cofactors/det (L0)
                const double I[6] = {
                    (G[3] * G[5] - G[4] * G[4]) / detG, (G[2] * G[4] - G[1] * G[5]) / detG,
                    (G[1] * G[4] - G[2] * G[3]) / detG, (G[0] * G[5] - G[2] * G[2]) / detG,
                    (G[1] * G[2] - G[0] * G[4]) / detG, (G[0] * G[3] - G[1] * G[1]) / detG};
                for (int k = 0; k < 6; ++k) G[k] = I[k] - G0[k];
            }
            for (size_t i = 0; i < nb.size(); ++i) {  // df = G & df
                const double d[3] = {df[3 * i], df[3 * i + 1], df[3 * i + 2]};
                df[3 * i] = G[0] * d[0] + G[1] * d[1] + G[2] * d[2];
                df[3 * i + 1] = G[1] * d[0] + G[3] * d[1] + G[4] * d[2];
                df[3 * i + 2] = G[2] * d[0] + G[4] * d[1] + G[5] * d[2];
            }
            GdfAll[facei] = df; wf2All[facei] = wf2;
        }
    }
    SurfField gradS(const VolField& iF) override {  // [extendedFaceStencilScalarGrad.C:50-114]
        SurfField sF = linearInterpolate(m, iF);
        SurfField sngF = fvcSnGrad(m, iF);
        SurfField g(m, 3);  // 0.0*nf*sngF
        for (int facei = 0; facei < m.nIF; ++facei) {
            double gf[3] = {0, 0, 0};
            const ivec& nb = neighbourCells[facei];
            for (size_t i = 0; i < nb.size(); ++i) {
                const double dphi = iF.in[nb[i]] - sF.v[facei];
                for (int k = 0; k < 3; ++k) gf[k] = gf[k] + (wf2All[facei][i] * GdfAll[facei][3 * i + k]) * dphi;
            }
            for (int k = 0; k < 3; ++k) g.v[3 * (size_t)facei + k] = gf[k];
        }
        for (int d : internalDegFaces)
            for (int k = 0; k < 3; ++k) g.v[3 * (size_t)d + k] = sngF.v[d] * nf[3 * (size_t)d + k];
        for (size_t p = 0; p < m.patches.size(); ++p) {
            const int t = m.patches[p].type;
            const bool constraint = t == PATCH_EMPTY || t == PATCH_WEDGE || t == PATCH_CYCLIC || t == PATCH_HALO ||
                                    t == PATCH_SYMMETRY || t == PATCH_SYMMETRYPLANE;
            if (constraint) continue;
            dvec sn((size_t)m.nBF(), 0.0);
            patchSnGrad(m, iF, (int)p, sn);
            for (int fc = m.patches[p].start; fc < m.patches[p].start + m.patches[p].size; ++fc)
                for (int k = 0; k < 3; ++k) g.v[3 * (size_t)fc + k] = nf[3 * (size_t)fc + k] * sn[fc - m.nIF];
        }
        return g;
    }
    SurfField gradV(const VolField& f) override {  // [leastSquaresStencil.C:145-196]
        SurfField c0 = gradS(component(m, f, 0)), c1 = gradS(component(m, f, 1)), c2 = gradS(component(m, f, 2));
        SurfField g(m, 9);
        for (int fc = 0; fc < m.nF; ++fc)
            for (int i = 0; i < 3; ++i) {
                g.v[9 * (size_t)fc + 3 * i + 0] = c0.v[3 * (size_t)fc + i];
                g.v[9 * (size_t)fc + 3 * i + 1] = c1.v[3 * (size_t)fc + i];
                g.v[9 * (size_t)fc + 3 * i + 2] = c2.v[3 * (size_t)fc + i];
            }
        return g;
    }
    SurfField divV(const VolField& f) override {  // [leastSquaresStencil.C:204-228]
        SurfField c0 = gradS(component(m, f, 0)), c1 = gradS(component(m, f, 1)), c2 = gradS(component(m, f, 2));
        SurfField d(m, 1);
        for (int fc = 0; fc < m.nF; ++fc) d.v[fc] = c0.v[3 * (size_t)fc] + c1.v[3 * (size_t)fc + 1] + c2.v[3 * (size_t)fc + 2];
        return d;
    }
    SurfField divT(const VolField& f) override {  // [leastSquaresStencil.C:236-275]
        SurfField g[9];
        for (int k = 0; k < 9; ++k) g[k] = gradS(component(m, f, k));
        SurfField d(m, 3);
        for (int fc = 0; fc < m.nF; ++fc)
            for (int j = 0; j < 3; ++j)
                d.v[3 * (size_t)fc + j] = g[j].v[3 * (size_t)fc] + g[3 + j].v[3 * (size_t)fc + 1] + g[6 + j].v[3 * (size_t)fc + 2];
        return d;
    }
};

// GaussVolPoint: 1-D / 2-D / 3-D bases + dispatcher
struct GaussVolPoint : Stencil {
    // ---- 3-D [GaussVolPointBase3D.C:41-159] --------------------------------
    ivec qf, tf, of;                        // internal quad / tri / other faces
    std::vector<ivec> bqf, btf, bof;        // per patch (patch-local indices)
    std::vector<dvec> aq[3], at[3];         // per face: 6 / 5 coefficients
    dvec vq, vt;
    std::vector<std::vector<dvec> > baq[3], bat[3];
    std::vector<dvec> bvq, bvt, bmvON;
    // ---- 2-D [GaussVolPointBase2D.C:44-293] --------------------------------
    dvec c1, c2, c3, c4, mv42, mv13;
    ivec ip3, ip1, ic4, ic2;
    int ie1 = -1, ie2 = -1, ie3 = -1;
    double e1[3] = {1, 0, 0}, e2[3] = {0, 1, 0};
    ivec ordinaryPatches;
    std::vector<ivec> ip3e, ip1e, ic4e;
    std::vector<dvec> c1e, c2e, c3e, c4e, mv42e, mv13e;

    explicit GaussVolPoint(const Mesh& mesh) : Stencil(mesh) { init2D(); init3D(); }

    void init3D() {
        const int np = (int)m.patches.size();
        bqf.assign(np, ivec()); btf.assign(np, ivec()); bof.assign(np, ivec());
        for (int i = 0; i < m.nIF; ++i) {
            if (m.fsize(i) == 3) tf.push_back(i);
            else if (m.fsize(i) == 4) qf.push_back(i);
            else of.push_back(i);
        }
        bmvON.assign(np, dvec());
        std::vector<dvec> vO(np), vN(np);
        for (int ip = 0; ip < np; ++ip) {
            if (!m.patchHasFields(ip)) continue;
            const PatchInfo& p = m.patches[ip];
            for (int i = 0; i < p.size; ++i) {
                const int n = m.fsize(p.start + i);
                if (n == 3) btf[ip].push_back(i); else if (n == 4) bqf[ip].push_back(i); else bof[ip].push_back(i);
            }
            vO[ip].resize(3 * (size_t)p.size); vN[ip].resize(3 * (size_t)p.size); bmvON[ip].resize(p.size);
            for (int i = 0; i < p.size; ++i) {
                const int gf = p.start + i;
                double d[3];
                for (int k = 0; k < 3; ++k) {
                    vO[ip][3 * i + k] = m.C[3 * (size_t)m.own[gf] + k];
                    // vN = vO + 2.0*(Cf - vO)  [3D.C:142-147]
                    vN[ip][3 * i + k] = vO[ip][3 * i + k] + 2.0 * (m.Cf[3 * (size_t)gf + k] - vO[ip][3 * i + k]);
                    d[k] = vO[ip][3 * i + k] - vN[ip][3 * i + k];
                }
                bmvON[ip][i] = mag3(d);
            }
        }
        const double OneBySix = (1.0 / 6.0);
        // ---- triangles [3D.C:161-318]
        for (int d = 0; d < 3; ++d) { at[d].assign(tf.size(), dvec()); bat[d].assign(np, std::vector<dvec>()); }
        vt.resize(tf.size()); bvt.assign(np, dvec());
        auto triCoeffs = [&](const double* p1, const double* p2, const double* p3, const double* own, const double* nei,
                             dvec& ax, dvec& ay, dvec& az, double& vol) {
            double a[3], b[3], cr[3], on[3];
            for (int k = 0; k < 3; ++k) { a[k] = p2[k] - p1[k]; b[k] = p3[k] - p1[k]; on[k] = own[k] - nei[k]; }
            cross3(a, b, cr);
            vol = dot3(cr, on);
            vol *= OneBySix;
            ax.resize(5); ay.resize(5); az.resize(5);
            // x: (y,z) ; y: (z,x) ; z: (x,y)  -- cyclic permutation of [3D.C:193-229]
            dvec* A[3] = {&ax, &ay, &az};
            for (int d = 0; d < 3; ++d) {
                const int u = (d + 1) % 3, v = (d + 2) % 3;  // x->(y,z), y->(z,x), z->(x,y)
                dvec& c = *A[d];
                c[0] = OneBySix * ((own[v] - nei[v]) * (p2[u] - p3[u]) + (nei[u] - own[u]) * (p2[v] - p3[v]));
                c[1] = OneBySix * ((nei[u] - own[u]) * (p3[v] - p1[v]) + (own[v] - nei[v]) * (p3[u] - p1[u]));
                c[2] = OneBySix * ((nei[u] - own[u]) * (p1[v] - p2[v]) + (own[v] - nei[v]) * (p1[u] - p2[u]));
                c[3] = OneBySix * (p1[v] * (p2[u] - p3[u]) + p2[v] * (p3[u] - p1[u]) + p3[v] * (p1[u] - p2[u]));
                c[4] = -c[3];
            }
        };
        for (size_t i = 0; i < tf.size(); ++i) {
            const int f = tf[i];
            const int* q = m.fp(f);
            triCoeffs(&m.pts[3 * (size_t)q[0]], &m.pts[3 * (size_t)q[1]], &m.pts[3 * (size_t)q[2]],
                      &m.C[3 * (size_t)m.own[f]], &m.C[3 * (size_t)m.nei[f]], at[0][i], at[1][i], at[2][i], vt[i]);
        }
        // ---- quads [3D.C:320-476]
        for (int d = 0; d < 3; ++d) { aq[d].assign(qf.size(), dvec()); baq[d].assign(np, std::vector<dvec>()); }
        vq.resize(qf.size()); bvq.assign(np, dvec());
        auto quadCoeffs = [&](const double* p1, const double* p2, const double* p3, const double* p4, const double* own,
                              const double* nei, dvec& ax, dvec& ay, dvec& az, double& vol) {
            double a[3], b[3], on[3], cr[3];
            for (int k = 0; k < 3; ++k) { a[k] = p3[k] - p1[k]; b[k] = p4[k] - p2[k]; on[k] = own[k] - nei[k]; }
            cross3(b, on, cr);
            vol = dot3(a, cr);
            vol *= OneBySix;
            ax.resize(6); ay.resize(6); az.resize(6);
            dvec* A[3] = {&ax, &ay, &az};
            for (int d = 0; d < 3; ++d) {
                const int u = (d + 1) % 3, v = (d + 2) % 3;
                dvec& c = *A[d];
                c[0] = OneBySix * ((nei[u] - own[u]) * (p2[v] - p4[v]) - (nei[v] - own[v]) * (p2[u] - p4[u]));
                c[1] = OneBySix * ((nei[u] - own[u]) * (p3[v] - p1[v]) - (nei[v] - own[v]) * (p3[u] - p1[u]));
                c[5] = OneBySix * ((p1[u] - p3[u]) * (p2[v] - p4[v]) - (p1[v] - p3[v]) * (p2[u] - p4[u]));
                c[2] = -c[0]; c[3] = -c[1]; c[4] = -c[5];
            }
        };
        for (size_t i = 0; i < qf.size(); ++i) {
            const int f = qf[i];
            const int* q = m.fp(f);
            quadCoeffs(&m.pts[3 * (size_t)q[0]], &m.pts[3 * (size_t)q[1]], &m.pts[3 * (size_t)q[2]], &m.pts[3 * (size_t)q[3]],
                       &m.C[3 * (size_t)m.own[f]], &m.C[3 * (size_t)m.nei[f]], aq[0][i], aq[1][i], aq[2][i], vq[i]);
        }
        for (int ip = 0; ip < np; ++ip) {
            for (int d = 0; d < 3; ++d) { bat[d][ip].assign(btf[ip].size(), dvec()); baq[d][ip].assign(bqf[ip].size(), dvec()); }
            bvt[ip].resize(btf[ip].size()); bvq[ip].resize(bqf[ip].size());
            for (size_t k = 0; k < btf[ip].size(); ++k) {
                const int li = btf[ip][k], gf = m.patches[ip].start + li;
                const int* q = m.fp(gf);
                triCoeffs(&m.pts[3 * (size_t)q[0]], &m.pts[3 * (size_t)q[1]], &m.pts[3 * (size_t)q[2]],
                          &vO[ip][3 * li], &vN[ip][3 * li], bat[0][ip][k], bat[1][ip][k], bat[2][ip][k], bvt[ip][k]);
            }
            for (size_t k = 0; k < bqf[ip].size(); ++k) {
                const int li = bqf[ip][k], gf = m.patches[ip].start + li;
                const int* q = m.fp(gf);
                quadCoeffs(&m.pts[3 * (size_t)q[0]], &m.pts[3 * (size_t)q[1]], &m.pts[3 * (size_t)q[2]], &m.pts[3 * (size_t)q[3]],
                           &vO[ip][3 * li], &vN[ip][3 * li], baq[0][ip][k], baq[1][ip][k], baq[2][ip][k], bvq[ip][k]);
            }
        }
    }

    // macro dfdxif [3D.C:488-513]: out[face][ocmpt] += (sum_k a_k phi_k)/V
    void dfdxif(const VolField& vf, const dvec& pf, SurfField& out, const ivec& fi, const dvec& vi,
                const std::vector<dvec>& ai, int icmpt, int ocmpt) const {
        if (fi.empty()) return;
        const int iown = (int)ai[0].size() - 1, inei = iown - 1;
        const int nc = vf.nc, no = out.nc;
        for (size_t i = 0; i < fi.size(); ++i) {
            const int facei = fi[i];
            double d = vf.in[(size_t)m.nei[facei] * nc + icmpt] * ai[i][inei];
            d += vf.in[(size_t)m.own[facei] * nc + icmpt] * ai[i][iown];
            const int* q = m.fp(facei);
            for (int k = 0; k < m.fsize(facei); ++k) d += pf[(size_t)q[k] * nc + icmpt] * ai[i][k];
            out.v[(size_t)facei * no + ocmpt] += (d / vi[i]);
        }
    }
    // macro dfdxbf [3D.C:515-539]
    void dfdxbf(const VolField& vf, const dvec& pf, int patchi, SurfField& out, const ivec& bfi, const dvec& bvfi,
                const std::vector<dvec>& bai, const dvec& psin, int icmpt, int ocmpt) const {
        if (bfi.empty()) return;
        const int iown = (int)bai[0].size() - 1, inei = iown - 1;
        const int nc = vf.nc, no = out.nc;
        for (size_t i = 0; i < bfi.size(); ++i) {
            const int gf = m.patches[patchi].start + bfi[i];
            const int b = gf - m.nIF;
            double d = psin[(size_t)b * nc + icmpt] * bai[i][inei];
            d += vf.in[(size_t)m.own[gf] * nc + icmpt] * bai[i][iown];  // psio = patchInternalField
            const int* q = m.fp(gf);
            for (int k = 0; k < m.fsize(gf); ++k) d += pf[(size_t)q[k] * nc + icmpt] * bai[i][k];
            out.v[(size_t)gf * no + ocmpt] += d / bvfi[i];
        }
    }
    // psin = boundaryField + snGrad*bmvON*0.5 [3D.C:790-793]
    dvec psiN(const VolField& vf) const {
        dvec sn = allPatchSnGrad(m, vf);
        dvec r((size_t)m.nBF() * vf.nc, 0.0);
        for (size_t ip = 0; ip < m.patches.size(); ++ip) {
            if (!m.patchHasFields((int)ip)) continue;
            for (int i = 0; i < m.patches[ip].size; ++i) {
                const int b = m.patches[ip].start + i - m.nIF;
                for (int k = 0; k < vf.nc; ++k)
                    r[(size_t)b * vf.nc + k] = vf.bf[(size_t)b * vf.nc + k] + sn[(size_t)b * vf.nc + k] * bmvON[ip][i] * 0.5;
            }
        }
        return r;
    }
    // apply a table of (direction, icmpt, ocmpt) on quads then on triangles,
    // copy dfdn on other faces; same for every patch
    struct Term { int dir, ic, oc; };
    void apply3D(const VolField& vf, SurfField& out, const SurfField& dfdn, const std::vector<Term>& quadTerms,
                 const std::vector<Term>& triTermsInternal, const std::vector<Term>& triTermsBoundary) const {
        dvec pf = volPointInterpolate(m, vf);
        for (const Term& t : quadTerms) dfdxif(vf, pf, out, qf, vq, aq[t.dir], t.ic, t.oc);
        for (const Term& t : triTermsInternal) dfdxif(vf, pf, out, tf, vt, at[t.dir], t.ic, t.oc);
        for (int f : of) for (int k = 0; k < out.nc; ++k) out.v[(size_t)f * out.nc + k] = dfdn.v[(size_t)f * out.nc + k];
        dvec psin = psiN(vf);
        for (size_t ip = 0; ip < m.patches.size(); ++ip) {
            if (!m.patchHasFields((int)ip)) continue;
            for (const Term& t : quadTerms) dfdxbf(vf, pf, (int)ip, out, bqf[ip], bvq[ip], baq[t.dir][ip], psin, t.ic, t.oc);
            for (const Term& t : triTermsBoundary) dfdxbf(vf, pf, (int)ip, out, btf[ip], bvt[ip], bat[t.dir][ip], psin, t.ic, t.oc);
            for (int li : bof[ip]) {
                const int gf = m.patches[ip].start + li;
                for (int k = 0; k < out.nc; ++k) out.v[(size_t)gf * out.nc + k] = dfdn.v[(size_t)gf * out.nc + k];
            }
        }
    }

    void init2D() {
        if (m.nGeomD != 2) return;
        const int nIF = m.nIF, np = (int)m.patches.size();
        c1.assign(nIF, 0); c2.assign(nIF, 0); c3.assign(nIF, 0); c4.assign(nIF, 0);
        mv42.assign(nIF, 1); mv13.assign(nIF, 1);
        ip3.assign(nIF, -1); ip1.assign(nIF, -1); ic4.assign(nIF, -1); ic2.assign(nIF, -1);
        for (int d = 0; d < 3; ++d) if (m.geomD[d] < 1) ie3 = d;
        if (ie3 == 0) { e1[0] = 0; e1[1] = 1; e1[2] = 0; e2[0] = 0; e2[1] = 0; e2[2] = 1; ie1 = 1; ie2 = 2; }
        if (ie3 == 1) { e1[0] = 1; e1[1] = 0; e1[2] = 0; e2[0] = 0; e2[1] = 0; e2[2] = 1; ie1 = 0; ie2 = 2; }
        if (ie3 == 2) { e1[0] = 1; e1[1] = 0; e1[2] = 0; e2[0] = 0; e2[1] = 1; e2[2] = 0; ie1 = 0; ie2 = 1; }
        int p1 = -1, p3 = -1;
        auto coeffs = [&](const double* v42, const double* v13, double& m42, double& m13, double& C1, double& C2, double& C3, double& C4) {
            m42 = mag3(v42); m13 = mag3(v13);
            double a[3], b[3];
            for (int k = 0; k < 3; ++k) { a[k] = v42[k] / m42; b[k] = v13[k] / m13; }
            const double cosa1 = dot3(a, e1), cosa2 = dot3(b, e1), sina1 = dot3(a, e2), sina2 = dot3(b, e2);
            const double den = sina2 * cosa1 - sina1 * cosa2;
            C1 = sina2 / den; C2 = sina1 / den; C3 = cosa1 / den; C4 = cosa2 / den;
        };
        for (int f = 0; f < nIF; ++f) {  // [2D.C:122-169]
            const int c2i = m.nei[f], c4i = m.own[f];
            const int* q = m.fp(f);
            const int n = m.fsize(f);
            for (int k = 0; k < n; ++k) if (m.pts[3 * (size_t)q[k] + ie3] >= m.C[3 * (size_t)c2i + ie3]) { p1 = q[k]; break; }
            for (int k = 0; k < n; ++k) if (m.pts[3 * (size_t)q[k] + ie3] >= m.C[3 * (size_t)c2i + ie3]) { if (p1 != q[k]) { p3 = q[k]; break; } }
            ic2[f] = c2i; ic4[f] = c4i; ip3[f] = p3; ip1[f] = p1;
            double v42[3], v13[3];
            for (int k = 0; k < 3; ++k) { v42[k] = m.C[3 * (size_t)c2i + k] - m.C[3 * (size_t)c4i + k]; v13[k] = m.pts[3 * (size_t)p3 + k] - m.pts[3 * (size_t)p1 + k]; }
            coeffs(v42, v13, mv42[f], mv13[f], c1[f], c2[f], c3[f], c4[f]);
        }
        ip3e.assign(np, ivec()); ip1e.assign(np, ivec()); ic4e.assign(np, ivec());
        c1e.assign(np, dvec()); c2e.assign(np, dvec()); c3e.assign(np, dvec()); c4e.assign(np, dvec());
        mv42e.assign(np, dvec()); mv13e.assign(np, dvec());
        for (int ip = 0; ip < np; ++ip) {  // [2D.C:172-291]
            const int t = m.patches[ip].type;
            if (t == PATCH_EMPTY || t == PATCH_WEDGE) continue;
            if (t == PATCH_CYCLIC) continue;  // coupled and not processor
            ordinaryPatches.push_back(ip);
            const int n = m.patches[ip].size;
            ic4e[ip].resize(n); ip3e[ip].resize(n); ip1e[ip].resize(n);
            c1e[ip].resize(n); c2e[ip].resize(n); c3e[ip].resize(n); c4e[ip].resize(n); mv42e[ip].resize(n); mv13e[ip].resize(n);
            for (int i = 0; i < n; ++i) {
                const int gf = m.patches[ip].start + i, c4i = m.own[gf];
                ic4e[ip][i] = c4i;
                double v42[3], v13[3];
                for (int k = 0; k < 3; ++k) v42[k] = 2.0 * (m.Cf[3 * (size_t)gf + k] - m.C[3 * (size_t)c4i + k]);
                int q1 = -1, q3 = -1;
                const int* q = m.fp(gf);
                const int nn = m.fsize(gf);
                for (int k = 0; k < nn; ++k) if (m.pts[3 * (size_t)q[k] + ie3] >= m.C[3 * (size_t)c4i + ie3]) { q1 = q[k]; break; }
                for (int k = 0; k < nn; ++k) if (m.pts[3 * (size_t)q[k] + ie3] >= m.C[3 * (size_t)c4i + ie3]) { if (q1 != q[k]) { q3 = q[k]; break; } }
                ip3e[ip][i] = q3; ip1e[ip][i] = q1;
                for (int k = 0; k < 3; ++k) v13[k] = m.pts[3 * (size_t)q3 + k] - m.pts[3 * (size_t)q1 + k];
                coeffs(v42, v13, mv42e[ip][i], mv13e[ip][i], c1e[ip][i], c2e[ip][i], c3e[ip][i], c4e[ip][i]);
            }
        }
    }
    // psi2 = boundaryField + snGrad*mv42e*0.5 [2D.C:343-346]
    dvec psi2(const VolField& f) const {
        dvec sn = allPatchSnGrad(m, f);
        dvec r((size_t)m.nBF() * f.nc, 0.0);
        for (int ip : ordinaryPatches)
            for (int i = 0; i < m.patches[ip].size; ++i) {
                const int b = m.patches[ip].start + i - m.nIF;
                for (int k = 0; k < f.nc; ++k) r[(size_t)b * f.nc + k] = f.bf[(size_t)b * f.nc + k] + sn[(size_t)b * f.nc + k] * mv42e[ip][i] * 0.5;
            }
        return r;
    }
    void faceGrad2D(const VolField& f, SurfField& g) const {  // [2D.C:301-367]
        dvec pF = volPointInterpolate(m, f);
        for (int fc = 0; fc < m.nIF; ++fc) {
            const double dfdn = (f.in[ic2[fc]] - f.in[ic4[fc]]) / mv42[fc];
            const double dfdt = (pF[ip3[fc]] - pF[ip1[fc]]) / mv13[fc];
            g.v[3 * (size_t)fc + ie1] = (dfdn * c1[fc] - dfdt * c2[fc]);
            g.v[3 * (size_t)fc + ie2] = (dfdt * c3[fc] - dfdn * c4[fc]);
            g.v[3 * (size_t)fc + ie3] = 0.0;
        }
        dvec p2 = psi2(f);
        for (int ip : ordinaryPatches)
            for (int i = 0; i < m.patches[ip].size; ++i) {
                const int gf = m.patches[ip].start + i, b = gf - m.nIF;
                const double dfdn = (p2[b] - f.in[ic4e[ip][i]]) / mv42e[ip][i];
                const double dfdt = (pF[ip3e[ip][i]] - pF[ip1e[ip][i]]) / mv13e[ip][i];
                g.v[3 * (size_t)gf + ie1] = (dfdn * c1e[ip][i] - dfdt * c2e[ip][i]);
                g.v[3 * (size_t)gf + ie2] = (dfdt * c3e[ip][i] - dfdn * c4e[ip][i]);
                g.v[3 * (size_t)gf + ie3] = 0.0;
            }
    }
    void faceDiv2D_V(const VolField& f, SurfField& d) const {  // [2D.C:369-438]
        dvec pF = volPointInterpolate(m, f);
        for (int fc = 0; fc < m.nIF; ++fc) {
            const double df1dn = (f.in[3 * (size_t)ic2[fc] + ie1] - f.in[3 * (size_t)ic4[fc] + ie1]) / mv42[fc];
            const double df2dn = (f.in[3 * (size_t)ic2[fc] + ie2] - f.in[3 * (size_t)ic4[fc] + ie2]) / mv42[fc];
            const double df1dt = (pF[3 * (size_t)ip3[fc] + ie1] - pF[3 * (size_t)ip1[fc] + ie1]) / mv13[fc];
            const double df2dt = (pF[3 * (size_t)ip3[fc] + ie2] - pF[3 * (size_t)ip1[fc] + ie2]) / mv13[fc];
            d.v[fc] = (df1dn * c1[fc] - df1dt * c2[fc]) + (df2dt * c3[fc] - df2dn * c4[fc]);
        }
        dvec p2 = psi2(f);
        for (int ip : ordinaryPatches)
            for (int i = 0; i < m.patches[ip].size; ++i) {
                const int gf = m.patches[ip].start + i, b = gf - m.nIF, o = ic4e[ip][i];
                const double df1dn = (p2[3 * (size_t)b + ie1] - f.in[3 * (size_t)o + ie1]) / mv42e[ip][i];
                const double df2dn = (p2[3 * (size_t)b + ie2] - f.in[3 * (size_t)o + ie2]) / mv42e[ip][i];
                const double df1dt = (pF[3 * (size_t)ip3e[ip][i] + ie1] - pF[3 * (size_t)ip1e[ip][i] + ie1]) / mv13e[ip][i];
                const double df2dt = (pF[3 * (size_t)ip3e[ip][i] + ie2] - pF[3 * (size_t)ip1e[ip][i] + ie2]) / mv13e[ip][i];
                d.v[gf] = (df1dn * c1e[ip][i] - df1dt * c2e[ip][i]) + (df2dt * c3e[ip][i] - df2dn * c4e[ip][i]);
            }
    }
    void faceDiv2D_T(const VolField& f, SurfField& d) const {  // [2D.C:440-539]
        const int i11 = ie1 * 3 + ie1, i21 = ie2 * 3 + ie1, i22 = ie2 * 3 + ie2, i12 = ie1 * 3 + ie2;
        dvec pF = volPointInterpolate(m, f);
        auto dn = [&](int fc, int c) { return (f.in[9 * (size_t)ic2[fc] + c] - f.in[9 * (size_t)ic4[fc] + c]) / mv42[fc]; };
        auto dt = [&](int fc, int c) { return (pF[9 * (size_t)ip3[fc] + c] - pF[9 * (size_t)ip1[fc] + c]) / mv13[fc]; };
        for (int fc = 0; fc < m.nIF; ++fc) {
            d.v[3 * (size_t)fc + ie1] = (dn(fc, i11) * c1[fc] - dt(fc, i11) * c2[fc]) + (dt(fc, i21) * c3[fc] - dn(fc, i21) * c4[fc]);
            d.v[3 * (size_t)fc + ie2] = (dn(fc, i12) * c1[fc] - dt(fc, i12) * c2[fc]) + (dt(fc, i22) * c3[fc] - dn(fc, i22) * c4[fc]);
        }
        dvec p2 = psi2(f);
        for (int ip : ordinaryPatches)
            for (int i = 0; i < m.patches[ip].size; ++i) {
                const int gf = m.patches[ip].start + i, b = gf - m.nIF, o = ic4e[ip][i];
                auto bn = [&](int c) { return (p2[9 * (size_t)b + c] - f.in[9 * (size_t)o + c]) / mv42e[ip][i]; };
                auto bt = [&](int c) { return (pF[9 * (size_t)ip3e[ip][i] + c] - pF[9 * (size_t)ip1e[ip][i] + c]) / mv13e[ip][i]; };
                d.v[3 * (size_t)gf + ie1] = (bn(i11) * c1e[ip][i] - bt(i11) * c2e[ip][i]) + (bt(i21) * c3e[ip][i] - bn(i21) * c4e[ip][i]);
                d.v[3 * (size_t)gf + ie2] = (bn(i12) * c1e[ip][i] - bt(i12) * c2e[ip][i]) + (bt(i22) * c3e[ip][i] - bn(i22) * c4e[ip][i]);
            }
    }

    // dispatcher [GaussVolPointBase.C:54-158] + [GaussVolPointStencil.C:71-129]
    // (the correctBoundaryConditions() of the input is the caller's business here)
    SurfField gradS(const VolField& f) override {
        if (m.nGeomD == 1) return nfOuterSnGrad(f);  // [GaussVolPointBase1D.C:49-55]
        SurfField g(m, 3);
        if (m.nGeomD == 2) { faceGrad2D(f, g); return g; }
        SurfField dfdn = nfOuterSnGrad(f);
        std::vector<Term> t = {{0, 0, 0}, {1, 0, 1}, {2, 0, 2}};  // [3D.C:750-757]
        apply3D(f, g, dfdn, t, t, t);
        return g;
    }
    SurfField gradV(const VolField& f) override {
        if (m.nGeomD == 1) return nfOuterSnGrad(f);
        SurfField g(m, 9);
        if (m.nGeomD == 2) {  // [GaussVolPointBase.C:79-116]
            SurfField gc[3] = {SurfField(m, 3), SurfField(m, 3), SurfField(m, 3)};
            for (int c = 0; c < 3; ++c) faceGrad2D(component(m, f, c), gc[c]);
            for (int fc = 0; fc < m.nF; ++fc)
                for (int i = 0; i < 3; ++i)
                    for (int c = 0; c < 3; ++c) g.v[9 * (size_t)fc + 3 * i + c] = gc[c].v[3 * (size_t)fc + i];
            return g;
        }
        SurfField dfdn = nfOuterSnGrad(f);
        std::vector<Term> q = {{0, 0, 0}, {0, 1, 1}, {0, 2, 2}, {1, 0, 3}, {1, 1, 4}, {1, 2, 5}, {2, 0, 6}, {2, 1, 7}, {2, 2, 8}};  // [3D.C:831-841]
        // interior triangles: the listing pairs (atx,0),(aty,1),(atz,2) three times [3D.C:844-854] (quirk B2)
        std::vector<Term> ti = {{0, 0, 0}, {1, 1, 1}, {2, 2, 2}, {0, 0, 3}, {1, 1, 4}, {2, 2, 5}, {0, 0, 6}, {1, 1, 7}, {2, 2, 8}};
        apply3D(f, g, dfdn, q, ti, q);  // boundary triangles use the quad pattern [3D.C:909-919]
        return g;
    }
    SurfField divV(const VolField& f) override {
        if (m.nGeomD == 1) return nfDotSnGrad(f);
        SurfField d(m, 1);
        if (m.nGeomD == 2) { faceDiv2D_V(f, d); return d; }
        SurfField dfdn = nfDotSnGrad(f);
        std::vector<Term> t = {{0, 0, 0}, {1, 1, 0}, {2, 2, 0}};  // [3D.C:553-560]
        apply3D(f, d, dfdn, t, t, t);
        return d;
    }
    SurfField divT(const VolField& f) override {
        if (m.nGeomD == 1) return nfDotSnGrad(f);
        SurfField d(m, 3);
        if (m.nGeomD == 2) { faceDiv2D_T(f, d); return d; }
        SurfField dfdn = nfDotSnGrad(f);
        std::vector<Term> t = {{0, 0, 0}, {1, 3, 0}, {2, 6, 0}, {0, 1, 1}, {1, 4, 1}, {2, 7, 1}, {0, 2, 2}, {1, 5, 2}, {2, 8, 2}};  // [3D.C:635-659]
        apply3D(f, d, dfdn, t, t, t);
        return d;
    }
};

// fvscOpName checks + fvscStencil::lookupOrNew cache [fvsc.C:47-85] [fvscStencil.C:98-118]
struct StencilCache {
    std::map<std::string, Stencil*> byName;
    ~StencilCache() { for (auto& kv : byName) delete kv.second; }
    int lookup(const Mesh& m, const std::string& word, Stencil** out) {
        if ((word == "leastSquares" || word == "leastSquaresOpt") && m.nGeomD == 3) return -4;
        if (word == "GaussVolPoint") {  // wedge patches + prism cells are fatal [fvsc.C:65-82]
            bool wedge = false;
            for (const PatchInfo& p : m.patches) wedge = wedge || (p.type == PATCH_WEDGE && p.size > 0);
            if (wedge)
                for (int c = 0; c < m.nC; ++c) {
                    int tri = 0, quad = 0, other = 0;
                    for (int f : m.cells[c]) { const int n = m.fsize(f); if (n == 3) ++tri; else if (n == 4) ++quad; else ++other; }
                    if (tri == 2 && quad == 3 && other == 0) return -4;  // prismMatcher (L0): 2 triangles + 3 quadrilaterals
                }
        }
        auto it = byName.find(word);
        if (it == byName.end()) {
            Stencil* s = nullptr;
            if (word == "reduced") s = new Reduced(m);
            else if (word == "leastSquares" || word == "leastSquaresOpt") s = new LeastSquares(m);
            else if (word == "GaussVolPoint") s = new GaussVolPoint(m);
            else return -5;
            it = byName.insert(std::make_pair(word, s)).first;
        }
        *out = it->second;
        return 0;
    }
};

struct MeshHandle {
    Mesh m;
    StencilCache cache;
};

// ---------------------------------------------------------------------------
// small tensor algebra in OpenFOAM's component order (L0 operator definitions)
// ---------------------------------------------------------------------------
inline void outer(const double* a, const double* b, double* T) { for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) T[3 * i + j] = a[i] * b[j]; }
inline void TdotV(const double* T, const double* v, double* r) { for (int i = 0; i < 3; ++i) r[i] = T[3 * i] * v[0] + T[3 * i + 1] * v[1] + T[3 * i + 2] * v[2]; }
inline void VdotT(const double* v, const double* T, double* r) { for (int j = 0; j < 3; ++j) r[j] = v[0] * T[j] + v[1] * T[3 + j] + v[2] * T[6 + j]; }
inline void TdotT(const double* A, const double* B, double* R) {
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}

// L0: fvc::grad(U) with the Gauss linear scheme: (1/V) sum_f Sf (x) Uf, tensor component 3*i+j = d_i U_j; patch values =
// owner cell's gradient (extrapolatedCalculated) with the normal part replaced by the patch snGrad on non-coupled patches
// (gaussGrad::correctBoundaryConditions)
VolField gaussGradVector(const Mesh& m, const std::vector<char>& liveFace, const VolField& f) {
    SurfField ff = linearInterpolate(m, f);
    VolField g(m, 9);
    for (int fc = 0; fc < m.nF; ++fc) {
        if (!liveFace[fc]) continue;
        const double* S = &m.Sf[3 * (size_t)fc];
        double* go = &g.in[9 * (size_t)m.own[fc]];
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) go[3 * i + j] += S[i] * ff.v[3 * (size_t)fc + j];
        if (fc < m.nIF) {
            double* gn = &g.in[9 * (size_t)m.nei[fc]];
            for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) gn[3 * i + j] -= S[i] * ff.v[3 * (size_t)fc + j];
        }
    }
    for (int c = 0; c < m.nC; ++c) for (int k = 0; k < 9; ++k) g.in[9 * (size_t)c + k] /= m.V[c];
    dvec sn = allPatchSnGrad(m, f);
    for (size_t ip = 0; ip < m.patches.size(); ++ip) {
        if (!m.patchHasFields((int)ip)) continue;
        for (int gf = m.patches[ip].start; gf < m.patches[ip].start + m.patches[ip].size; ++gf) {
            const int b = gf - m.nIF;
            double* gb = &g.bf[9 * (size_t)b];
            for (int k = 0; k < 9; ++k) gb[k] = g.in[9 * (size_t)m.own[gf] + k];
            if (m.coupled((int)ip)) continue;
            double n[3], ng[3];
            for (int k = 0; k < 3; ++k) n[k] = m.Sf[3 * (size_t)gf + k] / m.magSf[gf];
            VdotT(n, gb, ng);
            for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) gb[3 * i + j] += n[i] * (sn[3 * (size_t)b + j] - ng[j]);
        }
    }
    return g;
}

// Symmetric positive definite system of a Gauss laplacian plus a diagonal: (diag_c) x_c - sum_f a_f (x_nb - ... ) in the form
// y_c = diag_c x_c - sum_{faces} a_f x_nb; Jacobi-preconditioned conjugate gradients, OpenFOAM's normalised residual (L0)
int solveDiagLaplacian(const Mesh& m, const dvec& a, const dvec& diag, const dvec& rhs, double* x, double tol, int maxIter) {
    const int nC = m.nC;
    auto apply = [&](const double* v, dvec& y) {
        for (int c = 0; c < nC; ++c) y[c] = diag[c] * v[c];
        for (int f = 0; f < m.nIF; ++f) { y[m.own[f]] -= a[f] * v[m.nei[f]]; y[m.nei[f]] -= a[f] * v[m.own[f]]; }
    };
    dvec r((size_t)nC), z((size_t)nC), d((size_t)nC), q((size_t)nC), A1((size_t)nC), ones((size_t)nC, 1.0);
    apply(ones.data(), A1);
    apply(x, q);
    double xbar = 0;
    for (int c = 0; c < nC; ++c) xbar += x[c];
    xbar /= nC;
    double normFactor = 1e-20, sumAbs = 0, rz = 0;
    for (int c = 0; c < nC; ++c) {
        normFactor += std::fabs(q[c] - xbar * A1[c]) + std::fabs(rhs[c] - xbar * A1[c]);
        r[c] = rhs[c] - q[c]; z[c] = r[c] / diag[c]; d[c] = z[c];
        sumAbs += std::fabs(r[c]); rz += r[c] * z[c];
    }
    double res = sumAbs / normFactor;
    int it = 0;
    while (it < maxIter && !(res < tol)) {
        apply(d.data(), q);
        double dq = 0;
        for (int c = 0; c < nC; ++c) dq += d[c] * q[c];
        if (!(dq > 0) || !(rz > 0)) break;
        const double alpha = rz / dq;
        double rzNew = 0; sumAbs = 0;
        for (int c = 0; c < nC; ++c) {
            x[c] += alpha * d[c]; r[c] -= alpha * q[c]; z[c] = r[c] / diag[c];
            rzNew += r[c] * z[c]; sumAbs += std::fabs(r[c]);
        }
        res = sumAbs / normFactor;
        const double beta = rzNew / rz;
        for (int c = 0; c < nC; ++c) d[c] = z[c] + beta * d[c];
        rz = rzNew;
        ++it;
    }
    return it;
}

// ---------------------------------------------------------------------------
// QGDFoam case
// ---------------------------------------------------------------------------
inline double symmAbsN(const Mesh& m, int ip, int gf, int k) { double n[3]; m.symmNormal(ip, gf, n); return std::fabs(n[k]); }
// OpenFOAM gives a field on a constraint patch the patch's own field type whatever the field file says (L0: fvPatchField::New):
// empty / cut planes carry nothing, symmetryPlane / symmetry reflect vectors (basicSymmetry) and leave scalars zero-gradient
inline void constraintKinds(int ptype, int32_t& bcU, int32_t& bcT, int32_t& bcP) {
    if (ptype == PATCH_EMPTY || ptype == PATCH_HALO) { bcU = bcT = bcP = BC_NONE; }
    else if (ptype == PATCH_SYMMETRYPLANE || ptype == PATCH_SYMMETRY) { bcU = BC_SLIP; bcT = bcP = BC_ZEROGRADIENT; }
}
struct PatchBC { int32_t bcU = BC_ZEROGRADIENT, bcT = BC_ZEROGRADIENT, bcP = BC_ZEROGRADIENT; double vU[3] = {0, 0, 0}, vT = 0, vP = 0; };

struct Case {
    MeshHandle* mh;
    const Mesh& m;
    orc_case_options opt;
    std::vector<PatchBC> bc;
    Stencil* stencil = nullptr;
    std::string stencilWord;
    // per-term entries of fvSchemes.fvsc [fvsc.C L51-58]: grad(U), grad(e), grad(rho), grad(p); empty word = the default
    enum { TERM_U = 0, TERM_E = 1, TERM_RHO = 2, TERM_P = 3 };
    std::string termWord[4];
    Stencil* termStencil[4] = {nullptr, nullptr, nullptr, nullptr};
    const std::string& wordOf(int term) const { return termWord[term].empty() ? stencilWord : termWord[term]; }
    // thermo + state (cells + boundary)
    VolField p, T, e, U, rho, rhoU, rhoE, psi, mu, alpha, gamma, c, H;
    // QGDCoeffs
    VolField hQGD, aQGD, tauQGD, muQGD, alphauQGD, PrQGD, ScQGD;
    dvec aQGDin, aQGDbf, ScQGDin, ScQGDbf;   // non-uniform alphaQGD / ScQGD handed over by the caller (empty: uniform)
    SurfField hQGDf, tauQGDf;
    // face fields
    SurfField rhof, Uf, rhoUf, UrhoUf, pf, cf, gammaf, Hf, alphauf, muf;
    SurfField gradUf, divUf, gradef, gradRhof, gradPf, rhoW, phiw, jm, phiJm, phi, phiJmU, phiP, Pif, phiPi, phiJmH, qf, phiQ, phiPiU;
    SurfField tauMC, phiTauMC, phiSigmaDotU;   // implicitDiffusion branch [createFaceFluxes.H, updateFluxes.H:107-111, QGDUEqn.H:72-74]
    int lastIterU[3] = {0, 0, 0}, lastIterE = 0;
    bool phiwRegistered = false;  // "phiwStar" exists in the registry after createFaceFluxes.H
    double time = 0, deltaT = 0, CoNum = 0;
    long stepCount = 0;

    Case(MeshHandle* h, const orc_case_options& o) : mh(h), m(h->m), opt(o), bc(h->m.patches.size()) {
        deltaT = o.deltaT;
        liveFace.assign(h->m.nF, 1);
        for (size_t ip = 0; ip < h->m.patches.size(); ++ip)
            if (!h->m.patchHasFields((int)ip))
                for (int f = h->m.patches[ip].start; f < h->m.patches[ip].start + h->m.patches[ip].size; ++f) liveFace[f] = 0;
        for (size_t ip = 0; ip < bc.size(); ++ip) constraintKinds(m.patches[ip].type, bc[ip].bcU, bc[ip].bcT, bc[ip].bcP);
    }

    // ---- thermo closures (L0: perfectGas + eConst(Tref=0,Esref=0) + constTransport)
    double Cp() const { return opt.Cv + opt.R; }
    double HE(double /*p*/, double T) const { return opt.Cv * T; }
    double THE(double he, double /*p*/, double T0) const {  // L0: thermo::T Newton, tol 1e-4
        double Test = T0, Tnew = T0;
        const double Ttol = T0 * 1e-4;
        int iter = 0;
        do {
            Test = Tnew;
            Tnew = Test - (opt.Cv * Test - he) / opt.Cv;
            if (iter++ > 100) break;
        } while (std::fabs(Tnew - Test) > Ttol);
        return Tnew;
    }
    double psiOf(double /*p*/, double T) const { return 1.0 / (opt.R * T); }
    double alphahOf() const { const double rPr = 1.0 / opt.Pr; return (Cp() * opt.mu * rPr) / Cp(); }

    template <class Fn> void forPatchFaces(int ip, Fn fn) const {
        if (!m.patchHasFields(ip)) return;
        for (int gf = m.patches[ip].start; gf < m.patches[ip].start + m.patches[ip].size; ++gf) fn(gf, gf - m.nIF, m.own[gf]);
    }

    template <class Fn> void forAllPatchFaces(Fn fn) const { for (size_t ip = 0; ip < m.patches.size(); ++ip) forPatchFaces((int)ip, fn); }

    // ---- boundary-condition evaluation -----------------------------------
    void correctBC_U() {
        for (size_t ip = 0; ip < bc.size(); ++ip) forPatchFaces((int)ip, [&](int gf, int b, int o) {
            const PatchBC& B = bc[ip];
            if (B.bcU == BC_FIXEDVALUE) { for (int k = 0; k < 3; ++k) U.bf[3 * (size_t)b + k] = B.vU[k]; }
            else if (B.bcU == BC_SLIP) {  // L0 basicSymmetry::evaluate: (pif + transform(I - 2 nn, pif))/2
                double n[3], T[9];
                m.symmNormal((int)ip, gf, n);
                for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) T[3 * i + j] = (i == j ? 1.0 : 0.0) - 2.0 * (n[i] * n[j]);
                double tv[3];
                TdotV(T, &U.in[3 * (size_t)o], tv);
                for (int k = 0; k < 3; ++k) U.bf[3 * (size_t)b + k] = (U.in[3 * (size_t)o + k] + tv[k]) / 2.0;
            } else { for (int k = 0; k < 3; ++k) U.bf[3 * (size_t)b + k] = U.in[3 * (size_t)o + k]; }
        });
    }
    // L0 fixedEnergy / gradientEnergy for he [heThermo BCs]
    void correctBC_e() {
        for (size_t ip = 0; ip < bc.size(); ++ip) forPatchFaces((int)ip, [&](int gf, int b, int o) {
            if (bc[ip].bcT == BC_FIXEDVALUE) e.bf[b] = HE(p.bf[b], T.bf[b]);
            else {
                // gradient = Cv*Tw.snGrad() + deltaCoeffs*(he(pw,Tw) - he(pw,Tw)[faceCells]) = 0 for a
                // zeroGradient T; value = pif + gradient/deltaCoeffs
                const double g = opt.Cv * 0.0 + m.delta[gf] * (HE(p.bf[b], T.bf[b]) - HE(p.bf[b], T.bf[b]));
                e.bf[b] = e.in[o] + g / m.delta[gf];
            }
        });
    }
    // qgdFlux [qgdFluxFvPatchScalarField.C:159-197]
    void correctBC_p() {
        for (size_t ip = 0; ip < bc.size(); ++ip) forPatchFaces((int)ip, [&](int gf, int b, int o) {
            const PatchBC& B = bc[ip];
            if (B.bcP == BC_FIXEDVALUE) p.bf[b] = B.vP;
            else if (B.bcP == BC_QGDFLUX) {
                if (phiwRegistered) {
                    const double fluxSnGrad = phiw.v[gf] / tauQGDf.v[gf] / m.magSf[gf];
                    p.grad[b] = -fluxSnGrad;
                }
                p.bf[b] = p.in[o] + p.grad[b] / m.delta[gf];  // fixedGradient::evaluate
            } else p.bf[b] = p.in[o];
        });
    }

    // ---- hePsiQGDThermo::calculate [hePsiQGDThermo.C:38-126] ------------
    void thermoCalculate() {
        for (int ci = 0; ci < m.nC; ++ci) {
            T.in[ci] = THE(e.in[ci], p.in[ci], T.in[ci]);
            psi.in[ci] = psiOf(p.in[ci], T.in[ci]);
            mu.in[ci] = opt.mu;
            alpha.in[ci] = alphahOf();
        }
        for (size_t ip = 0; ip < bc.size(); ++ip) forPatchFaces((int)ip, [&](int, int b, int) {
            if (bc[ip].bcT == BC_FIXEDVALUE) e.bf[b] = HE(p.bf[b], T.bf[b]);
            else T.bf[b] = THE(e.bf[b], p.bf[b], T.bf[b]);
            psi.bf[b] = psiOf(p.bf[b], T.bf[b]);
            mu.bf[b] = opt.mu;
            alpha.bf[b] = alphahOf();
        });
        const double g = Cp() / opt.Cv;  // gamma_ == Cp()/Cv()
        for (double& x : gamma.in) x = g;
        for (double& x : gamma.bf) x = g;
        for (int ci = 0; ci < m.nC; ++ci) c.in[ci] = std::sqrt(gamma.in[ci] / psi.in[ci]);
        for (size_t ip = 0; ip < bc.size(); ++ip) forPatchFaces((int)ip, [&](int, int b, int) { c.bf[b] = std::sqrt(gamma.bf[b] / psi.bf[b]); });
        correctQGD();
    }
    // constScPrModel1::correct [constScPrModel1.C:97-131] + QGDThermo::correctQGD [QGDThermo.C:84-111]
    void computeTauQGDf() {
        VolField aOc(m, 1);
        for (int ci = 0; ci < m.nC; ++ci) aOc.in[ci] = aQGD.in[ci] / c.in[ci];
        for (size_t ip = 0; ip < bc.size(); ++ip) forPatchFaces((int)ip, [&](int, int b, int) { aOc.bf[b] = aQGD.bf[b] / c.bf[b]; });
        SurfField lin = linearInterpolate(m, aOc);
        for (int f = 0; f < m.nF; ++f) tauQGDf.v[f] = lin.v[f] * hQGDf.v[f];
    }
    void correctQGD() {
        computeTauQGDf();
        for (int ci = 0; ci < m.nC; ++ci) {
            tauQGD.in[ci] = aQGD.in[ci] * hQGD.in[ci] / c.in[ci];
            muQGD.in[ci] = p.in[ci] * ScQGD.in[ci] * tauQGD.in[ci];
            alphauQGD.in[ci] = muQGD.in[ci] / PrQGD.in[ci];
        }
        for (size_t ip = 0; ip < bc.size(); ++ip) forPatchFaces((int)ip, [&](int, int b, int) {
            tauQGD.bf[b] = aQGD.bf[b] * hQGD.bf[b] / c.bf[b];
            muQGD.bf[b] = p.bf[b] * ScQGD.bf[b] * tauQGD.bf[b];
            alphauQGD.bf[b] = muQGD.bf[b] / PrQGD.bf[b];
        });
        for (int ci = 0; ci < m.nC; ++ci) { mu.in[ci] += muQGD.in[ci]; alpha.in[ci] += alphauQGD.in[ci]; }
        for (size_t ip = 0; ip < bc.size(); ++ip) forPatchFaces((int)ip, [&](int, int b, int) { mu.bf[b] += muQGD.bf[b]; alpha.bf[b] += alphauQGD.bf[b]; });
    }
    // QGDCoeffs ctor + updateQGDLength [QGDCoeffs.C:195-199, 298-376]
    void initQGDCoeffs() {
        hQGDf = SurfField(m, 1); tauQGDf = SurfField(m, 1);
        hQGD = VolField(m, 1); aQGD = VolField(m, 1); tauQGD = VolField(m, 1); muQGD = VolField(m, 1);
        alphauQGD = VolField(m, 1); PrQGD = VolField(m, 1); ScQGD = VolField(m, 1);
        for (int f = 0; f < m.nF; ++f) hQGDf.v[f] = (m.delta[f] != 0.0) ? 1.0 / std::fabs(m.delta[f]) : 0.0;
        for (int f = 0; f < m.nIF; ++f) {
            double a[3], b[3];
            for (int k = 0; k < 3; ++k) { a[k] = m.C[3 * (size_t)m.own[f] + k] - m.Cf[3 * (size_t)f + k]; b[k] = m.C[3 * (size_t)m.nei[f] + k] - m.Cf[3 * (size_t)f + k]; }
            hQGDf.v[f] = 2.0 * std::min(mag3(a), mag3(b));
        }
        for (size_t ip = 0; ip < m.patches.size(); ++ip)
            if (!m.coupled((int)ip)) forPatchFaces((int)ip, [&](int gf, int, int) { hQGDf.v[gf] *= 2.0; });
        for (int f = 0; f < m.nF; ++f) if (!liveFace[f]) hQGDf.v[f] = 0.0;  // no field entries on empty patches
        for (int ci = 0; ci < m.nC; ++ci) {
            double hint = 0, surf = 0;
            for (int fid : m.cells[ci]) {
                if (fid < m.nIF) { hint += hQGDf.v[fid] * m.magSf[fid]; surf += m.magSf[fid]; }
                else {
                    int pid = -1;
                    for (size_t ip = 0; ip < m.patches.size(); ++ip)
                        if (fid >= m.patches[ip].start && fid < m.patches[ip].start + m.patches[ip].size) pid = (int)ip;
                    if (pid >= 0 && m.patches[pid].type != PATCH_EMPTY && m.patches[pid].type != PATCH_WEDGE) {
                        hint += hQGDf.v[fid] * m.magSf[fid];
                        surf += m.magSf[fid];
                    }
                }
            }
            hQGD.in[ci] = hint / surf;
        }
        for (size_t ip = 0; ip < m.patches.size(); ++ip) forPatchFaces((int)ip, [&](int gf, int b, int) { hQGD.bf[b] = hQGDf.v[gf] * 1.0; });
        for (double& x : aQGD.in) x = opt.alphaQGD;   // readOrCreateAlphaQGD: uniform, zeroGradient
        for (double& x : aQGD.bf) x = opt.alphaQGD;
        if (!aQGDin.empty()) { aQGD.in = aQGDin; aQGD.bf = aQGDbf; }   // the alphaQGD file [QGDCoeffs.C:119-143]
        for (double& x : PrQGD.in) x = opt.PrQGD;
        for (double& x : PrQGD.bf) x = opt.PrQGD;
        for (double& x : ScQGD.in) x = opt.ScQGD;
        for (double& x : ScQGD.bf) x = opt.ScQGD;
        if (!ScQGDin.empty()) { ScQGD.in = ScQGDin; ScQGD.bf = ScQGDbf; }   // the ScQGD file [constScPrModel1.C:66-79]
    }

    // createFields.H [QGDFoam/createFields.H:3-109] + createFaceFluxes.H
    int setFields(const double* U0, const double* T0, const double* p0) {
        int rc = mh->cache.lookup(m, stencilWord, &stencil);
        if (rc) return rc;
        for (int t = 0; t < 4; ++t) {
            rc = mh->cache.lookup(m, wordOf(t), &termStencil[t]);
            if (rc) return rc;
        }
        p = VolField(m, 1); T = VolField(m, 1); e = VolField(m, 1); U = VolField(m, 3); rho = VolField(m, 1);
        rhoU = VolField(m, 3); rhoE = VolField(m, 1); psi = VolField(m, 1); mu = VolField(m, 1); alpha = VolField(m, 1);
        gamma = VolField(m, 1); c = VolField(m, 1); H = VolField(m, 1);
        p.grad.assign(m.nBF(), 0.0);
        p.snKind.assign(bc.size(), SN_GENERIC); T.snKind = p.snKind; e.snKind = p.snKind; U.snKind = p.snKind;
        for (size_t ip = 0; ip < bc.size(); ++ip) {
            p.snKind[ip] = bc[ip].bcP == BC_ZEROGRADIENT ? SN_ZERO : (bc[ip].bcP == BC_QGDFLUX ? SN_GRADIENT : SN_GENERIC);
            T.snKind[ip] = bc[ip].bcT == BC_ZEROGRADIENT ? SN_ZERO : SN_GENERIC;
            // he: fixedEnergy (fixedValue) or gradientEnergy (fixedGradient, gradient()==0 here)
            e.snKind[ip] = bc[ip].bcT == BC_FIXEDVALUE ? SN_GENERIC : SN_ZERO;
            U.snKind[ip] = bc[ip].bcU == BC_ZEROGRADIENT ? SN_ZERO : (bc[ip].bcU == BC_SLIP ? SN_SYMM : SN_GENERIC);
            if (m.patches[ip].type == PATCH_HALO) p.snKind[ip] = T.snKind[ip] = e.snKind[ip] = U.snKind[ip] = SN_ZERO;
        }
        std::copy(U0, U0 + 3 * (size_t)m.nC, U.in.begin());
        std::copy(T0, T0 + m.nC, T.in.begin());
        std::copy(p0, p0 + m.nC, p.in.begin());
        // field construction evaluates the patches
        for (size_t ip = 0; ip < bc.size(); ++ip) forPatchFaces((int)ip, [&](int, int b, int o) {
            T.bf[b] = bc[ip].bcT == BC_FIXEDVALUE ? bc[ip].vT : T.in[o];
            p.bf[b] = bc[ip].bcP == BC_FIXEDVALUE ? bc[ip].vP : p.in[o];  // qgdFlux: patchInternalField, gradient 0
        });
        correctBC_U();
        initQGDCoeffs();
        // heThermo::init: he = he(p,T) on cells and patches
        for (int ci = 0; ci < m.nC; ++ci) e.in[ci] = HE(p.in[ci], T.in[ci]);
        for (int b = 0; b < m.nBF(); ++b) e.bf[b] = HE(p.bf[b], T.bf[b]);
        thermoCalculate();  // hePsiQGDThermo ctor
        thermoCalculate();  // thermo.correct() [createFields.H:8]
        for (int ci = 0; ci < m.nC; ++ci) rho.in[ci] = p.in[ci] * psi.in[ci];  // psiThermo::rho()
        for (int b = 0; b < m.nBF(); ++b) rho.bf[b] = p.bf[b] * psi.bf[b];
        auto cons = [&](const double* r, const double* u, const double* ee, double* ru, double* rE) {
            for (int k = 0; k < 3; ++k) ru[k] = r[0] * u[k];
            rE[0] = r[0] * ee[0] + r[0] * 0.5 * (u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
        };
        for (int ci = 0; ci < m.nC; ++ci) cons(&rho.in[ci], &U.in[3 * (size_t)ci], &e.in[ci], &rhoU.in[3 * (size_t)ci], &rhoE.in[ci]);
        for (int b = 0; b < m.nBF(); ++b) cons(&rho.bf[b], &U.bf[3 * (size_t)b], &e.bf[b], &rhoU.bf[3 * (size_t)b], &rhoE.bf[b]);
        for (int ci = 0; ci < m.nC; ++ci) H.in[ci] = (rhoE.in[ci] + p.in[ci]) / rho.in[ci];
        forAllPatchFaces([&](int, int b, int) { H.bf[b] = (rhoE.bf[b] + p.bf[b]) / rho.bf[b]; });
        // createFaceFluxes.H registers "phiwStar" = Sf & (tauQGDf*gradPf); with
        // GaussVolPoint that first grad(p) runs the qgdFlux BC before the name exists
        phiwRegistered = false;
        gradPf = fvscGrad(p, TERM_P);
        phiw = SurfField(m, 1);
        for (int f = 0; f < m.nF; ++f) {
            double rw[3];
            for (int k = 0; k < 3; ++k) rw[k] = tauQGDf.v[f] * gradPf.v[3 * (size_t)f + k];
            phiw.v[f] = dot3(&m.Sf[3 * (size_t)f], rw);
        }
        phiwRegistered = true;
        time = 0; stepCount = 0;
        return 0;
    }

    // fvsc::grad: GaussVolPoint re-evaluates the BCs of its input first
    // [GaussVolPointStencil.C:73,91]; of the case's fields only p's qgdFlux BC changes by that (B6)
    SurfField fvscGrad(VolField& f, int term) {
        if (term == TERM_P && wordOf(TERM_P) == "GaussVolPoint") correctBC_p();
        return f.nc == 1 ? termStencil[term]->gradS(f) : termStencil[term]->gradV(f);
    }

    // updateFields.H [QGDFoam/updateFields.H:45-80]
    void updateFields() {
        rhof = linearInterpolate(m, rho);
        Uf = linearInterpolate(m, U);
        rhoUf = linearInterpolate(m, rhoU);
        VolField UrhoU(m, 9);
        for (int ci = 0; ci < m.nC; ++ci) outer(&U.in[3 * (size_t)ci], &rhoU.in[3 * (size_t)ci], &UrhoU.in[9 * (size_t)ci]);
        for (int b = 0; b < m.nBF(); ++b) outer(&U.bf[3 * (size_t)b], &rhoU.bf[3 * (size_t)b], &UrhoU.bf[9 * (size_t)b]);
        UrhoUf = linearInterpolate(m, UrhoU);
        pf = linearInterpolate(m, p);
        cf = linearInterpolate(m, c);
        gammaf = linearInterpolate(m, gamma);
        for (int ci = 0; ci < m.nC; ++ci) H.in[ci] = (rhoE.in[ci] + p.in[ci]) / rho.in[ci];
        forAllPatchFaces([&](int, int b, int) { H.bf[b] = (rhoE.bf[b] + p.bf[b]) / rho.bf[b]; });
        Hf = linearInterpolate(m, H);
        // L0: laminar alphaEff() = thermo.alphaEff(alphat=0) = gamma*(alpha + 0) for an
        // internal-energy thermo; muEff() = mut(=0) + mu
        VolField aEff(m, 1), mEff(m, 1);
        for (int ci = 0; ci < m.nC; ++ci) { aEff.in[ci] = gamma.in[ci] * (alpha.in[ci] + 0.0); mEff.in[ci] = 0.0 + mu.in[ci]; }
        for (int b = 0; b < m.nBF(); ++b) { aEff.bf[b] = gamma.bf[b] * (alpha.bf[b] + 0.0); mEff.bf[b] = 0.0 + mu.bf[b]; }
        alphauf = linearInterpolate(m, aEff);
        muf = linearInterpolate(m, mEff);
    }

    // updateFluxes.H [QGDFoam/updateFluxes.H:41-139], explicit branch
    void updateFluxes() { updateFluxesA(); updateFluxesB(); }
    void updateFluxesA() {
        const int nF = m.nF;
        gradUf = fvscGrad(U, TERM_U);
        divUf = SurfField(m, 1);
        for (int f = 0; f < nF; ++f) divUf.v[f] = gradUf.v[9 * (size_t)f] + gradUf.v[9 * (size_t)f + 4] + gradUf.v[9 * (size_t)f + 8];
        gradef = fvscGrad(e, TERM_E);
        gradRhof = fvscGrad(rho, TERM_RHO);
        rhoW = SurfField(m, 3); phiw = SurfField(m, 1);
        for (int f = 0; f < nF; ++f) {
            if (!liveFace[f]) continue;
            const double* uf = &Uf.v[3 * (size_t)f];
            const double* ruf = &rhoUf.v[3 * (size_t)f];
            double A[9], t1[3], t3[3];
            outer(uf, &gradRhof.v[3 * (size_t)f], A);   // Uf * gradRhof
            TdotV(A, uf, t1);                           // & Uf
            VdotT(ruf, &gradUf.v[9 * (size_t)f], t3);   // rhoUf & gradUf
            for (int k = 0; k < 3; ++k) rhoW.v[3 * (size_t)f + k] = tauQGDf.v[f] * ((t1[k] + (ruf[k] * divUf.v[f])) + t3[k]);
            phiw.v[f] = dot3(&m.Sf[3 * (size_t)f], &rhoW.v[3 * (size_t)f]);
        }
        // fvsc::grad(p): under GaussVolPoint p's boundary conditions first (B6) -- the qgdFlux patches read the phiwStar just formed
        if (wordOf(TERM_P) == "GaussVolPoint") correctBC_p();
    }
    // ... and the rest of updateFluxes.H from the gradient of p on.  On a shard the patch faces of GHOST cells hold a mid-step patch
    // pressure formed from an incomplete stencil (their far vertices lack cells): their owner's values arrive between the two halves
    // (midHaloMove), because the vertex values of p on the wall mix them into the stencil of owned faces.
    void updateFluxesB() {
        const int nF = m.nF;
        gradPf = termStencil[TERM_P]->gradS(p);
        jm = SurfField(m, 3); phiJm = SurfField(m, 1); phi = SurfField(m, 1);
        for (int f = 0; f < nF; ++f) {
            if (!liveFace[f]) continue;
            for (int k = 0; k < 3; ++k) {
                rhoW.v[3 * (size_t)f + k] += tauQGDf.v[f] * gradPf.v[3 * (size_t)f + k];
                jm.v[3 * (size_t)f + k] = rhoUf.v[3 * (size_t)f + k] - rhoW.v[3 * (size_t)f + k];
            }
            phiJm.v[f] = dot3(&m.Sf[3 * (size_t)f], &jm.v[3 * (size_t)f]);
            phi.v[f] = dot3(&m.Sf[3 * (size_t)f], &rhoUf.v[3 * (size_t)f]);
        }
        phiJmU = SurfField(m, 3); phiP = SurfField(m, 3); Pif = SurfField(m, 9); phiPi = SurfField(m, 3);
        phiJmH = SurfField(m, 1); qf = SurfField(m, 3); phiQ = SurfField(m, 1); phiPiU = SurfField(m, 1);
        // qgdFlux(phiJm,U,Uf) / qgdFlux(phiJm,H,Hf): flux*psif, or fvc::flux when divSchemes has the flux's entry [QGDInterpolate.H L86-104]
        SurfField UfUp, HfUp;
        if (opt.fluxSchemeU) UfUp = upwindInterpolate(m, phiJm, U);
        if (opt.fluxSchemeH) HfUp = upwindInterpolate(m, phiJm, H);
        for (int f = 0; f < nF; ++f) {
            if (!liveFace[f]) continue;
            const double* uf = &Uf.v[3 * (size_t)f];
            const double* gU = &gradUf.v[9 * (size_t)f];
            const double* gP = &gradPf.v[3 * (size_t)f];
            const double tau = tauQGDf.v[f];
            for (int k = 0; k < 3; ++k) {
                phiJmU.v[3 * (size_t)f + k] = phiJm.v[f] * (opt.fluxSchemeU ? UfUp.v[3 * (size_t)f + k] : uf[k]);   // qgdFlux -> flux*psif [QGDInterpolate.H:104]
                phiP.v[3 * (size_t)f + k] = m.Sf[3 * (size_t)f + k] * pf.v[f];
            }
            double A[9], B[9];
            TdotT(&UrhoUf.v[9 * (size_t)f], gU, A);   // UrhoUf & gradUf
            outer(uf, gP, B);                         // Uf*gradPf
            const double sph = tau * (1.0 * (dot3(uf, gP) + (gammaf.v[f] * pf.v[f] * divUf.v[f])));
            double* Pi = &Pif.v[9 * (size_t)f];
            for (int k = 0; k < 9; ++k) Pi[k] = tau * (A[k] + B[k]);
            Pi[0] += sph; Pi[4] += sph; Pi[8] += sph;
            // explicit branch [updateFluxes.H:95-106]
            const double s23 = (2.0 / 3.0) * 1.0 * divUf.v[f];
            if (!opt.implicitDiffusion) for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
                double t = gU[3 * i + j] + gU[3 * j + i];
                if (i == j) t = t - s23;
                Pi[3 * i + j] += muf.v[f] * t;
            }
            VdotT(&m.Sf[3 * (size_t)f], Pi, &phiPi.v[3 * (size_t)f]);
            phiJmH.v[f] = phiJm.v[f] * (opt.fluxSchemeH ? HfUp.v[f] : Hf.v[f]);
            double g2[3], q[3];
            const double pr2 = pf.v[f] / rhof.v[f] / rhof.v[f];
            for (int k = 0; k < 3; ++k) g2[k] = gradef.v[3 * (size_t)f + k] - pr2 * gradRhof.v[3 * (size_t)f + k];
            TdotV(&UrhoUf.v[9 * (size_t)f], g2, q);
            for (int k = 0; k < 3; ++k) {
                qf.v[3 * (size_t)f + k] = (-tau) * q[k];
                if (!opt.implicitDiffusion) qf.v[3 * (size_t)f + k] -= alphauf.v[f] * gradef.v[3 * (size_t)f + k];   // [:131-135]
            }
            phiQ.v[f] = dot3(&m.Sf[3 * (size_t)f], &qf.v[3 * (size_t)f]);
            double piU[3];
            TdotV(Pi, uf, piU);
            phiPiU.v[f] = dot3(&m.Sf[3 * (size_t)f], piU);
        }
        if (opt.implicitDiffusion) {
            // tauMC = qgdInterpolate(muEff * dev2(T(fvc::grad(U)))), phiTauMC = Sf & tauMC [updateFluxes.H:107-111]
            VolField gU = gaussGradVector(m, liveFace, U);
            VolField prod(m, 9);
            auto dev2T = [](const double* g, double muEff, double* out) {   // dev2(A) = A - (2/3) tr(A) I, A = T(g)
                const double tr = g[0] + g[4] + g[8];
                for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
                    double a = g[3 * j + i];
                    if (i == j) a = a - (2.0 / 3.0) * tr;
                    out[3 * i + j] = muEff * a;
                }
            };
            for (int ci = 0; ci < m.nC; ++ci) dev2T(&gU.in[9 * (size_t)ci], 0.0 + mu.in[ci], &prod.in[9 * (size_t)ci]);
            for (int b = 0; b < m.nBF(); ++b) dev2T(&gU.bf[9 * (size_t)b], 0.0 + mu.bf[b], &prod.bf[9 * (size_t)b]);
            tauMC = linearInterpolate(m, prod);
            phiTauMC = SurfField(m, 3);
            for (int f = 0; f < nF; ++f) if (liveFace[f]) VdotT(&m.Sf[3 * (size_t)f], &tauMC.v[9 * (size_t)f], &phiTauMC.v[3 * (size_t)f]);
        }
    }

    // -----------------------------------------------------------------------------------------------------------------------------
    // The same flux assembly FUSED (bench.py cpu_baseline "fused"): one pass over the vertices for the six interpolated fields, one
    // pass over the faces that forms the four fvsc gradients, the 13 interpolations and the whole flux algebra in registers and stores
    // only the seven face fields the equations read -- what a CPU code written for speed would do with the reference's formulas,
    // where updateFields() / updateFluxes() above materialise ~50 face fields one at a time like the reference does.  Same operations
    // in the same order per face, so the results are those of the unfused path (tests/test_oracle_fused.py).  Scope: 3-D
    // GaussVolPoint, quadrilateral faces only, explicit branch, fixed deltaT, no qgdFlux patch (their mid-assembly boundary condition
    // would split the face pass); anything else returns false and the caller keeps the unfused path.
    // -----------------------------------------------------------------------------------------------------------------------------
    dvec fusedPv;
    bool fusedSupported() const {
        const GaussVolPoint* gv = dynamic_cast<const GaussVolPoint*>(stencil);
        if (!gv || m.nGeomD != 3 || opt.implicitDiffusion || opt.adjustTimeStep || !m.haloGhost.empty() || !m.pointOps.empty() || opt.fluxSchemeU || opt.fluxSchemeH) return false;
        for (int t = 0; t < 4; ++t) if (!termWord[t].empty() && termWord[t] != stencilWord) return false;   // mixed stencils: the field-at-a-time form only
        if (!gv->tf.empty() || !gv->of.empty()) return false;
        for (size_t ip = 0; ip < m.patches.size(); ++ip) {
            if (!m.patchHasFields((int)ip)) continue;
            if (!gv->btf[ip].empty() || !gv->bof[ip].empty() || bc[ip].bcP == BC_QGDFLUX) return false;
        }
        return true;
    }
    // flat copies of what the fused passes walk (the stencil keeps one small vector per face and per vertex, as the reference does)
    struct FusedTables {
        dvec coef;                 // 19 per face: a[direction][slot] (3 x 6: four vertices, "neighbour", owner), volume
        std::vector<int64_t> pcOff; ivec pcCell; dvec pcW;        // interior vertices: cells and weights (CSR)
        ivec bpPoint; std::vector<int64_t> bpOff; ivec bpFace; dvec bpW;   // patch vertices: patch faces (flatBoundaryField index, -1 = zero) and weights
    } ft;
    void buildFusedTables(const GaussVolPoint* gv) {
        ft.coef.assign((size_t)m.nF * 19, 0.0);
        auto put = [&](int f, const std::vector<dvec>* a, size_t ai, double vol) {
            double* c = &ft.coef[(size_t)f * 19];
            for (int d = 0; d < 3; ++d) for (int k = 0; k < 6; ++k) c[6 * d + k] = a[d][ai][k];
            c[18] = vol;
        };
        for (size_t i = 0; i < gv->qf.size(); ++i) put(gv->qf[i], gv->aq, i, gv->vq[i]);
        for (size_t ip = 0; ip < m.patches.size(); ++ip) {
            if (!m.patchHasFields((int)ip)) continue;
            const std::vector<dvec> a3[3] = {gv->baq[0][ip], gv->baq[1][ip], gv->baq[2][ip]};
            for (size_t k = 0; k < gv->bqf[ip].size(); ++k) put(m.patches[ip].start + gv->bqf[ip][k], a3, k, gv->bvq[ip][k]);
        }
        ft.pcOff.assign((size_t)m.nP + 1, 0);
        for (int pt = 0; pt < m.nP; ++pt) ft.pcOff[pt + 1] = ft.pcOff[pt] + (m.isPatchPoint[pt] ? 0 : (int64_t)m.pointCells[pt].size());
        ft.pcCell.resize((size_t)ft.pcOff[m.nP]); ft.pcW.resize((size_t)ft.pcOff[m.nP]);
        for (int pt = 0; pt < m.nP; ++pt) {
            if (m.isPatchPoint[pt]) continue;
            for (size_t i = 0; i < m.pointCells[pt].size(); ++i) { ft.pcCell[ft.pcOff[pt] + i] = m.pointCells[pt][i]; ft.pcW[ft.pcOff[pt] + i] = m.pointWeights[pt][i]; }
        }
        ft.bpPoint.clear(); ft.bpOff.assign(1, 0); ft.bpFace.clear(); ft.bpW.clear();
        for (size_t i = 0; i < m.bndMeshPoints.size(); ++i) {
            const int pt = m.bndMeshPoints[i];
            if (!m.isPatchPoint[pt]) continue;
            ft.bpPoint.push_back(pt);
            for (size_t j = 0; j < m.bndPointFaces[i].size(); ++j) {
                const int b = m.bndPointFaces[i][j];
                if (!m.isPatchFace[b]) continue;
                const int pid = patchOf(m.nIF + b);
                const bool flat = pid >= 0 && m.patches[pid].type != PATCH_EMPTY && !m.coupled(pid);   // flatBoundaryField: zero elsewhere
                ft.bpFace.push_back(flat ? b : -1);
                ft.bpW.push_back(m.bndPointWeights[i][j]);
            }
            ft.bpOff.push_back((int64_t)ft.bpFace.size());
        }
    }
    bool updateFluxesFused() {
        if (!fusedSupported()) return false;
        const GaussVolPoint* gv = dynamic_cast<const GaussVolPoint*>(stencil);
        if (ft.coef.size() != (size_t)m.nF * 19) buildFusedTables(gv);
        const int nF = m.nF, nP = m.nP;
        // H of updateFields.H L63
        for (int ci = 0; ci < m.nC; ++ci) H.in[ci] = (rhoE.in[ci] + p.in[ci]) / rho.in[ci];
        forAllPatchFaces([&](int, int b, int) { H.bf[b] = (rhoE.bf[b] + p.bf[b]) / rho.bf[b]; });
        // ---- vertex values of rho, U, p, e in one pass (volPointInterpolate of each, same sums) ----
        if (fusedPv.size() != (size_t)nP * 6) fusedPv.assign((size_t)nP * 6, 0.0);
        dvec& pv = fusedPv;
        for (int pt = 0; pt < nP; ++pt) {
            double* o = &pv[(size_t)pt * 6];
            for (int k = 0; k < 6; ++k) o[k] = 0.0;
            for (int64_t i = ft.pcOff[pt]; i < ft.pcOff[pt + 1]; ++i) {
                const double w = ft.pcW[i];
                const size_t ci = (size_t)ft.pcCell[i];
                o[0] += w * rho.in[ci];
                o[1] += w * U.in[3 * ci]; o[2] += w * U.in[3 * ci + 1]; o[3] += w * U.in[3 * ci + 2];
                o[4] += w * p.in[ci];
                o[5] += w * e.in[ci];
            }
        }
        for (size_t i = 0; i < ft.bpPoint.size(); ++i) {
            double* o = &pv[(size_t)ft.bpPoint[i] * 6];
            for (int64_t j = ft.bpOff[i]; j < ft.bpOff[i + 1]; ++j) {
                const double w = ft.bpW[j];
                const int b = ft.bpFace[j];
                if (b < 0) { for (int k = 0; k < 6; ++k) o[k] += w * 0.0; continue; }
                o[0] += w * rho.bf[b];
                o[1] += w * U.bf[3 * (size_t)b]; o[2] += w * U.bf[3 * (size_t)b + 1]; o[3] += w * U.bf[3 * (size_t)b + 2];
                o[4] += w * p.bf[b];
                o[5] += w * e.bf[b];
            }
        }
        // psin = patch value + snGrad * |vO - vN| / 2 of the four fields (boundary-sized)
        const dvec psinR = gv->psiN(rho), psinU = gv->psiN(U), psinP = gv->psiN(p), psinE = gv->psiN(e);
        for (SurfField* sf : {&phiJm, &phiJmH, &phiQ, &phiPiU, &phi}) if ((int)sf->v.size() != nF) *sf = SurfField(m, 1);
        for (SurfField* sf : {&phiJmU, &phiP, &phiPi}) if ((int)sf->v.size() != 3 * nF) *sf = SurfField(m, 3);
        const double gm = Cp() / opt.Cv;
        // one face: six gradients from the stencil's own coefficient tables, then everything else in registers
        auto face = [&](int f, bool boundary, int b) {
            const double* cf19 = &ft.coef[(size_t)f * 19];
            const double vol = cf19[18];
            const int o = m.own[f], n = boundary ? -1 : m.nei[f];
            const int* q = m.fp(f);
            // values at "neighbour" (psin on a patch), owner, the four vertices
            double vn[6], vo[6];
            vo[0] = rho.in[o]; vo[1] = U.in[3 * (size_t)o]; vo[2] = U.in[3 * (size_t)o + 1]; vo[3] = U.in[3 * (size_t)o + 2]; vo[4] = p.in[o]; vo[5] = e.in[o];
            if (boundary) {
                vn[0] = psinR[b]; vn[1] = psinU[3 * (size_t)b]; vn[2] = psinU[3 * (size_t)b + 1]; vn[3] = psinU[3 * (size_t)b + 2]; vn[4] = psinP[b]; vn[5] = psinE[b];
            } else {
                vn[0] = rho.in[n]; vn[1] = U.in[3 * (size_t)n]; vn[2] = U.in[3 * (size_t)n + 1]; vn[3] = U.in[3 * (size_t)n + 2]; vn[4] = p.in[n]; vn[5] = e.in[n];
            }
            double g[6][3];   // g[field][direction]
            for (int d = 0; d < 3; ++d) {
                const double* c = cf19 + 6 * d;
                for (int k = 0; k < 6; ++k) {
                    double s2 = vn[k] * c[4];
                    s2 += vo[k] * c[5];
                    for (int v = 0; v < 4; ++v) s2 += pv[(size_t)q[v] * 6 + k] * c[v];
                    g[k][d] = 0.0 + (s2 / vol);
                }
            }
            double gU[9], gRho[3], gP[3], gE[3];
            for (int d = 0; d < 3; ++d) { gRho[d] = g[0][d]; gP[d] = g[4][d]; gE[d] = g[5][d]; for (int c2 = 0; c2 < 3; ++c2) gU[3 * d + c2] = g[1 + c2][d]; }
            const double divU = gU[0] + gU[4] + gU[8];
            // the interpolations of updateFields.H
            auto lerp = [&](double ao, double an) { return m.w[f] * (ao - an) + an; };
            double rhof_, uf[3], ruf[3], UrU[9], pf_, cf_, gammaf_, Hf_, alphauf_, muf_;
            if (boundary) {
                rhof_ = rho.bf[b]; pf_ = p.bf[b]; cf_ = c.bf[b]; gammaf_ = gamma.bf[b]; Hf_ = H.bf[b];
                alphauf_ = gamma.bf[b] * (alpha.bf[b] + 0.0); muf_ = 0.0 + mu.bf[b];
                for (int k = 0; k < 3; ++k) { uf[k] = U.bf[3 * (size_t)b + k]; ruf[k] = rhoU.bf[3 * (size_t)b + k]; }
                outer(&U.bf[3 * (size_t)b], &rhoU.bf[3 * (size_t)b], UrU);
            } else {
                rhof_ = lerp(rho.in[o], rho.in[n]); pf_ = lerp(p.in[o], p.in[n]); cf_ = lerp(c.in[o], c.in[n]);
                gammaf_ = lerp(gamma.in[o], gamma.in[n]); Hf_ = lerp(H.in[o], H.in[n]);
                alphauf_ = lerp(gamma.in[o] * (alpha.in[o] + 0.0), gamma.in[n] * (alpha.in[n] + 0.0));
                muf_ = lerp(0.0 + mu.in[o], 0.0 + mu.in[n]);
                double Ao[9], An[9];
                outer(&U.in[3 * (size_t)o], &rhoU.in[3 * (size_t)o], Ao);
                outer(&U.in[3 * (size_t)n], &rhoU.in[3 * (size_t)n], An);
                for (int k = 0; k < 3; ++k) { uf[k] = lerp(U.in[3 * (size_t)o + k], U.in[3 * (size_t)n + k]); ruf[k] = lerp(rhoU.in[3 * (size_t)o + k], rhoU.in[3 * (size_t)n + k]); }
                for (int k = 0; k < 9; ++k) UrU[k] = lerp(Ao[k], An[k]);
            }
            (void)cf_; (void)gm;
            const double tau = tauQGDf.v[f];
            const double* S = &m.Sf[3 * (size_t)f];
            // updateFluxes.H L41-139, explicit branch, as updateFluxesA/B have it
            double A[9], t1[3], t3[3], rw[3], jm_[3];
            outer(uf, gRho, A);
            TdotV(A, uf, t1);
            VdotT(ruf, gU, t3);
            for (int k = 0; k < 3; ++k) rw[k] = tau * ((t1[k] + (ruf[k] * divU)) + t3[k]);
            for (int k = 0; k < 3; ++k) { rw[k] += tau * gP[k]; jm_[k] = ruf[k] - rw[k]; }
            const double pJm = dot3(S, jm_);
            phiJm.v[f] = pJm;
            phi.v[f] = dot3(S, ruf);
            for (int k = 0; k < 3; ++k) { phiJmU.v[3 * (size_t)f + k] = pJm * uf[k]; phiP.v[3 * (size_t)f + k] = S[k] * pf_; }
            double B[9], Pi[9];
            TdotT(UrU, gU, A);
            outer(uf, gP, B);
            const double sph = tau * (1.0 * (dot3(uf, gP) + (gammaf_ * pf_ * divU)));
            for (int k = 0; k < 9; ++k) Pi[k] = tau * (A[k] + B[k]);
            Pi[0] += sph; Pi[4] += sph; Pi[8] += sph;
            const double s23 = (2.0 / 3.0) * 1.0 * divU;
            for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
                double t = gU[3 * i + j] + gU[3 * j + i];
                if (i == j) t = t - s23;
                Pi[3 * i + j] += muf_ * t;
            }
            VdotT(S, Pi, &phiPi.v[3 * (size_t)f]);
            phiJmH.v[f] = pJm * Hf_;
            double g2[3], qv[3], qf_[3];
            const double pr2 = pf_ / rhof_ / rhof_;
            for (int k = 0; k < 3; ++k) g2[k] = gE[k] - pr2 * gRho[k];
            TdotV(UrU, g2, qv);
            for (int k = 0; k < 3; ++k) { qf_[k] = (-tau) * qv[k]; qf_[k] -= alphauf_ * gE[k]; }
            phiQ.v[f] = dot3(S, qf_);
            double piU[3];
            TdotV(Pi, uf, piU);
            phiPiU.v[f] = dot3(S, piU);
        };
        for (int f = 0; f < m.nIF; ++f) face(f, false, -1);
        for (size_t ip = 0; ip < m.patches.size(); ++ip) {
            if (!m.patchHasFields((int)ip)) continue;
            for (int gf = m.patches[ip].start; gf < m.patches[ip].start + m.patches[ip].size; ++gf) face(gf, true, gf - m.nIF);
        }
        return true;
    }
    // n steps with the fused assembly (the equations, thermo and boundary conditions stay stepPhase1); false: not supported here
    bool stepFused(int n) {
        if (!fusedSupported()) return false;
        for (int i = 0; i < n; ++i) { updateFluxesFused(); stepPhase1(); }
        return true;
    }

    // the matrix of fvm::ddt(rho, x) - fvm::laplacian(gamma_f, x) for one scalar (component): face coefficients a_f =
    // gamma_f |S_f| delta_f (uncorrected, L0), diagonal rDeltaT rho V + sum a_f + the patch internal coefficients
    // icoef(face) (gamma |S| x -gradientInternalCoeffs), source bsrc(face) (gamma |S| x gradientBoundaryCoeffs)
    template <class IC, class BS>
    int implicitDiffusionSolve(const SurfField& gammaf, double rDeltaT, const dvec& rhsCell, IC icoef, BS bsrc, double* x) {
        dvec a((size_t)m.nF, 0.0), diag((size_t)m.nC), rhs = rhsCell;
        for (int f = 0; f < m.nIF; ++f) a[f] = gammaf.v[f] * m.magSf[f] * m.nonOrthDelta[f];
        for (int ci = 0; ci < m.nC; ++ci) diag[ci] = rDeltaT * rho.in[ci] * m.V[ci];
        for (int f = 0; f < m.nIF; ++f) { diag[m.own[f]] += a[f]; diag[m.nei[f]] += a[f]; }
        for (size_t ip = 0; ip < bc.size(); ++ip) forPatchFaces((int)ip, [&](int gf, int b, int o) {
            if (m.coupled((int)ip)) return;
            const double gs = gammaf.v[gf] * m.magSf[gf];
            diag[o] += gs * icoef((int)ip, gf, b, o);
            rhs[o] += gs * bsrc((int)ip, gf, b, o);
        });
        return solveDiagLaplacian(m, a, diag, rhs, x, opt.implicitTol, opt.implicitMaxIter);
    }

    // L0 fvc::div(ssf) = surfaceIntegrate: owner += , neighbour -= , patches += , /V
    dvec fvcDiv(const SurfField& s) const {
        const int nc = s.nc;
        dvec d((size_t)m.nC * nc, 0.0);
        for (int f = 0; f < m.nIF; ++f)
            for (int k = 0; k < nc; ++k) { d[(size_t)m.own[f] * nc + k] += s.v[(size_t)f * nc + k]; d[(size_t)m.nei[f] * nc + k] -= s.v[(size_t)f * nc + k]; }
        for (size_t ip = 0; ip < m.patches.size(); ++ip) {
            if (!m.patchHasFields((int)ip)) continue;
            for (int f = m.patches[ip].start; f < m.patches[ip].start + m.patches[ip].size; ++f)
                for (int k = 0; k < nc; ++k) d[(size_t)m.own[f] * nc + k] += s.v[(size_t)f * nc + k];
        }
        for (int ci = 0; ci < m.nC; ++ci) for (int k = 0; k < nc; ++k) d[(size_t)ci * nc + k] /= m.V[ci];
        return d;
    }

    // [QGDCourantNo.H:36-53] [setDeltaT-QGDQHD.H:41-61]
    // local part of the reductions (a sharded run max/min-reduces redCo / redMinTau over the ranks in between)
    double redCo = -GREAT, redMinTau = GREAT;
    void courantLocal() {
        if (!opt.adjustTimeStep) return;
        double co = -GREAT, mintau = GREAT;
        {
            for (int f = 0; f < m.nF; ++f) {
                if (f >= m.nIF) {
                    int ip = patchOf(f);
                    if (ip < 0 || !m.patchHasFields(ip) || m.patches[ip].type == PATCH_HALO) continue;
                    if (!ghostFlag.empty() && ghostFlag[m.own[f]]) continue;
                }
                else if (!faceTouchesOwned(f)) continue;
                double nrm[3];
                for (int k = 0; k < 3; ++k) nrm[k] = m.Sf[3 * (size_t)f + k] / m.magSf[f];
                const double Unf = dot3(&Uf.v[3 * (size_t)f], nrm);
                const double cof = std::max(std::fabs(Unf + cf.v[f]), std::fabs(Unf - cf.v[f])) * deltaT / hQGDf.v[f];
                co = std::max(co, cof);
                mintau = std::min(mintau, tauQGDf.v[f]);
            }
        }
        redCo = co; redMinTau = mintau;
    }
    void setDeltaT() {
        if (!opt.adjustTimeStep) return;
        CoNum = redCo;
        const double mintau = redMinTau;
        const double maxDeltaTFact = opt.maxCo / (CoNum + SMALL);
        const double deltaTFact = std::min(std::min(maxDeltaTFact, 1.0 + 0.1 * maxDeltaTFact), 1.2);
        double maxDeltaT1 = opt.cTau * mintau;
        maxDeltaT1 = std::min(opt.maxDeltaT, maxDeltaT1);
        deltaT = std::min(deltaTFact * deltaT, maxDeltaT1);
    }
    int patchOf(int f) const {
        for (size_t ip = 0; ip < m.patches.size(); ++ip) if (f >= m.patches[ip].start && f < m.patches[ip].start + m.patches[ip].size) return (int)ip;
        return -1;
    }
    std::vector<char> ghostFlag;
    std::vector<char> liveFace;  // 0 on faces of empty patches (emptyFvPatch has size 0: no field entries there)
    bool faceTouchesOwned(int f) const {
        if (ghostFlag.empty()) return true;
        return !ghostFlag[m.own[f]] || !ghostFlag[m.nei[f]];
    }

    // loop body [QGDFoam.C:90-163]: phase 0 = flux assembly + local Courant/tau reductions, phase 1 = deltaT and the
    // rest of the body, phase 2 = refresh after a halo unpack
    void stepPhase0() {
        updateFields();
        updateFluxes();
        courantLocal();
    }
    // phase 0 in two halves for shards that need the mid-assembly message (midNeeded)
    void stepPhase5() { updateFields(); updateFluxesA(); }
    void stepPhase6() { updateFluxesB(); courantLocal(); }
    bool midNeeded() const {
        if (m.haloGhost.empty() || wordOf(TERM_P) != "GaussVolPoint") return false;
        for (const PatchBC& B : bc) if (B.bcP == BC_QGDFLUX) return true;
        return false;
    }
    // the mid-step patch pressure and its gradient on the patch faces of the boundary-layer cells (2 doubles per face)
    void midHaloCount(int side, int64_t* send, int64_t* recv) const {
        *send = *recv = 0;
        if (side < 0 || (size_t)side >= m.haloGhost.size()) return;
        *send = 2 * (int64_t)m.haloSendBF[side].size(); *recv = 2 * (int64_t)m.haloGhostBF[side].size();
    }
    void midHaloMove(int side, double* buf, bool pack) {
        if (side < 0 || (size_t)side >= m.haloGhost.size()) return;
        const ivec& facesL = pack ? m.haloSendBF[side] : m.haloGhostBF[side];
        size_t q = 0;
        for (int b : facesL) {
            if (pack) { buf[q++] = p.bf[b]; buf[q++] = p.grad[b]; }
            else { p.bf[b] = buf[q++]; p.grad[b] = buf[q++]; }
        }
    }
    void stepPhase1() {
        setDeltaT();
        time += deltaT; stepCount++;
        const double rDeltaT = 1.0 / deltaT;
        const dvec rhoOld = rho.in, rhoUOld = rhoU.in, UOld = U.in, rhoEOld = rhoE.in, eOld = e.in;
        // QGDRhoEqn.H [:40-47]: diag = rDeltaT*V, source = rDeltaT*rho.old*V - V*div(phiJm)
        {
            dvec d = fvcDiv(phiJm);
            for (int ci = 0; ci < m.nC; ++ci) {
                const double diag = rDeltaT * m.V[ci];
                double src = rDeltaT * rhoOld[ci] * m.V[ci];
                src -= m.V[ci] * d[ci];
                rho.in[ci] = src / diag;
            }
        }
        // QGDUEqn.H [:36-89]
        {
            dvec d1 = fvcDiv(phiJmU), d2 = fvcDiv(phiP), d3 = fvcDiv(phiPi);
            for (int ci = 0; ci < m.nC; ++ci)
                for (int k = 0; k < 3; ++k) {
                    const double diag = rDeltaT * m.V[ci];
                    double src = rDeltaT * rhoUOld[3 * (size_t)ci + k] * m.V[ci];
                    src -= m.V[ci] * d1[3 * (size_t)ci + k];
                    src -= m.V[ci] * d2[3 * (size_t)ci + k];
                    src += m.V[ci] * d3[3 * (size_t)ci + k];
                    rhoU.in[3 * (size_t)ci + k] = src / diag;
                }
            for (int ci = 0; ci < m.nC; ++ci) for (int k = 0; k < 3; ++k) U.in[3 * (size_t)ci + k] = rhoU.in[3 * (size_t)ci + k] / rho.in[ci];
            correctBC_U();
            if (opt.implicitDiffusion) {
                // UEqn: fvm::ddt(rho,U) - fvc::ddt(rho,U) - fvm::laplacian(muf,U) - fvc::div(phiTauMC) == rhoUSu [:56-68],
                // solved component by component (fvMatrix<vector>::solveSegregated, L0); patch coefficients:
                // fixedValue: -gradientInternalCoeffs = delta, gradientBoundaryCoeffs = delta*value;
                // basicSymmetry (slip): -gIC = delta*|n_cmpt|, gBC = snGrad - gIC*patchInternalField (transformFvPatchField);
                // zeroGradient: none
                dvec dT = fvcDiv(phiTauMC);
                dvec sn = allPatchSnGrad(m, U);
                const dvec Ucur = U.in;
                for (int k = 0; k < 3; ++k) {
                    if (m.geomD[k] < 0) continue;   // validComponents: the empty direction is not solved
                    dvec rhs((size_t)m.nC), x((size_t)m.nC);
                    for (int ci = 0; ci < m.nC; ++ci) {
                        // rDeltaT rho.old U.old V  +  V rDeltaT (rho U - rho.old U.old)  +  V div(phiTauMC)
                        double src = rDeltaT * rhoOld[ci] * UOld[3 * (size_t)ci + k] * m.V[ci];
                        src += m.V[ci] * (rDeltaT * (rho.in[ci] * Ucur[3 * (size_t)ci + k] - rhoOld[ci] * UOld[3 * (size_t)ci + k]));
                        src += m.V[ci] * dT[3 * (size_t)ci + k];
                        rhs[ci] = src;
                        x[ci] = Ucur[3 * (size_t)ci + k];
                    }
                    auto ic = [&](int ip, int gf, int, int) {
                        if (bc[ip].bcU == BC_FIXEDVALUE) return m.delta[gf];
                        if (bc[ip].bcU == BC_SLIP) return m.delta[gf] * symmAbsN(m, (int)ip, gf, k);
                        return 0.0;
                    };
                    auto bs = [&](int ip, int gf, int b, int o) {
                        if (bc[ip].bcU == BC_FIXEDVALUE) return m.delta[gf] * U.bf[3 * (size_t)b + k];
                        if (bc[ip].bcU == BC_SLIP)
                            return sn[3 * (size_t)b + k] + m.delta[gf] * symmAbsN(m, (int)ip, gf, k) * Ucur[3 * (size_t)o + k];
                        return 0.0;
                    };
                    lastIterU[k] = implicitDiffusionSolve(muf, rDeltaT, rhs, ic, bs, x.data());
                    for (int ci = 0; ci < m.nC; ++ci) U.in[3 * (size_t)ci + k] = x[ci];
                }
                correctBC_U();
                for (int ci = 0; ci < m.nC; ++ci) for (int k = 0; k < 3; ++k) rhoU.in[3 * (size_t)ci + k] = rho.in[ci] * U.in[3 * (size_t)ci + k];   // [:70]
                // sigmaDotU = (muf*linearInterpolate(fvc::grad(U)) + tauMC) & Uf;  phiSigmaDotU = Sf & sigmaDotU [:72-74]
                VolField gUn = gaussGradVector(m, liveFace, U);
                SurfField gUf = linearInterpolate(m, gUn);
                phiSigmaDotU = SurfField(m, 1);
                for (int f = 0; f < m.nF; ++f) {
                    if (!liveFace[f]) continue;
                    double A[9], sd[3];
                    for (int q = 0; q < 9; ++q) A[q] = muf.v[f] * gUf.v[9 * (size_t)f + q] + tauMC.v[9 * (size_t)f + q];
                    TdotV(A, &Uf.v[3 * (size_t)f], sd);
                    phiSigmaDotU.v[f] = dot3(&m.Sf[3 * (size_t)f], sd);
                }
            } else {
            // solve(fvm::ddt(rho,U) - fvc::ddt(rhoU) == rhoUSu) [:79-86]
            for (int ci = 0; ci < m.nC; ++ci)
                for (int k = 0; k < 3; ++k) {
                    const double diag = rDeltaT * rho.in[ci] * m.V[ci];
                    double src = rDeltaT * rhoOld[ci] * UOld[3 * (size_t)ci + k] * m.V[ci];
                    src += m.V[ci] * (rDeltaT * (rhoU.in[3 * (size_t)ci + k] - rhoUOld[3 * (size_t)ci + k]));
                    U.in[3 * (size_t)ci + k] = src / diag;
                }
            correctBC_U();  // fvMatrix::solve ends with psi.correctBoundaryConditions() (L0)
            }
            for (int b = 0; b < m.nBF(); ++b) for (int k = 0; k < 3; ++k) rhoU.bf[3 * (size_t)b + k] = rho.bf[b] * U.bf[3 * (size_t)b + k];
        }
        // QGDEEqn.H [:37-76] (phiSigmaDotU == 0 in the explicit branch)
        {
            dvec d1 = fvcDiv(phiJmH), d2 = fvcDiv(phiQ), d3 = fvcDiv(phiPiU);
            for (int ci = 0; ci < m.nC; ++ci) {
                const double diag = rDeltaT * m.V[ci];
                double src = rDeltaT * rhoEOld[ci] * m.V[ci];
                src -= m.V[ci] * d1[ci];
                src -= m.V[ci] * d2[ci];
                src += m.V[ci] * d3[ci];
                src += m.V[ci] * 0.0;  // - fvc::div(phiSigmaDotU), a zero field in the explicit branch
                rhoE.in[ci] = src / diag;
            }
            if (opt.implicitDiffusion) {   // the real phiSigmaDotU [:43]
                dvec d4 = fvcDiv(phiSigmaDotU);
                for (int ci = 0; ci < m.nC; ++ci) {
                    const double diag = rDeltaT * m.V[ci];
                    double src = rDeltaT * rhoEOld[ci] * m.V[ci];
                    src -= m.V[ci] * d1[ci];
                    src -= m.V[ci] * d2[ci];
                    src += m.V[ci] * d3[ci];
                    src += m.V[ci] * d4[ci];
                    rhoE.in[ci] = src / diag;
                }
            }
            for (int ci = 0; ci < m.nC; ++ci) {
                const double* u = &U.in[3 * (size_t)ci];
                e.in[ci] = rhoE.in[ci] / rho.in[ci] - 0.5 * (u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
            }
            correctBC_e();
            // solve(fvm::ddt(rho,e) - fvc::ddt(rhoE) == rhoESu) [:67-72] -- as written in the listing: rho*e advances by
            // the increment of rhoE, the kinetic energy is not taken out again.  consistentEnergy keeps the e of [:49]
            // (what fvc::ddt(rho,e), the form of the implicit branch [:57], would give with rhoESu = 0).
            if (opt.implicitDiffusion) {
                // solve(fvm::ddt(rho,e) - fvc::ddt(rho,e) - fvm::laplacian(alphauf,e) == rhoESu) [:55-61]; e's patches:
                // fixedEnergy = fixedValue, gradientEnergy = fixedGradient with a zero gradient here
                dvec rhs((size_t)m.nC), x = e.in;
                for (int ci = 0; ci < m.nC; ++ci) {
                    double src = rDeltaT * rhoOld[ci] * eOld[ci] * m.V[ci];
                    src += m.V[ci] * (rDeltaT * (rho.in[ci] * e.in[ci] - rhoOld[ci] * eOld[ci]));
                    rhs[ci] = src;
                }
                auto ic = [&](int ip, int gf, int, int) { return bc[ip].bcT == BC_FIXEDVALUE ? m.delta[gf] : 0.0; };
                auto bs = [&](int ip, int gf, int b, int) { return bc[ip].bcT == BC_FIXEDVALUE ? m.delta[gf] * e.bf[b] : 0.0; };
                lastIterE = implicitDiffusionSolve(alphauf, rDeltaT, rhs, ic, bs, x.data());
                e.in = x;
                correctBC_e();
                for (int ci = 0; ci < m.nC; ++ci) {
                    const double* u = &U.in[3 * (size_t)ci];
                    rhoE.in[ci] = rho.in[ci] * (e.in[ci] + 0.5 * (u[0] * u[0] + u[1] * u[1] + u[2] * u[2]));   // [:63]
                }
            } else if (!opt.consistentEnergy) for (int ci = 0; ci < m.nC; ++ci) {
                const double diag = rDeltaT * rho.in[ci] * m.V[ci];
                double src = rDeltaT * rhoOld[ci] * eOld[ci] * m.V[ci];
                src += m.V[ci] * (rDeltaT * (rhoE.in[ci] - rhoEOld[ci]));
                e.in[ci] = src / diag;
            }
            correctBC_e();
            for (int b = 0; b < m.nBF(); ++b) {
                const double* u = &U.bf[3 * (size_t)b];
                rhoE.bf[b] = rho.bf[b] * (e.bf[b] + 0.5 * (u[0] * u[0] + u[1] * u[1] + u[2] * u[2]));
            }
        }
        thermoCalculate();  // thermo.correct() [QGDFoam.C:149]
        for (int ci = 0; ci < m.nC; ++ci) p.in[ci] = rho.in[ci] / psi.in[ci];  // [:152-154]
        correctBC_p();                                                          // [:155]
        for (int b = 0; b < m.nBF(); ++b) rho.bf[b] = psi.bf[b] * p.bf[b];      // [:156]
    }
    void stepPhase2() {
        // after a halo unpack: tauQGDf is a pure function of the cell fields
        if (m.sharded()) computeTauQGDf();
    }

    // ---- the implicitDiffusion advance as phases (cell-range shards) ------------------------------------------------------------
    // Same protocol, control-block layout (68 doubles, slot-major ictl[slot * 4 + component]) and message kinds (1 fvc::grad(U),
    // 2 U, 3 search direction, 4 start values) as qgd_case_step_phase 20..35 of include/qgd_amd.h: the three velocity components advance in
    // lockstep as three right-hand sides, every sum runs over the OWNED cells and is reduced by the caller, ghost cells appear as
    // columns only.  The arithmetic is stepPhase1's (which stays the unsharded path).
    VolField gUold_, gUnew_;
    dvec rhoOld_, rhoUOld_, UOld_, rhoEOld_, eOld_, Ucur_, snU_;
    struct ISys { int NR = 0; dvec a, diag[3], rhs[3], x[3], r[3], d[3], q[3], A1[3]; bool valid[3] = {false, false, false}; } isys;
    double ictl[68] = {0};
    std::vector<char> ghostCell;
    bool ownedCell(int c) { if (ghostCell.empty() && m.sharded()) { ghostCell.assign(m.nC, 0); for (const ivec& gl : m.haloGhost) for (int g : gl) ghostCell[g] = 1; }
                            return ghostCell.empty() || !ghostCell[c]; }
    static int IC(int slot, int k) { return slot * 4 + k; }
    void patchValuesOfGrad(VolField& g, const VolField& f) {   // gaussGrad::correctBoundaryConditions on the patch values (L0)
        dvec sn = allPatchSnGrad(m, f);
        for (size_t ip = 0; ip < m.patches.size(); ++ip) {
            if (!m.patchHasFields((int)ip)) continue;
            for (int gf = m.patches[ip].start; gf < m.patches[ip].start + m.patches[ip].size; ++gf) {
                const int b = gf - m.nIF;
                double* gb = &g.bf[9 * (size_t)b];
                for (int k = 0; k < 9; ++k) gb[k] = g.in[9 * (size_t)m.own[gf] + k];
                if (m.coupled((int)ip)) continue;
                double n[3], ng[3];
                for (int k = 0; k < 3; ++k) n[k] = m.Sf[3 * (size_t)gf + k] / m.magSf[gf];
                VdotT(n, gb, ng);
                for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) gb[3 * i + j] += n[i] * (sn[3 * (size_t)b + j] - ng[j]);
            }
        }
    }
    template <class IC_, class BS_>
    void buildSystem(int k, const SurfField& gammaf, double rDeltaT, const dvec& rhsCell, IC_ icoef, BS_ bsrc) {
        isys.a.assign((size_t)m.nF, 0.0);
        for (int f = 0; f < m.nIF; ++f) isys.a[f] = gammaf.v[f] * m.magSf[f] * m.nonOrthDelta[f];
        dvec& diag = isys.diag[k];
        diag.assign((size_t)m.nC, 0.0);
        isys.rhs[k] = rhsCell;
        for (int ci = 0; ci < m.nC; ++ci) diag[ci] = rDeltaT * rho.in[ci] * m.V[ci];
        for (int f = 0; f < m.nIF; ++f) { diag[m.own[f]] += isys.a[f]; diag[m.nei[f]] += isys.a[f]; }
        for (size_t ip = 0; ip < bc.size(); ++ip) forPatchFaces((int)ip, [&](int gf, int b, int o) {
            if (m.coupled((int)ip)) return;
            const double gs = gammaf.v[gf] * m.magSf[gf];
            diag[o] += gs * icoef((int)ip, gf, b, o);
            isys.rhs[k][o] += gs * bsrc((int)ip, gf, b, o);
        });
    }
    void applySys(int k, const dvec& v, dvec& y) const {
        for (int c = 0; c < m.nC; ++c) y[c] = isys.diag[k][c] * v[c];
        for (int f = 0; f < m.nIF; ++f) { y[m.own[f]] -= isys.a[f] * v[m.nei[f]]; y[m.nei[f]] -= isys.a[f] * v[m.own[f]]; }
    }
    void solverPhase(int ph) {
        const int nC = m.nC, NR = isys.NR;
        const double tol = opt.implicitTol;
        const int maxIter = opt.implicitMaxIter;
        for (int k = 0; k < NR; ++k) {
            double* done = &ictl[IC(11, k)];
            if (ph == 0) {
                for (int s = 0; s < 16; ++s) ictl[IC(s, k)] = 0.0;
                *done = isys.valid[k] ? 0.0 : 3.0;
                if (!isys.valid[k]) continue;
                for (dvec* v : {&isys.r[k], &isys.d[k], &isys.q[k], &isys.A1[k]}) v->assign((size_t)nC, 0.0);
                dvec ones((size_t)nC, 1.0);
                applySys(k, ones, isys.A1[k]);
                applySys(k, isys.x[k], isys.q[k]);
                for (int c = 0; c < nC; ++c) {
                    isys.r[k][c] = isys.rhs[k][c] - isys.q[k][c];
                    if (ownedCell(c)) { ictl[IC(0, k)] += std::fabs(isys.r[k][c]); ictl[IC(1, k)] += isys.x[k][c]; ictl[IC(2, k)] += 1.0; }
                }
                continue;
            }
            if (*done != 0.0) continue;
            if (ph == 1) {
                const double xbar = ictl[IC(1, k)] / ictl[IC(2, k)];
                double t = 0;
                for (int c = 0; c < nC; ++c) if (ownedCell(c)) t += std::fabs(isys.q[k][c] - xbar * isys.A1[k][c]) + std::fabs(isys.rhs[k][c] - xbar * isys.A1[k][c]);
                ictl[IC(3, k)] = t;
            } else if (ph == 2) {
                const double nf = ictl[IC(3, k)] + 1e-20, res = ictl[IC(0, k)] / nf;
                ictl[IC(15, k)] = nf; ictl[IC(9, k)] = res; ictl[IC(10, k)] = res;
                if (res < tol || maxIter <= 0) { *done = 1.0; continue; }
                double rz = 0;
                for (int c = 0; c < nC; ++c) if (ownedCell(c)) { const double z = isys.r[k][c] / isys.diag[k][c]; isys.d[k][c] = z; rz += isys.r[k][c] * z; }
                ictl[IC(4, k)] = rz;
            } else if (ph == 3) {
                applySys(k, isys.d[k], isys.q[k]);
                double dq = 0;
                for (int c = 0; c < nC; ++c) if (ownedCell(c)) dq += isys.d[k][c] * isys.q[k][c];
                ictl[IC(5, k)] = dq;
            } else if (ph == 4) {
                const double dq = ictl[IC(5, k)], rz = ictl[IC(4, k)];
                if (!(dq > 0) || !(rz > 0)) { *done = 2.0; continue; }
                const double alpha = rz / dq;
                ictl[IC(13, k)] = alpha;
                double sa = 0, rzn = 0;
                for (int c = 0; c < nC; ++c) if (ownedCell(c)) {
                    isys.x[k][c] += alpha * isys.d[k][c];
                    isys.r[k][c] -= alpha * isys.q[k][c];
                    sa += std::fabs(isys.r[k][c]); rzn += isys.r[k][c] * (isys.r[k][c] / isys.diag[k][c]);
                }
                ictl[IC(6, k)] = sa; ictl[IC(7, k)] = rzn;
            } else if (ph == 5) {
                const double res = ictl[IC(6, k)] / ictl[IC(15, k)];
                ictl[IC(9, k)] = res; ictl[IC(12, k)] += 1.0;
                if (res < tol || ictl[IC(12, k)] >= (double)maxIter) { *done = 1.0; continue; }
                const double beta = ictl[IC(7, k)] / ictl[IC(4, k)];
                ictl[IC(14, k)] = beta; ictl[IC(4, k)] = ictl[IC(7, k)];
                for (int c = 0; c < nC; ++c) if (ownedCell(c)) isys.d[k][c] = isys.r[k][c] / isys.diag[k][c] + beta * isys.d[k][c];
            }
        }
        bool all = true;
        for (int k = 0; k < NR; ++k) all = all && ictl[IC(11, k)] != 0.0;
        ictl[64] = all ? 1.0 : 0.0;
    }
    void implicitPhase(int ph) {
        const int nC = m.nC, nF = m.nF;
        if (ph == 20) {
            setDeltaT();
            time += deltaT; stepCount++;
            rhoOld_ = rho.in; rhoUOld_ = rhoU.in; UOld_ = U.in; rhoEOld_ = rhoE.in; eOld_ = e.in;
            gUold_ = gaussGradVector(m, liveFace, U);
            return;
        }
        const double rDeltaT = 1.0 / deltaT;
        if (ph == 21) {
            // tauMC / phiTauMC from the (now complete) gradient [updateFluxes.H:107-111]
            patchValuesOfGrad(gUold_, U);
            VolField prod(m, 9);
            auto dev2T = [](const double* g, double muEff, double* out) {
                const double tr = g[0] + g[4] + g[8];
                for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
                    double a = g[3 * j + i];
                    if (i == j) a = a - (2.0 / 3.0) * tr;
                    out[3 * i + j] = muEff * a;
                }
            };
            for (int ci = 0; ci < nC; ++ci) dev2T(&gUold_.in[9 * (size_t)ci], 0.0 + mu.in[ci], &prod.in[9 * (size_t)ci]);
            for (int b = 0; b < m.nBF(); ++b) dev2T(&gUold_.bf[9 * (size_t)b], 0.0 + mu.bf[b], &prod.bf[9 * (size_t)b]);
            tauMC = linearInterpolate(m, prod);
            phiTauMC = SurfField(m, 3);
            for (int f = 0; f < nF; ++f) if (liveFace[f]) VdotT(&m.Sf[3 * (size_t)f], &tauMC.v[9 * (size_t)f], &phiTauMC.v[3 * (size_t)f]);
            // QGDRhoEqn.H, the explicit part of QGDUEqn.H [:36-51]
            {
                dvec d = fvcDiv(phiJm);
                for (int ci = 0; ci < nC; ++ci) {
                    const double diag = rDeltaT * m.V[ci];
                    double src = rDeltaT * rhoOld_[ci] * m.V[ci];
                    src -= m.V[ci] * d[ci];
                    rho.in[ci] = src / diag;
                }
            }
            dvec d1 = fvcDiv(phiJmU), d2 = fvcDiv(phiP), d3 = fvcDiv(phiPi);
            for (int ci = 0; ci < nC; ++ci)
                for (int k = 0; k < 3; ++k) {
                    const double diag = rDeltaT * m.V[ci];
                    double src = rDeltaT * rhoUOld_[3 * (size_t)ci + k] * m.V[ci];
                    src -= m.V[ci] * d1[3 * (size_t)ci + k];
                    src -= m.V[ci] * d2[3 * (size_t)ci + k];
                    src += m.V[ci] * d3[3 * (size_t)ci + k];
                    rhoU.in[3 * (size_t)ci + k] = src / diag;
                }
            for (int ci = 0; ci < nC; ++ci) for (int k = 0; k < 3; ++k) U.in[3 * (size_t)ci + k] = rhoU.in[3 * (size_t)ci + k] / rho.in[ci];
            correctBC_U();
            // the three systems of UEqn [:56-68]
            dvec dT = fvcDiv(phiTauMC);
            snU_ = allPatchSnGrad(m, U);
            Ucur_ = U.in;
            isys.NR = 3;
            for (int k = 0; k < 3; ++k) {
                isys.valid[k] = !(m.geomD[k] < 0);
                dvec rhs((size_t)nC);
                isys.x[k].assign((size_t)nC, 0.0);
                for (int ci = 0; ci < nC; ++ci) {
                    double src = rDeltaT * rhoOld_[ci] * UOld_[3 * (size_t)ci + k] * m.V[ci];
                    src += m.V[ci] * (rDeltaT * (rho.in[ci] * Ucur_[3 * (size_t)ci + k] - rhoOld_[ci] * UOld_[3 * (size_t)ci + k]));
                    src += m.V[ci] * dT[3 * (size_t)ci + k];
                    rhs[ci] = src;
                    isys.x[k][ci] = Ucur_[3 * (size_t)ci + k];
                }
                auto ic = [&](int ip, int gf, int, int) {
                    if (bc[ip].bcU == BC_FIXEDVALUE) return m.delta[gf];
                    if (bc[ip].bcU == BC_SLIP) return m.delta[gf] * symmAbsN(m, (int)ip, gf, k);
                    return 0.0;
                };
                auto bs = [&](int ip, int gf, int b, int o) {
                    if (bc[ip].bcU == BC_FIXEDVALUE) return m.delta[gf] * U.bf[3 * (size_t)b + k];
                    if (bc[ip].bcU == BC_SLIP)
                        return snU_[3 * (size_t)b + k] + m.delta[gf] * symmAbsN(m, (int)ip, gf, k) * Ucur_[3 * (size_t)o + k];
                    return 0.0;
                };
                buildSystem(k, muf, rDeltaT, rhs, ic, bs);
            }
        } else if (ph >= 22 && ph <= 27) solverPhase(ph - 22);
        else if (ph == 28) {
            for (int k = 0; k < 3; ++k) {
                lastIterU[k] = (int)ictl[IC(12, k)];
                if (isys.valid[k]) for (int ci = 0; ci < nC; ++ci) if (ownedCell(ci)) U.in[3 * (size_t)ci + k] = isys.x[k][ci];
            }
            correctBC_U();
        } else if (ph == 29) {
            for (int ci = 0; ci < nC; ++ci) for (int k = 0; k < 3; ++k) rhoU.in[3 * (size_t)ci + k] = rho.in[ci] * U.in[3 * (size_t)ci + k];   // [:70]
            gUnew_ = gaussGradVector(m, liveFace, U);
        } else if (ph == 30) {
            patchValuesOfGrad(gUnew_, U);
            SurfField gUf = linearInterpolate(m, gUnew_);
            phiSigmaDotU = SurfField(m, 1);
            for (int f = 0; f < nF; ++f) {
                if (!liveFace[f]) continue;
                double A[9], sd[3];
                for (int q = 0; q < 9; ++q) A[q] = muf.v[f] * gUf.v[9 * (size_t)f + q] + tauMC.v[9 * (size_t)f + q];
                TdotV(A, &Uf.v[3 * (size_t)f], sd);
                phiSigmaDotU.v[f] = dot3(&m.Sf[3 * (size_t)f], sd);
            }
            for (int b = 0; b < m.nBF(); ++b) for (int k = 0; k < 3; ++k) rhoU.bf[3 * (size_t)b + k] = rho.bf[b] * U.bf[3 * (size_t)b + k];
            // QGDEEqn.H [:37-50]
            dvec d1 = fvcDiv(phiJmH), d2 = fvcDiv(phiQ), d3 = fvcDiv(phiPiU), d4 = fvcDiv(phiSigmaDotU);
            for (int ci = 0; ci < nC; ++ci) {
                const double diag = rDeltaT * m.V[ci];
                double src = rDeltaT * rhoEOld_[ci] * m.V[ci];
                src -= m.V[ci] * d1[ci];
                src -= m.V[ci] * d2[ci];
                src += m.V[ci] * d3[ci];
                src += m.V[ci] * d4[ci];
                rhoE.in[ci] = src / diag;
            }
            for (int ci = 0; ci < nC; ++ci) {
                const double* u = &U.in[3 * (size_t)ci];
                e.in[ci] = rhoE.in[ci] / rho.in[ci] - 0.5 * (u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
            }
            correctBC_e();
            // the e system [:55-61]
            dvec rhs((size_t)nC);
            isys.NR = 1; isys.valid[0] = true;
            isys.x[0] = e.in;
            for (int ci = 0; ci < nC; ++ci) {
                double src = rDeltaT * rhoOld_[ci] * eOld_[ci] * m.V[ci];
                src += m.V[ci] * (rDeltaT * (rho.in[ci] * e.in[ci] - rhoOld_[ci] * eOld_[ci]));
                rhs[ci] = src;
            }
            auto ic = [&](int ip, int gf, int, int) { return bc[ip].bcT == BC_FIXEDVALUE ? m.delta[gf] : 0.0; };
            auto bs = [&](int ip, int gf, int b, int) { return bc[ip].bcT == BC_FIXEDVALUE ? m.delta[gf] * e.bf[b] : 0.0; };
            buildSystem(0, alphauf, rDeltaT, rhs, ic, bs);
        } else if (ph == 35) {
            lastIterE = (int)ictl[IC(12, 0)];
            for (int ci = 0; ci < nC; ++ci) if (ownedCell(ci)) e.in[ci] = isys.x[0][ci];
            correctBC_e();
            for (int ci = 0; ci < nC; ++ci) {
                const double* u = &U.in[3 * (size_t)ci];
                rhoE.in[ci] = rho.in[ci] * (e.in[ci] + 0.5 * (u[0] * u[0] + u[1] * u[1] + u[2] * u[2]));   // [:63]
            }
            correctBC_e();
            for (int b = 0; b < m.nBF(); ++b) {
                const double* u = &U.bf[3 * (size_t)b];
                rhoE.bf[b] = rho.bf[b] * (e.bf[b] + 0.5 * (u[0] * u[0] + u[1] * u[1] + u[2] * u[2]));
            }
            thermoCalculate();
            for (int ci = 0; ci < nC; ++ci) p.in[ci] = rho.in[ci] / psi.in[ci];
            correctBC_p();
            for (int b = 0; b < m.nBF(); ++b) rho.bf[b] = psi.bf[b] * p.bf[b];
        }
    }
    int implHaloWidth(int kind) const { return kind == 1 ? 9 : (kind == 2 ? 3 : isys.NR); }
    void implHaloMove(int side, int kind, double* buf, bool pack) {
        if (side < 0 || (size_t)side >= m.haloGhost.size()) return;
        const ivec& cells = pack ? m.haloSend[side] : m.haloGhost[side];
        size_t q = 0;
        auto io = [&](double& x) { if (pack) buf[q++] = x; else x = buf[q++]; };
        VolField& g = gUnew_.in.size() && kind == 1 && phaseNew_ ? gUnew_ : gUold_;
        for (int ci : cells) {
            if (kind == 1) for (int k = 0; k < 9; ++k) io(g.in[9 * (size_t)ci + k]);
            else if (kind == 2) for (int k = 0; k < 3; ++k) io(U.in[3 * (size_t)ci + k]);
            else if (kind == 3) for (int k = 0; k < isys.NR; ++k) { if (isys.d[k].size() != (size_t)m.nC) isys.d[k].assign((size_t)m.nC, 0.0); io(isys.d[k][ci]); }
            else for (int k = 0; k < isys.NR; ++k) io(isys.x[k][ci]);
        }
    }
    bool phaseNew_ = false;   // which gradient message kind 1 carries: the old state's (after phase 20) or the new velocity's (after 28)

    static const int kCellMsg = 15, kFaceMsg = 16;
    void haloCount(int side, int64_t* n, bool recv) const {
        *n = 0;
        if (side < 0 || (size_t)side >= m.haloGhost.size()) return;
        const ivec& cells = recv ? m.haloGhost[side] : m.haloSend[side];
        const ivec& faces = recv ? m.haloGhostBF[side] : m.haloSendBF[side];
        *n = (int64_t)cells.size() * kCellMsg + (int64_t)faces.size() * kFaceMsg;
    }
    void packOrUnpack(int side, double* buf, bool pack) {
        if (side < 0 || (size_t)side >= m.haloGhost.size()) return;
        const ivec& cellsL = pack ? m.haloSend[side] : m.haloGhost[side];
        const ivec& facesL = pack ? m.haloSendBF[side] : m.haloGhostBF[side];
        size_t q = 0;
        auto io = [&](double& x) { if (pack) buf[q++] = x; else x = buf[q++]; };
        for (int ci : cellsL) {
            io(rho.in[ci]); for (int k = 0; k < 3; ++k) io(U.in[3 * (size_t)ci + k]); for (int k = 0; k < 3; ++k) io(rhoU.in[3 * (size_t)ci + k]);
            io(rhoE.in[ci]); io(e.in[ci]); io(T.in[ci]); io(p.in[ci]); io(psi.in[ci]); io(mu.in[ci]); io(alpha.in[ci]); io(c.in[ci]);
        }
        for (int b : facesL) {
            io(rho.bf[b]); for (int k = 0; k < 3; ++k) io(U.bf[3 * (size_t)b + k]); for (int k = 0; k < 3; ++k) io(rhoU.bf[3 * (size_t)b + k]);
            io(rhoE.bf[b]); io(e.bf[b]); io(T.bf[b]); io(p.bf[b]); io(psi.bf[b]); io(mu.bf[b]); io(alpha.bf[b]); io(c.bf[b]); io(p.grad[b]);
        }
    }
};

}  // namespace

// ---------------------------------------------------------------------------
// C interface
// ---------------------------------------------------------------------------
extern "C" int orc_qhd_pressure(void* mp, const double* phiu, const double* phiwo, const double* taubyrhof, const int32_t* patchKind,
                                const double* pb, const double* gradb, double tolerance, double relTol, int32_t maxIter, int32_t pRefCell,
                                double pRefValue, double* p, double* phi, double info[3]);

// ---------------------------------------------------------------------------
// QHDFoam case: the loop body of QHDFoam.C L83-139, both branches of implicitDiffusion [QHDUEqn.H L46-85, QHDTEqn.H L69-92]
// (step(); the phase form for shards restates the explicit branch only), with rhoConst +
// constTransport thermo (rho, mu, alpha = mu/Pr uniform; the QHD closures keep muQGD = alphauQGD = 0 [T0byGr.C L62-72])
// and laminar transport.  thermo.correct() is not called inside the loop [QHDFoam.C L83-139], so rho, mu, alpha and
// tauQGDf are those of start-up.
// L0 pieces restated here: fvc::grad (Gauss linear + gaussGrad::correctBoundaryConditions), fvc::laplacian (Gauss,
// uncorrected snGrad), fvc::div (surfaceIntegrate), Euler ddt, setReference / needReference.
// ---------------------------------------------------------------------------
struct QhdCase {
    MeshHandle* mh;
    const Mesh& m;
    orc_qhd_options opt;
    std::vector<PatchBC> bc;
    Stencil* stencil = nullptr;
    std::string word;
    VolField U, T, p;
    SurfField tauQGDf, phi, phiu, phiwo;
    dvec hQGD, hQGDb, hQGDf;
    std::vector<char> liveFace;
    double time = 0;
    int64_t steps = 0;
    double lastPIter = 0, lastPRes0 = 0, lastPRes = 0;
    int lastIterU[3] = {0, 0, 0}, lastIterT = 0;   // implicitDiffusion: iterations of the U and T solves of the last step

    QhdCase(MeshHandle* h, const orc_qhd_options& o) : mh(h), m(h->m), opt(o), bc(h->m.patches.size()) {
        for (size_t ip = 0; ip < bc.size(); ++ip) constraintKinds(m.patches[ip].type, bc[ip].bcU, bc[ip].bcT, bc[ip].bcP);
        liveFace.assign(m.nF, 1);
        for (size_t ip = 0; ip < m.patches.size(); ++ip)
            if (!m.patchHasFields((int)ip))
                for (int f = m.patches[ip].start; f < m.patches[ip].start + m.patches[ip].size; ++f) liveFace[f] = 0;
    }
    template <class Fn> void forPatchFaces(int ip, Fn fn) const {
        if (!m.patchHasFields(ip)) return;
        const PatchInfo& pi = m.patches[ip];
        for (int gf = pi.start; gf < pi.start + pi.size; ++gf) fn(gf, gf - m.nIF, m.own[gf]);
    }
    void correctU() {
        for (size_t ip = 0; ip < bc.size(); ++ip) forPatchFaces((int)ip, [&](int gf, int b, int o) {
            const PatchBC& B = bc[ip];
            if (B.bcU == BC_FIXEDVALUE) { for (int k = 0; k < 3; ++k) U.bf[3 * (size_t)b + k] = B.vU[k]; }
            else if (B.bcU == BC_SLIP) {
                double n[3], Tm[9], tv[3];
                m.symmNormal((int)ip, gf, n);
                for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Tm[3 * i + j] = (i == j ? 1.0 : 0.0) - 2.0 * (n[i] * n[j]);
                TdotV(Tm, &U.in[3 * (size_t)o], tv);
                for (int k = 0; k < 3; ++k) U.bf[3 * (size_t)b + k] = (U.in[3 * (size_t)o + k] + tv[k]) / 2.0;
            } else { for (int k = 0; k < 3; ++k) U.bf[3 * (size_t)b + k] = U.in[3 * (size_t)o + k]; }
        });
    }
    void correctT() {
        for (size_t ip = 0; ip < bc.size(); ++ip) forPatchFaces((int)ip, [&](int, int b, int o) {
            T.bf[b] = bc[ip].bcT == BC_FIXEDVALUE ? bc[ip].vT : T.in[o];
        });
    }
    void correctP() {  // fixedValue | fixedGradient (qhdFlux keeps the gradient of its file: the registry lookup of
                       // "phiwStar" finds nothing in QHDFoam [qhdFluxFvPatchScalarField.C L166-168]) | zeroGradient
        for (size_t ip = 0; ip < bc.size(); ++ip) forPatchFaces((int)ip, [&](int gf, int b, int o) {
            const PatchBC& B = bc[ip];
            if (B.bcP == BC_FIXEDVALUE) p.bf[b] = B.vP;
            else if (B.bcP == BC_QGDFLUX) { p.grad[b] = B.vP; p.bf[b] = p.in[o] + p.grad[b] / m.delta[gf]; }
            else if (B.bcP == BC_QHDFLUX) {
                // qhdFlux with its flux registered (as the QHD solvers that register "phiwStar" do, e.g.
                // mulesQHDFoam/createFields.H): gradient = -(phiw/tauQGDf*rhof/|Sf|) [qhdFluxFvPatchScalarField.C L193-203]
                p.grad[b] = -(phiwo.v[gf] / tauQGDf.v[gf] * opt.rho0 / m.magSf[gf]);
                p.bf[b] = p.in[o] + p.grad[b] / m.delta[gf];
            } else p.bf[b] = p.in[o];
        });
    }
    // QGDCoeffs lengths [QGDCoeffs.C L195-199, L298-376] and the QHD closures [constTau.C L71-74, HbyUQHD.C L80-83,
    // T0byGr.C L84-87, H2bynuQHD.C L78-82]: tauQGD per cell and patch face, tauQGDf = linearInterpolate(tauQGD)
    void initTau() {
        hQGDf.assign(m.nF, 0.0); hQGD.assign(m.nC, 0.0); hQGDb.assign(m.nBF(), 0.0);
        for (int f = 0; f < m.nF; ++f) hQGDf[f] = (m.delta[f] != 0.0) ? 1.0 / std::fabs(m.delta[f]) : 0.0;
        for (int f = 0; f < m.nIF; ++f) {
            double a[3], b[3];
            for (int k = 0; k < 3; ++k) { a[k] = m.C[3 * (size_t)m.own[f] + k] - m.Cf[3 * (size_t)f + k]; b[k] = m.C[3 * (size_t)m.nei[f] + k] - m.Cf[3 * (size_t)f + k]; }
            hQGDf[f] = 2.0 * std::min(mag3(a), mag3(b));
        }
        for (size_t ip = 0; ip < m.patches.size(); ++ip)
            if (!m.coupled((int)ip)) forPatchFaces((int)ip, [&](int gf, int, int) { hQGDf[gf] *= 2.0; });
        for (int f = 0; f < m.nF; ++f) if (!liveFace[f]) hQGDf[f] = 0.0;
        {   // cut-plane faces are internal faces of the unsharded mesh: a ghost cell's hQGD needs their value THERE
            size_t k = 0;
            for (size_t ip = 0; ip < m.patches.size(); ++ip)
                if (m.patches[ip].type == PATCH_HALO)
                    for (int gf = m.patches[ip].start; gf < m.patches[ip].start + m.patches[ip].size && k < m.haloFaceH.size(); ++gf) hQGDf[gf] = m.haloFaceH[k++];
        }
        for (int ci = 0; ci < m.nC; ++ci) {
            double hint = 0, surf = 0;
            for (int fid : m.cells[ci]) {
                if (fid >= m.nIF) {
                    int pid = -1;
                    for (size_t ip = 0; ip < m.patches.size(); ++ip)
                        if (fid >= m.patches[ip].start && fid < m.patches[ip].start + m.patches[ip].size) pid = (int)ip;
                    if (pid < 0 || m.patches[pid].type == PATCH_EMPTY || m.patches[pid].type == PATCH_WEDGE) continue;
                }
                hint += hQGDf[fid] * m.magSf[fid];
                surf += m.magSf[fid];
            }
            hQGD[ci] = hint / surf;
        }
        for (size_t ip = 0; ip < m.patches.size(); ++ip) forPatchFaces((int)ip, [&](int gf, int b, int) { hQGDb[b] = hQGDf[gf] * 1.0; });
        VolField tau(m, 1);
        const double nu = opt.mu / opt.rho0;
        auto closure = [&](double h) {
            switch (opt.tauModel) {
                case 0: return opt.Tau;
                case 1: return opt.aQGD * h / opt.UQHD;
                case 2: return opt.T0 / opt.Gr;
                default: return opt.aQGD * h * h / nu;
            }
        };
        for (int ci = 0; ci < m.nC; ++ci) tau.in[ci] = closure(hQGD[ci]);
        for (int b = 0; b < m.nBF(); ++b) tau.bf[b] = closure(hQGDb[b]);
        tauQGDf = linearInterpolate(m, tau);
    }
    int setFields(const double* U0, const double* T0, const double* p0) {
        int rc = mh->cache.lookup(m, word, &stencil);
        if (rc) return rc;
        U = VolField(m, 3); T = VolField(m, 1); p = VolField(m, 1);
        p.grad.assign(m.nBF(), 0.0);
        p.snKind.assign(bc.size(), SN_GENERIC); T.snKind = p.snKind; U.snKind = p.snKind;
        for (size_t ip = 0; ip < bc.size(); ++ip) {
            p.snKind[ip] = bc[ip].bcP == BC_FIXEDVALUE ? SN_GENERIC : ((bc[ip].bcP == BC_QGDFLUX || bc[ip].bcP == BC_QHDFLUX) ? SN_GRADIENT : SN_ZERO);
            T.snKind[ip] = bc[ip].bcT == BC_FIXEDVALUE ? SN_GENERIC : SN_ZERO;
            U.snKind[ip] = bc[ip].bcU == BC_FIXEDVALUE ? SN_GENERIC : (bc[ip].bcU == BC_SLIP ? SN_SYMM : SN_ZERO);
            if (m.patches[ip].type == PATCH_HALO) p.snKind[ip] = T.snKind[ip] = U.snKind[ip] = SN_ZERO;
        }
        std::copy(U0, U0 + 3 * (size_t)m.nC, U.in.begin());
        std::copy(T0, T0 + m.nC, T.in.begin());
        std::copy(p0, p0 + m.nC, p.in.begin());
        phi = SurfField(m, 1); phiu = SurfField(m, 1); phiwo = SurfField(m, 1);
        initTau();
        correctU(); correctT(); correctP();
        time = 0; steps = 0;
        gUValid = false;
        return 0;
    }
    // L0 fvc::grad(U), Gauss linear, with gaussGrad::correctBoundaryConditions on the patch values
    VolField gaussGradV(const VolField& f) const {
        SurfField ff = linearInterpolate(m, f);
        VolField g(m, 9);
        for (int fc = 0; fc < m.nF; ++fc) {
            if (!liveFace[fc]) continue;
            const double* S = &m.Sf[3 * (size_t)fc];
            double* go = &g.in[9 * (size_t)m.own[fc]];
            for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) go[3 * i + j] += S[i] * ff.v[3 * (size_t)fc + j];
            if (fc < m.nIF) {
                double* gn = &g.in[9 * (size_t)m.nei[fc]];
                for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) gn[3 * i + j] -= S[i] * ff.v[3 * (size_t)fc + j];
            }
        }
        for (int c = 0; c < m.nC; ++c) for (int k = 0; k < 9; ++k) g.in[9 * (size_t)c + k] /= m.V[c];
        dvec sn = allPatchSnGrad(m, f);
        for (size_t ip = 0; ip < m.patches.size(); ++ip) {
            if (!m.patchHasFields((int)ip)) continue;
            for (int gf = m.patches[ip].start; gf < m.patches[ip].start + m.patches[ip].size; ++gf) {
                const int b = gf - m.nIF;
                double* gb = &g.bf[9 * (size_t)b];
                for (int k = 0; k < 9; ++k) gb[k] = g.in[9 * (size_t)m.own[gf] + k];   // extrapolatedCalculated
                if (m.coupled((int)ip)) continue;
                double n[3], ng[3];
                for (int k = 0; k < 3; ++k) n[k] = m.Sf[3 * (size_t)gf + k] / m.magSf[gf];
                VdotT(n, gb, ng);
                for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) gb[3 * i + j] += n[i] * (sn[3 * (size_t)b + j] - ng[j]);
            }
        }
        return g;
    }
    void step() {
        const int nC = m.nC, nF = m.nF, nB = m.nBF();
        const double dt = opt.deltaT;
        // updateFields.H L36-73
        SurfField gradUf = stencil->gradV(U), gradTf = stencil->gradS(T);
        SurfField Uf = linearInterpolate(m, U), Tf = linearInterpolate(m, T);
        VolField BdFrc(m, 3);
        for (int c = 0; c < nC; ++c) for (int k = 0; k < 3; ++k) BdFrc.in[3 * (size_t)c + k] = (opt.beta * T.in[c]) * opt.g[k];
        for (int b = 0; b < nB; ++b) for (int k = 0; k < 3; ++k) BdFrc.bf[3 * (size_t)b + k] = (opt.beta * T.bf[b]) * opt.g[k];
        SurfField BdFrcf = linearInterpolate(m, BdFrc);
        const double rhof = opt.rho0, muf = opt.mu, alphaf = opt.mu / opt.Pr;   // uniform: interpolation returns the value
        const double Hif = alphaf / rhof, nuf = muf / rhof;
        // updateFluxes.H L33-38
        SurfField UgU(m, 3), taubyrhof(m, 1), phiTauTReg(m, 1);
        for (int f = 0; f < nF; ++f) {
            if (!liveFace[f]) continue;
            const double* S = &m.Sf[3 * (size_t)f];
            phiu.v[f] = dot3(S, &Uf.v[3 * (size_t)f]);
            VdotT(&Uf.v[3 * (size_t)f], &gradUf.v[9 * (size_t)f], &UgU.v[3 * (size_t)f]);
            double wo[3];
            for (int k = 0; k < 3; ++k) wo[k] = tauQGDf.v[f] * (UgU.v[3 * (size_t)f + k] - BdFrcf.v[3 * (size_t)f + k]);
            phiwo.v[f] = dot3(S, wo);
            taubyrhof.v[f] = tauQGDf.v[f] / rhof;
            phiTauTReg.v[f] = tauQGDf.v[f] * phiu.v[f] * dot3(&Uf.v[3 * (size_t)f], &gradTf.v[3 * (size_t)f]);   // QHDTEqn.H L66
        }
        time += dt;
        // QHDpEqn.H L35-47
        correctP();
        std::vector<int32_t> kinds(bc.size());
        for (size_t ip = 0; ip < bc.size(); ++ip) kinds[ip] = bc[ip].bcP;
        dvec pb = p.bf, gb = p.grad;
        if (pb.empty()) { pb.assign(1, 0.0); gb.assign(1, 0.0); }
        double info[3];
        orc_qhd_pressure(mh, phiu.v.data(), phiwo.v.data(), taubyrhof.v.data(), kinds.data(), pb.data(), gb.data(), opt.pTol, opt.pRelTol,
                         opt.pMaxIter, opt.pRefCell, p.in[std::max(opt.pRefCell, 0)], p.in.data(), phi.v.data(), info);
        lastPIter = info[0]; lastPRes0 = info[1]; lastPRes = info[2];
        correctP();   // fvMatrix::solve ends in correctBoundaryConditions()
        // QHDUEqn.H L36-84 (explicit branch)
        SurfField gradPf = stencil->gradS(p);
        SurfField pf = linearInterpolate(m, p);
        VolField gU = gaussGradV(U);
        VolField gUT(m, 9);
        for (int c = 0; c < nC; ++c) for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) gUT.in[9 * (size_t)c + 3 * i + j] = gU.in[9 * (size_t)c + 3 * j + i];
        for (int b = 0; b < nB; ++b) for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) gUT.bf[9 * (size_t)b + 3 * i + j] = gU.bf[9 * (size_t)b + 3 * j + i];
        SurfField gUTf = linearInterpolate(m, gUT);
        SurfField snU = fvcSnGrad(m, U), snT = fvcSnGrad(m, T);
        const bool implicit = opt.implicitDiffusion != 0;
        // qgdFlux(phi,U,Uf) [QHDUEqn.H L41], qgdFlux(phi,T,Tf) [QHDTEqn.H L65]: fvc::flux when divSchemes has the flux's entry
        const SurfField UfFlux = opt.fluxSchemeU ? upwindInterpolate(m, phi, U) : Uf, TfFlux = opt.fluxSchemeT ? upwindInterpolate(m, phi, T) : Tf;
        dvec FU(3 * (size_t)nF, 0.0), FT((size_t)nF, 0.0), Gp(3 * (size_t)nF, 0.0);
        for (int f = 0; f < nF; ++f) {
            if (!liveFace[f]) continue;
            const double* S = &m.Sf[3 * (size_t)f];
            double Wf[3], UW[9], uw[3], ext[3];
            for (int k = 0; k < 3; ++k)
                Wf[k] = tauQGDf.v[f] * ((UgU.v[3 * (size_t)f + k] + gradPf.v[3 * (size_t)f + k] / rhof) - BdFrcf.v[3 * (size_t)f + k]);   // L37
            outer(&Uf.v[3 * (size_t)f], Wf, UW);
            VdotT(S, UW, uw);                                                                          // L39
            VdotT(S, &gUTf.v[9 * (size_t)f], ext);                                                     // Sf & lin(T(grad U)), L56 / L76
            for (int k = 0; k < 3; ++k) {
                const double phiUf = phi.v[f] * UfFlux.v[3 * (size_t)f + k] - uw[k];                   // L41-43
                const double lap = nuf * snU.v[3 * (size_t)f + k] * m.magSf[f];                        // fvc::laplacian(muf/rhof, U), L74
                FU[3 * (size_t)f + k] = implicit ? phiUf - nuf * ext[k] : (phiUf - lap) - nuf * ext[k];   // L54: the laplacian is in the matrix
                Gp[3 * (size_t)f + k] = S[k] * pf.v[f];                                                // fvc::grad(p), Gauss linear
            }
            FT[f] = implicit ? phi.v[f] * TfFlux.v[f] - phiTauTReg.v[f]                                // QHDTEqn.H L73-76
                             : (phi.v[f] * TfFlux.v[f] - Hif * snT.v[f] * m.magSf[f]) - phiTauTReg.v[f];   // QHDTEqn.H L65-66, L85-88
        }
        dvec sumU(3 * (size_t)nC, 0.0), sumT((size_t)nC, 0.0), sumG(3 * (size_t)nC, 0.0);
        for (int f = 0; f < nF; ++f) {   // surfaceIntegrate order
            if (!liveFace[f]) continue;
            const int o = m.own[f];
            for (int k = 0; k < 3; ++k) { sumU[3 * (size_t)o + k] += FU[3 * (size_t)f + k]; sumG[3 * (size_t)o + k] += Gp[3 * (size_t)f + k]; }
            sumT[o] += FT[f];
            if (f < m.nIF) {
                const int n = m.nei[f];
                for (int k = 0; k < 3; ++k) { sumU[3 * (size_t)n + k] -= FU[3 * (size_t)f + k]; sumG[3 * (size_t)n + k] -= Gp[3 * (size_t)f + k]; }
                sumT[n] -= FT[f];
            }
        }
        if (!implicit) {
            for (int c = 0; c < nC; ++c) {
                const double rV = 1.0 / m.V[c];
                for (int k = 0; k < 3; ++k)
                    U.in[3 * (size_t)c + k] += dt * ((-(sumU[3 * (size_t)c + k] * rV) - (sumG[3 * (size_t)c + k] * rV) / opt.rho0) + BdFrc.in[3 * (size_t)c + k]);
                T.in[c] += dt * (-(sumT[c] * rV));
            }
        } else {
            // fvm::ddt(U) - fvm::laplacian(muf/rhof, U) per component [QHDUEqn.H L48-64] and fvm::ddt(T) - fvm::laplacian(Hif, T)
            // [QHDTEqn.H L71-79]: face coefficients gamma |Sf| delta_f (Gauss linear uncorrected, L0), diagonal V/deltaT + sum a_f +
            // the patch internal coefficients; patch coefficients of -fvm::laplacian (L0): fixedValue delta / delta*value;
            // basicSymmetry (slip) delta*|n_k| / snGrad_k + delta*|n_k|*patchInternalField_k (transformFvPatchField);
            // zeroGradient none.  Components along empty directions are not solved (fvMatrix<vector>::solveSegregated, L0).
            const double rDeltaT = 1.0 / dt;
            dvec gS((size_t)nF, 0.0);
            for (int f = 0; f < nF; ++f) gS[f] = m.magSf[f] * (f < m.nIF ? m.nonOrthDelta[f] : m.delta[f]);
            const dvec snUb = allPatchSnGrad(m, U);
            const dvec Ucur = U.in;
            auto solveOne = [&](const double gamma, const dvec& rhsCell, auto icoef, auto bsrc, double* x) {
                dvec a((size_t)nF, 0.0), diag((size_t)nC), rhs = rhsCell;
                for (int f = 0; f < m.nIF; ++f) a[f] = gamma * gS[f];
                for (int c = 0; c < nC; ++c) diag[c] = rDeltaT * m.V[c];
                for (int f = 0; f < m.nIF; ++f) { diag[m.own[f]] += a[f]; diag[m.nei[f]] += a[f]; }
                for (size_t ip = 0; ip < bc.size(); ++ip) forPatchFaces((int)ip, [&](int gf, int b, int o) {
                    if (m.coupled((int)ip)) return;
                    const double gs = gamma * m.magSf[gf];
                    diag[o] += gs * icoef((int)ip, gf, b, o);
                    rhs[o] += gs * bsrc((int)ip, gf, b, o);
                });
                return solveDiagLaplacian(m, a, diag, rhs, x, opt.implicitTol, opt.implicitMaxIter);
            };
            for (int k = 0; k < 3; ++k) {
                if (m.geomD[k] < 0) continue;
                dvec rhs((size_t)nC), x((size_t)nC);
                for (int c = 0; c < nC; ++c) {
                    const double rV = 1.0 / m.V[c];
                    // rDeltaT U.old V  +  V (-div(phiUf - nu Sf & lin(T(grad U))) - grad(p)/rho + BdFrc)
                    rhs[c] = rDeltaT * Ucur[3 * (size_t)c + k] * m.V[c]
                             + m.V[c] * ((-(sumU[3 * (size_t)c + k] * rV) - (sumG[3 * (size_t)c + k] * rV) / opt.rho0) + BdFrc.in[3 * (size_t)c + k]);
                    x[c] = Ucur[3 * (size_t)c + k];
                }
                auto ic = [&](int ip, int gf, int, int) {
                    if (bc[ip].bcU == BC_FIXEDVALUE) return m.delta[gf];
                    if (bc[ip].bcU == BC_SLIP) return m.delta[gf] * symmAbsN(m, (int)ip, gf, k);
                    return 0.0;
                };
                auto bs = [&](int ip, int gf, int b, int o) {
                    if (bc[ip].bcU == BC_FIXEDVALUE) return m.delta[gf] * U.bf[3 * (size_t)b + k];
                    if (bc[ip].bcU == BC_SLIP)
                        return snUb[3 * (size_t)b + k] + m.delta[gf] * symmAbsN(m, (int)ip, gf, k) * Ucur[3 * (size_t)o + k];
                    return 0.0;
                };
                lastIterU[k] = solveOne(nuf, rhs, ic, bs, x.data());
                for (int c = 0; c < nC; ++c) U.in[3 * (size_t)c + k] = x[c];
            }
            {
                dvec rhs((size_t)nC), x((size_t)nC);
                for (int c = 0; c < nC; ++c) {
                    const double rV = 1.0 / m.V[c];
                    rhs[c] = rDeltaT * T.in[c] * m.V[c] + m.V[c] * (-(sumT[c] * rV));
                    x[c] = T.in[c];
                }
                auto ic = [&](int ip, int gf, int, int) { return bc[ip].bcT == BC_FIXEDVALUE ? m.delta[gf] : 0.0; };
                auto bs = [&](int ip, int gf, int b, int) { return bc[ip].bcT == BC_FIXEDVALUE ? m.delta[gf] * T.bf[b] : 0.0; };
                lastIterT = solveOne(Hif, rhs, ic, bs, x.data());
                for (int c = 0; c < nC; ++c) T.in[c] = x[c];
            }
        }
        correctU(); correctT();
        // QHDFoam.C L123-130
        bool anyFixed = false;
        for (size_t ip = 0; ip < bc.size(); ++ip) anyFixed = anyFixed || (bc[ip].bcP == BC_FIXEDVALUE && m.patches[ip].size > 0 && m.patches[ip].type == PATCH_GENERIC);
        if (!anyFixed && opt.pRefCell >= 0) {
            const double shift = opt.pRefValue - p.in[opt.pRefCell];
            for (double& x : p.in) x += shift;
            for (int b = 0; b < nB; ++b) p.bf[b] += shift;
        }
        ++steps;
    }

    // ---- the same step as phases, for cell-range shards (one rank per shard) --------------------------------------------
    // What OpenFOAM does through processor patches inside fvm::laplacian / PCG / fvc::grad under MPI [QHDpEqn.H L35-47,
    // QHDUEqn.H L36-84]: between the phases the caller SUMS the named ctl slots over the ranks and exchanges the halo messages
    // (kind 0: {U,T} per cell and patch face; kind 1: p + fvc::grad(U) per cell, p's patch value and gradient; kind 2: the
    // search direction).  Same protocol, slots and message layouts as qgd_qhd_case_step_phase of include/qgd_amd.h.
    // The preconditioner is Jacobi (local and exact under sharding), as in orc_qhd_pressure.
    SurfField UgU_, BdFrcf_, Uf_, Tf_, phiTauTReg_, taubyrhof_;
    VolField BdFrc_, gU_;
    bool gUValid = false;
    dvec pa_, pdiag_, prhs_, pr_, pz_, pd_, pq_, pA1_;
    double ctl[16] = {0};
    double normF_ = 0;
    bool refSet = false, needRef_ = false;
    int refLocal_ = -1;
    std::vector<char> pkind_, liveB_;
    bool isOwned(int c) const { return ghostFlag.empty() || !ghostFlag[c]; }
    std::vector<char> ghostFlag;
    void ensureShardInfo() {
        if (ghostFlag.empty() && m.sharded()) {
            ghostFlag.assign(m.nC, 0);
            for (const ivec& gl : m.haloGhost) for (int g : gl) ghostFlag[g] = 1;
        }
        if (!refSet) {   // unsharded default: the rule of step()
            bool anyFixed = false;
            for (size_t ip = 0; ip < bc.size(); ++ip) anyFixed = anyFixed || (bc[ip].bcP == BC_FIXEDVALUE && m.patches[ip].size > 0 && m.patches[ip].type == PATCH_GENERIC);
            needRef_ = !anyFixed && opt.pRefCell >= 0;
            refLocal_ = needRef_ ? opt.pRefCell : -1;
            refSet = true;
        }
    }
    void applyA(const double* x, dvec& y) const {
        for (int c = 0; c < m.nC; ++c) y[c] = pdiag_[c] * x[c];
        for (int f = 0; f < m.nIF; ++f) { y[m.own[f]] -= pa_[f] * x[m.nei[f]]; y[m.nei[f]] -= pa_[f] * x[m.own[f]]; }
    }
    bool converged(double res) const { return res < opt.pTol || (opt.pRelTol > 0 && res < opt.pRelTol * ctl[10]); }
    void phase(int ph) {
        const int nC = m.nC, nF = m.nF, nB = m.nBF(), nIF = m.nIF;
        const double dt = opt.deltaT;
        const double rhof = opt.rho0, muf = opt.mu, alphaf = opt.mu / opt.Pr;
        const double Hif = alphaf / rhof, nuf = muf / rhof;
        ensureShardInfo();
        if (ph == 0) {
            gU_ = gaussGradV(U); gUValid = true;   // fvc::grad(U) of QHDUEqn.H L76 (ghost rows arrive with the pressure message)
            // updateFields.H L36-73, updateFluxes.H L33-38 (as in step())
            SurfField gradUf = stencil->gradV(U), gradTf = stencil->gradS(T);
            Uf_ = linearInterpolate(m, U); Tf_ = linearInterpolate(m, T);
            BdFrc_ = VolField(m, 3);
            for (int c = 0; c < nC; ++c) for (int k = 0; k < 3; ++k) BdFrc_.in[3 * (size_t)c + k] = (opt.beta * T.in[c]) * opt.g[k];
            for (int b = 0; b < nB; ++b) for (int k = 0; k < 3; ++k) BdFrc_.bf[3 * (size_t)b + k] = (opt.beta * T.bf[b]) * opt.g[k];
            BdFrcf_ = linearInterpolate(m, BdFrc_);
            UgU_ = SurfField(m, 3); taubyrhof_ = SurfField(m, 1); phiTauTReg_ = SurfField(m, 1);
            for (int f = 0; f < nF; ++f) {
                if (!liveFace[f]) continue;
                const double* S = &m.Sf[3 * (size_t)f];
                phiu.v[f] = dot3(S, &Uf_.v[3 * (size_t)f]);
                VdotT(&Uf_.v[3 * (size_t)f], &gradUf.v[9 * (size_t)f], &UgU_.v[3 * (size_t)f]);
                double wo[3];
                for (int k = 0; k < 3; ++k) wo[k] = tauQGDf.v[f] * (UgU_.v[3 * (size_t)f + k] - BdFrcf_.v[3 * (size_t)f + k]);
                phiwo.v[f] = dot3(S, wo);
                taubyrhof_.v[f] = tauQGDf.v[f] / rhof;
                phiTauTReg_.v[f] = tauQGDf.v[f] * phiu.v[f] * dot3(&Uf_.v[3 * (size_t)f], &gradTf.v[3 * (size_t)f]);
            }
            time += dt;
            correctP();
            // the pressure equation's rows [QHDpEqn.H L36-44], assembled like orc_qhd_pressure
            pa_.assign(nF, 0.0); pdiag_.assign(nC, 0.0); prhs_.assign(nC, 0.0);
            pkind_.assign(std::max(nB, 1), 0); liveB_.assign(std::max(nB, 1), 1);
            for (size_t ip = 0; ip < m.patches.size(); ++ip) {
                int k = bc[ip].bcP;
                if (m.patches[ip].type != PATCH_GENERIC) k = BC_NONE;
                const int kk = k == BC_FIXEDVALUE ? 1 : ((k == BC_QGDFLUX || k == BC_QHDFLUX) ? 2 : 0);
                for (int f = m.patches[ip].start; f < m.patches[ip].start + m.patches[ip].size; ++f) {
                    pkind_[f - nIF] = (char)kk;
                    if (!m.patchHasFields((int)ip)) liveB_[f - nIF] = 0;
                }
            }
            for (int f = 0; f < nF; ++f) pa_[f] = taubyrhof_.v[f] * m.magSf[f] * (f < nIF ? m.nonOrthDelta[f] : m.delta[f]);
            for (int f = 0; f < nF; ++f) {
                if (f >= nIF && !liveB_[f - nIF]) continue;
                const double flux = phiu.v[f] - phiwo.v[f];
                prhs_[m.own[f]] -= flux;
                if (f < nIF) { prhs_[m.nei[f]] += flux; pdiag_[m.own[f]] += pa_[f]; pdiag_[m.nei[f]] += pa_[f]; }
                else {
                    const int b = f - nIF;
                    if (pkind_[b] == 1) { pdiag_[m.own[f]] += pa_[f]; prhs_[m.own[f]] += pa_[f] * p.bf[b]; }
                    else if (pkind_[b] == 2) prhs_[m.own[f]] += taubyrhof_.v[f] * m.magSf[f] * p.grad[b];
                }
            }
            if (refLocal_ >= 0) { prhs_[refLocal_] += pdiag_[refLocal_] * p.in[refLocal_]; pdiag_[refLocal_] += pdiag_[refLocal_]; }
            pr_.assign(nC, 0.0); pz_.assign(nC, 0.0); pd_.assign(nC, 0.0); pq_.assign(nC, 0.0); pA1_.assign(nC, 0.0);
            applyA(p.in.data(), pq_);
            for (double& x : ctl) x = 0.0;
            for (int c = 0; c < nC; ++c) {
                pr_[c] = prhs_[c] - pq_[c];
                if (isOwned(c)) { ctl[0] += std::fabs(pr_[c]); ctl[1] += p.in[c]; ctl[2] += 1.0; }
            }
        } else if (ph == 1) {
            dvec ones((size_t)nC, 1.0);
            applyA(ones.data(), pA1_);
            const double xbar = ctl[1] / ctl[2];
            ctl[3] = 0.0;
            for (int c = 0; c < nC; ++c) if (isOwned(c)) ctl[3] += std::fabs(pq_[c] - xbar * pA1_[c]) + std::fabs(prhs_[c] - xbar * pA1_[c]);
        } else if (ph == 2) {
            normF_ = ctl[3] + 1e-20;
            ctl[9] = ctl[0] / normF_; ctl[10] = ctl[9];
            ctl[11] = (ctl[9] < opt.pTol || opt.pMaxIter <= 0) ? 1.0 : 0.0;
            ctl[12] = 0.0; ctl[4] = 0.0;
            if (ctl[11] == 0.0)
                for (int c = 0; c < nC; ++c) if (isOwned(c)) { pz_[c] = pr_[c] / pdiag_[c]; pd_[c] = pz_[c]; ctl[4] += pr_[c] * pz_[c]; }
        } else if (ph == 3) {
            if (ctl[11] != 0.0) return;
            applyA(pd_.data(), pq_);
            ctl[5] = 0.0;
            for (int c = 0; c < nC; ++c) if (isOwned(c)) ctl[5] += pd_[c] * pq_[c];
        } else if (ph == 4) {
            if (ctl[11] != 0.0) return;
            if (!(ctl[5] > 0) || !(ctl[4] > 0)) { ctl[11] = 2.0; return; }
            const double alpha = ctl[4] / ctl[5];
            ctl[13] = alpha; ctl[6] = 0.0; ctl[7] = 0.0;
            for (int c = 0; c < nC; ++c) if (isOwned(c)) {
                p.in[c] += alpha * pd_[c]; pr_[c] -= alpha * pq_[c]; pz_[c] = pr_[c] / pdiag_[c];
                ctl[7] += pr_[c] * pz_[c]; ctl[6] += std::fabs(pr_[c]);
            }
        } else if (ph == 5) {
            if (ctl[11] != 0.0) return;
            ctl[9] = ctl[6] / normF_;
            ctl[12] += 1.0;
            if (converged(ctl[9]) || ctl[12] >= (double)opt.pMaxIter) { ctl[11] = 1.0; return; }
            const double beta = ctl[7] / ctl[4];
            ctl[14] = beta; ctl[4] = ctl[7];
            for (int c = 0; c < nC; ++c) if (isOwned(c)) pd_[c] = pz_[c] + beta * pd_[c];
        } else if (ph == 6) {
            lastPIter = ctl[12]; lastPRes0 = ctl[10]; lastPRes = ctl[9];
            correctP();   // fvMatrix::solve ends in correctBoundaryConditions()
        } else if (ph == 7) {
            for (int f = 0; f < nF; ++f) {   // phi = phiu - phiwo + pEqn.flux()
                double corr = 0;
                if (f < nIF) corr = -pa_[f] * (p.in[m.nei[f]] - p.in[m.own[f]]);
                else {
                    const int b = f - nIF;
                    if (pkind_[b] == 1) corr = -pa_[f] * (p.bf[b] - p.in[m.own[f]]);
                    else if (pkind_[b] == 2) corr = -taubyrhof_.v[f] * m.magSf[f] * p.grad[b];
                }
                phi.v[f] = (f < nIF || liveB_[f - nIF]) ? (phiu.v[f] - phiwo.v[f]) + corr : 0.0;
            }
            // QHDUEqn.H L36-84, QHDTEqn.H L65-91 (as in step(), with fvc::grad(U) kept from the end of the previous step)
            SurfField gradPf = stencil->gradS(p);
            SurfField pf = linearInterpolate(m, p);
            const VolField& gU = gU_;
            VolField gUT(m, 9);
            for (int c = 0; c < nC; ++c) for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) gUT.in[9 * (size_t)c + 3 * i + j] = gU.in[9 * (size_t)c + 3 * j + i];
            for (int b = 0; b < nB; ++b) for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) gUT.bf[9 * (size_t)b + 3 * i + j] = gU.bf[9 * (size_t)b + 3 * j + i];
            SurfField gUTf = linearInterpolate(m, gUT);
            SurfField snU = fvcSnGrad(m, U), snT = fvcSnGrad(m, T);
            const SurfField UfFlux = opt.fluxSchemeU ? upwindInterpolate(m, phi, U) : Uf_, TfFlux = opt.fluxSchemeT ? upwindInterpolate(m, phi, T) : Tf_;
            dvec FU(3 * (size_t)nF, 0.0), FT((size_t)nF, 0.0), Gp(3 * (size_t)nF, 0.0);
            for (int f = 0; f < nF; ++f) {
                if (!liveFace[f]) continue;
                const double* S = &m.Sf[3 * (size_t)f];
                double Wf[3], UW[9], uw[3], ext[3];
                for (int k = 0; k < 3; ++k)
                    Wf[k] = tauQGDf.v[f] * ((UgU_.v[3 * (size_t)f + k] + gradPf.v[3 * (size_t)f + k] / rhof) - BdFrcf_.v[3 * (size_t)f + k]);
                outer(&Uf_.v[3 * (size_t)f], Wf, UW);
                VdotT(S, UW, uw);
                VdotT(S, &gUTf.v[9 * (size_t)f], ext);
                for (int k = 0; k < 3; ++k) {
                    const double phiUf = phi.v[f] * UfFlux.v[3 * (size_t)f + k] - uw[k];
                    const double lap = nuf * snU.v[3 * (size_t)f + k] * m.magSf[f];
                    FU[3 * (size_t)f + k] = (phiUf - lap) - nuf * ext[k];
                    Gp[3 * (size_t)f + k] = S[k] * pf.v[f];
                }
                FT[f] = (phi.v[f] * TfFlux.v[f] - Hif * snT.v[f] * m.magSf[f]) - phiTauTReg_.v[f];
            }
            dvec sumU(3 * (size_t)nC, 0.0), sumT((size_t)nC, 0.0), sumG(3 * (size_t)nC, 0.0);
            for (int f = 0; f < nF; ++f) {
                if (!liveFace[f]) continue;
                const int o = m.own[f];
                for (int k = 0; k < 3; ++k) { sumU[3 * (size_t)o + k] += FU[3 * (size_t)f + k]; sumG[3 * (size_t)o + k] += Gp[3 * (size_t)f + k]; }
                sumT[o] += FT[f];
                if (f < m.nIF) {
                    const int n = m.nei[f];
                    for (int k = 0; k < 3; ++k) { sumU[3 * (size_t)n + k] -= FU[3 * (size_t)f + k]; sumG[3 * (size_t)n + k] -= Gp[3 * (size_t)f + k]; }
                    sumT[n] -= FT[f];
                }
            }
            for (int c = 0; c < nC; ++c) {
                if (!isOwned(c)) continue;   // ghost cells are refreshed by the halo message
                const double rV = 1.0 / m.V[c];
                for (int k = 0; k < 3; ++k)
                    U.in[3 * (size_t)c + k] += dt * ((-(sumU[3 * (size_t)c + k] * rV) - (sumG[3 * (size_t)c + k] * rV) / opt.rho0) + BdFrc_.in[3 * (size_t)c + k]);
                T.in[c] += dt * (-(sumT[c] * rV));
            }
            correctU(); correctT();
            ctl[8] = (needRef_ && refLocal_ >= 0) ? opt.pRefValue - p.in[refLocal_] : 0.0;   // QHDFoam.C L123-130, this rank's share
        } else if (ph == 8) {
            if (needRef_) {
                for (double& x : p.in) x += ctl[8];
                for (int b = 0; b < nB; ++b) p.bf[b] += ctl[8];
            }
            ++steps;
        }
    }
    int haloWidth(int kind, bool face) const { return face ? (kind == 0 ? 4 : (kind == 1 ? 2 : 0)) : (kind == 0 ? 4 : (kind == 1 ? 10 : 1)); }
    int64_t haloCount(int side, int kind, bool recv) const {
        if (side < 0 || (size_t)side >= m.haloGhost.size()) return 0;
        const ivec& cells = recv ? m.haloGhost[side] : m.haloSend[side];
        const ivec& faces = recv ? m.haloGhostBF[side] : m.haloSendBF[side];
        return (int64_t)haloWidth(kind, false) * (int64_t)cells.size() + (int64_t)haloWidth(kind, true) * (int64_t)faces.size();
    }
    void haloMove(int side, int kind, double* buf, bool pack) {
        if (side < 0 || (size_t)side >= m.haloGhost.size()) return;
        const ivec& cells = pack ? m.haloSend[side] : m.haloGhost[side];
        const ivec& faces = pack ? m.haloSendBF[side] : m.haloGhostBF[side];
        if (kind == 1 && !gUValid) { gU_ = gaussGradV(U); gUValid = true; }
        if (kind == 2 && pd_.size() != (size_t)m.nC) pd_.assign(m.nC, 0.0);
        auto mv = [&](double& field, double& slot) { if (pack) slot = field; else field = slot; };
        size_t k = 0;
        for (int c : cells) {
            if (kind == 0) {
                for (int q = 0; q < 3; ++q) mv(U.in[3 * (size_t)c + q], buf[k++]);
                mv(T.in[c], buf[k++]);
            } else if (kind == 1) {
                mv(p.in[c], buf[k++]);
                for (int q = 0; q < 9; ++q) mv(gU_.in[9 * (size_t)c + q], buf[k++]);
            } else mv(pd_[c], buf[k++]);
        }
        if (kind == 2) return;
        for (int b : faces) {
            if (kind == 0) { for (int q = 0; q < 3; ++q) mv(U.bf[3 * (size_t)b + q], buf[k++]); mv(T.bf[b], buf[k++]); }
            else { mv(p.bf[b], buf[k++]); mv(p.grad[b], buf[k++]); }
        }
    }
};

extern "C" {

void* orc_mesh_create(int32_t nPoints, const double* points, int32_t nFaces, const int32_t* faceOffsets,
                      const int32_t* facePoints, int32_t nInternalFaces, const int32_t* owner, const int32_t* neighbour,
                      int32_t nCells, int32_t nPatches, const int32_t* patchStart, const int32_t* patchSize,
                      const int32_t* patchType) {
    MeshHandle* h = new MeshHandle();
    Mesh& m = h->m;
    m.nP = nPoints; m.nF = nFaces; m.nIF = nInternalFaces; m.nC = nCells;
    m.pts.assign(points, points + 3 * (size_t)nPoints);
    m.fOff.assign(faceOffsets, faceOffsets + nFaces + 1);
    m.fPts.assign(facePoints, facePoints + faceOffsets[nFaces]);
    m.own.assign(owner, owner + nFaces);
    m.nei.assign(neighbour, neighbour + nInternalFaces);
    for (int i = 0; i < nPatches; ++i) m.patches.push_back(PatchInfo{patchType[i], patchStart[i], patchSize[i]});
    m.geometry();
    m.addressing();
    m.pointInterpolationWeights();
    return h;
}
void orc_mesh_free(void* m) { delete (MeshHandle*)m; }

int orc_mesh_get(void* mp, const char* name, double* out, int64_t n) {
    const Mesh& m = ((MeshHandle*)mp)->m;
    const dvec* src = nullptr;
    std::string s(name);
    if (s == "Sf") src = &m.Sf; else if (s == "magSf") src = &m.magSf; else if (s == "Cf") src = &m.Cf;
    else if (s == "C") src = &m.C; else if (s == "V") src = &m.V; else if (s == "weights") src = &m.w;
    else if (s == "deltaCoeffs") src = &m.delta; else if (s == "nonOrthDeltaCoeffs") src = &m.nonOrthDelta;
    if (!src) return -5;
    if ((int64_t)src->size() > n) return -1;
    std::copy(src->begin(), src->end(), out);
    return 0;
}
/* geometry handed over by the caller (what an OpenFOAM adapter does with mesh.Sf(), Cf(), C(), V(); counterpart of
 * qgd_mesh_set_geometry): everything derived from it is rebuilt, stencils built so far are dropped */
int orc_mesh_set_geometry(void* mp, const double* Sf, const double* Cf, const double* C, const double* V) {
    MeshHandle* h = (MeshHandle*)mp;
    Mesh& m = h->m;
    m.Sf.assign(Sf, Sf + 3 * (size_t)m.nF); m.Cf.assign(Cf, Cf + 3 * (size_t)m.nF);
    m.C.assign(C, C + 3 * (size_t)m.nC); m.V.assign(V, V + (size_t)m.nC);
    for (int f = 0; f < m.nF; ++f) m.magSf[f] = mag3(&m.Sf[3 * (size_t)f]);
    m.derivedGeometry();
    m.pointInterpolationWeights();
    for (auto& kv : h->cache.byName) delete kv.second;
    h->cache.byName.clear();
    return 0;
}
int orc_mesh_info(void* mp, int64_t info[4]) {
    const Mesh& m = ((MeshHandle*)mp)->m;
    info[0] = m.nGeomD; info[1] = m.geomD[0]; info[2] = m.geomD[1]; info[3] = m.geomD[2];
    return 0;
}
int orc_mesh_set_halo(void* mp, int side, int32_t nGhost, const int32_t* ghost, int32_t nSend, const int32_t* send) {
    Mesh& m = ((MeshHandle*)mp)->m;
    if (side < 0) return -1;
    if ((size_t)side >= m.haloGhost.size()) { m.haloGhost.resize((size_t)side + 1); m.haloSend.resize((size_t)side + 1); }
    m.haloGhost[side].assign(ghost, ghost + nGhost);
    m.haloSend[side].assign(send, send + nSend);
    m.haloFaces();
    return 0;
}

int orc_mesh_set_degenerate_faces(void* mp, int32_t n, const int32_t* faces) {
    MeshHandle* h = (MeshHandle*)mp;
    h->m.userDegenerateFaces.assign(faces, faces + n);
    for (auto& kv : h->cache.byName) delete kv.second;   // stencils built so far are dropped
    h->cache.byName.clear();
    return 0;
}
/* the leastSquares stencil of internal face `face` in the order the reference builds it [extendedFaceStencilFindNeighbours.C:48-84];
 * returns the number of cells (the first `cap` are written), < 0 when the mesh has no leastSquares stencil (3-D) */
int orc_mesh_lsq_stencil(void* mp, int32_t face, int32_t* cells, int32_t cap) {
    MeshHandle* h = (MeshHandle*)mp;
    Stencil* st = nullptr;
    if (h->cache.lookup(h->m, "leastSquares", &st) != 0) return -1;
    const LeastSquares* ls = dynamic_cast<const LeastSquares*>(st);
    if (!ls || face < 0 || face >= h->m.nIF) return -1;
    const ivec& nb = ls->neighbourCells[face];
    for (int i = 0; i < (int)nb.size() && i < cap; ++i) cells[i] = nb[i];
    return (int)nb.size();
}
int orc_mesh_set_halo_face_h(void* mp, int32_t n, const double* h) {
    ((MeshHandle*)mp)->m.haloFaceH.assign(h, h + n);
    return 0;
}

int orc_fvsc(void* mp, const char* scheme, const char* op, const double* cell, const double* bnd, double* out) {
    MeshHandle* h = (MeshHandle*)mp;
    const Mesh& m = h->m;
    Stencil* s = nullptr;
    int rc = h->cache.lookup(m, scheme, &s);
    if (rc) return rc;
    std::string o(op);
    const int nc = (o == "grad_s") ? 1 : (o == "div_t" ? 9 : 3);
    VolField f(m, nc);
    std::copy(cell, cell + (size_t)m.nC * nc, f.in.begin());
    std::copy(bnd, bnd + (size_t)m.nBF() * nc, f.bf.begin());
    SurfField r;
    if (o == "grad_s") r = s->gradS(f); else if (o == "grad_v") r = s->gradV(f);
    else if (o == "div_v") r = s->divV(f); else if (o == "div_t") r = s->divT(f);
    else return -5;
    std::copy(r.v.begin(), r.v.end(), out);
    return 0;
}

// QHDFoam face-flux parts, field at a time:
// [QHDFoam/updateFields.H:36-73] [QHDFoam/updateFluxes.H:33-38] [QHDUEqn.H:36-43] [QHDTEqn.H:65-66]
int orc_qhd_fluxes(void* mp, const char* scheme, const orc_qhd_inputs* in, orc_qhd_outputs* out) {
    MeshHandle* h = (MeshHandle*)mp;
    const Mesh& m = h->m;
    Stencil* st = nullptr;
    int rc = h->cache.lookup(m, scheme, &st);
    if (rc) return rc;
    const int nC = m.nC, nB = m.nBF(), nF = m.nF;
    VolField U(m, 3), T(m, 1), p(m, 1), rho(m, 1), BdFrc(m, 3);
    std::copy(in->U, in->U + 3 * (size_t)nC, U.in.begin());
    std::copy(in->T, in->T + nC, T.in.begin());
    std::copy(in->rho, in->rho + nC, rho.in.begin());
    if (nB) { std::copy(in->Ub, in->Ub + 3 * (size_t)nB, U.bf.begin()); std::copy(in->Tb, in->Tb + nB, T.bf.begin()); std::copy(in->rhob, in->rhob + nB, rho.bf.begin()); }
    if (in->p) { std::copy(in->p, in->p + nC, p.in.begin()); if (nB) std::copy(in->pb, in->pb + nB, p.bf.begin()); }
    // updateFields.H
    SurfField gradUf = st->gradV(U);                      // L36
    SurfField gradTf = st->gradS(T);                      // L40
    SurfField rhof = linearInterpolate(m, rho);           // L57
    SurfField Uf = linearInterpolate(m, U);               // L60
    SurfField Tf = linearInterpolate(m, T);               // L63
    for (int c = 0; c < nC; ++c) for (int k = 0; k < 3; ++k) BdFrc.in[3 * (size_t)c + k] = (in->beta * T.in[c]) * in->g[k];   // L66
    for (int b = 0; b < nB; ++b) for (int k = 0; k < 3; ++k) BdFrc.bf[3 * (size_t)b + k] = (in->beta * T.bf[b]) * in->g[k];
    SurfField BdFrcf = linearInterpolate(m, BdFrc);       // L67
    std::vector<char> live(nF, 1);
    for (size_t ip = 0; ip < m.patches.size(); ++ip)
        if (!m.patchHasFields((int)ip)) for (int f = m.patches[ip].start; f < m.patches[ip].start + m.patches[ip].size; ++f) live[f] = 0;
    // updateFluxes.H
    SurfField phiu(m, 1), phiwo(m, 1), taubyrhof(m, 1), UgU(m, 3);
    for (int f = 0; f < nF; ++f) {
        if (!live[f]) continue;
        const double* S = &m.Sf[3 * (size_t)f];
        phiu.v[f] = dot3(S, &Uf.v[3 * (size_t)f]);                                           // L33
        VdotT(&Uf.v[3 * (size_t)f], &gradUf.v[9 * (size_t)f], &UgU.v[3 * (size_t)f]);        // Uf & gradUf
        double wo[3];
        for (int k = 0; k < 3; ++k) wo[k] = in->tauQGDf[f] * (UgU.v[3 * (size_t)f + k] - BdFrcf.v[3 * (size_t)f + k]);
        phiwo.v[f] = dot3(S, wo);                                                            // L35
        taubyrhof.v[f] = in->tauQGDf[f] / rhof.v[f];                                         // L38
    }
    auto give = [&](double* dst, const SurfField& s) { if (dst) std::copy(s.v.begin(), s.v.end(), dst); };
    give(out->gradUf, gradUf); give(out->gradTf, gradTf); give(out->phiu, phiu); give(out->phiwo, phiwo); give(out->taubyrhof, taubyrhof);
    // QHDTEqn.H L66
    if (out->phiTauTReg) for (int f = 0; f < nF; ++f)
        out->phiTauTReg[f] = live[f] ? in->tauQGDf[f] * phiu.v[f] * dot3(&Uf.v[3 * (size_t)f], &gradTf.v[3 * (size_t)f]) : 0.0;
    if (out->phiTf) for (int f = 0; f < nF; ++f) out->phiTf[f] = live[f] ? in->phi[f] * Tf.v[f] : 0.0;   // L65 qgdFlux -> flux*psif
    if (in->p) {
        SurfField gradPf = st->gradS(p);                                                     // QHDUEqn.H L36
        SurfField Wf(m, 3);
        for (int f = 0; f < nF; ++f) {
            if (!live[f]) continue;
            for (int k = 0; k < 3; ++k)
                Wf.v[3 * (size_t)f + k] = in->tauQGDf[f] * ((UgU.v[3 * (size_t)f + k] + gradPf.v[3 * (size_t)f + k] / rhof.v[f]) - BdFrcf.v[3 * (size_t)f + k]);  // L37
        }
        give(out->gradPf, gradPf); give(out->Wf, Wf);
        if (out->phiUf) for (int f = 0; f < nF; ++f) {
            if (!live[f]) { for (int k = 0; k < 3; ++k) out->phiUf[3 * (size_t)f + k] = 0.0; continue; }
            double UW[9], uw[3];
            outer(&Uf.v[3 * (size_t)f], &Wf.v[3 * (size_t)f], UW);                           // Uf * Wf
            VdotT(&m.Sf[3 * (size_t)f], UW, uw);                                             // L39
            for (int k = 0; k < 3; ++k) out->phiUf[3 * (size_t)f + k] = in->phi[f] * Uf.v[3 * (size_t)f + k] - uw[k];   // L41-43
        }
    }
    return 0;
}

// cyclic / wedge patches with faces: the cases restate no coupled / wedge patch field (NULL, like qgd_case_create's refusal)
static bool caseServesMesh(const Mesh& m) {
    for (const PatchInfo& p : m.patches) if ((p.type == PATCH_CYCLIC || p.type == PATCH_WEDGE) && p.size > 0) return false;
    return true;
}
void* orc_case_create(void* mesh, const orc_case_options* opt) {
    if (!caseServesMesh(((MeshHandle*)mesh)->m)) return nullptr;
    Case* c = new Case((MeshHandle*)mesh, *opt);
    auto wordOfId = [](int id) { return std::string(id == FVSC_REDUCED ? "reduced" : (id == FVSC_LEASTSQUARES ? "leastSquares" : "GaussVolPoint")); };
    c->stencilWord = wordOfId(opt->stencil);
    for (int t = 0; t < 4; ++t) if (opt->termStencil[t] != 0) c->termWord[t] = wordOfId(opt->termStencil[t] - 1);
    const Mesh& m = c->m;
    if (m.sharded()) {
        c->ghostFlag.assign(m.nC, 0);
        for (const ivec& gl : m.haloGhost) for (int g : gl) c->ghostFlag[g] = 1;
    }
    return c;
}
void orc_case_free(void* c) { delete (Case*)c; }
int orc_case_set_bc(void* cp, int32_t patch, int32_t bcU, const double* valueU, int32_t bcT, double valueT, int32_t bcP, double valueP) {
    Case* c = (Case*)cp;
    if (patch < 0 || patch >= (int)c->bc.size()) return -1;
    PatchBC& b = c->bc[patch];
    constraintKinds(c->m.patches[patch].type, bcU, bcT, bcP);
    b.bcU = bcU; b.bcT = bcT; b.bcP = bcP; b.vT = valueT; b.vP = valueP;
    if (valueU) for (int k = 0; k < 3; ++k) b.vU[k] = valueU[k];
    return 0;
}
int orc_case_set_qgd_coeffs(void* cp, const double* a, const double* ab, const double* sc, const double* scb) {
    Case* c = (Case*)cp;
    const size_t nC = (size_t)c->m.nC, nB = (size_t)c->m.nBF();
    c->aQGDin.clear(); c->aQGDbf.clear(); c->ScQGDin.clear(); c->ScQGDbf.clear();
    if (a) { c->aQGDin.assign(a, a + nC); c->aQGDbf.assign(nB, 0.0); if (nB) std::copy(ab, ab + nB, c->aQGDbf.begin()); }
    if (sc) { c->ScQGDin.assign(sc, sc + nC); c->ScQGDbf.assign(nB, 0.0); if (nB) std::copy(scb, scb + nB, c->ScQGDbf.begin()); }
    return 0;
}
int orc_case_set_fields(void* cp, const double* U, const double* T, const double* p) { return ((Case*)cp)->setFields(U, T, p); }
int orc_case_update_fluxes(void* cp) { Case* c = (Case*)cp; c->updateFields(); c->updateFluxes(); return 0; }
int orc_case_step(void* cp, int32_t n) {
    Case* c = (Case*)cp;
    for (int i = 0; i < n; ++i) { c->stepPhase0(); c->stepPhase1(); c->stepPhase2(); }
    return 0;
}
int orc_case_step_phase(void* cp, int phase) {
    Case* c = (Case*)cp;
    if (phase >= 20) {
        if (phase == 20) c->phaseNew_ = false;
        c->implicitPhase(phase);
        if (phase == 29) c->phaseNew_ = true;
        return 0;
    }
    if (phase == 0) c->stepPhase0(); else if (phase == 1) c->stepPhase1(); else if (phase == 5) c->stepPhase5(); else if (phase == 6) c->stepPhase6();
    else c->stepPhase2();
    return 0;
}
/* nSteps with the fused flux assembly (bench.py cpu_baseline "fused"); 1 = this case is outside the fused path's scope */
int orc_case_step_fused(void* cp, int32_t nSteps) { return ((Case*)cp)->stepFused(nSteps) ? 0 : 1; }
int orc_case_mid_exchange_needed(void* cp) { return ((Case*)cp)->midNeeded() ? 1 : 0; }
int orc_case_mid_halo_count(void* cp, int side, int64_t* send, int64_t* recv) { ((Case*)cp)->midHaloCount(side, send, recv); return 0; }
int orc_case_mid_halo_pack(void* cp, int side, double* buf) { ((Case*)cp)->midHaloMove(side, buf, true); return 0; }
int orc_case_mid_halo_unpack(void* cp, int side, const double* buf) { ((Case*)cp)->midHaloMove(side, const_cast<double*>(buf), false); return 0; }
int orc_case_implicit_control(void* cp, double* buf68, int set) {
    Case* c = (Case*)cp;
    for (int k = 0; k < 68; ++k) { if (set) c->ictl[k] = buf68[k]; else buf68[k] = c->ictl[k]; }
    return 0;
}
int orc_case_implicit_halo_count(void* cp, int side, int kind, int64_t* send, int64_t* recv) {
    Case* c = (Case*)cp;
    *send = *recv = 0;
    if (side < 0 || (size_t)side >= c->m.haloGhost.size()) return 0;
    const int w = kind >= 3 ? 3 : c->implHaloWidth(kind);
    *send = (int64_t)w * (int64_t)c->m.haloSend[side].size(); *recv = (int64_t)w * (int64_t)c->m.haloGhost[side].size();
    return 0;
}
int orc_case_implicit_halo_pack(void* cp, int side, int kind, double* buf) { ((Case*)cp)->implHaloMove(side, kind, buf, true); return 0; }
int orc_case_implicit_halo_unpack(void* cp, int side, int kind, const double* buf) { ((Case*)cp)->implHaloMove(side, kind, const_cast<double*>(buf), false); return 0; }
// [0] = max Cof, [1] = -min tauQGDf of this shard: MAX-reduce over the ranks between phase 0 and phase 1
int orc_case_reduction(void* cp, double* buf, int set) {
    Case* c = (Case*)cp;
    if (set) { c->redCo = buf[0]; c->redMinTau = -buf[1]; } else { buf[0] = c->redCo; buf[1] = -c->redMinTau; }
    return 0;
}

int orc_case_get_field(void* cp, const char* name, double* out, int64_t n) {
    Case* c = (Case*)cp;
    std::string s(name);
    bool bnd = false;
    const std::string suffix = ".boundary";
    if (s.size() > suffix.size() && s.compare(s.size() - suffix.size(), suffix.size(), suffix) == 0) { bnd = true; s = s.substr(0, s.size() - suffix.size()); }
    std::map<std::string, const VolField*> vf = {
        {"rho", &c->rho}, {"U", &c->U}, {"p", &c->p}, {"e", &c->e}, {"T", &c->T}, {"rhoU", &c->rhoU}, {"rhoE", &c->rhoE},
        {"c", &c->c}, {"psi", &c->psi}, {"mu", &c->mu}, {"alphau", &c->alpha}, {"tauQGD", &c->tauQGD}, {"muQGD", &c->muQGD},
        {"alphauQGD", &c->alphauQGD}, {"hQGD", &c->hQGD}, {"H", &c->H}, {"gamma", &c->gamma}};
    std::map<std::string, const SurfField*> sf = {
        {"phiJm", &c->phiJm}, {"phiJmU", &c->phiJmU}, {"phiP", &c->phiP}, {"phiPi", &c->phiPi}, {"phiJmH", &c->phiJmH},
        {"phiQ", &c->phiQ}, {"phiPiU", &c->phiPiU}, {"phiwStar", &c->phiw}, {"phi", &c->phi}, {"tauQGDf", &c->tauQGDf},
        {"hQGDf", &c->hQGDf}, {"gradUf", &c->gradUf}, {"gradef", &c->gradef}, {"gradRhof", &c->gradRhof}, {"gradPf", &c->gradPf},
        {"rhof", &c->rhof}, {"Uf", &c->Uf}, {"pf", &c->pf}, {"Hf", &c->Hf}, {"muf", &c->muf}, {"alphauf", &c->alphauf}, {"cf", &c->cf},
        {"Pif", &c->Pif}, {"qf", &c->qf}, {"jm", &c->jm},
        {"tauMC", &c->tauMC}, {"phiTauMC", &c->phiTauMC}, {"phiSigmaDotU", &c->phiSigmaDotU}};   // implicitDiffusion branch
    const dvec* src = nullptr;
    auto iv = vf.find(s);
    if (iv != vf.end()) src = bnd ? &iv->second->bf : &iv->second->in;
    else { auto is = sf.find(s); if (is != sf.end() && !bnd) src = &is->second->v; }
    if (!src) return -5;
    if ((int64_t)src->size() > n) return -1;
    std::copy(src->begin(), src->end(), out);
    return 0;
}
int orc_case_info(void* cp, double info[6]) {
    Case* c = (Case*)cp;
    double mr = GREAT, me = GREAT;
    for (int ci = 0; ci < c->m.nC; ++ci) {
        if (!c->ghostFlag.empty() && c->ghostFlag[ci]) continue;
        mr = std::min(mr, c->rho.in[ci]); me = std::min(me, c->e.in[ci]);
    }
    info[0] = c->time; info[1] = c->deltaT; info[2] = c->CoNum; info[3] = mr; info[4] = me; info[5] = (double)c->stepCount;
    return 0;
}

// Species block of reactingLagrangianQGDFoam/updateFluxes.H L117-132 for one species:
//   gradYf = fvsc::grad(Y);  phiJmY = qgdFlux(phiJm, Y, Yf) [= phiJm*Yf, QGDInterpolate.H L104];
//   dydtflux = -phi*tauQGDf*(Uf & gradYf);  phiJmY += dydtflux;  diffusiveFlux = dydtflux
int orc_species_flux(void* mp, const char* scheme, const double* Yc, const double* Yb, const double* Uc, const double* Ub,
                     const double* phiJm, const double* phi, const double* tauQGDf, double* phiJmY, double* diffusiveFlux,
                     double* gradYfOut) {
    MeshHandle* h = (MeshHandle*)mp;
    const Mesh& m = h->m;
    Stencil* st = nullptr;
    int rc = h->cache.lookup(m, scheme, &st);
    if (rc) return rc;
    const int nC = m.nC, nB = m.nBF(), nF = m.nF;
    VolField Y(m, 1), U(m, 3);
    std::copy(Yc, Yc + nC, Y.in.begin());
    std::copy(Uc, Uc + 3 * (size_t)nC, U.in.begin());
    if (nB) { std::copy(Yb, Yb + nB, Y.bf.begin()); std::copy(Ub, Ub + 3 * (size_t)nB, U.bf.begin()); }
    SurfField gradYf = st->gradS(Y);
    SurfField Yf = linearInterpolate(m, Y);
    SurfField Uf = linearInterpolate(m, U);
    std::vector<char> live(nF, 1);
    for (size_t ip = 0; ip < m.patches.size(); ++ip)
        if (!m.patchHasFields((int)ip)) for (int f = m.patches[ip].start; f < m.patches[ip].start + m.patches[ip].size; ++f) live[f] = 0;
    for (int f = 0; f < nF; ++f) {
        if (!live[f]) { phiJmY[f] = 0.0; diffusiveFlux[f] = 0.0; continue; }
        const double dydt = (-phi[f]) * tauQGDf[f] * dot3(&Uf.v[3 * (size_t)f], &gradYf.v[3 * (size_t)f]);
        phiJmY[f] = phiJm[f] * Yf.v[f] + dydt;
        diffusiveFlux[f] = dydt;
    }
    if (gradYfOut) std::copy(gradYf.v.begin(), gradYf.v.end(), gradYfOut);
    return 0;
}

// QGDYEqn.H L44-45, L67-86 for ONE species, explicit branch, the sources (combustion->R, parcels.SYi) handed in as an explicit field:
//   solve(fvm::ddt(rho,Yi) + fvc::div(phiJmYi) - fvc::laplacian(muf/Sc_i, Yi) == Su);
//   diffusiveFlux_i += (muf/Sc_i) * fvc::snGrad(Yi.oldTime()) * mesh.magSf();   Yi.max(0.0);
// L0: Euler ddt, fvc::div = surfaceIntegrate, Gauss laplacian with the uncorrected snGrad (nonOrthDeltaCoeffs inside, patch snGrad =
// deltaCoeffs (patch value - cell value) on the patches).  The inert species (L65/L83, L90-91) is the caller's two axpys.
int orc_species_step(void* mp, const double* Yc, const double* Yb, const double* rhoOld, const double* rho, const double* phiJmY,
                     const double* muf, double Sc, double deltaT, const double* Su, double* diffusiveFlux, double* Ynew) {
    MeshHandle* h = (MeshHandle*)mp;
    const Mesh& m = h->m;
    const int nC = m.nC, nF = m.nF;
    std::vector<char> live(nF, 1);
    for (size_t ip = 0; ip < m.patches.size(); ++ip)
        if (!m.patchHasFields((int)ip) || m.patches[ip].type == PATCH_HALO)
            for (int f = m.patches[ip].start; f < m.patches[ip].start + m.patches[ip].size; ++f) live[f] = 0;
    dvec lapFlux((size_t)nF, 0.0);   // (muf/Sc) snGrad(Yi) |Sf|
    for (int f = 0; f < nF; ++f) {
        if (!live[f]) continue;
        const double sn = f < m.nIF ? m.nonOrthDelta[f] * (Yc[m.nei[f]] - Yc[m.own[f]]) : m.delta[f] * (Yb[f - m.nIF] - Yc[m.own[f]]);
        lapFlux[f] = (muf[f] / Sc) * sn * m.magSf[f];
    }
    auto div = [&](auto flux) {
        dvec d((size_t)nC, 0.0);
        for (int f = 0; f < m.nIF; ++f) { const double x = flux(f); d[m.own[f]] += x; d[m.nei[f]] -= x; }
        for (int f = m.nIF; f < nF; ++f) if (live[f]) d[m.own[f]] += flux(f);
        for (int ci = 0; ci < nC; ++ci) d[ci] /= m.V[ci];
        return d;
    };
    const dvec d1 = div([&](int f) { return live[f] ? phiJmY[f] : 0.0; }), d2 = div([&](int f) { return lapFlux[f]; });
    const double rDeltaT = 1.0 / deltaT;
    for (int ci = 0; ci < nC; ++ci) {
        const double diag = rDeltaT * rho[ci] * m.V[ci];
        double src = rDeltaT * rhoOld[ci] * Yc[ci] * m.V[ci];
        src -= m.V[ci] * d1[ci];
        src += m.V[ci] * d2[ci];
        if (Su) src += m.V[ci] * Su[ci];
        Ynew[ci] = std::max(src / diag, 0.0);
    }
    for (int f = 0; f < nF; ++f) if (live[f]) diffusiveFlux[f] += lapFlux[f];
    return 0;
}

// QGDYEqn.H L47-66, one species, implicitDiffusion branch: fvm::ddt(rho,Yi) + fvc::div(phiJmYi) - fvm::laplacian(muf/Sc, Yi) == Su;
// diffusiveFlux += YEqn.flux() (L0 fvMatrix::flux: upper psi_N - lower psi_O with upper = lower = -a for "- fvm::laplacian", patch
// faces internalCoeffs psi_P - boundaryCoeffs); Yi.max(0).  fixedFace[b] != 0: the face belongs to a fixedValue patch of Yi.
int orc_species_step_implicit(void* mp, const double* Yc, const double* Yb, const uint8_t* fixedFace, const double* rhoOld, const double* rho,
                              const double* phiJmY, const double* muf, double Sc, double deltaT, const double* Su, double tol, int32_t maxIter,
                              double* diffusiveFlux, double* Ynew, double info[3]) {
    MeshHandle* h = (MeshHandle*)mp;
    const Mesh& m = h->m;
    const int nC = m.nC, nF = m.nF;
    std::vector<char> live(nF, 1);
    for (size_t ip = 0; ip < m.patches.size(); ++ip)
        if (!m.patchHasFields((int)ip) || m.patches[ip].type == PATCH_HALO)
            for (int f = m.patches[ip].start; f < m.patches[ip].start + m.patches[ip].size; ++f) live[f] = 0;
    dvec a((size_t)nF, 0.0), diag((size_t)nC), rhs((size_t)nC), d1((size_t)nC, 0.0);
    for (int f = 0; f < nF; ++f) if (live[f]) a[f] = (muf[f] / Sc) * m.magSf[f] * (f < m.nIF ? m.nonOrthDelta[f] : m.delta[f]);
    for (int f = 0; f < m.nIF; ++f) { d1[m.own[f]] += phiJmY[f]; d1[m.nei[f]] -= phiJmY[f]; }
    for (int f = m.nIF; f < nF; ++f) if (live[f]) d1[m.own[f]] += phiJmY[f];
    const double rDeltaT = 1.0 / deltaT;
    for (int ci = 0; ci < nC; ++ci) {
        diag[ci] = rDeltaT * rho[ci] * m.V[ci];
        double src = rDeltaT * rhoOld[ci] * Yc[ci] * m.V[ci];
        src -= m.V[ci] * (d1[ci] / m.V[ci]);
        if (Su) src += m.V[ci] * Su[ci];
        rhs[ci] = src;
    }
    for (int f = 0; f < m.nIF; ++f) { diag[m.own[f]] += a[f]; diag[m.nei[f]] += a[f]; }
    for (int f = m.nIF; f < nF; ++f)
        if (live[f] && fixedFace && fixedFace[f - m.nIF]) { diag[m.own[f]] += a[f]; rhs[m.own[f]] += a[f] * Yb[f - m.nIF]; }
    dvec ai(a.begin(), a.end());
    for (int f = m.nIF; f < nF; ++f) ai[f] = 0.0;
    dvec x(Yc, Yc + nC);
    // initial / final normalised residuals as OpenFOAM prints them
    auto resid = [&](const dvec& v) {
        dvec q((size_t)nC), A1((size_t)nC);
        for (int c = 0; c < nC; ++c) { q[c] = diag[c] * v[c]; A1[c] = diag[c]; }
        for (int f = 0; f < m.nIF; ++f) { q[m.own[f]] -= a[f] * v[m.nei[f]]; q[m.nei[f]] -= a[f] * v[m.own[f]]; A1[m.own[f]] -= a[f]; A1[m.nei[f]] -= a[f]; }
        double xbar = 0; for (int c = 0; c < nC; ++c) xbar += v[c];
        xbar /= nC;
        double nf = 1e-20, sr = 0;
        for (int c = 0; c < nC; ++c) { nf += std::fabs(q[c] - xbar * A1[c]) + std::fabs(rhs[c] - xbar * A1[c]); sr += std::fabs(rhs[c] - q[c]); }
        return sr / nf;
    };
    info[1] = resid(x);
    info[0] = solveDiagLaplacian(m, ai, diag, rhs, x.data(), tol, maxIter);
    info[2] = resid(x);
    for (int f = 0; f < nF; ++f) {
        if (!live[f]) continue;
        double fl = 0.0;
        if (f < m.nIF) fl = -(a[f] * (x[m.nei[f]] - x[m.own[f]]));
        else if (fixedFace && fixedFace[f - m.nIF]) fl = -(a[f] * (Yb[f - m.nIF] - x[m.own[f]]));
        diffusiveFlux[f] += fl;
    }
    for (int ci = 0; ci < nC; ++ci) Ynew[ci] = std::max(x[ci], 0.0);
    return 0;
}

// QHDpEqn.H L35-47: fvc::div(phiu) - fvc::div(phiwo) - fvm::laplacian(taubyrhof, p) == 0 with setReference and
// phi = phiu - phiwo + pEqn.flux().  L0 assumptions: Gauss laplacian, uncorrected snGrad (nonOrthDeltaCoeffs inside,
// deltaCoeffs on patches); fixedValue / fixedGradient / zeroGradient patch coefficients; the linear solver is a plain
// diagonal-preconditioned CG run to `tolerance` on OpenFOAM's normalised residual (sequential sums).
int orc_qhd_pressure(void* mp, const double* phiu, const double* phiwo, const double* taubyrhof, const int32_t* patchKind,
                     const double* pb, const double* gradb, double tolerance, double relTol, int32_t maxIter, int32_t pRefCell,
                     double pRefValue, double* p, double* phi, double info[3]) {
    MeshHandle* h = (MeshHandle*)mp;
    const Mesh& m = h->m;
    const int nC = m.nC, nF = m.nF, nIF = m.nIF;
    dvec a((size_t)nF), diag((size_t)nC, 0.0), rhs((size_t)nC, 0.0);
    std::vector<int> kind((size_t)std::max(m.nBF(), 1), 0);
    bool anyFixed = false;
    for (size_t ip = 0; ip < m.patches.size(); ++ip) {
        int k = patchKind[ip];
        if (m.patches[ip].type != PATCH_GENERIC) k = BC_NONE;
        const int kk = k == BC_FIXEDVALUE ? 1 : ((k == BC_QGDFLUX || k == BC_QHDFLUX) ? 2 : 0);
        if (kk == 1 && m.patches[ip].size > 0) anyFixed = true;
        for (int f = m.patches[ip].start; f < m.patches[ip].start + m.patches[ip].size; ++f) kind[f - nIF] = kk;
    }
    for (int f = 0; f < nF; ++f) a[f] = taubyrhof[f] * m.magSf[f] * (f < nIF ? m.nonOrthDelta[f] : m.delta[f]);
    std::vector<char> liveB((size_t)std::max(m.nBF(), 1), 1);
    for (size_t ip = 0; ip < m.patches.size(); ++ip)
        if (!m.patchHasFields((int)ip))
            for (int f = m.patches[ip].start; f < m.patches[ip].start + m.patches[ip].size; ++f) liveB[f - nIF] = 0;
    auto live = [&](int f) { return f < nIF || liveB[f - nIF]; };
    for (int f = 0; f < nF; ++f) {  // ascending face order per cell, as the device gathers
        if (!live(f)) continue;
        const double flux = phiu[f] - phiwo[f];
        rhs[m.own[f]] -= flux;
        if (f < nIF) { rhs[m.nei[f]] += flux; diag[m.own[f]] += a[f]; diag[m.nei[f]] += a[f]; }
        else {
            const int b = f - nIF;
            if (kind[b] == 1) { diag[m.own[f]] += a[f]; rhs[m.own[f]] += a[f] * pb[b]; }
            else if (kind[b] == 2) rhs[m.own[f]] += taubyrhof[f] * m.magSf[f] * gradb[b];
        }
    }
    if (!anyFixed && pRefCell >= 0) { rhs[pRefCell] += diag[pRefCell] * pRefValue; diag[pRefCell] += diag[pRefCell]; }
    auto apply = [&](const double* x, dvec& y) {
        for (int c = 0; c < nC; ++c) y[c] = diag[c] * x[c];
        for (int f = 0; f < nIF; ++f) { y[m.own[f]] -= a[f] * x[m.nei[f]]; y[m.nei[f]] -= a[f] * x[m.own[f]]; }
    };
    dvec r((size_t)nC), z((size_t)nC), d((size_t)nC), q((size_t)nC), A1((size_t)nC), ones((size_t)nC, 1.0);
    apply(ones.data(), A1);
    apply(p, q);
    double xbar = 0;
    for (int c = 0; c < nC; ++c) xbar += p[c];
    xbar /= nC;
    double normFactor = 1e-20, sumAbs = 0, rz = 0;
    for (int c = 0; c < nC; ++c) {
        normFactor += std::fabs(q[c] - xbar * A1[c]) + std::fabs(rhs[c] - xbar * A1[c]);
        r[c] = rhs[c] - q[c]; z[c] = r[c] / diag[c]; d[c] = z[c];
        sumAbs += std::fabs(r[c]); rz += r[c] * z[c];
    }
    double res = sumAbs / normFactor;
    const double res0 = res;
    int it = 0;
    while (it < maxIter && !(res < tolerance || (relTol > 0 && res < relTol * res0))) {
        apply(d.data(), q);
        double dq = 0;
        for (int c = 0; c < nC; ++c) dq += d[c] * q[c];
        if (!(dq > 0) || !(rz > 0)) break;
        const double alpha = rz / dq;
        double rzNew = 0; sumAbs = 0;
        for (int c = 0; c < nC; ++c) {
            p[c] += alpha * d[c]; r[c] -= alpha * q[c]; z[c] = r[c] / diag[c];
            rzNew += r[c] * z[c]; sumAbs += std::fabs(r[c]);
        }
        res = sumAbs / normFactor;
        const double beta = rzNew / rz;
        for (int c = 0; c < nC; ++c) d[c] = z[c] + beta * d[c];
        rz = rzNew;
        ++it;
    }
    for (int f = 0; f < nF; ++f) {
        double corr = 0;
        if (f < nIF) corr = -a[f] * (p[m.nei[f]] - p[m.own[f]]);
        else {
            const int b = f - nIF;
            if (kind[b] == 1) corr = -a[f] * (pb[b] - p[m.own[f]]);
            else if (kind[b] == 2) corr = -taubyrhof[f] * m.magSf[f] * gradb[b];
        }
        phi[f] = live(f) ? (phiu[f] - phiwo[f]) + corr : 0.0;
    }
    if (info) { info[0] = it; info[1] = res0; info[2] = res; }
    return 0;
}

// STREAM triad a = b + s*c (24 bytes per element by the STREAM convention): bench.py times it on the same host cores as
// the oracle ranks to bound what ANY fused CPU implementation of the step could reach there (bytes per cell-step / bandwidth)
void* orc_qhd_case_create(void* mesh, const orc_qhd_options* opt) {
    if (!caseServesMesh(((MeshHandle*)mesh)->m)) return nullptr;
    QhdCase* c = new QhdCase((MeshHandle*)mesh, *opt);
    c->word = opt->stencil == FVSC_REDUCED ? "reduced" : (opt->stencil == FVSC_LEASTSQUARES ? "leastSquares" : "GaussVolPoint");
    return c;
}
void orc_qhd_case_free(void* c) { delete (QhdCase*)c; }
int orc_qhd_case_set_bc(void* cp, int32_t patch, int32_t bcU, const double* vU, int32_t bcT, double vT, int32_t bcP, double vP) {
    QhdCase* c = (QhdCase*)cp;
    if (patch < 0 || patch >= (int32_t)c->bc.size()) return -1;
    PatchBC& b = c->bc[patch];
    constraintKinds(c->m.patches[patch].type, bcU, bcT, bcP);
    b.bcU = bcU; b.bcT = bcT; b.bcP = bcP; b.vT = vT; b.vP = vP;
    if (vU) for (int k = 0; k < 3; ++k) b.vU[k] = vU[k];
    return 0;
}
int orc_qhd_case_set_fields(void* cp, const double* U, const double* T, const double* p) { return ((QhdCase*)cp)->setFields(U, T, p); }
int orc_qhd_case_step(void* cp, int32_t n) { for (int i = 0; i < n; ++i) ((QhdCase*)cp)->step(); return 0; }
int orc_qhd_case_step_phase(void* cp, int phase) {
    if (phase < 0 || phase > 8 || ((QhdCase*)cp)->opt.implicitDiffusion) return -1;   // the phase form restates the explicit branch only
    ((QhdCase*)cp)->phase(phase);
    return 0;
}
int orc_qhd_case_control(void* cp, double* buf16, int set) {
    QhdCase* c = (QhdCase*)cp;
    for (int k = 0; k < 16; ++k) { if (set) c->ctl[k] = buf16[k]; else buf16[k] = c->ctl[k]; }
    return 0;
}
/* whether p needs a reference level is a property of the whole mesh, and pRefCell may belong to another shard: the caller
 * tells a shard (needRef, local label of the reference cell or -1) */
int orc_qhd_case_set_reference(void* cp, int needRef, int localRefCell) {
    QhdCase* c = (QhdCase*)cp;
    c->needRef_ = needRef != 0; c->refLocal_ = needRef ? localRefCell : -1; c->refSet = true;
    return 0;
}
int orc_qhd_case_halo_count(void* cp, int side, int kind, int64_t* send, int64_t* recv) {
    QhdCase* c = (QhdCase*)cp;
    *send = c->haloCount(side, kind, false); *recv = c->haloCount(side, kind, true);
    return 0;
}
int orc_qhd_case_halo_pack(void* cp, int side, int kind, double* buf) { ((QhdCase*)cp)->haloMove(side, kind, buf, true); return 0; }
int orc_qhd_case_halo_unpack(void* cp, int side, int kind, const double* buf) { ((QhdCase*)cp)->haloMove(side, kind, const_cast<double*>(buf), false); return 0; }
int orc_qhd_case_get_field(void* cp, const char* name, double* out, int64_t n) {
    QhdCase* c = (QhdCase*)cp;
    const std::string s(name);
    const dvec* src = nullptr;
    if (s == "U") src = &c->U.in; else if (s == "T") src = &c->T.in; else if (s == "p") src = &c->p.in;
    else if (s == "U.boundary") src = &c->U.bf; else if (s == "T.boundary") src = &c->T.bf; else if (s == "p.boundary") src = &c->p.bf;
    else if (s == "phi") src = &c->phi.v; else if (s == "phiu") src = &c->phiu.v; else if (s == "phiwo") src = &c->phiwo.v;
    else if (s == "tauQGDf") src = &c->tauQGDf.v;
    if (!src) return -5;
    if ((int64_t)src->size() > n) return -1;
    std::copy(src->begin(), src->end(), out);
    return 0;
}
int orc_qhd_case_info(void* cp, double info[6]) {
    QhdCase* c = (QhdCase*)cp;
    info[0] = c->time; info[1] = c->opt.deltaT; info[2] = c->lastPIter; info[3] = c->lastPRes0; info[4] = c->lastPRes; info[5] = (double)c->steps;
    return 0;
}

void orc_stream_triad(double* a, const double* b, const double* c, double s, int64_t n, int32_t reps) {
    for (int32_t r = 0; r < reps; ++r) {
        for (int64_t i = 0; i < n; ++i) a[i] = b[i] + s * c[i];
        s += 1e-9 * a[(size_t)(n / 2)];  // keep the repetitions dependent
    }
}

int orc_case_halo_count(void* cp, int side, int64_t* count) { ((Case*)cp)->haloCount(side, count, false); return 0; }
int orc_case_halo_recv_count(void* cp, int side, int64_t* count) { ((Case*)cp)->haloCount(side, count, true); return 0; }
int orc_case_halo_pack(void* cp, int side, double* sendBuf) { ((Case*)cp)->packOrUnpack(side, sendBuf, true); return 0; }
int orc_case_halo_unpack(void* cp, int side, const double* recvBuf) { ((Case*)cp)->packOrUnpack(side, const_cast<double*>(recvBuf), false); return 0; }

}  // extern "C"
