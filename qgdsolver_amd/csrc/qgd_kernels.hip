// qgd_kernels.hip -- hand-written HIP kernels (gfx950 / CDNA4, wave64) of the
// QGDFoam face-flux path.  No MFMA: the path has no dense contraction; every
// kernel is an HBM/L2-bound indirect-addressing loop (DESIGN.md "Kernels").
//
//   P   pointInterpKernel      cell -> vertex inverse-distance interpolation
//   PB  boundaryPointKernel    patch points: average of boundary-face values
//   F   faceFlux{Gvp3,Gvp2,Lsq,Reduced}Kernel   fused: 13 face interpolations + 4 fvsc
//                              gradients + all QGD fluxes, one thread per internal face,
//                              one loads-first kernel per stencil
//   FB  boundaryFaceFluxKernel same on boundary faces (mirror-point stencil)
//   C   cellUpdateKernel       deterministic gather of face fluxes + explicit
//                              Euler update + thermo + QGD coefficients
//   B   boundaryUpdateKernel   boundary-condition refresh
//   PFC fusedFaceCellKernel    P + F + C of a block of <= 128 cells in one workgroup (QGD_FUSED, the default of fixed-deltaT explicit
//                              3-D GaussVolPoint cases): vertex values, internal faces and cell update out of LDS
//
// Reference restated (listing lines under /root/reference/docs/html/):
//   interpolations   QGDFoam_2updateFields_8H_source.html L45-80
//   flux algebra     QGDFoam_2updateFluxes_8H_source.html L41-139 (explicit branch)
//   GaussVolPoint    GaussVolPointBase3D_8C_source.html L488-539 (dfdxif/dfdxbf),
//                    GaussVolPointBase2D_8C_source.html L315-360
//   leastSquares     extendedFaceStencilScalarGrad_8C_source.html L62-109
//   reduced          reducedFaceNormalStencil_8C_source.html L69-108
//   cell update      QGDRhoEqn_8H L40-47, QGDUEqn_8H L36-89, QGDEEqn_8H L37-76,
//                    QGDFoam_8C L149-156, hePsiQGDThermo_8C L38-126,
//                    constScPrModel1_8C L97-131, QGDThermo_8C L84-111
#include "qgd_device.hpp"

#include <stdexcept>

#include "../../include/qgd_amd.h"
#include "qgd_stencil_dev.hpp"
#include "qgd_implicit_dev.hpp"

namespace qgd {

// partial slots of the internal-face kernels are laid out for the smallest face tile (64)
__device__ __forceinline__ int faceBlocksDev(const MeshView& m) { return (m.nIF + 63) / 64; }

// ---------------------------------------------------------------------------
// QGD flux algebra on one face [QGDFoam/updateFluxes.H L41-139, explicit branch]
// g: gradients of (rho, Ux, Uy, Uz, p, e) with stride 6.
// ---------------------------------------------------------------------------
struct FaceState {
    double rhof, Uf[3], rhoUf[3], UrhoUf[9], pf, cf, Hf, gammaf, alphauf, muf, tauf;
    int implicitDiffusion;   // 1: the Navier-Stokes part of Pi and the Fourier part of q are left to the implicit solves
    // the UPW instantiations only (a `div(phiJm,U|H) Gauss upwind` entry): the two cells' U and H, which of the two fluxes takes them
    double Uo[3], Un[3], Ho, Hn;
    int upwindU, upwindH;
};

// UPW: qgdFlux with a divSchemes entry `Gauss upwind` [QGDInterpolate.H L86-104 -> fvc::flux]: psi_f = lambda (psi_O - psi_N) + psi_N with
// lambda = pos0(phiJm) (L0: upwind::weights, surfaceInterpolationScheme::interpolate); internal faces only -- a patch face keeps its patch
// value, which is what s.Uf / s.Hf hold there, so the boundary kernel instantiates UPW = false
template <bool DBG, bool UPW = false>
__device__ __forceinline__ void qgdFluxes(const FaceState& s, const double* __restrict__ g, const double S[3],
                                          double out[5], double& phiwStar, double* __restrict__ dbg, size_t dbgStride) {
    double gR[3], gP[3], gE[3], gU[9];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        gR[i] = g[i * 6 + 0];
        gU[3 * i + 0] = g[i * 6 + 1];
        gU[3 * i + 1] = g[i * 6 + 2];
        gU[3 * i + 2] = g[i * 6 + 3];
        gP[i] = g[i * 6 + 4];
        gE[i] = g[i * 6 + 5];
    }
    const double tau = s.tauf;
    const double divU = gU[0] + gU[4] + gU[8];
    // continuity [L54-72]
    double rhoW[3], jm[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double t1 = (s.Uf[k] * gR[0]) * s.Uf[0] + (s.Uf[k] * gR[1]) * s.Uf[1] + (s.Uf[k] * gR[2]) * s.Uf[2];
        const double t3 = s.rhoUf[0] * gU[0 + k] + s.rhoUf[1] * gU[3 + k] + s.rhoUf[2] * gU[6 + k];
        rhoW[k] = tau * ((t1 + (s.rhoUf[k] * divU)) + t3);
    }
    phiwStar = S[0] * rhoW[0] + S[1] * rhoW[1] + S[2] * rhoW[2];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        rhoW[k] += tau * gP[k];
        jm[k] = s.rhoUf[k] - rhoW[k];
    }
    const double phiJm = S[0] * jm[0] + S[1] * jm[1] + S[2] * jm[2];
    // momentum [L78-113]
    double Pi[9];
    const double sph = tau * ((s.Uf[0] * gP[0] + s.Uf[1] * gP[1] + s.Uf[2] * gP[2]) + (s.gammaf * s.pf * divU));
    const double s23 = (2.0 / 3.0) * divU;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const double a = s.UrhoUf[3 * i] * gU[j] + s.UrhoUf[3 * i + 1] * gU[3 + j] + s.UrhoUf[3 * i + 2] * gU[6 + j];
            double pij = tau * (a + s.Uf[i] * gP[j]);
            double t = gU[3 * i + j] + gU[3 * j + i];
            if (i == j) { pij += sph; t = t - s23; }
            Pi[3 * i + j] = s.implicitDiffusion ? pij : pij + s.muf * t;   // [updateFluxes.H L95-106]
        }
    double phiPi[3], phiJmU[3], phiP[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        phiPi[j] = S[0] * Pi[j] + S[1] * Pi[3 + j] + S[2] * Pi[6 + j];
        phiJmU[j] = phiJm * ((UPW && s.upwindU) ? lerpf(phiJm >= 0.0 ? 1.0 : 0.0, s.Uo[j], s.Un[j]) : s.Uf[j]);
        phiP[j] = S[j] * s.pf;
    }
    // energy [L119-139]
    const double phiJmH = phiJm * ((UPW && s.upwindH) ? lerpf(phiJm >= 0.0 ? 1.0 : 0.0, s.Ho, s.Hn) : s.Hf);
#if QGD_F_DIET
    const double rrho = rcpNewton(s.rhof);
    const double pr2 = s.pf * rrho * rrho;
#else
    const double pr2 = s.pf / s.rhof / s.rhof;
#endif
    double g2[3], qf[3], piU[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) g2[k] = gE[k] - pr2 * gR[k];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const double q = s.UrhoUf[3 * i] * g2[0] + s.UrhoUf[3 * i + 1] * g2[1] + s.UrhoUf[3 * i + 2] * g2[2];
        qf[i] = s.implicitDiffusion ? (-tau) * q : (-tau) * q - s.alphauf * gE[i];   // [updateFluxes.H L131-135]
        piU[i] = Pi[3 * i] * s.Uf[0] + Pi[3 * i + 1] * s.Uf[1] + Pi[3 * i + 2] * s.Uf[2];
    }
    const double phiQ = S[0] * qf[0] + S[1] * qf[1] + S[2] * qf[2];
    const double phiPiU = S[0] * piU[0] + S[1] * piU[1] + S[2] * piU[2];
    // net fluxes consumed by the three equations [QGDRhoEqn/QGDUEqn/QGDEEqn]
    out[0] = phiJm;
#pragma unroll
    for (int j = 0; j < 3; ++j) out[1 + j] = phiJmU[j] + phiP[j] - phiPi[j];
    out[4] = phiJmH + phiQ - phiPiU;
    if (DBG) {
        auto put = [&](int slot, double x) { dbg[(size_t)slot * dbgStride] = x; };
        put(DBG_PHIJM, phiJm);
        for (int j = 0; j < 3; ++j) { put(DBG_PHIJMU + j, phiJmU[j]); put(DBG_PHIP + j, phiP[j]); put(DBG_PHIPI + j, phiPi[j]); }
        put(DBG_PHIJMH, phiJmH); put(DBG_PHIQ, phiQ); put(DBG_PHIPIU, phiPiU); put(DBG_PHIW, phiwStar);
        put(DBG_PHI, S[0] * s.rhoUf[0] + S[1] * s.rhoUf[1] + S[2] * s.rhoUf[2]);
        put(DBG_TAU, tau);
        for (int k = 0; k < 9; ++k) put(DBG_GRADU + k, gU[k]);
        for (int k = 0; k < 3; ++k) { put(DBG_GRADE + k, gE[k]); put(DBG_GRADRHO + k, gR[k]); put(DBG_GRADP + k, gP[k]); }
    }
}

__device__ __forceinline__ void loadVals(const RecA& a, double* o) {
    o[0] = a.rho; o[1] = a.ux; o[2] = a.uy; o[3] = a.uz; o[4] = a.p; o[5] = a.e;
}

// ---------------------------------------------------------------------------
// What follows the gradient on an internal face: the 13 interpolations of updateFields.H, the flux algebra, the five
// net fluxes, the face's share of the Courant number.  Shared by the loads-first kernels of the 2-D stencils below.
// ---------------------------------------------------------------------------
template <bool DBG, bool UPW = false>
__device__ __forceinline__ void finishInternalFace(const MeshView& m, const CaseView& c, const GasModel& gm, const int f, const int o,
                                                   const int n, const RecA& Ao, const RecA& An, const RecB& Bo, const RecB& Bn,
                                                   const double w, const double hf, const double S[3], const double* __restrict__ g,
                                                   const int fp, const int adjustDt, double& cof, double& tauMin) {
    const size_t nF = (size_t)m.nF;
    FaceState s;
    s.rhof = lerpf(w, Ao.rho, An.rho);
    const double Uo[3] = {Ao.ux, Ao.uy, Ao.uz}, Un[3] = {An.ux, An.uy, An.uz};
    double rUo[3], rUn[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        s.Uf[k] = lerpf(w, Uo[k], Un[k]);
        rUo[k] = Ao.rho * Uo[k];
        rUn[k] = An.rho * Un[k];
        s.rhoUf[k] = lerpf(w, rUo[k], rUn[k]);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) s.UrhoUf[3 * i + j] = lerpf(w, Uo[i] * rUo[j], Un[i] * rUn[j]);
    s.pf = lerpf(w, Ao.p, An.p);
    s.cf = lerpf(w, Bo.c, Bn.c);
    s.Hf = lerpf(w, Bo.H, Bn.H);
    s.gammaf = lerpf(w, gm.gamma, gm.gamma);
    s.alphauf = lerpf(w, alphaEffOf(gm, Bo.muQGD), alphaEffOf(gm, Bn.muQGD));
    s.muf = lerpf(w, muEffOf(gm, Bo.muQGD), muEffOf(gm, Bn.muQGD));
    s.tauf = lerpf(w, Bo.aOc, Bn.aOc) * hf;  // tauQGDf = lin(aQGD/c)*hQGDf [constScPrModel1_8C L103]
    s.implicitDiffusion = gm.implicitDiffusion;
    if (UPW) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { s.Uo[k] = Uo[k]; s.Un[k] = Un[k]; }
        s.Ho = Bo.H; s.Hn = Bn.H; s.upwindU = gm.upwindU; s.upwindH = gm.upwindH;
    }
    double out[5], phiw;
    qgdFluxes<DBG, UPW>(s, g, S, out, phiw, DBG ? c.dbg + f : nullptr, nF);
#pragma unroll
    for (int k = 0; k < 5; ++k) c.flux[(size_t)k * nF + fp] = out[k];
    if (adjustDt) {
        const bool counted = (m.ghost == nullptr) || !(m.ghost[o] == 1 && m.ghost[n] == 1);
        if (counted) {
            const double ms = sqrt(S[0] * S[0] + S[1] * S[1] + S[2] * S[2]);
            const double Unf = s.Uf[0] * (S[0] / ms) + s.Uf[1] * (S[1] / ms) + s.Uf[2] * (S[2] / ms);
            cof = fmax(fabs(Unf + s.cf), fabs(Unf - s.cf)) * c.dt[0] / hf;  // [QGDCourantNo_8H L44-48]
            tauMin = s.tauf;
        }
    }
}

// ---------------------------------------------------------------------------
// leastSquares internal faces (1-D / 2-D meshes, config 2): the arithmetic of faceGradient<ST_LSQ>
// [extendedFaceStencilScalarGrad_8C L50-88], written loads-first like the 3-D kernel: labels and counts; the streamed face
// data and the first QGD_LSQ_SLOTS stencil entries (neighbour label + wf2*Gdf, coalesced out of the sliced ELL); every
// gathered record; then the ordered accumulation out of registers.  Longer stencils finish in a loop.
// ---------------------------------------------------------------------------
#define QGD_LSQ_SLOTS 6
template <bool DBG, bool UPW = false>
__global__ __launch_bounds__(QGD_BLOCK) __attribute__((amdgpu_waves_per_eu(2, 3)))
void faceFluxLsqKernel(const MeshView m, const CaseView c, const GasModel gm, const int adjustDt) {
    const int tile = xcdTile((int)gridDim.x, m.xcdRun);
    const int f = tile * QGD_BLOCK + (int)threadIdx.x;
    double cof = -1e300, tauMin = 1e300;
    if (f < m.nIF) {
        // (0) labels
        const int o = ldStream(m.own + f), n = ldStream(m.nei + f), fp = ldStream(m.fpos + f);
        const int cnt = m.lsqCnt[f];
        const bool degenerate = m.lsqDeg[f] != 0;
        const size_t base = (size_t)m.lsqSlice[f >> 6] * 64 + (f & 63);
        // (1) streamed face data and stencil entries
        const double w = ldStream(m.w + f);
        const double hf = ldStream(m.hf + f);
        const double S[3] = {ldStream(m.Sx + f), ldStream(m.Sy + f), ldStream(m.Sz + f)};
        int sc[QGD_LSQ_SLOTS];
        double gx[QGD_LSQ_SLOTS], gy[QGD_LSQ_SLOTS], gz[QGD_LSQ_SLOTS];
#pragma unroll
        for (int e = 0; e < QGD_LSQ_SLOTS; ++e) {
            const size_t i = base + (size_t)e * 64;
            const bool on = e < cnt;
            sc[e] = on ? ldStream(m.lsqCell + i) : o;
            gx[e] = on ? ldStream(m.lsqGx + i) : 0.0;
            gy[e] = on ? ldStream(m.lsqGy + i) : 0.0;
            gz[e] = on ? ldStream(m.lsqGz + i) : 0.0;
        }
        // (2) gathered records
        const RecA Ao = c.A[o], An = c.A[n];
        const RecB Bo = c.B[o], Bn = c.B[n];
        RecA R[QGD_LSQ_SLOTS];
#pragma unroll
        for (int e = 0; e < QGD_LSQ_SLOTS; ++e) R[e] = c.A[sc[e]];
        __builtin_amdgcn_sched_barrier(0);

        FaceVals<6> v;
        loadVals(Ao, v.o);
        loadVals(An, v.n);
        double g[18];
        if (degenerate) {
            faceGradient<ST_LSQ, 6, 1>(m, f, v, reinterpret_cast<const double*>(c.A), nullptr, g);  // nf*snGrad [L76-83]
        } else {
#pragma unroll
            for (int i = 0; i < 18; ++i) g[i] = 0.0;
            double pf[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) pf[k] = lerpf(w, v.o[k], v.n[k]);
#pragma unroll
            for (int e = 0; e < QGD_LSQ_SLOTS; ++e) {
                if (e < cnt) {
                    double cv[6];
                    loadVals(R[e], cv);
#pragma unroll
                    for (int k = 0; k < 6; ++k) {
                        const double dphi = cv[k] - pf[k];
                        g[0 * 6 + k] = g[0 * 6 + k] + gx[e] * dphi;
                        g[1 * 6 + k] = g[1 * 6 + k] + gy[e] * dphi;
                        g[2 * 6 + k] = g[2 * 6 + k] + gz[e] * dphi;
                    }
                }
            }
            for (int e = QGD_LSQ_SLOTS; e < cnt; ++e) {
                const size_t i = base + (size_t)e * 64;
                double cv[6];
                loadVals(c.A[m.lsqCell[i]], cv);
                const double ex = m.lsqGx[i], ey = m.lsqGy[i], ez = m.lsqGz[i];
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    const double dphi = cv[k] - pf[k];
                    g[0 * 6 + k] = g[0 * 6 + k] + ex * dphi;
                    g[1 * 6 + k] = g[1 * 6 + k] + ey * dphi;
                    g[2 * 6 + k] = g[2 * 6 + k] + ez * dphi;
                }
            }
        }
        finishInternalFace<DBG, UPW>(m, c, gm, f, o, n, Ao, An, Bo, Bn, w, hf, S, g, fp, adjustDt, cof, tauMin);
    }
    if (adjustDt) blockMaxMin(cof, tauMin, c.blkFace + 2 * (size_t)tile, false);
}

// ---------------------------------------------------------------------------
// reduced stencil internal faces: nf (x) snGrad [reducedFaceNormalStencil_8C L69-108], loads-first.
// ---------------------------------------------------------------------------
template <bool DBG, bool UPW = false>
__global__ __launch_bounds__(QGD_BLOCK) __attribute__((amdgpu_waves_per_eu(3, 4)))
void faceFluxReducedKernel(const MeshView m, const CaseView c, const GasModel gm, const int adjustDt) {
    const int tile = xcdTile((int)gridDim.x, m.xcdRun);
    const int f = tile * QGD_BLOCK + (int)threadIdx.x;
    double cof = -1e300, tauMin = 1e300;
    if (f < m.nIF) {
        const int o = ldStream(m.own + f), n = ldStream(m.nei + f), fp = ldStream(m.fpos + f);
        const double w = ldStream(m.w + f);
        const double hf = ldStream(m.hf + f);
        const double S[3] = {ldStream(m.Sx + f), ldStream(m.Sy + f), ldStream(m.Sz + f)};
        const double ms = ldStream(m.magSf + f), dn = ldStream(m.dn + f);
        const RecA Ao = c.A[o], An = c.A[n];
        const RecB Bo = c.B[o], Bn = c.B[n];
        __builtin_amdgcn_sched_barrier(0);

        double vo[6], vn[6], g[18];
        loadVals(Ao, vo); loadVals(An, vn);
        const double nx = S[0] / ms, ny = S[1] / ms, nz = S[2] / ms;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const double sn = dn * (vn[k] - vo[k]);
            g[0 * 6 + k] = nx * sn;
            g[1 * 6 + k] = ny * sn;
            g[2 * 6 + k] = nz * sn;
        }
        finishInternalFace<DBG, UPW>(m, c, gm, f, o, n, Ao, An, Bo, Bn, w, hf, S, g, fp, adjustDt, cof, tauMin);
    }
    if (adjustDt) blockMaxMin(cof, tauMin, c.blkFace + 2 * (size_t)tile, false);
}

// ---------------------------------------------------------------------------
// GaussVolPoint 2-D internal faces: the arithmetic of faceGradient<ST_GVP2> [GaussVolPointBase2D_8C L301-367], loads-first: labels and the
// two face vertices; the streamed face data incl. the six coefficients; the two cell and two vertex records.
// ---------------------------------------------------------------------------
template <bool DBG, bool UPW = false>
__global__ __launch_bounds__(QGD_BLOCK) __attribute__((amdgpu_waves_per_eu(3, 4)))
void faceFluxGvp2Kernel(const MeshView m, const CaseView c, const GasModel gm, const int adjustDt) {
    const int tile = xcdTile((int)gridDim.x, m.xcdRun);
    const int f = tile * QGD_BLOCK + (int)threadIdx.x;
    double cof = -1e300, tauMin = 1e300;
    if (f < m.nIF) {
        const size_t nF = (size_t)m.nF;
        const int o = ldStream(m.own + f), n = ldStream(m.nei + f), fp = ldStream(m.fpos + f);
        const int2 ip = m.ip13[f];
        const double w = ldStream(m.w + f);
        const double hf = ldStream(m.hf + f);
        const double S[3] = {ldStream(m.Sx + f), ldStream(m.Sy + f), ldStream(m.Sz + f)};
        const double c1 = ldStream(m.c2d + 0 * nF + f), c2 = ldStream(m.c2d + 1 * nF + f), c3 = ldStream(m.c2d + 2 * nF + f),
                     c4 = ldStream(m.c2d + 3 * nF + f), mv42 = ldStream(m.c2d + 4 * nF + f), mv13 = ldStream(m.c2d + 5 * nF + f);
        const RecA Ao = c.A[o], An = c.A[n];
        const RecB Bo = c.B[o], Bn = c.B[n];
        const RecA Pa = c.P[ip.x], Pb = c.P[ip.y];
        __builtin_amdgcn_sched_barrier(0);

        double vo[6], vn[6], pa[6], pb[6], g[18];
        loadVals(Ao, vo); loadVals(An, vn); loadVals(Pa, pa); loadVals(Pb, pb);
        const int ie1 = m.ie1, ie2 = m.ie2;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const double dfdn = (vn[k] - vo[k]) / mv42;   // [2D.C L317-321]
            const double dfdt = (pb[k] - pa[k]) / mv13;   // [2D.C L322-328]
            const double g1 = (dfdn * c1 - dfdt * c2), g2 = (dfdt * c3 - dfdn * c4);
#pragma unroll
            for (int d = 0; d < 3; ++d) g[d * 6 + k] = (d == ie1) ? g1 : ((d == ie2) ? g2 : 0.0);  // no dynamic register index
        }
        finishInternalFace<DBG, UPW>(m, c, gm, f, o, n, Ao, An, Bo, Bn, w, hf, S, g, fp, adjustDt, cof, tauMin);
    }
    if (adjustDt) blockMaxMin(cof, tauMin, c.blkFace + 2 * (size_t)tile, false);
}

// ---------------------------------------------------------------------------
// Per-term fvsc entries [fvsc_8C L51-58]: `fvsc{default GaussVolPoint; grad(p) reduced;}` gives the four gradients of updateFluxes.H
// L41-65 different stencils.  Internal faces of such a case: the six-component gradient by STA and by STB through the generic
// faceGradient (the arithmetic the fused kernels restate), component k taken from STB's where bit k of maskB is set
// (k = rho, Ux, Uy, Uz, p, e), then the common tail.  Twice the gradient work of a uniform case; the flux algebra is the same.
// ---------------------------------------------------------------------------
template <int STA, int STB, bool DBG, bool UPW>
__global__ __launch_bounds__(QGD_BLOCK) void faceFluxMixedKernel(const MeshView m, const CaseView c, const GasModel gm, const int adjustDt,
                                                                const int maskB) {
    const int tile = xcdTile((int)gridDim.x, m.xcdRun);
    const int f = tile * QGD_BLOCK + (int)threadIdx.x;
    double cof = -1e300, tauMin = 1e300;
    if (f < m.nIF) {
        const int o = m.own[f], n = m.nei[f], fp = m.fpos[f];
        const double w = m.w[f], hf = m.hf[f];
        const double S[3] = {m.Sx[f], m.Sy[f], m.Sz[f]};
        const RecA Ao = c.A[o], An = c.A[n];
        const RecB Bo = c.B[o], Bn = c.B[n];
        FaceVals<6> v;
        loadVals(Ao, v.o); loadVals(An, v.n);
#pragma unroll
        for (int k = 0; k < 6; ++k) v.sn[k] = 0.0;
        double gA[18], gB[18], g[18];
        faceGradient<STA, 6, 1>(m, f, v, reinterpret_cast<const double*>(c.A), reinterpret_cast<const double*>(c.P), gA);
        faceGradient<STB, 6, 1>(m, f, v, reinterpret_cast<const double*>(c.A), reinterpret_cast<const double*>(c.P), gB);
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int k = 0; k < 6; ++k) g[i * 6 + k] = ((maskB >> k) & 1) ? gB[i * 6 + k] : gA[i * 6 + k];
        finishInternalFace<DBG, UPW>(m, c, gm, f, o, n, Ao, An, Bo, Bn, w, hf, S, g, fp, adjustDt, cof, tauMin);
    }
    if (adjustDt) blockMaxMin(cof, tauMin, c.blkFace + 2 * (size_t)tile, false);
}

// ---------------------------------------------------------------------------
// GaussVolPoint 3-D internal faces: the bench path.  Same arithmetic as faceGradient<ST_GVP3>, written so that
// every load of a face is in flight before the first use: (0) labels, (1) the 18 streamed doubles of the face,
// (2) the 2 cell and 4 vertex records; then ~600 fp64 operations out of registers.  One memory round trip per
// dependency level instead of one per gradient component; the register budget is traded for that on purpose.
// ---------------------------------------------------------------------------
#ifndef QGD_F_WAVES_MIN
#define QGD_F_WAVES_MIN 2
#endif
#ifndef QGD_F_WAVES_MAX
#define QGD_F_WAVES_MAX 3
#endif
// Gauss coefficients of one internal face from the geometry records alone (so that a kernel holding its records in LDS can
// let go of the geometry before it touches the field values).  Quad [GaussVolPointBase3D_8C L346-389, L488-513] in difference
// form: with a2=-a0, a3=-a1, a4=-a5,
//   V d_d phi = a5_d (phi_O - phi_N) + a0_d (phi_1 - phi_3) + a1_d (phi_2 - phi_4),
//   6 a0 = (N-O) x (p2-p4),  6 a1 = (N-O) x (p3-p1),  6 a5 = (p1-p3) x (p2-p4),  6 V = -(p3-p1).(6 a0)
// (the 1/6 cancel): coef = {a0[3], a1[3], a5[3]}, rV = 1/(6V).  Triangle [L193-229]: coef = t[12] of gvpTriCoef.
// the quadrilateral's coefficients from its three difference vectors N-O, p2-p4, p3-p1 (one definition: the fused kernel forms the
// differences one after the other out of LDS)
__device__ __forceinline__ void gvp3QuadCoefs(const double (&NO)[3], const double (&d24)[3], const double (&d31)[3], double (&coef)[12], double& rV) {
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const int u = (d + 1) % 3, w2 = (d + 2) % 3;
        coef[d] = NO[u] * d24[w2] - NO[w2] * d24[u];
        coef[3 + d] = NO[u] * d31[w2] - NO[w2] * d31[u];
        coef[6 + d] = d24[u] * d31[w2] - d24[w2] * d31[u];
    }
    coef[9] = coef[10] = coef[11] = 0.0;
    rV = -QGD_RCP(d31[0] * coef[0] + d31[1] * coef[1] + d31[2] * coef[2]);
}
__device__ __forceinline__ void gvp3Coefs(const int kind, const double4& cO, const double4& cN, const double4& x0, const double4& x1,
                                          const double4& x2, const double4& x3, double (&coef)[12], double& rV) {
    if (kind == 0) {
        const double NO[3] = {cN.x - cO.x, cN.y - cO.y, cN.z - cO.z};
        const double d24[3] = {x1.x - x3.x, x1.y - x3.y, x1.z - x3.z};
        const double d31[3] = {x2.x - x0.x, x2.y - x0.y, x2.z - x0.z};
        gvp3QuadCoefs(NO, d24, d31, coef, rV);
    } else if (kind == 1) {
        gvpTriCoef(cO, cN, x0, x1, x2, coef, rV);
    } else {
#pragma unroll
        for (int i = 0; i < 12; ++i) coef[i] = 0.0;
        rV = 0.0;
    }
}

// QGD_F_PRIO: wave priority inside the staged face kernel.  Its rate is tiles in flight / length of a tile's chain of dependent round
// trips (profiles/r04_ab_face_latency_chain.txt), and the instructions between those round trips -- address arithmetic, LDS stores, the
// barrier -- compete for issue with the flux algebra of the other waves of the SIMD: 0 never raised, 1 raised until the piece loads are
// out, 2 until the tile is in LDS (default: F 6.75 / 6.63 / 6.50 ms for 0 / 1 / 2; kept up until the records are in registers -- with the
// Gauss coefficients' arithmetic inside the window -- 7.0-7.3 ms).
#ifndef QGD_F_PRIO
#define QGD_F_PRIO 2
#endif
#ifndef QGD_P_PRIO
#define QGD_P_PRIO 1   // the same in the vertex kernel, until its gathers are out: P 2.53 -> 2.48 ms (nothing in the cell kernel: not kept there)
#endif
// everything after the loads of one internal face: gradient coefficients from the geometry, the 6-component gradient, the 13
// interpolations, the flux algebra, the five net fluxes (slot-major position fp), the face's share of the Courant number
// component k of the quadrilateral's gradient [GaussVolPointBase3D_8C L346-389, L488-513 in difference form, gvp3Coefs]: one definition for the
// kernels that hold the six records in registers and the one that reads them out of LDS component by component
__device__ __forceinline__ void gvp3QuadGradK(const double (&coef)[12], const double rV6, const double vo, const double vn, const double p0,
                                              const double p1, const double p2, const double p3, double (&g)[18], const int k) {
    const double D5 = (vo - vn) * rV6, D0 = (p0 - p2) * rV6, D1 = (p1 - p3) * rV6;
#pragma unroll
    for (int d = 0; d < 3; ++d) g[d * 6 + k] = coef[6 + d] * D5 + coef[d] * D0 + coef[3 + d] * D1;
}

// what follows the gradient on an internal face of the 3-D GaussVolPoint kernels: the 13 interpolations, the flux algebra, the five net
// fluxes (fluxOut / fluxStride: c.flux + fp at a stride of nF faces, or registers / LDS of the fused kernel), the face's share of the Courant number
template <bool DBG, bool UPW = false>
__device__ __forceinline__ void gvp3FaceTail(const MeshView& m, const CaseView& c, const GasModel& gm, const int f, const double w, const double hf,
                                             const double (&S)[3], const RecA& Ao, const RecA& An, const RecB& Bo, const RecB& Bn,
                                             const double (&g)[18], const int adjustDt, double& cof, double& tauMin, double* const fluxOut,
                                             const size_t fluxStride) {
    const size_t nF = (size_t)m.nF;
    FaceState s;
    s.rhof = lerpf(w, Ao.rho, An.rho);
    const double Uo[3] = {Ao.ux, Ao.uy, Ao.uz}, Un[3] = {An.ux, An.uy, An.uz};
    double rUo[3], rUn[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        s.Uf[k] = lerpf(w, Uo[k], Un[k]);
        rUo[k] = Ao.rho * Uo[k];
        rUn[k] = An.rho * Un[k];
        s.rhoUf[k] = lerpf(w, rUo[k], rUn[k]);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) s.UrhoUf[3 * i + j] = lerpf(w, Uo[i] * rUo[j], Un[i] * rUn[j]);
    s.pf = lerpf(w, Ao.p, An.p);
    s.cf = lerpf(w, Bo.c, Bn.c);
    s.Hf = lerpf(w, Bo.H, Bn.H);
    s.gammaf = lerpf(w, gm.gamma, gm.gamma);
    s.alphauf = lerpf(w, alphaEffOf(gm, Bo.muQGD), alphaEffOf(gm, Bn.muQGD));
    s.muf = lerpf(w, muEffOf(gm, Bo.muQGD), muEffOf(gm, Bn.muQGD));
    s.tauf = lerpf(w, Bo.aOc, Bn.aOc) * hf;
    s.implicitDiffusion = gm.implicitDiffusion;
    if (UPW) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { s.Uo[k] = Uo[k]; s.Un[k] = Un[k]; }
        s.Ho = Bo.H; s.Hn = Bn.H; s.upwindU = gm.upwindU; s.upwindH = gm.upwindH;
    }
    double out[5], phiw;
    qgdFluxes<DBG, UPW>(s, g, S, out, phiw, DBG ? c.dbg + f : nullptr, nF);
#pragma unroll
    for (int k = 0; k < 5; ++k) fluxOut[(size_t)k * fluxStride] = out[k];
    if (adjustDt) {
        bool counted = true;
        if (m.ghost != nullptr) counted = !(m.ghost[m.own[f]] == 1 && m.ghost[m.nei[f]] == 1);
        if (counted) {
            const double ms = sqrt(S[0] * S[0] + S[1] * S[1] + S[2] * S[2]);
            const double Unf = s.Uf[0] * (S[0] / ms) + s.Uf[1] * (S[1] / ms) + s.Uf[2] * (S[2] / ms);
            cof = fmax(fabs(Unf + s.cf), fabs(Unf - s.cf)) * c.dt[0] / hf;
            tauMin = s.tauf;
        }
    }
}


template <bool DBG, bool UPW = false>
__device__ __forceinline__ void gvp3FaceBody(const MeshView& m, const CaseView& c, const GasModel& gm, const int f, const int fp, const int kind,
                                             const double w, const double hf, const double (&S)[3], const RecA& Ao, const RecA& An,
                                             const RecB& Bo, const RecB& Bn, const RecA& q0, const RecA& q1, const RecA& q2, const RecA& q3,
                                             const double (&coef)[12], const double rVc, const double msO, const double dnO,
                                             const int adjustDt, double& cof, double& tauMin, double* const fluxOut, const size_t fluxStride) {
    // fluxOut / fluxStride: where the five net fluxes go -- c.flux + fp at a stride of nF faces, or a slot of the fused kernel's LDS
    // msO, dnO: |Sf| and deltaCoeffs of the face, read only on meshes that have faces with more than four vertices
    FaceVals<6> v;
    loadVals(Ao, v.o);
    loadVals(An, v.n);
    double g[18];
    if (kind == 0) {
        // Quad: coef = {a0, a1, a5}, rVc = 1/(6V) (gvp3Coefs)
        double p0[6], p1[6], p2[6], p3[6];
        loadVals(q0, p0); loadVals(q1, p1); loadVals(q2, p2); loadVals(q3, p3);
#pragma unroll
        for (int k = 0; k < 6; ++k) gvp3QuadGradK(coef, rVc, v.o[k], v.n[k], p0[k], p1[k], p2[k], p3[k], g, k);
    } else if (kind == 1) {
        // Triangle [GaussVolPointBase3D_8C L193-229, L844-854], out of the records already in registers: the same
        // operations in the same order as faceGradient<ST_GVP3> (no second round of loads in mixed wavefronts)
        const double* t = coef;
        const double rV = rVc;
        double p0[6], p1[6], p2[6];
        loadVals(q0, p0); loadVals(q1, p1); loadVals(q2, p2);
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const double a0 = t[4 * d], a1 = t[4 * d + 1], a2 = t[4 * d + 2], a3 = t[4 * d + 3];
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                double sg = v.n[k] * a3;
                sg += v.o[k] * (-a3);
                sg += p0[k] * a0;
                sg += p1[k] * a1;
                sg += p2[k] * a2;
                g[d * 6 + k] = sg * rV;
            }
        }
        double dg[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) dg[j] = g[j * 6 + 1 + j];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) g[i * 6 + 1 + j] = dg[j];
    } else if (kind == 2) {
        // more than four vertices: nf (x) snGrad [GaussVolPointBase3D_8C L759-768], as faceGradient<ST_GVP3> has it
        const double nx = S[0] / msO, ny = S[1] / msO, nz = S[2] / msO;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const double sn = dnO * (v.n[k] - v.o[k]);
            g[0 * 6 + k] = nx * sn;
            g[1 * 6 + k] = ny * sn;
            g[2 * 6 + k] = nz * sn;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 18; ++i) g[i] = 0.0;
    }
    gvp3FaceTail<DBG, UPW>(m, c, gm, f, w, hf, S, Ao, An, Bo, Bn, g, adjustDt, cof, tauMin, fluxOut, fluxStride);
}

template <bool DBG, int FB, bool SGEO = false, bool UPW = false>
__global__ __launch_bounds__(FB) __attribute__((amdgpu_waves_per_eu(QGD_F_WAVES_MIN, QGD_F_WAVES_MAX)))
void faceFluxGvp3Kernel(const MeshView m, const CaseView c, const GasModel gm, const int adjustDt, const int32_t* __restrict__ tileList) {
    // tileList: the tiles the staged kernel below leaves to this one (nullptr: every tile)
    const int tile = tileList ? tileList[blockIdx.x] : xcdTile((int)gridDim.x, m.xcdRun * (QGD_BLOCK / FB));
    const int f = tile * FB + (int)threadIdx.x;
    double cof = -1e300, tauMin = 1e300;
    if (f < m.nIF) {
        // (0) labels
        const int o = ldStream(m.own + f), n = ldStream(m.nei + f), fp = ldStream(m.fpos + f);
        const int4 vt = m.verts[f];
        const int kind = m.fkind[f];
        // (1) streamed face data
        const double w = ldStream(m.w + f);
        const double hf = ldStream(m.hf + f);
        double S[3] = {0.0, 0.0, 0.0};
        if (!SGEO || kind != 0) { S[0] = ldStream(m.Sx + f); S[1] = ldStream(m.Sy + f); S[2] = ldStream(m.Sz + f); }
        double msO = 1.0, dnO = 0.0;
        if (m.hasOther) { msO = m.magSf[f]; dnO = m.dn[f]; }
        // (2) gathered records (vertex 3 is clamped for triangles; unused there)
        const RecA Ao = c.A[o], An = c.A[n];
        const RecB Bo = c.B[o], Bn = c.B[n];
        const int v3 = vt.w < 0 ? 0 : vt.w;
        const RecA q0 = c.P[vt.x], q1 = c.P[vt.y], q2 = c.P[vt.z], q3 = c.P[v3];
        // geometry the Gauss coefficients are rebuilt from (cached: shared by ~12 faces per vertex, 6 per cell)
        const double4 cO = ld3(m.Cc, o), cN = ld3(m.Cc, n);
        const double4 x0 = ld3(m.X, vt.x), x1 = ld3(m.X, vt.y), x2 = ld3(m.X, vt.z), x3 = ld3(m.X, v3);
        __builtin_amdgcn_sched_barrier(0);

        double coef[12], rVc;
        gvp3Coefs(kind, cO, cN, x0, x1, x2, x3, coef, rVc);
        if (SGEO && kind == 0) { S[0] = 0.5 * coef[6]; S[1] = 0.5 * coef[7]; S[2] = 0.5 * coef[8]; }   // Sf = (p3-p1) x (p4-p2) / 2
        gvp3FaceBody<DBG, UPW>(m, c, gm, f, fp, kind, w, hf, S, Ao, An, Bo, Bn, q0, q1, q2, q3, coef, rVc, msO, dnO, adjustDt, cof, tauMin,
                               c.flux + fp, (size_t)m.nF);
    }
    if (adjustDt) blockMaxMin<FB>(cof, tauMin, c.blkFace + 2 * (size_t)tile, false);
}

// ---------------------------------------------------------------------------
// The same faces through LDS (qgd_setup.hpp FaceTiles).  The gather kernel above asks the vector L1 for 2 + 4 records per
// face as 16-B pieces at a 48-B stride: ~40 cache-line lookups per wave instruction, 44 such instructions per wave, and
// the counters show the L1 tag pipe busy all of the kernel (TCP_GATE_EN ~ 100 %, TA address stalls 42 %) while neither
// the time nor the HBM bytes respond to better cell orders.  Here the workgroup loads each DISTINCT record of its tile
// once, as consecutive 16-B (records) and 8-B (geometry) pieces -- lane q takes piece q, so a wave instruction covers
// whole cache lines -- parks them in LDS and every face reads its six records from there.  Same arithmetic, same
// operation order, bit-identical fluxes.
// ---------------------------------------------------------------------------
typedef double v2d __attribute__((ext_vector_type(2)));   // one 16-B piece
constexpr int kFusedCapCDev = 320, kFusedCapVDev = 256, kFusedCapFDev = 512, kFusedCapTotDev = 384;   // = kFusedCap{C,V,F,Tot} of qgd_setup.hpp
#ifndef QGD_F_BUF
#define QGD_F_BUF 0
#endif
#ifndef QGD_FT_WAVES_MIN
#define QGD_FT_WAVES_MIN 2
#endif
#ifndef QGD_FT_WAVES_MAX
#define QGD_FT_WAVES_MAX 3
#endif
// FIXED: the tile lists at a fixed stride (MeshView::tileCellsFix / tileVertsFix: every tile's list padded to the longest one by
// repeating its last label, tileFlag = 1 for the tiles left to the gather kernel).  A tile's life is a chain of dependent memory
// round trips -- list offsets -> labels -> pieces -> LDS -> algebra -- and with 6 workgroups per CU in flight the kernel's rate is
// tiles in flight / length of that chain, not bytes (29 % fewer L2 misses under a pencil order: -1.6 % time,
// profiles/r04_ab_pencil_xcd_matched.txt).  With computed offsets the chain is one round trip shorter.
template <int FB, int WAVES, bool SGEO = false, bool FIXED = false, bool UPW = false>
__global__ __launch_bounds__(FB) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES)))
void faceFluxGvp3TileKernel(const MeshView m, const CaseView c, const GasModel gm, const int adjustDt) {
    extern __shared__ v2d tileLds[];
#if QGD_F_PRIO
    __builtin_amdgcn_s_setprio(3);   // the address arithmetic in front of the loads ahead of the other waves' flux algebra
#endif
    constexpr int KC = 4, KB2 = 3, KV = 5;   // piece loads per thread: ceil(3 capC / FB), ceil(2 capC / FB), ceil(3 capV / FB) (faceTileCap*)
    static_assert(3 * (FB + FB / 16) <= KC * FB && 2 * (FB + FB / 16) <= KB2 * FB && 3 * (((FB * 23) / 16 + 7) / 8 * 8) <= KV * FB, "caps");
    const int tile = xcdTile((int)gridDim.x, m.xcdRun * (QGD_BLOCK / FB));
    const int tid = (int)threadIdx.x;
    const int f = tile * FB + tid;
    const bool active = f < m.nIF;
    int cOff, vOff, nUc, nUv, spill = 0;
    const int32_t* __restrict__ tCells = m.tileCells;
    const int32_t* __restrict__ tVerts = m.tileVerts;
    if (FIXED) {
        nUc = m.tileMaxC; nUv = m.tileMaxV; cOff = tile * nUc; vOff = tile * nUv;
        tCells = m.tileCellsFix; tVerts = m.tileVertsFix;
        spill = m.tileFlag[tile];          // (arrives with the labels: no round trip of its own)
    } else {
        cOff = m.tileOff[2 * tile]; vOff = m.tileOff[2 * tile + 1];
        nUc = m.tileOff[2 * tile + 2] - cOff; nUv = m.tileOff[2 * tile + 3] - vOff;
        if (nUc == 0) return;   // more distinct records than the piece loads below cover: in m.tileSpill, done by the gather kernel
    }
    v2d* const sA = tileLds;               // 3 nUc pieces: cell RecA
    v2d* const sB = sA + 3 * nUc;          // 2 nUc: cell RecB
    v2d* const sP = sB + 2 * nUc;          // 3 nUv: vertex RecA
    double* const sC = reinterpret_cast<double*>(sP + 3 * nUv);   // 3 nUc: cell centres
    double* const sX = sC + 3 * nUc;           // 3 nUv: vertex coordinates
    // (0) labels of this thread's pieces; the face's own streamed data
    const int fl = active ? f : m.nIF - 1;
    const unsigned lc = ldStream(m.locC + fl);
    const uint2 lv = m.locV[fl];
    const int fp = ldStream(m.fpos + fl);
    const int kind = m.fkind[fl];
    const double w = ldStream(m.w + fl);
    const double hf = ldStream(m.hf + fl);
    double S[3] = {0.0, 0.0, 0.0};
    if (!SGEO) { S[0] = ldStream(m.Sx + fl); S[1] = ldStream(m.Sy + fl); S[2] = ldStream(m.Sz + fl); }
    // (what depends on the face's kind is loaded BELOW, after the piece loads have gone out: a branch on a loaded value here makes
    // the label loads wait for a whole memory round trip)
    int idC[KC], idB[KB2], idV[KV];
#pragma unroll
    for (int k = 0; k < KC; ++k) {
        const int q = tid + k * FB, r = (q * 43691) >> 17;   // q / 3 for q < 98304
        idC[k] = tCells[cOff + min(r, nUc - 1)] * 3 + (q - 3 * r);
    }
#pragma unroll
    for (int k = 0; k < KB2; ++k) {
        const int q = tid + k * FB, r = q >> 1;
        idB[k] = tCells[cOff + min(r, nUc - 1)] * 2 + (q & 1);
    }
#pragma unroll
    for (int k = 0; k < KV; ++k) {
        const int q = tid + k * FB, r = (q * 43691) >> 17;
        idV[k] = tVerts[vOff + min(r, nUv - 1)] * 3 + (q - 3 * r);
    }
    // (1) the distinct records of the tile, piece by piece (pieces past the end repeat the last record and are dropped)
    const v2d* __restrict__ gA = reinterpret_cast<const v2d*>(c.A);
    const v2d* __restrict__ gB = reinterpret_cast<const v2d*>(c.B);
    const v2d* __restrict__ gP = reinterpret_cast<const v2d*>(c.P);
    v2d dA[KC], dB[KB2], dP[KV];
    double dC[KC], dX[KV];
#if QGD_F_BUF
    // QGD_F_BUF (compile-time experiment): the piece gathers as raw buffer loads -- a 32-bit byte offset per
    // load instead of a 64-bit address (one v_lshlrev_b32 in place of v_ashrrev + v_lshl_add_u64); needs every array below 4 GiB
    (void)gA; (void)gB; (void)gP;
    typedef unsigned int v4u __attribute__((ext_vector_type(4)));
    typedef unsigned int v2u __attribute__((ext_vector_type(2)));
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)c.A, 0, (int)0xffffffffu, 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)c.B, 0, (int)0xffffffffu, 0x00020000);
    const __amdgpu_buffer_rsrc_t rP = __builtin_amdgcn_make_buffer_rsrc((void*)c.P, 0, (int)0xffffffffu, 0x00020000);
    const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc((void*)m.Cc, 0, (int)0xffffffffu, 0x00020000);
    const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void*)m.X, 0, (int)0xffffffffu, 0x00020000);
#pragma unroll
    for (int k = 0; k < KC; ++k) {
        dA[k] = __builtin_bit_cast(v2d, __builtin_amdgcn_raw_buffer_load_b128(rA, (int)((unsigned)idC[k] << 4), 0, 0));
        dC[k] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rC, (int)((unsigned)idC[k] << 3), 0, 0));
    }
#pragma unroll
    for (int k = 0; k < KB2; ++k) dB[k] = __builtin_bit_cast(v2d, __builtin_amdgcn_raw_buffer_load_b128(rB, (int)((unsigned)idB[k] << 4), 0, 0));
#pragma unroll
    for (int k = 0; k < KV; ++k) {
        dP[k] = __builtin_bit_cast(v2d, __builtin_amdgcn_raw_buffer_load_b128(rP, (int)((unsigned)idV[k] << 4), 0, 0));
        dX[k] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rX, (int)((unsigned)idV[k] << 3), 0, 0));
    }
#else
#pragma unroll
    for (int k = 0; k < KC; ++k) { dA[k] = gA[idC[k]]; dC[k] = m.Cc[idC[k]]; }
#pragma unroll
    for (int k = 0; k < KB2; ++k) dB[k] = gB[idB[k]];
#pragma unroll
    for (int k = 0; k < KV; ++k) { dP[k] = gP[idV[k]]; dX[k] = m.X[idV[k]]; }
#endif
#if QGD_F_PRIO == 1
    __builtin_amdgcn_s_setprio(0);
#endif
    double msO = 1.0, dnO = 0.0;
    if (SGEO && kind != 0) { S[0] = ldStream(m.Sx + fl); S[1] = ldStream(m.Sy + fl); S[2] = ldStream(m.Sz + fl); }
    if (m.hasOther) { msO = m.magSf[fl]; dnO = m.dn[fl]; }
    __builtin_amdgcn_sched_barrier(0);
    if (FIXED && spill) return;   // in m.tileSpill, done by the gather kernel (the piece loads above went to record 0: harmless)
#pragma unroll
    for (int k = 0; k < KC; ++k) { const int q = tid + k * FB; if (q < 3 * nUc) { sA[q] = dA[k]; sC[q] = dC[k]; } }
#pragma unroll
    for (int k = 0; k < KB2; ++k) { const int q = tid + k * FB; if (q < 2 * nUc) sB[q] = dB[k]; }
#pragma unroll
    for (int k = 0; k < KV; ++k) { const int q = tid + k * FB; if (q < 3 * nUv) { sP[q] = dP[k]; sX[q] = dX[k]; } }
    __syncthreads();
#if QGD_F_PRIO == 2
    __builtin_amdgcn_s_setprio(0);
#endif
    // (2) every face picks its records out of LDS
    double cof = -1e300, tauMin = 1e300;
    if (active) {
        const int lo = (int)(lc & 0xffffu), ln = (int)(lc >> 16);
        const int v0 = (int)(lv.x & 0xffffu), v1 = (int)(lv.x >> 16), v2 = (int)(lv.y & 0xffffu), v3 = (int)(lv.y >> 16);
        // geometry first: once the Gauss coefficients are there its 36 registers are free for the field values
        auto l3 = [](const double* p, int i) { return make_double4(p[3 * i], p[3 * i + 1], p[3 * i + 2], 0.0); };
        double coef[12], rVc;
        {
            const double4 cO = l3(sC, lo), cN = l3(sC, ln);
            const double4 x0 = l3(sX, v0), x1 = l3(sX, v1), x2 = l3(sX, v2), x3 = l3(sX, v3);
            gvp3Coefs(kind, cO, cN, x0, x1, x2, x3, coef, rVc);
            if (SGEO && kind == 0) { S[0] = 0.5 * coef[6]; S[1] = 0.5 * coef[7]; S[2] = 0.5 * coef[8]; }   // Sf = (p3-p1) x (p4-p2) / 2
        }
        __builtin_amdgcn_sched_barrier(0);
        const RecA Ao = *reinterpret_cast<const RecA*>(sA + 3 * lo), An = *reinterpret_cast<const RecA*>(sA + 3 * ln);
        const RecB Bo = *reinterpret_cast<const RecB*>(sB + 2 * lo), Bn = *reinterpret_cast<const RecB*>(sB + 2 * ln);
        const RecA q0 = *reinterpret_cast<const RecA*>(sP + 3 * v0), q1 = *reinterpret_cast<const RecA*>(sP + 3 * v1),
                   q2 = *reinterpret_cast<const RecA*>(sP + 3 * v2), q3 = *reinterpret_cast<const RecA*>(sP + 3 * v3);
        gvp3FaceBody<false, UPW>(m, c, gm, f, fp, kind, w, hf, S, Ao, An, Bo, Bn, q0, q1, q2, q3, coef, rVc, msO, dnO, adjustDt, cof, tauMin,
                                 c.flux + fp, (size_t)m.nF);
    }
    if (adjustDt) blockMaxMin<FB>(cof, tauMin, c.blkFace + 2 * (size_t)tile, false);
}

// patch snGrad of the six case fields on boundary face (global label f)
// perComponent: the 2-D GaussVolPoint gradient of U goes component by component [GaussVolPointBase.C L79-87], each a scalar field whose
// patch field on a symmetryPlane / symmetry patch is the scalar symmetry one (L0: fvPatchField::New lets the constraint type win), snGrad = 0
__device__ __forceinline__ void boundaryVals(const MeshView& m, const CaseView& c, const PatchBCDev& bc, const int f,
                                             const RecA& Ao, const RecA& Ab, FaceVals<6>& v, const bool perComponent) {
    const int b = f - m.nIF;
    loadVals(Ao, v.o);
    loadVals(Ab, v.n);
    v.n[4] = c.bPmid[b];        // patch pressure after the mid-step BC evaluation
    const double dc = m.dn[f];  // deltaCoeffs on boundary faces
#pragma unroll
    for (int k = 0; k < 6; ++k) v.sn[k] = dc * (v.n[k] - v.o[k]);  // fvPatchField::snGrad (L0)
    if (bc.bcU == QGD_BC_SLIP) {
        // basicSymmetry::snGrad (L0): (transform(I - 2 nn, pif) - pif)*(deltaCoeffs/2)
        double n[3];
        symmNormal(m, bc, f, n);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const double tv = ((i == 0 ? 1.0 : 0.0) - 2.0 * (n[i] * n[0])) * v.o[1] + ((i == 1 ? 1.0 : 0.0) - 2.0 * (n[i] * n[1])) * v.o[2] +
                              ((i == 2 ? 1.0 : 0.0) - 2.0 * (n[i] * n[2])) * v.o[3];
            v.sn[1 + i] = (tv - v.o[1 + i]) * (dc / 2.0);
        }
    } else if (bc.bcU == QGD_BC_ZEROGRADIENT || bc.bcU == QGD_BC_NONE) {
        v.sn[1] = v.sn[2] = v.sn[3] = 0.0;
    }
    if (bc.bcP == QGD_BC_QGDFLUX) v.sn[4] = c.bG[b];             // fixedGradient::snGrad = gradient()
    else if (bc.bcP != QGD_BC_FIXEDVALUE) v.sn[4] = 0.0;          // zeroGradient
    if (bc.bcT != QGD_BC_FIXEDVALUE) v.sn[5] = 0.0;               // gradientEnergy with zero gradient
    if (bc.ptype == QGD_PATCH_SYMMETRYPLANE || bc.ptype == QGD_PATCH_SYMMETRY) {
        v.sn[0] = 0.0;   // rho's calculated patch is the scalar symmetry patch there: snGrad = 0 whatever psi_b p_b rounds to
        if (perComponent) v.sn[1] = v.sn[2] = v.sn[3] = 0.0;
    }
}

template <int ST, bool DBG, int STB = ST>
__global__ __launch_bounds__(QGD_BLOCK) void boundaryFaceFluxKernel(const MeshView m, const CaseView c, const GasModel gm,
                                                                   const PatchBCDev* __restrict__ bcs, const int phiwOnly,
                                                                   const int adjustDt, const int maskB = 0) {
    // STB != ST: per-term fvsc entries (faceFluxMixedKernel): component k of the gradient by STB where bit k of maskB is set
    // phiwOnly: 0 = fluxes; 1 = phiwStar + the mid-step pressure of qgdFlux patches only; 2 = fluxes, then the patch
    // pressure becomes the mid-step one
    const int b = blockIdx.x * QGD_BLOCK + threadIdx.x;
    double cof = -1e300, tauMin = 1e300;
    if (b < m.nBF) {
        const int f = m.nIF + b;
        const PatchBCDev bc = bcs[m.bPatch[b]];
        const bool live = m.fkind[f] != 3 && bc.ptype != QGD_PATCH_HALO && !(phiwOnly == 1 && bc.bcP != QGD_BC_QGDFLUX);
        if (live) {
            const int o = m.own[f];
            const RecA Ao = c.A[o], Ab = c.bA[b];
            const RecB Bb = c.bB[b];
            FaceVals<6> v;
            boundaryVals(m, c, bc, f, Ao, Ab, v, (STB != ST && (maskB & 2) ? STB : ST) == ST_GVP2);   // (the stencil of grad(U))
            double g[18];
            faceGradient<ST, 6, 1>(m, f, v, reinterpret_cast<const double*>(c.A), reinterpret_cast<const double*>(c.P), g);
            if (STB != ST) {
                double gB[18];
                faceGradient<STB, 6, 1>(m, f, v, reinterpret_cast<const double*>(c.A), reinterpret_cast<const double*>(c.P), gB);
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int k = 0; k < 6; ++k) if ((maskB >> k) & 1) g[i * 6 + k] = gB[i * 6 + k];
            }
            FaceState s;
            s.rhof = Ab.rho;
            const double Ub[3] = {Ab.ux, Ab.uy, Ab.uz};
            double rUb[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) { s.Uf[k] = Ub[k]; rUb[k] = c.bRhoLag[b] * Ub[k]; s.rhoUf[k] = rUb[k]; }  // rhoU_b [QGDUEqn_8H L88-89]
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) s.UrhoUf[3 * i + j] = Ub[i] * rUb[j];
            s.pf = Ab.p; s.cf = Bb.c; s.Hf = Bb.H; s.gammaf = gm.gamma;
            s.alphauf = alphaEffOf(gm, Bb.muQGD);
            s.muf = muEffOf(gm, Bb.muQGD);
            const double hf = m.hf[f];
            s.tauf = Bb.aOc * hf;
            s.implicitDiffusion = gm.implicitDiffusion;
            const double S[3] = {m.Sx[f], m.Sy[f], m.Sz[f]};
            double out[5], phiw;
            qgdFluxes<DBG>(s, g, S, out, phiw, DBG ? c.dbg + f : nullptr, (size_t)m.nF);
            c.bPhiw[b] = phiw;
            if (phiwOnly == 1) {
                // p's boundary conditions re-evaluated in the middle of updateFluxes.H with the fresh phiwStar
                // [qgdFluxFvPatchScalarField_8C L166-192]; bPmid holds the patch pressure after that evaluation
                const double grad = -(phiw / s.tauf / m.magSf[f]);
                c.bG[b] = grad;
                c.bPmid[b] = Ao.p + grad / m.dn[f];
            } else {
#pragma unroll
                for (int k = 0; k < 5; ++k) c.flux[(size_t)k * m.nF + f] = out[k];
                if (phiwOnly == 2) {
                    // the face value pf above came from the old patch pressure; from here on the patch field holds the
                    // mid-step value (one array in the reference: p.boundaryField())
                    const double pm = c.bPmid[b];
                    if (Ab.p != pm) {
                        RecA a = Ab;
                        a.p = pm;
                        c.bA[b] = a;
                        const double rE = c.bRhoLag[b] * (a.e + 0.5 * (a.ux * a.ux + a.uy * a.uy + a.uz * a.uz));
                        c.bB[b].H = (rE + a.p) / a.rho;
                    }
                }
                if (adjustDt && !(m.ghost && m.ghost[o] == 1)) {
                    const double ms = m.magSf[f];
                    const double Unf = s.Uf[0] * (S[0] / ms) + s.Uf[1] * (S[1] / ms) + s.Uf[2] * (S[2] / ms);
                    cof = fmax(fabs(Unf + s.cf), fabs(Unf - s.cf)) * c.dt[0] / hf;
                    tauMin = s.tauf;
                }
            }
        }
    }
    if (adjustDt && phiwOnly != 1) blockMaxMin(cof, tauMin, c.blkFace + 2 * ((size_t)faceBlocksDev(m) + blockIdx.x), false);
}

// ---------------------------------------------------------------------------
// vertex interpolation (L0: volPointInterpolation)
// ---------------------------------------------------------------------------
template <int NC>
__global__ __launch_bounds__(QGD_BLOCK) void pointInterpKernel(const MeshView m, const double* __restrict__ cellF,
                                                              const int cellStride, double* __restrict__ ptF) {
    const int p = xcdTile((int)gridDim.x, m.xcdRun) * QGD_BLOCK + threadIdx.x;
    if (p >= m.nP) return;
    const int n = m.pcCount[p];
    if (n == 0) return;  // patch point: written by boundaryPointKernel
    const size_t base = (size_t)m.pcSlice[p >> 6] * 64 + (p & 63);
    double acc[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) acc[k] = 0.0;
    for (int i = 0; i < n; ++i) {
        const double w = m.pcW[base + (size_t)i * 64];
        const double* cv = cellF + (size_t)m.pcCell[base + (size_t)i * 64] * cellStride;
#pragma unroll
        for (int k = 0; k < NC; ++k) acc[k] += w * cv[k];
    }
    double* o = ptF + (size_t)p * NC;
#pragma unroll
    for (int k = 0; k < NC; ++k) o[k] = acc[k];
}

// The case's vertex kernel: the gather list of a wave is read as contiguous runs (sliced ELL), the cell records
// as whole 48-B records (3 x dwordx4), all gathers of a point are issued before the first use.
#ifndef QGD_P_WAVES_MIN
#define QGD_P_WAVES_MIN 3
#endif
#ifndef QGD_P_WAVES_MAX
#define QGD_P_WAVES_MAX 4
#endif
template <int PB>
__global__ __launch_bounds__(PB) __attribute__((amdgpu_waves_per_eu(QGD_P_WAVES_MIN, QGD_P_WAVES_MAX)))
void pointInterpRecKernel(const MeshView m, const RecA* __restrict__ A, RecA* __restrict__ P) {
#if QGD_P_PRIO
    __builtin_amdgcn_s_setprio(3);
#endif
    const int p = xcdTile((int)gridDim.x, m.xcdRun * (QGD_BLOCK / PB)) * PB + threadIdx.x;
    if (p >= m.nP) return;
    const int n = m.pcCount[p];
    if (n == 0) return;
    const size_t base = (size_t)m.pcSlice[p >> 6] * 64 + (p & 63);
    RecA acc = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    // eight cells per pass, every 48-B gather of the pass in flight before the ordered sum.  A wavefront of interior
    // hexahedral vertices (8 cells each) takes the unpredicated pass; any other (2-D meshes: 4 cells; triangles,
    // polyhedra) the predicated loop as a whole.
    if (__ballot(n != 8) == 0) {
        int id[8];
        double w[8];
        RecA r[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { id[q] = m.pcCell[base + (size_t)q * 64]; w[q] = m.pcW[base + (size_t)q * 64]; }
#pragma unroll
        for (int q = 0; q < 8; ++q) r[q] = A[id[q]];
#if QGD_P_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            acc.rho += w[q] * r[q].rho; acc.ux += w[q] * r[q].ux; acc.uy += w[q] * r[q].uy;
            acc.uz += w[q] * r[q].uz; acc.p += w[q] * r[q].p; acc.e += w[q] * r[q].e;
        }
    } else {
        for (int i = 0; i < n; i += 8) {
            int id[8];
            double w[8];
            RecA r[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const bool on = i + q < n;
                id[q] = on ? m.pcCell[base + (size_t)(i + q) * 64] : 0;
                w[q] = on ? m.pcW[base + (size_t)(i + q) * 64] : 0.0;
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) if (i + q < n) r[q] = A[id[q]];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (i + q < n) {
                    acc.rho += w[q] * r[q].rho; acc.ux += w[q] * r[q].ux; acc.uy += w[q] * r[q].uy;
                    acc.uz += w[q] * r[q].uz; acc.p += w[q] * r[q].p; acc.e += w[q] * r[q].e;
                }
            }
        }
    }
    P[p] = acc;
}

// QGDRhoEqn / QGDUEqn / QGDEEqn + thermo + QGD coefficients of one cell from the sum of its net face fluxes (shared by the cell kernel and
// the fused face + cell kernel)
__device__ __forceinline__ void advanceCell(const CaseView& c, const GasModel& gm, const int ci, const RecA& A, const double rEold,
                                            const double Vc, const double hq, const double (&sum)[5], RecA& An, RecB& Bn, double& rEnew) {
    const double dtV = c.dt[0] / Vc;
    // QGDRhoEqn / QGDUEqn / QGDEEqn: explicit Euler on rho, rhoU, rhoE
    const double rho = A.rho - dtV * sum[0];
    // rhoU is not stored: it equals rho*U up to rounding by the re-solve identity below, so its increment is taken
    // directly (rhoU_new - rhoU_old = -dtV*sum); rhoE is an independent field (the explicit energy re-solve as listed does
    // not keep rhoE = rho*(e + |U|^2/2)), 8 B per cell
    rEnew = rEold - dtV * sum[4];
    // solve(fvm::ddt(rho,U) - fvc::ddt(rhoU)) [QGDUEqn_8H L79-86]
    An.rho = rho;
    An.ux = (A.rho * A.ux + (-(dtV * sum[1]))) / rho;
    An.uy = (A.rho * A.uy + (-(dtV * sum[2]))) / rho;
    An.uz = (A.rho * A.uz + (-(dtV * sum[3]))) / rho;
    // solve(fvm::ddt(rho,e) - fvc::ddt(rhoE)) [QGDEEqn_8H L67-72], as written in the listing
    An.e = gm.consistentEnergy ? rEnew / rho - 0.5 * (An.ux * An.ux + An.uy * An.uy + An.uz * An.uz)   // e of [QGDEEqn_8H L49] kept
                               : (A.rho * A.e + (rEnew - rEold)) / rho;
    // thermo.correct(): eConst + perfectGas [hePsiQGDThermo_8C L48-64, L123-124]
    const double T = An.e / gm.Cv;
    const double psi = 1.0 / (gm.R * T);
    const double cs = sqrt(gm.gamma / psi);
    // constScPrModel1 [L103-115]: the pressure seen here is still the old one [QGDFoam_8C L149-154]
    const double aq = c.aQ ? c.aQ[ci] : gm.alphaQGD, scq = c.sc ? c.sc[ci] : gm.ScQGD;
    const double tauQGD = aq * hq / cs;
    Bn.muQGD = A.p * scq * tauQGD;
    Bn.c = cs;
    Bn.aOc = aq / cs;
    An.p = rho / psi;                  // [QGDFoam_8C L152-154]
    Bn.H = (rEnew + An.p) / rho;       // H = (rhoE + p)/rho [QGDFoam/updateFields.H L71]
}

// ---------------------------------------------------------------------------
// cell update: gather of the net face fluxes in ascending face order (the
// summation order of fvc::surfaceIntegrate), explicit Euler, thermo, QGD coeffs
// ---------------------------------------------------------------------------
#ifndef QGD_C_WAVES_MIN
#define QGD_C_WAVES_MIN 3
#endif
#ifndef QGD_C_WAVES_MAX
#define QGD_C_WAVES_MAX 4
#endif
template <int CB>
__global__ __launch_bounds__(CB) __attribute__((amdgpu_waves_per_eu(QGD_C_WAVES_MIN, QGD_C_WAVES_MAX)))
void cellUpdateKernel(const MeshView m, const CaseView c, const GasModel gm, const int mode,
                      const int32_t* __restrict__ list, const int nList, const int slotBase) {
    // mode 0: every cell but the ghosts; mode 1: the cells of `list` (boundary layer of a shard: its records are what the
    // neighbours wait for); mode 2: ordinary owned cells only (the rest, while the exchange is in flight)
    const int tile = (mode == 1) ? (int)blockIdx.x : xcdTile((int)gridDim.x, m.xcdRun * (QGD_BLOCK / CB));
    const int idx = tile * CB + threadIdx.x;
    int ci = -1;
    if (mode == 1) { if (idx < nList) ci = list[idx]; }
    else if (idx < m.nC) {
        const int role = m.ghost ? m.ghost[idx] : 0;
        if (role != 1 && !(mode == 2 && role == 2)) ci = idx;
    }
    double rmin = 1e300, emin = 1e300;
    if (ci >= 0) {
        double sum[5] = {0, 0, 0, 0, 0};
        const int n = m.cfCount[ci];
        const size_t base = (size_t)m.cfSlice[ci >> 6] * 64 + (ci & 63);
        const size_t nF = (size_t)m.nF;
        // the cell's own records stream in while the flux gather is in flight
        const RecA A = c.A[ci];
        const double rEold = c.rE[ci];
        const double Vc = m.V[ci], hq = m.hQGD[ci];
        // six faces per pass, every flux load of the pass in flight before the ordered sum, ascending face label.
        // A wavefront of hexahedra only takes the unpredicated pass; one with other cells (tetrahedra, prisms, split
        // cells) takes the predicated loop as a whole, so no wavefront ever runs both.
        if (__ballot(n != 6) == 0) {
            int it[6];
            double fl[6][5];
#pragma unroll
            for (int q = 0; q < 6; ++q) it[q] = m.cfPos[base + (size_t)q * 64];
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                const size_t f = (size_t)(it[q] >= 0 ? it[q] : ~it[q]);
#pragma unroll
                for (int k = 0; k < 5; ++k) fl[q][k] = c.flux[k * nF + f];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 6; ++q)
#pragma unroll
                for (int k = 0; k < 5; ++k) sum[k] = (it[q] >= 0) ? sum[k] + fl[q][k] : sum[k] - fl[q][k];
        } else {
            for (int i = 0; i < n; i += 6) {
                int it[6];
                double fl[6][5];
#pragma unroll
                for (int q = 0; q < 6; ++q) it[q] = (i + q < n) ? m.cfPos[base + (size_t)(i + q) * 64] : 0;
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    const size_t f = (size_t)(it[q] >= 0 ? it[q] : ~it[q]);
#pragma unroll
                    for (int k = 0; k < 5; ++k) fl[q][k] = (i + q < n) ? c.flux[k * nF + f] : 0.0;
                }
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    if (i + q < n) {
#pragma unroll
                        for (int k = 0; k < 5; ++k) sum[k] = (it[q] >= 0) ? sum[k] + fl[q][k] : sum[k] - fl[q][k];
                    }
                }
            }
        }
        RecA An;
        RecB Bn;
        double rEnew;
        advanceCell(c, gm, ci, A, rEold, Vc, hq, sum, An, Bn, rEnew);
        const double rho = An.rho;
        c.A[ci] = An;
        c.B[ci] = Bn;
        c.rE[ci] = rEnew;
        // a NaN must not hide behind fmin(): it counts as a lost positivity
        rmin = (rho == rho) ? rho : -1e300;
        emin = (An.e == An.e) ? An.e : -1e300;
    }
    // positivity monitor [QGDFoam_8C L142]: one plain store pair per workgroup, no atomics
    blockMaxMin<CB>(-rmin, emin, c.blkCell + 2 * (size_t)(slotBase + tile), true);
}

// ---------------------------------------------------------------------------
// QGD_FUSED: the explicit step of a BLOCK of cells in one workgroup (qgd_setup.hpp FusedBlocks) -- vertex values, internal faces, cell update.
// The three kernels of the explicit step (P, F, C above) run at what the memory system delivers (profiles/r05_ab_face_four_waves.txt), and
// most of the step's bytes are what they hand each other through HBM: 48 B per vertex written by P and read back by F, 40 B per face written
// by F and read back twice by C.  Here a workgroup (256 threads) takes <= 128 cells that are compact in space -- an 8x4x4 brick on a box --
// and
//   (0) reads its lists (no address depends on a loaded value: they are padded to fixed strides) and, one round trip later, stages in LDS
//       RecA of its own cells, of the cells across its surface and of the edge / corner cells around its vertices (360 on a box), RecB and
//       the centres of the first two groups (288), the coordinates of its vertices (225);
//   (1) thread v forms vertex v: volPointInterpolation's weighted sum over pointCells, in their order (pointInterpRecKernel's arithmetic,
//       out of the staged records); a patch point takes the patch-point kernel's value from the vertex records;
//   (2) every thread computes two of the block's internal faces (464 on a box: the 80 surface faces are computed by the block on the other
//       side as well, +21 % face arithmetic) -- Gauss coefficients with the quadrilateral's three differences formed one after the other,
//       the gradient component by component out of LDS (144-152 VGPRs: three blocks per CU), then gvp3FaceTail -- and keeps the fluxes;
//   (3) once every face is done with the vertex records and coordinates, the fluxes take their place in LDS, and threads 0..127 advance one
//       own cell each: the ordered sum of its faces' fluxes in ascending face label = fvc::surfaceIntegrate's order (patch faces from c.flux,
//       where the patch-face kernel put them), advanceCell, 88 B of new records.
// The block reads the OLD records of its neighbours while other blocks write new ones: the step writes A2 / B2, the host swaps them with A / B.
// Same arithmetic per vertex, face and cell as P + F + C, same orders: bit-identical states (tests/test_fused_step_gpu.py).  Fixed deltaT,
// no debug fields (everything else keeps the three kernels); UPW = `Gauss upwind` fluxes; shards run it too (the boundary-layer blocks first,
// stepAdvance).
// ---------------------------------------------------------------------------
#ifndef QGD_FU_WAVES
#define QGD_FU_WAVES 3
#endif
// wave priority raised from the kernel's start: 0 never, 1 until the loads are out, 2 until the block is staged, 3 until its vertex values are formed
#ifndef QGD_FU_PRIO
#define QGD_FU_PRIO 3
#endif
// IMPL = the implicitDiffusion branch [QGDUEqn.H L36-68, updateFluxes.H L95-111]: the same block forms its vertex values and the QGD fluxes of
// its faces (without the Navier-Stokes / Fourier parts), then -- with fvc::grad(U) of its own and across-a-face cells staged where the vertex
// records were -- tauMC, phiTauMC and the laplacian coefficients of every face (implInternalFace: implFaceTileKernel's expressions), and instead
// of advancing its cells it assembles their rows of the three U systems out of LDS (implCellU: implCellUKernel's): rho, the predictor
// U = rhoU/rho, diagonal, right-hand side, start value.  What the solves and the energy equation read later goes to device memory once, from the
// block that owns the face's owner: the laplacian coefficients, Uf, Sf.(tauMC & Uf), muf, the net energy flux, phiTauMC.  The vertex kernel, the
// QGD face kernel, implFaceTileKernel and implCellUKernel are this one launch, in the explicit step's LDS (three blocks per CU).  Unsharded cases.
// ADJ = Courant-number control [QGDCourantNo.H L36-53, setDeltaT-QGDQHD.H L41-61]: the new deltaT needs every face's Courant number and
// tauQGDf before the first cell may advance, so the block stops after its ordered sums: it leaves max Cof / min tauQGDf of its faces in its
// slot of blkFace (the three kernels' partial slots; faceReduceKernel folds them, deltaTKernel follows) and the five net flux sums of each own
// cell in cellSum; cellFinishKernel advances the cells from there.  Vertex values and face fluxes still never reach device memory.
template <bool SGEO, bool UPW = false, bool IMPL = false, bool ADJ = false>
// (IMPL with `Gauss upwind` fluxes needs a few registers more than three waves per SIMD leave: that instantiation is compiled for two -- no scratch)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((IMPL && UPW) ? 2 : QGD_FU_WAVES, (IMPL && UPW) ? 2 : QGD_FU_WAVES)))
void fusedFaceCellKernel(const MeshView m, const CaseView c, const GasModel gm, const int firstBlock, const ImplView iv,
                         const PatchBCDev* __restrict__ bcs) {
    extern __shared__ v2d tileLds[];
#if QGD_FU_CLOCK   // timing probe (scripts/fused_phase_clock.py): where a block's lifetime goes, in shader-clock ticks, left in its cells' new records
    const uint64_t tk0 = __builtin_readcyclecounter();
    uint64_t tk1 = 0, tk2 = 0, tk3 = 0, tk4 = 0, tk5 = 0;
#endif
#if QGD_FU_PRIO
    __builtin_amdgcn_s_setprio(3);
#endif
    constexpr int NT = 256, KC = 5, KCC = 4, KB2 = 3, KV = 3, KF = 2, KE = 6, KP = 8, KG = 12;
    static_assert(9 * kFusedCapCDev <= KG * NT, "caps");
    // piece loads per thread: RecA of <= 384 staged cells, centres and RecB of the <= 320 own + across-a-face cells, coordinates of <= 256
    // vertices; faces per thread; face entries of a cell / cells of a vertex held in registers
    static_assert(3 * kFusedCapTotDev <= KC * NT && 3 * kFusedCapCDev <= KCC * NT && 2 * kFusedCapCDev <= KB2 * NT && 3 * kFusedCapVDev <= KV * NT &&
                  kFusedCapFDev <= KF * NT && kFusedCapVDev <= NT, "caps");
    const int blk = firstBlock + xcdTile((int)gridDim.x, m.fuXcdRun);
    const int tid = (int)threadIdx.x;
    const int capC = m.fuCapC, capV = m.fuCapV, capF = m.fuCapF, capPE = m.fuCapPE;
    const int32_t* __restrict__ tCells = m.fuCells + (size_t)blk * capC;
    const int32_t* __restrict__ tVerts = m.fuVerts + (size_t)blk * capV;
    const int32_t* __restrict__ tFaceLabel = m.fuFaceLabel + (size_t)blk * capF;
    // (0) everything whose address does not depend on a loaded value: the counts and the template id, the lists (padded to their strides with
    // their last entry, so no count is needed to read them), the labels of this thread's two faces, its vertex's weights
    const int4 hdr = m.fuHdr[blk];
    const int4 hdr2 = m.fuHdr2[blk];
    const int nTot = hdr2.x;
    int fl[KF];
#pragma unroll
    for (int j = 0; j < KF; ++j) fl[j] = tFaceLabel[min(tid + j * NT, capF - 1)];
    int idC[KC], idB[KB2], idV[KV];
#pragma unroll
    for (int k = 0; k < KC; ++k) {
        const int q = tid + k * NT, r = (q * 43691) >> 17;
        idC[k] = tCells[min(r, capC - 1)] * 3 + (q - 3 * r);
    }
#pragma unroll
    for (int k = 0; k < KB2; ++k) {
        const int q = tid + k * NT, r = q >> 1;
        idB[k] = tCells[min(r, capC - 1)] * 2 + (q & 1);
    }
#pragma unroll
    for (int k = 0; k < KV; ++k) {
        const int q = tid + k * NT, r = (q * 43691) >> 17;
        idV[k] = tVerts[min(r, capV - 1)] * 3 + (q - 3 * r);
    }
    const int ci = tCells[min(tid & 127, capC - 1)];   // (threads beyond the block's cells repeat its last label: loads stay inside the lists)
    const int nEraw = (int)m.fuNEntry[(size_t)blk * 128 + (tid & 127)];
    // this thread's vertex (thread v forms vertex v of the block's list)
    const int vt = min(tid, capV - 1);
    const int myVert = tVerts[vt];
    const int nPc = (int)m.fuVCount[(size_t)blk * capV + vt];
    const double* __restrict__ vW = m.fuVW + (size_t)blk * capPE * capV + vt;
    double pcW[KP];
#pragma unroll
    for (int i = 0; i < KP; ++i) pcW[i] = vW[(size_t)min(i, capPE - 1) * capV];
    // (1) one round trip later: the records, piece by piece; the faces' streams; the cell's own scalars; a patch point's record -- and the
    // block's local topology out of its TEMPLATE (hdr2.y; qgd_setup.hpp FusedBlocks: the interior bricks of a structured region share a few
    // hundred templates, which stay in L2): the positions of this thread's two faces' cells and vertices in the staged lists, its cell's face
    // entries, its vertex's cell positions.  None of it is needed before the records are staged, so the template costs no round trip.
#if QGD_FU_CLOCK
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (probe only: the two rounds one after the other, to time them apart)
    tk1 = __builtin_readcyclecounter();
#endif
    const size_t tpl = (size_t)hdr2.y;
    struct Pos3 { uint32_t c, va, vb; };
    const Pos3* __restrict__ tFacePos = reinterpret_cast<const Pos3*>(m.fuFacePos) + tpl * capF;
    const int32_t* __restrict__ ent = m.fuEntry + tpl * m.fuCapE * 128 + (tid & 127);
    const uint16_t* __restrict__ vPos = m.fuVPos + tpl * capPE * capV + vt;
    Pos3 fp[KF];
#pragma unroll
    for (int j = 0; j < KF; ++j) fp[j] = tFacePos[min(tid + j * NT, capF - 1)];
    int e6[KE];
#pragma unroll
    for (int i = 0; i < KE; ++i) e6[i] = ent[(size_t)min(i, m.fuCapE - 1) * 128];
    int pcPos[KP];
#pragma unroll
    for (int i = 0; i < KP; ++i) pcPos[i] = (int)vPos[(size_t)min(i, capPE - 1) * capV];
    const v2d* __restrict__ gA = reinterpret_cast<const v2d*>(c.A);
    const v2d* __restrict__ gB = reinterpret_cast<const v2d*>(c.B);
    const v2d* __restrict__ gP = reinterpret_cast<const v2d*>(c.P);
    v2d dA[KC], dB[KB2];
    double dC[KCC], dX[KV];
#pragma unroll
    for (int k = 0; k < KC; ++k) dA[k] = gA[idC[k]];
#pragma unroll
    for (int k = 0; k < KCC; ++k) dC[k] = m.Cc[idC[k]];
#pragma unroll
    for (int k = 0; k < KB2; ++k) dB[k] = gB[idB[k]];
#pragma unroll
    for (int k = 0; k < KV; ++k) dX[k] = m.X[idV[k]];
    double fw[KF], fh[KF];
    int fk[KF];
#pragma unroll
    for (int j = 0; j < KF; ++j) { fw[j] = ldStream(m.w + fl[j]); fh[j] = ldStream(m.hf + fl[j]); fk[j] = m.fkind[fl[j]]; }
    const double rEold = c.rE[ci], Vc = m.V[ci], hq = m.hQGD[ci];
    v2d dPt[3];
    dPt[0] = dPt[1] = dPt[2] = v2d{0.0, 0.0};
    if (nPc == 0) {   // a patch point: the patch-point kernel has put its value into the vertex records
#pragma unroll
        for (int k = 0; k < 3; ++k) dPt[k] = gP[(size_t)myVert * 3 + k];
    }
    __builtin_amdgcn_sched_barrier(0);
#if QGD_FU_PRIO == 1
    __builtin_amdgcn_s_setprio(0);
#endif
    const int nOwn = hdr.x, nUc = hdr.y, nUv = hdr.z, nFc = hdr.w;
    // LDS, laid out by THIS block's counts (the launch reserves what the block that needs most takes): RecA of every staged cell and RecB of
    // the own + across-a-face cells stay to the end (an own cell's old record is read by its update); the vertex records -- formed HERE, from
    // the staged cells -- and all coordinates are dead once every face has its fluxes in registers, and the fluxes take their place.  The
    // parked face entries of the own cells sit at a fixed place behind all that.
    v2d* const sA = tileLds;                     // 3 nTot pieces
    v2d* const sB = sA + 3 * nTot;               // 2 nAll
    v2d* const sP = sB + 2 * hdr.y;              // 3 nV: vertex RecA
    double* const sX = reinterpret_cast<double*>(sP + 3 * hdr.z);   // 3 nV: vertex coordinates
    double* const sC = sX + 3 * hdr.z;           // 3 nAll: cell centres
    double* const sF = reinterpret_cast<double*>(sP);   // 5 nF: net fluxes, plane by plane (after the third barrier)
    double* const sG = reinterpret_cast<double*>(sP);   // IMPL: 9 nAll: fvc::grad(U) of the own + across-a-face cells, between the two rounds of flux planes
    int* const sE = reinterpret_cast<int*>(reinterpret_cast<double*>(tileLds) + (IMPL ? m.fuLdsCellImpl : m.fuLdsCell));   // (IMPL: the same figure unless a block's gradients need more than its vertex region)   // 6 x 128: an own cell's first six face entries, parked until its update
    const int strideF = hdr.w;
    if (tid < 128) {
#pragma unroll
        for (int i = 0; i < KE; ++i) sE[i * 128 + tid] = e6[i];
    }
#pragma unroll
    for (int k = 0; k < KC; ++k) { const int q = tid + k * NT; if (q < 3 * nTot) sA[q] = dA[k]; }
#pragma unroll
    for (int k = 0; k < KCC; ++k) { const int q = tid + k * NT; if (q < 3 * nUc) sC[q] = dC[k]; }
#pragma unroll
    for (int k = 0; k < KB2; ++k) { const int q = tid + k * NT; if (q < 2 * nUc) sB[q] = dB[k]; }
#pragma unroll
    for (int k = 0; k < KV; ++k) { const int q = tid + k * NT; if (q < 3 * nUv) sX[q] = dX[k]; }
#if QGD_FU_CLOCK
    tk2 = __builtin_readcyclecounter();
#endif
    __syncthreads();
#if QGD_FU_PRIO == 2
    __builtin_amdgcn_s_setprio(0);
#endif
    // (1b) the vertex values [volPointInterpolation: inverse-distance weights over pointCells, in their order -- pointInterpRecKernel's
    // arithmetic, out of the staged cell records]
    if (tid < nUv) {
        if (__ballot(nPc != KP) == 0) {
            // a wavefront of interior vertices of hexahedra: eight cells each, no predicates
            RecA acc = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int i = 0; i < KP; ++i) {
                const RecA r = *reinterpret_cast<const RecA*>(sA + 3 * pcPos[i]);
                acc.rho += pcW[i] * r.rho; acc.ux += pcW[i] * r.ux; acc.uy += pcW[i] * r.uy;
                acc.uz += pcW[i] * r.uz; acc.p += pcW[i] * r.p; acc.e += pcW[i] * r.e;
            }
            *reinterpret_cast<RecA*>(sP + 3 * tid) = acc;
        } else if (nPc > 0) {
            RecA acc = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
            for (int i = 0; i < nPc; ++i) {
                int pos = 0;
                double w = 0.0;
                if (i < KP) {
#pragma unroll
                    for (int q = 0; q < KP; ++q) { pos = (i == q) ? pcPos[q] : pos; w = (i == q) ? pcW[q] : w; }
                } else {
                    pos = (int)vPos[(size_t)i * capV];
                    w = vW[(size_t)i * capV];
                }
                const RecA r = *reinterpret_cast<const RecA*>(sA + 3 * pos);
                acc.rho += w * r.rho; acc.ux += w * r.ux; acc.uy += w * r.uy;
                acc.uz += w * r.uz; acc.p += w * r.p; acc.e += w * r.e;
            }
            *reinterpret_cast<RecA*>(sP + 3 * tid) = acc;
        } else {
#pragma unroll
            for (int k = 0; k < 3; ++k) sP[3 * tid + k] = dPt[k];
        }
    }
#if QGD_FU_CLOCK
    tk3 = __builtin_readcyclecounter();
#endif
    __syncthreads();
#if QGD_FU_PRIO == 3
    __builtin_amdgcn_s_setprio(0);
#endif
    // (2) the faces: fluxes into registers
    auto l3 = [](const double* p, int i) { return make_double4(p[3 * i], p[3 * i + 1], p[3 * i + 2], 0.0); };
    double out[KF][5];
    double cofMax = -1e300, tauMinAll = 1e300;   // ADJ: this thread's faces
#pragma unroll
    for (int j = 0; j < KF; ++j) {
        const int lf = tid + j * NT;
        if (lf < nFc) {
            const int f = fl[j], kind = fk[j];
            const unsigned lc = fp[j].c, lva = fp[j].va, lvb = fp[j].vb;
            const int lo = (int)(lc & 0xffffu), ln = (int)(lc >> 16);
            const int v0 = (int)(lva & 0xffffu), v1 = (int)(lva >> 16), v2 = (int)(lvb & 0xffffu), v3 = (int)(lvb >> 16);
            double S[3] = {0.0, 0.0, 0.0};
            double msO = 1.0, dnO = 0.0;
            if (!SGEO || kind != 0) { S[0] = ldStream(m.Sx + f); S[1] = ldStream(m.Sy + f); S[2] = ldStream(m.Sz + f); }
            if (m.hasOther) { msO = m.magSf[f]; dnO = m.dn[f]; }
            // Gauss coefficients; the quadrilateral's three differences one after the other (six coordinates live, not eighteen)
            double coef[12], rVc;
            // (the branch below asks a copy of `kind` the compiler cannot see through: with the same condition here and in front of the
            // gradient it threads the two, compiles the flux algebra once per kind, and the two copies contract their multiply-adds
            // differently -- 1e-16 away from the two-kernel step)
            int kindC = kind;
            asm volatile("" : "+v"(kindC));
            if (kindC == 0) {
                double NO[3], d24[3], d31[3];
                { const double4 cO = l3(sC, lo), cN = l3(sC, ln); NO[0] = cN.x - cO.x; NO[1] = cN.y - cO.y; NO[2] = cN.z - cO.z; }
                __builtin_amdgcn_sched_barrier(0);
                { const double4 x1 = l3(sX, v1), x3 = l3(sX, v3); d24[0] = x1.x - x3.x; d24[1] = x1.y - x3.y; d24[2] = x1.z - x3.z; }
                __builtin_amdgcn_sched_barrier(0);
                { const double4 x2 = l3(sX, v2), x0 = l3(sX, v0); d31[0] = x2.x - x0.x; d31[1] = x2.y - x0.y; d31[2] = x2.z - x0.z; }
                gvp3QuadCoefs(NO, d24, d31, coef, rVc);
                if (SGEO) { S[0] = 0.5 * coef[6]; S[1] = 0.5 * coef[7]; S[2] = 0.5 * coef[8]; }
            } else {
                const double4 cO = l3(sC, lo), cN = l3(sC, ln);
                const double4 x0 = l3(sX, v0), x1 = l3(sX, v1), x2 = l3(sX, v2), x3 = l3(sX, v3);
                gvp3Coefs(kind, cO, cN, x0, x1, x2, x3, coef, rVc);
            }
            __builtin_amdgcn_sched_barrier(0);
            // the six-component gradient, one component of the six records at a time out of LDS (gvp3FaceBody holds the records in registers)
            const double* const vO = reinterpret_cast<const double*>(sA + 3 * lo);
            const double* const vN = reinterpret_cast<const double*>(sA + 3 * ln);
            const double* const r0 = reinterpret_cast<const double*>(sP + 3 * v0);
            const double* const r1 = reinterpret_cast<const double*>(sP + 3 * v1);
            const double* const r2 = reinterpret_cast<const double*>(sP + 3 * v2);
            const double* const r3 = reinterpret_cast<const double*>(sP + 3 * v3);
            double cof = -1e300, tauMin = 1e300;
            double g[18];
            if (kind == 0) {
#pragma unroll
                for (int k = 0; k < 6; ++k) gvp3QuadGradK(coef, rVc, vO[k], vN[k], r0[k], r1[k], r2[k], r3[k], g, k);
            } else if (kind == 1) {
                const double* t = coef;
                const double rV = rVc;
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    const double a0 = t[4 * d], a1 = t[4 * d + 1], a2 = t[4 * d + 2], a3 = t[4 * d + 3];
#pragma unroll
                    for (int k = 0; k < 6; ++k) {
                        double sg = vN[k] * a3;
                        sg += vO[k] * (-a3);
                        sg += r0[k] * a0;
                        sg += r1[k] * a1;
                        sg += r2[k] * a2;
                        g[d * 6 + k] = sg * rV;
                    }
                }
                double dg[3];
#pragma unroll
                for (int q = 0; q < 3; ++q) dg[q] = g[q * 6 + 1 + q];
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int q = 0; q < 3; ++q) g[i * 6 + 1 + q] = dg[q];
            } else if (kind == 2) {
                const double nx = S[0] / msO, ny = S[1] / msO, nz = S[2] / msO;
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    const double sn = dnO * (vN[k] - vO[k]);
                    g[0 * 6 + k] = nx * sn;
                    g[1 * 6 + k] = ny * sn;
                    g[2 * 6 + k] = nz * sn;
                }
            } else {
#pragma unroll
                for (int i = 0; i < 18; ++i) g[i] = 0.0;
            }
            __builtin_amdgcn_sched_barrier(0);
            const RecA Ao = *reinterpret_cast<const RecA*>(sA + 3 * lo), An = *reinterpret_cast<const RecA*>(sA + 3 * ln);
            const RecB Bo = *reinterpret_cast<const RecB*>(sB + 2 * lo), Bn = *reinterpret_cast<const RecB*>(sB + 2 * ln);
            gvp3FaceTail<false, UPW>(m, c, gm, f, fw[j], fh[j], S, Ao, An, Bo, Bn, g, ADJ ? 1 : 0, cof, tauMin, &out[j][0], (size_t)1);
            if constexpr (ADJ) { cofMax = fmax(cofMax, cof); tauMinAll = fmin(tauMinAll, tauMin); }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#if QGD_FU_CLOCK
    tk4 = __builtin_readcyclecounter();
#endif
    __syncthreads();   // every face has read its vertex records and coordinates
    if constexpr (IMPL) {
        // The implicitDiffusion branch keeps the explicit step's LDS (three blocks per CU) by using the dead vertex region three times:
        // (a) the four mass / momentum flux planes -> the own cells' ordered sums; (b) fvc::grad(U) of the own + across-a-face cells ->
        // tauMC, phiTauMC, the laplacian coefficients of every face; (c) the phiTauMC / coefficient planes -> the cells' second sums and
        // their rows of the U systems.  The gradients and the faces' streamed data are requested before (a) and arrive behind it.
        const size_t nF = (size_t)m.nF;
        double dG[KG], fgsd[KF], fS[KF][3];
        int fps[KF];
#pragma unroll
        for (int k = 0; k < KG; ++k) {
            const int q = tid + k * NT, r = q / 9;
            dG[k] = iv.gUc[(size_t)tCells[min(r, capC - 1)] * 9 + (q - 9 * r)];
        }
#pragma unroll
        for (int j = 0; j < KF; ++j) {
            fps[j] = ldStream(m.fpos + fl[j]);
            fgsd[j] = ldStream(m.magSf + fl[j]) * ldStream(m.dn + fl[j]);   // |Sf| * nonOrthDeltaCoeffs
            // the STREAMED Sf: implFaceTileKernel multiplies with that one, the QGD fluxes above with the Sf rebuilt from the vertices
            fS[j][0] = ldStream(m.Sx + fl[j]); fS[j][1] = ldStream(m.Sy + fl[j]); fS[j][2] = ldStream(m.Sz + fl[j]);
        }
#pragma unroll
        for (int j = 0; j < KF; ++j) {
            const int lf = tid + j * NT;
            if (lf < nFc) {
#pragma unroll
                for (int k = 0; k < 4; ++k) sF[k * strideF + lf] = out[j][k];
            }
        }
        __syncthreads();
        double sum[4] = {0, 0, 0, 0}, dTau[3] = {0, 0, 0}, diagBase = 0;
        const int nE = nEraw;
        if (tid < nOwn) {
            for (int i = 0; i < nE; ++i) {
                const int e = (i < KE) ? sE[i * 128 + tid] : ent[(size_t)i * 128];
                double fx[4];
                if (e >= 0) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) fx[k] = sF[k * strideF + (e >> 1)];
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) fx[k] = c.flux[k * nF + (size_t)(~e)];
                }
                const bool plus = e < 0 || !(e & 1);
#pragma unroll
                for (int k = 0; k < 4; ++k) sum[k] = plus ? sum[k] + fx[k] : sum[k] - fx[k];
            }
        }
        __syncthreads();   // the flux planes are summed: the gradients take their place
#pragma unroll
        for (int k = 0; k < KG; ++k) { const int q = tid + k * NT; if (q < 9 * nUc) sG[q] = dG[k]; }
        __syncthreads();
        double oxi[KF][4];   // phiTauMC (3) and the laplacian coefficient of the U systems
#pragma unroll
        for (int j = 0; j < KF; ++j) {
            const int lf = tid + j * NT;
            if (lf < nFc) {
                const int f = fl[j];
                const int lo = (int)(fp[j].c & 0xffffu), ln = (int)(fp[j].c >> 16);
                const RecA Ao = *reinterpret_cast<const RecA*>(sA + 3 * lo), An = *reinterpret_cast<const RecA*>(sA + 3 * ln);
                const double muQo = reinterpret_cast<const RecB*>(sB + 2 * lo)->muQGD, muQn = reinterpret_cast<const RecB*>(sB + 2 * ln)->muQGD;
                const double uo[3] = {Ao.ux, Ao.uy, Ao.uz}, un[3] = {An.ux, An.uy, An.uz};
                ImplFaceOut r;
                implInternalFace(gm, fw[j], muQo, muQn, uo, un, sG + 9 * lo, sG + 9 * ln, fS[j], fgsd[j], r);
                oxi[j][0] = r.phiTau[0]; oxi[j][1] = r.phiTau[1]; oxi[j][2] = r.phiTau[2]; oxi[j][3] = r.aU;
                if (lo < nOwn) {   // the block that owns the face's owner writes what the solves and the energy equation read later
                    const size_t pos = (size_t)fps[j];
                    c.flux[4 * nF + pos] = out[j][4];
#pragma unroll
                    for (int k = 0; k < 3; ++k) { iv.phiTau[(size_t)k * nF + pos] = r.phiTau[k]; iv.UfS[(size_t)k * nF + f] = r.Uf[k]; }
                    iv.sTau[f] = r.sTau;
                    iv.mufS[f] = r.muf;
                    iv.aU[pos] = r.aU;
                    iv.aE[pos] = r.aE;
                }
            }
        }
        __syncthreads();   // every face has read its two gradients: the second round of planes takes their place
#pragma unroll
        for (int j = 0; j < KF; ++j) {
            const int lf = tid + j * NT;
            if (lf < nFc) {
#pragma unroll
                for (int k = 0; k < 4; ++k) sF[k * strideF + lf] = oxi[j][k];
            }
        }
        __syncthreads();
        // (3') the rows of the U systems of the block's own cells: implCellUKernel's ordered sums and arithmetic
        if (tid < nOwn) {
            for (int i = 0; i < nE; ++i) {
                const int e = (i < KE) ? sE[i * 128 + tid] : ent[(size_t)i * 128];
                double tx[3];
                if (e >= 0) {
#pragma unroll
                    for (int k = 0; k < 3; ++k) tx[k] = sF[k * strideF + (e >> 1)];
                } else {
#pragma unroll
                    for (int k = 0; k < 3; ++k) tx[k] = iv.phiTau[k * nF + (size_t)(~e)];
                }
                const bool plus = e < 0 || !(e & 1);
#pragma unroll
                for (int k = 0; k < 3; ++k) dTau[k] = plus ? dTau[k] + tx[k] : dTau[k] - tx[k];
                if (e >= 0) diagBase += sF[3 * strideF + (e >> 1)];
            }
            const RecA A = *reinterpret_cast<const RecA*>(sA + 3 * tid);
            implCellU(m, c, iv, bcs, ci, A, Vc, sum, dTau, diagBase, nE, [&](int i) {
                const int e = (i < KE) ? sE[i * 128 + tid] : ent[(size_t)i * 128];
                return e < 0 ? ~e : -1;   // a patch face's label; internal faces carry no patch coefficient
            });
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < KF; ++j) {
        const int lf = tid + j * NT;
        if (lf < nFc) {
#pragma unroll
            for (int k = 0; k < 5; ++k) sF[k * strideF + lf] = out[j][k];
        }
    }
    __syncthreads();
#if QGD_FU_CLOCK
    tk5 = __builtin_readcyclecounter();
#endif
    // (3) the block's own cells out of LDS
    double rmin = 1e300, emin = 1e300;
    if (tid < nOwn) {
        const size_t nF = (size_t)m.nF;
        double sum[5] = {0, 0, 0, 0, 0};
        const int nE = nEraw;
        int eq[KE];
#pragma unroll
        for (int i = 0; i < KE; ++i) eq[i] = sE[i * 128 + tid];
        bool inner = nE == KE;
#pragma unroll
        for (int i = 0; i < KE; ++i) inner = inner && eq[i] >= 0;
        if (__ballot(!inner) == 0) {
            // a wavefront of hexahedra away from the patches (cellUpdateKernel's first branch): the thirty flux terms in flight out of LDS before
            // the ordered sums -- the loop below waits for every face's terms before it asks for the next face's
            double x[KE][5];
#pragma unroll
            for (int i = 0; i < KE; ++i)
#pragma unroll
                for (int k = 0; k < 5; ++k) x[i][k] = sF[k * strideF + (eq[i] >> 1)];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < KE; ++i)
#pragma unroll
                for (int k = 0; k < 5; ++k) sum[k] = (eq[i] & 1) ? sum[k] - x[i][k] : sum[k] + x[i][k];
        } else {
            for (int i = 0; i < nE; ++i) {
                const int e = (i < KE) ? sE[i * 128 + tid] : ent[(size_t)i * 128];
                double fl[5];
                if (e >= 0) {
#pragma unroll
                    for (int k = 0; k < 5; ++k) fl[k] = sF[k * strideF + (e >> 1)];
                } else {
#pragma unroll
                    for (int k = 0; k < 5; ++k) fl[k] = c.flux[k * nF + (size_t)(~e)];
                }
#pragma unroll
                for (int k = 0; k < 5; ++k) sum[k] = (e < 0 || !(e & 1)) ? sum[k] + fl[k] : sum[k] - fl[k];
            }
        }
        if constexpr (ADJ) {
#pragma unroll
            for (int k = 0; k < 5; ++k) c.cellSum[(size_t)k * m.nC + ci] = sum[k];
        } else {
            const RecA A = *reinterpret_cast<const RecA*>(sA + 3 * tid);
            RecA An;
            RecB Bn;
            double rEnew;
            advanceCell(c, gm, ci, A, rEold, Vc, hq, sum, An, Bn, rEnew);
#if QGD_FU_CLOCK
            {
                asm volatile("" ::"v"(An.rho), "v"(An.e), "v"(Bn.muQGD), "v"(rEnew));
                const uint64_t tk6 = __builtin_readcyclecounter();
                // lists (round 0) | records (round 1) + staging | barrier + vertex values | barrier + faces | barriers + flux planes | sums + advanceCell
                An.rho = (double)(tk1 - tk0); An.ux = (double)(tk2 - tk1); An.uy = (double)(tk3 - tk2); An.uz = (double)(tk4 - tk3);
                An.p = (double)(tk5 - tk4); An.e = (double)(tk6 - tk5);
            }
#endif
            c.A2[ci] = An;
            c.B2[ci] = Bn;
            c.rE[ci] = rEnew;
            rmin = (An.rho == An.rho) ? An.rho : -1e300;
            emin = (An.e == An.e) ? An.e : -1e300;
        }
    }
    if constexpr (ADJ) blockMaxMin<NT>(cofMax, tauMinAll, c.blkFace + 2 * (size_t)blk, false);
    else blockMaxMin<NT>(-rmin, emin, c.blkCell + 2 * (size_t)blk, true);
}

// Courant-number control with the fused step: the cells advance from the flux sums the blocks left in cellSum, once deltaT is known
// (advanceCell: the cell kernel's arithmetic; mode / list as in cellUpdateKernel)
template <int CB>
__global__ __launch_bounds__(CB) void cellFinishKernel(const MeshView m, const CaseView c, const GasModel gm, const int mode,
                                                      const int32_t* __restrict__ list, const int nList, const int slotBase) {
    const int tile = (mode == 1) ? (int)blockIdx.x : xcdTile((int)gridDim.x, m.xcdRun * (QGD_BLOCK / CB));
    const int idx = tile * CB + threadIdx.x;
    int ci = -1;
    if (mode == 1) { if (idx < nList) ci = list[idx]; }
    else if (idx < m.nC) {
        const int role = m.ghost ? m.ghost[idx] : 0;
        if (role != 1 && !(mode == 2 && role == 2)) ci = idx;
    }
    double rmin = 1e300, emin = 1e300;
    if (ci >= 0) {
        double sum[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) sum[k] = c.cellSum[(size_t)k * m.nC + ci];
        const RecA A = c.A[ci];
        RecA An;
        RecB Bn;
        double rEnew;
        advanceCell(c, gm, ci, A, c.rE[ci], m.V[ci], m.hQGD[ci], sum, An, Bn, rEnew);
        c.A[ci] = An;
        c.B[ci] = Bn;
        c.rE[ci] = rEnew;
        rmin = (An.rho == An.rho) ? An.rho : -1e300;
        emin = (An.e == An.e) ? An.e : -1e300;
    }
    blockMaxMin<CB>(-rmin, emin, c.blkCell + 2 * (size_t)(slotBase + tile), true);
}

// createFields.H for the cells [QGDFoam_2createFields_8H L3-109]
__global__ __launch_bounds__(QGD_BLOCK) void cellInitKernel(const MeshView m, const CaseView c, const GasModel gm,
                                                           const double* __restrict__ U, const double* __restrict__ T,
                                                           const double* __restrict__ p) {
    const int ci = blockIdx.x * QGD_BLOCK + threadIdx.x;
    if (ci >= m.nC) return;
    const double Tc = T[ci], pc = p[ci];
    const double e = gm.Cv * Tc;
    const double psi = 1.0 / (gm.R * Tc);
    const double cs = sqrt(gm.gamma / psi);
    const double rho = pc * psi;
    RecA A;
    A.rho = rho; A.ux = U[3 * (size_t)ci]; A.uy = U[3 * (size_t)ci + 1]; A.uz = U[3 * (size_t)ci + 2]; A.p = pc; A.e = e;
    const double rE = rho * e + rho * 0.5 * (A.ux * A.ux + A.uy * A.uy + A.uz * A.uz);
    RecB B;
    const double aq = c.aQ ? c.aQ[ci] : gm.alphaQGD, scq = c.sc ? c.sc[ci] : gm.ScQGD;
    const double tauQGD = aq * m.hQGD[ci] / cs;
    B.muQGD = pc * scq * tauQGD;
    B.c = cs;
    B.aOc = aq / cs;
    B.H = (rE + pc) / rho;
    c.A[ci] = A; c.B[ci] = B; c.rE[ci] = rE;
}

// Boundary-condition refresh of every patch face, in the order the loop body
// applies them: U, e, thermo patch values, QGD coefficients, p (qgdFlux), rho
// [QGDUEqn_8H L51, QGDEEqn_8H L50, hePsiQGDThermo_8C L84-121, QGDFoam_8C L155-156]
__global__ __launch_bounds__(QGD_BLOCK) void boundaryUpdateKernel(const MeshView m, const CaseView c, const GasModel gm,
                                                                 const PatchBCDev* __restrict__ bcs, const int init,
                                                                 const int phiwRegistered, const int mode,
                                                                 const int32_t* __restrict__ list, const int nList) {
    // modes as in cellUpdateKernel, by the role of the face's owner cell; init evaluates every face
    const int idx = blockIdx.x * QGD_BLOCK + threadIdx.x;
    int b;
    if (mode == 1) { if (idx >= nList) return; b = list[idx]; }
    else { if (idx >= m.nBF) return; b = idx; }
    const int f = m.nIF + b;
    if (m.fkind[f] == 3) return;
    const PatchBCDev bc = bcs[m.bPatch[b]];
    const int o = m.own[f];
    if (!init && mode != 1 && m.ghost) {
        const int role = m.ghost[o];
        if (role == 1 || (mode == 2 && role == 2)) return;
    }
    const RecA Ao = c.A[o];
    RecA Ab;
    // U
    if (bc.bcU == QGD_BC_FIXEDVALUE) { Ab.ux = bc.vU[0]; Ab.uy = bc.vU[1]; Ab.uz = bc.vU[2]; }
    else if (bc.bcU == QGD_BC_SLIP) {
        double n[3];
        symmNormal(m, bc, f, n);
        const double u[3] = {Ao.ux, Ao.uy, Ao.uz};
        double r[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const double tv = ((i == 0 ? 1.0 : 0.0) - 2.0 * (n[i] * n[0])) * u[0] + ((i == 1 ? 1.0 : 0.0) - 2.0 * (n[i] * n[1])) * u[1] +
                              ((i == 2 ? 1.0 : 0.0) - 2.0 * (n[i] * n[2])) * u[2];
            r[i] = (u[i] + tv) / 2.0;
        }
        Ab.ux = r[0]; Ab.uy = r[1]; Ab.uz = r[2];
    } else { Ab.ux = Ao.ux; Ab.uy = Ao.uy; Ab.uz = Ao.uz; }
    // e / T
    double Tb;
    if (bc.bcT == QGD_BC_FIXEDVALUE) { Tb = bc.vT; Ab.e = gm.Cv * Tb; }
    else { Ab.e = Ao.e; Tb = Ab.e / gm.Cv; }
    const double psi = 1.0 / (gm.R * Tb);
    const double cs = sqrt(gm.gamma / psi);
    RecB Bb;
    const double aq = c.aQb ? c.aQb[b] : gm.alphaQGD, scq = c.scb ? c.scb[b] : gm.ScQGD;
    Bb.c = cs;
    Bb.aOc = aq / cs;
    // pressure seen by constScPrModel1 at thermo.correct(): the patch value before p's BC update
    const double pOld = init ? ((bc.bcP == QGD_BC_FIXEDVALUE) ? bc.vP : Ao.p) : c.bPmid[b];
    const double tauQGD = aq * m.hQGDb[b] / cs;
    Bb.muQGD = pOld * scq * tauQGD;
    // p
    if (bc.bcP == QGD_BC_FIXEDVALUE) Ab.p = bc.vP;
    else if (bc.bcP == QGD_BC_QGDFLUX) {
        double grad = init ? 0.0 : c.bG[b];
        if (phiwRegistered) {
            const double tauf = Bb.aOc * m.hf[f];
            grad = -(c.bPhiw[b] / tauf / m.magSf[f]);  // [qgdFluxFvPatchScalarField_8C L184-192]
        }
        c.bG[b] = grad;
        Ab.p = Ao.p + grad / m.dn[f];
    } else Ab.p = Ao.p;
    Ab.rho = init ? Ab.p * psi : psi * Ab.p;  // thermo.rho() at start-up, psi_b*p_b afterwards [QGDFoam_8C L156]
    // rhoU_b and rhoE_b are assigned inside the U and E equations [QGDUEqn_8H L88-89, QGDEEqn_8H L75-76],
    // i.e. with the patch density of the step before; rho_b itself is refreshed last [QGDFoam_8C L156]
    const double rhoLag = init ? Ab.rho : c.bA[b].rho;
    const double rE = init ? (Ab.rho * Ab.e + Ab.rho * 0.5 * (Ab.ux * Ab.ux + Ab.uy * Ab.uy + Ab.uz * Ab.uz))
                           : (rhoLag * (Ab.e + 0.5 * (Ab.ux * Ab.ux + Ab.uy * Ab.uy + Ab.uz * Ab.uz)));
    Bb.H = (rE + Ab.p) / Ab.rho;
    c.bRhoLag[b] = rhoLag;
    c.bA[b] = Ab;
    c.bB[b] = Bb;
    c.bPmid[b] = Ab.p;
}

// adjustTimeStep: one workgroup folds the per-workgroup partials of the face kernels into red[0] = max Cof and
// red[1] = -min tauQGDf (one MAX all-reduce serves both in a sharded run) [QGDCourantNo_8H L50, setDeltaT-QGDQHD_8H L46]
// (two levels since round 6: ONE workgroup walking the three million partial slots of a 64 M-cell mesh took 6 ms per step -- a third of the
// adjusted step; max and min do not care about the order, so the result is the same bit for bit)
__global__ __launch_bounds__(QGD_BLOCK) void faceReduceStage1Kernel(const CaseView c) {
    double co = -1e300, tmin = 1e300;
    for (int i = blockIdx.x * QGD_BLOCK + threadIdx.x; i < c.nBlkFace; i += gridDim.x * QGD_BLOCK) {
        co = fmax(co, c.blkFace[2 * (size_t)i]);
        tmin = fmin(tmin, c.blkFace[2 * (size_t)i + 1]);
    }
    blockMaxMin(co, tmin, c.blkFace2 + 2 * (size_t)blockIdx.x, false);
}
__global__ __launch_bounds__(QGD_BLOCK) void faceReduceKernel(const CaseView c, const int nPartials) {
    double co = -1e300, tmin = 1e300;
    for (int i = threadIdx.x; i < nPartials; i += QGD_BLOCK) {
        co = fmax(co, c.blkFace2[2 * (size_t)i]);
        tmin = fmin(tmin, c.blkFace2[2 * (size_t)i + 1]);
    }
    blockMaxMin(co, tmin, c.red, false);
    __syncthreads();
    if (threadIdx.x == 0) c.red[1] = -c.red[1];
}
// deltaT on the device [setDeltaT-QGDQHD_8H L41-61]
__global__ void deltaTKernel(const CaseView c, const double maxCo, const double maxDeltaT, const double cTau) {
    const double CoNum = c.red[0], minTau = -c.red[1];
    const double maxDeltaTFact = maxCo / (CoNum + 1e-15);
    const double deltaTFact = fmin(fmin(maxDeltaTFact, 1.0 + 0.1 * maxDeltaTFact), 1.2);
    double maxDeltaT1 = cTau * minTau;
    maxDeltaT1 = fmin(maxDeltaT, maxDeltaT1);
    const double dt = fmin(deltaTFact * c.dt[0], maxDeltaT1);
    c.dt[0] = dt;
    c.dt[1] += dt;  // time
    c.dt[2] = CoNum;
}
__global__ __launch_bounds__(QGD_BLOCK) void resetReductionsKernel(const CaseView c) {
    for (int i = blockIdx.x * QGD_BLOCK + threadIdx.x; i < c.nBlkCell; i += gridDim.x * QGD_BLOCK) {
        c.blkCell[2 * (size_t)i] = -1e300;
        c.blkCell[2 * (size_t)i + 1] = 1e300;
    }
    for (int i = blockIdx.x * QGD_BLOCK + threadIdx.x; i < c.nBlkFace; i += gridDim.x * QGD_BLOCK) {
        c.blkFace[2 * (size_t)i] = -1e300;
        c.blkFace[2 * (size_t)i + 1] = 1e300;
    }
}
// min(rho), min(e) over the per-workgroup monitors -> red[2], red[3]; then restart the monitors
__global__ __launch_bounds__(QGD_BLOCK) void cellMinReduceKernel(const CaseView c) {
    double a = -1e300, b = 1e300;
    for (int i = threadIdx.x; i < c.nBlkCell; i += QGD_BLOCK) {
        a = fmax(a, c.blkCell[2 * (size_t)i]);
        b = fmin(b, c.blkCell[2 * (size_t)i + 1]);
        c.blkCell[2 * (size_t)i] = -1e300;
        c.blkCell[2 * (size_t)i + 1] = 1e300;
    }
    blockMaxMin(a, b, c.red + 2, false);
    __syncthreads();
    if (threadIdx.x == 0) c.red[2] = -c.red[2];
}

// halo message = 8 doubles per listed cell -- the six primitives of SURVEY 8(e) {rho, U, p, e} plus the two derived quantities a
// receiver cannot rebuild from them: H (rhoE is an independent field under the listing's explicit energy re-solve, QGDEEqn_8H L67-72)
// and muQGD (formed with the PREVIOUS step's pressure, QGDFoam_8C L149-154, and the owner's hQGD, a mean over ALL faces of the cell, which
// a ghost cell does not have here); c and alphaQGD/c are recomputed from e by the formulas of the cell update -- then 12 per listed
// boundary face (RecA, RecB, p gradient, lagged patch density: patch records follow their boundary conditions, not these formulas).
// A ghost cell's conserved record is rebuilt from what arrives (it only feeds the ghost's own, discarded, update).
#define QGD_HALO_CELL_DOUBLES 8
__global__ __launch_bounds__(QGD_BLOCK) void haloKernel(const CaseView c, const GasModel gm, const int32_t* __restrict__ cells, const int nCells,
                                                       const int32_t* __restrict__ bfaces, const int nFaces,
                                                       double* __restrict__ buf, const int pack) {
    const int i = blockIdx.x * QGD_BLOCK + threadIdx.x;
    if (i < nCells) {
        const int ci = cells[i];
        double* q = buf + QGD_HALO_CELL_DOUBLES * (size_t)i;
        if (pack) {
            const RecA a = c.A[ci]; const RecB b = c.B[ci];
            q[0] = a.rho; q[1] = a.ux; q[2] = a.uy; q[3] = a.uz; q[4] = a.p; q[5] = a.e; q[6] = b.H; q[7] = b.muQGD;
        } else {
            RecA a; RecB b;
            a.rho = q[0]; a.ux = q[1]; a.uy = q[2]; a.uz = q[3]; a.p = q[4]; a.e = q[5]; b.H = q[6]; b.muQGD = q[7];
            // thermo.correct() of the cell update [hePsiQGDThermo_8C L48-64], the same operations in the same order: the same bits
            const double T = a.e / gm.Cv;
            const double psi = 1.0 / (gm.R * T);
            const double cs = sqrt(gm.gamma / psi);
            b.c = cs;
            b.aOc = (c.aQ ? c.aQ[ci] : gm.alphaQGD) / cs;
            c.A[ci] = a; c.B[ci] = b; c.rE[ci] = b.H * a.rho - a.p;
        }
    } else if (i < nCells + nFaces) {
        const int j = i - nCells;
        const int bi = bfaces[j];
        double* q = buf + QGD_HALO_CELL_DOUBLES * (size_t)nCells + 12 * (size_t)j;
        if (pack) {
            const RecA a = c.bA[bi]; const RecB b = c.bB[bi];
            q[0] = a.rho; q[1] = a.ux; q[2] = a.uy; q[3] = a.uz; q[4] = a.p; q[5] = a.e; q[6] = b.H; q[7] = b.c; q[8] = b.muQGD; q[9] = b.aOc;
            q[10] = c.bG[bi]; q[11] = c.bRhoLag[bi];
        } else {
            RecA a; RecB b;
            a.rho = q[0]; a.ux = q[1]; a.uy = q[2]; a.uz = q[3]; a.p = q[4]; a.e = q[5]; b.H = q[6]; b.c = q[7]; b.muQGD = q[8]; b.aOc = q[9];
            c.bA[bi] = a; c.bB[bi] = b; c.bG[bi] = q[10]; c.bPmid[bi] = a.p; c.bRhoLag[bi] = q[11];
        }
    }
}

// ---------------------------------------------------------------------------
// fvsc operators on plain NC-component fields (the four fvscStencil virtuals)
// ---------------------------------------------------------------------------
template <int ST, int NC, int OP>
__global__ __launch_bounds__(QGD_BLOCK) void fvscOpKernel(const MeshView m, const double* __restrict__ cellF,
                                                         const double* __restrict__ bndF, const double* __restrict__ ptF,
                                                         double* __restrict__ out, const int f0) {
    const int f = f0 + blockIdx.x * QGD_BLOCK + threadIdx.x;
    if (f >= m.nF) return;
    FaceVals<NC> v;
    const int o = m.own[f];
#pragma unroll
    for (int k = 0; k < NC; ++k) v.o[k] = cellF[(size_t)o * NC + k];
    if (f < m.nIF) {
        const int n = m.nei[f];
#pragma unroll
        for (int k = 0; k < NC; ++k) { v.n[k] = cellF[(size_t)n * NC + k]; v.sn[k] = 0.0; }
    } else {
        const int b = f - m.nIF;
        const double dc = m.dn[f];
#pragma unroll
        for (int k = 0; k < NC; ++k) { v.n[k] = bndF[(size_t)b * NC + k]; v.sn[k] = dc * (v.n[k] - v.o[k]); }
        // symmetryPlane / symmetry / wedge patches: a SCALAR patch field there has snGrad = 0 by type (L0: basicSymmetry / wedge
        // specialisations), and the 2-D GaussVolPoint gradient of a vector sees three scalar fields [GaussVolPointBase.C L79-87]
        if ((NC == 1 || (NC == 3 && OP == 0 && ST == ST_GVP2)) && m.bSymm && m.bSymm[b]) {
#pragma unroll
            for (int k = 0; k < NC; ++k) v.sn[k] = 0.0;
        }
    }
    double g[3 * NC];
    // the vector-gradient operator (NC==3, OP==0) carries the interior-triangle pattern
    faceGradient<ST, NC, (NC == 3 && OP == 0) ? 0 : -1>(m, f, v, cellF, ptF, g);
    if (OP == 0) {
#pragma unroll
        for (int k = 0; k < 3 * NC; ++k) out[(size_t)f * 3 * NC + k] = g[k];
    } else {
        constexpr int NO = NC / 3;  // vector -> scalar, tensor -> vector: div_j = sum_i d_i T_ij
#pragma unroll
        for (int j = 0; j < NO; ++j) {
            double d = g[0 * NC + 0 * NO + j] + g[1 * NC + 1 * NO + j] + g[2 * NC + 2 * NO + j];
            // the 2-D tensor divergence fills the two in-plane components only [GaussVolPointBase2D_8C L471-484]
            if (ST == ST_GVP2 && NO == 3 && j == m.ie3) d = 0.0;
            out[(size_t)f * NO + j] = d;
        }
    }
}

// ---------------------------------------------------------------------------
// The drop-in fvsc::grad of the 3-D GaussVolPoint stencil (what an unmodified updateFluxes.H calls four times per step,
// QGDFoam/updateFluxes.H L41-65): loads-first like the fused flux kernel -- labels; cell + vertex values and the geometry the
// Gauss coefficients are rebuilt from, all in flight before the first use -- and the 3*NC results of a 256-face tile go out
// through LDS as one contiguous block (the AoS face field of the reference's surfaceVector/TensorField, 24/72 B per face).
// Internal faces only; patch faces take fvscOpKernel.
// ---------------------------------------------------------------------------
template <int NC>
__global__ __launch_bounds__(QGD_BLOCK) void fvscGradGvp3Kernel(const MeshView m, const double* __restrict__ cellF,
                                                               const double* __restrict__ ptF, double* __restrict__ out) {
    __shared__ double stage[QGD_BLOCK * 3 * NC];
    const int tile = xcdTile((int)gridDim.x, m.xcdRun);
    const int f = tile * QGD_BLOCK + (int)threadIdx.x;
    double g[3 * NC];
#pragma unroll
    for (int i = 0; i < 3 * NC; ++i) g[i] = 0.0;
    if (f < m.nIF) {
        const int o = ldStream(m.own + f), n = ldStream(m.nei + f);
        const int4 vt = m.verts[f];
        const int kind = m.fkind[f];
        const int v3 = vt.w < 0 ? 0 : vt.w;
        FaceVals<NC> v;
        double p0[NC], p1[NC], p2[NC], p3[NC];
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            v.o[k] = cellF[(size_t)o * NC + k]; v.n[k] = cellF[(size_t)n * NC + k]; v.sn[k] = 0.0;
            p0[k] = ptF[(size_t)vt.x * NC + k]; p1[k] = ptF[(size_t)vt.y * NC + k];
            p2[k] = ptF[(size_t)vt.z * NC + k]; p3[k] = ptF[(size_t)v3 * NC + k];
        }
        const double4 cO = ld3(m.Cc, o), cN = ld3(m.Cc, n);
        const double4 x0 = ld3(m.X, vt.x), x1 = ld3(m.X, vt.y), x2 = ld3(m.X, vt.z), x3 = ld3(m.X, v3);
        __builtin_amdgcn_sched_barrier(0);
        if (kind == 0) {
            // quad, difference form (see faceFluxGvp3Kernel): V d_d phi = a5_d (phi_O - phi_N) + a0_d (phi_1 - phi_3) + a1_d (phi_2 - phi_4)
            const double NO[3] = {cN.x - cO.x, cN.y - cO.y, cN.z - cO.z};
            const double d24[3] = {x1.x - x3.x, x1.y - x3.y, x1.z - x3.z};
            const double d31[3] = {x2.x - x0.x, x2.y - x0.y, x2.z - x0.z};
            double A0[3], A1[3], A5[3];
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const int u = (d + 1) % 3, w2 = (d + 2) % 3;
                A0[d] = NO[u] * d24[w2] - NO[w2] * d24[u];
                A1[d] = NO[u] * d31[w2] - NO[w2] * d31[u];
                A5[d] = d24[u] * d31[w2] - d24[w2] * d31[u];
            }
            const double rV6 = -QGD_RCP(d31[0] * A0[0] + d31[1] * A0[1] + d31[2] * A0[2]);
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                const double D5 = (v.o[k] - v.n[k]) * rV6, D0 = (p0[k] - p2[k]) * rV6, D1 = (p1[k] - p3[k]) * rV6;
#pragma unroll
                for (int d = 0; d < 3; ++d) g[d * NC + k] = A5[d] * D5 + A0[d] * D0 + A1[d] * D1;
            }
        } else if (kind == 1) {
            double t[12], rV;
            gvpTriCoef(cO, cN, x0, x1, x2, t, rV);
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const double a0 = t[4 * d], a1 = t[4 * d + 1], a2 = t[4 * d + 2], a3 = t[4 * d + 3];
#pragma unroll
                for (int k = 0; k < NC; ++k) {
                    double sg = v.n[k] * a3;
                    sg += v.o[k] * (-a3);
                    sg += p0[k] * a0;
                    sg += p1[k] * a1;
                    sg += p2[k] * a2;
                    g[d * NC + k] = sg * rV;
                }
            }
            if (NC == 3) {  // interior triangles of a vector field: every row holds (dxUx, dyUy, dzUz) [3D.C L844-854]
                double dg[3];
#pragma unroll
                for (int j = 0; j < 3; ++j) dg[j] = g[j * NC + (j % NC)];
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) g[i * NC + (j % NC)] = dg[j];
            }
        } else {
            faceGradient<ST_GVP3, NC, (NC == 3) ? 0 : -1>(m, f, v, cellF, ptF, g);  // > 4 vertices: nf*snGrad [3D.C L759-768]
        }
    }
#pragma unroll
    for (int i = 0; i < 3 * NC; ++i) stage[(size_t)threadIdx.x * 3 * NC + i] = g[i];
    __syncthreads();
    const int64_t first = (int64_t)tile * QGD_BLOCK;
    const int64_t cnt = ((m.nIF - first) < QGD_BLOCK ? (m.nIF - first) : QGD_BLOCK) * 3 * NC;
    double* dst = out + first * 3 * NC;
    for (int i = threadIdx.x; i < cnt; i += QGD_BLOCK) dst[i] = stage[i];
}

// ---------------------------------------------------------------------------
// QHDFoam face fluxes [QHDFoam/updateFields.H L36-73, updateFluxes.H L33-38, QHDUEqn_8H L36-43, QHDTEqn_8H L65-66]
// one thread per face (internal and patch faces); fields U,T,p travel as one 5-component record so the three
// fvsc::grad calls are a single stencil walk.
// ---------------------------------------------------------------------------
template <int ST>
__global__ __launch_bounds__(QGD_BLOCK) void qhdFaceKernel(const MeshView m, const double* __restrict__ cell5,
                                                          const double* __restrict__ bnd5, const double* __restrict__ pt5,
                                                          const double* __restrict__ rho, const double* __restrict__ rhob,
                                                          const double* __restrict__ tauF, const double* __restrict__ phiF,
                                                          const double beta, const double gx, const double gy, const double gz,
                                                          double* __restrict__ out) {
    const int f = blockIdx.x * QGD_BLOCK + threadIdx.x;
    if (f >= m.nF) return;
    if (m.fkind[f] == 3) return;  // empty patches carry no field
    const bool internal = f < m.nIF;
    const int o = m.own[f];
    FaceVals<5> v;
#pragma unroll
    for (int k = 0; k < 5; ++k) v.o[k] = cell5[(size_t)o * 5 + k];
    double rhof, w = 1.0;
    if (internal) {
        const int n = m.nei[f];
#pragma unroll
        for (int k = 0; k < 5; ++k) { v.n[k] = cell5[(size_t)n * 5 + k]; v.sn[k] = 0.0; }
        w = m.w[f];
        rhof = lerpf(w, rho[o], rho[n]);
    } else {
        const int b = f - m.nIF;
        const double dc = m.dn[f];
#pragma unroll
        for (int k = 0; k < 5; ++k) { v.n[k] = bnd5[(size_t)b * 5 + k]; v.sn[k] = dc * (v.n[k] - v.o[k]); }
        rhof = rhob[b];
    }
    double g[15];
    faceGradient<ST, 5, 0>(m, f, v, cell5, pt5, g);
    double gU[9], gT[3], gP[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        gU[3 * i] = g[i * 5]; gU[3 * i + 1] = g[i * 5 + 1]; gU[3 * i + 2] = g[i * 5 + 2];
        gT[i] = g[i * 5 + 3]; gP[i] = g[i * 5 + 4];
    }
    const double gv[3] = {gx, gy, gz};
    double Uf[3], Bf[3];
    double Tf;
    if (internal) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            Uf[k] = lerpf(w, v.o[k], v.n[k]);
            Bf[k] = lerpf(w, (beta * v.o[3]) * gv[k], (beta * v.n[3]) * gv[k]);  // BdFrc = beta*T*g [updateFields.H L66-67]
        }
        Tf = lerpf(w, v.o[3], v.n[3]);
    } else {
#pragma unroll
        for (int k = 0; k < 3; ++k) { Uf[k] = v.n[k]; Bf[k] = (beta * v.n[3]) * gv[k]; }
        Tf = v.n[3];
    }
    const double S[3] = {m.Sx[f], m.Sy[f], m.Sz[f]};
    const double tau = tauF[f];
    const size_t nF = (size_t)m.nF;
    auto put = [&](int slot, double x) { out[(size_t)slot * nF + f] = x; };
    const double phiu = S[0] * Uf[0] + S[1] * Uf[1] + S[2] * Uf[2];                       // [updateFluxes.H L33]
    double UgU[3], wo[3], Wf[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        UgU[j] = Uf[0] * gU[j] + Uf[1] * gU[3 + j] + Uf[2] * gU[6 + j];                  // Uf & gradUf
        wo[j] = tau * (UgU[j] - Bf[j]);
        Wf[j] = tau * ((UgU[j] + gP[j] / rhof) - Bf[j]);                                  // [QHDUEqn_8H L37]
    }
    const double phiwo = S[0] * wo[0] + S[1] * wo[1] + S[2] * wo[2];                      // [updateFluxes.H L35]
    for (int k = 0; k < 9; ++k) put(QHD_GRADU + k, gU[k]);
    for (int k = 0; k < 3; ++k) { put(QHD_GRADT + k, gT[k]); put(QHD_GRADP + k, gP[k]); put(QHD_WF + k, Wf[k]); }
    put(QHD_PHIU, phiu);
    put(QHD_PHIWO, phiwo);
    put(QHD_TAUBYRHO, tau / rhof);                                                        // [updateFluxes.H L38]
    const double phi = phiF ? phiF[f] : 0.0;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const double uw = S[0] * (Uf[0] * Wf[j]) + S[1] * (Uf[1] * Wf[j]) + S[2] * (Uf[2] * Wf[j]);  // Sf & (Uf*Wf)
        put(QHD_PHIUF + j, phi * Uf[j] - uw);                                             // [QHDUEqn_8H L39-43]
    }
    put(QHD_PHITF, phi * Tf);                                                             // [QHDTEqn_8H L65]
    put(QHD_PHITAUT, tau * phiu * (Uf[0] * gT[0] + Uf[1] * gT[1] + Uf[2] * gT[2]));       // [QHDTEqn_8H L66]
}

// Named cell / patch field out of the records into a dense array (the accessor path of qgd_case_get_field: one pass on
// the device and one copy of what was asked for, instead of shipping every record to the host).
__global__ __launch_bounds__(QGD_BLOCK) void extractFieldKernel(const RecA* __restrict__ A, const RecB* __restrict__ B,
                                                               const double* __restrict__ rE, const double* __restrict__ hq,
                                                               const double* __restrict__ aQ,
                                                               const int64_t n, const GasModel g, const int field,
                                                               double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * QGD_BLOCK + threadIdx.x;
    if (i >= n) return;
    const RecA a = A[i];
    const RecB b = B[i];
    const double ke = 0.5 * (a.ux * a.ux + a.uy * a.uy + a.uz * a.uz);
    switch (field) {
        case XF_RHO: out[i] = a.rho; break;
        case XF_U: out[3 * i] = a.ux; out[3 * i + 1] = a.uy; out[3 * i + 2] = a.uz; break;
        case XF_P: out[i] = a.p; break;
        case XF_E: out[i] = a.e; break;
        case XF_T: out[i] = a.e / g.Cv; break;
        case XF_RHOU:
            out[3 * i] = a.rho * a.ux; out[3 * i + 1] = a.rho * a.uy; out[3 * i + 2] = a.rho * a.uz;
            break;
        case XF_RHOE: out[i] = rE ? rE[i] : a.rho * (a.e + ke); break;
        case XF_C: out[i] = b.c; break;
        case XF_PSI: out[i] = 1.0 / (g.R * (a.e / g.Cv)); break;
        case XF_MU: out[i] = g.mu0 + b.muQGD; break;
#if QGD_F_DIET
        case XF_ALPHAU: out[i] = g.alphah0 + b.muQGD * g.rPrQGD; break;     // as alphaEffOf forms it in the face kernels
#else
        case XF_ALPHAU: out[i] = g.alphah0 + b.muQGD / g.PrQGD; break;
#endif
        case XF_TAUQGD: out[i] = (aQ ? aQ[i] : g.alphaQGD) * hq[i] / b.c; break;
        case XF_MUQGD: out[i] = b.muQGD; break;
#if QGD_F_DIET
        case XF_ALPHAUQGD: out[i] = b.muQGD * g.rPrQGD; break;
#else
        case XF_ALPHAUQGD: out[i] = b.muQGD / g.PrQGD; break;
#endif
        case XF_HQGD: out[i] = hq[i]; break;
        case XF_H: out[i] = b.H; break;
        default: out[i] = g.gamma; break;
    }
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
static inline int gridFor(int64_t n) { return (int)((n + QGD_BLOCK - 1) / QGD_BLOCK); }
#define QGD_TIMED(L, k, stmt)            \
    do {                                 \
        if ((L).pre) (L).pre((L).ctx, k); \
        stmt;                            \
        if ((L).post) (L).post((L).ctx, k); \
    } while (0)

void launchPointInterp(const Launcher& L, const MeshView& m, const CaseView& c) {
    if (m.nP == 0) return;
    if (m.pblock == 64) QGD_TIMED(L, QGD_K_POINT, (pointInterpRecKernel<64><<<(m.nP + 63) / 64, 64, 0, L.stream>>>(m, c.A, c.P)));
    else if (m.pblock == 128) QGD_TIMED(L, QGD_K_POINT, (pointInterpRecKernel<128><<<(m.nP + 127) / 128, 128, 0, L.stream>>>(m, c.A, c.P)));
    else QGD_TIMED(L, QGD_K_POINT, (pointInterpRecKernel<256><<<gridFor(m.nP), QGD_BLOCK, 0, L.stream>>>(m, c.A, c.P)));
}
void launchBoundaryPoints(const Launcher& L, const MeshView& m, const CaseView& c, bool pOnly) {
    if (m.nBP == 0) return;
    if (pOnly)
        QGD_TIMED(L, QGD_K_BPOINT, (boundaryPointKernel<1><<<gridFor(m.nBP), QGD_BLOCK, 0, L.stream>>>(
            m, c.bPmid, 1, reinterpret_cast<double*>(c.P), 6, 4, -1)));
    else
        QGD_TIMED(L, QGD_K_BPOINT, (boundaryPointKernel<6><<<gridFor(m.nBP), QGD_BLOCK, 0, L.stream>>>(
            m, reinterpret_cast<const double*>(c.bA), 6, reinterpret_cast<double*>(c.P), 6, 0, m.nGeomD == 3 ? 1 : -1)));   // U of the records: a pointVectorField in 3-D only
}
template <bool DBG, bool UPW>
static void launchFaceFluxT(const Launcher& L, int stencil, const MeshView& m, const CaseView& c, const GasModel& g, bool adj) {
    const int grid = gridFor(m.nIF);
    if (grid == 0) return;
    switch (stencil) {
        case ST_REDUCED: faceFluxReducedKernel<DBG, UPW><<<grid, QGD_BLOCK, 0, L.stream>>>(m, c, g, adj); break;
        case ST_LSQ: faceFluxLsqKernel<DBG, UPW><<<grid, QGD_BLOCK, 0, L.stream>>>(m, c, g, adj); break;
        case ST_GVP3:
            if constexpr (UPW) {
                // `Gauss upwind` fluxes: the default tile configuration has its upwind instantiation, every other one takes the gather kernel
                if (!DBG && m.tileOff != nullptr && m.fblock == 128 && m.tileWaves == 3 && m.sGeo && m.tileFlag) {
                    faceFluxGvp3TileKernel<128, 3, true, true, true><<<(m.nIF + 127) / 128, 128, m.tileLds, L.stream>>>(m, c, g, adj);
                    if (m.nTileSpill > 0) faceFluxGvp3Kernel<false, 128, true, true><<<m.nTileSpill, 128, 0, L.stream>>>(m, c, g, adj, m.tileSpill);
                } else if (m.fblock == 64) faceFluxGvp3Kernel<DBG, 64, false, true><<<(m.nIF + 63) / 64, 64, 0, L.stream>>>(m, c, g, adj, nullptr);
                else if (m.fblock == 128) faceFluxGvp3Kernel<DBG, 128, false, true><<<(m.nIF + 127) / 128, 128, 0, L.stream>>>(m, c, g, adj, nullptr);
                else faceFluxGvp3Kernel<DBG, 256, false, true><<<grid, QGD_BLOCK, 0, L.stream>>>(m, c, g, adj, nullptr);
                break;
            }
            if (!DBG && m.tileOff != nullptr) {
                if (m.fblock == 64) faceFluxGvp3TileKernel<64, 3><<<(m.nIF + 63) / 64, 64, m.tileLds, L.stream>>>(m, c, g, adj);
                else if (m.fblock == 256) faceFluxGvp3TileKernel<256, 3><<<grid, QGD_BLOCK, m.tileLds, L.stream>>>(m, c, g, adj);
                else if (m.tileWaves == 2) faceFluxGvp3TileKernel<128, 2><<<(m.nIF + 127) / 128, 128, m.tileLds, L.stream>>>(m, c, g, adj);
                else if (m.tileWaves == 4) faceFluxGvp3TileKernel<128, 4><<<(m.nIF + 127) / 128, 128, m.tileLds, L.stream>>>(m, c, g, adj);
                else if (m.sGeo && m.tileFlag) faceFluxGvp3TileKernel<128, 3, true, true><<<(m.nIF + 127) / 128, 128, m.tileLds, L.stream>>>(m, c, g, adj);
                else if (m.tileFlag) faceFluxGvp3TileKernel<128, 3, false, true><<<(m.nIF + 127) / 128, 128, m.tileLds, L.stream>>>(m, c, g, adj);
                else if (m.sGeo) faceFluxGvp3TileKernel<128, 3, true><<<(m.nIF + 127) / 128, 128, m.tileLds, L.stream>>>(m, c, g, adj);
                else faceFluxGvp3TileKernel<128, 3><<<(m.nIF + 127) / 128, 128, m.tileLds, L.stream>>>(m, c, g, adj);
                if (m.nTileSpill > 0) {
                    if (m.sGeo && m.fblock == 128 && m.tileWaves == 3) faceFluxGvp3Kernel<false, 128, true><<<m.nTileSpill, 128, 0, L.stream>>>(m, c, g, adj, m.tileSpill);
                    else if (m.fblock == 64) faceFluxGvp3Kernel<false, 64><<<m.nTileSpill, 64, 0, L.stream>>>(m, c, g, adj, m.tileSpill);
                    else if (m.fblock == 256) faceFluxGvp3Kernel<false, 256><<<m.nTileSpill, 256, 0, L.stream>>>(m, c, g, adj, m.tileSpill);
                    else faceFluxGvp3Kernel<false, 128><<<m.nTileSpill, 128, 0, L.stream>>>(m, c, g, adj, m.tileSpill);
                }
            } else if (m.fblock == 64) faceFluxGvp3Kernel<DBG, 64><<<(m.nIF + 63) / 64, 64, 0, L.stream>>>(m, c, g, adj, nullptr);
            else if (m.fblock == 128 && m.sGeo && m.tileWaves == 3) faceFluxGvp3Kernel<DBG, 128, true><<<(m.nIF + 127) / 128, 128, 0, L.stream>>>(m, c, g, adj, nullptr);
            else if (m.fblock == 128) faceFluxGvp3Kernel<DBG, 128><<<(m.nIF + 127) / 128, 128, 0, L.stream>>>(m, c, g, adj, nullptr);
            else faceFluxGvp3Kernel<DBG, 256><<<grid, QGD_BLOCK, 0, L.stream>>>(m, c, g, adj, nullptr);
            break;
        default: faceFluxGvp2Kernel<DBG, UPW><<<grid, QGD_BLOCK, 0, L.stream>>>(m, c, g, adj); break;
    }
}
// the two-stencil walk of a case with per-term fvsc entries; (a, b) normalised to a < b by the caller
template <bool DBG, bool UPW>
static void launchFaceFluxMixedT(const Launcher& L, int a, int b, int maskB, const MeshView& m, const CaseView& c, const GasModel& g, bool adj) {
    const int grid = gridFor(m.nIF);
    if (grid == 0) return;
    if (a == ST_REDUCED && b == ST_LSQ) faceFluxMixedKernel<ST_REDUCED, ST_LSQ, DBG, UPW><<<grid, QGD_BLOCK, 0, L.stream>>>(m, c, g, adj, maskB);
    else if (a == ST_REDUCED && b == ST_GVP3) faceFluxMixedKernel<ST_REDUCED, ST_GVP3, DBG, UPW><<<grid, QGD_BLOCK, 0, L.stream>>>(m, c, g, adj, maskB);
    else if (a == ST_REDUCED && b == ST_GVP2) faceFluxMixedKernel<ST_REDUCED, ST_GVP2, DBG, UPW><<<grid, QGD_BLOCK, 0, L.stream>>>(m, c, g, adj, maskB);
    else if (a == ST_LSQ && b == ST_GVP2) faceFluxMixedKernel<ST_LSQ, ST_GVP2, DBG, UPW><<<grid, QGD_BLOCK, 0, L.stream>>>(m, c, g, adj, maskB);
    else throw std::invalid_argument("launchFaceFluxMixed: no such pair of stencils on one mesh");
}
void launchFaceFluxMixed(const Launcher& L, int a, int b, int maskB, const MeshView& m, const CaseView& c, const GasModel& g, bool adjustDt) {
    const bool upw = g.upwindU || g.upwindH;
    QGD_TIMED(L, QGD_K_FACE, (c.dbg ? (upw ? launchFaceFluxMixedT<true, true>(L, a, b, maskB, m, c, g, adjustDt) : launchFaceFluxMixedT<true, false>(L, a, b, maskB, m, c, g, adjustDt))
                                    : (upw ? launchFaceFluxMixedT<false, true>(L, a, b, maskB, m, c, g, adjustDt) : launchFaceFluxMixedT<false, false>(L, a, b, maskB, m, c, g, adjustDt))));
}
template <bool DBG>
static void launchBFaceFluxMixedT(const Launcher& L, int a, int b, int maskB, const MeshView& m, const CaseView& c, const GasModel& g,
                                  const PatchBCDev* bc, int phiwOnly, bool adj) {
    const int grid = gridFor(m.nBF);
    if (grid == 0) return;
    if (a == ST_REDUCED && b == ST_LSQ) boundaryFaceFluxKernel<ST_REDUCED, DBG, ST_LSQ><<<grid, QGD_BLOCK, 0, L.stream>>>(m, c, g, bc, phiwOnly, adj, maskB);
    else if (a == ST_REDUCED && b == ST_GVP3) boundaryFaceFluxKernel<ST_REDUCED, DBG, ST_GVP3><<<grid, QGD_BLOCK, 0, L.stream>>>(m, c, g, bc, phiwOnly, adj, maskB);
    else if (a == ST_REDUCED && b == ST_GVP2) boundaryFaceFluxKernel<ST_REDUCED, DBG, ST_GVP2><<<grid, QGD_BLOCK, 0, L.stream>>>(m, c, g, bc, phiwOnly, adj, maskB);
    else if (a == ST_LSQ && b == ST_GVP2) boundaryFaceFluxKernel<ST_LSQ, DBG, ST_GVP2><<<grid, QGD_BLOCK, 0, L.stream>>>(m, c, g, bc, phiwOnly, adj, maskB);
    else throw std::invalid_argument("launchBoundaryFaceFluxMixed: no such pair of stencils on one mesh");
}
void launchBoundaryFaceFluxMixed(const Launcher& L, int a, int b, int maskB, const MeshView& m, const CaseView& c, const GasModel& g,
                                 const PatchBCDev* bc, int phiwOnly, bool adjustDt) {
    QGD_TIMED(L, QGD_K_BFACE, (c.dbg && phiwOnly != 1 ? launchBFaceFluxMixedT<true>(L, a, b, maskB, m, c, g, bc, phiwOnly, adjustDt)
                                                   : launchBFaceFluxMixedT<false>(L, a, b, maskB, m, c, g, bc, phiwOnly, adjustDt)));
}
void launchFaceFlux(const Launcher& L, int stencil, const MeshView& m, const CaseView& c, const GasModel& g, bool adjustDt) {
    const bool upw = g.upwindU || g.upwindH;
    QGD_TIMED(L, QGD_K_FACE, (c.dbg ? (upw ? launchFaceFluxT<true, true>(L, stencil, m, c, g, adjustDt) : launchFaceFluxT<true, false>(L, stencil, m, c, g, adjustDt))
                                    : (upw ? launchFaceFluxT<false, true>(L, stencil, m, c, g, adjustDt) : launchFaceFluxT<false, false>(L, stencil, m, c, g, adjustDt))));
}
template <bool DBG>
static void launchBFaceFluxT(const Launcher& L, int stencil, const MeshView& m, const CaseView& c, const GasModel& g,
                             const PatchBCDev* bc, int phiwOnly, bool adj) {
    const int grid = gridFor(m.nBF);
    if (grid == 0) return;
    switch (stencil) {
        case ST_REDUCED: boundaryFaceFluxKernel<ST_REDUCED, DBG><<<grid, QGD_BLOCK, 0, L.stream>>>(m, c, g, bc, phiwOnly, adj); break;
        case ST_LSQ: boundaryFaceFluxKernel<ST_LSQ, DBG><<<grid, QGD_BLOCK, 0, L.stream>>>(m, c, g, bc, phiwOnly, adj); break;
        case ST_GVP3: boundaryFaceFluxKernel<ST_GVP3, DBG><<<grid, QGD_BLOCK, 0, L.stream>>>(m, c, g, bc, phiwOnly, adj); break;
        default: boundaryFaceFluxKernel<ST_GVP2, DBG><<<grid, QGD_BLOCK, 0, L.stream>>>(m, c, g, bc, phiwOnly, adj); break;
    }
}
void launchBoundaryFaceFlux(const Launcher& L, int stencil, const MeshView& m, const CaseView& c, const GasModel& g,
                            const PatchBCDev* bc, int phiwOnly, bool adjustDt) {
    QGD_TIMED(L, QGD_K_BFACE, (c.dbg && phiwOnly != 1 ? launchBFaceFluxT<true>(L, stencil, m, c, g, bc, phiwOnly, adjustDt)
                                                   : launchBFaceFluxT<false>(L, stencil, m, c, g, bc, phiwOnly, adjustDt)));
}
void launchFusedFaceCell(const Launcher& L, const MeshView& m, const CaseView& c, const GasModel& g, int firstBlock, int nBlocks) {
    if (nBlocks <= 0) return;
    const bool upw = g.upwindU || g.upwindH;   // a `Gauss upwind` entry for div(phiJm,U) or div(phiJm,H)
    const ImplView none{};
    if (m.sGeo && upw) QGD_TIMED(L, QGD_K_FACE, (fusedFaceCellKernel<true, true><<<nBlocks, 256, m.fuLds, L.stream>>>(m, c, g, firstBlock, none, nullptr)));
    else if (upw) QGD_TIMED(L, QGD_K_FACE, (fusedFaceCellKernel<false, true><<<nBlocks, 256, m.fuLds, L.stream>>>(m, c, g, firstBlock, none, nullptr)));
    else if (m.sGeo) QGD_TIMED(L, QGD_K_FACE, (fusedFaceCellKernel<true><<<nBlocks, 256, m.fuLds, L.stream>>>(m, c, g, firstBlock, none, nullptr)));
    else QGD_TIMED(L, QGD_K_FACE, (fusedFaceCellKernel<false><<<nBlocks, 256, m.fuLds, L.stream>>>(m, c, g, firstBlock, none, nullptr)));
}
// Courant-number control: all blocks up to their flux sums (+ the Courant partials), then -- after deltaTKernel -- the cells
void launchFusedAdjust(const Launcher& L, const MeshView& m, const CaseView& c, const GasModel& g) {
    if (m.fuBlocks <= 0) return;
    const bool upw = g.upwindU || g.upwindH;
    const ImplView none{};
    const int n = m.fuBlocks;
    if (m.sGeo && upw) QGD_TIMED(L, QGD_K_FACE, (fusedFaceCellKernel<true, true, false, true><<<n, 256, m.fuLds, L.stream>>>(m, c, g, 0, none, nullptr)));
    else if (upw) QGD_TIMED(L, QGD_K_FACE, (fusedFaceCellKernel<false, true, false, true><<<n, 256, m.fuLds, L.stream>>>(m, c, g, 0, none, nullptr)));
    else if (m.sGeo) QGD_TIMED(L, QGD_K_FACE, (fusedFaceCellKernel<true, false, false, true><<<n, 256, m.fuLds, L.stream>>>(m, c, g, 0, none, nullptr)));
    else QGD_TIMED(L, QGD_K_FACE, (fusedFaceCellKernel<false, false, false, true><<<n, 256, m.fuLds, L.stream>>>(m, c, g, 0, none, nullptr)));
}
void launchCellFinish(const Launcher& L, const MeshView& m, const CaseView& c, const GasModel& g, int mode, const int32_t* list, int nList) {
    const int n = (mode == 1) ? nList : m.nC;
    if (n == 0) return;
    // (mode 1 uses the monitor slots behind the fused kernel's / the cell kernel's, like launchCellUpdate)
    const int slotBase = (mode == 1) ? std::max((m.nC + m.cblock - 1) / m.cblock, m.fuBlocks) : 0;
    QGD_TIMED(L, QGD_K_CELL, (cellFinishKernel<256><<<(n + 255) / 256, 256, 0, L.stream>>>(m, c, g, mode, list, nList, slotBase)));
}
// the implicitDiffusion branch's block-fused assembly of the U systems (fusedFaceCellKernel<..., IMPL = true>): more dynamic LDS than the
// 64 KB a launch gets by default, so the kernels' limit is raised first; false when the device refuses (the caller keeps the separate kernels)
template <bool SGEO, bool UPW>
static bool fusedImplLimit(int lds) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&fusedFaceCellKernel<SGEO, UPW, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess;
}
bool fusedImplUPrepare(const MeshView& m, const GasModel& g) {
    if (m.fuBlocks <= 0 || m.fuLdsImpl <= 0) return false;
    const bool upw = g.upwindU || g.upwindH;
    const bool ok = m.sGeo ? (upw ? fusedImplLimit<true, true>(m.fuLdsImpl) : fusedImplLimit<true, false>(m.fuLdsImpl))
                           : (upw ? fusedImplLimit<false, true>(m.fuLdsImpl) : fusedImplLimit<false, false>(m.fuLdsImpl));
    if (!ok) (void)hipGetLastError();
    return ok;
}
void launchFusedImplU(const Launcher& L, const MeshView& m, const CaseView& c, const GasModel& g, const ImplView& iv, const PatchBCDev* bc) {
    const bool upw = g.upwindU || g.upwindH;
    const int n = m.fuBlocks;
    if (m.sGeo && upw) QGD_TIMED(L, QGD_K_FACE, (fusedFaceCellKernel<true, true, true><<<n, 256, m.fuLdsImpl, L.stream>>>(m, c, g, 0, iv, bc)));
    else if (upw) QGD_TIMED(L, QGD_K_FACE, (fusedFaceCellKernel<false, true, true><<<n, 256, m.fuLdsImpl, L.stream>>>(m, c, g, 0, iv, bc)));
    else if (m.sGeo) QGD_TIMED(L, QGD_K_FACE, (fusedFaceCellKernel<true, false, true><<<n, 256, m.fuLdsImpl, L.stream>>>(m, c, g, 0, iv, bc)));
    else QGD_TIMED(L, QGD_K_FACE, (fusedFaceCellKernel<false, false, true><<<n, 256, m.fuLdsImpl, L.stream>>>(m, c, g, 0, iv, bc)));
}
void launchCellUpdate(const Launcher& L, const MeshView& m, const CaseView& c, const GasModel& g, int mode,
                      const int32_t* list, int nList) {
    const int n = (mode == 1) ? nList : m.nC;
    if (n == 0) return;
    const int slotBase = (mode == 1) ? cellBlocks(m) : 0;  // the list launch monitors min(rho), min(e) in its own slots
    if (m.cblock == 64) QGD_TIMED(L, QGD_K_CELL, (cellUpdateKernel<64><<<(n + 63) / 64, 64, 0, L.stream>>>(m, c, g, mode, list, nList, slotBase)));
    else if (m.cblock == 128) QGD_TIMED(L, QGD_K_CELL, (cellUpdateKernel<128><<<(n + 127) / 128, 128, 0, L.stream>>>(m, c, g, mode, list, nList, slotBase)));
    else QGD_TIMED(L, QGD_K_CELL, (cellUpdateKernel<256><<<gridFor(n), QGD_BLOCK, 0, L.stream>>>(m, c, g, mode, list, nList, slotBase)));
}
void launchBoundaryUpdate(const Launcher& L, const MeshView& m, const CaseView& c, const GasModel& g, const PatchBCDev* bc,
                          bool init, bool phiwRegistered, int mode, const int32_t* list, int nList) {
    const int n = (mode == 1) ? nList : m.nBF;
    if (n == 0) return;
    QGD_TIMED(L, QGD_K_BC, (boundaryUpdateKernel<<<gridFor(n), QGD_BLOCK, 0, L.stream>>>(m, c, g, bc, init, phiwRegistered, mode,
                                                                                          list, nList)));
}
void launchCellInit(const Launcher& L, const MeshView& m, const CaseView& c, const GasModel& g, const double* U, const double* T,
                    const double* p) {
    cellInitKernel<<<gridFor(m.nC), QGD_BLOCK, 0, L.stream>>>(m, c, g, U, T, p);
}
void launchDeltaT(const Launcher& L, const CaseView& c, double maxCo, double maxDeltaT, double cTau) {
    deltaTKernel<<<1, 1, 0, L.stream>>>(c, maxCo, maxDeltaT, cTau);
}
void launchFaceReduce(const Launcher& L, const CaseView& c) {
    const int g = std::max(1, std::min(QGD_FACE_REDUCE_PARTIALS, (c.nBlkFace + QGD_BLOCK - 1) / QGD_BLOCK));
    faceReduceStage1Kernel<<<g, QGD_BLOCK, 0, L.stream>>>(c);
    faceReduceKernel<<<1, QGD_BLOCK, 0, L.stream>>>(c, g);
}
void launchResetReductions(const Launcher& L, const CaseView& c) { resetReductionsKernel<<<64, QGD_BLOCK, 0, L.stream>>>(c); }
void launchCellMinReduce(const Launcher& L, const CaseView& c) { cellMinReduceKernel<<<1, QGD_BLOCK, 0, L.stream>>>(c); }
int faceBlocks(const MeshView& m) { return (m.nIF + 63) / 64; }
int bfaceBlocks(const MeshView& m) { return gridFor(m.nBF); }
int cellBlocks(const MeshView& m) { return (m.nC + 63) / 64; }  // slots laid out for the smallest cell tile
// the message in the middle of the flux assembly: mid-step patch pressure and qgdFlux gradient of patch faces (2 doubles each)
__global__ __launch_bounds__(QGD_BLOCK) void midHaloKernel(const CaseView c, const int32_t* __restrict__ bfaces, const int n,
                                                          double* __restrict__ buf, const int pack) {
    const int i = blockIdx.x * QGD_BLOCK + threadIdx.x;
    if (i >= n) return;
    const int b = bfaces[i];
    if (pack) { buf[2 * (size_t)i] = c.bPmid[b]; buf[2 * (size_t)i + 1] = c.bG[b]; }
    else { c.bPmid[b] = buf[2 * (size_t)i]; c.bG[b] = buf[2 * (size_t)i + 1]; }
}
void launchMidHalo(hipStream_t s, const CaseView& c, const int32_t* bfaces, int32_t n, double* buf, bool pack) {
    if (n > 0) midHaloKernel<<<gridFor(n), QGD_BLOCK, 0, s>>>(c, bfaces, n, buf, pack ? 1 : 0);
}
void launchHaloPack(const Launcher& L, const CaseView& c, const GasModel& g, const int32_t* cells, int32_t nCells, const int32_t* bfaces,
                    int32_t nFaces, double* buf, bool pack) {
    const int n = nCells + nFaces;
    if (n == 0) return;
    haloKernel<<<gridFor(n), QGD_BLOCK, 0, L.stream>>>(c, g, cells, nCells, bfaces, nFaces, buf, pack ? 1 : 0);
}

// helpers of the device-pointer operator entries: {U,T,p} -> 5-component records; SoA result slots -> one AoS face field
__global__ __launch_bounds__(QGD_BLOCK) void pack5Kernel(const int64_t n, const double* __restrict__ U, const double* __restrict__ T,
                                                        const double* __restrict__ p, double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * QGD_BLOCK + threadIdx.x;
    if (i >= n) return;
    out[5 * i] = U[3 * i]; out[5 * i + 1] = U[3 * i + 1]; out[5 * i + 2] = U[3 * i + 2];
    out[5 * i + 3] = T[i];
    out[5 * i + 4] = p ? p[i] : 0.0;
}
__global__ __launch_bounds__(QGD_BLOCK) void soaToAosKernel(const int64_t n, const int nc, const double* __restrict__ src, double* __restrict__ dst) {
    const int64_t i = (int64_t)blockIdx.x * QGD_BLOCK + threadIdx.x;
    if (i >= n) return;
    for (int k = 0; k < nc; ++k) dst[i * nc + k] = src[(size_t)k * n + i];
}
void launchPack5(hipStream_t s, int64_t n, const double* U, const double* T, const double* p, double* out) {
    if (n > 0) pack5Kernel<<<gridFor(n), QGD_BLOCK, 0, s>>>(n, U, T, p, out);
}
void launchSoaToAos(hipStream_t s, int64_t n, int nc, const double* src, double* dst) {
    if (n > 0) soaToAosKernel<<<gridFor(n), QGD_BLOCK, 0, s>>>(n, nc, src, dst);
}

void launchExtractField(hipStream_t s, const RecA* A, const RecB* B, const double* K, const double* hq, const double* aQ, int64_t n,
                        const GasModel& g, int field, double* out) {
    if (n == 0) return;
    extractFieldKernel<<<gridFor(n), QGD_BLOCK, 0, s>>>(A, B, K, hq, aQ, n, g, field, out);
}

template <int ST, int NC>
static void launchFvscOpT(hipStream_t s, int op, const MeshView& m, const double* cell, const double* bnd, double* pt, double* out) {
    if (ST == ST_GVP3 || ST == ST_GVP2) {
        pointInterpFastKernel<NC><<<gridFor(m.nP), QGD_BLOCK, 0, s>>>(m, cell, pt);
        // what the reference interpolates as a pointVectorField / pointTensorField: the 3-D operators [GaussVolPointBase3D.C L938, L969,
        // L1000, L1026] and the 2-D divergences [GaussVolPointBase2D.C L376-382, L454-460]; the 2-D gradient of a vector goes
        // component by component [GaussVolPointBase.C L79-87]
        const int vecMode = NC == 9 ? -2 : ((NC == 3 && (op != 0 || ST == ST_GVP3)) ? 0 : -1);
        if (m.nBP) boundaryPointKernel<NC><<<gridFor(m.nBP), QGD_BLOCK, 0, s>>>(m, bnd, NC, pt, NC, 0, vecMode);
    }
    if (ST == ST_GVP3 && op == 0 && NC <= 3) {
        // the drop-in path of updateFluxes.H: internal faces through the loads-first kernel, patch faces through the generic one
        if (m.nIF) fvscGradGvp3Kernel<(NC <= 3 ? NC : 1)><<<gridFor(m.nIF), QGD_BLOCK, 0, s>>>(m, cell, pt, out);
        if (m.nBF) fvscOpKernel<ST, NC, 0><<<gridFor(m.nBF), QGD_BLOCK, 0, s>>>(m, cell, bnd, pt, out, m.nIF);
        return;
    }
    if (op == 0) fvscOpKernel<ST, NC, 0><<<gridFor(m.nF), QGD_BLOCK, 0, s>>>(m, cell, bnd, pt, out, 0);
    else fvscOpKernel<ST, NC, (NC >= 3 ? 1 : 0)><<<gridFor(m.nF), QGD_BLOCK, 0, s>>>(m, cell, bnd, pt, out, 0);
}
template <int ST>
static void launchFvscOpS(hipStream_t s, int op, int NC, const MeshView& m, const double* cell, const double* bnd, double* pt, double* out) {
    if (NC == 1) launchFvscOpT<ST, 1>(s, op, m, cell, bnd, pt, out);
    else if (NC == 3) launchFvscOpT<ST, 3>(s, op, m, cell, bnd, pt, out);
    else launchFvscOpT<ST, 9>(s, op, m, cell, bnd, pt, out);
}
void launchFvscOp(hipStream_t s, int stencil, int op, int NC, const MeshView& m, const double* cell, const double* bnd, double* pt,
                  double* out) {
    switch (stencil) {
        case ST_REDUCED: launchFvscOpS<ST_REDUCED>(s, op, NC, m, cell, bnd, pt, out); break;
        case ST_LSQ: launchFvscOpS<ST_LSQ>(s, op, NC, m, cell, bnd, pt, out); break;
        case ST_GVP3: launchFvscOpS<ST_GVP3>(s, op, NC, m, cell, bnd, pt, out); break;
        default: launchFvscOpS<ST_GVP2>(s, op, NC, m, cell, bnd, pt, out); break;
    }
}

// qgdInterpolate with the default (linear) scheme [QGDInterpolate_8H L38-67]
__global__ __launch_bounds__(QGD_BLOCK) void interpolateKernel(const MeshView m, const int NC, const double* __restrict__ cell,
                                                              const double* __restrict__ bnd, double* __restrict__ out) {
    const int f = blockIdx.x * QGD_BLOCK + threadIdx.x;
    if (f >= m.nF) return;
    if (f < m.nIF) {
        const int o = m.own[f], n = m.nei[f];
        const double w = m.w[f];
        for (int k = 0; k < NC; ++k) out[(size_t)f * NC + k] = lerpf(w, cell[(size_t)o * NC + k], cell[(size_t)n * NC + k]);
    } else {
        const bool live = m.fkind[f] != 3;
        for (int k = 0; k < NC; ++k) out[(size_t)f * NC + k] = live ? bnd[(size_t)(f - m.nIF) * NC + k] : 0.0;
    }
}
void launchInterpolate(hipStream_t s, int NC, const MeshView& m, const double* cell, const double* bnd, double* out) {
    interpolateKernel<<<gridFor(m.nF), QGD_BLOCK, 0, s>>>(m, NC, cell, bnd, out);
}
// qgdFlux with a `Gauss upwind` entry [QGDInterpolate_8H L86-104 -> fvc::flux]: flux * (pos0(flux) (psi_O - psi_N) + psi_N), the patch
// value on patch faces (L0: upwind::weights, surfaceInterpolationScheme::interpolate)
__global__ __launch_bounds__(QGD_BLOCK) void fluxUpwindKernel(const MeshView m, const int NC, const double* __restrict__ flux,
                                                             const double* __restrict__ cell, const double* __restrict__ bnd,
                                                             double* __restrict__ out) {
    const int f = blockIdx.x * QGD_BLOCK + threadIdx.x;
    if (f >= m.nF) return;
    const double phi = flux[f];
    if (f < m.nIF) {
        const int o = m.own[f], n = m.nei[f];
        const double lambda = phi >= 0.0 ? 1.0 : 0.0;
        for (int k = 0; k < NC; ++k) out[(size_t)f * NC + k] = phi * lerpf(lambda, cell[(size_t)o * NC + k], cell[(size_t)n * NC + k]);
    } else {
        const bool live = m.fkind[f] != 3;
        for (int k = 0; k < NC; ++k) out[(size_t)f * NC + k] = live ? phi * bnd[(size_t)(f - m.nIF) * NC + k] : 0.0;
    }
}
void launchFluxUpwind(hipStream_t s, int NC, const MeshView& m, const double* flux, const double* cell, const double* bnd, double* out) {
    fluxUpwindKernel<<<gridFor(m.nF), QGD_BLOCK, 0, s>>>(m, NC, flux, cell, bnd, out);
}

template <int ST>
static void launchQhdT(hipStream_t s, const MeshView& m, const double* cell5, const double* bnd5, double* pt5, const double* rho,
                       const double* rhob, const double* tau, const double* phi, double beta, double gx, double gy, double gz,
                       double* out) {
    if (ST == ST_GVP3 || ST == ST_GVP2) {
        pointInterpKernel<5><<<gridFor(m.nP), QGD_BLOCK, 0, s>>>(m, cell5, 5, pt5);
        if (m.nBP) boundaryPointKernel<5><<<gridFor(m.nBP), QGD_BLOCK, 0, s>>>(m, bnd5, 5, pt5, 5, 0, ST == ST_GVP3 ? 0 : -1);
    }
    qhdFaceKernel<ST><<<gridFor(m.nF), QGD_BLOCK, 0, s>>>(m, cell5, bnd5, pt5, rho, rhob, tau, phi, beta, gx, gy, gz, out);
}
void launchQhdFluxes(hipStream_t s, int stencil, const MeshView& m, const double* cell5, const double* bnd5, double* pt5,
                     const double* rho, const double* rhob, const double* tau, const double* phi, double beta, double gx,
                     double gy, double gz, double* out) {
    switch (stencil) {
        case ST_REDUCED: launchQhdT<ST_REDUCED>(s, m, cell5, bnd5, pt5, rho, rhob, tau, phi, beta, gx, gy, gz, out); break;
        case ST_LSQ: launchQhdT<ST_LSQ>(s, m, cell5, bnd5, pt5, rho, rhob, tau, phi, beta, gx, gy, gz, out); break;
        case ST_GVP3: launchQhdT<ST_GVP3>(s, m, cell5, bnd5, pt5, rho, rhob, tau, phi, beta, gx, gy, gz, out); break;
        default: launchQhdT<ST_GVP2>(s, m, cell5, bnd5, pt5, rho, rhob, tau, phi, beta, gx, gy, gz, out); break;
    }
}

// ---------------------------------------------------------------------------
// Species block [reactingLagrangianQGDFoam/updateFluxes.H L117-132]: one stencil walk for fvsc::grad(Y), the two
// interpolations and the regularised species flux.  out: phiJmY, diffusiveFlux, gradYf(3) as SoA slots of nF.
// ---------------------------------------------------------------------------
template <int ST>
__global__ __launch_bounds__(QGD_BLOCK) void speciesFaceKernel(const MeshView m, const double* __restrict__ Yc,
                                                              const double* __restrict__ Yb, const double* __restrict__ ptY,
                                                              const double* __restrict__ Uc, const double* __restrict__ Ub,
                                                              const double* __restrict__ phiJm, const double* __restrict__ phi,
                                                              const double* __restrict__ tauF, double* __restrict__ out) {
    const int f = blockIdx.x * QGD_BLOCK + threadIdx.x;
    if (f >= m.nF) return;
    const size_t nF = (size_t)m.nF;
    if (m.fkind[f] == 3) {  // empty patches carry no field
        for (int k = 0; k < 5; ++k) out[(size_t)k * nF + f] = 0.0;
        return;
    }
    const bool internal = f < m.nIF;
    const int o = m.own[f];
    FaceVals<1> v;
    v.o[0] = Yc[o];
    double Uf[3], Yf;
    if (internal) {
        const int n = m.nei[f];
        v.n[0] = Yc[n];
        v.sn[0] = 0.0;
        const double w = m.w[f];
        Yf = lerpf(w, v.o[0], v.n[0]);
#pragma unroll
        for (int k = 0; k < 3; ++k) Uf[k] = lerpf(w, Uc[3 * (size_t)o + k], Uc[3 * (size_t)n + k]);
    } else {
        const int b = f - m.nIF;
        v.n[0] = Yb[b];
        v.sn[0] = m.dn[f] * (v.n[0] - v.o[0]);  // fvPatchField::snGrad (L0)
        Yf = v.n[0];
#pragma unroll
        for (int k = 0; k < 3; ++k) Uf[k] = Ub[3 * (size_t)b + k];
    }
    double g[3];
    faceGradient<ST, 1, -1>(m, f, v, Yc, ptY, g);
    const double dydt = (-phi[f]) * tauF[f] * (Uf[0] * g[0] + Uf[1] * g[1] + Uf[2] * g[2]);  // L124-125
    out[f] = phiJm[f] * Yf + dydt;                                                             // L123, L126
    out[nF + f] = dydt;                                                                        // L127
#pragma unroll
    for (int k = 0; k < 3; ++k) out[(size_t)(2 + k) * nF + f] = g[k];
}

// QGDYEqn.H L67-86, one species, explicit branch.  Face pass: the explicit laplacian flux (muf/Sc) snGrad(Yi) |Sf| (uncorrected snGrad, L0),
// added to diffusiveFlux as L82 does, and the net flux phiJmYi - that of the cell pass.  Empty faces carry nothing.
__global__ __launch_bounds__(QGD_BLOCK) void speciesLapKernel(const MeshView m, const double* __restrict__ Yc, const double* __restrict__ Yb,
                                                             const double* __restrict__ phiJmY, const double* __restrict__ muf, const double Sc,
                                                             double* __restrict__ diffusiveFlux, double* __restrict__ net) {
    const int f = blockIdx.x * QGD_BLOCK + threadIdx.x;
    if (f >= m.nF) return;
    if (m.fkind[f] == 3) { net[f] = 0.0; return; }   // (a shard's cut faces touch ghost cells only, whose values the caller refreshes)
    const double yo = Yc[m.own[f]];
    const double sn = f < m.nIF ? m.dn[f] * (Yc[m.nei[f]] - yo) : m.dn[f] * (Yb[f - m.nIF] - yo);
    const double lap = (muf[f] / Sc) * sn * m.magSf[f];
    diffusiveFlux[f] += lap;
    net[f] = phiJmY[f] - lap;
}
// cell pass: fvm::ddt(rho,Yi) + fvc::div(net) == Su (Euler), the divergence gathered in ascending face label; Yi.max(0)
__global__ __launch_bounds__(QGD_BLOCK) void speciesCellKernel(const MeshView m, const double* __restrict__ net, const double* __restrict__ Yc,
                                                              const double* __restrict__ rhoOld, const double* __restrict__ rho, const double dt,
                                                              const double* __restrict__ Su, double* __restrict__ Ynew) {
    const int c = blockIdx.x * QGD_BLOCK + threadIdx.x;
    if (c >= m.nC) return;
    const int n = m.cfCount[c];
    const size_t base = (size_t)m.cfSlice[c >> 6] * 64 + (c & 63);
    double s = 0.0;
    for (int i = 0; i < n; ++i) {
        const int it = m.cfItem[base + (size_t)i * 64];
        const double x = net[it >= 0 ? it : ~it];
        s = it >= 0 ? s + x : s - x;
    }
    const double V = m.V[c], rDeltaT = 1.0 / dt;
    double src = rDeltaT * rhoOld[c] * Yc[c] * V - s;
    if (Su) src += V * Su[c];
    Ynew[c] = fmax(src / (rDeltaT * rho[c] * V), 0.0);
}
// MeshView::geoPos: the face geometry fvc::grad(U) per cell needs, at the faces' slot-major positions
__global__ __launch_bounds__(QGD_BLOCK) void faceGeoPosKernel(const MeshView m, double4* __restrict__ out) {
    const int f = blockIdx.x * QGD_BLOCK + threadIdx.x;
    if (f >= m.nF) return;
    const size_t pos = f < m.nIF ? (size_t)m.fpos[f] : (size_t)f;
    out[pos] = make_double4(m.fkind[f] == 3 ? -1.0 : m.w[f], m.Sx[f], m.Sy[f], m.Sz[f]);
}
void launchFaceGeoPos(hipStream_t s, const MeshView& m, double4* out) {
    if (m.nF) faceGeoPosKernel<<<gridFor(m.nF), QGD_BLOCK, 0, s>>>(m, out);
}
void launchSpeciesStep(hipStream_t s, const MeshView& m, const double* Yc, const double* Yb, const double* rhoOld, const double* rho,
                       const double* phiJmY, const double* muf, double Sc, double dt, const double* Su, double* diffusiveFlux, double* net,
                       double* Ynew) {
    if (m.nF) speciesLapKernel<<<gridFor(m.nF), QGD_BLOCK, 0, s>>>(m, Yc, Yb, phiJmY, muf, Sc, diffusiveFlux, net);
    if (m.nC) speciesCellKernel<<<gridFor(m.nC), QGD_BLOCK, 0, s>>>(m, net, Yc, rhoOld, rho, dt, Su, Ynew);
}

template <int ST>
static void launchSpeciesT(hipStream_t s, const MeshView& m, const double* Y, const double* Yb, double* ptY, const double* U,
                           const double* Ub, const double* phiJm, const double* phi, const double* tau, double* out) {
    if (ST == ST_GVP3 || ST == ST_GVP2) {
        pointInterpKernel<1><<<gridFor(m.nP), QGD_BLOCK, 0, s>>>(m, Y, 1, ptY);
        if (m.nBP) boundaryPointKernel<1><<<gridFor(m.nBP), QGD_BLOCK, 0, s>>>(m, Yb, 1, ptY, 1, 0, -1);
    }
    speciesFaceKernel<ST><<<gridFor(m.nF), QGD_BLOCK, 0, s>>>(m, Y, Yb, ptY, U, Ub, phiJm, phi, tau, out);
}
void launchSpeciesFlux(hipStream_t s, int stencil, const MeshView& m, const double* Y, const double* Yb, double* ptY,
                       const double* U, const double* Ub, const double* phiJm, const double* phi, const double* tau, double* out) {
    switch (stencil) {
        case ST_REDUCED: launchSpeciesT<ST_REDUCED>(s, m, Y, Yb, ptY, U, Ub, phiJm, phi, tau, out); break;
        case ST_LSQ: launchSpeciesT<ST_LSQ>(s, m, Y, Yb, ptY, U, Ub, phiJm, phi, tau, out); break;
        case ST_GVP3: launchSpeciesT<ST_GVP3>(s, m, Y, Yb, ptY, U, Ub, phiJm, phi, tau, out); break;
        default: launchSpeciesT<ST_GVP2>(s, m, Y, Yb, ptY, U, Ub, phiJm, phi, tau, out); break;
    }
}

}  // namespace qgd
