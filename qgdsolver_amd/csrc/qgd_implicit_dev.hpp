// qgd_implicit_dev.hpp -- device helpers of the implicitDiffusion branch [QGDUEqn.H L54-75, QGDEEqn.H L53-64] shared by its own kernels
// (qgd_implicit.hip) and by the block-fused assembly of the U systems (qgd_kernels.hip fusedFaceCellKernel<..., IMPL = true>): the same
// inlined expressions in both, so that the two paths agree bit for bit.
#pragma once
#include "qgd_device.hpp"
#include "qgd_stencil_dev.hpp"

namespace qgd {

// patch snGrad of U on boundary face f from the owner's and the patch's velocity
__device__ __forceinline__ void patchSnGradU(const MeshView& m, const PatchBCDev& bc, const int f, const double uo[3], const double ub[3],
                                             double sn[3]) {
#pragma clang fp contract(off)   // (no fusing left to the compiler: inlined into different kernels it would fuse different products, see implInternalFace)
    const double dc = m.dn[f];
    if (bc.bcU == QGD_BC_FIXEDVALUE) { for (int k = 0; k < 3; ++k) sn[k] = dc * (ub[k] - uo[k]); }
    else if (bc.bcU == QGD_BC_SLIP) {
        double n[3];
        symmNormal(m, bc, f, n);
        for (int i = 0; i < 3; ++i) {
            const double tv = ((i == 0 ? 1.0 : 0.0) - 2.0 * (n[i] * n[0])) * uo[0] + ((i == 1 ? 1.0 : 0.0) - 2.0 * (n[i] * n[1])) * uo[1] +
                              ((i == 2 ? 1.0 : 0.0) - 2.0 * (n[i] * n[2])) * uo[2];
            sn[i] = (tv - uo[i]) * (dc / 2.0);
        }
    } else { sn[0] = sn[1] = sn[2] = 0.0; }
}
// patch value of fvc::grad(U): the owner's gradient with its normal part replaced by the patch snGrad (L0)
__device__ __forceinline__ void patchGradU(const MeshView& m, const PatchBCDev& bc, const int f, const double* gOwner, const double sn[3],
                                           double gb[9]) {
    for (int k = 0; k < 9; ++k) gb[k] = gOwner[k];
    if (bc.ptype == QGD_PATCH_HALO || bc.ptype == QGD_PATCH_CYCLIC) return;
    const double ms = m.magSf[f];
    const double n[3] = {m.Sx[f] / ms, m.Sy[f] / ms, m.Sz[f] / ms};
    double ng[3];
    for (int j = 0; j < 3; ++j) ng[j] = n[0] * gb[j] + n[1] * gb[3 + j] + n[2] * gb[6 + j];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) gb[3 * i + j] += n[i] * (sn[j] - ng[j]);
}
// mu * dev2(T(g)):  dev2(A) = A - (2/3) tr(A) I
__device__ __forceinline__ void muDev2T(const double* g, const double mu, double out[9]) {
#pragma clang fp contract(off)   // (no fusing left to the compiler: inlined into different kernels it would fuse different products, see implInternalFace)
    const double tr = g[0] + g[4] + g[8];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double a = g[3 * j + i];
            if (i == j) a = a - (2.0 / 3.0) * tr;
            out[3 * i + j] = mu * a;
        }
}

// ---- start values of the two solves -------------------------------------------------------------------------------------------------------
// OpenFOAM starts a solve from the field as it stands: the predictor U = rhoU/rho [QGDUEqn.H L48-50], e = rhoE/rho - |U|^2/2 [QGDEEqn.H L49]
// (QGD_IMPL_XEXTRAP=0).  What the solve adds to the predictor -- the implicit part of the viscous / conductive update -- changes slowly from
// step to step, so the default starts from predictor + the correction of the steps before extrapolated in time (=1: the last one, =2:
// 2 d1 - d2, =3, default: 3 d1 - 3 d2 + d3, =4: 4 d1 - 6 d2 + 4 d3 - d4 -- no further gain): the same system, the same right-hand side, the same tolerance, a first residual smaller by
// orders of magnitude and correspondingly fewer Chebyshev steps (profiles/r05_ab_implicit_start_values.txt).  A solver-internal choice like
// the pressure solve's (qgd_qhd.hip qhdExtrapolatePKernel): the answer is the same to the solve's tolerance, the "Initial residual" of the
// log is not.  Ghost columns of a shard receive their neighbours' start values with message kind 4 as before.
// The extrapolated correction is LIMITED to twice the last one per value (see qhdExtrapolatePKernel): where the history is not smooth the start
// value falls back towards the predictor.
__device__ __forceinline__ double startValue(const ImplView& iv, const size_t j, const double pred) {
    if (iv.pred == nullptr) return pred;
    iv.pred[j] = pred;
    const int k = iv.have < iv.order ? iv.have : iv.order;
    if (k == 0) return pred;
    const double d1 = iv.dh[0][j];
    double e = d1;
    if (iv.w != nullptr) {   // adjustTimeStep: steps of different length (implStartWeightsKernel)
        e = iv.w[0] * d1;
        if (k >= 2) e += iv.w[1] * iv.dh[1][j];
        if (k >= 3) e += iv.w[2] * iv.dh[2][j];
        if (k >= 4) e += iv.w[3] * iv.dh[3][j];
    } else if (k == 2) e = 2.0 * d1 - iv.dh[1][j];
    else if (k == 3) e = (3.0 * d1 - 3.0 * iv.dh[1][j]) + iv.dh[2][j];
    else if (k >= 4) e = ((4.0 * d1 - 6.0 * iv.dh[1][j]) + 4.0 * iv.dh[2][j]) - iv.dh[3][j];
    const double lim = 2.0 * fabs(d1);
    return pred + fmin(fmax(e, -lim), lim);
}
// after a solve: this step's correction into the oldest slot (the host rotates the pointers at the end of the step)
__device__ __forceinline__ void keepCorrection(const ImplView& iv, const size_t j, const double solved) {
    if (iv.pred != nullptr) iv.dh[iv.order - 1][j] = solved - iv.pred[j];
}

// ---- one internal face of the branch [updateFluxes.H L95-111]: muf, alphauf, Uf, tauMC = lin(muEff dev2(T(grad U))) -> phiTauMC = Sf & tauMC,
// Sf.(tauMC & Uf), the laplacian coefficients of the U and e systems.  uo / un: the two cells' velocity, gO / gN: their fvc::grad(U) ----
struct ImplFaceOut { double phiTau[3], Uf[3], sTau, muf, aU, aE; };
__device__ __forceinline__ void implInternalFace(const GasModel& gm, const double w, const double muQo, const double muQn, const double uo[3],
                                                 const double un[3], const double* gO, const double* gN, const double S[3], const double gsd,
                                                 ImplFaceOut& o) {
#pragma clang fp contract(off)   // (no fusing left to the compiler: inlined into different kernels it would fuse different products, see implInternalFace)
    const double muf = lerpf(w, muEffOf(gm, muQo), muEffOf(gm, muQn));
    const double alf = lerpf(w, alphaEffOf(gm, muQo), alphaEffOf(gm, muQn));
    o.Uf[0] = lerpf(w, uo[0], un[0]); o.Uf[1] = lerpf(w, uo[1], un[1]); o.Uf[2] = lerpf(w, uo[2], un[2]);
    double to[9], tn[9], tau[9];
    muDev2T(gO, muEffOf(gm, muQo), to);
    muDev2T(gN, muEffOf(gm, muQn), tn);
    for (int k = 0; k < 9; ++k) tau[k] = lerpf(w, to[k], tn[k]);
    // (the three-term products with their multiply-adds spelled out: left to the compiler's contraction, `a*b + c*d + e*f` may fuse one product
    // in one kernel and another in the other, and the two callers of this function would differ in the last bit)
    auto dot3 = [](double a0, double b0, double a1, double b1, double a2, double b2) { return fma(a2, b2, fma(a1, b1, a0 * b0)); };
    double tU[3];
    for (int i = 0; i < 3; ++i) tU[i] = dot3(tau[3 * i], o.Uf[0], tau[3 * i + 1], o.Uf[1], tau[3 * i + 2], o.Uf[2]);   // tauMC & Uf
    for (int j = 0; j < 3; ++j) o.phiTau[j] = dot3(S[0], tau[j], S[1], tau[3 + j], S[2], tau[6 + j]);                  // Sf & tauMC
    o.sTau = dot3(S[0], tU[0], S[1], tU[1], S[2], tU[2]);
    o.muf = muf;
    o.aU = muf * gsd;
    o.aE = alf * gsd;
}

// ---- one cell of QGDRhoEqn.H, the first solve of QGDUEqn.H (rhoU), U = rhoU/rho, and the matrix + source of UEqn per component, from the
// ordered sums over the cell's faces (net mass + momentum fluxes, phiTauMC, the laplacian coefficients of its internal faces); the patch
// coefficients of -fvm::laplacian(muf, U) come from the cell's boundary faces, handed over as `nPatchFaces` labels through `patchFace(i)` ----
template <class PatchFaceFn>
__device__ __forceinline__ void implCellU(const MeshView& m, const CaseView& c, const ImplView& iv, const PatchBCDev* __restrict__ bcs, const int ci,
                                          const RecA& A, const double V, const double sum[4], const double dTau[3], const double diagBase,
                                          const int nPatchFaces, PatchFaceFn patchFace) {
#pragma clang fp contract(off)   // (no fusing left to the compiler: inlined into different kernels it would fuse different products, see implInternalFace)
    const size_t nC = (size_t)m.nC;
    const double dt = c.dt[0], dtV = dt / V, rDeltaT = 1.0 / dt;
    const double rho = A.rho - dtV * sum[0];
    const double uo[3] = {A.ux, A.uy, A.uz};
    double Ucur[3];
    for (int k = 0; k < 3; ++k) Ucur[k] = fma(-dtV, sum[1 + k], A.rho * uo[k]) / rho;   // rhoU/rho [QGDUEqn.H L36-50] (one spelling of the multiply-add, see implInternalFace)
    double diag[3], rhs[3];
    for (int k = 0; k < 3; ++k) {
        diag[k] = rDeltaT * rho * V + diagBase;
        rhs[k] = rDeltaT * rho * Ucur[k] * V + dTau[k];   // fvm::ddt(rho,U) - fvc::ddt(rho,U) - fvc::div(phiTauMC) [L58-60]
    }
    // patch coefficients of -fvm::laplacian(muf, U) (L0): fixedValue: delta / delta*value; basicSymmetry: delta*|n_k| /
    // snGrad_k + delta*|n_k|*patchInternalField_k (transformFvPatchField); zeroGradient: none
    for (int i = 0; i < nPatchFaces; ++i) {
        const int f = patchFace(i);
        if (f < m.nIF) continue;
        const int b = f - m.nIF;
        if (m.fkind[f] == 3) continue;
        const PatchBCDev bc = bcs[m.bPatch[b]];
        if (bc.ptype == QGD_PATCH_HALO || bc.ptype == QGD_PATCH_CYCLIC) continue;
        const double a = iv.aU[f];
        if (bc.bcU == QGD_BC_FIXEDVALUE) {
            for (int k = 0; k < 3; ++k) { diag[k] += a; rhs[k] += a * bc.vU[k]; }
        } else if (bc.bcU == QGD_BC_SLIP) {
            const double ms = m.magSf[f], dc = m.dn[f], gs = iv.mufS[f] * ms;
            double nv[3];
            symmNormal(m, bc, f, nv);
            double sn[3];
            patchSnGradU(m, bc, f, Ucur, Ucur, sn);
            for (int k = 0; k < 3; ++k) { diag[k] += a * fabs(nv[k]); rhs[k] += gs * (sn[k] + dc * fabs(nv[k]) * Ucur[k]); }
        }
    }
    iv.rhoNew[ci] = rho;
    for (int k = 0; k < 3; ++k) {
        iv.xU[(size_t)k * nC + ci] = startValue(iv, (size_t)k * nC + ci, Ucur[k]);
        iv.diagU[(size_t)k * nC + ci] = diag[k]; iv.rhsU[(size_t)k * nC + ci] = rhs[k];
    }
}

}  // namespace qgd
