// qgd_implicit.hip -- the implicitDiffusion branch of QGDFoam (the reference's default, QGDThermo_8C_source.html L70-82):
//
//   updateFluxes.H L107-111   tauMC = qgdInterpolate(muEff*dev2(T(fvc::grad(U)))), phiTauMC = Sf & tauMC
//   QGDUEqn.H L54-75          UEqn: fvm::ddt(rho,U) - fvc::ddt(rho,U) - fvm::laplacian(muf,U) - fvc::div(phiTauMC); rhoU = rho*U;
//                             sigmaDotU = (muf*lin(fvc::grad(U)) + tauMC) & Uf; phiSigmaDotU = Sf & sigmaDotU
//   QGDEEqn.H L37-64          EEqn with -fvc::div(phiSigmaDotU); fvm::ddt(rho,e) - fvc::ddt(rho,e) - fvm::laplacian(alphauf,e)
//
// The face-flux kernels of qgd_kernels.hip run with GasModel::implicitDiffusion = 1 (Pi without its Navier-Stokes part, q
// without its Fourier part); what is here sits between them and the boundary refresh.  L0 pieces: fvc::grad (Gauss linear +
// gaussGrad::correctBoundaryConditions), fvm::laplacian (Gauss, uncorrected snGrad; patch coefficients of fixedValue,
// zeroGradient and basicSymmetry patches), segregated component solves.  Not a benchmark path: generic one-thread-per-item
// kernels, the two linear solves by the reproducible Jacobi-PCG of qgd_poisson.hip.
#include "../../include/qgd_amd.h"
#include "qgd_device.hpp"
#include "qgd_stencil_dev.hpp"

namespace qgd {

namespace {

// patch snGrad of U on boundary face f from the owner's and the patch's velocity
__device__ __forceinline__ void patchSnGradU(const MeshView& m, const PatchBCDev& bc, const int f, const double uo[3], const double ub[3],
                                             double sn[3]) {
    const double dc = m.dn[f];
    if (bc.bcU == QGD_BC_FIXEDVALUE) { for (int k = 0; k < 3; ++k) sn[k] = dc * (ub[k] - uo[k]); }
    else if (bc.bcU == QGD_BC_SLIP) {
        const double ms = m.magSf[f];
        const double n[3] = {m.Sx[f] / ms, m.Sy[f] / ms, m.Sz[f] / ms};
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
    const double tr = g[0] + g[4] + g[8];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double a = g[3 * j + i];
            if (i == j) a = a - (2.0 / 3.0) * tr;
            out[3 * i + j] = mu * a;
        }
}

// fvc::grad(U), Gauss linear: cell gather in ascending face order
__global__ __launch_bounds__(QGD_BLOCK) void implCellGradKernel(const MeshView m, const CaseView c, const ImplView iv) {
    const int ci = blockIdx.x * QGD_BLOCK + threadIdx.x;
    if (ci >= m.nC) return;
    const int n = m.cfCount[ci];
    const size_t base = (size_t)m.cfSlice[ci >> 6] * 64 + (ci & 63);
    double G[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < n; ++i) {
        const int it = m.cfItem[base + (size_t)i * 64];
        const int f = it >= 0 ? it : ~it;
        if (m.fkind[f] == 3) continue;
        double Uf[3];
        if (f < m.nIF) {
            const RecA a = c.A[m.own[f]], b = c.A[m.nei[f]];
            const double w = m.w[f];
            Uf[0] = lerpf(w, a.ux, b.ux); Uf[1] = lerpf(w, a.uy, b.uy); Uf[2] = lerpf(w, a.uz, b.uz);
        } else {
            const RecA b = c.bA[f - m.nIF];
            Uf[0] = b.ux; Uf[1] = b.uy; Uf[2] = b.uz;
        }
        const double S[3] = {m.Sx[f], m.Sy[f], m.Sz[f]};
        for (int a = 0; a < 3; ++a)
            for (int j = 0; j < 3; ++j) G[3 * a + j] = it >= 0 ? G[3 * a + j] + S[a] * Uf[j] : G[3 * a + j] - S[a] * Uf[j];
    }
    const double V = m.V[ci];
    for (int k = 0; k < 9; ++k) iv.gUc[(size_t)ci * 9 + k] = G[k] / V;
}

// per face: muf, alphauf, Uf, tauMC -> phiTauMC, Sf.(tauMC & Uf), the laplacian coefficients [updateFluxes.H L107-111]
__global__ __launch_bounds__(QGD_BLOCK) void implFaceKernel(const MeshView m, const CaseView c, const ImplView iv, const GasModel gm,
                                                           const PatchBCDev* __restrict__ bcs) {
    const int f = blockIdx.x * QGD_BLOCK + threadIdx.x;
    if (f >= m.nF) return;
    const size_t nF = (size_t)m.nF;
    if (m.fkind[f] == 3) {
        for (int k = 0; k < 3; ++k) { iv.phiTau[(size_t)k * nF + f] = 0.0; iv.UfS[(size_t)k * nF + f] = 0.0; }
        iv.sTau[f] = iv.mufS[f] = iv.aU[f] = iv.aE[f] = 0.0;
        return;
    }
    const int o = m.own[f];
    const RecA Ao = c.A[o];
    const RecB Bo = c.B[o];
    double muf, alf, Uf[3], tau[9];
    if (f < m.nIF) {
        const int n = m.nei[f];
        const RecA An = c.A[n];
        const RecB Bn = c.B[n];
        const double w = m.w[f];
        muf = lerpf(w, muEffOf(gm, Bo.muQGD), muEffOf(gm, Bn.muQGD));
        alf = lerpf(w, alphaEffOf(gm, Bo.muQGD), alphaEffOf(gm, Bn.muQGD));
        Uf[0] = lerpf(w, Ao.ux, An.ux); Uf[1] = lerpf(w, Ao.uy, An.uy); Uf[2] = lerpf(w, Ao.uz, An.uz);
        double to[9], tn[9];
        muDev2T(iv.gUc + (size_t)o * 9, muEffOf(gm, Bo.muQGD), to);
        muDev2T(iv.gUc + (size_t)n * 9, muEffOf(gm, Bn.muQGD), tn);
        for (int k = 0; k < 9; ++k) tau[k] = lerpf(w, to[k], tn[k]);
    } else {
        const int b = f - m.nIF;
        const PatchBCDev bc = bcs[m.bPatch[b]];
        const RecA Ab = c.bA[b];
        const RecB Bb = c.bB[b];
        muf = muEffOf(gm, Bb.muQGD);
        alf = alphaEffOf(gm, Bb.muQGD);
        Uf[0] = Ab.ux; Uf[1] = Ab.uy; Uf[2] = Ab.uz;
        const double uo[3] = {Ao.ux, Ao.uy, Ao.uz};
        double sn[3], gb[9];
        patchSnGradU(m, bc, f, uo, Uf, sn);
        patchGradU(m, bc, f, iv.gUc + (size_t)o * 9, sn, gb);
        muDev2T(gb, muf, tau);
    }
    const double S[3] = {m.Sx[f], m.Sy[f], m.Sz[f]};
    double tU[3];
    for (int i = 0; i < 3; ++i) tU[i] = tau[3 * i] * Uf[0] + tau[3 * i + 1] * Uf[1] + tau[3 * i + 2] * Uf[2];   // tauMC & Uf
    for (int j = 0; j < 3; ++j) {
        iv.phiTau[(size_t)j * nF + f] = S[0] * tau[j] + S[1] * tau[3 + j] + S[2] * tau[6 + j];                  // Sf & tauMC
        iv.UfS[(size_t)j * nF + f] = Uf[j];
    }
    iv.sTau[f] = S[0] * tU[0] + S[1] * tU[1] + S[2] * tU[2];
    iv.mufS[f] = muf;
    const double gsd = m.magSf[f] * m.dn[f];   // |Sf| * (nonOrthDeltaCoeffs inside, deltaCoeffs on patches)
    iv.aU[f] = muf * gsd;
    iv.aE[f] = alf * gsd;
}

// QGDRhoEqn.H, the first solve of QGDUEqn.H (rhoU), U = rhoU/rho, and the matrix + source of UEqn per component
__global__ __launch_bounds__(QGD_BLOCK) void implCellUKernel(const MeshView m, const CaseView c, const ImplView iv, const PatchBCDev* __restrict__ bcs) {
    const int ci = blockIdx.x * QGD_BLOCK + threadIdx.x;
    if (ci >= m.nC) return;
    const int n = m.cfCount[ci];
    const size_t base = (size_t)m.cfSlice[ci >> 6] * 64 + (ci & 63);
    const size_t nF = (size_t)m.nF, nC = (size_t)m.nC;
    double sum[4] = {0, 0, 0, 0}, dTau[3] = {0, 0, 0}, diagBase = 0;
    for (int i = 0; i < n; ++i) {
        const int it = m.cfItem[base + (size_t)i * 64];
        const int f = it >= 0 ? it : ~it;
        const size_t pos = f < m.nIF ? (size_t)m.fpos[f] : (size_t)f;
        for (int k = 0; k < 4; ++k) { const double x = c.flux[(size_t)k * nF + pos]; sum[k] = it >= 0 ? sum[k] + x : sum[k] - x; }
        for (int k = 0; k < 3; ++k) { const double x = iv.phiTau[(size_t)k * nF + f]; dTau[k] = it >= 0 ? dTau[k] + x : dTau[k] - x; }
        if (f < m.nIF) diagBase += iv.aU[f];
    }
    const RecA A = c.A[ci];
    const double V = m.V[ci], dt = c.dt[0], dtV = dt / V, rDeltaT = 1.0 / dt;
    const double rho = A.rho - dtV * sum[0];
    const double uo[3] = {A.ux, A.uy, A.uz};
    double Ucur[3];
    for (int k = 0; k < 3; ++k) Ucur[k] = (A.rho * uo[k] - dtV * sum[1 + k]) / rho;   // rhoU/rho [QGDUEqn.H L36-50]
    double diag[3], rhs[3];
    for (int k = 0; k < 3; ++k) {
        diag[k] = rDeltaT * rho * V + diagBase;
        rhs[k] = rDeltaT * rho * Ucur[k] * V + dTau[k];   // fvm::ddt(rho,U) - fvc::ddt(rho,U) - fvc::div(phiTauMC) [L58-60]
    }
    // patch coefficients of -fvm::laplacian(muf, U) (L0): fixedValue: delta / delta*value; basicSymmetry: delta*|n_k| /
    // snGrad_k + delta*|n_k|*patchInternalField_k (transformFvPatchField); zeroGradient: none
    for (int i = 0; i < n; ++i) {
        const int it = m.cfItem[base + (size_t)i * 64];
        if (it < m.nIF) continue;   // owner-side boundary faces only (it >= nIF implies it >= 0)
        const int f = it, b = f - m.nIF;
        if (m.fkind[f] == 3) continue;
        const PatchBCDev bc = bcs[m.bPatch[b]];
        if (bc.ptype == QGD_PATCH_HALO || bc.ptype == QGD_PATCH_CYCLIC) continue;
        const double a = iv.aU[f];
        if (bc.bcU == QGD_BC_FIXEDVALUE) {
            for (int k = 0; k < 3; ++k) { diag[k] += a; rhs[k] += a * bc.vU[k]; }
        } else if (bc.bcU == QGD_BC_SLIP) {
            const double ms = m.magSf[f], dc = m.dn[f], gs = iv.mufS[f] * ms;
            const double nv[3] = {m.Sx[f] / ms, m.Sy[f] / ms, m.Sz[f] / ms};
            double sn[3];
            patchSnGradU(m, bc, f, Ucur, Ucur, sn);
            for (int k = 0; k < 3; ++k) { diag[k] += a * fabs(nv[k]); rhs[k] += gs * (sn[k] + dc * fabs(nv[k]) * Ucur[k]); }
        }
    }
    iv.rhoNew[ci] = rho;
    for (int k = 0; k < 3; ++k) { iv.xU[(size_t)k * nC + ci] = Ucur[k]; iv.diagU[(size_t)k * nC + ci] = diag[k]; iv.rhsU[(size_t)k * nC + ci] = rhs[k]; }
}

// after the U solve: rho and U of the records (p and e stay those of the old time level), patch values of U
__global__ __launch_bounds__(QGD_BLOCK) void implStoreUKernel(const MeshView m, const CaseView c, const ImplView iv) {
    const int ci = blockIdx.x * QGD_BLOCK + threadIdx.x;
    if (ci >= m.nC) return;
    const size_t nC = (size_t)m.nC;
    RecA a = c.A[ci];
    a.rho = iv.rhoNew[ci];
    a.ux = iv.xU[ci]; a.uy = iv.xU[nC + ci]; a.uz = iv.xU[2 * nC + ci];
    c.A[ci] = a;
}
__global__ __launch_bounds__(QGD_BLOCK) void implBcUKernel(const MeshView m, const CaseView c, const PatchBCDev* __restrict__ bcs) {
    const int b = blockIdx.x * QGD_BLOCK + threadIdx.x;
    if (b >= m.nBF) return;
    const int f = m.nIF + b;
    if (m.fkind[f] == 3) return;
    const PatchBCDev bc = bcs[m.bPatch[b]];
    if (bc.ptype == QGD_PATCH_HALO) return;
    const RecA Ao = c.A[m.own[f]];
    RecA Ab = c.bA[b];
    if (bc.bcU == QGD_BC_FIXEDVALUE) { Ab.ux = bc.vU[0]; Ab.uy = bc.vU[1]; Ab.uz = bc.vU[2]; }
    else if (bc.bcU == QGD_BC_SLIP) {
        const double ms = m.magSf[f];
        const double n[3] = {m.Sx[f] / ms, m.Sy[f] / ms, m.Sz[f] / ms};
        const double u[3] = {Ao.ux, Ao.uy, Ao.uz};
        double r[3];
        for (int i = 0; i < 3; ++i) {
            const double tv = ((i == 0 ? 1.0 : 0.0) - 2.0 * (n[i] * n[0])) * u[0] + ((i == 1 ? 1.0 : 0.0) - 2.0 * (n[i] * n[1])) * u[1] +
                              ((i == 2 ? 1.0 : 0.0) - 2.0 * (n[i] * n[2])) * u[2];
            r[i] = (u[i] + tv) / 2.0;
        }
        Ab.ux = r[0]; Ab.uy = r[1]; Ab.uz = r[2];
    } else { Ab.ux = Ao.ux; Ab.uy = Ao.uy; Ab.uz = Ao.uz; }
    c.bA[b] = Ab;
}

// phiSigmaDotU = Sf & ((muf*lin(fvc::grad(U)) + tauMC) & Uf) with the new U's gradient [QGDUEqn.H L72-74]
__global__ __launch_bounds__(QGD_BLOCK) void implSigmaKernel(const MeshView m, const CaseView c, const ImplView iv, const PatchBCDev* __restrict__ bcs) {
    const int f = blockIdx.x * QGD_BLOCK + threadIdx.x;
    if (f >= m.nF) return;
    if (m.fkind[f] == 3) { iv.phiSig[f] = 0.0; return; }
    const size_t nF = (size_t)m.nF;
    const int o = m.own[f];
    double g[9];
    if (f < m.nIF) {
        const int n = m.nei[f];
        const double w = m.w[f];
        for (int k = 0; k < 9; ++k) g[k] = lerpf(w, iv.gUc[(size_t)o * 9 + k], iv.gUc[(size_t)n * 9 + k]);
    } else {
        const int b = f - m.nIF;
        const PatchBCDev bc = bcs[m.bPatch[b]];
        const RecA Ao = c.A[o], Ab = c.bA[b];
        const double uo[3] = {Ao.ux, Ao.uy, Ao.uz}, ub[3] = {Ab.ux, Ab.uy, Ab.uz};
        double sn[3];
        patchSnGradU(m, bc, f, uo, ub, sn);
        patchGradU(m, bc, f, iv.gUc + (size_t)o * 9, sn, g);
    }
    const double Uf[3] = {iv.UfS[f], iv.UfS[nF + f], iv.UfS[2 * nF + f]};
    const double S[3] = {m.Sx[f], m.Sy[f], m.Sz[f]};
    double gu[3];
    for (int i = 0; i < 3; ++i) gu[i] = g[3 * i] * Uf[0] + g[3 * i + 1] * Uf[1] + g[3 * i + 2] * Uf[2];
    iv.phiSig[f] = iv.mufS[f] * (S[0] * gu[0] + S[1] * gu[1] + S[2] * gu[2]) + iv.sTau[f];
}

// EEqn [QGDEEqn.H L37-50] and the matrix + source of the e equation [L55-61]
__global__ __launch_bounds__(QGD_BLOCK) void implCellEKernel(const MeshView m, const CaseView c, const ImplView iv, const GasModel gm,
                                                            const PatchBCDev* __restrict__ bcs) {
    const int ci = blockIdx.x * QGD_BLOCK + threadIdx.x;
    if (ci >= m.nC) return;
    const int n = m.cfCount[ci];
    const size_t base = (size_t)m.cfSlice[ci >> 6] * 64 + (ci & 63);
    const size_t nF = (size_t)m.nF;
    double sum = 0, diag = 0, rhs = 0;
    for (int i = 0; i < n; ++i) {
        const int it = m.cfItem[base + (size_t)i * 64];
        const int f = it >= 0 ? it : ~it;
        const size_t pos = f < m.nIF ? (size_t)m.fpos[f] : (size_t)f;
        const double x = c.flux[4 * nF + pos] - iv.phiSig[f];   // phiJmH + phiQ - phiPiU - phiSigmaDotU
        sum = it >= 0 ? sum + x : sum - x;
        if (f < m.nIF) diag += iv.aE[f];
        else if (m.fkind[f] != 3) {
            const PatchBCDev bc = bcs[m.bPatch[f - m.nIF]];
            if (bc.ptype != QGD_PATCH_HALO && bc.ptype != QGD_PATCH_CYCLIC && bc.bcT == QGD_BC_FIXEDVALUE) {
                diag += iv.aE[f];                       // fixedEnergy = fixedValue
                rhs += iv.aE[f] * (gm.Cv * bc.vT);
            }
        }
    }
    const RecA A = c.A[ci];   // rho, U of the new time level
    const double V = m.V[ci], dt = c.dt[0], rDeltaT = 1.0 / dt;
    const double rE = c.rE[ci] - (dt / V) * sum;
    const double ecur = rE / A.rho - 0.5 * (A.ux * A.ux + A.uy * A.uy + A.uz * A.uz);   // [L49]
    iv.xE[ci] = ecur;
    iv.diagE[ci] = rDeltaT * A.rho * V + diag;
    iv.rhsE[ci] = rDeltaT * A.rho * ecur * V + rhs;   // fvm::ddt(rho,e) - fvc::ddt(rho,e) [L57]
}

// rhoE = rho*(e + |U|^2/2) [L63], thermo.correct(), p = rho/psi [QGDFoam.C L149-154]
__global__ __launch_bounds__(QGD_BLOCK) void implFinishKernel(const MeshView m, const CaseView c, const ImplView iv, const GasModel gm) {
    const int ci = blockIdx.x * QGD_BLOCK + threadIdx.x;
    double rmin = 1e300, emin = 1e300;
    if (ci < m.nC) {
        RecA A = c.A[ci];
        const double pOld = A.p;
        A.e = iv.xE[ci];
        const double rE = A.rho * (A.e + 0.5 * (A.ux * A.ux + A.uy * A.uy + A.uz * A.uz));
        const double T = A.e / gm.Cv;
        const double psi = 1.0 / (gm.R * T);
        const double cs = sqrt(gm.gamma / psi);
        const double aq = c.aQ ? c.aQ[ci] : gm.alphaQGD, scq = c.sc ? c.sc[ci] : gm.ScQGD;
        const double tauQGD = aq * m.hQGD[ci] / cs;
        RecB B;
        B.muQGD = pOld * scq * tauQGD;   // the pressure constScPrModel1 sees is still the old one [QGDFoam.C L149-154]
        B.c = cs;
        B.aOc = aq / cs;
        A.p = A.rho / psi;
        B.H = (rE + A.p) / A.rho;
        c.A[ci] = A; c.B[ci] = B; c.rE[ci] = rE;
        rmin = (A.rho == A.rho) ? A.rho : -1e300;
        emin = (A.e == A.e) ? A.e : -1e300;
    }
    blockMaxMin<QGD_BLOCK>(-rmin, emin, c.blkCell + 2 * (size_t)blockIdx.x, true);
}

inline int gridOf(int64_t n) { return (int)((n + QGD_BLOCK - 1) / QGD_BLOCK); }

}  // namespace

// One implicit-diffusion advance after the flux assembly.  work: 6*nC + 3*ceil(nC/256) + 8 doubles.  iters[0..2] = U
// components, iters[3] = e; resid[2k], resid[2k+1] = initial and final normalised residual of solve k.
void launchImplicitAdvance(hipStream_t s, const MeshView& m, const CaseView& c, const ImplView& iv, const GasModel& g, const PatchBCDev* bc,
                           double tol, int maxIter, double* work, int iters[4], double resid[8]) {
    const int gc = gridOf(m.nC), gf = gridOf(m.nF), gb = gridOf(m.nBF);
    const size_t nC = (size_t)m.nC;
    double res[2];
    implCellGradKernel<<<gc, QGD_BLOCK, 0, s>>>(m, c, iv);
    implFaceKernel<<<gf, QGD_BLOCK, 0, s>>>(m, c, iv, g, bc);
    implCellUKernel<<<gc, QGD_BLOCK, 0, s>>>(m, c, iv, bc);
    for (int k = 0; k < 3; ++k) {
        iters[k] = 0;
        resid[2 * k] = resid[2 * k + 1] = 0.0;
        const bool valid = !(m.nGeomD < 3 && m.emptyDir[k]);   // validComponents: empty directions are not solved (L0)
        if (!valid) continue;
        iters[k] = diagLaplacianPcg(s, m, iv.aU, iv.diagU + k * nC, iv.rhsU + k * nC, iv.xU + k * nC, work, tol, maxIter, res);
        resid[2 * k] = res[0]; resid[2 * k + 1] = res[1];
    }
    implStoreUKernel<<<gc, QGD_BLOCK, 0, s>>>(m, c, iv);
    if (m.nBF) implBcUKernel<<<gb, QGD_BLOCK, 0, s>>>(m, c, bc);
    implCellGradKernel<<<gc, QGD_BLOCK, 0, s>>>(m, c, iv);
    implSigmaKernel<<<gf, QGD_BLOCK, 0, s>>>(m, c, iv, bc);
    implCellEKernel<<<gc, QGD_BLOCK, 0, s>>>(m, c, iv, g, bc);
    iters[3] = diagLaplacianPcg(s, m, iv.aE, iv.diagE, iv.rhsE, iv.xE, work, tol, maxIter, res);
    resid[6] = res[0]; resid[7] = res[1];
    implFinishKernel<<<gc, QGD_BLOCK, 0, s>>>(m, c, iv, g);
}

}  // namespace qgd
