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
#include <algorithm>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/qgd_amd.h"
#include "qgd_device.hpp"
#include "qgd_stencil_dev.hpp"
#include "qgd_implicit_dev.hpp"

namespace qgd {

namespace {

// Under Courant-number control [setDeltaT-QGDQHD.H L41-61] the steps of the history differ in length and what a solve adds to its predictor
// scales with the step: the smooth quantity is correction / deltaT at the END of its step.  With T_n = 0 the end of the step about to be
// taken, the corrections of the k steps before sit at tau_j = -(deltaT_n + ... + deltaT_{n-j+1}), j = 1..k, and the start value is
// deltaT_n * sum_j L_j(0) d_j / deltaT_{n-j} with the Lagrange polynomials of those nodes -- the binomial weights 3, -3, 1 when all steps are equal.
// One thread, once per step, behind deltaTKernel: rotates the ring of step lengths and writes w[0..3] for min(have, order) nodes.
__global__ void implStartWeightsKernel(const CaseView c, const ImplView iv) {
    double* h = iv.dtHist;
    for (int j = 4; j >= 1; --j) h[j] = h[j - 1];
    h[0] = c.dt[0];
    const int k = iv.have < iv.order ? iv.have : iv.order;
    double tau[4] = {0, 0, 0, 0}, t = 0.0;
    for (int j = 0; j < k; ++j) { t -= h[j]; tau[j] = t; }
    for (int j = 0; j < 4; ++j) {
        double L = 0.0;
        if (j < k && h[j + 1] > 0.0) {
            L = 1.0;
            for (int q = 0; q < k; ++q) if (q != j) L *= (0.0 - tau[q]) / (tau[j] - tau[q]);
            L *= h[0] / h[j + 1];
        }
        iv.w[j] = L;
    }
}


// fvc::grad(U), Gauss linear: cell gather in ascending face order.  The cell's own velocity once, per face the neighbour cell's
// (cfNbr) or the patch value, the weight and Sf: a third of the bytes of walking owner and neighbour records face by face
// (1.44 -> 0.5 ms at 8 M cells), same operations in the same order.
__global__ __launch_bounds__(QGD_BLOCK) void implCellGradKernel(const MeshView m, const CaseView c, const ImplView iv) {
    const int ci = xcdTile((int)gridDim.x, m.xcdRun) * QGD_BLOCK + threadIdx.x;   // runs of consecutive blocks per XCD: neighbours meet in one L2
    const bool live = ci < m.nC && !(m.ghost && m.ghost[ci] == 1);   // a ghost cell lacks faces here: its gradient arrives by message
    if (__ballot(live) == 0) return;
    if (live) cellGradGauss<6, 1, 6, 1>(m, ci, reinterpret_cast<const double*>(c.A), reinterpret_cast<const double*>(c.bA), iv.gUc);
}

// per face: muf, alphauf, Uf, tauMC -> phiTauMC, Sf.(tauMC & Uf), the laplacian coefficients [updateFluxes.H L107-111]
// (tileList != nullptr: 128 threads per workgroup, the 128-face tiles the staged kernel below leaves to this one; else the faces from fBegin on)
__global__ __launch_bounds__(QGD_BLOCK) void implFaceKernel(const MeshView m, const CaseView c, const ImplView iv, const GasModel gm,
                                                           const PatchBCDev* __restrict__ bcs, const int32_t* __restrict__ tileList, const int fBegin) {
    const int f = tileList ? tileList[blockIdx.x] * 128 + (int)threadIdx.x
                           : fBegin + xcdTile((int)gridDim.x, m.xcdRun) * QGD_BLOCK + (int)threadIdx.x;   // runs of consecutive blocks per XCD: neighbours meet in one L2
    if (f >= (tileList ? m.nIF : m.nF)) return;
    const size_t nF = (size_t)m.nF;
    // what the cell kernels and the matrix products gather (phiTauMC, the laplacian coefficients, phiSigmaDotU) sits at the face's
    // slot-major POSITION like the net fluxes (MeshView::fpos, cfPos): consecutive cells find it at consecutive addresses, by label
    // they would touch every third double of the lines they fetch
    // (labels, kind and weight go out together and the owner's records right behind them: a branch on the loaded kind in front of
    // them would cost every face one more memory round trip)
    const size_t pos = f < m.nIF ? (size_t)m.fpos[f] : (size_t)f;
    const int kind = m.fkind[f];
    const int o = m.own[f];
    const int nLab = f < m.nIF ? m.nei[f] : o;
    const double wLin = f < m.nIF ? m.w[f] : 1.0;
    const RecA Ao = c.A[o];
    const RecB Bo = c.B[o];
    if (kind == 3) {
        for (int k = 0; k < 3; ++k) { iv.phiTau[(size_t)k * nF + pos] = 0.0; iv.UfS[(size_t)k * nF + f] = 0.0; }
        iv.sTau[f] = iv.mufS[f] = iv.aU[pos] = iv.aE[pos] = 0.0;
        return;
    }
    double muf, alf, Uf[3], tau[9];
    if (f < m.nIF) {
        const int n = nLab;
        const RecA An = c.A[n];
        const RecB Bn = c.B[n];
        const double w = wLin;
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
        iv.phiTau[(size_t)j * nF + pos] = S[0] * tau[j] + S[1] * tau[3 + j] + S[2] * tau[6 + j];                  // Sf & tauMC
        iv.UfS[(size_t)j * nF + f] = Uf[j];
    }
    iv.sTau[f] = S[0] * tU[0] + S[1] * tU[1] + S[2] * tU[2];
    iv.mufS[f] = muf;
    const double gsd = m.magSf[f] * m.dn[f];   // |Sf| * (nonOrthDeltaCoeffs inside, deltaCoeffs on patches)
    iv.aU[pos] = muf * gsd;
    iv.aE[pos] = alf * gsd;
}

// The same for the internal faces of a face tile (qgd_setup.hpp FaceTiles, 3-D meshes), its distinct cell records -- the velocity, muQGD and
// the nine gradients that the generic walk gathers as eleven scattered pieces per cell and face -- staged through LDS once: the internal-face
// branch of implFaceKernel, expression by expression.
__global__ __launch_bounds__(128) void implFaceTileKernel(const MeshView m, const CaseView c, const ImplView iv, const GasModel gm) {
    constexpr int FB = 128;
    static_assert(2 * (FB + FB / 16) <= 3 * FB && (FB + FB / 16) <= 2 * FB && 9 * (FB + FB / 16) <= 10 * FB, "faceTileCapCells");
    extern __shared__ v2dTile implLds[];
    const int tile = xcdTile((int)gridDim.x, m.xcdRun * (QGD_BLOCK / FB));
    const int tid = (int)threadIdx.x;
    const int f = tile * FB + tid;
    const bool active = f < m.nIF;
    const int cOff = m.tileOff[2 * tile];
    const int nUc = m.tileOff[2 * tile + 2] - cOff;
    if (nUc == 0) return;   // beyond the caps: in m.tileSpill, done by the generic kernel
    v2dTile* const sA = implLds;                                  // 2 nUc pieces: {rho, ux} {uy, uz} of the cell record
    v2dTile* const sB = sA + 2 * nUc;                             // nUc: {muQGD, alphaQGD / c}
    double* const sG = reinterpret_cast<double*>(sB + nUc);       // 9 nUc: fvc::grad(U)
    const int fl = active ? f : m.nIF - 1;
    const unsigned lc = ldStream(m.locC + fl);
    const size_t nF = (size_t)m.nF, pos = (size_t)ldStream(m.fpos + fl);
    const int kind = m.fkind[fl];
    const double w = ldStream(m.w + fl);
    const double S[3] = {ldStream(m.Sx + fl), ldStream(m.Sy + fl), ldStream(m.Sz + fl)};
    const double gsd = ldStream(m.magSf + fl) * ldStream(m.dn + fl);   // |Sf| * nonOrthDeltaCoeffs
    TileStager<v2dTile, 3, 2, FB, 3, 0> gA;
    TileStager<v2dTile, 2, 1, FB, 2, 1> gB;
    TileStager<double, 10, 9, FB> gG;
    gA.load(reinterpret_cast<const v2dTile*>(c.A), m.tileCells + cOff, nUc, tid);
    gB.load(reinterpret_cast<const v2dTile*>(c.B), m.tileCells + cOff, nUc, tid);
    gG.load(iv.gUc, m.tileCells + cOff, nUc, tid);
    __builtin_amdgcn_sched_barrier(0);
    gA.store(sA, nUc, tid); gB.store(sB, nUc, tid); gG.store(sG, nUc, tid);
    __syncthreads();
    if (!active) return;
    if (kind == 3) {
        for (int k = 0; k < 3; ++k) { iv.phiTau[(size_t)k * nF + pos] = 0.0; iv.UfS[(size_t)k * nF + f] = 0.0; }
        iv.sTau[f] = iv.mufS[f] = iv.aU[pos] = iv.aE[pos] = 0.0;
        return;
    }
    const int lo = (int)(lc & 0xffffu), ln = (int)(lc >> 16);
    const v2dTile a0 = sA[2 * lo], a1 = sA[2 * lo + 1], n0 = sA[2 * ln], n1 = sA[2 * ln + 1];
    const double muQo = sB[lo].x, muQn = sB[ln].x;
    const double uo[3] = {a0.y, a1.x, a1.y}, un[3] = {n0.y, n1.x, n1.y};
    ImplFaceOut r;
    implInternalFace(gm, w, muQo, muQn, uo, un, sG + 9 * lo, sG + 9 * ln, S, gsd, r);   // (shared with the block-fused assembly, qgd_implicit_dev.hpp)
    for (int j = 0; j < 3; ++j) {
        iv.phiTau[(size_t)j * nF + pos] = r.phiTau[j];
        iv.UfS[(size_t)j * nF + f] = r.Uf[j];
    }
    iv.sTau[f] = r.sTau;
    iv.mufS[f] = r.muf;
    iv.aU[pos] = r.aU;
    iv.aE[pos] = r.aE;
}

// QGDRhoEqn.H, the first solve of QGDUEqn.H (rhoU), U = rhoU/rho, and the matrix + source of UEqn per component
__global__ __launch_bounds__(QGD_BLOCK) void implCellUKernel(const MeshView m, const CaseView c, const ImplView iv, const PatchBCDev* __restrict__ bcs) {
    const int ci = xcdTile((int)gridDim.x, m.xcdRun) * QGD_BLOCK + threadIdx.x;   // runs of consecutive blocks per XCD: neighbours meet in one L2
    if (ci >= m.nC) return;
    if (m.ghost && m.ghost[ci] == 1) return;   // ghost rows belong to another shard
    const int n = m.cfCount[ci];
    const size_t base = (size_t)m.cfSlice[ci >> 6] * 64 + (ci & 63);
    const size_t nF = (size_t)m.nF;
    double sum[4] = {0, 0, 0, 0}, dTau[3] = {0, 0, 0}, diagBase = 0;
    if (__ballot(n != 6) == 0) {
        // a wavefront of hexahedra: labels and flux positions of the six faces, then their 48 values in flight before the ordered sums
        int it[6], ps[6];
        double fx[6][4], tx[6][3], ax[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) { it[i] = m.cfItem[base + (size_t)i * 64]; ps[i] = m.cfPos[base + (size_t)i * 64]; }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const size_t pos = (size_t)(ps[i] >= 0 ? ps[i] : ~ps[i]);
#pragma unroll
            for (int k = 0; k < 4; ++k) fx[i][k] = c.flux[(size_t)k * nF + pos];
#pragma unroll
            for (int k = 0; k < 3; ++k) tx[i][k] = iv.phiTau[(size_t)k * nF + pos];
            ax[i] = iv.aU[pos];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int f = it[i] >= 0 ? it[i] : ~it[i];
#pragma unroll
            for (int k = 0; k < 4; ++k) sum[k] = it[i] >= 0 ? sum[k] + fx[i][k] : sum[k] - fx[i][k];
#pragma unroll
            for (int k = 0; k < 3; ++k) dTau[k] = it[i] >= 0 ? dTau[k] + tx[i][k] : dTau[k] - tx[i][k];
            if (f < m.nIF) diagBase += ax[i];
        }
    } else {
        // any cell shapes: eight faces per pass (positions, then their 64 values in flight, then the ordered sums); a position below
        // nIF is an internal face's, patch faces keep their label (MeshView::fpos)
        for (int i0 = 0; i0 < n; i0 += 8) {
            int ps[8];
            double fx[8][4], tx[8][3], ax[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) ps[u] = i0 + u < n ? m.cfPos[base + (size_t)(i0 + u) * 64] : 0;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const bool on = i0 + u < n;
                const size_t pos = (size_t)(ps[u] >= 0 ? ps[u] : ~ps[u]);
#pragma unroll
                for (int k = 0; k < 4; ++k) fx[u][k] = on ? c.flux[(size_t)k * nF + pos] : 0.0;
#pragma unroll
                for (int k = 0; k < 3; ++k) tx[u][k] = on ? iv.phiTau[(size_t)k * nF + pos] : 0.0;
                ax[u] = on ? iv.aU[pos] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (i0 + u >= n) continue;
                const int pos = ps[u] >= 0 ? ps[u] : ~ps[u];
#pragma unroll
                for (int k = 0; k < 4; ++k) sum[k] = ps[u] >= 0 ? sum[k] + fx[u][k] : sum[k] - fx[u][k];
#pragma unroll
                for (int k = 0; k < 3; ++k) dTau[k] = ps[u] >= 0 ? dTau[k] + tx[u][k] : dTau[k] - tx[u][k];
                if (pos < m.nIF) diagBase += ax[u];
            }
        }
    }
    const RecA A = c.A[ci];
    implCellU(m, c, iv, bcs, ci, A, m.V[ci], sum, dTau, diagBase, n, [&](int i) { return m.cfItem[base + (size_t)i * 64]; });   // (qgd_implicit_dev.hpp)
}

// after the U solve: rho and U of the records (p and e stay those of the old time level), patch values of U
__global__ __launch_bounds__(QGD_BLOCK) void implStoreUKernel(const MeshView m, const CaseView c, const ImplView iv) {
    const int ci = blockIdx.x * QGD_BLOCK + threadIdx.x;
    if (ci >= m.nC) return;
    if (m.ghost && m.ghost[ci] == 1) return;
    const size_t nC = (size_t)m.nC;
    RecA a = c.A[ci];
    a.rho = iv.rhoNew[ci];
    a.ux = iv.xU[ci]; a.uy = iv.xU[nC + ci]; a.uz = iv.xU[2 * nC + ci];
    c.A[ci] = a;
    keepCorrection(iv, ci, a.ux); keepCorrection(iv, nC + ci, a.uy); keepCorrection(iv, 2 * nC + ci, a.uz);
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
        double n[3];
        symmNormal(m, bc, f, n);
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
    const int f = xcdTile((int)gridDim.x, m.xcdRun) * QGD_BLOCK + threadIdx.x;   // runs of consecutive blocks per XCD: neighbours meet in one L2
    if (f >= m.nF) return;
    const size_t nF = (size_t)m.nF, pos = f < m.nIF ? (size_t)m.fpos[f] : (size_t)f;   // phiSigmaDotU at the face's position (see implFaceKernel)
    const int kind = m.fkind[f];          // (loaded together with the labels: no branch on it in front of them, see implFaceKernel)
    const int o = m.own[f];
    const int nLab = f < m.nIF ? m.nei[f] : o;
    const double wLin = f < m.nIF ? m.w[f] : 1.0;
    if (kind == 3) { iv.phiSig[pos] = 0.0; return; }
    double g[9];
    if (f < m.nIF) {
        const int n = nLab;
        const double w = wLin;
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
    iv.phiSig[pos] = iv.mufS[f] * (S[0] * gu[0] + S[1] * gu[1] + S[2] * gu[2]) + iv.sTau[f];
}

// EEqn [QGDEEqn.H L37-50] and the matrix + source of the e equation [L55-61]
__global__ __launch_bounds__(QGD_BLOCK) void implCellEKernel(const MeshView m, const CaseView c, const ImplView iv, const GasModel gm,
                                                            const PatchBCDev* __restrict__ bcs) {
    const int ci = xcdTile((int)gridDim.x, m.xcdRun) * QGD_BLOCK + threadIdx.x;   // runs of consecutive blocks per XCD: neighbours meet in one L2
    if (ci >= m.nC) return;
    if (m.ghost && m.ghost[ci] == 1) return;
    const int n = m.cfCount[ci];
    const size_t base = (size_t)m.cfSlice[ci >> 6] * 64 + (ci & 63);
    const size_t nF = (size_t)m.nF;
    double sum = 0, diag = 0, rhs = 0;
    // one face of the cell: its net energy flux, the laplacian coefficient, the patch coefficients of a fixedEnergy face
    auto face = [&](const int it, const int f, const double fl, const double sg, const double a) {
        const double x = fl - sg;   // phiJmH + phiQ - phiPiU - phiSigmaDotU
        sum = it >= 0 ? sum + x : sum - x;
        if (f < m.nIF) diag += a;
        else if (m.fkind[f] != 3) {
            const PatchBCDev bc = bcs[m.bPatch[f - m.nIF]];
            if (bc.ptype != QGD_PATCH_HALO && bc.ptype != QGD_PATCH_CYCLIC && bc.bcT == QGD_BC_FIXEDVALUE) {
                diag += a;                       // fixedEnergy = fixedValue
                rhs += a * (gm.Cv * bc.vT);
            }
        }
    };
    if (__ballot(n != 6) == 0) {
        int it[6], ps[6];
        double fl[6], sg[6], a[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) { it[i] = m.cfItem[base + (size_t)i * 64]; ps[i] = m.cfPos[base + (size_t)i * 64]; }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const size_t pos = (size_t)(ps[i] >= 0 ? ps[i] : ~ps[i]);
            fl[i] = c.flux[4 * nF + pos]; sg[i] = iv.phiSig[pos]; a[i] = iv.aE[pos];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 6; ++i) face(it[i], it[i] >= 0 ? it[i] : ~it[i], fl[i], sg[i], a[i]);
    } else {
        for (int i0 = 0; i0 < n; i0 += 8) {   // any cell shapes: eight faces per pass, as in implCellUKernel
            int ps[8];
            double fl[8], sg[8], a[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) ps[u] = i0 + u < n ? m.cfPos[base + (size_t)(i0 + u) * 64] : 0;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const bool on = i0 + u < n;
                const size_t pos = (size_t)(ps[u] >= 0 ? ps[u] : ~ps[u]);
                fl[u] = on ? c.flux[4 * nF + pos] : 0.0; sg[u] = on ? iv.phiSig[pos] : 0.0; a[u] = on ? iv.aE[pos] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (i0 + u >= n) continue;
                const int pos = ps[u] >= 0 ? ps[u] : ~ps[u];   // = the label for a patch face, < nIF for an internal one: all face() asks
                face(ps[u], pos, fl[u], sg[u], a[u]);
            }
        }
    }
    const RecA A = c.A[ci];   // rho, U of the new time level
    const double V = m.V[ci], dt = c.dt[0], rDeltaT = 1.0 / dt;
    const double rE = c.rE[ci] - (dt / V) * sum;
    const double ecur = rE / A.rho - 0.5 * (A.ux * A.ux + A.uy * A.uy + A.uz * A.uz);   // [L49]
    iv.xE[ci] = startValue(iv, 3 * (size_t)m.nC + ci, ecur);
    iv.diagE[ci] = rDeltaT * A.rho * V + diag;
    iv.rhsE[ci] = rDeltaT * A.rho * ecur * V + rhs;   // fvm::ddt(rho,e) - fvc::ddt(rho,e) [L57]
}

// rhoE = rho*(e + |U|^2/2) [L63], thermo.correct(), p = rho/psi [QGDFoam.C L149-154]
__global__ __launch_bounds__(QGD_BLOCK) void implFinishKernel(const MeshView m, const CaseView c, const ImplView iv, const GasModel gm) {
    const int ci = blockIdx.x * QGD_BLOCK + threadIdx.x;
    double rmin = 1e300, emin = 1e300;
    if (ci < m.nC && !(m.ghost && m.ghost[ci] == 1)) {
        RecA A = c.A[ci];
        const double pOld = A.p;
        A.e = iv.xE[ci];
        keepCorrection(iv, 3 * (size_t)m.nC + ci, A.e);
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

// ---------------------------------------------------------------------------------------------------------------------
// The linear solves of the branch:  y_c = diag_c x_c - gamma_k sum_{internal faces of c} a_f x_nb
// (fvm::ddt(rho, .) - fvm::laplacian(gamma_f, .) of QGDUEqn.H L56-68 / QGDEEqn.H L55-61; QHDUEqn.H L48-64 / QHDTEqn.H L71-79 with
// a_f = |Sf| delta_f and gamma_k = nu, nu, nu, Hi) for up to FOUR right-hand sides AT ONCE -- the components share the face
// coefficients, so one walk over the matrix serves all of them --, with OpenFOAM's normalised residual per component,
// reproducible two-level sums, and every scalar of the loop in a control block ON THE DEVICE (slot-major:
// ctl[slot * 4 + component]), so that no host synchronisation sits inside a solve and a sharded caller reduces / exchanges
// between the phases exactly like for the QHD pressure equation.  Rows are the owned cells [ob, oe) of a shard; ghost cells
// appear only as columns.
//
// Default algorithm (round 4): CHEBYSHEV iteration on the Jacobi-preconditioned system.  These matrices are strictly
// diagonally dominant -- the ddt term carries the diagonal, the laplacian adds rho_c = gamma sum_f a_f / diag_c (0.09 at the
// bench's 200^3 case) -- so Gershgorin bounds the spectrum of D^-1 A by [1 - delta, 1 + delta], delta = max_c rho_c, and the
// Chebyshev polynomial for that interval converges at the conjugate-gradient bound WITHOUT dot products:
//     z_i = D^-1 (b - A x_i);  d_i = c1_i d_(i-1) + c2_i z_i;  x_(i+1) = x_i + d_i          [Saad, Iterative Methods, Alg. 12.1]
//     sigma = 1/delta, rho_0 = delta, c1_0 = 0, c2_0 = 1;  rho_i = 1/(2 sigma - rho_(i-1)), c1_i = rho_i rho_(i-1), c2_i = 2 rho_i/delta
// ONE kernel per iteration does the product, the update of d and x and the partial sums of |b - A x_i| (the residual OpenFOAM
// prints, here of the iterate the step started from), a second one folds them and, unsharded, runs the control logic in the same
// launch: 2 launches and 216 B per cell instead of 7 launches and 408 B of the conjugate-gradient loop (three components,
// hexahedra).  x ping-pongs between the caller's vector and one of the solver's (neighbours read the old iterate while rows write
// the new one); the iterate after i steps sits in buffer i & 1, and implicitSolveEnd copies odd ones home.
//   phase 0  q = A x, r = b - q, A1 = A 1, rho_c      -> {sum|r|, sum x, rows} per component (slots 0..2: SUM), delta (slot 8: MAX)
//   phase 1  normFactor pieces with the global xbar    -> slot 3
//   phase 2  first residual, done?, the constants; x_1 = x_0 + D^-1 r_0             -> then the ghost entries of the iterate
//   phase 3  one step: the kernel above + the fold      -> slot 5 (sum |b - A x_i|); then the ghost entries of the new iterate
//   phase 4  residual of x_i, iteration count, done?, c1, c2 (unsharded: done inside phase 3's fold launch)
//   phase 5  nothing (kept so that drivers written for the conjugate-gradient phases run both)
// QGD_IMPL_SOLVER=pcg keeps round 3's Jacobi-preconditioned conjugate gradients (same phases: 2 d = r/diag, r.z -> slot 4 + ghost
// entries of d | 3 q = A d, d.q -> slot 5 | 4 alpha, x, r, {sum|r|, r.z} -> slots 6, 7 | 5 residual, done?, beta, d -> ghost entries).
// ---------------------------------------------------------------------------------------------------------------------
enum ICtl : int { I_ABSR = 0, I_SUMX = 1, I_N = 2, I_NORM = 3, I_RZ = 4, I_DQ = 5, I_ABSR2 = 6, I_RZNEW = 7, I_DELTA = 8, I_RES = 9, I_RES0 = 10,
                  I_DONE = 11, I_ITER = 12, I_ALPHA = 13, I_BETA = 14, I_NORMF = 15, I_SLOTS = 16, I_ALLDONE = 64, I_COUNT = 68,
                  // Chebyshev: slot 5 holds sum |b - A x_i|, slots 13 / 14 the coefficients c1 / c2
                  I_CABSR = 5, I_C1 = 13, I_C2 = 14, I_BEST = 0, I_STALL = 1 };   // (slots 0, 1: reused from phase 2 on, see chebAfterStep)
#define ICTL(slot, k) ((slot) * 4 + (k))

__device__ __forceinline__ double iBlockSum(double v) {
    __shared__ double s[QGD_BLOCK / 64];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < QGD_BLOCK / 64; ++i) t += s[i];
    }
    __syncthreads();
    return t;
}

__device__ __forceinline__ double iBlockMax(double v) {
    __shared__ double s[QGD_BLOCK / 64];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, 64));
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < QGD_BLOCK / 64; ++i) t = fmax(t, s[i]);
    }
    __syncthreads();
    return t;
}

struct ISolveView {
    int NR, ob, n, nC;                 // right-hand sides, first row, rows, vector stride
    double gam[4];                     // gamma_k: the face coefficients of component k are gam[k] * a
    double* xb;                        // Chebyshev: the second buffer of the iterate (NR * nC)
    const double* a;                   // nF, at the faces' slot-major positions (MeshView::fpos / cfPos)
    const double* diag; const double* rhs; double* x;    // NR * nC each, component-major
    double *r, *d, *q;                 // NR * nC each
    float* df;                         // Chebyshev: d_{i-1} of the recurrence d_i = c1 d_{i-1} + c2 z, kept in single precision (see iChebKernel)
    double* part;                      // partial sums: (row * NR + k) * nBlocks + block
    double* ctl;
    int nBlocks;
    int xrun;                          // row blocks per XCD run of the matrix product (xcdRunBlock in qgd_device.hpp), 0: plain order
};

// MODE 0: q = A x, r = b - q, d <- A 1 (kept until phase 1), partial {|r|, x};  MODE 1: q = A d, partial {d.q}
template <int NR, int MODE>
__global__ __launch_bounds__(QGD_BLOCK) void iApplyKernel(const MeshView m, const ISolveView v) {
    if (v.ctl[I_ALLDONE] != 0.0 && MODE == 1) return;
    const int blk = xcdRunBlock(v.xrun);   // runs of row blocks per XCD (qgd_device.hpp); the partial sums stay in block order
    const int i = blk * QGD_BLOCK + threadIdx.x;
    double s0[NR], s1[NR], s2[NR];
#pragma unroll
    for (int k = 0; k < NR; ++k) s0[k] = s1[k] = s2[k] = 0.0;
    if (i < v.n) {
        const int c = v.ob + i;
        const int cnt = m.cfCount[c];
        const size_t base = (size_t)m.cfSlice[c >> 6] * 64 + (c & 63);
        const double* __restrict__ src = MODE == 0 ? v.x : v.d;
        double acc[NR], rowsum = 0.0;
#pragma unroll
        for (int k = 0; k < NR; ++k) acc[k] = 0.0;
        // eight entries at a time, every level of the chain (lists -> face coefficient and neighbour values) requested before the
        // first use; entries past the row end and boundary faces (nb < 0) add 0 * 0 in their place, which changes no bit
        constexpr int U = 8;
        for (int e0 = 0; e0 < cnt; e0 += U) {
            int nb[U], it[U];
            double af[U], xs[NR][U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool in = e0 + u < cnt;
                nb[u] = in ? m.cfNbr[base + (size_t)(e0 + u) * 64] : -1;
                it[u] = in ? m.cfPos[base + (size_t)(e0 + u) * 64] : 0;      // v.a is stored by position
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                af[u] = nb[u] >= 0 ? v.a[it[u] >= 0 ? it[u] : ~it[u]] : 0.0;
#pragma unroll
                for (int k = 0; k < NR; ++k) xs[k][u] = nb[u] >= 0 ? src[(size_t)k * v.nC + nb[u]] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                rowsum += af[u];
#pragma unroll
                for (int k = 0; k < NR; ++k) acc[k] += af[u] * xs[k][u];
            }
        }
#pragma unroll
        for (int k = 0; k < NR; ++k) {
            const size_t j = (size_t)k * v.nC + c;
            if (MODE == 1 && v.ctl[ICTL(I_DONE, k)] != 0.0) continue;
            const double dg = v.diag[j], xv = src[j];
            const double y = dg * xv - v.gam[k] * acc[k];
            v.q[j] = y;
            if (MODE == 0) {
                const double rc = v.rhs[j] - y;
                v.r[j] = rc;
                v.d[j] = dg - v.gam[k] * rowsum;          // (A 1)_c
                s0[k] = fabs(rc); s1[k] = xv;
                s2[k] = (v.gam[k] * rowsum) / dg;          // Gershgorin radius of row c of D^-1 A
            } else s0[k] = xv * y;
        }
    }
#pragma unroll
    for (int k = 0; k < NR; ++k) {
        const double t0 = iBlockSum(s0[k]);
        if (threadIdx.x == 0) v.part[(size_t)(0 * NR + k) * v.nBlocks + blk] = t0;
        if (MODE == 0) {
            const double t1 = iBlockSum(s1[k]);
            if (threadIdx.x == 0) v.part[(size_t)(1 * NR + k) * v.nBlocks + blk] = t1;
            const double t2 = iBlockMax(s2[k]);
            if (threadIdx.x == 0) v.part[(size_t)(2 * NR + k) * v.nBlocks + blk] = t2;
        }
    }
}

// One Chebyshev step (see the head of this section).  FIRST: x_1 = x_0 + D^-1 r_0 from the residual phase 0 left (no product);
// otherwise the product on the current iterate (buffer step & 1), the new d and the next iterate into the other buffer, and the
// partial sums of |b - A x_i|.  Components that are done are not touched; all others have made the same number of steps.
#ifndef QGD_CHEB_DF
#define QGD_CHEB_DF 1   // 0: the recurrence's d in double (A/B: profiles/r05_ab_implicit_direction_f32.txt)
#endif
template <int NR, int FIRST>
__global__ __launch_bounds__(QGD_BLOCK) void iChebKernel(const MeshView m, const ISolveView v) {
    const int blk = FIRST ? (int)blockIdx.x : xcdRunBlock(v.xrun);
    const int i = blk * QGD_BLOCK + threadIdx.x;
    int step = -1;                                        // steps made so far by the components still running (all the same number)
#pragma unroll
    for (int k = 0; k < NR; ++k) if (v.ctl[ICTL(I_DONE, k)] == 0.0) step = (int)v.ctl[ICTL(I_ITER, k)];
    if (step < 0) return;
    const double* __restrict__ src = (step & 1) ? v.xb : v.x;
    double* __restrict__ dst = (step & 1) ? v.x : v.xb;
    double s0[NR];
#pragma unroll
    for (int k = 0; k < NR; ++k) s0[k] = 0.0;
    if (i < v.n) {
        const int c = v.ob + i;
        double acc[NR];
#pragma unroll
        for (int k = 0; k < NR; ++k) acc[k] = 0.0;
        if (!FIRST) {
            const int cnt = m.cfCount[c];
            const size_t base = (size_t)m.cfSlice[c >> 6] * 64 + (c & 63);
            constexpr int U = 8;   // eight entries at a time, every level of the chain requested before the first use (as iApplyKernel)
            for (int e0 = 0; e0 < cnt; e0 += U) {
                int nb[U], it[U];
                double af[U], xs[NR][U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const bool in = e0 + u < cnt;
                    nb[u] = in ? m.cfNbr[base + (size_t)(e0 + u) * 64] : -1;
                    it[u] = in ? m.cfPos[base + (size_t)(e0 + u) * 64] : 0;
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    af[u] = nb[u] >= 0 ? v.a[it[u] >= 0 ? it[u] : ~it[u]] : 0.0;
#pragma unroll
                    for (int k = 0; k < NR; ++k) xs[k][u] = nb[u] >= 0 ? src[(size_t)k * v.nC + nb[u]] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int k = 0; k < NR; ++k) acc[k] += af[u] * xs[k][u];
            }
        }
#pragma unroll
        for (int k = 0; k < NR; ++k) {
            if (v.ctl[ICTL(I_DONE, k)] != 0.0) continue;
            const size_t j = (size_t)k * v.nC + c;
            const double dg = v.diag[j], xv = src[j];
            double rc, dn;
            if (FIRST) { rc = v.r[j]; dn = rc / dg; }
            else {
                rc = v.rhs[j] - (dg * xv - v.gam[k] * acc[k]);
#if QGD_CHEB_DF
                dn = v.ctl[ICTL(I_C1, k)] * (double)v.df[j] + v.ctl[ICTL(I_C2, k)] * (rc / dg);
#else
                dn = v.ctl[ICTL(I_C1, k)] * v.d[j] + v.ctl[ICTL(I_C2, k)] * (rc / dg);
#endif
            }
            // the update x_{i+1} = x_i + d_i is made with d_i in double; what the NEXT step's recurrence reads back is d_i rounded to single
            // precision (4 + 4 instead of 8 + 8 B per component and step).  z is the true residual of the iterate every step, so the rounding
            // (6e-8 of a term that c1 < 1 damps) perturbs the polynomial, not the fixed point: same solution to the same tolerance
#if QGD_CHEB_DF
            v.df[j] = (float)dn;
#else
            v.d[j] = dn;
#endif
            dst[j] = xv + dn;
            s0[k] = fabs(rc);
        }
    }
    if (!FIRST) {
#pragma unroll
        for (int k = 0; k < NR; ++k) {
            const double t0 = iBlockSum(s0[k]);
            if (threadIdx.x == 0) v.part[(size_t)k * v.nBlocks + blk] = t0;
        }
    }
}
// after the last step: iterates that ended in the solver's buffer (odd step counts) go home to the caller's vector
__global__ __launch_bounds__(QGD_BLOCK) void iChebHomeKernel(const ISolveView v) {
    const int i = blockIdx.x * QGD_BLOCK + threadIdx.x;
    if (i >= v.n) return;
    for (int k = 0; k < v.NR; ++k) {
        if (v.ctl[ICTL(I_DONE, k)] == 3.0 || (((int)v.ctl[ICTL(I_ITER, k)]) & 1) == 0) continue;
        const size_t j = (size_t)k * v.nC + v.ob + i;
        v.x[j] = v.xb[j];
    }
}
// phase 1: sum(|A x - xbar A 1| + |b - xbar A 1|)
template <int NR>
__global__ __launch_bounds__(QGD_BLOCK) void iNormKernel(const ISolveView v) {
    const int i = blockIdx.x * QGD_BLOCK + threadIdx.x;
#pragma unroll
    for (int k = 0; k < NR; ++k) {
        double t = 0.0;
        if (i < v.n) {
            const size_t j = (size_t)k * v.nC + v.ob + i;
            const double ref = (v.ctl[ICTL(I_SUMX, k)] / v.ctl[ICTL(I_N, k)]) * v.d[j];
            t = fabs(v.q[j] - ref) + fabs(v.rhs[j] - ref);
        }
        t = iBlockSum(t);
        if (threadIdx.x == 0) v.part[(size_t)k * v.nBlocks + blockIdx.x] = t;
    }
}
// phase 2 (FIRST = 1): d = r/diag, partial r.z;  phase 5 (FIRST = 0): d = r/diag + beta d
template <int NR, int FIRST>
__global__ __launch_bounds__(QGD_BLOCK) void iDirectionKernel(const ISolveView v) {
    if (v.ctl[I_ALLDONE] != 0.0) return;
    const int i = blockIdx.x * QGD_BLOCK + threadIdx.x;
#pragma unroll
    for (int k = 0; k < NR; ++k) {
        double t = 0.0;
        const bool on = v.ctl[ICTL(I_DONE, k)] == 0.0;
        if (i < v.n && on) {
            const size_t j = (size_t)k * v.nC + v.ob + i;
            const double rc = v.r[j], z = rc / v.diag[j];
            v.d[j] = FIRST ? z : z + v.ctl[ICTL(I_BETA, k)] * v.d[j];
            t = rc * z;
        }
        if (FIRST) {
            t = iBlockSum(t);
            if (threadIdx.x == 0) v.part[(size_t)k * v.nBlocks + blockIdx.x] = t;
        }
    }
}
// phase 4: x += alpha d, r -= alpha q; partial {|r|, r.z}
template <int NR>
__global__ __launch_bounds__(QGD_BLOCK) void iUpdateKernel(const ISolveView v) {
    if (v.ctl[I_ALLDONE] != 0.0) return;
    const int i = blockIdx.x * QGD_BLOCK + threadIdx.x;
#pragma unroll
    for (int k = 0; k < NR; ++k) {
        double t0 = 0.0, t1 = 0.0;
        const bool on = v.ctl[ICTL(I_DONE, k)] == 0.0;
        if (i < v.n && on) {
            const size_t j = (size_t)k * v.nC + v.ob + i;
            const double alpha = v.ctl[ICTL(I_ALPHA, k)];
            v.x[j] += alpha * v.d[j];
            const double rc = v.r[j] - alpha * v.q[j];
            v.r[j] = rc;
            t0 = fabs(rc); t1 = rc * (rc / v.diag[j]);
        }
        t0 = iBlockSum(t0); t1 = iBlockSum(t1);
        if (threadIdx.x == 0) { v.part[(size_t)(0 * NR + k) * v.nBlocks + blockIdx.x] = t0; v.part[(size_t)(1 * NR + k) * v.nBlocks + blockIdx.x] = t1; }
    }
}
// folds rows x NR rows of partials into ctl[(firstSlot + row) * 4 + k], one workgroup per row; components that are done keep
// their values.  maxRow >= 0: that row is folded with max into ctl[(maxSlot) * 4 + k] (the Gershgorin radius of phase 0).
__device__ __forceinline__ double iFoldRow(const double* __restrict__ p, const int n, const bool useMax) {
    // eight independent chains per thread, their loads requested together (31 250 partials at 8 M cells: 15 round trips instead of 30)
    double w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int i = threadIdx.x;
    for (; i + 7 * QGD_BLOCK < n; i += 8 * QGD_BLOCK) {
        double x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = p[i + j * QGD_BLOCK];
#pragma unroll
        for (int j = 0; j < 8; ++j) w[j] = useMax ? fmax(w[j], x[j]) : w[j] + x[j];
    }
    for (; i < n; i += QGD_BLOCK) w[0] = useMax ? fmax(w[0], p[i]) : w[0] + p[i];
    if (useMax) return iBlockMax(fmax(fmax(fmax(w[0], w[1]), fmax(w[2], w[3])), fmax(fmax(w[4], w[5]), fmax(w[6], w[7]))));
    return iBlockSum(((w[0] + w[1]) + (w[2] + w[3])) + ((w[4] + w[5]) + (w[6] + w[7])));
}
__global__ __launch_bounds__(QGD_BLOCK) void iFoldKernel(const ISolveView v, const int NR, const int rows, const int firstSlot, const int always,
                                                        const int maxRow, const int maxSlot) {
    if (!always && v.ctl[I_ALLDONE] != 0.0) return;
    const int row = blockIdx.x / NR, k = blockIdx.x % NR;
    if (!always && v.ctl[ICTL(I_DONE, k)] != 0.0) return;   // uniform over the workgroup
    const double t = iFoldRow(v.part + (size_t)(row * NR + k) * v.nBlocks, v.nBlocks, row == maxRow);
    if (threadIdx.x == 0) v.ctl[ICTL(row == maxRow ? maxSlot : firstSlot + row, k)] = t;
}
// the Chebyshev constants of the step after `rhoPrev` (see the head of this section)
__device__ __forceinline__ void chebNext(double* __restrict__ ctl, const int k, const double rhoPrev) {
    const double delta = ctl[ICTL(I_DELTA, k)];
    const double rho = 1.0 / (2.0 / delta - rhoPrev);
    ctl[ICTL(I_C1, k)] = rho * rhoPrev;
    ctl[ICTL(I_C2, k)] = 2.0 * rho / delta;
}
// what follows a Chebyshev step for component k, whose sum |b - A x_i| is in slot I_CABSR: the residual of the iterate the step
// started from, the step count, done?, the next constants
__device__ __forceinline__ void chebAfterStep(double* __restrict__ ctl, const int k, const double tol, const int maxIter) {
    const double res = ctl[ICTL(I_CABSR, k)] / ctl[ICTL(I_NORMF, k)], it = ctl[ICTL(I_ITER, k)] + 1.0;
    ctl[ICTL(I_RES, k)] = res; ctl[ICTL(I_ITER, k)] = it;
    // the residual here is the TRUE one, b - A x_i, not a recurrence: it stalls at the rounding floor of the product (the conjugate-
    // gradient loop's recurrence residual keeps falling below it and never notices; on a nearly uniform field OpenFOAM's normFactor is
    // small against |b| and the floor of the NORMALISED residual can sit at 1e-12).  A tolerance under that floor would burn maxIter
    // steps for nothing.  The 1-norm of a Chebyshev residual is not monotone, so one slow pair of steps proves nothing: the best
    // residual so far (slot 0) has to go unimproved by a factor 2 for several times as many steps as the Chebyshev bound needs to
    // halve it (slot 1 counts them) -- then the component stops (done = 4).  Only looked at below 1e-8; the Gershgorin interval cannot
    // be wrong, so a stall there is rounding, not divergence.  (Slots 0 and 1 are free from phase 2 on: the first residual and the
    // mean of x were consumed by phases 1 and 2, and no later reduction touches them.)
    const double sigma = 1.0 / ctl[ICTL(I_DELTA, k)], rate = sigma - sqrt(fmax(sigma * sigma - 1.0, 0.0));
    const double halving = rate < 1.0 ? log(0.5) / log(fmax(rate, 1e-300)) : 1e9;   // steps the bound needs to halve the residual
    double best = ctl[ICTL(I_BEST, k)], count = ctl[ICTL(I_STALL, k)];
    if (res < 0.5 * best) { best = res; count = 0.0; } else count += 1.0;
    ctl[ICTL(I_BEST, k)] = best; ctl[ICTL(I_STALL, k)] = count;
    const bool stalled = res < 1e-8 && count >= 8.0 + 4.0 * halving;
    if (res < tol || it >= (double)maxIter) ctl[ICTL(I_DONE, k)] = 1.0;
    else if (stalled) ctl[ICTL(I_DONE, k)] = 4.0;   // at the rounding floor of b - A x: as converged as this arithmetic gets; not counted as a failed solve
    else chebNext(ctl, k, ctl[ICTL(I_C2, k)] * ctl[ICTL(I_DELTA, k)] * 0.5);   // rho_i = c2_i delta / 2
}
// one rank: the fold of a Chebyshev step and its control logic in ONE launch, one workgroup per component (the components do not
// depend on each other; nothing reads I_ALLDONE on this path)
__global__ __launch_bounds__(QGD_BLOCK) void iFoldChebKernel(const ISolveView v, const double tol, const int maxIter) {
    const int k = blockIdx.x;
    if (v.ctl[ICTL(I_DONE, k)] != 0.0) return;   // uniform over the workgroup
    const double t = iFoldRow(v.part + (size_t)k * v.nBlocks, v.nBlocks, false);
    if (threadIdx.x == 0) {
        v.ctl[ICTL(I_CABSR, k)] = t;
        chebAfterStep(v.ctl, k, tol, maxIter);
    }
}
// bookkeeping, one thread per component.  stage 0: start (valid mask: components along empty directions are "done" from the start);
// 1: first residual; 2: alpha or breakdown; 3: residual, iteration count, done?, beta (conjugate gradients);  Chebyshev: 1 also sets
// the constants of step 1, 4: after the first step (x_1 stands), 5: after any later step.  Last, thread-independent: all done?
// ctl[I_ALLDONE + 1] = Chebyshev steps made so far by the components still running (which buffer holds the iterate).
__global__ void iCtlKernel(double* __restrict__ ctl, const int NR, const int stage, const double nRows, const int validMask, const double tol,
                           const int maxIter, const int cheb) {
    const int k = threadIdx.x;
    if (k < NR) {
        if (stage == 0) {
            for (int s = 0; s < I_SLOTS; ++s) ctl[ICTL(s, k)] = 0.0;
            ctl[ICTL(I_N, k)] = nRows;
            ctl[ICTL(I_DONE, k)] = (validMask >> k) & 1 ? 0.0 : 3.0;   // 3: not solved
        } else if (ctl[ICTL(I_DONE, k)] == 0.0) {
            if (stage == 1) {
                const double nf = ctl[ICTL(I_NORM, k)] + 1e-20, res = ctl[ICTL(I_ABSR, k)] / nf;
                ctl[ICTL(I_NORMF, k)] = nf; ctl[ICTL(I_RES, k)] = res; ctl[ICTL(I_RES0, k)] = res;
                if (res < tol || maxIter <= 0) ctl[ICTL(I_DONE, k)] = 1.0;
                else if (cheb) {
                    ctl[ICTL(I_BEST, k)] = res; ctl[ICTL(I_STALL, k)] = 0.0;   // stall detection of chebAfterStep
                    // Gershgorin: the spectrum of D^-1 A lies in [1 - delta, 1 + delta]; strictly below 1 by diagonal dominance.  The
                    // clamps keep the recurrence finite for a purely diagonal matrix (delta = 0: x_1 is exact) and for a singular one
                    const double delta = fmin(fmax(ctl[ICTL(I_DELTA, k)], 1e-30), 1.0 - 1e-9);
                    ctl[ICTL(I_DELTA, k)] = delta;
                    chebNext(ctl, k, delta);   // rho_0 = delta
                }
            } else if (stage == 2) {
                const double dq = ctl[ICTL(I_DQ, k)], rz = ctl[ICTL(I_RZ, k)];
                if (!(dq > 0) || !(rz > 0)) ctl[ICTL(I_DONE, k)] = 2.0;
                else ctl[ICTL(I_ALPHA, k)] = rz / dq;
            } else if (stage == 3) {
                const double res = ctl[ICTL(I_ABSR2, k)] / ctl[ICTL(I_NORMF, k)], it = ctl[ICTL(I_ITER, k)] + 1.0;
                ctl[ICTL(I_RES, k)] = res; ctl[ICTL(I_ITER, k)] = it;
                if (res < tol || it >= (double)maxIter) ctl[ICTL(I_DONE, k)] = 1.0;
                else { ctl[ICTL(I_BETA, k)] = ctl[ICTL(I_RZNEW, k)] / ctl[ICTL(I_RZ, k)]; ctl[ICTL(I_RZ, k)] = ctl[ICTL(I_RZNEW, k)]; }
            } else if (stage == 4) {
                ctl[ICTL(I_ITER, k)] = 1.0;                       // x_1 stands (its residual is known after the next step)
                if (maxIter <= 1) ctl[ICTL(I_DONE, k)] = 1.0;
            } else if (stage == 5) chebAfterStep(ctl, k, tol, maxIter);
        }
    }
    __syncthreads();
    if (k == 0) {
        bool all = true;
        for (int j = 0; j < NR; ++j) all = all && ctl[ICTL(I_DONE, j)] != 0.0;
        ctl[I_ALLDONE] = all ? 1.0 : 0.0;
        if (stage == 0) ctl[I_ALLDONE + 1] = 0.0;
    }
}

// halo messages of the branch on a shard.  kind 1: fvc::grad(U) (9 per cell); kind 2: the velocity after its solve (3 per cell);
// kind 3: the search direction of the solve in flight (its right-hand sides per cell, component-major on the device); kind 4: its
// initial guess (same shape): the first matrix product needs the neighbours' start values in the ghost columns
__global__ __launch_bounds__(QGD_BLOCK) void implHaloKernel(const CaseView c, const ImplView iv, double* __restrict__ dirn, const int nC, const int kind,
                                                           const int NR, const int32_t* __restrict__ cells, const int nCells,
                                                           double* __restrict__ buf, const int pack) {
    const int i = blockIdx.x * QGD_BLOCK + threadIdx.x;
    if (i >= nCells) return;
    const size_t ci = (size_t)cells[i];
    if (kind == 1) {
        double* b = buf + (size_t)i * 9;
        for (int k = 0; k < 9; ++k) { if (pack) b[k] = iv.gUc[ci * 9 + k]; else iv.gUc[ci * 9 + k] = b[k]; }
    } else if (kind == 2) {
        double* b = buf + (size_t)i * 3;
        if (pack) { const RecA a = c.A[ci]; b[0] = a.ux; b[1] = a.uy; b[2] = a.uz; }
        else { RecA a = c.A[ci]; a.ux = b[0]; a.uy = b[1]; a.uz = b[2]; c.A[ci] = a; }
    } else {
        double* b = buf + (size_t)i * NR;
        for (int k = 0; k < NR; ++k) { if (pack) b[k] = dirn[(size_t)k * nC + ci]; else dirn[(size_t)k * nC + ci] = b[k]; }
    }
}

// statistics of a step without a host round trip: stats[0] = steps in which a solve stopped above its tolerance (iteration limit or
// breakdown), stats[1] = the flag of the step in flight; stats[2] / stats[3] the same for solves that the Chebyshev iteration ended at
// the rounding floor of its true residual (done = 4) while that floor lay ABOVE the tolerance: not a failure of the iteration, but not
// "solved to implicitTol" either -- OpenFOAM would have gone on to maxIter and printed the residual; callers report the count.
// mode 0: start of a step; 1: after a solve; 2: end of a step
#define ISTAT_COUNT 4
__global__ void iStatKernel(const double* __restrict__ ctl, double* __restrict__ stats, const int NR, const int mode, const double tol) {
    if (mode == 0) stats[1] = stats[3] = 0.0;
    else if (mode == 1) {
        for (int k = 0; k < NR; ++k) {
            const double dn = ctl[ICTL(I_DONE, k)];
            if (dn == 2.0 || (dn == 1.0 && !(ctl[ICTL(I_RES, k)] < tol))) stats[1] = 1.0;
            if (dn == 4.0 && !(ctl[ICTL(I_RES, k)] < tol)) stats[3] = 1.0;
        }
    } else { stats[0] += stats[1]; stats[2] += stats[3]; }
}

}  // namespace

void launchImplicitStartWeights(hipStream_t s, const CaseView& c, const ImplView& iv) {
    if (iv.w == nullptr || iv.pred == nullptr) return;
    implStartWeightsKernel<<<1, 1, 0, s>>>(c, iv);
}

// ---- host side of the solves ---------------------------------------------------------------------------------------------
struct ImplicitSolver {
    MeshView m{};
    hipStream_t stream = nullptr;
    int ob = 0, oe = 0;
    double *r = nullptr, *d = nullptr, *q = nullptr, *xb = nullptr, *part = nullptr, *ctl = nullptr, *hostCtl = nullptr;
    float* df = nullptr;
    bool cheb = true;           // QGD_IMPL_SOLVER: "cheb" (default) | "pcg"
    int hostSteps = 0;          // Chebyshev steps queued so far in the solve in flight (which buffer a halo message moves)
    double* stats = nullptr;    // device: [0] unconverged steps, [1] flag of the step in flight, [2], [3] the same for stalled solves (iStatKernel), then the control blocks of the last U and e solves
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    ISolveView v{};
    int NR = 1, validMask = 1, maxIter = 0;
    double tol = 0;
    int rowRun = 64;            // QGD_ROW_XCD_RUN (0 ... 4096): row blocks per XCD run of the matrix product, 0: plain order (measured: 0.286 -> 0.274 ms per product at 8 M cells)
};
#define ICHECK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) { (void)hipGetLastError(); throw std::runtime_error(std::string("HIP: ") + hipGetErrorString(e_)); } \
    } while (0)

ImplicitSolver* implicitSolverCreate(hipStream_t stream, const MeshView& m, int ownedBegin, int ownedEnd) {
    ImplicitSolver* S = new ImplicitSolver();
    S->m = m; S->stream = stream; S->ob = ownedBegin; S->oe = ownedEnd < 0 ? m.nC : ownedEnd;
    if (const char* e = std::getenv("QGD_ROW_XCD_RUN")) {
        char* end = nullptr;
        const long v = std::strtol(e, &end, 10);
        if (!end || *end != '\0' || v < 0 || v > 4096) { delete S; throw std::invalid_argument(std::string("QGD_ROW_XCD_RUN=") + e + " is outside [0, 4096]"); }
        S->rowRun = (int)v;
    }
    if (const char* e = std::getenv("QGD_IMPL_SOLVER")) {
        const std::string w(e);
        if (w != "cheb" && w != "pcg") { delete S; throw std::invalid_argument("QGD_IMPL_SOLVER=" + w + " is not a supported value (cheb, pcg)"); }
        S->cheb = w == "cheb";
    }
    const size_t nC = (size_t)m.nC, nb = (size_t)gridOf(S->oe - S->ob);
    try {
        ICHECK(hipMalloc((void**)&S->r, sizeof(double) * 4 * nC)); ICHECK(hipMalloc((void**)&S->d, sizeof(double) * 4 * nC)); ICHECK(hipMalloc((void**)&S->df, sizeof(float) * 4 * nC)); ICHECK(hipMemset(S->df, 0, sizeof(float) * 4 * nC));
        ICHECK(hipMalloc((void**)&S->q, sizeof(double) * 4 * nC)); ICHECK(hipMalloc((void**)&S->part, sizeof(double) * 12 * std::max<size_t>(nb, 1)));
        ICHECK(hipMalloc((void**)&S->xb, sizeof(double) * 4 * nC));
        ICHECK(hipMemset(S->xb, 0, sizeof(double) * 4 * nC));
        ICHECK(hipMalloc((void**)&S->ctl, sizeof(double) * I_COUNT));
        ICHECK(hipMalloc((void**)&S->stats, sizeof(double) * (ISTAT_COUNT + 2 * I_COUNT)));
        ICHECK(hipMemset(S->stats, 0, sizeof(double) * (ISTAT_COUNT + 2 * I_COUNT)));
        ICHECK(hipMemset(S->d, 0, sizeof(double) * 4 * nC));     // ghost entries of the direction are read before the first exchange fills them
        ICHECK(hipMemset(S->ctl, 0, sizeof(double) * I_COUNT));
        ICHECK(hipStreamSynchronize(nullptr));   // null-stream zero-fills are done before anything runs on the solver's non-blocking stream
        ICHECK(hipHostMalloc((void**)&S->hostCtl, sizeof(double) * I_COUNT * 5, hipHostMallocDefault));
        for (hipEvent_t& e : S->ev) ICHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    } catch (...) { implicitSolverFree(S); throw; }
    return S;
}
void implicitSolverFree(ImplicitSolver* S) {
    if (!S) return;
    (void)hipFree(S->r); (void)hipFree(S->d); (void)hipFree(S->df); (void)hipFree(S->q); (void)hipFree(S->xb); (void)hipFree(S->part); (void)hipFree(S->ctl); (void)hipFree(S->stats);
    if (S->hostCtl) (void)hipHostFree(S->hostCtl);
    for (hipEvent_t e : S->ev) if (e) (void)hipEventDestroy(e);
    delete S;
}
int64_t implicitSolverBytes(const ImplicitSolver* S) { return S ? (int64_t)sizeof(double) * (16 * (int64_t)S->m.nC + 12 * gridOf(S->oe - S->ob) + I_COUNT) : 0; }
double* implicitSolverCtl(ImplicitSolver* S) { return S->ctl; }
double* implicitSolverDirection(ImplicitSolver* S) { return S->d; }
int implicitSolverRhs(const ImplicitSolver* S) { return S->NR; }
bool implicitSolverChebyshev(const ImplicitSolver* S) { return S->cheb; }
double* implicitSolverIterate(ImplicitSolver* S) { return (S->hostSteps & 1) ? S->xb : S->v.x; }

template <int NR>
static void iPhaseT(ImplicitSolver* S, int phase, bool fused = false) {
    const ISolveView& v = S->v;
    hipStream_t s = S->stream;
    const int nb = v.nBlocks, cheb = S->cheb ? 1 : 0;
    switch (phase) {
        case 0:
            S->hostSteps = 0;
            iCtlKernel<<<1, 4, 0, s>>>(v.ctl, NR, 0, (double)v.n, S->validMask, S->tol, S->maxIter, cheb);
            iApplyKernel<NR, 0><<<nb, QGD_BLOCK, 0, s>>>(S->m, v);
            iFoldKernel<<<3 * NR, QGD_BLOCK, 0, s>>>(v, NR, 3, I_ABSR, 1, 2, I_DELTA);   // rows: sum|r|, sum x | max rho_c -> slot 8
            break;
        case 1:
            iNormKernel<NR><<<nb, QGD_BLOCK, 0, s>>>(v);
            iFoldKernel<<<1 * NR, QGD_BLOCK, 0, s>>>(v, NR, 1, I_NORM, 1, -1, 0);
            break;
        case 2:
            iCtlKernel<<<1, 4, 0, s>>>(v.ctl, NR, 1, 0.0, 0, S->tol, S->maxIter, cheb);
            if (S->cheb) {
                iChebKernel<NR, 1><<<nb, QGD_BLOCK, 0, s>>>(S->m, v);
                iCtlKernel<<<1, 4, 0, s>>>(v.ctl, NR, 4, 0.0, 0, S->tol, S->maxIter, cheb);
                S->hostSteps = 1;
            } else {
                iDirectionKernel<NR, 1><<<nb, QGD_BLOCK, 0, s>>>(v);
                iFoldKernel<<<1 * NR, QGD_BLOCK, 0, s>>>(v, NR, 1, I_RZ, 0, -1, 0);
            }
            break;
        case 3:
            if (S->cheb) {
                iChebKernel<NR, 0><<<nb, QGD_BLOCK, 0, s>>>(S->m, v);
                if (fused) iFoldChebKernel<<<NR, QGD_BLOCK, 0, s>>>(v, S->tol, S->maxIter);
                else iFoldKernel<<<1 * NR, QGD_BLOCK, 0, s>>>(v, NR, 1, I_CABSR, 0, -1, 0);
                S->hostSteps++;
            } else {
                iApplyKernel<NR, 1><<<nb, QGD_BLOCK, 0, s>>>(S->m, v);
                iFoldKernel<<<1 * NR, QGD_BLOCK, 0, s>>>(v, NR, 1, I_DQ, 0, -1, 0);
            }
            break;
        case 4:
            if (S->cheb) { if (!fused) iCtlKernel<<<1, 4, 0, s>>>(v.ctl, NR, 5, 0.0, 0, S->tol, S->maxIter, cheb); }
            else {
                iCtlKernel<<<1, 4, 0, s>>>(v.ctl, NR, 2, 0.0, 0, S->tol, S->maxIter, cheb);
                iUpdateKernel<NR><<<nb, QGD_BLOCK, 0, s>>>(v);
                iFoldKernel<<<2 * NR, QGD_BLOCK, 0, s>>>(v, NR, 2, I_ABSR2, 0, -1, 0);
            }
            break;
        case 5:
            if (!S->cheb) {
                iCtlKernel<<<1, 4, 0, s>>>(v.ctl, NR, 3, 0.0, 0, S->tol, S->maxIter, cheb);
                iDirectionKernel<NR, 0><<<nb, QGD_BLOCK, 0, s>>>(v);
            }
            break;
        default: throw std::invalid_argument("implicitSolvePhase: phase must be 0..5");
    }
    ICHECK(hipGetLastError());
}
static void iPhase(ImplicitSolver* S, int phase, bool fused) {
    if (S->NR == 4) iPhaseT<4>(S, phase, fused); else if (S->NR == 3) iPhaseT<3>(S, phase, fused); else iPhaseT<1>(S, phase, fused);
}
// a, diag, rhs, x: the system (diag, rhs, x component-major with stride nC, nRhs in {1, 3, 4}); validMask: bit k = component k is solved;
// gamma (nullptr: ones): the face coefficients of component k are gamma[k] * a
void implicitSolveSetup(ImplicitSolver* S, int nRhs, int validMask, const double* a, const double* diag, const double* rhs, double* x, double tol,
                        int maxIter, const double* gamma) {
    if (nRhs != 1 && nRhs != 3 && nRhs != 4) throw std::invalid_argument("implicitSolveSetup: 1, 3 or 4 right-hand sides");
    ISolveView& v = S->v;
    v.NR = nRhs; v.ob = S->ob; v.n = S->oe - S->ob; v.nC = S->m.nC; v.a = a; v.diag = diag; v.rhs = rhs; v.x = x;
    for (int k = 0; k < 4; ++k) v.gam[k] = gamma && k < nRhs ? gamma[k] : 1.0;
    v.r = S->r; v.d = S->d; v.df = S->df; v.q = S->q; v.xb = S->xb; v.part = S->part; v.ctl = S->ctl; v.nBlocks = gridOf(v.n);
    v.xrun = S->rowRun;
    S->NR = nRhs; S->validMask = validMask; S->tol = tol; S->maxIter = maxIter;
}
void implicitSolvePhase(ImplicitSolver* S, int phase) { iPhase(S, phase, false); }
static bool iAllDone(const ImplicitSolver* S, const double* h) {
    for (int k = 0; k < S->NR; ++k) if (h[ICTL(I_DONE, k)] == 0.0) return false;
    return true;
}
// {all done, iterations[k], initial[k], final[k]} for k < 3; waits for the stream
void implicitSolveStatus(ImplicitSolver* S, double* allDone, int iters[3], double res0[3], double res[3]) {
    double* h = S->hostCtl + 4 * I_COUNT;
    ICHECK(hipMemcpyAsync(h, S->ctl, sizeof(double) * I_COUNT, hipMemcpyDeviceToHost, S->stream));
    ICHECK(hipStreamSynchronize(S->stream));
    *allDone = iAllDone(S, h) ? 1.0 : 0.0;
    for (int k = 0; k < 3; ++k) {
        const bool on = k < S->NR && h[ICTL(I_DONE, k)] != 3.0;
        iters[k] = on ? (int)h[ICTL(I_ITER, k)] : 0; res0[k] = on ? h[ICTL(I_RES0, k)] : 0.0; res[k] = on ? h[ICTL(I_RES, k)] : 0.0;
    }
}
void implicitSolveStatus4(ImplicitSolver* S, double* allDone, int iters[4], double res0[4], double res[4]) {
    double* h = S->hostCtl + 4 * I_COUNT;
    ICHECK(hipMemcpyAsync(h, S->ctl, sizeof(double) * I_COUNT, hipMemcpyDeviceToHost, S->stream));
    ICHECK(hipStreamSynchronize(S->stream));
    *allDone = iAllDone(S, h) ? 1.0 : 0.0;
    for (int k = 0; k < 4; ++k) {
        const bool on = k < S->NR && h[ICTL(I_DONE, k)] != 3.0;
        iters[k] = on ? (int)h[ICTL(I_ITER, k)] : 0; res0[k] = on ? h[ICTL(I_RES0, k)] : 0.0; res[k] = on ? h[ICTL(I_RES, k)] : 0.0;
    }
}
double implicitSolverUnconverged(ImplicitSolver* S, double* stalledSteps) {
    double h[ISTAT_COUNT] = {0, 0, 0, 0};
    ICHECK(hipMemcpyAsync(h, S->stats, sizeof(h), hipMemcpyDeviceToHost, S->stream));
    ICHECK(hipStreamSynchronize(S->stream));
    if (stalledSteps) *stalledSteps = h[2];
    return h[0];
}
__global__ __launch_bounds__(QGD_BLOCK) void iVecHaloKernel(double* __restrict__ vec, const int nC, const int NR, const int32_t* __restrict__ cells,
                                                          const int nCells, double* __restrict__ buf, const int pack) {
    const int i = blockIdx.x * QGD_BLOCK + threadIdx.x;
    if (i >= nCells) return;
    const size_t ci = (size_t)cells[i];
    double* b = buf + (size_t)i * NR;
    for (int k = 0; k < NR; ++k) { if (pack) b[k] = vec[(size_t)k * nC + ci]; else vec[(size_t)k * nC + ci] = b[k]; }
}
void launchSolverHalo(hipStream_t s, ImplicitSolver* S, const int32_t* cells, int nCells, double* buf, bool pack) {
    if (nCells > 0) iVecHaloKernel<<<gridOf(nCells), QGD_BLOCK, 0, s>>>(S->cheb ? implicitSolverIterate(S) : S->d, S->m.nC, S->NR, cells, nCells, buf, pack ? 1 : 0);
    ICHECK(hipGetLastError());
}
// the whole solve after implicitSolveSetup (see pressureSolveRun): at most two iterations queued ahead of the last "done" flags read back
void implicitSolveRun(ImplicitSolver* S, const SolveHooks* hooks) {
    double* ctl = S->ctl;
    const bool sharded = hooks && hooks->allreduce;
    auto reduce = [&](int firstSlot, int slots) { if (sharded) hooks->allreduce(ctl + 4 * firstSlot, 4 * slots); };
    auto halo = [&]() { if (hooks && hooks->haloDirection) hooks->haloDirection(); };
    if (hooks && hooks->haloGuess) hooks->haloGuess();     // the neighbours' initial guesses into the ghost columns
    iPhase(S, 0, false);
    reduce(I_ABSR, 3);
    if (sharded && S->cheb) {
        if (!hooks->allreduceBuf) throw std::logic_error("implicitSolveRun: the Chebyshev solver needs a MAX all-reduce (SolveHooks::allreduceBuf)");
        hooks->allreduceBuf(ctl + 4 * I_DELTA, 4, 3);      // the spectral bound is the maximum over the ranks
    }
    iPhase(S, 1, false);
    reduce(I_NORM, 1);
    iPhase(S, 2, false);
    if (!S->cheb) reduce(I_RZ, 1);
    halo();
    const int ahead = 2;
    for (int it = 0; it < S->maxIter; ++it) {
        if (it >= ahead) {
            const int slot = (it - ahead) & 3;
            ICHECK(hipEventSynchronize(S->ev[slot]));
            if (iAllDone(S, S->hostCtl + slot * I_COUNT)) break;
        }
        iPhase(S, 3, !sharded);
        reduce(I_DQ, 1);                                   // conjugate gradients: d.q; Chebyshev: sum |b - A x_i| (the same slot)
        if (S->cheb) halo();                               // the new iterate's ghost entries do not wait for the control logic
        iPhase(S, 4, !sharded);
        if (!S->cheb) {
            reduce(I_ABSR2, 2);
            iPhase(S, 5, false);
            halo();
        }
        const int slot = it & 3;
        ICHECK(hipMemcpyAsync(S->hostCtl + slot * I_COUNT, ctl, sizeof(double) * I_COUNT, hipMemcpyDeviceToHost, S->stream));
        ICHECK(hipEventRecord(S->ev[slot], S->stream));
    }
}

// after a solve has finished: its control block is kept for qgd_case_implicit_info (which = 0: U, 1: e), the step's flag updated
void implicitSolveEnd(ImplicitSolver* S, int which) {
    if (S->cheb) iChebHomeKernel<<<S->v.nBlocks, QGD_BLOCK, 0, S->stream>>>(S->v);   // iterates that ended in the solver's buffer
    iStatKernel<<<1, 1, 0, S->stream>>>(S->ctl, S->stats, S->NR, 1, S->tol);
    ICHECK(hipMemcpyAsync(S->stats + ISTAT_COUNT + (size_t)which * I_COUNT, S->ctl, sizeof(double) * I_COUNT, hipMemcpyDeviceToDevice, S->stream));
}
void implicitStepMark(ImplicitSolver* S, bool begin) { iStatKernel<<<1, 1, 0, S->stream>>>(S->ctl, S->stats, 0, begin ? 0 : 2, 0.0); }
void implicitStatsReset(ImplicitSolver* S) { ICHECK(hipMemsetAsync(S->stats, 0, sizeof(double) * (ISTAT_COUNT + 2 * I_COUNT), S->stream)); }
// waits for the stream: iterations / initial / final residual of Ux, Uy, Uz, e in the last step, steps with an unconverged solve
void implicitSolverInfo(ImplicitSolver* S, int iters[4], double res0[4], double res[4], double* unconvergedSteps, double* stalledSteps) {
    std::vector<double> h(ISTAT_COUNT + 2 * I_COUNT);
    ICHECK(hipMemcpyAsync(h.data(), S->stats, sizeof(double) * h.size(), hipMemcpyDeviceToHost, S->stream));
    ICHECK(hipStreamSynchronize(S->stream));
    *unconvergedSteps = h[0];
    if (stalledSteps) *stalledSteps = h[2];
    const double* u = h.data() + ISTAT_COUNT;
    const double* e = u + I_COUNT;
    for (int k = 0; k < 3; ++k) {
        const bool on = u[ICTL(I_DONE, k)] != 3.0;
        iters[k] = on ? (int)u[ICTL(I_ITER, k)] : 0; res0[k] = on ? u[ICTL(I_RES0, k)] : 0.0; res[k] = on ? u[ICTL(I_RES, k)] : 0.0;
    }
    iters[3] = (int)e[ICTL(I_ITER, 0)]; res0[3] = e[ICTL(I_RES0, 0)]; res[3] = e[ICTL(I_RES, 0)];
}

int implicitHaloWidth(const ImplicitSolver* S, int kind) { return kind == 1 ? 9 : (kind == 2 ? 3 : S->NR); }   // kinds 3, 4: one per right-hand side
void launchImplicitHalo(hipStream_t s, const MeshView& m, const CaseView& c, const ImplView& iv, ImplicitSolver* S, int kind, const int32_t* cells,
                        int nCells, double* buf, bool pack) {
    if (nCells > 0) implHaloKernel<<<gridOf(nCells), QGD_BLOCK, 0, s>>>(c, iv, kind == 4 ? S->v.x : (S->cheb ? implicitSolverIterate(S) : S->d), m.nC, kind, S->NR, cells, nCells, buf, pack ? 1 : 0);
    ICHECK(hipGetLastError());
}
void implicitSolverSetStream(ImplicitSolver* S, hipStream_t s) { S->stream = s; }
// measurement: `reps` launches of the kernel the branch spends most of its time in -- one Chebyshev step of the three-component U
// system (iChebKernel<3, 0>: product, d, next iterate, partial residual sums) or, with QGD_IMPL_SOLVER=pcg, its matrix product
// q = A d (iApplyKernel<3, 1>) -- on the vectors the last step left, between two HIP events; returns the average ms.  The control
// block is cleared first (a finished solve makes every launch return at once); it is only kept for implicitSolverInfo, which
// reads its own copy.  Scratch vectors of the solver are overwritten (the search direction, the second iterate buffer): call between steps.
double implicitApplyMs(ImplicitSolver* S, const ImplView& iv, int reps, int* rows) {
    implicitSolveSetup(S, 3, 7, iv.aU, iv.diagU, iv.rhsU, iv.xU, S->tol, S->maxIter);
    const ISolveView& v = S->v;
    *rows = v.n;
    hipEvent_t a, b;
    ICHECK(hipEventCreate(&a)); ICHECK(hipEventCreate(&b));
    ICHECK(hipMemsetAsync(S->ctl, 0, sizeof(double) * I_COUNT, S->stream));   // every component "running", step 0, c1 = c2 = 0
    auto launch = [&]() {
        if (S->cheb) iChebKernel<3, 0><<<v.nBlocks, QGD_BLOCK, 0, S->stream>>>(S->m, v);
        else iApplyKernel<3, 1><<<v.nBlocks, QGD_BLOCK, 0, S->stream>>>(S->m, v);
    };
    launch();
    ICHECK(hipEventRecord(a, S->stream));
    for (int i = 0; i < reps; ++i) launch();
    ICHECK(hipEventRecord(b, S->stream));
    ICHECK(hipGetLastError());
    ICHECK(hipEventSynchronize(b));
    float ms = 0;
    ICHECK(hipEventElapsedTime(&ms, a, b));
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    return (double)ms / reps;
}

// ---- the species equation, implicitDiffusion branch [QGDYEqn.H L47-66], one species, as an operator on plain device fields ----------
//     fvScalarMatrix YEqn(fvm::ddt(rho,Yi) + fvc::div(phiJmYi) - fvm::laplacian(muf/ScNumbers[i],Yi) == R(Yi) + SYi);  YEqn.solve();
//     diffusiveFlux[i] += YEqn.flux();   Yi.max(0.0);
// Face coefficients a_f = (muf/Sc) |Sf| delta_f (Gauss, uncorrected snGrad, L0) at the faces' slot-major positions; patch faces flagged
// fixedValue add a_b to the diagonal and a_b Yb to the source, the others (zeroGradient) nothing.  YEqn.flux() is the matrix's own face
// flux (L0 fvMatrix::flux): -a_f (Y_N - Y_O) inside, -a_b (Y_b - Y_P) on fixedValue faces, of the NEW Y -- the sign the listing gets
// from "- fvm::laplacian", opposite to the explicit branch's "+ (muf/Sc) snGrad |Sf|" of L82; replicated as listed.
namespace {
__global__ __launch_bounds__(QGD_BLOCK) void speciesImplFaceKernel(const MeshView m, const double* __restrict__ muf, const double Sc, double* __restrict__ a) {
    const int f = blockIdx.x * QGD_BLOCK + threadIdx.x;
    if (f >= m.nF) return;
    const size_t pos = f < m.nIF ? (size_t)m.fpos[f] : (size_t)f;
    a[pos] = m.fkind[f] == 3 ? 0.0 : (muf[f] / Sc) * m.magSf[f] * m.dn[f];
}
__global__ __launch_bounds__(QGD_BLOCK) void speciesImplCellKernel(const MeshView m, const double* __restrict__ a, const double* __restrict__ Yc,
                                                                  const double* __restrict__ Yb, const uint8_t* __restrict__ fixedFace,
                                                                  const double* __restrict__ rhoOld, const double* __restrict__ rho,
                                                                  const double* __restrict__ phiJmY, const double dt, const double* __restrict__ Su,
                                                                  double* __restrict__ diag, double* __restrict__ rhs, double* __restrict__ x) {
    const int c = blockIdx.x * QGD_BLOCK + threadIdx.x;
    if (c >= m.nC) return;
    x[c] = Yc[c];
    if (m.ghost && m.ghost[c] == 1) { diag[c] = 1.0; rhs[c] = 0.0; return; }
    const int n = m.cfCount[c];
    const size_t base = (size_t)m.cfSlice[c >> 6] * 64 + (c & 63);
    double s = 0.0, dsum = 0.0, src = 0.0;
    for (int i = 0; i < n; ++i) {
        const int it = m.cfItem[base + (size_t)i * 64], ps = m.cfPos[base + (size_t)i * 64];
        const int f = it >= 0 ? it : ~it;
        if (m.fkind[f] == 3) continue;
        const double j = phiJmY[f];
        s = it >= 0 ? s + j : s - j;                       // fvc::div(phiJmYi): surfaceIntegrate, ascending face label
        const double af = a[ps >= 0 ? ps : ~ps];
        if (f < m.nIF) dsum += af;
        else if (fixedFace && fixedFace[f - m.nIF]) { dsum += af; src += af * Yb[f - m.nIF]; }
    }
    const double V = m.V[c], rD = 1.0 / dt;
    diag[c] = rD * rho[c] * V + dsum;
    double r = rD * rhoOld[c] * Yc[c] * V - s;
    if (Su) r += V * Su[c];
    rhs[c] = r + src;
}
__global__ __launch_bounds__(QGD_BLOCK) void speciesImplFluxKernel(const MeshView m, const double* __restrict__ a, const double* __restrict__ x,
                                                                  const double* __restrict__ Yb, const uint8_t* __restrict__ fixedFace,
                                                                  double* __restrict__ diffusiveFlux) {
    const int f = blockIdx.x * QGD_BLOCK + threadIdx.x;
    if (f >= m.nF) return;
    if (m.fkind[f] == 3) return;
    const size_t pos = f < m.nIF ? (size_t)m.fpos[f] : (size_t)f;
    double fl = 0.0;
    if (f < m.nIF) fl = -(a[pos] * (x[m.nei[f]] - x[m.own[f]]));
    else if (fixedFace && fixedFace[f - m.nIF]) fl = -(a[pos] * (Yb[f - m.nIF] - x[m.own[f]]));
    diffusiveFlux[f] += fl;                                                                       // L64
}
__global__ __launch_bounds__(QGD_BLOCK) void speciesImplMaxKernel(const int n, const double* __restrict__ x, double* __restrict__ Ynew) {
    const int c = blockIdx.x * QGD_BLOCK + threadIdx.x;
    if (c < n) Ynew[c] = fmax(x[c], 0.0);                                                          // L86
}
}  // namespace
// work: 3 * nC + nF doubles of device scratch (diag, rhs, x | a).  Returns after queueing; info (host) is filled after a wait.
void launchSpeciesStepImplicit(ImplicitSolver* S, const MeshView& m, const double* Yc, const double* Yb, const uint8_t* fixedFace, const double* rhoOld,
                               const double* rho, const double* phiJmY, const double* muf, double Sc, double dt, const double* Su, double tol, int maxIter,
                               double* work, double* diffusiveFlux, double* Ynew, double info[3]) {
    hipStream_t s = S->stream;
    const size_t nC = (size_t)m.nC;
    double *diag = work, *rhs = work + nC, *x = work + 2 * nC, *a = work + 3 * nC;
    if (m.nF) speciesImplFaceKernel<<<gridOf(m.nF), QGD_BLOCK, 0, s>>>(m, muf, Sc, a);
    speciesImplCellKernel<<<gridOf(m.nC), QGD_BLOCK, 0, s>>>(m, a, Yc, Yb, fixedFace, rhoOld, rho, phiJmY, dt, Su, diag, rhs, x);
    implicitSolveSetup(S, 1, 1, a, diag, rhs, x, tol, maxIter);
    implicitSolveRun(S, nullptr);
    implicitSolveEnd(S, 1);
    if (m.nF) speciesImplFluxKernel<<<gridOf(m.nF), QGD_BLOCK, 0, s>>>(m, a, x, Yb, fixedFace, diffusiveFlux);
    speciesImplMaxKernel<<<gridOf(m.nC), QGD_BLOCK, 0, s>>>(m.nC, x, Ynew);
    ICHECK(hipGetLastError());
    double done = 0;
    int it[4];
    double r0[4], r1[4];
    implicitSolveStatus4(S, &done, it, r0, r1);
    info[0] = it[0]; info[1] = r0[0]; info[2] = r1[0];
}

// ---- the advance as phases (stream-ordered; a sharded caller exchanges between them, see include/qgd_amd.h) ---------------
//   A  fvc::grad(U) of the state before the step                                   -> ghost gradients
//   B  tauMC / phiTauMC and the laplacian coefficients per face, rho, rhoU, U = rhoU/rho, the U systems (their solve follows)
//   C  U of the records, U's boundary conditions                                   -> ghost velocities
//   D  fvc::grad(U) of the new velocity                                            -> ghost gradients
//   E  phiSigmaDotU, the energy equation's explicit part, the e system (its solve follows)
//   F  rhoE = rho (e + |U|^2/2), thermo, p
void launchImplicitPart(hipStream_t s, const MeshView& m, const CaseView& c, const ImplView& iv, const GasModel& g, const PatchBCDev* bc,
                        ImplicitSolver* S, double tol, int maxIter, int part, bool fusedU) {
    const int gc = gridOf(m.nC), gf = gridOf(m.nF), gb = gridOf(m.nBF);
    switch (part) {
        case 0: implCellGradKernel<<<gc, QGD_BLOCK, 0, s>>>(m, c, iv); break;
        case 1: {
            if (fusedU) {
                // the block-fused assembly (qgd_kernels.hip fusedFaceCellKernel<..., IMPL>): the patch faces first -- the blocks read their
                // phiTauMC and laplacian coefficients --, then vertex values, QGD fluxes, tauMC and the rows of the U systems in ONE launch
                if (m.nBF > 0) implFaceKernel<<<gb, QGD_BLOCK, 0, s>>>(m, c, iv, g, bc, nullptr, m.nIF);
                const Launcher L{s, nullptr, nullptr, nullptr};
                launchFusedImplU(L, m, c, g, iv, bc);
            } else {
            if (m.tileOff != nullptr && m.fblock == 128 && m.implTiles) {
                // internal faces of the staged tiles out of LDS, the tiles beyond the caps and the boundary faces through the generic walk
                const size_t lds = ((size_t)m.tileMaxC * 120 + 255) / 256 * 256;
                implFaceTileKernel<<<(m.nIF + 127) / 128, 128, lds, s>>>(m, c, iv, g);
                if (m.nTileSpill > 0) implFaceKernel<<<m.nTileSpill, 128, 0, s>>>(m, c, iv, g, bc, m.tileSpill, 0);
                if (m.nBF > 0) implFaceKernel<<<gb, QGD_BLOCK, 0, s>>>(m, c, iv, g, bc, nullptr, m.nIF);
            } else implFaceKernel<<<gf, QGD_BLOCK, 0, s>>>(m, c, iv, g, bc, nullptr, 0);
            implCellUKernel<<<gc, QGD_BLOCK, 0, s>>>(m, c, iv, bc);
            }
            int mask = 0;
            for (int k = 0; k < 3; ++k) if (!(m.nGeomD < 3 && m.emptyDir[k])) mask |= 1 << k;   // validComponents (L0)
            implicitSolveSetup(S, 3, mask, iv.aU, iv.diagU, iv.rhsU, iv.xU, tol, maxIter);
            break;
        }
        case 2:
            implStoreUKernel<<<gc, QGD_BLOCK, 0, s>>>(m, c, iv);
            if (m.nBF) implBcUKernel<<<gb, QGD_BLOCK, 0, s>>>(m, c, bc);
            break;
        case 3: implCellGradKernel<<<gc, QGD_BLOCK, 0, s>>>(m, c, iv); break;
        case 4:
            implSigmaKernel<<<gf, QGD_BLOCK, 0, s>>>(m, c, iv, bc);
            implCellEKernel<<<gc, QGD_BLOCK, 0, s>>>(m, c, iv, g, bc);
            implicitSolveSetup(S, 1, 1, iv.aE, iv.diagE, iv.rhsE, iv.xE, tol, maxIter);
            break;
        case 5: implFinishKernel<<<gc, QGD_BLOCK, 0, s>>>(m, c, iv, g); break;
        default: throw std::invalid_argument("launchImplicitPart: part must be 0..5");
    }
    ICHECK(hipGetLastError());
}

}  // namespace qgd
