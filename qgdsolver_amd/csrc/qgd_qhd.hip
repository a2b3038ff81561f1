// qgd_qhd.hip -- QHDFoam's loop body resident on the device (QHDFoam_8C_source.html L83-139; both branches of implicitDiffusion:
// fvc::laplacian in the face terms, or fvm::laplacian as ONE four-component system solved by qgd_implicit.hip's solver):
//
//   updateFields.H L36-73      gradUf, gradTf, Uf, Tf, BdFrcf                          face pass 1
//   updateFluxes.H L33-38      phiu, phiwo, taubyrhof                                  face pass 1
//   QHDpEqn.H L35-47           p BCs, pressure equation, phi                           qgd_poisson.hip
//   QHDUEqn.H L36-84           gradPf, Wf, phiUf, fvc::laplacian, fvc::div(nu Sf & lin(T(grad U))), -grad(p)/rho + BdFrc
//   QHDTEqn.H L65-91           phiTf, phiTauTReg, fvc::laplacian(Hif, T)               face pass 2 + cell update
//   QHDFoam.C L123-130         reference level of p
//
// Thermo: rhoConst + constTransport (rho, mu, alpha = mu/Pr uniform; the QHD closures keep muQGD = alphauQGD = 0,
// T0byGr_8C L62-72), not corrected inside the loop (the listing never calls thermo.correct() there), so tauQGDf and the
// pressure matrix are those of start-up.  State: cell records {Ux,Uy,Uz,T} (32 B) and p separately (the solver's vector).
#include "../../include/qgd_amd.h"
#include "qgd_device.hpp"
#include "qgd_stencil_dev.hpp"

namespace qgd {

namespace {

// patch snGrad of {U,T} (4 components) on boundary face f: fvPatchField::snGrad = deltaCoeffs*(value - internal) (L0);
// basicSymmetry::snGrad for slip
// perComponent (the 2-D GaussVolPoint gradient of U [GaussVolPointBase.C L79-87]: one scalar field per component, whose patch field on a
// symmetryPlane / symmetry patch is the scalar symmetry one, snGrad = 0 -- L0: fvPatchField::New lets the constraint patch type win)
__device__ __forceinline__ void qhdBoundaryVals4(const MeshView& m, const PatchBCDev& bc, const int f, const double* o4, const double* b4,
                                                 FaceVals<4>& v, const bool perComponent = false) {
    const double dc = m.dn[f];
#pragma unroll
    for (int k = 0; k < 4; ++k) { v.o[k] = o4[k]; v.n[k] = b4[k]; v.sn[k] = dc * (b4[k] - o4[k]); }
    if (bc.bcU == QGD_BC_SLIP) {
        double n[3];
        symmNormal(m, bc, f, n);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const double tv = ((i == 0 ? 1.0 : 0.0) - 2.0 * (n[i] * n[0])) * o4[0] + ((i == 1 ? 1.0 : 0.0) - 2.0 * (n[i] * n[1])) * o4[1] +
                              ((i == 2 ? 1.0 : 0.0) - 2.0 * (n[i] * n[2])) * o4[2];
            v.sn[i] = (tv - o4[i]) * (dc / 2.0);
        }
    } else if (bc.bcU != QGD_BC_FIXEDVALUE) v.sn[0] = v.sn[1] = v.sn[2] = 0.0;
    if (bc.bcT != QGD_BC_FIXEDVALUE) v.sn[3] = 0.0;
    if (perComponent && (bc.ptype == QGD_PATCH_SYMMETRYPLANE || bc.ptype == QGD_PATCH_SYMMETRY)) v.sn[0] = v.sn[1] = v.sn[2] = 0.0;
}

// face pass 1 [updateFields.H L36-73, updateFluxes.H L33-38, QHDTEqn.H L66]
// (tileList != nullptr: 128 threads per workgroup, the 128-face tiles the staged kernel leaves to this one; else the faces from fBegin on)
template <int ST>
__global__ __launch_bounds__(QGD_BLOCK) void qhdFace1Kernel(const MeshView m, const QhdView q, const PatchBCDev* __restrict__ bcs,
                                                            const int32_t* __restrict__ tileList, const int fBegin) {
    const int f = tileList ? tileList[blockIdx.x] * 128 + (int)threadIdx.x
                           : fBegin + xcdTile((int)gridDim.x, m.xcdRun) * QGD_BLOCK + (int)threadIdx.x;   // runs of consecutive blocks per XCD: neighbours meet in one L2
    if (f >= (tileList ? m.nIF : m.nF)) return;
    if (m.fkind[f] == 3) return;
    const bool internal = f < m.nIF;
    const int o = m.own[f];
    FaceVals<4> v;
    double w = 1.0;
    if (internal) {
        const int n = m.nei[f];
#pragma unroll
        for (int k = 0; k < 4; ++k) { v.o[k] = q.c4[(size_t)o * 4 + k]; v.n[k] = q.c4[(size_t)n * 4 + k]; v.sn[k] = 0.0; }
        w = m.w[f];
    } else {
        const int b = f - m.nIF;
        qhdBoundaryVals4(m, bcs[m.bPatch[b]], f, q.c4 + (size_t)o * 4, q.b4 + (size_t)b * 4, v, ST == ST_GVP2);
    }
    double g[12];
    faceGradient<ST, 4, 0>(m, f, v, q.c4, q.pt4, g);   // g[i*4 + k] = d_i {Ux,Uy,Uz,T}_k
    const double gv[3] = {q.g[0], q.g[1], q.g[2]};
    double Uf[3], Bf[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        Uf[k] = internal ? lerpf(w, v.o[k], v.n[k]) : v.n[k];
        Bf[k] = internal ? lerpf(w, (q.beta * v.o[3]) * gv[k], (q.beta * v.n[3]) * gv[k]) : (q.beta * v.n[3]) * gv[k];   // L66-67
    }
    const double S[3] = {m.Sx[f], m.Sy[f], m.Sz[f]};
    const double tau = q.tauF[f];
    const size_t nF = (size_t)m.nF;
    const double phiu = S[0] * Uf[0] + S[1] * Uf[1] + S[2] * Uf[2];
    double wo[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const double UgU = Uf[0] * g[0 * 4 + j] + Uf[1] * g[1 * 4 + j] + Uf[2] * g[2 * 4 + j];   // Uf & gradUf
        q.ugu[(size_t)j * nF + f] = UgU;   // (BdFrcf is not stored: face pass 2 holds T of both cells and forms it again from the same expression)
        wo[j] = tau * (UgU - Bf[j]);
    }
    q.phiu[f] = phiu;
    q.phiwo[f] = S[0] * wo[0] + S[1] * wo[1] + S[2] * wo[2];
    q.phitr[f] = tau * phiu * (Uf[0] * g[0 * 4 + 3] + Uf[1] * g[1 * 4 + 3] + Uf[2] * g[2 * 4 + 3]);
}

// p's boundary conditions [QHDpEqn.H L35]: fixedValue | fixedGradient (qhdFlux with the gradient of its file) |
// qhdFlux fed by the registered flux [qhdFluxFvPatchScalarField_8C L193-203] | zeroGradient
__global__ __launch_bounds__(QGD_BLOCK) void qhdPressureBcKernel(const MeshView m, const QhdView q, const PatchBCDev* __restrict__ bcs) {
    const int b = blockIdx.x * QGD_BLOCK + threadIdx.x;
    if (b >= m.nBF) return;
    const int f = m.nIF + b;
    if (m.fkind[f] == 3) return;
    const PatchBCDev bc = bcs[m.bPatch[b]];
    const double po = q.p[m.own[f]];
    double pb = po, gb = 0.0;
    if (bc.bcP == QGD_BC_FIXEDVALUE) pb = bc.vP;
    else if (bc.bcP == QGD_BC_QGDFLUX) { gb = bc.vP; pb = po + gb / m.dn[f]; }
    else if (bc.bcP == QGD_BC_QHDFLUX) { gb = -(q.phiwo[f] / q.tauF[f] * q.rho0 / m.magSf[f]); pb = po + gb / m.dn[f]; }
    q.pb[b] = pb;
    q.pgb[b] = gb;
}

// fvc::grad(U), Gauss linear (L0): cell gather in ascending face order -- the cell's own velocity once, per face the neighbour cell's
// (cfNbr) or the patch value, the weight and Sf (same operations, same order as walking owner and neighbour face by face)
__global__ __launch_bounds__(QGD_BLOCK) void qhdCellGradKernel(const MeshView m, const QhdView q) {
    const int c = xcdTile((int)gridDim.x, m.xcdRun) * QGD_BLOCK + threadIdx.x;   // runs of consecutive blocks per XCD: neighbours meet in one L2
    const bool live = c < m.nC && !(m.ghost && m.ghost[c] == 1);   // a ghost cell lacks faces here: its gradient arrives with the halo message
    if (__ballot(live) == 0) return;
    if (live) cellGradGauss<4, 0, 4, 0>(m, c, q.c4, q.b4, q.gUc);
}

// face pass 2 [QHDUEqn.H L36-84, QHDTEqn.H L65-91]: the net face terms of the U and T equations
template <int ST>
__global__ __launch_bounds__(QGD_BLOCK) void qhdFace2Kernel(const MeshView m, const QhdView q, const PatchBCDev* __restrict__ bcs,
                                                            const int32_t* __restrict__ tileList, const int fBegin) {
    const int f = tileList ? tileList[blockIdx.x] * 128 + (int)threadIdx.x
                           : fBegin + xcdTile((int)gridDim.x, m.xcdRun) * QGD_BLOCK + (int)threadIdx.x;   // runs of consecutive blocks per XCD: neighbours meet in one L2
    if (f >= (tileList ? m.nIF : m.nF)) return;
    // the net face terms go to the face's slot-major position (MeshView::fpos), where the cell update finds those of consecutive cells
    // at consecutive addresses (by label: every third double of the lines it fetches)
    const size_t nF = (size_t)m.nF, pos = f < m.nIF ? (size_t)m.fpos[f] : (size_t)f;
    if (m.fkind[f] == 3) { for (int k = 0; k < 4; ++k) q.F[(size_t)k * nF + pos] = 0.0; return; }
    const bool internal = f < m.nIF;
    const int o = m.own[f];
    const double S[3] = {m.Sx[f], m.Sy[f], m.Sz[f]};
    const double magS = m.magSf[f];
    double Uo[3], Un[3], To, Tn, snU[3], snT, pf, w = 1.0;
    FaceVals<1> vp;
    vp.o[0] = q.p[o];
    double gUT[9];   // T(grad U) at the face: gUT[3i+j] = lin(gradU)[3j+i]
#pragma unroll
    for (int k = 0; k < 3; ++k) Uo[k] = q.c4[(size_t)o * 4 + k];
    To = q.c4[(size_t)o * 4 + 3];
    if (internal) {
        const int n = m.nei[f];
        w = m.w[f];
#pragma unroll
        for (int k = 0; k < 3; ++k) Un[k] = q.c4[(size_t)n * 4 + k];
        Tn = q.c4[(size_t)n * 4 + 3];
        vp.n[0] = q.p[n]; vp.sn[0] = 0.0;
        const double dn = m.dn[f];
#pragma unroll
        for (int k = 0; k < 3; ++k) snU[k] = dn * (Un[k] - Uo[k]);   // fvc::snGrad, uncorrected (L0)
        snT = dn * (Tn - To);
        pf = lerpf(w, vp.o[0], vp.n[0]);
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) gUT[3 * i + j] = lerpf(w, q.gUc[(size_t)o * 9 + 3 * j + i], q.gUc[(size_t)n * 9 + 3 * j + i]);
    } else {
        const int b = f - m.nIF;
        const PatchBCDev bc = bcs[m.bPatch[b]];
        FaceVals<4> v4;
        qhdBoundaryVals4(m, bc, f, q.c4 + (size_t)o * 4, q.b4 + (size_t)b * 4, v4);
#pragma unroll
        for (int k = 0; k < 3; ++k) { Un[k] = v4.n[k]; snU[k] = v4.sn[k]; }
        Tn = v4.n[3]; snT = v4.sn[3];
        vp.n[0] = q.pb[b];
        vp.sn[0] = (bc.bcP == QGD_BC_QGDFLUX || bc.bcP == QGD_BC_QHDFLUX) ? q.pgb[b] : (bc.bcP == QGD_BC_FIXEDVALUE ? m.dn[f] * (vp.n[0] - vp.o[0]) : 0.0);
        pf = vp.n[0];
        // patch value of fvc::grad(U): the owner's gradient with its normal part replaced by the patch snGrad
        // (gaussGrad::correctBoundaryConditions, L0); cut (halo) faces keep the extrapolated value
        double gb[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) gb[k] = q.gUc[(size_t)o * 9 + k];
        if (bc.ptype != QGD_PATCH_HALO && bc.ptype != QGD_PATCH_CYCLIC) {
            const double n[3] = {S[0] / magS, S[1] / magS, S[2] / magS};
            double ng[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) ng[j] = n[0] * gb[j] + n[1] * gb[3 + j] + n[2] * gb[6 + j];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) gb[3 * i + j] += n[i] * (snU[j] - ng[j]);
        }
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) gUT[3 * i + j] = gb[3 * j + i];
    }
    double gP[3];
    faceGradient<ST, 1, -1>(m, f, vp, q.p, q.ptp, gP);                                           // QHDUEqn.H L36
    const double tau = q.tauF[f], phi = q.phi[f];
    double Uf[3], Wf[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        Uf[k] = internal ? lerpf(w, Uo[k], Un[k]) : Un[k];
        const double Bf = internal ? lerpf(w, (q.beta * To) * q.g[k], (q.beta * Tn) * q.g[k]) : (q.beta * Tn) * q.g[k];   // BdFrcf [updateFields.H L66-67], as in face pass 1
        Wf[k] = tau * ((q.ugu[(size_t)k * nF + f] + gP[k] / q.rho0) - Bf);   // L37
    }
    const double Tf = internal ? lerpf(w, To, Tn) : Tn;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const double uw = S[0] * (Uf[0] * Wf[j]) + S[1] * (Uf[1] * Wf[j]) + S[2] * (Uf[2] * Wf[j]);   // Sf & (Uf*Wf), L39
        // qgdFlux(phi,U,Uf) [L41]: phi*Uf, or with `div(phi,U) Gauss upwind` fvc::flux = phi * (pos0(phi) (U_O - U_N) + U_N) inside,
        // the patch value on patch faces [QGDInterpolate.H L86-104]
        const double phiUf = phi * ((q.upwindU && internal) ? lerpf(phi >= 0.0 ? 1.0 : 0.0, Uo[j], Un[j]) : Uf[j]) - uw;   // L41-43
        const double lap = q.nu * snU[j] * magS;                                                     // fvc::laplacian(muf/rhof, U), L74
        const double ext = S[0] * gUT[0 * 3 + j] + S[1] * gUT[1 * 3 + j] + S[2] * gUT[2 * 3 + j];    // Sf & lin(T(grad U)), L56 / L76
        // the Gauss term of -fvc::grad(p)/rho (uniform rho) rides in the same face flux: S_j p_f / rho;
        // implicitDiffusion: the laplacian sits in the matrix (fvm::laplacian, L54)
        q.F[(size_t)j * nF + pos] = (q.implicit ? phiUf - q.nu * ext : (phiUf - lap) - q.nu * ext) + (S[j] * pf) / q.rho0;
    }
    const double phiTf = phi * ((q.upwindT && internal) ? lerpf(phi >= 0.0 ? 1.0 : 0.0, To, Tn) : Tf);   // qgdFlux(phi,T,Tf) [QHDTEqn.H L65]
    q.F[3 * nF + pos] = q.implicit ? phiTf - q.phitr[f]                                                // QHDTEqn.H L73-76
                                   : (phiTf - q.Hi * snT * magS) - q.phitr[f];                         // QHDTEqn.H L65-66, L85-88
}

// ---------------------------------------------------------------------------
// The two face passes through LDS on 3-D GaussVolPoint meshes, like the QGD face kernel (qgd_kernels.hip faceFluxGvp3TileKernel,
// qgd_setup.hpp FaceTiles): a workgroup owns a tile of 128 consecutive internal faces, brings every DISTINCT cell / vertex record of
// the tile in once as consecutive 16-B or 8-B pieces (lane q takes piece q: whole cache lines per wave instruction instead of one
// scattered record per lane) and every face picks its records out of LDS.  Pass 2 gains most: fvc::grad(U) alone is 2 x 72 B per
// face as nine scattered doubles in the generic walk.  Same expressions as the generic kernels (which keep the boundary faces and
// the tiles beyond the caps).
// ---------------------------------------------------------------------------
typedef double v2d __attribute__((ext_vector_type(2)));
constexpr int kQhdFB = 128;
static_assert(2 * (kQhdFB + kQhdFB / 16) <= 3 * kQhdFB && 3 * (kQhdFB + kQhdFB / 16) <= 4 * kQhdFB && 9 * (kQhdFB + kQhdFB / 16) <= 10 * kQhdFB &&
              2 * (((kQhdFB * 23) / 16 + 7) / 8 * 8) <= 3 * kQhdFB && 3 * (((kQhdFB * 23) / 16 + 7) / 8 * 8) <= 5 * kQhdFB, "faceTileCap*");

// face pass 1 on the staged tiles: the internal-face branch of qhdFace1Kernel
__global__ __launch_bounds__(kQhdFB) void qhdFace1TileKernel(const MeshView m, const QhdView q) {
    constexpr int FB = kQhdFB;
    extern __shared__ v2d qhdLds[];
    const int tile = xcdTile((int)gridDim.x, m.xcdRun * (QGD_BLOCK / FB));
    const int tid = (int)threadIdx.x;
    const int f = tile * FB + tid;
    const bool active = f < m.nIF;
    const int cOff = m.tileOff[2 * tile], vOff = m.tileOff[2 * tile + 1];
    const int nUc = m.tileOff[2 * tile + 2] - cOff, nUv = m.tileOff[2 * tile + 3] - vOff;
    if (nUc == 0) return;   // beyond the caps: in m.tileSpill, done by the generic kernel
    v2d* const s4 = qhdLds;                                    // 2 nUc pieces: cell {U,T}
    v2d* const sP = s4 + 2 * nUc;                              // 2 nUv: vertex {U,T}
    double* const sC = reinterpret_cast<double*>(sP + 2 * nUv);   // 3 nUc: cell centres
    double* const sX = sC + 3 * nUc;                           // 3 nUv: vertex coordinates
    const int fl = active ? f : m.nIF - 1;
    const unsigned lc = ldStream(m.locC + fl);
    const uint2 lv = m.locV[fl];
    const int kind = m.fkind[fl];
    const double w = ldStream(m.w + fl), tau = ldStream(q.tauF + fl);
    const double S[3] = {ldStream(m.Sx + fl), ldStream(m.Sy + fl), ldStream(m.Sz + fl)};
    TileStager<v2d, 3, 2, FB> g4, gP;
    TileStager<double, 4, 3, FB> gC;
    TileStager<double, 5, 3, FB> gX;
    g4.load(reinterpret_cast<const v2d*>(q.c4), m.tileCells + cOff, nUc, tid);
    gC.load(m.Cc, m.tileCells + cOff, nUc, tid);
    gP.load(reinterpret_cast<const v2d*>(q.pt4), m.tileVerts + vOff, nUv, tid);
    gX.load(m.X, m.tileVerts + vOff, nUv, tid);
    __builtin_amdgcn_sched_barrier(0);
    g4.store(s4, nUc, tid); gC.store(sC, nUc, tid); gP.store(sP, nUv, tid); gX.store(sX, nUv, tid);
    __syncthreads();
    if (!active) return;
    const int lo = (int)(lc & 0xffffu), ln = (int)(lc >> 16);
    const int v0 = (int)(lv.x & 0xffffu), v1 = (int)(lv.x >> 16), v2 = (int)(lv.y & 0xffffu), v3 = (int)(lv.y >> 16);
    auto l3 = [](const double* p, int i) { return make_double4(p[3 * i], p[3 * i + 1], p[3 * i + 2], 0.0); };
    auto l4 = [](const v2d* p, int i, double* o) { const v2d a = p[2 * i], b = p[2 * i + 1]; o[0] = a.x; o[1] = a.y; o[2] = b.x; o[3] = b.y; };
    FaceVals<4> v;
    l4(s4, lo, v.o); l4(s4, ln, v.n);
#pragma unroll
    for (int k = 0; k < 4; ++k) v.sn[k] = 0.0;
    double g[12];
    if (kind == 2) faceGradient<ST_GVP3, 4, 0>(m, f, v, q.c4, q.pt4, g);   // more than four vertices: nf (x) snGrad, out of global memory
    else {
        double q0[4], q1[4], q2[4], q3[4];
        l4(sP, v0, q0); l4(sP, v1, q1); l4(sP, v2, q2); l4(sP, v3, q3);
        gvp3GradCore<4, 0>(kind, true, l3(sC, lo), l3(sC, ln), l3(sX, v0), l3(sX, v1), l3(sX, v2), l3(sX, v3), v.o, v.n, q0, q1, q2, q3, g);
    }
    const double gv[3] = {q.g[0], q.g[1], q.g[2]};
    double Uf[3], Bf[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        Uf[k] = lerpf(w, v.o[k], v.n[k]);
        Bf[k] = lerpf(w, (q.beta * v.o[3]) * gv[k], (q.beta * v.n[3]) * gv[k]);   // L66-67
    }
    const size_t nF = (size_t)m.nF;
    const double phiu = S[0] * Uf[0] + S[1] * Uf[1] + S[2] * Uf[2];
    double wo[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const double UgU = Uf[0] * g[0 * 4 + j] + Uf[1] * g[1 * 4 + j] + Uf[2] * g[2 * 4 + j];   // Uf & gradUf
        q.ugu[(size_t)j * nF + f] = UgU;
        wo[j] = tau * (UgU - Bf[j]);
    }
    q.phiu[f] = phiu;
    q.phiwo[f] = S[0] * wo[0] + S[1] * wo[1] + S[2] * wo[2];
    q.phitr[f] = tau * phiu * (Uf[0] * g[0 * 4 + 3] + Uf[1] * g[1 * 4 + 3] + Uf[2] * g[2 * 4 + 3]);
}

// face pass 2 on the staged tiles: the internal-face branch of qhdFace2Kernel
__global__ __launch_bounds__(kQhdFB) void qhdFace2TileKernel(const MeshView m, const QhdView q) {
    constexpr int FB = kQhdFB;
    extern __shared__ v2d qhdLds[];
    const int tile = xcdTile((int)gridDim.x, m.xcdRun * (QGD_BLOCK / FB));
    const int tid = (int)threadIdx.x;
    const int f = tile * FB + tid;
    const bool active = f < m.nIF;
    const int cOff = m.tileOff[2 * tile], vOff = m.tileOff[2 * tile + 1];
    const int nUc = m.tileOff[2 * tile + 2] - cOff, nUv = m.tileOff[2 * tile + 3] - vOff;
    if (nUc == 0) return;
    v2d* const s4 = qhdLds;                                       // 2 nUc pieces: cell {U,T}
    double* const sG = reinterpret_cast<double*>(s4 + 2 * nUc);   // 9 nUc: fvc::grad(U)
    double* const sC = sG + 9 * nUc;                              // 3 nUc: cell centres
    double* const sp = sC + 3 * nUc;                              // nUc: p
    double* const sX = sp + nUc;                                  // 3 nUv: vertex coordinates
    double* const sq = sX + 3 * nUv;                              // nUv: vertex p
    const int fl = active ? f : m.nIF - 1;
    const unsigned lc = ldStream(m.locC + fl);
    const uint2 lv = m.locV[fl];
    const int kind = m.fkind[fl];
    const size_t nF = (size_t)m.nF;
    const size_t pos = (size_t)ldStream(m.fpos + fl);
    const double w = ldStream(m.w + fl), dn = ldStream(m.dn + fl), magS = ldStream(m.magSf + fl);
    const double S[3] = {ldStream(m.Sx + fl), ldStream(m.Sy + fl), ldStream(m.Sz + fl)};
    const double tau = ldStream(q.tauF + fl), phi = q.phi[fl], phitr = q.phitr[fl];
    const double ugu[3] = {q.ugu[fl], q.ugu[nF + fl], q.ugu[2 * nF + fl]};
    TileStager<v2d, 3, 2, FB> g4;
    TileStager<double, 10, 9, FB> gG;
    TileStager<double, 4, 3, FB> gC;
    TileStager<double, 2, 1, FB> gp, gq;
    TileStager<double, 5, 3, FB> gX;
    g4.load(reinterpret_cast<const v2d*>(q.c4), m.tileCells + cOff, nUc, tid);
    gG.load(q.gUc, m.tileCells + cOff, nUc, tid);
    gC.load(m.Cc, m.tileCells + cOff, nUc, tid);
    gp.load(q.p, m.tileCells + cOff, nUc, tid);
    gX.load(m.X, m.tileVerts + vOff, nUv, tid);
    gq.load(q.ptp, m.tileVerts + vOff, nUv, tid);
    __builtin_amdgcn_sched_barrier(0);
    g4.store(s4, nUc, tid); gG.store(sG, nUc, tid); gC.store(sC, nUc, tid); gp.store(sp, nUc, tid); gX.store(sX, nUv, tid); gq.store(sq, nUv, tid);
    __syncthreads();
    if (!active) return;
    const int lo = (int)(lc & 0xffffu), ln = (int)(lc >> 16);
    const int v0 = (int)(lv.x & 0xffffu), v1 = (int)(lv.x >> 16), v2 = (int)(lv.y & 0xffffu), v3 = (int)(lv.y >> 16);
    auto l3 = [](const double* p, int i) { return make_double4(p[3 * i], p[3 * i + 1], p[3 * i + 2], 0.0); };
    auto l4 = [](const v2d* p, int i, double* o) { const v2d a = p[2 * i], b = p[2 * i + 1]; o[0] = a.x; o[1] = a.y; o[2] = b.x; o[3] = b.y; };
    if (kind == 3) { for (int k = 0; k < 4; ++k) q.F[(size_t)k * nF + pos] = 0.0; return; }
    double o4[4], n4[4];
    l4(s4, lo, o4); l4(s4, ln, n4);
    const double Uo[3] = {o4[0], o4[1], o4[2]}, Un[3] = {n4[0], n4[1], n4[2]}, To = o4[3], Tn = n4[3];
    FaceVals<1> vp;
    vp.o[0] = sp[lo]; vp.n[0] = sp[ln]; vp.sn[0] = 0.0;
    double snU[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) snU[k] = dn * (Un[k] - Uo[k]);   // fvc::snGrad, uncorrected (L0)
    const double snT = dn * (Tn - To);
    const double pf = lerpf(w, vp.o[0], vp.n[0]);
    double gUT[9];   // T(grad U) at the face: gUT[3i+j] = lin(gradU)[3j+i]
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) gUT[3 * i + j] = lerpf(w, sG[9 * lo + 3 * j + i], sG[9 * ln + 3 * j + i]);
    double gP[3];
    if (kind == 2) faceGradient<ST_GVP3, 1, -1>(m, f, vp, q.p, q.ptp, gP);
    else gvp3GradCore<1, -1>(kind, true, l3(sC, lo), l3(sC, ln), l3(sX, v0), l3(sX, v1), l3(sX, v2), l3(sX, v3), vp.o, vp.n, sq + v0, sq + v1, sq + v2,
                             sq + v3, gP);                                                       // QHDUEqn.H L36
    double Uf[3], Wf[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        Uf[k] = lerpf(w, Uo[k], Un[k]);
        const double Bf = lerpf(w, (q.beta * To) * q.g[k], (q.beta * Tn) * q.g[k]);   // BdFrcf [updateFields.H L66-67], as in face pass 1
        Wf[k] = tau * ((ugu[k] + gP[k] / q.rho0) - Bf);   // L37
    }
    const double Tf = lerpf(w, To, Tn);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const double uw = S[0] * (Uf[0] * Wf[j]) + S[1] * (Uf[1] * Wf[j]) + S[2] * (Uf[2] * Wf[j]);   // Sf & (Uf*Wf), L39
        const double phiUf = phi * (q.upwindU ? lerpf(phi >= 0.0 ? 1.0 : 0.0, Uo[j], Un[j]) : Uf[j]) - uw;   // L41-43 (upwind: see qhdFace2Kernel)
        const double lap = q.nu * snU[j] * magS;                                                     // fvc::laplacian(muf/rhof, U), L74
        const double ext = S[0] * gUT[0 * 3 + j] + S[1] * gUT[1 * 3 + j] + S[2] * gUT[2 * 3 + j];    // Sf & lin(T(grad U)), L56 / L76
        q.F[(size_t)j * nF + pos] = (q.implicit ? phiUf - q.nu * ext : (phiUf - lap) - q.nu * ext) + (S[j] * pf) / q.rho0;
    }
    const double phiTf = phi * (q.upwindT ? lerpf(phi >= 0.0 ? 1.0 : 0.0, To, Tn) : Tf);
    q.F[3 * nF + pos] = q.implicit ? phiTf - phitr : (phiTf - q.Hi * snT * magS) - phitr;   // QHDTEqn.H L65-66, L73-76, L85-88
}

// explicit Euler of the U and T equations [QHDUEqn.H L68-84, QHDTEqn.H L83-91]
__global__ __launch_bounds__(QGD_BLOCK) void qhdCellUpdateKernel(const MeshView m, const QhdView q) {
    const int c = xcdTile((int)gridDim.x, m.xcdRun) * QGD_BLOCK + threadIdx.x;   // runs of consecutive blocks per XCD: neighbours meet in one L2
    if (c >= m.nC) return;
    if (m.ghost && m.ghost[c] == 1) return;
    const int n = m.cfCount[c];
    const size_t base = (size_t)m.cfSlice[c >> 6] * 64 + (c & 63);
    const size_t nF = (size_t)m.nF;
    double s[4] = {0, 0, 0, 0};
    if (__ballot(n != 6) == 0) {
        // a wavefront of hexahedra: the 24 face terms in flight before the ordered sums (ascending face label, as below; cfPos keeps cfItem's order)
        int it[6];
        double x[6][4];
#pragma unroll
        for (int i = 0; i < 6; ++i) it[i] = m.cfPos[base + (size_t)i * 64];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const size_t pos = (size_t)(it[i] >= 0 ? it[i] : ~it[i]);
#pragma unroll
            for (int k = 0; k < 4; ++k) x[i][k] = q.F[(size_t)k * nF + pos];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int k = 0; k < 4; ++k) s[k] = it[i] >= 0 ? s[k] + x[i][k] : s[k] - x[i][k];
    } else {
        // any cell shapes: eight faces per pass, positions first, then the 32 terms in flight before the ordered sums
        for (int i0 = 0; i0 < n; i0 += 8) {
            int it[8];
            double x[8][4];
#pragma unroll
            for (int u = 0; u < 8; ++u) it[u] = i0 + u < n ? m.cfPos[base + (size_t)(i0 + u) * 64] : 0;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const bool on = i0 + u < n;
                const size_t pos = (size_t)(it[u] >= 0 ? it[u] : ~it[u]);
#pragma unroll
                for (int k = 0; k < 4; ++k) x[u][k] = on ? q.F[(size_t)k * nF + pos] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (i0 + u >= n) continue;
#pragma unroll
                for (int k = 0; k < 4; ++k) s[k] = it[u] >= 0 ? s[k] + x[u][k] : s[k] - x[u][k];
            }
        }
    }
    const double rV = 1.0 / m.V[c];
    double* rec = q.c4 + (size_t)c * 4;
    const double T = rec[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) rec[k] += q.dt * (-(s[k] * rV) + (q.beta * T) * q.g[k]);   // BdFrc = beta*T*g of updateFields.H L66
    rec[3] = T + q.dt * (-(s[3] * rV));
}

// ---------------------------------------------------------------------------
// QGD_QHD_FUSED: the U and T equations of a BLOCK of cells in one workgroup, on the cell blocks of QGDFoam's one-launch step (qgd_setup.hpp
// FusedBlocks; qgd_kernels.hip fusedFaceCellKernel) -- the vertex values of p, face pass 2 and the explicit Euler update [QHDUEqn.H L36-84,
// QHDTEqn.H L65-91] without the 8 B per vertex and the 2 x 32 B per face the three kernels hand each other through device memory.  A workgroup
// (256 threads) stages {U,T}, fvc::grad(U) and the centres of its own + across-a-face cells, p of those and of the edge / corner cells around
// its vertices, the vertex coordinates; thread v forms p at vertex v (pointInterpFastKernel's sum, out of LDS; a patch point takes
// boundaryPointKernel's value); every thread computes the net terms of two of the block's internal faces (qhdFace2TileKernel's expressions);
// the terms take the gradients' place in LDS and threads 0..127 advance one own cell each (qhdCellUpdateKernel's ordered sums; patch faces
// from q.F, where qhdFace2Kernel put them).  The block reads the OLD {U,T} of its neighbours while other blocks write new ones: the update
// goes to a second record array, the host swaps the two.  Unsharded cases, implicitDiffusion false.
// ---------------------------------------------------------------------------
constexpr int kQhdFuCapC = 320, kQhdFuCapV = 256, kQhdFuCapF = 512, kQhdFuCapTot = 384;   // = kFusedCap{C,V,F,Tot} of qgd_setup.hpp
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3)))
void qhdFusedAdvanceKernel(const MeshView m, const QhdView q, double* __restrict__ c4new, const int ldsCell) {
    extern __shared__ v2d qhdLds[];
    constexpr int NT = 256, KB2 = 3, KCC = 4, KG = 12, KPC = 2, KV = 3, KF = 2, KE = 6, KP = 8;
    static_assert(2 * kQhdFuCapC <= KB2 * NT && 3 * kQhdFuCapC <= KCC * NT && 9 * kQhdFuCapC <= KG * NT && kQhdFuCapTot <= KPC * NT &&
                  3 * kQhdFuCapV <= KV * NT && kQhdFuCapF <= KF * NT && kQhdFuCapV <= NT, "caps");
    const int blk = xcdTile((int)gridDim.x, m.fuXcdRun);
    const int tid = (int)threadIdx.x;
    const int capC = m.fuCapC, capV = m.fuCapV, capF = m.fuCapF, capPE = m.fuCapPE;
    const int32_t* __restrict__ tCells = m.fuCells + (size_t)blk * capC;
    const int32_t* __restrict__ tVerts = m.fuVerts + (size_t)blk * capV;
    const int32_t* __restrict__ tFaceLabel = m.fuFaceLabel + (size_t)blk * capF;
    // (0) the lists: nothing here depends on a loaded value
    const int4 hdr = m.fuHdr[blk];
    const int4 hdr2 = m.fuHdr2[blk];
    int fl[KF];
#pragma unroll
    for (int j = 0; j < KF; ++j) fl[j] = tFaceLabel[min(tid + j * NT, capF - 1)];
    // the cell list goes through LDS: every thread reads two labels, the 21 piece indices below come out of LDS instead of 21 more loads
    // (the CU's address unit is what a block's rounds of loads queue in: profiles/r06_fused_phase_clock.txt)
    int* const sList = reinterpret_cast<int*>(reinterpret_cast<double*>(qhdLds) + ldsCell) + KE * 128;
    int idP[KPC];
#pragma unroll
    for (int k = 0; k < KPC; ++k) idP[k] = tCells[min(tid + k * NT, capC - 1)];
#pragma unroll
    for (int k = 0; k < KPC; ++k) { if (tid + k * NT < capC) sList[tid + k * NT] = idP[k]; }
    __syncthreads();
    int idB[KB2], idC[KCC], idG[KG], idV[KV];
#pragma unroll
    for (int k = 0; k < KB2; ++k) { const int qq = tid + k * NT, r = qq >> 1; idB[k] = sList[min(r, capC - 1)] * 2 + (qq & 1); }
#pragma unroll
    for (int k = 0; k < KCC; ++k) { const int qq = tid + k * NT, r = (qq * 43691) >> 17; idC[k] = sList[min(r, capC - 1)] * 3 + (qq - 3 * r); }
#pragma unroll
    for (int k = 0; k < KG; ++k) { const int qq = tid + k * NT, r = qq / 9; idG[k] = sList[min(r, capC - 1)] * 9 + (qq - 9 * r); }
#pragma unroll
    for (int k = 0; k < KV; ++k) { const int qq = tid + k * NT, r = (qq * 43691) >> 17; idV[k] = tVerts[min(r, capV - 1)] * 3 + (qq - 3 * r); }
    const int ci = sList[min(tid & 127, capC - 1)];
    const int nEraw = (int)m.fuNEntry[(size_t)blk * 128 + (tid & 127)];
    const int vt = min(tid, capV - 1);
    const int myVert = tVerts[vt];
    const int nPc = (int)m.fuVCount[(size_t)blk * capV + vt];
    const double* __restrict__ vW = m.fuVW + (size_t)blk * capPE * capV + vt;
    double pcW[KP];
#pragma unroll
    for (int i = 0; i < KP; ++i) pcW[i] = vW[(size_t)min(i, capPE - 1) * capV];
    // (1) one round trip later: the records piece by piece, the faces' streams, the block's local topology out of its template
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
    v2d d4[KB2];
    double dC[KCC], dG[KG], dP[KPC], dX[KV];
    const v2d* __restrict__ g4 = reinterpret_cast<const v2d*>(q.c4);
#pragma unroll
    for (int k = 0; k < KB2; ++k) d4[k] = g4[idB[k]];
#pragma unroll
    for (int k = 0; k < KCC; ++k) dC[k] = m.Cc[idC[k]];
#pragma unroll
    for (int k = 0; k < KG; ++k) dG[k] = q.gUc[idG[k]];
#pragma unroll
    for (int k = 0; k < KPC; ++k) dP[k] = q.p[idP[k]];
#pragma unroll
    for (int k = 0; k < KV; ++k) dX[k] = m.X[idV[k]];
    const size_t nF = (size_t)m.nF;
    const double Vc = m.V[ci];
    double dPt = 0.0;
    if (nPc == 0) dPt = q.ptp[myVert];   // a patch point: boundaryPointKernel has put its value there
    __builtin_amdgcn_sched_barrier(0);
    const int nOwn = hdr.x, nUc = hdr.y, nUv = hdr.z, nFc = hdr.w, nTot = hdr2.x;
    // LDS by THIS block's counts: {U,T}, fvc::grad(U) (the four planes of net terms take their place once every face is done), centres, p, vertex
    // coordinates, vertex p; the own cells' parked face entries at a fixed place behind
    const int nG = max(9 * nUc, 4 * nFc);
    v2d* const s4 = qhdLds;                                        // 2 nUc pieces
    double* const sG = reinterpret_cast<double*>(s4 + 2 * nUc);    // 9 nUc
    double* const sC = sG + nG;                                    // 3 nUc
    double* const sp = sC + 3 * nUc;                               // nTot
    double* const sX = sp + nTot;                                  // 3 nUv
    double* const sq = sX + 3 * nUv;                               // nUv
    double* const sF = sG;
    int* const sE = reinterpret_cast<int*>(reinterpret_cast<double*>(qhdLds) + ldsCell);
    const int strideF = nFc;
    if (tid < 128) {
#pragma unroll
        for (int i = 0; i < KE; ++i) sE[i * 128 + tid] = e6[i];
    }
#pragma unroll
    for (int k = 0; k < KB2; ++k) { const int qq = tid + k * NT; if (qq < 2 * nUc) s4[qq] = d4[k]; }
#pragma unroll
    for (int k = 0; k < KCC; ++k) { const int qq = tid + k * NT; if (qq < 3 * nUc) sC[qq] = dC[k]; }
#pragma unroll
    for (int k = 0; k < KG; ++k) { const int qq = tid + k * NT; if (qq < 9 * nUc) sG[qq] = dG[k]; }
#pragma unroll
    for (int k = 0; k < KPC; ++k) { const int qq = tid + k * NT; if (qq < nTot) sp[qq] = dP[k]; }
#pragma unroll
    for (int k = 0; k < KV; ++k) { const int qq = tid + k * NT; if (qq < 3 * nUv) sX[qq] = dX[k]; }
    __builtin_amdgcn_sched_barrier(0);
    // the first face's streams are asked for now that the staging registers are free, and arrive behind the vertex values; the second face's
    // in the middle of the first face (two faces' streams next to one face's algebra do not fit three waves per SIMD)
    struct FaceStreams { double w, dn, magS, S[3], tau, phi, phitr, ugu[3]; int kind; };
    auto loadStreams = [&](FaceStreams& t, const int f) {
        t.w = ldStream(m.w + f); t.dn = ldStream(m.dn + f); t.magS = ldStream(m.magSf + f);
        t.S[0] = ldStream(m.Sx + f); t.S[1] = ldStream(m.Sy + f); t.S[2] = ldStream(m.Sz + f);
        t.tau = ldStream(q.tauF + f); t.phi = q.phi[f]; t.phitr = q.phitr[f];
        t.ugu[0] = q.ugu[f]; t.ugu[1] = q.ugu[nF + f]; t.ugu[2] = q.ugu[2 * nF + f];
        t.kind = m.fkind[f];
    };
    FaceStreams st[KF];
    loadStreams(st[0], fl[0]);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    // (1b) p at the vertices [volPointInterpolation: pointInterpFastKernel's sum in pointCells order, out of the staged cells]
    if (tid < nUv) {
        double acc = 0.0;
        if (__ballot(nPc != KP) == 0) {
#pragma unroll
            for (int i = 0; i < KP; ++i) acc += pcW[i] * sp[pcPos[i]];
        } else if (nPc > 0) {
            for (int i = 0; i < nPc; ++i) {
                int pos = 0;
                double w = 0.0;
                if (i < KP) {
#pragma unroll
                    for (int u = 0; u < KP; ++u) { pos = (i == u) ? pcPos[u] : pos; w = (i == u) ? pcW[u] : w; }
                } else {
                    pos = (int)vPos[(size_t)i * capV];
                    w = vW[(size_t)i * capV];
                }
                acc += w * sp[pos];
            }
        } else acc = dPt;
        sq[tid] = acc;
    }
    __syncthreads();
    // (2) the faces: qhdFace2TileKernel's expressions, the net terms into registers
    auto l3 = [](const double* p, int i) { return make_double4(p[3 * i], p[3 * i + 1], p[3 * i + 2], 0.0); };
    auto l4 = [](const v2d* p, int i, double* o) { const v2d a = p[2 * i], b = p[2 * i + 1]; o[0] = a.x; o[1] = a.y; o[2] = b.x; o[3] = b.y; };
    double out[KF][4];
#pragma unroll
    for (int j = 0; j < KF; ++j) {
        const int lf = tid + j * NT;
        if (lf < nFc) {
            const int f = fl[j], kind = st[j].kind;
            const int lo = (int)(fp[j].c & 0xffffu), ln = (int)(fp[j].c >> 16);
            const int v0 = (int)(fp[j].va & 0xffffu), v1 = (int)(fp[j].va >> 16), v2 = (int)(fp[j].vb & 0xffffu), v3 = (int)(fp[j].vb >> 16);
            if (kind == 3) {
                out[j][0] = out[j][1] = out[j][2] = out[j][3] = 0.0;
                if (j + 1 < KF) loadStreams(st[j + 1], fl[j + 1]);
            } else {
                const double w = st[j].w, dn = st[j].dn, magS = st[j].magS, tau = st[j].tau, phi = st[j].phi, phitr = st[j].phitr;
                const double S[3] = {st[j].S[0], st[j].S[1], st[j].S[2]};
                double o4[4], n4[4];
                l4(s4, lo, o4); l4(s4, ln, n4);
                const double Uo[3] = {o4[0], o4[1], o4[2]}, Un[3] = {n4[0], n4[1], n4[2]}, To = o4[3], Tn = n4[3];
                FaceVals<1> vp;
                vp.o[0] = sp[lo]; vp.n[0] = sp[ln]; vp.sn[0] = 0.0;
                double snU[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) snU[k] = dn * (Un[k] - Uo[k]);   // fvc::snGrad, uncorrected (L0)
                const double snT = dn * (Tn - To);
                const double pf = lerpf(w, vp.o[0], vp.n[0]);
                double gUT[9];   // T(grad U) at the face: gUT[3i+j] = lin(gradU)[3j+i]
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int jj = 0; jj < 3; ++jj) gUT[3 * i + jj] = lerpf(w, sG[9 * lo + 3 * jj + i], sG[9 * ln + 3 * jj + i]);
                double gP[3];
                if (kind == 2) faceGradient<ST_GVP3, 1, -1>(m, f, vp, q.p, q.ptp, gP);
                else gvp3GradCore<1, -1>(kind, true, l3(sC, lo), l3(sC, ln), l3(sX, v0), l3(sX, v1), l3(sX, v2), l3(sX, v3), vp.o, vp.n, sq + v0, sq + v1,
                                         sq + v2, sq + v3, gP);                                      // QHDUEqn.H L36
                if (j + 1 < KF) {
                    __builtin_amdgcn_sched_barrier(0);
                    loadStreams(st[j + 1], fl[j + 1]);
                    __builtin_amdgcn_sched_barrier(0);
                }
                double Uf[3], Wf[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    Uf[k] = lerpf(w, Uo[k], Un[k]);
                    const double Bf = lerpf(w, (q.beta * To) * q.g[k], (q.beta * Tn) * q.g[k]);   // BdFrcf [updateFields.H L66-67], as in face pass 1
                    Wf[k] = tau * ((st[j].ugu[k] + gP[k] / q.rho0) - Bf);   // L37
                }
                const double Tf = lerpf(w, To, Tn);
#pragma unroll
                for (int jj = 0; jj < 3; ++jj) {
                    const double uw = S[0] * (Uf[0] * Wf[jj]) + S[1] * (Uf[1] * Wf[jj]) + S[2] * (Uf[2] * Wf[jj]);   // Sf & (Uf*Wf), L39
                    const double phiUf = phi * (q.upwindU ? lerpf(phi >= 0.0 ? 1.0 : 0.0, Uo[jj], Un[jj]) : Uf[jj]) - uw;   // L41-43
                    const double lap = q.nu * snU[jj] * magS;                                                      // fvc::laplacian(muf/rhof, U), L74
                    const double ext = S[0] * gUT[0 * 3 + jj] + S[1] * gUT[1 * 3 + jj] + S[2] * gUT[2 * 3 + jj];   // Sf & lin(T(grad U)), L56 / L76
                    out[j][jj] = ((phiUf - lap) - q.nu * ext) + (S[jj] * pf) / q.rho0;
                }
                const double phiTf = phi * (q.upwindT ? lerpf(phi >= 0.0 ? 1.0 : 0.0, To, Tn) : Tf);
                out[j][3] = (phiTf - q.Hi * snT * magS) - phitr;   // QHDTEqn.H L65-66, L85-88
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();   // every face has read its gradients: the planes of net terms take their place
#pragma unroll
    for (int j = 0; j < KF; ++j) {
        const int lf = tid + j * NT;
        if (lf < nFc) {
#pragma unroll
            for (int k = 0; k < 4; ++k) sF[k * strideF + lf] = out[j][k];
        }
    }
    __syncthreads();
    // (3) the block's own cells [QHDUEqn.H L68-84, QHDTEqn.H L83-91]: qhdCellUpdateKernel's ordered sums (ascending face label) and update
    if (tid < nOwn) {
        double s[4] = {0, 0, 0, 0};
        const int nE = nEraw;
        int eq[KE];
#pragma unroll
        for (int i = 0; i < KE; ++i) eq[i] = sE[i * 128 + tid];
        bool inner = nE == KE;
#pragma unroll
        for (int i = 0; i < KE; ++i) inner = inner && eq[i] >= 0;
        if (__ballot(!inner) == 0) {
            double x[KE][4];
#pragma unroll
            for (int i = 0; i < KE; ++i)
#pragma unroll
                for (int k = 0; k < 4; ++k) x[i][k] = sF[k * strideF + (eq[i] >> 1)];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < KE; ++i)
#pragma unroll
                for (int k = 0; k < 4; ++k) s[k] = (eq[i] & 1) ? s[k] - x[i][k] : s[k] + x[i][k];
        } else {
            for (int i = 0; i < nE; ++i) {
                const int e = (i < KE) ? sE[i * 128 + tid] : ent[(size_t)i * 128];
                double x[4];
                if (e >= 0) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) x[k] = sF[k * strideF + (e >> 1)];
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) x[k] = q.F[(size_t)k * nF + (size_t)(~e)];   // a patch face: the owner's side, plus
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) s[k] = (e < 0 || !(e & 1)) ? s[k] + x[k] : s[k] - x[k];
            }
        }
        const double rV = 1.0 / Vc;
        double rec[4];
        l4(s4, tid, rec);
        const double T = rec[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) rec[k] += q.dt * (-(s[k] * rV) + (q.beta * T) * q.g[k]);   // BdFrc = beta*T*g of updateFields.H L66
        rec[3] = T + q.dt * (-(s[3] * rV));
        v2d* const dst = reinterpret_cast<v2d*>(c4new) + 2 * (size_t)ci;
        dst[0] = v2d{rec[0], rec[1]};
        dst[1] = v2d{rec[2], rec[3]};
    }
}

// ---- implicitDiffusion [QHDUEqn.H L46-65, QHDTEqn.H L69-80] ------------------------------------------------------------------------
// patch coefficients of -fvm::laplacian(gamma, x) on owner-side boundary face f (L0): fixedValue: internal delta, source delta*value;
// basicSymmetry (slip, U only): internal delta*|n_k|, source snGrad_k + delta*|n_k|*patchInternalField_k (transformFvPatchField);
// zeroGradient: none.  ic[k] / bs[k] for k = Ux, Uy, Uz, T, in units of delta_f (the caller multiplies by gamma_k |Sf|); cur = the
// owner's current {U, T} (nullptr: coefficients only)
__device__ __forceinline__ void qhdPatchCoeffs(const MeshView& m, const PatchBCDev& bc, const int f, const double* cur, double ic[4], double bs[4]) {
#pragma unroll
    for (int k = 0; k < 4; ++k) ic[k] = bs[k] = 0.0;
    if (bc.ptype == QGD_PATCH_HALO || bc.ptype == QGD_PATCH_CYCLIC) return;
    const double dc = m.dn[f];
    if (bc.bcU == QGD_BC_FIXEDVALUE) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { ic[k] = dc; bs[k] = dc * bc.vU[k]; }
    } else if (bc.bcU == QGD_BC_SLIP) {
        double n[3];
        symmNormal(m, bc, f, n);
#pragma unroll
        for (int k = 0; k < 3; ++k) ic[k] = dc * fabs(n[k]);
        if (cur) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const double tv = ((i == 0 ? 1.0 : 0.0) - 2.0 * (n[i] * n[0])) * cur[0] + ((i == 1 ? 1.0 : 0.0) - 2.0 * (n[i] * n[1])) * cur[1] +
                                  ((i == 2 ? 1.0 : 0.0) - 2.0 * (n[i] * n[2])) * cur[2];
                bs[i] = (tv - cur[i]) * (dc / 2.0) + ic[i] * cur[i];   // basicSymmetry::snGrad + gIC * patchInternalField
            }
        }
    }
    if (bc.bcT == QGD_BC_FIXEDVALUE) { ic[3] = dc; bs[3] = dc * bc.vT; }
}
// the shared face coefficient |Sf| delta_f at the face's slot-major position (what the matrix products gather)
__global__ __launch_bounds__(QGD_BLOCK) void qhdImplicitFaceCoefKernel(const MeshView m, const QhdView q) {
    const int f = blockIdx.x * QGD_BLOCK + threadIdx.x;
    if (f >= m.nF) return;
    const size_t pos = f < m.nIF ? (size_t)m.fpos[f] : (size_t)f;
    q.aG[pos] = m.fkind[f] == 3 ? 0.0 : m.magSf[f] * m.dn[f];   // nonOrthDeltaCoeffs inside, deltaCoeffs on patches
}
// the diagonals: V/deltaT + gamma_k (sum of the internal faces' coefficients + the patch internal coefficients)
__global__ __launch_bounds__(QGD_BLOCK) void qhdImplicitDiagKernel(const MeshView m, const QhdView q, const PatchBCDev* __restrict__ bcs) {
    const int c = blockIdx.x * QGD_BLOCK + threadIdx.x;
    if (c >= m.nC) return;
    const size_t nC = (size_t)m.nC;
    if (m.ghost && m.ghost[c] == 1) { for (int k = 0; k < 4; ++k) q.diag4[(size_t)k * nC + c] = 1.0; return; }   // never a row
    const int n = m.cfCount[c];
    const size_t base = (size_t)m.cfSlice[c >> 6] * 64 + (c & 63);
    double inner = 0.0, pc[4] = {0, 0, 0, 0};
    for (int i = 0; i < n; ++i) {
        const int it = m.cfItem[base + (size_t)i * 64], ps = m.cfPos[base + (size_t)i * 64];
        const int f = it >= 0 ? it : ~it;
        const double g = q.aG[ps >= 0 ? ps : ~ps];
        if (f < m.nIF) { inner += g; continue; }
        if (m.fkind[f] == 3) continue;
        double ic[4], bs[4];
        qhdPatchCoeffs(m, bcs[m.bPatch[f - m.nIF]], f, nullptr, ic, bs);
        const double ms = m.magSf[f];
        for (int k = 0; k < 4; ++k) pc[k] += ms * ic[k];
    }
    const double rDV = (1.0 / q.dt) * m.V[c];
    for (int k = 0; k < 4; ++k) {
        const double gam = k < 3 ? q.nu : q.Hi;
        q.diag4[(size_t)k * nC + c] = (rDV + gam * inner) + gam * pc[k];
    }
}
// right-hand sides and start values of the four systems: fvm::ddt's source + V (-div(face terms) + BdFrc) + the patch sources
__global__ __launch_bounds__(QGD_BLOCK) void qhdImplicitRhsKernel(const MeshView m, const QhdView q, const PatchBCDev* __restrict__ bcs) {
    const int c = xcdTile((int)gridDim.x, m.xcdRun) * QGD_BLOCK + threadIdx.x;
    if (c >= m.nC) return;
    const size_t nC = (size_t)m.nC, nF = (size_t)m.nF;
    double cur[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { cur[k] = q.c4[(size_t)c * 4 + k]; q.x4[(size_t)k * nC + c] = cur[k]; }   // ghost columns start from the state message
    // start values: the current fields (OpenFOAM's) + their time increments of the steps before extrapolated (QGD_IMPL_XEXTRAP, see the "start
    // values" note in qgd_implicit.hip).  The history is kept here, for EVERY cell: a ghost cell's record is its owner's, bit for bit (state
    // message), so both ranks form the same start value and the first product needs no extra message.  xd[0..] = the fields one, two, ...
    // steps back (the oldest doubles as the slot the current fields go into; the host rotates the pointers after the launch)
    if (q.xOrder > 0) {
        const int kk = q.xHave < q.xOrder ? q.xHave : q.xOrder;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const size_t j = (size_t)k * nC + c;
            double v[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) v[t] = t < q.xOrder ? q.xd[t][j] : 0.0;
            if (kk >= 1) {
                const double d0 = cur[k] - v[0];
                double e = d0;
                if (kk == 2) e = (2.0 * cur[k] - 3.0 * v[0]) + v[1];
                else if (kk == 3) e = ((3.0 * cur[k] - 6.0 * v[0]) + 4.0 * v[1]) - v[2];
                else if (kk >= 4) e = (((4.0 * cur[k] - 10.0 * v[0]) + 10.0 * v[1]) - 5.0 * v[2]) + v[3];
                const double lim = 2.0 * fabs(d0);     // limited like the pressure's (qhdExtrapolatePKernel)
                q.x4[j] = cur[k] + fmin(fmax(e, -lim), lim);
            }
            q.xd[q.xOrder - 1][j] = cur[k];            // the current fields into the oldest slot (the host rotates the pointers)
        }
    }
    if (m.ghost && m.ghost[c] == 1) return;
    const int n = m.cfCount[c];
    const size_t base = (size_t)m.cfSlice[c >> 6] * 64 + (c & 63);
    double s[4] = {0, 0, 0, 0}, src[4] = {0, 0, 0, 0};
    for (int i0 = 0; i0 < n; i0 += 8) {   // eight faces per pass: positions, then the 32 terms in flight before the ordered sums (ascending face label)
        int it[8], ps[8];
        double x[8][4];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const bool on = i0 + u < n;
            it[u] = on ? m.cfItem[base + (size_t)(i0 + u) * 64] : 0;
            ps[u] = on ? m.cfPos[base + (size_t)(i0 + u) * 64] : 0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const bool on = i0 + u < n;
            const size_t pos = (size_t)(ps[u] >= 0 ? ps[u] : ~ps[u]);
#pragma unroll
            for (int k = 0; k < 4; ++k) x[u][k] = on ? q.F[(size_t)k * nF + pos] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (i0 + u >= n) continue;
#pragma unroll
            for (int k = 0; k < 4; ++k) s[k] = ps[u] >= 0 ? s[k] + x[u][k] : s[k] - x[u][k];
            const int f = it[u] >= 0 ? it[u] : ~it[u];
            if (f >= m.nIF && m.fkind[f] != 3) {
                double ic[4], bs[4];
                qhdPatchCoeffs(m, bcs[m.bPatch[f - m.nIF]], f, cur, ic, bs);
                const double ms = m.magSf[f];
                for (int k = 0; k < 4; ++k) src[k] += ((k < 3 ? q.nu : q.Hi) * ms) * bs[k];
            }
        }
    }
    const double V = m.V[c], rV = 1.0 / V, rD = 1.0 / q.dt;
#pragma unroll
    for (int k = 0; k < 3; ++k)
        q.rhs4[(size_t)k * nC + c] = (rD * cur[k] * V + V * (-(s[k] * rV) + (q.beta * cur[3]) * q.g[k])) + src[k];   // BdFrc = beta*T*g [updateFields.H L66]
    q.rhs4[3 * nC + c] = (rD * cur[3] * V + V * (-(s[3] * rV))) + src[3];
}
// the solution into the records (components that were not solved keep their values)
__global__ __launch_bounds__(QGD_BLOCK) void qhdImplicitStoreKernel(const MeshView m, const QhdView q, const int validMask) {
    const int c = blockIdx.x * QGD_BLOCK + threadIdx.x;
    if (c >= m.nC) return;
    if (m.ghost && m.ghost[c] == 1) return;
    const size_t nC = (size_t)m.nC;
#pragma unroll
    for (int k = 0; k < 4; ++k) if ((validMask >> k) & 1) q.c4[(size_t)c * 4 + k] = q.x4[(size_t)k * nC + c];
}

// correctBoundaryConditions of U and T after their solves
__global__ __launch_bounds__(QGD_BLOCK) void qhdBcKernel(const MeshView m, const QhdView q, const PatchBCDev* __restrict__ bcs) {
    const int b = blockIdx.x * QGD_BLOCK + threadIdx.x;
    if (b >= m.nBF) return;
    const int f = m.nIF + b;
    if (m.fkind[f] == 3) return;
    const PatchBCDev bc = bcs[m.bPatch[b]];
    if (bc.ptype == QGD_PATCH_HALO) return;
    const double* o4 = q.c4 + (size_t)m.own[f] * 4;
    double* b4 = q.b4 + (size_t)b * 4;
    if (bc.bcU == QGD_BC_FIXEDVALUE) { b4[0] = bc.vU[0]; b4[1] = bc.vU[1]; b4[2] = bc.vU[2]; }
    else if (bc.bcU == QGD_BC_SLIP) {
        double n[3];
        symmNormal(m, bc, f, n);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const double tv = ((i == 0 ? 1.0 : 0.0) - 2.0 * (n[i] * n[0])) * o4[0] + ((i == 1 ? 1.0 : 0.0) - 2.0 * (n[i] * n[1])) * o4[1] +
                              ((i == 2 ? 1.0 : 0.0) - 2.0 * (n[i] * n[2])) * o4[2];
            b4[i] = (o4[i] + tv) / 2.0;
        }
    } else { b4[0] = o4[0]; b4[1] = o4[1]; b4[2] = o4[2]; }
    b4[3] = bc.bcT == QGD_BC_FIXEDVALUE ? bc.vT : o4[3];
}

// p += pRefValue - p[pRefCell] when the field needs a reference level [QHDFoam.C L123-130]
__global__ void qhdRefReadKernel(const QhdView q, const int refCell, const double refValue, double* __restrict__ shift) {
    shift[0] = refCell >= 0 ? refValue - q.p[refCell] : 0.0;   // refCell < 0: another shard owns it (the shifts are summed over the ranks)
}
// halo messages of a sharded QHD case.  kind 0: the state after a step, {Ux,Uy,Uz,T} per cell and per patch face (4 + 4);
// kind 1: after the pressure solve, p + fvc::grad(U) per cell (1 + 9: the gradient was formed at the start of the step, when the
// ghost cells held the new velocity), p's patch value and gradient per patch face (2); kind 2: the search direction of the
// pressure solve (1 per cell)
__global__ __launch_bounds__(QGD_BLOCK) void qhdHaloKernel(const QhdView q, double* __restrict__ dirn, const int kind, const int32_t* __restrict__ cells,
                                                          const int nCells, const int32_t* __restrict__ bfaces, const int nFaces,
                                                          double* __restrict__ buf, const int pack) {
    const int i = blockIdx.x * QGD_BLOCK + threadIdx.x;
    if (i < nCells) {
        const size_t c = (size_t)cells[i];
        if (kind == 0) {
            double* b = buf + (size_t)i * 4;
#pragma unroll
            for (int k = 0; k < 4; ++k) { if (pack) b[k] = q.c4[c * 4 + k]; else q.c4[c * 4 + k] = b[k]; }
        } else if (kind == 1) {
            double* b = buf + (size_t)i * 10;
            if (pack) b[0] = q.p[c]; else q.p[c] = b[0];
#pragma unroll
            for (int k = 0; k < 9; ++k) { if (pack) b[1 + k] = q.gUc[c * 9 + k]; else q.gUc[c * 9 + k] = b[1 + k]; }
        } else { if (pack) buf[i] = dirn[c]; else dirn[c] = buf[i]; }
    } else if (i < nCells + nFaces && kind != 2) {
        const int j = i - nCells;
        const size_t f = (size_t)bfaces[j];
        if (kind == 0) {
            double* b = buf + (size_t)nCells * 4 + (size_t)j * 4;
#pragma unroll
            for (int k = 0; k < 4; ++k) { if (pack) b[k] = q.b4[f * 4 + k]; else q.b4[f * 4 + k] = b[k]; }
        } else {
            double* b = buf + (size_t)nCells * 10 + (size_t)j * 2;
            if (pack) { b[0] = q.pb[f]; b[1] = q.pgb[f]; } else { q.pb[f] = b[0]; q.pgb[f] = b[1]; }
        }
    }
}

__global__ __launch_bounds__(QGD_BLOCK) void qhdRefShiftKernel(const int nC, const int nBF, const QhdView q, const double* __restrict__ shift) {
    const int i = blockIdx.x * QGD_BLOCK + threadIdx.x;
    if (i < nC) q.p[i] += shift[0];
    else if (i < nC + nBF) q.pb[i - nC] += shift[0];
}

// createFields: cell records from U, T (host order) and p
__global__ __launch_bounds__(QGD_BLOCK) void qhdInitKernel(const int nC, const QhdView q, const double* __restrict__ U, const double* __restrict__ T,
                                                          const double* __restrict__ p) {
    const int c = blockIdx.x * QGD_BLOCK + threadIdx.x;
    if (c >= nC) return;
    q.c4[(size_t)c * 4] = U[3 * (size_t)c]; q.c4[(size_t)c * 4 + 1] = U[3 * (size_t)c + 1]; q.c4[(size_t)c * 4 + 2] = U[3 * (size_t)c + 2];
    q.c4[(size_t)c * 4 + 3] = T[c];
    q.p[c] = p[c];
}
// tauQGDf = linearInterpolate(tauQGD) of the QHD closures [constTau_8C L71-74, HbyUQHD_8C L80-83, T0byGr_8C L84-87,
// H2bynuQHD_8C L78-82]; taubyrhof = tauQGDf/rhof [updateFluxes.H L38]
__device__ __forceinline__ double qhdTauOf(const QhdView& q, const double h) {
    switch (q.tauModel) {
        case 0: return q.Tau;
        case 1: return q.aQGD * h / q.UQHD;
        case 2: return q.T0 / q.Gr;
        default: return q.aQGD * h * h / q.nu;
    }
}
__global__ __launch_bounds__(QGD_BLOCK) void qhdTauKernel(const MeshView m, const QhdView q, double* __restrict__ tauF, double* __restrict__ tbr) {
    const int f = blockIdx.x * QGD_BLOCK + threadIdx.x;
    if (f >= m.nF) return;
    double t = 0.0;
    if (m.fkind[f] != 3) {
        if (f < m.nIF) t = lerpf(m.w[f], qhdTauOf(q, m.hQGD[m.own[f]]), qhdTauOf(q, m.hQGD[m.nei[f]]));
        else t = qhdTauOf(q, m.hQGDb[f - m.nIF]);
    }
    tauF[f] = t;
    tbr[f] = t / q.rho0;
}
// named field out of the records
__global__ __launch_bounds__(QGD_BLOCK) void qhdExtractKernel(const int64_t n, const double* __restrict__ rec4, const int field, double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * QGD_BLOCK + threadIdx.x;
    if (i >= n) return;
    if (field == 0) { out[3 * i] = rec4[4 * i]; out[3 * i + 1] = rec4[4 * i + 1]; out[3 * i + 2] = rec4[4 * i + 2]; }
    else out[i] = rec4[4 * i + 3];
}

inline int gridOf(int64_t n) { return (int)((n + QGD_BLOCK - 1) / QGD_BLOCK); }

// 3-D GaussVolPoint with face tiles of 128 (MeshView::qhdTiles): the staged kernels take the internal faces, the generic ones the
// tiles beyond the caps and the boundary faces
inline bool staged(int st, const MeshView& m) { return st == ST_GVP3 && m.qhdTiles != 0 && m.tileOff != nullptr && m.fblock == kQhdFB; }
template <int ST>
void face1(hipStream_t s, const MeshView& m, const QhdView& q, const PatchBCDev* bc) {
    if (staged(ST, m)) {
        const size_t lds = ((size_t)m.tileMaxC * 56 + (size_t)m.tileMaxV * 56 + 255) / 256 * 256;
        qhdFace1TileKernel<<<(m.nIF + kQhdFB - 1) / kQhdFB, kQhdFB, lds, s>>>(m, q);
        if (m.nTileSpill > 0) qhdFace1Kernel<ST><<<m.nTileSpill, 128, 0, s>>>(m, q, bc, m.tileSpill, 0);
        if (m.nBF > 0) qhdFace1Kernel<ST><<<gridOf(m.nBF), QGD_BLOCK, 0, s>>>(m, q, bc, nullptr, m.nIF);
    } else qhdFace1Kernel<ST><<<gridOf(m.nF), QGD_BLOCK, 0, s>>>(m, q, bc, nullptr, 0);
}
template <int ST>
void face2(hipStream_t s, const MeshView& m, const QhdView& q, const PatchBCDev* bc) {
    if (staged(ST, m)) {
        const size_t lds = ((size_t)m.tileMaxC * 136 + (size_t)m.tileMaxV * 32 + 255) / 256 * 256;
        qhdFace2TileKernel<<<(m.nIF + kQhdFB - 1) / kQhdFB, kQhdFB, lds, s>>>(m, q);
        if (m.nTileSpill > 0) qhdFace2Kernel<ST><<<m.nTileSpill, 128, 0, s>>>(m, q, bc, m.tileSpill, 0);
        if (m.nBF > 0) qhdFace2Kernel<ST><<<gridOf(m.nBF), QGD_BLOCK, 0, s>>>(m, q, bc, nullptr, m.nIF);
    } else qhdFace2Kernel<ST><<<gridOf(m.nF), QGD_BLOCK, 0, s>>>(m, q, bc, nullptr, 0);
}

}  // namespace

void launchQhdInit(hipStream_t s, const MeshView& m, const QhdView& q, const PatchBCDev* bc, const double* U, const double* T, const double* p,
                   double* tauF, double* taubyrho) {
    qhdInitKernel<<<gridOf(m.nC), QGD_BLOCK, 0, s>>>(m.nC, q, U, T, p);
    qhdTauKernel<<<gridOf(m.nF), QGD_BLOCK, 0, s>>>(m, q, tauF, taubyrho);
    if (m.nBF) {
        qhdBcKernel<<<gridOf(m.nBF), QGD_BLOCK, 0, s>>>(m, q, bc);
        qhdPressureBcKernel<<<gridOf(m.nBF), QGD_BLOCK, 0, s>>>(m, q, bc);
    }
}
// everything of the step before the pressure equation
void launchQhdAssemble(hipStream_t s, int stencil, bool usesPoints, const MeshView& m, const QhdView& q, const PatchBCDev* bc) {
    if (usesPoints) {
        pointInterpFastKernel<4><<<gridOf(m.nP), QGD_BLOCK, 0, s>>>(m, q.c4, q.pt4);
        if (m.nBP) boundaryPointKernel<4><<<gridOf(m.nBP), QGD_BLOCK, 0, s>>>(m, q.b4, 4, q.pt4, 4, 0, m.nGeomD == 3 ? 0 : -1);
    }
    switch (stencil) {
        case ST_REDUCED: face1<ST_REDUCED>(s, m, q, bc); break;
        case ST_LSQ: face1<ST_LSQ>(s, m, q, bc); break;
        case ST_GVP3: face1<ST_GVP3>(s, m, q, bc); break;
        default: face1<ST_GVP2>(s, m, q, bc); break;
    }
    if (m.nBF) qhdPressureBcKernel<<<gridOf(m.nBF), QGD_BLOCK, 0, s>>>(m, q, bc);   // p.correctBoundaryConditions() [QHDpEqn.H L35]
    // fvc::grad(U) of QHDUEqn.H L76, formed here: U does not change before the U equation, and a shard's ghost cells hold the new
    // velocity now (their own gradient travels with the pressure message, after the solve)
    qhdCellGradKernel<<<gridOf(m.nC), QGD_BLOCK, 0, s>>>(m, q);
}
// The start value of the pressure solve [QHDpEqn.H L45]: OpenFOAM starts from the field as it stands, p^n (QGD_QHD_PEXTRAP=0).  The default
// starts from the polynomial extrapolation in time through the last k + 1 fields, sum_j (-1)^j C(k+1, j+1) p^(n-j) (k = QGD_QHD_PEXTRAP:
// 1 linear 2 p^n - p^(n-1), 2 quadratic, 3 cubic, 4 quartic): the same system, the same tolerance, a smaller first residual on a flow that
// evolves smoothly -- 6 -> 5 / 3-4 / 2 conjugate-gradient iterations per step on the 8 M-cell cavity for k = 1 / 2 / 3, 9 -> 7 / 5 / 3 on the
// 16 M-cell irregular mesh (profiles/r05_ab_qhd_pressure_start_value*.txt).  A solver-internal choice: the answer is the same to the
// solve's tolerance, the "Initial residual" of the log is not.  The extrapolated increment is LIMITED to twice the last one per cell
// (|p_start - p^n| <= 2 |p^n - p^(n-1)|): where the history is not smooth (a turning point, a change of a boundary value) the start
// value falls back towards p^n, so it is never further from the new solution than three of the last increments -- at worst an iteration
// more than OpenFOAM's start, never a different answer.  h[0..3]: the fields one to four steps back, rotated here (have = how many are valid).
struct TimeHistory { double* h[4]; int have; int order; };
__global__ __launch_bounds__(QGD_BLOCK) void qhdExtrapolatePKernel(const int n, double* __restrict__ p, const TimeHistory H) {
    const int i = blockIdx.x * QGD_BLOCK + threadIdx.x;
    if (i >= n) return;
    const double pn = p[i];
    double v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = (j < H.order && H.h[j]) ? H.h[j][i] : 0.0;
    const int k = H.have < H.order ? H.have : H.order;
    if (k >= 1) {
        const double d0 = pn - v[0];
        double e = d0;                                                                             // 2 p^n - p^(n-1)
        if (k == 2) e = (2.0 * pn - 3.0 * v[0]) + v[1];                                            // 3, -3, 1
        else if (k == 3) e = ((3.0 * pn - 6.0 * v[0]) + 4.0 * v[1]) - v[2];                        // 4, -6, 4, -1
        else if (k >= 4) e = (((4.0 * pn - 10.0 * v[0]) + 10.0 * v[1]) - 5.0 * v[2]) + v[3];       // 5, -10, 10, -5, 1
        const double lim = 2.0 * fabs(d0);
        e = fmin(fmax(e, -lim), lim);
        p[i] = pn + e;
    }
    // the ring: h[j] <- h[j-1], h[0] <- p^n (each lane moves its own entries; no other lane reads them)
#pragma unroll
    for (int j = 3; j >= 1; --j) if (j < H.order && H.h[j]) H.h[j][i] = v[j - 1];
    if (H.order >= 1 && H.h[0]) H.h[0][i] = pn;
}
void launchQhdExtrapolateP(hipStream_t s, int nC, double* p, double* const hist[4], int have, int order) {
    if (nC <= 0 || order <= 0) return;
    TimeHistory H;
    for (int j = 0; j < 4; ++j) H.h[j] = hist[j];
    H.have = have; H.order = order;
    qhdExtrapolatePKernel<<<gridOf(nC), QGD_BLOCK, 0, s>>>(nC, p, H);
}
// after the solve: solve() ends in correctBoundaryConditions() (a shard then sends p and these patch values to its neighbours)
void launchQhdPostSolve(hipStream_t s, const MeshView& m, const QhdView& q, const PatchBCDev* bc) {
    if (m.nBF) qhdPressureBcKernel<<<gridOf(m.nBF), QGD_BLOCK, 0, s>>>(m, q, bc);
}
// the U and T equations (phi must be there: pressureSolveFlux).  shift (device, 1 double) receives this rank's share of the
// reference-level shift of p [QHDFoam.C L123-130]: pRefValue - p[refCell] on the rank that owns refCell (localRefCell >= 0), else 0
// c4new != nullptr (qhdFusedAdvanceEligible): vertex values of p, internal faces and the cell update are ONE launch on the cell blocks; the new
// {U,T} land in c4new and the caller swaps it with q.c4
bool qhdFusedAdvanceEligible(int stencil, const MeshView& m, const QhdView& q, int* ldsBytes, int* ldsCell) {
    if (stencil != ST_GVP3 || m.fuBlocks <= 0 || m.fuHdr == nullptr || m.fuHdr2 == nullptr || m.ghost != nullptr || q.implicit || m.nGeomD != 3) return false;
    if (m.fuMaxAll > kQhdFuCapC || m.fuMaxTot > kQhdFuCapTot || m.fuMaxV > kQhdFuCapV || m.fuMaxF > kQhdFuCapF) return false;
    const int cell = 7 * m.fuMaxAll + std::max(9 * m.fuMaxAll, 4 * m.fuMaxF) + m.fuMaxTot + 4 * m.fuMaxV;
    const int lds = 8 * cell + 6 * 128 * 4 + 4 * m.fuCapC;   // records, the own cells' parked face entries, the cell list
    if (lds > 65536) return false;
    if (ldsBytes) *ldsBytes = lds;
    if (ldsCell) *ldsCell = cell;
    return true;
}
bool launchQhdAdvance(hipStream_t s, int stencil, bool usesPoints, const MeshView& m, const QhdView& q, const PatchBCDev* bc, bool needRef,
                      int localRefCell, double refValue, double* shift, double* c4new) {
    int lds = 0, ldsCell = 0;
    if (c4new && qhdFusedAdvanceEligible(stencil, m, q, &lds, &ldsCell)) {
        if (m.nBP) boundaryPointKernel<1><<<gridOf(m.nBP), QGD_BLOCK, 0, s>>>(m, q.pb, 1, q.ptp, 1, 0, -1);
        if (m.nBF > 0) qhdFace2Kernel<ST_GVP3><<<gridOf(m.nBF), QGD_BLOCK, 0, s>>>(m, q, bc, nullptr, m.nIF);
        qhdFusedAdvanceKernel<<<m.fuBlocks, 256, lds, s>>>(m, q, c4new, ldsCell);
        QhdView q2 = q;
        q2.c4 = c4new;
        if (m.nBF) qhdBcKernel<<<gridOf(m.nBF), QGD_BLOCK, 0, s>>>(m, q2, bc);
        if (needRef) qhdRefReadKernel<<<1, 1, 0, s>>>(q2, localRefCell, refValue, shift);
        return true;
    }
    if (usesPoints) {
        pointInterpFastKernel<1><<<gridOf(m.nP), QGD_BLOCK, 0, s>>>(m, q.p, q.ptp);
        if (m.nBP) boundaryPointKernel<1><<<gridOf(m.nBP), QGD_BLOCK, 0, s>>>(m, q.pb, 1, q.ptp, 1, 0, -1);
    }
    switch (stencil) {
        case ST_REDUCED: face2<ST_REDUCED>(s, m, q, bc); break;
        case ST_LSQ: face2<ST_LSQ>(s, m, q, bc); break;
        case ST_GVP3: face2<ST_GVP3>(s, m, q, bc); break;
        default: face2<ST_GVP2>(s, m, q, bc); break;
    }
    qhdCellUpdateKernel<<<gridOf(m.nC), QGD_BLOCK, 0, s>>>(m, q);
    if (m.nBF) qhdBcKernel<<<gridOf(m.nBF), QGD_BLOCK, 0, s>>>(m, q, bc);
    if (needRef) qhdRefReadKernel<<<1, 1, 0, s>>>(q, localRefCell, refValue, shift);
    return false;
}
void launchQhdImplicitMatrix(hipStream_t s, const MeshView& m, const QhdView& q, const PatchBCDev* bc) {
    qhdImplicitFaceCoefKernel<<<gridOf(m.nF), QGD_BLOCK, 0, s>>>(m, q);
    qhdImplicitDiagKernel<<<gridOf(m.nC), QGD_BLOCK, 0, s>>>(m, q, bc);
}
void launchQhdImplicitAdvance(hipStream_t s, int stencil, bool usesPoints, const MeshView& m, const QhdView& q, const PatchBCDev* bc, int part,
                              int validMask, bool needRef, int localRefCell, double refValue, double* shift) {
    if (part == 0) {
        if (usesPoints) {
            pointInterpFastKernel<1><<<gridOf(m.nP), QGD_BLOCK, 0, s>>>(m, q.p, q.ptp);
            if (m.nBP) boundaryPointKernel<1><<<gridOf(m.nBP), QGD_BLOCK, 0, s>>>(m, q.pb, 1, q.ptp, 1, 0, -1);
        }
        switch (stencil) {
            case ST_REDUCED: face2<ST_REDUCED>(s, m, q, bc); break;
            case ST_LSQ: face2<ST_LSQ>(s, m, q, bc); break;
            case ST_GVP3: face2<ST_GVP3>(s, m, q, bc); break;
            default: face2<ST_GVP2>(s, m, q, bc); break;
        }
        qhdImplicitRhsKernel<<<gridOf(m.nC), QGD_BLOCK, 0, s>>>(m, q, bc);
    } else {
        qhdImplicitStoreKernel<<<gridOf(m.nC), QGD_BLOCK, 0, s>>>(m, q, validMask);
        if (m.nBF) qhdBcKernel<<<gridOf(m.nBF), QGD_BLOCK, 0, s>>>(m, q, bc);
        if (needRef) qhdRefReadKernel<<<1, 1, 0, s>>>(q, localRefCell, refValue, shift);
    }
}
// end of the step: the (global) shift of p [QHDFoam.C L123-130]
void launchQhdFinish(hipStream_t s, const MeshView& m, const QhdView& q, bool needRef, const double* shift) {
    if (needRef) qhdRefShiftKernel<<<gridOf((int64_t)m.nC + m.nBF), QGD_BLOCK, 0, s>>>(m.nC, m.nBF, q, shift);
}
// message kind 3: one value per listed cell of a single-precision vector (the iterate of the multigrid level that spans the ranks)
__global__ __launch_bounds__(QGD_BLOCK) void qhdHaloFloatKernel(float* __restrict__ vec, const int32_t* __restrict__ cells, const int nCells,
                                                               double* __restrict__ buf, const int pack) {
    const int i = blockIdx.x * QGD_BLOCK + threadIdx.x;
    if (i >= nCells) return;
    if (pack) buf[i] = (double)vec[cells[i]]; else vec[cells[i]] = (float)buf[i];
}
void launchQhdHaloFloat(hipStream_t s, float* vec, const int32_t* cells, int nCells, double* buf, bool pack) {
    if (nCells > 0) qhdHaloFloatKernel<<<gridOf(nCells), QGD_BLOCK, 0, s>>>(vec, cells, nCells, buf, pack ? 1 : 0);
}
void launchQhdHalo(hipStream_t s, const QhdView& q, double* direction, int kind, const int32_t* cells, int nCells, const int32_t* bfaces, int nFaces,
                   double* buf, bool pack) {
    const int n = nCells + (kind == 2 ? 0 : nFaces);
    if (n > 0) qhdHaloKernel<<<gridOf(n), QGD_BLOCK, 0, s>>>(q, direction, kind, cells, nCells, bfaces, kind == 2 ? 0 : nFaces, buf, pack ? 1 : 0);
}
void launchQhdExtract(hipStream_t s, int64_t n, const double* rec4, int field, double* out) {
    if (n) qhdExtractKernel<<<gridOf(n), QGD_BLOCK, 0, s>>>(n, rec4, field, out);
}

}  // namespace qgd
