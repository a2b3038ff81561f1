// qgd_device.hpp -- device-side views and launch entry points (gfx950).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <functional>

namespace qgd {

// ---- gathered records (AoS, 16-B aligned so they move as dwordx4) ----------
struct alignas(16) RecA { double rho, ux, uy, uz, p, e; };   // 48 B: fields whose face gradients are needed
struct alignas(16) RecB { double H, c, muQGD, aOc; };        // 32 B: derived per-cell quantities the face kernel interpolates

enum StencilKind : int { ST_REDUCED = 0, ST_LSQ = 1, ST_GVP3 = 2, ST_GVP2 = 3 };

// Static mesh data on the device; every face array is indexed by global face label.
struct MeshView {
    int32_t nP, nF, nIF, nC, nBF;
    int32_t ie1, ie2, ie3;
    int32_t nGeomD;          // mesh.nGeometricD()
    int32_t emptyDir[3];     // 1 for the directions of empty patches (vector components the segregated solves skip)
    int32_t cblock, pblock;  // tiles of the cell-update and vertex kernels (64, 128 or 256)
    int32_t fblock;          // face tile of the 3-D GaussVolPoint kernel: 64, 128 or 256 faces per workgroup
    int32_t hasOther;        // 1: some internal face has more than four vertices (FK_OTHER)
    int32_t xcdRun;          // tiles per XCD run of the workgroup->tile map (0: one contiguous eighth per XCD)
    const int32_t* own;      // nF
    const int32_t* nei;      // nIF
    const int4* verts;       // nF
    const uint8_t* fkind;    // nF
    const double* Sx; const double* Sy; const double* Sz;  // nF
    const double* magSf;     // nF
    const double* w;         // nF
    const double* hf;        // nF
    const double* dn;        // nF
    const double* X;         // 3*nP vertex coordinates, packed
    const double* Cc;        // 3*nC cell centres, packed
    const double4* bN;       // nBF mirror points of boundary faces
    const double* bmvON;     // nBF
    const int2* ip13;        // nF
    const double* c2d;       // 6*nF
    const int32_t* lsqSlice; const uint8_t* lsqCnt; const int32_t* lsqCell;               // sliced ELL (64-face slices)
    const double* lsqGx; const double* lsqGy; const double* lsqGz; const uint8_t* lsqDeg; const uint8_t* lsqBndZero;
    const int32_t* pcSlice; const uint8_t* pcCount; const int32_t* pcCell; const double* pcW;  // sliced ELL (64-point slices)
    int32_t nBP; const int32_t* bpPoint; const int32_t* bpOff; const int32_t* bpFace; const double* bpW;
    // point constraints of vector / tensor vertex fields on symmetryPlane / symmetry / wedge patches (qgd_setup.hpp StaticData::cpOff);
    // cpOff == nullptr: the mesh has none
    const int32_t* cpOff; const uint8_t* cpKind; const double* cpT;
    const uint8_t* bSymm;    // nBF: 1 on faces of symmetryPlane / symmetry / wedge patches (nullptr: the mesh has none)
    const int32_t* cfSlice; const uint8_t* cfCount; const int32_t* cfItem;                     // sliced ELL (64-cell slices)
    const int32_t* fpos;     // nIF: storage position of an internal face's net fluxes (slot-major, qgd_setup.hpp)
    const int32_t* cfPos;    // cfItem with positions instead of labels: gather list of the cell kernel
    const int32_t* cfNbr;    // cfItem's neighbour cells (-1: boundary face): gather list of the matrix products
    // {w, Sx, Sy, Sz} of every face at its slot-major POSITION (fpos; patch faces at their label), w = -1 on faces of empty patches:
    // what fvc::grad(U) per cell gathers (cellGradGauss).  nullptr until a QHD or implicitDiffusion case asks for it (32 B per face).
    const double4* geoPos;
    // face tiles of the LDS-staged 3-D GaussVolPoint kernel (qgd_setup.hpp FaceTiles); tileOff == nullptr: gather kernel
    const int32_t* tileOff; const int32_t* tileCells; const int32_t* tileVerts;
    const uint32_t* locC; const uint2* locV;
    const int32_t* tileSpill; int32_t nTileSpill;   // tiles left to the gather kernel
    int32_t tileLds;         // dynamic LDS bytes of the largest tile
    int32_t tileMaxC, tileMaxV;   // distinct cells / vertices of the largest staged tile
    // QGD_FTILE_FIXED (default 1; 128-face tiles, 3 waves per SIMD): the same lists at a fixed stride of tileMaxC /
    // tileMaxV labels per tile (padded with the tile's last label; all zero for the tiles of tileSpill, which carry tileFlag = 1)
    const int32_t* tileCellsFix; const int32_t* tileVertsFix; const uint8_t* tileFlag;
    int32_t qhdTiles;        // QGD_QHD_TILES (default 1): QHD's two face passes use the tiles too (qgd_qhd.hip qhdFace{1,2}TileKernel)
    int32_t implTiles;       // QGD_IMPL_TILES (default 1): so does the face kernel of QGDFoam's implicit branch (qgd_implicit.hip implFaceTileKernel)
    int32_t tileWaves;       // waves per SIMD the staged kernel is compiled for (2, 3 or 4)
    int32_t sGeo;            // 1: the 3-D GaussVolPoint kernels rebuild Sf of quadrilateral faces from the vertices (no Sf stream)
    const double* V; const double* hQGD; const uint8_t* ghost;
    const int32_t* bPatch; const double* hQGDb;
    // cell blocks of the fused face + cell kernel (qgd_setup.hpp FusedBlocks); fuBlocks == 0: not built
    int32_t fuBlocks, fuLayerBlocks, fuCapC, fuCapV, fuCapF, fuCapE, fuLds, fuLdsCell;   // fuLayerBlocks: a shard's boundary-layer blocks come first;   // fuLdsCell: where the per-cell park of the kernel starts, in doubles
    const int4* fuHdr; const int32_t* fuCells; const int32_t* fuVerts; const int32_t* fuFaceLabel; const uint8_t* fuNEntry;
    // a block's local topology sits in its template (fuHdr2[blk].y): 3 position words per face, the own cells' face entries, the vertices' cell positions
    const uint32_t* fuFacePos; const int32_t* fuEntry; int32_t fuTemplates;
    int32_t fuXcdRun;        // QGD_FU_XCD_RUN (default 64): consecutive blocks dealt to one XCD before the next XCD's run (the face tiles' xcdRun is 16)
    int32_t fuLdsImpl, fuLdsCellImpl;   // the same two figures as fuLds / fuLdsCell for the implicitDiffusion branch's layout (+ 72 B of fvc::grad(U) per own / across-a-face cell, eight flux planes)
    int32_t fuCapPE, fuMaxTot, fuMaxAll, fuMaxV, fuMaxF;   // cells per vertex (stride); staged cells incl. / without the extra ones, vertices, faces: maxima over the blocks (information; a block lays its LDS out by its own counts)
    const int4* fuHdr2; const uint8_t* fuVCount; const uint16_t* fuVPos; const double* fuVW;   // the vertex values are formed inside the block
};

// Per-patch boundary-condition table (device copy, <= 64 patches)
struct PatchBCDev {
    int32_t bcU, bcT, bcP, ptype; double vU[3]; double vT, vP;
    // symmetryPlane patches: the one normal of the patch (qgd_mesh.hpp Patch::nHat), used on every face where basicSymmetry (slip,
    // symmetry) uses the face's own normal (L0: symmetryPlaneFvPatchField::evaluate / snGrad / snGradTransformDiag)
    double nHat[3]; int32_t planeN, pad_;
};
#define QGD_MAX_PATCHES 64

struct GasModel {
    double R, Cv, mu0, Pr, ScQGD, PrQGD, alphaQGD;
    double gamma;     // Cp/Cv
    double alphah0;   // (Cp*mu*rPr)/Cp
    int32_t consistentEnergy;  // qgd_case_options::consistentEnergy
    int32_t implicitDiffusion; // qgd_case_options::implicitDiffusion
    double rPrQGD;             // 1/PrQGD
    int32_t upwindU, upwindH;  // qgd_case_options::fluxSchemeU / fluxSchemeH == QGD_FLUX_UPWIND
};

// Mutable case state on the device
#define QGD_FACE_REDUCE_PARTIALS 1024
struct CaseView {
    RecA* A; RecB* B;               // nC
    RecA* A2; RecB* B2;             // nC, fused step only: the records the step writes (it reads its neighbours' old ones); swapped with A, B after the step
    double* rE;                     // nC total energy rho*E (rhoU is rho*U of the record: the explicit re-solve keeps them equal)
    RecA* P;                        // nP vertex records
    RecA* bA; RecB* bB;             // nBF boundary records
    double* bG;                     // nBF p gradient (qgdFlux)
    double* bPhiw;                  // nBF phiwStar on boundary faces
    double* bPmid;                  // nBF patch pressure after GaussVolPoint's mid-step BC evaluation
    double* bRhoLag;                // nBF patch density of the previous step: rhoU_b, rhoE_b are built with it
                                    //     [QGDUEqn_8H L88-89, QGDEEqn_8H L75-76 run before QGDFoam_8C L156]
    double* cellSum;                // 5*nC, fused step under Courant-number control only: the ordered net flux sums of every owned cell (the blocks stop there until deltaT is known)
    double* flux;                   // 5*nF net face fluxes, SoA: flux[k*nF + fpos[f]] (boundary faces: their label)
    double* red;                    // [0]=max Co, [1]=min tauQGDf, [2]=min rho, [3]=min e
    double* blkFace;                // 2 per face-kernel workgroup (internal then boundary): max Cof, min tauQGDf
    double* blkFace2;               // 2 x QGD_FACE_REDUCE_PARTIALS: the first level of their fold (launchFaceReduce)
    double* blkCell;                // 2 per cell-kernel workgroup: min rho, min e since the last query
    int32_t nBlkFace, nBlkCell;
    double* dt;                     // [0]=deltaT (device resident so adjustTimeStep needs no host round trip)
    double* dbg;                    // optional debug face fields (nullptr in the product path)
    // non-uniform alphaQGD / ScQGD (the READ_IF_PRESENT fields of QGDCoeffs_8C L119-160, constScPrModel1_8C L66-79);
    // nullptr = the uniform value of GasModel
    const double* aQ; const double* aQb; const double* sc; const double* scb;
};

enum DebugSlot : int {
    DBG_PHIJM = 0, DBG_PHIJMU = 1, DBG_PHIP = 4, DBG_PHIPI = 7, DBG_PHIJMH = 10, DBG_PHIQ = 11, DBG_PHIPIU = 12,
    DBG_PHIW = 13, DBG_PHI = 14, DBG_TAU = 15, DBG_GRADU = 16, DBG_GRADE = 25, DBG_GRADRHO = 28, DBG_GRADP = 31,
    DBG_COUNT = 34
};

struct Launcher {
    hipStream_t stream;
    // timing hooks (set by the C-ABI layer)
    void (*pre)(void* ctx, int k);
    void (*post)(void* ctx, int k);
    void* ctx;
};

// ---- case kernels -------------------------------------------------------------
void launchPointInterp(const Launcher& L, const MeshView& m, const CaseView& c);
void launchBoundaryPoints(const Launcher& L, const MeshView& m, const CaseView& c, bool pOnly);
void launchFaceFlux(const Launcher& L, int stencil, const MeshView& m, const CaseView& c, const GasModel& g, bool adjustDt);
void launchBoundaryFaceFlux(const Launcher& L, int stencil, const MeshView& m, const CaseView& c, const GasModel& g,
                            const PatchBCDev* bc, int phiwOnly, bool adjustDt);
// per-term fvsc entries: stencils a < b, component k of the six-component gradient by b where bit k of maskB is set (rho, Ux, Uy, Uz, p, e)
void launchFaceFluxMixed(const Launcher& L, int a, int b, int maskB, const MeshView& m, const CaseView& c, const GasModel& g, bool adjustDt);
void launchBoundaryFaceFluxMixed(const Launcher& L, int a, int b, int maskB, const MeshView& m, const CaseView& c, const GasModel& g,
                                 const PatchBCDev* bc, int phiwOnly, bool adjustDt);
void launchFusedFaceCell(const Launcher& L, const MeshView& m, const CaseView& c, const GasModel& g, int firstBlock, int nBlocks);
void launchFusedAdjust(const Launcher& L, const MeshView& m, const CaseView& c, const GasModel& g);   // Courant-number control: every block up to its flux sums
void launchCellFinish(const Launcher& L, const MeshView& m, const CaseView& c, const GasModel& g, int mode, const int32_t* list, int nList);
struct ImplView;
bool fusedImplUPrepare(const MeshView& m, const GasModel& g);   // raises the dynamic-LDS limit of the IMPL instantiation; false: not available
void launchFusedImplU(const Launcher& L, const MeshView& m, const CaseView& c, const GasModel& g, const ImplView& iv, const PatchBCDev* bc);
void launchCellUpdate(const Launcher& L, const MeshView& m, const CaseView& c, const GasModel& g, int mode,
                      const int32_t* list, int nList);
void launchBoundaryUpdate(const Launcher& L, const MeshView& m, const CaseView& c, const GasModel& g, const PatchBCDev* bc,
                          bool init, bool phiwRegistered, int mode, const int32_t* list, int nList);
void launchCellInit(const Launcher& L, const MeshView& m, const CaseView& c, const GasModel& g, const double* U,
                    const double* T, const double* p);
void launchDeltaT(const Launcher& L, const CaseView& c, double maxCo, double maxDeltaT, double cTau);
void launchFaceReduce(const Launcher& L, const CaseView& c);
void launchResetReductions(const Launcher& L, const CaseView& c);
void launchCellMinReduce(const Launcher& L, const CaseView& c);
int faceBlocks(const MeshView& m);
int bfaceBlocks(const MeshView& m);
int cellBlocks(const MeshView& m);
#define QGD_HALO_CELL_DOUBLES_HOST 8   // doubles per cell of the state message (qgd_kernels.hip haloKernel), 12 per patch face
void launchHaloPack(const Launcher& L, const CaseView& c, const GasModel& g, const int32_t* cells, int32_t nCells, const int32_t* bfaces,
                    int32_t nFaces, double* buf, bool pack);
void launchMidHalo(hipStream_t s, const CaseView& c, const int32_t* bfaces, int32_t n, double* buf, bool pack);
void launchFaceGeoPos(hipStream_t s, const MeshView& m, double4* out);   // fills MeshView::geoPos

// ---- accessor: one named cell / patch field out of the records (K == nullptr on patches) ----------------------------
enum ExtractField : int { XF_RHO = 0, XF_U, XF_P, XF_E, XF_T, XF_RHOU, XF_RHOE, XF_C, XF_PSI, XF_MU, XF_ALPHAU, XF_TAUQGD, XF_MUQGD,
                          XF_ALPHAUQGD, XF_HQGD, XF_H, XF_GAMMA };
void launchExtractField(hipStream_t s, const RecA* A, const RecB* B, const double* rE, const double* hq, const double* aQ, int64_t n,
                        const GasModel& g, int field, double* out);

// ---- fvsc operators on plain fields ----------------------------------------------
// op: 0 grad (out 3*NC per face), 1 div (out NC/3 per face); NC in {1,3,9}
void launchFvscOp(hipStream_t s, int stencil, int op, int NC, const MeshView& m, const double* cell, const double* bnd,
                  double* pt, double* out);

void launchPack5(hipStream_t s, int64_t n, const double* U, const double* T, const double* p, double* out);
void launchSoaToAos(hipStream_t s, int64_t n, int nc, const double* src, double* dst);
void launchInterpolate(hipStream_t s, int NC, const MeshView& m, const double* cell, const double* bnd, double* out);
void launchFluxUpwind(hipStream_t s, int NC, const MeshView& m, const double* flux, const double* cell, const double* bnd, double* out);

// ---- QHDFoam face fluxes --------------------------------------------------------
// rec5 = {Ux,Uy,Uz,T,p} per cell / patch face / vertex; out = 26 SoA slots of nF doubles (see QhdSlot)
enum QhdSlot : int { QHD_GRADU = 0, QHD_GRADT = 9, QHD_PHIU = 12, QHD_PHIWO = 13, QHD_TAUBYRHO = 14, QHD_GRADP = 15,
                     QHD_WF = 18, QHD_PHIUF = 21, QHD_PHITF = 24, QHD_PHITAUT = 25, QHD_COUNT = 26 };
void launchQhdFluxes(hipStream_t s, int stencil, const MeshView& m, const double* cell5, const double* bnd5, double* pt5,
                     const double* rho, const double* rhob, const double* tau, const double* phi, double beta,
                     double gx, double gy, double gz, double* out);

// ---- species flux block: out = 5 SoA slots of nF doubles {phiJmY, diffusiveFlux, gradYf(3)} -------------------
void launchSpeciesFlux(hipStream_t s, int stencil, const MeshView& m, const double* Y, const double* Yb, double* ptY,
                       const double* U, const double* Ub, const double* phiJm, const double* phi, const double* tau, double* out);
void launchSpeciesStep(hipStream_t s, const MeshView& m, const double* Yc, const double* Yb, const double* rhoOld, const double* rho,
                       const double* phiJmY, const double* muf, double Sc, double dt, const double* Su, double* diffusiveFlux, double* net,
                       double* Ynew);

// the species equation with fvm::laplacian [QGDYEqn_8H L47-66], one species; work = 3 * nC + nF doubles (qgd_implicit.hip)
struct ImplicitSolver;
void launchSpeciesStepImplicit(ImplicitSolver* S, const MeshView& m, const double* Yc, const double* Yb, const uint8_t* fixedFace, const double* rhoOld,
                               const double* rho, const double* phiJmY, const double* muf, double Sc, double dt, const double* Su, double tol, int maxIter,
                               double* work, double* diffusiveFlux, double* Ynew, double info[3]);

// ---- implicitDiffusion branch of QGDFoam (qgd_implicit.hip) -----------------------------------------------------------
#if defined(__HIPCC__)
// Workgroup b runs on XCD b % 8, each XCD with a private 4 MiB L2.  Dealt round-robin, the 256-row blocks of a row-wise kernel put a row
// and its neighbours one mesh row away on different XCDs, and every gathered line is fetched by about five of them (measured: the
// implicit matrix product read 217 B per cell where 120 are compulsory, a level-0 multigrid sweep 73 B per row where 60 are).  Runs of
// `run` consecutive blocks per XCD keep those neighbours in one L2; run <= 0: the plain order.  (xcdTile of the explicit kernels is the
// same map, with run 0 meaning one contiguous eighth per XCD.)
__device__ __forceinline__ int xcdRunBlock(const int run) {
    const int b = blockIdx.x;
    if (run <= 0) return b;
    const int span = run << 3, full = ((int)gridDim.x / span) * span;
    if (b >= full) return b;
    const int xcd = b & 7, i = b >> 3;
    return ((i / run) * 8 + xcd) * run + (i % run);
}
#endif

struct ImplView {
    double* gUc;                         // 9*nC fvc::grad(U)
    double *phiTau, *UfS;                // 3*nF SoA: Sf & tauMC, Uf
    double *sTau, *mufS, *aU, *aE, *phiSig;   // nF: Sf.(tauMC & Uf), muf, laplacian coefficients of U and e, phiSigmaDotU
    double* rhoNew;                      // nC
    double *xU, *diagU, *rhsU;           // 3*nC SoA: the component systems of UEqn
    double *xE, *diagE, *rhsE;           // nC
    // start values of the two solves (QGD_IMPL_XEXTRAP, qgd_implicit.hip "start values"): the predictor each solve would start from as the
    // listing has it (U = rhoU/rho, e = rhoE/rho - |U|^2/2), and the corrections solution - predictor of the last `have` steps, newest first
    // (the oldest doubles as this step's write target: dh[order - 1]); 4*nC each, component-major {Ux, Uy, Uz, e}; pred == nullptr: off
    double *pred, *dh[4];
    int have, order;
    // adjustTimeStep: the steps of the history differ in length, so the extrapolation takes Lagrange weights from the deltaT ratios instead of
    // the binomial ones (w[0..3], written on the device by implStartWeightsKernel from the ring dtHist[0..4] = deltaT of this step and of the
    // four before); nullptr with a fixed deltaT
    double *w, *dtHist;
};
// the branch as parts 0..5 (gradient | faces + U systems | store U | gradient of the new U | sigma + e system | finish) around its two
// multi-right-hand-side Jacobi-PCG solves, all stream-ordered with the scalars of the solves in a device control block
struct ImplicitSolver;
struct SolveHooks;
void launchImplicitStartWeights(hipStream_t s, const CaseView& c, const ImplView& iv);   // after deltaTKernel, before the step's first start value
ImplicitSolver* implicitSolverCreate(hipStream_t stream, const MeshView& m, int ownedBegin = 0, int ownedEnd = -1);
void implicitSolverFree(ImplicitSolver* S);
int64_t implicitSolverBytes(const ImplicitSolver* S);
double* implicitSolverCtl(ImplicitSolver* S);          // 68 doubles, slot-major: ctl[slot * 4 + component]; reduced slots 0..2, 3, 4, 5, 6..7
double* implicitSolverDirection(ImplicitSolver* S);    // 3 * nC doubles, component-major
int implicitSolverRhs(const ImplicitSolver* S);        // right-hand sides of the solve in flight (3: U, 1: e)
void implicitSolveSetup(ImplicitSolver* S, int nRhs, int validMask, const double* a, const double* diag, const double* rhs, double* x, double tol,
                        int maxIter, const double* gamma = nullptr);   // gamma[k] scales the face coefficients of component k (nullptr: 1)
bool implicitSolverChebyshev(const ImplicitSolver* S);   // the algorithm of the solves: Chebyshev (default) or conjugate gradients (QGD_IMPL_SOLVER=pcg)
double* implicitSolverIterate(ImplicitSolver* S);        // the buffer that holds the iterate after the steps queued so far (halo kind 3, Chebyshev)
void implicitSolvePhase(ImplicitSolver* S, int phase);
void implicitSolveRun(ImplicitSolver* S, const SolveHooks* hooks);
void implicitSolveStatus(ImplicitSolver* S, double* allDone, int iters[3], double res0[3], double res[3]);
void implicitSolveStatus4(ImplicitSolver* S, double* allDone, int iters[4], double res0[4], double res[4]);   // the same for up to four right-hand sides
double implicitSolverUnconverged(ImplicitSolver* S, double* stalledSteps);   // steps since implicitStatsReset in which a solve stopped above its tolerance (waits)
// ghost entries of what the next matrix product reads (Chebyshev: the iterate; conjugate gradients: the search direction): its
// right-hand sides per listed cell, cell-major in the message
void launchSolverHalo(hipStream_t s, ImplicitSolver* S, const int32_t* cells, int nCells, double* buf, bool pack);
void implicitSolveEnd(ImplicitSolver* S, int which);             // which = 0: the U solve, 1: the e solve
double implicitApplyMs(ImplicitSolver* S, const ImplView& iv, int reps, int* rows);   // measurement: average ms of the U system's matrix product
void implicitStepMark(ImplicitSolver* S, bool begin);
void implicitStatsReset(ImplicitSolver* S);
void implicitSolverInfo(ImplicitSolver* S, int iters[4], double res0[4], double res[4], double* unconvergedSteps, double* stalledSteps);
int implicitHaloWidth(const ImplicitSolver* S, int kind);   // doubles per cell of message kind 1 (grad U), 2 (U), 3 (search direction), 4 (initial guess)
void launchImplicitHalo(hipStream_t s, const MeshView& m, const CaseView& c, const ImplView& iv, ImplicitSolver* S, int kind, const int32_t* cells,
                        int nCells, double* buf, bool pack);
void implicitSolverSetStream(ImplicitSolver* S, hipStream_t s);
void launchImplicitPart(hipStream_t s, const MeshView& m, const CaseView& c, const ImplView& iv, const GasModel& g, const PatchBCDev* bc,
                        ImplicitSolver* S, double tol, int maxIter, int part, bool fusedU = false);

// ---- QHDFoam case resident on the device (qgd_qhd.hip) ---------------------------------------------------------------
struct QhdView {
    double* c4;  double* b4;  double* pt4;     // {Ux,Uy,Uz,T}: cells (nC*4), patch faces (nBF*4), vertices (nP*4)
    double* p;   double* pb;  double* pgb;  double* ptp;   // p: cells, patch values, patch gradients, vertices
    const double* tauF;                        // nF tauQGDf
    double *phiu, *phiwo, *phi, *phitr;        // nF
    double* ugu;                               // 3*nF SoA: Uf & gradUf (BdFrcf is formed again in face pass 2)
    double* gUc;                               // 9*nC fvc::grad(U)
    double* F;                                 // 4*nF SoA at the faces' slot-major positions (MeshView::fpos): net face terms of the U (3) and T equations
    double rho0, nu, Hi, beta, g[3], dt;
    int32_t upwindU, upwindT;                  // qgd_qhd_options::fluxSchemeU / fluxSchemeT == QGD_FLUX_UPWIND
    // implicitDiffusion [QHDUEqn_8H L46-65, QHDTEqn_8H L69-80]: the four systems {Ux, Uy, Uz, T} share the face coefficients
    // aG = |Sf| delta_f (at the faces' slot-major positions) up to gamma = {nu, nu, nu, Hi}; the matrix does not change in time
    // (thermo is not corrected inside the loop, deltaT is fixed): diag4 is built once, rhs4 / x4 every step (component-major, 4 * nC)
    int32_t implicit;
    double *aG, *diag4, *rhs4, *x4;
    // start values of the four systems (QGD_IMPL_XEXTRAP): the fields {U, T} of the last `xHave` steps before the current one, newest first
    // (the oldest, xd[xOrder - 1], doubles as the slot the current fields go into); 4 * nC each, component-major; xOrder == 0: off (start from
    // the current fields, as OpenFOAM does)
    double* xd[4]; int xHave, xOrder;
    int32_t tauModel;                          // 0 constTau, 1 HbyUQHD, 2 T0byGr, 3 H2bynuQHD
    double Tau, aQGD, UQHD, T0, Gr;
};
void launchQhdInit(hipStream_t s, const MeshView& m, const QhdView& q, const PatchBCDev* bc, const double* U, const double* T, const double* p,
                   double* tauF, double* taubyrho);
void launchQhdAssemble(hipStream_t s, int stencil, bool usesPoints, const MeshView& m, const QhdView& q, const PatchBCDev* bc);
void launchQhdExtrapolateP(hipStream_t s, int nC, double* p, double* const hist[4], int have, int order);
void launchQhdPostSolve(hipStream_t s, const MeshView& m, const QhdView& q, const PatchBCDev* bc);
bool launchQhdAdvance(hipStream_t s, int stencil, bool usesPoints, const MeshView& m, const QhdView& q, const PatchBCDev* bc, bool needRef,
                      int localRefCell, double refValue, double* shift, double* c4new = nullptr);   // true: the new {U,T} are in c4new
bool qhdFusedAdvanceEligible(int stencil, const MeshView& m, const QhdView& q, int* ldsBytes, int* ldsCell);
void launchQhdFinish(hipStream_t s, const MeshView& m, const QhdView& q, bool needRef, const double* shift);
// implicitDiffusion: the constant matrix (set-up), then per step part 0 = face pass 2 + right-hand sides and start values (the
// solve of the four systems follows: implicitSolveSetup with gamma = {nu, nu, nu, Hi}), part 1 = the solution into the records,
// U / T boundary conditions, this rank's share of the reference-level shift of p
void launchQhdImplicitMatrix(hipStream_t s, const MeshView& m, const QhdView& q, const PatchBCDev* bc);
void launchQhdImplicitAdvance(hipStream_t s, int stencil, bool usesPoints, const MeshView& m, const QhdView& q, const PatchBCDev* bc, int part,
                              int validMask, bool needRef, int localRefCell, double refValue, double* shift);
// halo messages of a sharded QHD case: kind 0 = state {U,T} (4 per cell, 4 per patch face), 1 = p + fvc::grad(U) (10 per cell, 2 per
// patch face), 2 = the search direction of the pressure solve (1 per cell)
void launchQhdHalo(hipStream_t s, const QhdView& q, double* direction, int kind, const int32_t* cells, int nCells, const int32_t* bfaces, int nFaces,
                   double* buf, bool pack);
void launchQhdHaloFloat(hipStream_t s, float* vec, const int32_t* cells, int nCells, double* buf, bool pack);
void launchQhdExtract(hipStream_t s, int64_t n, const double* rec4, int field, double* out);

// persistent pressure solver: PCG preconditioned by aggregation multigrid (precond 1) or Jacobi (0); qgd_poisson.hip
struct PressureSolver;
// ownedBegin/ownedEnd: the rows of the system, i.e. the owned cells of a shard (0, -1: every cell)
PressureSolver* pressureSolverCreate(hipStream_t stream, const MeshView& m, const double* taubyrho, const uint8_t* bKind, int refCell, int precond,
                                     int ownedBegin = 0, int ownedEnd = -1, const int32_t* cellGlobal = nullptr, int64_t cellGlobalOffset = 0,
                                     bool sharded = false);   // sharded: cellGlobal (or local + offset) = labels of the unsharded mesh
void pressureSolverFree(PressureSolver* S);
int64_t pressureSolverBytes(const PressureSolver* S);
int pressureSolverLevels(const PressureSolver* S, int* sizes, int cap);
int pressureSolve(PressureSolver* S, const double* phiu, const double* phiwo, const double* pb, const double* gb, double tolerance,
                  double relTol, int maxIter, double* p, double* phi, double residuals[2]);
// the same solve as stream-ordered phases (qgd_poisson.hip): a sharded caller reduces the control block and exchanges the
// ghost entries of the search direction between them
struct SolveHooks {
    std::function<void(double* devicePtr, int n)> allreduce;   // SUM over the ranks, in place, stream-ordered
    // the distributed multigrid hierarchy of a sharded pressure solve (pressureSolvePending): op 2 = SUM, 3 = MAX over the ranks of a
    // device buffer, in place; haloMg: ghost entries of the multigrid iterate (message kind 3 of the QHD case)
    std::function<void(double* devicePtr, int64_t n, int op)> allreduceBuf;
    std::function<void()> haloMg;
    std::function<void()> haloDirection;                       // ghost entries of pressureSolverDirection()
    std::function<void()> haloGuess;                           // implicit branch: ghost entries of the initial guess, before the first product
};
void pressureSolveBegin(PressureSolver* S, const double* phiu, const double* phiwo, const double* pb, const double* gb, double tolerance,
                        double relTol, int maxIter, double* p);
void pressureSolvePhase(PressureSolver* S, int phase);
// what the caller owes the phase in flight before pressureSolveContinue: 0 nothing (the phase is complete), 1 halo of the multigrid
// iterate (pressureSolverMgHaloVec: one float per local cell), 2 / 3 SUM / MAX all-reduce of *buf[0, *n)
int pressureSolvePending(PressureSolver* S, double** buf, int64_t* n);
void pressureSolveContinue(PressureSolver* S);
float* pressureSolverMgHaloVec(PressureSolver* S);
int pressureSolveRun(PressureSolver* S, const SolveHooks* hooks, double residuals[2]);
void pressureSolveFlux(PressureSolver* S, double* phi);
void pressureSolveStatus(PressureSolver* S, double out[4]);
double pressureSolverSweepMs(PressureSolver* S, int reps, int* rows, double* width);   // measurement: one level-0 smoothing sweep
bool pressureSolverSinglePrecisionCycle(const PressureSolver* S);
double* pressureSolverCtl(PressureSolver* S);         // control block: slots [0,3) [3] [4] [5] [6,8) [8] are the sums a sharded run reduces
double* pressureSolverDirection(PressureSolver* S);   // nC doubles by local cell label

// ---- QHDFoam pressure equation (qgd_poisson.hip) -------------------------------------
// all pointers are device memory; work holds 8*nC + nF + max(nBF,1) + 3*ceil(nC/256) + 8 doubles
int solveQhdPressure(hipStream_t stream, const MeshView& m, const double* gamma, const double* phiu, const double* phiwo,
                     const uint8_t* bKind, const double* pb, const double* gb, int refCell, double refValue, double tolerance,
                     double relTol, int maxIter, double* p, double* phi, double* work, double residuals[2]);

}  // namespace qgd
