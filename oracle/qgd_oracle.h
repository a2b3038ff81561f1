/*
 * qgd_oracle.h -- C interface of the CPU ORACLE.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it, and only as the checker.  The product library (libqgd_amd.so) does
 * not link, include or call anything in this directory.
 *
 * PARITY UNPINNED: the reference snapshot (/root/reference) holds no source
 * files, tests, tutorials or golden vectors -- only Doxygen listings -- and its
 * dependency OpenFOAM v2312 is absent, so the reference cannot be built or run
 * here.  This oracle is a restatement of those listings,
 * operation by operation in the reference's evaluation order, pinned only by
 * the analytic known-answer properties in tests/test_oracle_properties.py.
 */
#ifndef QGD_ORACLE_H
#define QGD_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* numeric codes are the same as include/qgd_amd.h (QGD_PATCH_*, QGD_BC_*,
 * QGD_FVSC_*) so the Python tests can share them */

typedef struct orc_case_options {
    int32_t stencil, implicitDiffusion, adjustTimeStep, consistentEnergy;
    double R, Cv, mu, Pr, ScQGD, PrQGD, alphaQGD, deltaT, maxCo, maxDeltaT, cTau;
    double implicitTol;        /* fvSolution tolerance of the two implicit-diffusion solves */
    int32_t implicitMaxIter;
    int32_t fluxSchemeU, fluxSchemeH;   /* 0 flux*psif (= Gauss linear), 1 Gauss upwind: divSchemes entry of qgdFlux's flux [QGDInterpolate.H L86-104] */
    int32_t pad_;
    int32_t termStencil[4];             /* fvsc entries of grad(U), grad(e), grad(rho), grad(p): 0 = default, else 1 + FVSC_* [fvsc.C L51-58] */
} orc_case_options;

void* orc_mesh_create(int32_t nPoints, const double* points, int32_t nFaces,
                      const int32_t* faceOffsets, const int32_t* facePoints,
                      int32_t nInternalFaces, const int32_t* owner,
                      const int32_t* neighbour, int32_t nCells, int32_t nPatches,
                      const int32_t* patchStart, const int32_t* patchSize,
                      const int32_t* patchType);
void orc_mesh_free(void* m);
/* "Sf","magSf","Cf","C","V","weights","deltaCoeffs","nonOrthDeltaCoeffs" */
int orc_mesh_get(void* m, const char* name, double* out, int64_t n);
/* geometry supplied by the caller (counterpart of qgd_mesh_set_geometry); derived coefficients are rebuilt */
int orc_mesh_set_geometry(void* m, const double* Sf, const double* Cf, const double* C, const double* V);
/* info[0]=nGeometricD, info[1..3]=geometricD */
int orc_mesh_info(void* m, int64_t info[4]);
/* halo lists of a cell-range shard, one call per halo slot (= neighbouring shard; box slabs: 0 lower, 1 upper) */
int orc_mesh_set_halo(void* m, int side, int32_t nGhost, const int32_t* ghost,
                      int32_t nSend, const int32_t* send);

/* the faceSet degenerateStencilFaces of the leastSquares stencil (internal face labels) */
int orc_mesh_set_degenerate_faces(void* m, int32_t n, const int32_t* faces);
/* hQGDf of the halo-patch faces as the unsharded mesh has it (patch order), so that a ghost cell's hQGD comes out right */
int orc_mesh_set_halo_face_h(void* m, int32_t n, const double* h);

/* fvsc operators: scheme word as in fvSchemes ("reduced","leastSquares",
 * "leastSquaresOpt","GaussVolPoint"); op = "grad_s","grad_v","div_v","div_t".
 * Returns 0, or -4 when the scheme is refused (leastSquares in 3-D),
 * -5 unknown word. */
int orc_fvsc(void* m, const char* scheme, const char* op, const double* cell,
             const double* bnd, double* out);

/* QHDFoam face fluxes; same argument meaning as qgd_qhd_fluxes (pointers may be NULL as there) */
typedef struct orc_qhd_inputs {
    const double *U, *Ub, *T, *Tb, *p, *pb, *rho, *rhob, *tauQGDf, *phi;
    double beta; double g[3];
} orc_qhd_inputs;
typedef struct orc_qhd_outputs {
    double *gradUf, *gradTf, *phiu, *phiwo, *taubyrhof, *gradPf, *Wf, *phiUf, *phiTf, *phiTauTReg;
} orc_qhd_outputs;
int orc_qhd_fluxes(void* mesh, const char* scheme, const orc_qhd_inputs* in, orc_qhd_outputs* out);

/* species flux block of reactingLagrangianQGDFoam/updateFluxes.H L117-132; same argument meaning as qgd_species_flux */
int orc_species_flux(void* mesh, const char* scheme, const double* Y, const double* Yb, const double* U, const double* Ub,
                     const double* phiJm, const double* phi, const double* tauQGDf, double* phiJmY, double* diffusiveFlux,
                     double* gradYf);

/* one species of QGDYEqn.H L44-45, L67-86 (explicit branch, explicit source Su or NULL); same argument meaning as qgd_species_step */
int orc_species_step(void* mesh, const double* Y, const double* Yb, const double* rhoOld, const double* rho, const double* phiJmY,
                     const double* muf, double Sc, double deltaT, const double* Su, double* diffusiveFlux, double* Ynew);
/* QGDYEqn.H L47-66: the implicitDiffusion branch of the same equation (fvm::laplacian, diffusiveFlux += YEqn.flux()); info = {iterations,
 * initial, final residual} */
int orc_species_step_implicit(void* mesh, const double* Y, const double* Yb, const uint8_t* fixedValueFace, const double* rhoOld, const double* rho,
                              const double* phiJmY, const double* muf, double Sc, double deltaT, const double* Su, double tolerance, int32_t maxIter,
                              double* diffusiveFlux, double* Ynew, double info[3]);

/* QHDFoam pressure equation; same argument meaning as qgd_qhd_pressure (QHDpEqn.H L35-47) */
int orc_qhd_pressure(void* mesh, const double* phiu, const double* phiwo, const double* taubyrhof, const int32_t* patchKind,
                     const double* pb, const double* gradb, double tolerance, double relTol, int32_t maxIter, int32_t pRefCell,
                     double pRefValue, double* p, double* phi, double info[3]);

/* QHDFoam case (explicit branch of QHDFoam.C L83-139); same layout and meaning as qgd_qhd_options of include/qgd_amd.h */
typedef struct orc_qhd_options {
    int32_t stencil, implicitDiffusion, tauModel, pRefCell, pMaxIter, precond;
    double rho0, mu, Pr, beta, g[3], deltaT, Tau, aQGD, UQHD, T0, Gr, pTol, pRelTol, pRefValue;
    double implicitTol; int32_t implicitMaxIter;
    int32_t fluxSchemeU, fluxSchemeT, pad_;
} orc_qhd_options;
void* orc_qhd_case_create(void* mesh, const orc_qhd_options* opt);
void orc_qhd_case_free(void* c);
int orc_qhd_case_set_bc(void* c, int32_t patch, int32_t bcU, const double* valueU, int32_t bcT, double valueT, int32_t bcP, double valueP);
int orc_qhd_case_set_fields(void* c, const double* U, const double* T, const double* p);
int orc_qhd_case_step(void* c, int32_t nSteps);
/* the same step as phases for cell-range shards; protocol, control slots and message kinds as qgd_qhd_case_step_phase */
int orc_qhd_case_step_phase(void* c, int phase);
int orc_qhd_case_control(void* c, double* buf16, int set);
int orc_qhd_case_set_reference(void* c, int needRef, int localRefCell);
int orc_qhd_case_halo_count(void* c, int side, int kind, int64_t* send, int64_t* recv);
int orc_qhd_case_halo_pack(void* c, int side, int kind, double* buf);
int orc_qhd_case_halo_unpack(void* c, int side, int kind, const double* buf);
int orc_qhd_case_get_field(void* c, const char* name, double* out, int64_t n);
int orc_qhd_case_info(void* c, double info[6]);

void* orc_case_create(void* mesh, const orc_case_options* opt);
void orc_case_free(void* c);
int orc_case_set_bc(void* c, int32_t patch, int32_t bcU, const double* valueU,
                    int32_t bcT, double valueT, int32_t bcP, double valueP);
/* non-uniform alphaQGD / ScQGD (cell + patch values; NULL = uniform), before orc_case_set_fields */
int orc_case_set_qgd_coeffs(void* c, const double* alphaQGD, const double* alphaQGDb, const double* ScQGD, const double* ScQGDb);
int orc_case_set_fields(void* c, const double* U, const double* T, const double* p);
int orc_case_update_fluxes(void* c);
int orc_case_step(void* c, int32_t nSteps);
int orc_case_get_field(void* c, const char* name, double* out, int64_t n);
int orc_case_info(void* c, double info[6]);
int orc_case_halo_count(void* c, int side, int64_t* count);
int orc_case_halo_recv_count(void* c, int side, int64_t* count);
int orc_case_halo_pack(void* c, int side, double* sendBuf);
int orc_case_halo_unpack(void* c, int side, const double* recvBuf);
int orc_case_step_phase(void* c, int phase);   /* 0, 1, 2; 5 + 6 = the two halves of 0; implicitDiffusion on shards: 20..30, 35 as qgd_case_step_phase */
/* the message between phases 5 and 6 (qgd_case_mid_*): mid-step patch pressure + gradient of the boundary-layer cells' patch faces */
/* nSteps of the explicit branch with the flux assembly fused into one vertex pass + one face pass (same arithmetic as orc_case_step,
 * seven face fields stored instead of ~50): the "fused CPU" baseline of bench.py.  Returns 1 when the case is outside its scope
 * (3-D GaussVolPoint, quadrilateral faces, explicit, fixed deltaT, no qgdFlux patch, unsharded). */
int orc_case_step_fused(void* c, int32_t nSteps);
/* cells of the leastSquares stencil of an internal face, in the reference's order; returns their number (< 0: no such stencil) */
int orc_mesh_lsq_stencil(void* mesh, int32_t face, int32_t* cells, int32_t cap);
int orc_case_mid_exchange_needed(void* c);
int orc_case_mid_halo_count(void* c, int side, int64_t* send, int64_t* recv);
int orc_case_mid_halo_pack(void* c, int side, double* buf);
int orc_case_mid_halo_unpack(void* c, int side, const double* buf);
int orc_case_implicit_control(void* c, double* buf68, int set);
int orc_case_implicit_halo_count(void* c, int side, int kind, int64_t* send, int64_t* recv);
int orc_case_implicit_halo_pack(void* c, int side, int kind, double* buf);
int orc_case_implicit_halo_unpack(void* c, int side, int kind, const double* buf);
/* STREAM triad on the calling core (host-bandwidth yardstick for bench.py's cpu_baseline) */
void orc_stream_triad(double* a, const double* b, const double* c, double s, int64_t n, int32_t reps);
int orc_case_reduction(void* c, double* buf2, int set);

#ifdef __cplusplus
}
#endif
#endif
