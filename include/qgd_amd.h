/*
 * qgd_amd.h -- C-ABI of the MI355X-native QGD face-flux path.
 *
 * This is the drop-in boundary for the per-timestep face-flux assembly of
 * QGDFoam (unicfdlab/QGDsolver).  Plain pointers and sizes only; no C++ or
 * torch types cross this boundary.  Every entry returns an int status
 * (QGD_OK == 0, negative == error) and never throws.
 *
 * Citations are to the Doxygen listings of the reference:
 *   /root/reference/docs/html/<name>_source.html, listing lines L..
 * (physical HTML line = listing line + 101).
 *
 * Reference interface each group replaces:
 *   qgd_fvsc_*            fvsc::grad / fvsc::div free functions
 *                         [fvsc_8H_source.html L46-68, fvsc_8C_source.html L87-167]
 *                         and the four fvscStencil virtuals Grad/Grad/Div/Div
 *                         [fvscStencil_8H_source.html L105-130]
 *   qgd_stencil_lookup    fvscStencil::New / lookupOrNew run-time selection by word
 *                         [fvscStencil_8C_source.html L59-118], with the scheme
 *                         checks of fvscOpName [fvsc_8C_source.html L47-85]
 *   qgd_case_update_fluxes  QGDFoam/updateFields.H + QGDFoam/updateFluxes.H
 *                         [QGDFoam_2updateFields_8H_source.html L45-80,
 *                          QGDFoam_2updateFluxes_8H_source.html L41-139]
 *   qgd_case_step         one pass of the QGDFoam while-loop body
 *                         [QGDFoam_8C_source.html L90-163]
 *   qgd_case_get_field    QGDThermo accessors tauQGDf/hQGDf/tauQGD/hQGD/muQGD/
 *                         alphauQGD/c/p/rho/mu [QGDThermo_8H_source.html L99-135]
 *                         and the registered face fields the qgdFlux BC reads by
 *                         name ("phiwStar", "tauQGDf")
 *                         [qgdFluxFvPatchScalarField_8C_source.html L166-192]
 *
 * Layout conventions (same as OpenFOAM's):
 *   - labels are int32, scalars are fp64
 *   - vectors are 3 contiguous doubles, tensors 9 row-major doubles,
 *     T[3*i+j]; a face gradient of a vector is out[3*i+j] = d_i U_j
 *     [leastSquaresStencil_8C_source.html L155-165]
 *   - a surface field is nFaces records: internal faces 0..nInternalFaces-1
 *     followed by the boundary faces patch by patch in mesh order.  Faces of
 *     `empty` patches keep a (zero) slot so that a record is addressed by its
 *     global face label (OpenFOAM gives those patches size 0).
 *   - a boundary ("patch") field is nFaces-nInternalFaces records addressed by
 *     (global face label - nInternalFaces).
 */
#ifndef QGD_AMD_H
#define QGD_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes --------------------------------------------------------- */
enum {
    QGD_OK = 0,
    QGD_ERR_INVALID = -1,     /* bad argument / inconsistent sizes               */
    QGD_ERR_NO_DEVICE = -2,   /* no HIP device: the product path has NO CPU fallback */
    QGD_ERR_HIP = -3,         /* a HIP runtime call failed (see qgd_last_error)  */
    QGD_ERR_SCHEME = -4,      /* scheme rejected, e.g. leastSquares on a 3-D mesh
                                 [fvsc_8C_source.html L60-63]                     */
    QGD_ERR_UNKNOWN_NAME = -5,/* unknown stencil / field / model word            */
    QGD_ERR_NOT_IMPLEMENTED = -6 /* mirrors notImplemented(...) of the base class
                                 [fvscStencil_8H_source.html L105-130]           */
};

/* ---- patch (mesh boundary) types: the fvPatch classes the path tests for -- */
enum {
    QGD_PATCH_GENERIC = 0,        /* patch / wall: ordinary                      */
    QGD_PATCH_EMPTY = 1,          /* emptyFvPatch (2-D / 1-D cases)              */
    QGD_PATCH_SYMMETRYPLANE = 2,  /* symmetryPlaneFvPatch                        */
    QGD_PATCH_SYMMETRY = 3,       /* symmetryFvPatch                             */
    QGD_PATCH_WEDGE = 4,          /* wedgeFvPatch                                */
    QGD_PATCH_CYCLIC = 5,         /* coupled, not processor                      */
    QGD_PATCH_HALO = 6            /* cut plane of a cell-range shard (plays the
                                     role of processorFvPatch): its faces are
                                     never integrated, ghost cells behind it are
                                     refreshed by the halo exchange              */
};

/* ---- boundary-condition kinds of the QGDFoam case ------------------------- */
enum {
    QGD_BC_ZEROGRADIENT = 0,
    QGD_BC_FIXEDVALUE = 1,
    QGD_BC_SLIP = 2,       /* basicSymmetry: U only (also used on symmetryPlane)  */
    QGD_BC_QGDFLUX = 3,    /* p only: qgdFluxFvPatchScalarField
                              [qgdFluxFvPatchScalarField_8C_source.html L159-197] */
    QGD_BC_NONE = 4,       /* empty / halo patches                                */
    QGD_BC_QHDFLUX = 5     /* p of the QHD case only: qhdFlux fed by the registered flux, gradient =
                              -(phiw/tauQGDf*rhof/|Sf|) [qhdFluxFvPatchScalarField_8C_source.html L193-203], as in the
                              QHD solvers that register "phiwStar" (mulesQHDFoam/createFields.H); QHDFoam itself does not,
                              there the patch keeps the gradient of its file = QGD_BC_QGDFLUX with that value */
};

/* ---- fvsc stencil words [fvsc_8C_source.html L60-65] ---------------------- */
enum {
    QGD_FVSC_REDUCED = 0,
    QGD_FVSC_LEASTSQUARES = 1,
    QGD_FVSC_GAUSSVOLPOINT = 2
};

/* ---- opaque handles -------------------------------------------------------- */
typedef struct qgd_mesh_s* qgd_mesh_t;     /* host polyMesh + derived geometry  */
typedef struct qgd_device_s* qgd_device_t; /* device-resident mesh + stencils   */
typedef struct qgd_case_s* qgd_case_t;     /* a QGDFoam case on one device      */

/* ---- library ---------------------------------------------------------------- */
/* Bumped whenever an options struct grows or an entry changes its meaning (3: round 3). */
#define QGD_ABI_VERSION 7
const char* qgd_version(void);
/* sizes[0..2] = sizeof(qgd_case_options), sizeof(qgd_qhd_options), sizeof(qgd_poisson_control) as THIS library was built,
 * sizes[3] = its QGD_ABI_VERSION: a host compiled against another header compares before it passes a struct (the structs
 * carry no size member; the library copies sizeof(its own) bytes). */
int qgd_struct_sizes(int64_t sizes[4]);
/* Last error text of the calling thread (never NULL). */
const char* qgd_last_error(void);
/* Number of visible HIP devices (0 when there is none; never fails). */
int qgd_device_count(void);

/* ---- mesh (host side, plain C++; no device needed) ------------------------ */

/* polyMesh arrays exactly as constant/polyMesh/{points,faces,owner,neighbour,
 * boundary}: faces in upper-triangular order, boundary faces grouped per patch.
 * Arrays are copied.  Geometry (Sf, Cf, C, V) is computed with OpenFOAM's
 * face/cell decomposition rules unless supplied via qgd_mesh_set_geometry. */
int qgd_mesh_create(int32_t nPoints, const double* points,
                    int32_t nFaces, const int32_t* faceOffsets /*nFaces+1*/,
                    const int32_t* facePoints,
                    int32_t nInternalFaces, const int32_t* owner /*nFaces*/,
                    const int32_t* neighbour /*nInternalFaces*/,
                    int32_t nCells,
                    int32_t nPatches, const int32_t* patchStart,
                    const int32_t* patchSize, const int32_t* patchType,
                    qgd_mesh_t* out);

/* Structured box in blockMesh numbering: cell i + nx*(j + ny*k), internal faces
 * sorted by (owner, neighbour), six patches xMin,xMax,yMin,yMax,zMin,zMax.
 * patchTypes[6] gives the QGD_PATCH_* of each.  kLo/kHi select a k-slab
 * [kLo,kHi) of a taller box of nzGlobal cells (kLo=0,kHi=nz for the whole box);
 * z-coordinates stay those of the global box so a shard is bit-identical to the
 * same cells of the global mesh. */
int qgd_mesh_box(int32_t nx, int32_t ny, int32_t nzGlobal,
                 int32_t kLo, int32_t kHi,
                 const double lo[3], const double hi[3],
                 const int32_t patchTypes[6],
                 qgd_mesh_t* out);

/* Forward-facing-step planform (one cell thick in z): nx*ny cells with the
 * block [ixStep,nx) x [0,iyStep) removed.  Patches: inlet, outlet, bottom,
 * top, step (walls of the obstacle), frontAndBack (empty). */
int qgd_mesh_forward_step(int32_t nx, int32_t ny, int32_t ixStep, int32_t iyStep,
                          double lx, double ly, double lz, qgd_mesh_t* out);

/* Move every point by jitter*h*uniform(-1,1) per coordinate (boundary points
 * slide in their plane only), then recompute geometry.  Test helper for
 * non-orthogonal hexahedra. */
int qgd_mesh_jitter(qgd_mesh_t m, double amplitude, uint64_t seed);
/* Split every `stride`-th internal and boundary quad into two triangles
 * (cells stay closed polyhedra): exercises the triangle branch
 * [GaussVolPointBase3D_8C_source.html L161-318]. */
int qgd_mesh_split_quads(qgd_mesh_t m, int32_t stride);
/* Insert a mid-edge vertex on edge 0 of every `stride`-th internal quad; every face sharing that edge gains the vertex
 * (pentagons, hexagons): exercises the "other faces" fallback to nf*snGrad
 * [GaussVolPointBase3D_8C_source.html L759-768].  Test helper. */
int qgd_mesh_split_edges(qgd_mesh_t m, int32_t stride);

/* ---- renumbering and cell-range sharding of any polyMesh ("ranges after a bandwidth-reducing
 * renumbering"; stands in for OpenFOAM's renumberMesh / decomposePar, which are not part of the reference tree) ---- */
/* Relabel the cells in place, newOfOld[old cell] = new cell (a permutation).  Internal faces are re-oriented and
 * re-sorted into upper-triangular order, boundary faces keep their labels, points are relabelled to follow their
 * cells (by the lowest new cell label using them).  faceNewOfOld (nFaces, may be NULL)
 * receives each old face's new label, or -1-label when the face was reversed (fluxes change sign). */
int qgd_mesh_renumber(qgd_mesh_t m, const int32_t* newOfOld, int32_t* faceNewOfOld);
/* newOfOld (nCells) of a reverse Cuthill-McKee ordering of the face-neighbour graph. */
int qgd_mesh_rcm_order(qgd_mesh_t m, int32_t* newOfOld);
/* newOfOld (nCells) of the Morton (Z-curve) order of the cell centres: spatial neighbours stay close in memory at
 * every scale (cache locality on the device); also a reasonable order to cut cell ranges from. */
int qgd_mesh_morton_order(qgd_mesh_t m, int32_t* newOfOld);
/* The shard of `rank` when the cells of `global` are cut into the contiguous ranges cellStart[r]..cellStart[r+1]
 * (cellStart[0]=0, cellStart[nRanks]=nCells): owned cells + one vertex-connected ghost layer (what
 * volPointInterpolation and the leastSquares stencil reach, extendedFaceStencilFindNeighbours_8C_source.html L55-80),
 * all faces of those cells, the global patches (same indices, possibly empty) followed by one QGD_PATCH_HALO patch,
 * and one halo slot per neighbouring rank.  qgd_mesh_get names of a shard: "cellGlobal","faceGlobal" (-1-label when
 * reversed),"pointGlobal","haloPeer","haloGhost<slot>","haloSend<slot>". */
int qgd_mesh_shard(qgd_mesh_t global, int32_t nRanks, const int32_t* cellStart, int32_t rank, qgd_mesh_t* out);
/* Translational cyclic patch pairs served by ghost cells.  The reference's stencils reach across a coupled patch through
 * patchNeighbourField and the true neighbour centre [GaussVolPointBase3D_8C_source.html L398-415, L588, L688, L783-794;
 * extendedFaceStencilScalarGrad_8C_source.html L90-101]; here the two halves of each pair are glued: `out` is the mesh followed by one
 * vertex-connected layer of translated copies of its own cells behind every half (diagonal copies at the edges / corners where several pairs
 * meet), the pairs' faces turned into internal faces between a real cell and a copy, the cyclic patches left empty (same indices), a
 * trailing QGD_PATCH_HALO patch, and halo slots whose messages stay on the rank: qgd_mesh_get "haloSelf" names, per slot, the slot whose
 * packed message it unpacks.  pairs = 2 * nPairs patch indices {A, B} with face i of A = face i of B shifted by one vector (checked:
 * rotational pairs are refused with QGD_ERR_NOT_IMPLEMENTED); nPairs = 0 pairs consecutive cyclic patches.  Real cells, points and patch
 * faces keep their labels ("cellGlobal" = the original of every cell).  A case on such a mesh steps with plain qgd_case_step (explicit branch):
 * the library refreshes the copies from their originals after every step, on the case's stream. */
int qgd_mesh_unroll_cyclic(qgd_mesh_t mesh, int32_t nPairs, const int32_t* pairs, qgd_mesh_t* out);
/* number of halo slots (neighbouring shards) of a mesh; 0 when unsharded, 2 for a qgd_mesh_box slab */
int qgd_mesh_halo_slots(qgd_mesh_t m, int32_t* nSlots);

/* The faceSet "degenerateStencilFaces" of the leastSquares stencil [leastSquaresStencil_8C_source.html L58-133]: faces the user
 * wants treated like the ones whose weight matrix is degenerate, i.e. the face gradient falls back to nf * snGrad
 * [extendedFaceStencilScalarGrad_8C_source.html L76-83].  Internal face labels count (the listing keeps boundary ones only on
 * processor patches); the list follows the faces through qgd_mesh_renumber / qgd_mesh_shard.  Call before qgd_device_create.
 * qgd_mesh_get name: "degenerateFaces". */
int qgd_mesh_set_degenerate_faces(qgd_mesh_t m, int32_t n, const int32_t* faces);
int qgd_mesh_set_geometry(qgd_mesh_t m, const double* Sf, const double* Cf,
                          const double* C, const double* V);
int qgd_mesh_free(qgd_mesh_t m);

/* sizes[0..6] = nPoints, nFaces, nInternalFaces, nCells, nPatches,
 *               nFacePoints (sum of face sizes), nGeometricD */
int qgd_mesh_sizes(qgd_mesh_t m, int64_t sizes[7]);
/* Copy out a named array: "points","faceOffsets","facePoints","owner",
 * "neighbour","patchStart","patchSize","patchType" (int32 / double as natural),
 * "Sf","magSf","Cf","C","V","weights","deltaCoeffs","nonOrthDeltaCoeffs".
 * outBytes < 0 asks for the size: *(int64_t*)out receives the array's length in bytes. */
int qgd_mesh_get(qgd_mesh_t m, const char* name, void* out, int64_t outBytes);

/* ---- device mesh + fvsc operators ----------------------------------------- */

/* Upload the mesh to HIP device `deviceId` and build the static stencil data.
 * Fails with QGD_ERR_NO_DEVICE when no GPU is present. */
int qgd_device_create(qgd_mesh_t m, int deviceId, qgd_device_t* out);
/* The same with the caller's choice about the block tables of QGDFoam's fused explicit step (qgd_case_fused_info; 3-D meshes only: ~30 KB per
 * block of 128 cells on the device and seconds of host set-up at tens of millions of cells), instead of the QGD_FUSED environment variable:
 *   QGD_DEVICE_NO_FUSED_TABLES   do not build them: a device whose cases never run the fused step (QHDFoam, implicitDiffusion, adjustTimeStep
 *                                with QGD_FUSED_ADJUST=0, reduced / leastSquares stencils) -- every case then steps with the separate kernels;
 *   QGD_DEVICE_FUSED_ANY_BLOCKS  build them and use them whatever the blocks look like (the default leaves a mesh whose blocks average under
 *                                88 cells to the separate kernels, which are faster there).
 * flags = 0 is qgd_device_create. */
#define QGD_DEVICE_NO_FUSED_TABLES 1
#define QGD_DEVICE_FUSED_ANY_BLOCKS 2
int qgd_device_create_with(qgd_mesh_t m, int deviceId, int32_t flags, qgd_device_t* out);
/* Ownership (fvscStencil_8C_source.html L57, L104-117: the reference's stencils live in the mesh's registry and die with it):
 * every case (qgd_case_t, qgd_qhd_case_t) keeps a pointer to the device it was created on and uses its stream until it is
 * freed, so free the cases BEFORE their device: qgd_device_free returns QGD_ERR_INVALID (and frees nothing) while cases created on
 * the device are still open.  The Python mirror enforces the order (fvsc.Device.close frees the cases still open on it first). */
int qgd_device_free(qgd_device_t d);

/* Run-time selection by word, like fvscStencil::New: "reduced",
 * "leastSquares", "leastSquaresOpt" (alias of leastSquares' serial formula),
 * "GaussVolPoint".  Applies the checks of fvscOpName: leastSquares* is refused
 * on 3-D meshes (QGD_ERR_SCHEME); unknown word -> QGD_ERR_UNKNOWN_NAME. */
int qgd_stencil_lookup(qgd_device_t d, const char* word, int* stencilId);

/* The four fvscStencil operators.  cell = nCells*ncomp, bnd = boundary (patch)
 * values nBoundaryFaces*ncomp, out = nFaces*ncompOut, all HOST pointers (the
 * OpenFOAM adapter hands over Field<Type>::cdata()).
 *   grad_s : ncomp 1 -> 3      grad_v : ncomp 3 -> 9
 *   div_v  : ncomp 3 -> 1      div_t  : ncomp 9 -> 3
 * The patch surface-normal gradient the boundary-face rules need is
 * deltaCoeffs*(bnd - internal), fvPatchField::snGrad (L0). */
int qgd_fvsc_grad_s(qgd_device_t d, int stencilId, const double* cell,
                    const double* bnd, double* out);
int qgd_fvsc_grad_v(qgd_device_t d, int stencilId, const double* cell,
                    const double* bnd, double* out);
int qgd_fvsc_div_v(qgd_device_t d, int stencilId, const double* cell,
                   const double* bnd, double* out);
int qgd_fvsc_div_t(qgd_device_t d, int stencilId, const double* cell,
                   const double* bnd, double* out);

/* The same four operators, qgdInterpolate, the QHD flux block and the species block on DEVICE pointers (buffers of
 * qgd_device_alloc or any allocation of this HIP device): same layouts, nothing staged, nothing crosses PCIe, stream-ordered on
 * the device handle's own stream -- qgd_device_sync waits.  For hosts whose fields already live in HBM (level 1 of
 * INTEGRATION.md without the 80 ms of PCIe per step the host-pointer entries cost at 8 M cells).  The struct members of
 * qgd_qhd_inputs / qgd_qhd_outputs are device pointers here.  qgd_device_copy: a plain synchronous copy for callers that hold
 * no HIP runtime of their own (toDevice != 0: host -> device). */
int qgd_fvsc_grad_s_dev(qgd_device_t d, int stencilId, const double* cellDev, const double* bndDev, double* outDev);
int qgd_fvsc_grad_v_dev(qgd_device_t d, int stencilId, const double* cellDev, const double* bndDev, double* outDev);
int qgd_fvsc_div_v_dev(qgd_device_t d, int stencilId, const double* cellDev, const double* bndDev, double* outDev);
int qgd_fvsc_div_t_dev(qgd_device_t d, int stencilId, const double* cellDev, const double* bndDev, double* outDev);
int qgd_interpolate_dev(qgd_device_t d, int32_t ncomp, const double* cellDev, const double* bndDev, double* outDev);
int qgd_device_sync(qgd_device_t d);
int qgd_device_copy(qgd_device_t d, void* dst, const void* src, int64_t bytes, int toDevice);
/* The leastSquares stencil of internal face `face` as the device holds it: the cells in the order they are summed, which is the order
 * the reference builds [extendedFaceStencilFindNeighbours_8C_source.html L48-84: the face's points in order, each point's cells in
 * pointCells() order, first occurrence kept].  *count = their number (the first `cap` are written).  Diagnostic entry (a few small
 * copies per call): tests/test_ref_expr_gpu.py compares it with the order executed from the listing text.  QGD_ERR_SCHEME on 3-D meshes. */
int qgd_device_lsq_stencil(qgd_device_t d, int32_t face, int32_t* cells, int32_t cap, int32_t* count);

/* Where the time of the last qgd_fvsc_* call on this device went: ms[0] = host -> device copies, ms[1] = kernels (HIP events),
 * ms[2] = device -> host copy.  Pageable caller memory moves through two pinned staging chunks in a double-buffered
 * pipeline; buffers of the persistent per-device workspace are reused from call to call (no allocation per call). */
int qgd_device_op_times(qgd_device_t d, double ms[3]);

/* How the internal faces of a 3-D mesh go through the GaussVolPoint flux kernel: info[0] = faces per tile (workgroup) of the
 * LDS-staged kernel, 0 when every face goes through the gather kernel (not a 3-D mesh, QGD_FTILE=0); info[1] = tiles;
 * info[2] = tiles left to the gather kernel (more distinct cell / vertex records than the staged kernel brings in);
 * info[3] = LDS bytes per workgroup.  Both kernels do the arithmetic of GaussVolPointBase3D_8C_source.html L346-389,
 * L488-513 + QGDFoam_2updateFluxes_8H_source.html L41-139 in the same order: results are bit-identical. */
int qgd_device_face_tiles(qgd_device_t d, int64_t info[4]);
/* The block tables of the fused explicit step on this device (all 0 when they were not built or not accepted; qgd_case_fused_info says whether
 * a case uses them): info[0] = blocks, [1] = of which the leading boundary-layer blocks of a shard, [2] = distinct templates of the blocks'
 * local topology (per-face list positions, per-cell face entries, per-vertex cell positions: blocks that are alike share one, so on a
 * structured mesh those tables stay in L2 instead of being streamed per block per step), [3] = LDS bytes per workgroup, [4] = bytes of a
 * block's own lists (labels, weights, counts), [5] = bytes of one template, [6] = host milliseconds the builder took, [7] = the lattice brick
 * the blocks were cut from, bx | by << 8 | bz << 16 (0: count-based runs of the Morton order). */
int qgd_device_fused_blocks(qgd_device_t d, int64_t info[8]);

/* qgdInterpolate / linearInterpolate [QGDInterpolate_8H_source.html L38-67]: face = w*(phi_O - phi_N) + phi_N, patch faces
 * take the patch value.  cell nCells*ncomp, bnd nBoundaryFaces*ncomp, out nFaces*ncomp (HOST pointers), ncomp in 1..9. */
int qgd_interpolate(qgd_device_t d, int32_t ncomp, const double* cell, const double* bnd, double* out);
/* qgdFlux [QGDInterpolate_8H_source.html L76-118] without a divSchemes entry: out = flux*psif (flux nFaces, psif and out
 * nFaces*ncomp).  Pure host arithmetic (one multiply per value): kept so call sites read like the reference. */
int qgd_flux(qgd_device_t d, int32_t ncomp, const double* flux, const double* psif, double* out);
/* qgdFlux WITH a divSchemes entry `Gauss upwind` for the flux's name [QGDInterpolate_8H_source.html L86-104 -> fvc::flux]:
 * out = flux * (pos0(flux)*(psi_O - psi_N) + psi_N) on internal faces, flux * patch value on patch faces (`Gauss linear` is qgd_flux
 * on the linear psif: the same numbers).  flux nFaces, cell nCells*ncomp, bnd nBoundaryFaces*ncomp, out nFaces*ncomp, HOST pointers. */
int qgd_flux_upwind(qgd_device_t d, int32_t ncomp, const double* flux, const double* cell, const double* bnd, double* out);
/* Static QGD length scales of the mesh [QGDCoeffs_8C_source.html L298-376]: name = "hQGD" (nCells), "hQGDf" (nFaces),
 * "hQGD.boundary" (nBoundaryFaces).  The QHD tau closures (constTau, HbyUQHD, T0byGr, H2bynuQHD) are one line each on
 * top of these and qgd_interpolate; see qgdsolver_amd/qhdfoam.py. */
int qgd_device_get(qgd_device_t d, const char* name, double* out, int64_t outDoubles);

/* ---- QHDFoam flux assembly (stateless, host pointers) -------------------- */
/* The face-flux parts of the QHDFoam step: QHDFoam/updateFields.H L36-73 (gradients, interpolations, BdFrc),
 * QHDFoam/updateFluxes.H L33-38 (phiu, phiwo, taubyrhof), QHDUEqn.H L36-43 (gradPf, Wf, phiUf) and QHDTEqn.H
 * L65-66 (phiTf, phiTauTReg).  The implicit pressure Poisson solve between them (QHDpEqn.H L35-47) stays with the
 * caller: call once with p = phi = NULL for the first group, again with p and phi for the rest (or once with
 * everything).  cell arrays: nCells*ncomp, patch arrays: nBoundaryFaces*ncomp, face arrays: nFaces*ncomp. */
typedef struct qgd_qhd_inputs {
    const double* U;   const double* Ub;     /* 3 */
    const double* T;   const double* Tb;     /* 1 */
    const double* p;   const double* pb;     /* 1, nullable */
    const double* rho; const double* rhob;   /* 1 */
    const double* tauQGDf;                   /* nFaces: thermo.tauQGDf() of the chosen QGDCoeffs model */
    const double* phi;                       /* nFaces, nullable: phiu - phiwo + pEqn.flux() */
    double beta;                             /* thermal expansion coefficient */
    double g[3];                             /* gravitational acceleration */
} qgd_qhd_inputs;
typedef struct qgd_qhd_outputs {             /* every pointer may be NULL */
    double* gradUf;      /* 9: fvsc::grad(U) */
    double* gradTf;      /* 3: fvsc::grad(T) */
    double* phiu;        /* 1 */
    double* phiwo;       /* 1 */
    double* taubyrhof;   /* 1 */
    double* gradPf;      /* 3: needs p */
    double* Wf;          /* 3: needs p */
    double* phiUf;       /* 3: needs p and phi */
    double* phiTf;       /* 1: needs phi */
    double* phiTauTReg;  /* 1 */
} qgd_qhd_outputs;
int qgd_qhd_fluxes(qgd_device_t d, int stencilId, const qgd_qhd_inputs* in, qgd_qhd_outputs* out);
int qgd_qhd_fluxes_dev(qgd_device_t d, int stencilId, const qgd_qhd_inputs* in, qgd_qhd_outputs* out);   /* device pointers */

/* Species flux block -- reactingLagrangianQGDFoam_2updateFluxes_8H_source.html L117-132, one species:
 *     gradYf = fvsc::grad(Y);  phiJmY = qgdFlux(phiJm, Y, Yf);  dydtflux = -phi*tauQGDf*(Uf & gradYf);
 *     phiJmY += dydtflux;  diffusiveFlux = dydtflux;
 * Y[nCells], Yb[nBoundaryFaces], U[3 nCells], Ub[3 nBoundaryFaces] (patch values after their BCs), phiJm, phi, tauQGDf [nFaces]
 * as the QGDFoam block produced them (qgd_case_get_field "phiJm","phi","tauQGDf").  Out: phiJmY, diffusiveFlux [nFaces],
 * gradYf [3 nFaces] (may be NULL).  What QGDYEqn_8H_source.html L35-93 then does with them stays with the caller. */
int qgd_species_flux(qgd_device_t d, int stencilId, const double* Y, const double* Yb, const double* U, const double* Ub,
                     const double* phiJm, const double* phi, const double* tauQGDf, double* phiJmY, double* diffusiveFlux,
                     double* gradYf);
int qgd_species_flux_dev(qgd_device_t d, int stencilId, const double* Y, const double* Yb, const double* U, const double* Ub,
                         const double* phiJm, const double* phi, const double* tauQGDf, double* phiJmY, double* diffusiveFlux,
                         double* gradYf);
/* The species equation itself, one species, explicit branch -- QGDYEqn_8H_source.html L44-45, L67-86:
 *     solve(fvm::ddt(rho,Yi) + fvc::div(phiJmYi) - fvc::laplacian(muf/ScNumbers[i],Yi) == combustion->R(Yi) + parcels.SYi(i, Yi));
 *     diffusiveFlux[i] += (muf/ScNumbers[i]) * fvc::snGrad(Yi.oldTime()) * mesh.magSf();   Yi.max(0.0);
 * with the right-hand side handed in as ONE explicit source field Su [nCells] (kg/m^3/s; NULL: none) -- combustion and parcels stay with
 * the caller.  Y [nCells], Yb [nBoundaryFaces] (old time level, patch values after their BCs), rhoOld, rho [nCells] (before / after
 * QGDRhoEqn), phiJmY [nFaces] (qgd_species_flux), muf [nFaces] (qgd_case_get_field "muf"), Sc = ScNumbers[i], deltaT.
 * In/out: diffusiveFlux [nFaces].  Out: Ynew [nCells].  Euler ddt, fvc::div = surfaceIntegrate in ascending face label, Gauss laplacian
 * with the uncorrected snGrad (L0).  The inert species (L65 / L83, L90-91: Y_inert = 1 - sum, diffusiveFlux_inert -= diffusiveFlux_i)
 * is two axpys over these arrays and stays with the caller (qgdsolver_amd.qgdfoam.QGDYEqn does it). */
int qgd_species_step(qgd_device_t d, const double* Y, const double* Yb, const double* rhoOld, const double* rho, const double* phiJmY,
                     const double* muf, double Sc, double deltaT, const double* Su, double* diffusiveFlux, double* Ynew);
int qgd_species_step_dev(qgd_device_t d, const double* Y, const double* Yb, const double* rhoOld, const double* rho, const double* phiJmY,
                         const double* muf, double Sc, double deltaT, const double* Su, double* diffusiveFlux, double* Ynew);   /* device pointers */   /* device pointers */

/* The same equation, implicitDiffusion branch -- QGDYEqn_8H_source.html L47-66:
 *     fvScalarMatrix YEqn(fvm::ddt(rho,Yi) + fvc::div(phiJmYi) - fvm::laplacian(muf/ScNumbers[i],Yi) == combustion->R(Yi) + parcels.SYi(i, Yi));
 *     YEqn.solve();   diffusiveFlux[i] += YEqn.flux();   Yi.max(0.0);
 * Arguments as above plus fixedValueFace [nBoundaryFaces] (1: the face belongs to a fixedValue patch of Yi, value Yb: diagonal and source
 * coefficients |Sf| delta (muf/Sc); 0: zeroGradient, no contribution; NULL: all zeroGradient), and the controls fvSolution gives the
 * solver of Yi: tolerance on OpenFOAM's normalised residual, maxIter.  YEqn.flux() is the matrix's own face flux (L0 fvMatrix::flux) of
 * the NEW Yi: -a_f (Y_N - Y_O) inside, -a_b (Y_b - Y_P) on fixedValue faces -- the sign "- fvm::laplacian" gives it, opposite to the
 * explicit branch's +(muf/Sc) snGrad |Sf| of L82; replicated as listed.  info (host) = {iterations, initial, final residual}.  Solved by
 * the library's device-resident solver (Chebyshev iteration, or conjugate gradients with QGD_IMPL_SOLVER=pcg); one device, no shards. */
int qgd_species_step_implicit(qgd_device_t d, const double* Y, const double* Yb, const uint8_t* fixedValueFace, const double* rhoOld,
                              const double* rho, const double* phiJmY, const double* muf, double Sc, double deltaT, const double* Su,
                              double tolerance, int32_t maxIter, double* diffusiveFlux, double* Ynew, double info[3]);
int qgd_species_step_implicit_dev(qgd_device_t d, const double* Y, const double* Yb, const uint8_t* fixedValueFace, const double* rhoOld,
                                  const double* rho, const double* phiJmY, const double* muf, double Sc, double deltaT, const double* Su,
                                  double tolerance, int32_t maxIter, double* diffusiveFlux, double* Ynew, double info[3]);   /* device pointers */

/* QHDFoam's pressure equation -- QHDpEqn_8H_source.html L35-47:
 *     fvScalarMatrix pEqn(fvc::div(phiu) - fvc::div(phiwo) - fvm::laplacian(taubyrhof, p));
 *     pEqn.setReference(pRefCell, pRefValue);  pEqn.solve();  phi = phiu - phiwo + pEqn.flux();
 * assembled and solved on the device: Gauss laplacian with the uncorrected surface-normal gradient (L0), a
 * Jacobi-preconditioned conjugate gradient with reproducible reductions, converged on OpenFOAM's normalised residual
 * sum|b-Ax|/normFactor < tolerance (or < relTol x initial), at most maxIter iterations.
 * phiu, phiwo, taubyrhof: nFaces.  patchKind[nPatches]: QGD_BC_ZEROGRADIENT, QGD_BC_FIXEDVALUE (values pb[nBoundaryFaces])
 * or QGD_BC_QGDFLUX = fixedGradient with gradb[nBoundaryFaces] (what qhdFluxFvPatchScalarField::updateCoeffs leaves in
 * gradient(), qhdFluxFvPatchScalarField_8C_source.html L159-208); constraint patches are skipped.
 * p[nCells]: initial guess in, solution out.  phi[nFaces] out.  info = {iterations, initial residual, final residual}.
 * The reference level is applied only when no patch is fixedValue (fvMatrix::setReference, L0); pRefCell < 0: never. */
typedef struct qgd_poisson_control {
    double tolerance, relTol;
    int32_t maxIter, pRefCell;
    double pRefValue;
} qgd_poisson_control;
int qgd_poisson_control_default(qgd_poisson_control* c);
int qgd_qhd_pressure(qgd_device_t d, const double* phiu, const double* phiwo, const double* taubyrhof,
                     const int32_t* patchKind, const double* pb, const double* gradb, const qgd_poisson_control* ctl,
                     double* p, double* phi, double info[3]);

/* ---- QHDFoam case resident on the device --------------------------------------------------------------------------- */
/* The loop body of QHDFoam [QHDFoam_8C_source.html L83-139], both branches of implicitDiffusion: updateFields.H,
 * updateFluxes.H, the pressure equation QHDpEqn.H L35-47 (conjugate gradients preconditioned by an aggregation multigrid
 * built once: the matrix does not change, QHDFoam never re-corrects its thermo inside the loop), QHDUEqn.H L36-84,
 * QHDTEqn.H L65-91, the reference level of p (L123-130).  Thermo: rhoConst + constTransport (uniform rho0, mu, Pr; the QHD
 * closures leave muQGD = alphauQGD = 0 [T0byGr_8C_source.html L62-72]), laminar; L0 discretisation assumed: Gauss linear
 * gradients, Gauss linear uncorrected laplacians, Euler ddt.  Sharded meshes: see "the QHD case on a cell-range shard" below. */
/* qgdFlux [QGDInterpolate_8H_source.html L76-105]: the flux's own divSchemes entry decides */
enum qgd_flux_scheme {
    QGD_FLUX_LINEAR = 0,   /* no entry (flux*psif) or `Gauss linear` (fvc::flux with linear weights: the same numbers) */
    QGD_FLUX_UPWIND = 1    /* `Gauss upwind`: psi_f = pos0(flux)*(psi_O - psi_N) + psi_N (L0: upwind::weights = pos0(faceFlux),
                              surfaceInterpolationScheme::interpolate), patch faces keep the patch value               */
};

typedef struct qgd_qhd_options {
    int32_t stencil;          /* QGD_FVSC_*                                                                    */
    int32_t implicitDiffusion;/* 0: fvc::laplacian in both equations; 1 (the reference's default, QGDThermo_8C_source.html L70-82):
                                 fvm::laplacian(muf/rhof, U) and fvm::laplacian(Hif, T) [QHDUEqn_8H_source.html L46-65,
                                 QHDTEqn_8H_source.html L69-80]: the four systems as ONE matrix walk (see implicitTol)     */
    int32_t tauModel;         /* QGDCoeffs closure: 0 constTau (Tau), 1 HbyUQHD (aQGD, UQHD), 2 T0byGr (T0, Gr),
                                 3 H2bynuQHD (aQGD; nu = mu/rho0)                                                */
    int32_t pRefCell;         /* fvSolution pRefCell; < 0: no reference level                                  */
    int32_t pMaxIter;
    int32_t precond;          /* 1 aggregation multigrid (default), 0 Jacobi                                    */
    double rho0, mu, Pr, beta, g[3], deltaT;
    double Tau, aQGD, UQHD, T0, Gr;
    double pTol, pRelTol, pRefValue;
    double implicitTol;       /* implicitDiffusion: tolerance (OpenFOAM's normalised residual) of the U and T solves (1e-10) */
    int32_t implicitMaxIter;  /* ... and their iteration limit (1000)                                          */
    int32_t fluxSchemeU;      /* QGD_FLUX_* of qgdFlux(phi,U,Uf) [QHDUEqn_8H_source.html L41], entry `div(phi,U)`          */
    int32_t fluxSchemeT;      /* ... of qgdFlux(phi,T,Tf) [QHDTEqn_8H_source.html L65], entry `div(phi,T)`                 */
    int32_t pad_;
} qgd_qhd_options;
typedef struct qgd_qhd_case_s* qgd_qhd_case_t;
int qgd_qhd_options_default(qgd_qhd_options* opt);
int qgd_qhd_case_create(qgd_device_t d, const qgd_qhd_options* opt, qgd_qhd_case_t* out);
int qgd_qhd_case_free(qgd_qhd_case_t c);
/* per patch: U zeroGradient | fixedValue | slip; T zeroGradient | fixedValue; p zeroGradient | fixedValue (valueP) |
 * QGD_BC_QGDFLUX = fixedGradient with gradient valueP (what qhdFlux is inside QHDFoam) | QGD_BC_QHDFLUX (see the enum) */
int qgd_qhd_case_set_bc(qgd_qhd_case_t c, int32_t patch, int32_t bcU, const double* valueU, int32_t bcT, double valueT,
                        int32_t bcP, double valueP);
int qgd_qhd_case_set_fields(qgd_qhd_case_t c, const double* U, const double* T, const double* p);
int qgd_qhd_case_step(qgd_qhd_case_t c, int32_t nSteps);
/* "U","T","p" (+ ".boundary"), face fields "phi","phiu","phiwo","tauQGDf" */
int qgd_qhd_case_get_field(qgd_qhd_case_t c, const char* name, double* out, int64_t outDoubles);
/* info[0]=time, [1]=deltaT, [2..4]= iterations / initial / final normalised residual of the last pressure solve,
 * [5]=steps, [6]=multigrid levels, [7]=milliseconds of the last pressure solve */
int qgd_qhd_case_info(qgd_qhd_case_t c, double info[8]);
/* QGD_QHD_FUSED=1 (off by default: at 8 M cells the one launch takes what the three kernels take, profiles/r06_ab_qhd_fused_advance.txt):
 * the U and T equations of the explicit branch [QHDUEqn_8H L36-84, QHDTEqn_8H L65-91] -- vertex values of p, face pass 2, explicit Euler
 * update -- run as ONE launch on the cell blocks of QGDFoam's one-launch step (qgd_qhd.hip qhdFusedAdvanceKernel).  info[0] = bit 0: they do
 * (bit 1 is reserved for the part in front of the pressure equation, not built); info[1] = blocks; info[2] = LDS bytes per workgroup;
 * info[3] = 0.  Needs a device with block tables (qgd_device_create, or _with QGD_DEVICE_FUSED_ANY_BLOCKS), 3-D GaussVolPoint, an unsharded
 * mesh, implicitDiffusion false; everything else keeps the separate kernels. */
int qgd_qhd_case_fused_info(qgd_qhd_case_t c, int64_t info[4]);
/* implicitDiffusion: the solve of the last step -- info[0..3] = iterations of Ux, Uy, Uz, T, [4..7] = initial, [8..11] = final
 * normalised residuals (what OpenFOAM prints as "Solving for Ux, Initial residual = ..."), [12] = steps since
 * qgd_qhd_case_set_fields in which a component stopped above implicitTol, [13] = 0 explicit branch | 1 conjugate gradients
 * (QGD_IMPL_SOLVER=pcg) | 2 Chebyshev iteration (default), [14] = steps in which a component ended at the rounding floor of its
 * true residual ABOVE implicitTol (see qgd_case_implicit_info), [15] reserved.  The four systems share |Sf| delta_f up to
 * {nu, nu, nu, Hi} and are solved as four right-hand sides of one matrix walk. */
int qgd_qhd_case_implicit_info(qgd_qhd_case_t c, double info[16]);

/* ---- the QHD case on a cell-range shard (qgd_mesh_box slabs, qgd_mesh_shard) ------------------------------------------------
 * What the reference does through processor patches inside fvm::laplacian / PCG / fvc::grad under MPI
 * [QHDpEqn_8H_source.html L35-47, QHDUEqn_8H_source.html L36-84] becomes, per step, with ONE rank per shard:
 *   phase 0   flux assembly (updateFields.H, updateFluxes.H), p's boundary conditions, fvc::grad(U) of the owned cells, the rows
 *             of the pressure equation of the owned cells, first residual   -> DRAIN PENDING; all-reduce (SUM) control[0..3)
 *   phase 1   normFactor                                        -> all-reduce control[3]
 *   phase 2   first preconditioned residual, search direction   -> DRAIN PENDING; all-reduce control[4]; exchange message kind 2
 *   repeat until qgd_qhd_case_solve_status says done (every rank sees the same flag: it is computed from reduced sums):
 *     phase 3 A d, d.Ad                                         -> all-reduce control[5]
 *     phase 4 x, r, z = M r, |r|, r.z                            -> DRAIN PENDING; all-reduce control[6..8)
 *     phase 5 residual, iteration count, done?, new direction   -> exchange message kind 2
 *   phase 6   p's boundary conditions after the solve           -> exchange message kind 1
 *   phase 7   phi, QHDUEqn.H, QHDTEqn.H, U/T boundary conditions -> all-reduce control[8] (only when p needs a reference level)
 *             implicitDiffusion: phi, the face terms without the laplacians, the right-hand sides of the four systems; then the
 *             solve with ITS OWN 68-double control block (qgd_qhd_case_implicit_control*), phases and reductions exactly those of
 *             "the implicitDiffusion branch on a cell-range shard" below, numbered 10..15 here instead of 22..27, message kind 4 in
 *             place of its kind 3 (no start-value message: the ghost columns start from the state message), until
 *             qgd_qhd_case_implicit_solve_status says done; then
 *   phase 16  (implicitDiffusion) the solution into the records, U/T boundary conditions -> all-reduce control[8] as after phase 7
 *   phase 8   reference level of p                              -> exchange message kind 0
 * DRAIN PENDING = the comm points of the preconditioner, which fall INSIDE phases 0, 2 and 4: after the phase call, and again after
 * every qgd_qhd_case_step_phase(c, 9), ask qgd_qhd_case_pending and perform what it names until it answers 0 (protocol below).  A
 * caller that implements only the table and never drains gets an error status from the next phase ("a collective of the phase in
 * flight is pending"), not a wrong answer.
 * Preconditioner: by DEFAULT the smoothed-aggregation hierarchy SPANS THE RANKS (level 0 distributed, coarse levels replicated;
 * same iteration count as the unsharded solve).  It replicates the global matrix on every rank at set-up, so it is built only up
 * to QGD_MG_DIST_MAX_CELLS cells of the unsharded mesh (default 20 000 000; memory per rank ~ nCells (1 + 2K) doubles on host and
 * device, K = couplings per cell to higher-numbered neighbours, plus a host coarsening of nCells rows).  Above that, with
 * QGD_MG_DIST=0, with the plain-aggregation or double-precision cycle, or for a caller that never performs the set-up all-reduces,
 * every rank preconditions with the hierarchy of its own rows (block Jacobi across the shards: couplings to ghost cells stay in the
 * diagonal; 7 -> 65 / 97 / 140 iterations on 2 / 4 / 8 shards of a 128^3 box) and a line on stderr says so.  Everything else is the
 * unsharded arithmetic.
 * control: 16 device doubles (qgd_qhd_case_control_ptr); message kinds: 0 = the new state, {U,T} per cell (4) and per patch face
 * (4); 1 = p + fvc::grad(U) per cell (1 + 9: a ghost cell's gradient cannot be formed locally, it lacks faces) and p's patch value
 * + gradient per patch face (2); 2 = the search direction per cell (1); 3 = the multigrid iterate per cell (1), see qgd_qhd_case_pending;
 * 4 = what the implicitDiffusion solve's next matrix product reads in the ghost columns, {Ux, Uy, Uz, T} per cell (4).
 * pRefCell is a cell label of the UNSHARDED mesh.  All entries are stream-ordered on the device's stream. */
int qgd_qhd_case_step_phase(qgd_qhd_case_t c, int phase);
/* The hierarchy that spans the ranks (default up to QGD_MG_DIST_MAX_CELLS, see above).  Level 0 stays distributed --
 * each rank smooths its own rows, the ghost entries of the iterate refreshed before every sweep (message kind 3: one double per cell)
 * -- and every level below it is replicated: the ranks all-reduce their shares of the level-1 right-hand side and run the coarse part
 * of the cycle redundantly.  The comm points fall INSIDE phases 0 (set-up, first step only: the global matrix is gathered by two
 * all-reduces), 2 and 4 (the cycle), so after EVERY qgd_qhd_case_step_phase call the caller asks
 *     qgd_qhd_case_pending(c, &action, &ptr, &count)
 * and, while action != 0, performs it and calls qgd_qhd_case_step_phase(c, 9):
 *     action 1  exchange message kind 3 (halo_pack / halo_unpack with kind 3)
 *     action 2  SUM all-reduce of the `count` device doubles at ptr, in place
 *     action 3  MAX all-reduce of the same
 * then goes on with the reductions and messages the table above lists for that phase.  Every rank sees the same sequence.
 * qgd_qhd_case_step_sharded does all of this itself. */
int qgd_qhd_case_pending(qgd_qhd_case_t c, int32_t* action, void** devicePtr, int64_t* count);
/* the control block of the implicitDiffusion solve (68 doubles, control[slot * 4 + component], as qgd_case_implicit_control) and
 * its state: status[0] != 0: every component is done; status[1] = right-hand sides (4) */
int qgd_qhd_case_implicit_control(qgd_qhd_case_t c, double control[68], int set);
int qgd_qhd_case_implicit_control_ptr(qgd_qhd_case_t c, void** devicePtr);
int qgd_qhd_case_implicit_solve_status(qgd_qhd_case_t c, double status[2]);
int qgd_qhd_case_control_ptr(qgd_qhd_case_t c, void** devicePtr);
/* host copy of the control block out (set == 0) or in (set != 0), after everything queued so far: for transports that reduce on
 * the host (MPI_Allreduce of 16 doubles, torch.distributed over gloo) */
int qgd_qhd_case_control(qgd_qhd_case_t c, double control[16], int set);
/* waits for the stream; status = {done (0 no, 1 converged or out of iterations, 2 breakdown), iterations, initial, final residual} */
int qgd_qhd_case_solve_status(qgd_qhd_case_t c, double status[4]);
int qgd_qhd_case_sync(qgd_qhd_case_t c);
/* measurement: `reps` smoothing sweeps of multigrid level 0 (the kernel the pressure solve spends most of its time in) between two
 * HIP events; info = {average ms per sweep, rows, ELL width, bytes per matrix value and vector entry (4: single-precision cycle)} */
int qgd_qhd_case_sweep_time(qgd_qhd_case_t c, int reps, double info[4]);
int qgd_qhd_case_halo_count(qgd_qhd_case_t c, int slot, int kind, int64_t* sendCount, int64_t* recvCount);
int qgd_qhd_case_halo_pack(qgd_qhd_case_t c, int slot, int kind, double* sendBufDevice);
int qgd_qhd_case_halo_unpack(qgd_qhd_case_t c, int slot, int kind, const double* recvBufDevice);

/* ---- QGDFoam case ----------------------------------------------------------- */

typedef struct qgd_case_options {
    int32_t stencil;          /* QGD_FVSC_* : fvSchemes fvsc{default ...;}        */
    int32_t implicitDiffusion;/* 0: the explicit branch [QGDFoam_2updateFluxes_8H_source.html L95-106];
                                 1: the reference's default [QGDThermo_8C_source.html L70-82]: viscous stress and heat
                                 conduction implicit, tauMC / phiSigmaDotU [updateFluxes.H L107-111, QGDUEqn_8H L54-75,
                                 QGDEEqn_8H L53-64]; on shards see "the implicitDiffusion branch on a cell-range shard"   */
    int32_t adjustTimeStep;   /* 1: Courant/deltaT control
                                 [QGDCourantNo_8H_source.html L36-53,
                                  setDeltaT-QGDQHD_8H_source.html L41-61]         */
    int32_t consistentEnergy; /* 0: the explicit energy re-solve as the listing has it, fvm::ddt(rho,e) - fvc::ddt(rhoE)
                                 [QGDEEqn_8H_source.html L67-72]: rho*e advances by the increment of rhoE and never gives
                                 the kinetic energy back (Sod's plateau comes out at p* = 0.330 instead of 0.303);
                                 1: keep e = rhoE/rho - |U|^2/2 of L49 (what the implicit branch's fvc::ddt(rho,e) form,
                                 L57, gives with a zero source): the physically consistent update                  */
    double R;                 /* perfectGas: specific gas constant                */
    double Cv;                /* eConst (Tref = 0, Esref = 0)                     */
    double mu;                /* constTransport: mu                               */
    double Pr;                /* constTransport: Pr                               */
    double ScQGD;             /* constScPrModel1 [constScPrModel1_8C_source.html L58-89] */
    double PrQGD;
    double alphaQGD;          /* uniform alphaQGD (0.5 when the file is absent
                                 [QGDCoeffs_8C_source.html L145-159])             */
    double deltaT;            /* (initial) time step                              */
    double maxCo, maxDeltaT, cTau; /* used when adjustTimeStep                    */
    double implicitTol;       /* implicitDiffusion: tolerance (OpenFOAM's normalised residual) of the U and e solves,
                                 fvSolution's `tolerance` of those fields                                         */
    int32_t implicitMaxIter;  /* ... and their iteration limit                                                    */
    int32_t fluxSchemeU;      /* QGD_FLUX_*: what qgdFlux(phiJm,U,Uf) [QGDFoam_2updateFluxes_8H_source.html L78] does: LINEAR = flux*psif,
                                 the default branch and `divSchemes{div(phiJm,U) Gauss linear;}` (same numbers); UPWIND =
                                 `div(phiJm,U) Gauss upwind;` -> fvc::flux [QGDInterpolate_8H_source.html L86-104]: the upwind cell's
                                 U on internal faces, the patch value on patch faces                                  */
    int32_t fluxSchemeH;      /* the same for qgdFlux(phiJm,H,Hf) [updateFluxes.H L119], entry `div(phiJm,H)`          */
    int32_t pad_;
    int32_t termStencil[4];   /* per-term entries of fvSchemes.fvsc [fvsc_8C_source.html L51-58] for the four face gradients of the step,
                                 in the order grad(U), grad(e), grad(rho), grad(p) [QGDFoam_2updateFluxes_8H_source.html L41-65]:
                                 0 = no entry of its own (the `default` word = `stencil`), else 1 + QGD_FVSC_*.  At most two distinct
                                 stencils per case (QGD_ERR_NOT_IMPLEMENTED beyond); a mixed case walks its faces through the generic
                                 gather kernels (each gradient formed by its own stencil, the flux algebra unchanged), a uniform one
                                 through the fused kernels as before.  GaussVolPoint's re-evaluation of its input's boundary
                                 conditions [GaussVolPointStencil_8C_source.html L73] follows grad(p)'s word.                   */
} qgd_case_options;

int qgd_case_options_default(qgd_case_options* opt);

int qgd_case_create(qgd_device_t d, const qgd_case_options* opt, qgd_case_t* out);
int qgd_case_free(qgd_case_t c);

/* Boundary conditions, per patch.  value: 3 doubles for U, 1 for T and p
 * (uniform fixedValue); ignored for the other kinds. */
int qgd_case_set_bc(qgd_case_t c, int32_t patch, int32_t bcU, const double* valueU,
                    int32_t bcT, double valueT, int32_t bcP, double valueP);

/* Non-uniform alphaQGD / ScQGD: the READ_IF_PRESENT volScalarFields "alphaQGD" [QGDCoeffs_8C_source.html L119-160] and
 * "ScQGD" [constScPrModel1_8C_source.html L66-79] of the time directory, cell values (nCells) and patch values
 * (nBoundaryFaces) as the files' own boundary conditions evaluate them.  NULL keeps the uniform value of the options.
 * Call before qgd_case_set_fields. */
int qgd_case_set_qgd_coeffs(qgd_case_t c, const double* alphaQGD, const double* alphaQGDb, const double* ScQGD,
                            const double* ScQGDb);

/* Initial cell fields U (nCells*3), T, p (HOST pointers); evaluates the BCs,
 * thermo.correct() and the derived conserved fields like createFields.H
 * [QGDFoam_2createFields_8H_source.html L3-109]. */
int qgd_case_set_fields(qgd_case_t c, const double* U, const double* T,
                        const double* p);

/* updateFields.H + updateFluxes.H on the current state.  Afterwards the face
 * fields "phiJm","phiJmU","phiP","phiPi","phiJmH","phiQ","phiPiU","phiwStar",
 * "phi","tauQGDf" are readable with qgd_case_get_field (debug / BC coupling
 * path: it materialises them; qgd_case_step does not). */
int qgd_case_update_fluxes(qgd_case_t c);

/* nSteps passes of the QGDFoam loop body, fully device resident. */
int qgd_case_step(qgd_case_t c, int32_t nSteps);

/* Named field copy-out to HOST.  Cell fields: "rho","U","p","e","T","rhoU",
 * "rhoE","c","psi","mu","alphau","tauQGD","muQGD","alphauQGD","hQGD","H".
 * Face fields: see qgd_case_update_fluxes plus "hQGDf"; with implicitDiffusion true also "phiTauMC" (3 per face)
 * and "phiSigmaDotU" as the last step formed them [QGDFoam_2updateFluxes_8H_source.html L107-111, QGDUEqn_8H_source.html L72-74].
 * Boundary fields: "<cellfield>.boundary".  outDoubles = capacity of out. */
int qgd_case_get_field(qgd_case_t c, const char* name, double* out,
                       int64_t outDoubles);

/* info[0]=time, [1]=deltaT, [2]=CoNum, [3]=min(rho), [4]=min(e), [5]=step count */
int qgd_case_info(qgd_case_t c, double info[6]);
/* Whether this case advances with the fused kernel of the explicit step (QGD_FUSED, default on): a uniform 3-D GaussVolPoint case, explicit
 * branch, fixed deltaT (linear or `Gauss upwind` qgdFlux schemes); shards included.  One workgroup per block of <= 128 cells then stages the records of those
 * cells and of the cells around them in LDS, forms the vertex values of the block from them (volPointInterpolation's weights, pointCells
 * order), computes every internal face of its cells into LDS and advances the cells from there, in the summation order of
 * fvc::surfaceIntegrate: neither the vertex values nor the net fluxes of internal faces reach device memory, and the vertex kernel, the face
 * kernel and the cell kernel are one launch.  Same arithmetic, bit-identical states.  info[0] = 1 when in use (2: the same blocks assemble
 * the three U systems of an unsharded fixed-deltaT implicitDiffusion case in one launch -- vertex values, QGD fluxes, tauMC, phiTauMC, the rows
 * [QGDUEqn_8H_source.html L36-68, QGDFoam_2updateFluxes_8H_source.html L95-111], QGD_IMPL_FUSED; 3: Courant-number control, adjustTimeStep
 * [QGDCourantNo_8H_source.html L36-53, setDeltaT-QGDQHD_8H_source.html L41-61] -- the blocks run up to their cells' flux sums and their faces'
 * Courant partials, deltaT follows on the device, a cell kernel advances: two launches, QGD_FUSED_ADJUST; 0: the separate kernels), [1] = blocks, [2] = internal
 * faces computed per step (faces on a block's surface are computed by the block on either side), [3] = LDS bytes per workgroup, [4] = cell
 * records staged per step over all blocks, [5] = of which with their second record and centre (own cells + cells across a face), [6] =
 * vertex values formed per step over all blocks, [7] = on a shard, the leading blocks that hold the cells a neighbouring rank waits for (phase 10
 * launches them, phase 11 the others; whole bricks, so they hold other owned cells too).  Owned cells / info[1] = mean cells per block: 128
 * on a box its bricks divide, 123 on a 50-plane slab between two cuts.  qgd_case_update_fluxes and every other branch keep the separate kernels. */
int qgd_case_fused_info(qgd_case_t c, int64_t info[8]);
/* The linear solves of the implicitDiffusion branch in the last step (what OpenFOAM prints as "Solving for Ux, Initial
 * residual = ..., Final residual = ..., No Iterations ...") [QGDUEqn_8H_source.html L54-68, QGDEEqn_8H_source.html L53-61]:
 * info[0..3] = iterations of Ux, Uy, Uz, e; [4..7] = initial, [8..11] = final normalised residuals; [12] = number of steps
 * since qgd_case_set_fields in which a solve stopped above implicitTol (iteration limit or breakdown: the step keeps the last
 * iterate, as OpenFOAM does, and counts here); [13] = 0 explicit branch | 1 implicit, conjugate gradients (QGD_IMPL_SOLVER=pcg) |
 * 2 implicit, Chebyshev iteration (the default).  The Chebyshev iteration measures the TRUE residual b - A x of its iterate (the
 * conjugate-gradient loop a recurrence); when that stops falling below 1e-8 -- the rounding floor of the product, which OpenFOAM's
 * normalisation can lift to 1e-12 on nearly uniform fields -- the component stops there and is NOT counted as unconverged in [12];
 * [14] = number of steps in which a solve ended that way ABOVE implicitTol (OpenFOAM would have iterated on to maxIter and printed the
 * residual: "solved to implicitTol" holds only while [12] and [14] are both 0); [15] reserved. */
int qgd_case_implicit_info(qgd_case_t c, double info[16]);
/* measurement: `reps` matrix products of the three-component U system (QGDUEqn_8H_source.html L54-68: the `fvm::laplacian(muf,U)`
 * matrix applied to a search direction; the kernel the branch spends most of its time in) between two HIP events, on the vectors
 * the last step left; info = {average ms per product, rows} */
int qgd_case_implicit_apply_time(qgd_case_t c, int reps, double info[2]);

/* ---- the implicitDiffusion branch on a cell-range shard -------------------------------------------------------------------------
 * implicitDiffusion true is the reference's default [QGDThermo_8C_source.html L70-82].  Its two implicit equations
 * [QGDUEqn_8H_source.html L54-75, QGDEEqn_8H_source.html L53-64] are solved here with the scalars of the solve in a control block
 * on the device (the three velocity components as three right-hand sides of ONE matrix walk).  The solver is a Chebyshev iteration
 * on the Jacobi-preconditioned system by default (these matrices are strictly diagonally dominant, Gershgorin bounds the spectrum;
 * no dot products), or Jacobi-preconditioned conjugate gradients with QGD_IMPL_SOLVER=pcg; the phases, reductions and messages below
 * serve both (a slot the algorithm in use does not write is zero, a phase it does not need does nothing).  Under MPI the reference
 * reaches across ranks through processor patches inside fvm::laplacian, the linear solver and fvc::grad.  With one rank per shard
 * the advance (after phase 0, the flux assembly) is, through qgd_case_step_phase:
 *   phase 20  deltaT, fvc::grad(U) of the old state          -> exchange message kind 1
 *   phase 21  tauMC / phiTauMC, rho, rhoU, the three U systems and their start values -> exchange kind 4
 *   phase 22  first solver phase (A x, r, row radii)          -> SUM-reduce control[0..12), MAX-reduce control[32..36)
 *   phase 23  normFactor                                      -> SUM-reduce control[12..16)
 *   phase 24  first residual; Chebyshev: x_1 | CG: search direction, r.z -> SUM-reduce control[16..20); exchange kind 3
 *   repeat until qgd_case_implicit_solve_status says done:
 *     phase 25 Chebyshev: one step, sum |b - A x| | CG: A d, d.Ad    -> SUM-reduce control[20..24)
 *     phase 26 Chebyshev: residual, done?, next constants | CG: x, r, |r|, r.z -> SUM-reduce control[24..32)
 *     phase 27 Chebyshev: nothing | CG: new direction           -> exchange kind 3
 *   phase 28  U into the records, its boundary conditions     -> exchange kind 2
 *   phase 29  fvc::grad(U) of the new velocity                -> exchange kind 1
 *   phase 30  phiSigmaDotU, the energy equation, the e system and its start value -> exchange kind 4; then 22, 23, 24, (25, 26, 27)*
 *   phase 35  rhoE, thermo, p, boundary refresh               -> the state message (qgd_case_halo_pack / unpack), phase 2
 * control: 68 device doubles, slot-major, control[slot * 4 + component].  Message kinds: 1 = fvc::grad(U), 9 per cell; 2 = U, 3 per
 * cell; 3 = what the next matrix product reads in the ghost columns (Chebyshev: the iterate, CG: the search direction), 4 = the start
 * value of the solve in flight, one per right-hand side per cell (counts report room for three).
 * An unsharded case runs the same phases back to back inside qgd_case_step; qgd_case_step_sharded drives them over RCCL. */
int qgd_case_implicit_halo_count(qgd_case_t c, int slot, int kind, int64_t* sendCount, int64_t* recvCount);
int qgd_case_implicit_halo_pack(qgd_case_t c, int slot, int kind, double* sendBufDevice);
int qgd_case_implicit_halo_unpack(qgd_case_t c, int slot, int kind, const double* recvBufDevice);
int qgd_case_implicit_control(qgd_case_t c, double control[68], int set);
int qgd_case_implicit_control_ptr(qgd_case_t c, void** devicePtr);
/* waits for the stream; status[0] != 0: every right-hand side of the solve in flight is done; status[1] = their number */
int qgd_case_implicit_solve_status(qgd_case_t c, double status[2]);

/* ---- halo exchange of ghost-cell primitives (multi-GPU) ------------------- */
/* A shard has one halo slot per neighbouring shard: a qgd_mesh_box slab (kLo>0 or kHi<nzGlobal) has slot 0 = lower k
 * and slot 1 = upper k; a qgd_mesh_shard mesh has one per rank in "haloPeer".  count / recv_count = number of doubles
 * in the message sent to / received from that neighbour.  pack gathers the owned boundary-layer cells' records into
 * the DEVICE buffer sendBuf; unpack scatters recvBuf into the ghost cells.
 * The state message carries 8 doubles per cell -- {rho, Ux, Uy, Uz, p, e} as SURVEY 8(e) plans, plus H and muQGD, the two derived
 * quantities a receiver cannot rebuild from them (rhoE is an independent field under the listing's energy re-solve; muQGD is formed with
 * the previous step's pressure and the owner's hQGD); c and alphaQGD/c are recomputed on arrival -- and 12 per patch face of those cells
 * (both records, the qgdFlux gradient, the lagged patch density).  Always ask qgd_case_halo_count: the layout may change.
 * Both are asynchronous on the case's stream; qgd_case_stream_sync waits. */
/* Plain device buffers for callers that own the transport (GPU-aware MPI, RCCL, ...). */
int qgd_device_alloc(qgd_device_t d, int64_t bytes, void** devicePtr);
int qgd_device_release(qgd_device_t d, void* devicePtr);
int qgd_case_halo_count(qgd_case_t c, int slot, int64_t* count);
int qgd_case_halo_recv_count(qgd_case_t c, int slot, int64_t* count);
int qgd_case_halo_pack(qgd_case_t c, int slot, double* sendBufDevice);
int qgd_case_halo_unpack(qgd_case_t c, int slot, const double* recvBufDevice);
int qgd_case_stream_sync(qgd_case_t c);
/* Run the case's kernels on a caller-owned HIP stream (hipStream_t passed as
 * void*), e.g. the stream the RCCL halo transfers are ordered on, so that
 * compute and exchange need no host synchronisation between them. */
int qgd_case_set_stream(qgd_case_t c, void* hipStream);
/* One step in two stream-ordered phases so that the exchanges between them are the caller's:
 *   phase 0 = flux assembly; with adjustTimeStep it leaves {max Cof, -min tauQGDf} of this shard in the
 *             2-double device buffer of qgd_case_reduction_ptr -- MAX-all-reduce it over the ranks in place
 *             [QGDCourantNo_8H L50, setDeltaT-QGDQHD_8H L46 are global reductions];
 *   phase 1 = deltaT, cell update, boundary refresh; then halo_pack / exchange / halo_unpack.
 * To overlap the exchange with the bulk of the cell update, phase 1 splits further:
 *   phase 10 = deltaT + update of the shard's boundary layer only (the cells a neighbour needs and their patch faces),
 *   phase 11 = update of all remaining owned cells and patch faces;
 * halo_pack may start as soon as phase 10 is done (on the halo stream of qgd_case_set_halo_stream, ordered after the
 * compute stream by the caller), phase 11 runs meanwhile, and the next phase 0 waits for halo_unpack.
 * Ghost cells and their patch faces are written by halo_unpack only.  phase 2 is a no-op hook.
 * phase 3 = one whole step of an UNSHARDED case, stream-ordered like the others (qgd_case_step without its host synchronisation; the
 *   fused face + cell kernel when qgd_case_fused_info says the case uses it).
 * Phase 0 itself splits when qgd_case_mid_exchange_needed says so (a shard whose GaussVolPoint stencil meets a wall with the qgdFlux
 * pressure condition): GaussVolPoint re-evaluates p's boundary conditions inside fvsc::grad(p) [GaussVolPointStencil_8C_source.html
 * L73 -> qgdFluxFvPatchScalarField_8C_source.html L184-192], a ghost cell's patch face forms that mid-step patch pressure from an
 * incomplete stencil, and the vertex values of p on the wall carry it into the stencil of owned faces (1e-7 in a few cells).  So
 *   phase 5 = the assembly up to and including that re-evaluation  -> exchange the mid message (qgd_case_mid_halo_*: 2 doubles per
 *             patch face of the boundary-layer cells)
 *   phase 6 = the rest of the assembly (+ the Courant reduction of phase 0).
 * What the reference gets from the processor-patch evaluation inside correctBoundaryConditions(). */
int qgd_case_step_phase(qgd_case_t c, int phase);
int qgd_case_mid_exchange_needed(qgd_case_t c, int32_t* needed);
int qgd_case_mid_halo_count(qgd_case_t c, int slot, int64_t* sendCount, int64_t* recvCount);   /* in doubles */
int qgd_case_mid_halo_pack(qgd_case_t c, int slot, double* sendBufDevice);
int qgd_case_mid_halo_unpack(qgd_case_t c, int slot, const double* recvBufDevice);
int qgd_case_reduction_ptr(qgd_case_t c, void** devicePtr);
/* Stream (hipStream_t as void*) the halo pack/unpack kernels run on; default: the case's compute stream. */
int qgd_case_set_halo_stream(qgd_case_t c, void* hipStream);

/* ---- native halo transport (RCCL over xGMI, inside the library) ------------------------------------------------------ */
/* For a C++/MPI host: what replaces the reference's per-gradient-call PstreamBuffers exchange
 * [extendedFaceStencilScalarGrad_8C_source.html L145-233] and the processor-patch evaluation behind
 * correctBoundaryConditions() [GaussVolPointStencil_8C_source.html L73] is ONE grouped ncclSend/ncclRecv pair per
 * neighbouring rank per step.  RCCL is bound at run time (an RCCL already loaded in the process is reused, otherwise
 * librccl.so of the ROCm installation; QGD_RCCL_LIB overrides); without it these entries return QGD_ERR_NOT_IMPLEMENTED
 * and the pack/unpack entries above remain for callers with their own transport (GPU-aware MPI, torch.distributed).
 * Bootstrap like NCCL's: rank 0 asks for the 128-byte unique id, the host broadcasts it (MPI_Bcast / Pstream), every rank
 * creates its communicator for its HIP device. */
typedef struct qgd_comm_s* qgd_comm_t;
int qgd_comm_unique_id(void* id128);
int qgd_comm_create(int deviceId, int rank, int nRanks, const void* id128, qgd_comm_t* out);
int qgd_comm_free(qgd_comm_t comm);
/* info = {this rank (ncclCommUserRank), ranks RCCL itself counts in the communicator (ncclCommCount), HIP device (ncclCommCuDevice)}:
 * what the transport SAW, for logs and bench lines -- not what the caller asked for */
int qgd_comm_info(qgd_comm_t comm, int32_t info[3]);
/* pack -> send/recv with peers[slot] (rank behind each halo slot; < 0: skip the slot) -> unpack, stream-ordered on the
 * case's stream, buffers owned by the library.  No-op for an unsharded case. */
int qgd_case_halo_exchange(qgd_case_t c, qgd_comm_t comm, const int32_t* peers, int nSlots);
/* MAX all-reduce of the {max Cof, -min tauQGDf} device buffer between step phases 0 and 1 (adjustTimeStep)
 * [QGDCourantNo_8H_source.html L50, setDeltaT-QGDQHD_8H_source.html L46]; no-op on one rank */
int qgd_case_allreduce_max(qgd_case_t c, qgd_comm_t comm);
/* One step of a sharded case: assemble, (adjustTimeStep: all-reduce), advance, exchange.  overlapped != 0: the shard's
 * boundary layer is updated first and the exchange runs on the library's halo stream while the compute stream updates the
 * remaining cells; the next assembly waits for the unpack through an event.  Stream-ordered (qgd_case_stream_sync waits). */
int qgd_case_step_sharded(qgd_case_t c, qgd_comm_t comm, const int32_t* peers, int nSlots, int overlapped);
/* The QHD case over the same transport: one message of kind 0 / 1 / 2 per neighbouring rank (after qgd_qhd_case_set_fields:
 * kind 0 once), and nSteps whole steps -- phases, ncclAllReduce of the control block, exchanges -- with no host wait except the
 * run-ahead check of the pressure solve (the host reads the "done" flag of iteration i-2 before it queues iteration i). */
int qgd_qhd_case_halo_exchange(qgd_qhd_case_t c, qgd_comm_t comm, const int32_t* peers, int nSlots, int kind);
int qgd_qhd_case_step_sharded(qgd_qhd_case_t c, qgd_comm_t comm, const int32_t* peers, int nSlots, int32_t nSteps);

/* ---- measurement ------------------------------------------------------------ */
/* Kernel ids for qgd_case_kernel_time. */
enum {
    QGD_K_POINT = 0,   /* cell -> vertex interpolation                          */
    QGD_K_FACE = 1,    /* fused internal-face gradient + flux kernel            */
    QGD_K_BFACE = 2,   /* boundary-face flux kernel                             */
    QGD_K_CELL = 3,    /* flux gather + Euler update + thermo + QGD coefficients */
    QGD_K_BC = 4,      /* boundary-condition refresh                            */
    QGD_K_BPOINT = 5,  /* patch points: mean of the surrounding patch-face values */
    QGD_K_COUNT = 6
};
/* Enable HIP-event timing of every launch of kernel `k` on the case's own
 * stream; totals since the last reset. */
int qgd_case_timing(qgd_case_t c, int enable);
int qgd_case_kernel_time(qgd_case_t c, int k, double* totalMs, int64_t* launches);
int qgd_case_timing_reset(qgd_case_t c);
/* Bytes resident on the device for this case + its mesh. */
int qgd_case_device_bytes(qgd_case_t c, int64_t* bytes);

#ifdef __cplusplus
}
#endif
#endif /* QGD_AMD_H */
