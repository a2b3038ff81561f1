// qgd_setup.hpp -- static, per-mesh stencil data in the flat layouts the HIP
// kernels stream (built once on the host, uploaded by qgd_device_create).
//
// Layout rules (DESIGN.md "Data layout in HBM"):
//  * every per-face array is indexed by the GLOBAL face label (internal faces
//    first, then boundary faces) so internal- and boundary-face kernels share it
//  * streamed per-face data is SoA (one array per scalar) for coalesced loads
//  * gathered per-cell / per-vertex data is AoS records (one record = one
//    contiguous 16-B aligned chunk)
#pragma once
#include <cstdint>
#include <memory>
#include <type_traits>
#include <utility>
#include <vector>

#include "qgd_mesh.hpp"

namespace qgd {

enum FaceKind : uint8_t { FK_QUAD = 0, FK_TRI = 1, FK_OTHER = 2, FK_SKIP = 3 };

struct StaticData {
    int32_t nP = 0, nF = 0, nIF = 0, nC = 0, nBF = 0;
    int32_t nGeomD = 3, ie1 = 0, ie2 = 1, ie3 = 2;
    bool hasTri = false;

    // ---- faces --------------------------------------------------------------
    RawVec<int32_t> own;     // nF   (RawVec: the per-face / per-cell tables are filled by parallel loops, qgd_setup.cpp)
    RawVec<int32_t> nei;     // nIF
    RawVec<int32_t> verts;   // 4*nF (tri: [3] = -1)
    RawVec<uint8_t> fkind;   // nF
    RawVec<double> Sf[3];    // nF each
    RawVec<double> magSf;    // nF
    RawVec<double> w;        // nF linear weights
    RawVec<double> hf;       // nF hQGDf
    RawVec<double> dn;       // nF: nonOrthDeltaCoeffs (internal), deltaCoeffs (boundary)
    // GaussVolPoint 3-D: the 10 (quad) / 13 (triangle) Gauss coefficients of a face are NOT stored: the face kernel
    // rebuilds them from the vertex coordinates and the two cell centres, which are gathered (and cached) instead
    // of streaming 80 B per face.  Boundary faces use the mirror point C_O + 2 (C_f - C_O) as "neighbour centre".
    RawVec<double> X;        // 3*nP: vertex coordinates (packed 24-B records)
    RawVec<double> Cc;       // 3*nC: cell centres (packed 24-B records)
    std::vector<double> bN;       // 4*nBF: mirror points of the boundary faces
    std::vector<double> bmvON;    // nBF
    // GaussVolPoint 2-D
    std::vector<int32_t> ip13;    // 2*nF (ip1, ip3)
    std::vector<double> c2d;      // 6*nF: c1,c2,c3,c4,mv42,mv13 (SoA: [k*nF+f])
    // leastSquares (internal faces)
    // leastSquares stencil as sliced ELL over the internal faces (64-face slices, entry e of face f at
    // (lsqSlice[f/64] + e)*64 + f%64): neighbour cell and the three components of wf2*Gdf
    std::vector<int32_t> lsqSlice;  // nSlices+1
    std::vector<uint8_t> lsqCnt;    // nIF
    std::vector<int32_t> lsqCell;
    std::vector<double> lsqGx, lsqGy, lsqGz;
    std::vector<uint8_t> lsqDeg;  // nIF
    std::vector<uint8_t> lsqBndZero;  // nBF: 1 on constraint patches (gradient left zero)
    std::vector<uint8_t> bSymm;       // nBF: 1 on symmetryPlane / symmetry / wedge patches (empty: none) -- scalar patch fields there have snGrad = 0 by type

    // ---- points -------------------------------------------------------------
    // point -> cells in sliced-ELL form: a slice is 64 consecutive points (one wavefront); entry i of the 64
    // points of slice s sits at (pcSlice[s] + i)*64 + lane, so every gather-list load of a wave is one
    // contiguous 256/512-B run.  Rows keep OpenFOAM's pointCells order (ascending cell label); patch points
    // have count 0.
    std::vector<int32_t> pcSlice; // nSlices+1, in rows of 64 entries
    std::vector<uint8_t> pcCount; // nP
    RawVec<int32_t> pcCell;  // padded entries: -1
    RawVec<double> pcW;
    std::vector<int32_t> bpPoint; // patch points
    std::vector<int32_t> bpOff;   // bpPoint.size()+1
    std::vector<int32_t> bpFace;  // boundary-face index (global label - nIF)
    std::vector<double> bpW;
    // Point constraints of VECTOR / TENSOR vertex fields (L0 assumption: volPointInterpolation::interpolateBoundaryField ends with
    // pointConstraints::constrain): per patch point (row i of bpPoint) the operations cpOff[i] .. cpOff[i+1], applied in order to the
    // weighted mean of the patch-face values.  kind 0: x = (x + transform(T, x))/2 with T = I - 2 nn -- the evaluate() of a
    // symmetryPlane (n = the patch normal) or symmetry (n = the point normal) point patch, in patch order; kind 1: x = transform(T, x)
    // -- a wedge point patch (T = I - nn) and, last, the patch-patch ("corner") constraint of points on the rim of one or more
    // constraint patches (T = I - nn | the edge direction squared | 0 for one | two | three independent normals).  Scalars are
    // untouched (transform is the identity on them).  Empty when the mesh has no symmetryPlane / symmetry / wedge patch.
    std::vector<int32_t> cpOff;   // bpPoint.size()+1 (empty: no constraints)
    std::vector<uint8_t> cpKind;
    std::vector<double> cpT;      // 9 per operation, row-major

    // ---- cells --------------------------------------------------------------
    // cell -> faces flux gather list, sliced-ELL like pc*; rows in ascending face label (== the summation order of
    // fvc::surfaceIntegrate); item = f (cell is owner, +) or ~f (cell is neighbour, -); empty/halo faces left out
    std::vector<int32_t> cfSlice; // nSlices+1
    std::vector<uint8_t> cfCount; // nC
    RawVec<int32_t> cfItem;
    RawVec<int32_t> cfNbr;   // like cfItem: the cell across the face, -1 for boundary faces
    // Storage order of the net face fluxes (CaseView::flux).  Internal face f keeps its five fluxes at position
    // fpos[f] of each SoA plane: faces are bucketed by their rank among the faces their owner owns (bucket 0 = every
    // cell's first owned face, in cell order, then bucket 1, ...), so that consecutive cells find their own faces AND
    // the faces they are the neighbour of (on a hexahedral box: the i-1, j-1 and k-1 faces) at consecutive positions.
    // Boundary faces stay at their label.  cfPos = cfItem with the label replaced by the position (same row order, ~pos
    // when the cell is the neighbour): the cell kernel's gather list.
    std::vector<int32_t> fpos;    // nIF
    RawVec<int32_t> cfPos;   // like cfItem
    std::vector<double> V;        // nC
    std::vector<double> hQGD;     // nC
    std::vector<uint8_t> ghost;   // nC cell role (empty when unsharded): 0 owned, 1 ghost, 2 owned + sent to a neighbour

    // ---- boundary faces -----------------------------------------------------
    std::vector<int32_t> bPatch;  // nBF patch index
    std::vector<double> hQGDb;    // nBF: hQGD boundary (= hQGDf boundary)

    // ---- halo -----------------------------------------------------------------
    std::vector<std::vector<int32_t>> haloGhost, haloSend, haloGhostBF, haloSendBF;  // one entry per halo slot

    int64_t bytes() const;
};

// build every table above; `needLsq` / `needGvp2` are derived from nGeometricD
StaticData buildStaticData(const HostMesh& m);

// ---- face tiles of the LDS-staged 3-D GaussVolPoint kernel ------------------------------------------------------------
// A tile is `fb` consecutive internal faces (one workgroup).  Its faces reference far fewer distinct cell and vertex
// records than 2 + 4 per face (blockMesh numbering, fb = 128: 130 cells and 175 vertices instead of 256 + 512), so the
// workgroup brings each distinct record in ONCE, as contiguous 16-B pieces, stages it in LDS and lets every face pick
// its six records from there: cells[]/verts[] list the distinct labels of each tile in ascending order (off[] = the
// two CSR offsets per tile), locC/locV hold each face's positions in those lists (16 bit each).
struct FaceTiles {
    int32_t fb = 0;                  // 0: not built (not a 3-D mesh or switched off)
    int32_t maxCells = 0, maxVerts = 0;   // over the staged tiles
    std::vector<int32_t> off;        // 2*(nTiles+1): {cell offset, vertex offset}
    std::vector<int32_t> spill;      // tiles with more distinct records than the caps below (e.g. the last rows of a box, whose
                                     // cells own one internal face each): empty lists here, they go through the gather kernel
    std::vector<int32_t> cells, verts;
    std::vector<uint32_t> locC;      // nIF: owner position | neighbour position << 16
    std::vector<uint32_t> locV;      // 2*nIF: v0 | v1 << 16,  v2 | v3 << 16 (absent vertex: 0)
};
// distinct records a tile may hold: what the kernel's fixed number of piece loads per thread covers
// (4, 3 and 5 rounds of 16-B pieces for cell RecA / RecB / vertex RecA).  A box in blockMesh numbering has fb + 3 cells and
// 11/8 fb vertices in all but the row-end tiles (< 1 %), which the tighter vertex cap leaves to the gather kernel: LDS per
// workgroup stays at 26.8 KB for fb = 128, six workgroups per CU.
inline int32_t faceTileCapCells(int32_t fb) { return fb + fb / 16; }
inline int32_t faceTileCapVerts(int32_t fb) { return ((fb * 23) / 16 + 7) / 8 * 8; }
FaceTiles buildFaceTiles(const StaticData& s, int32_t fb);


// ---- cell blocks of the fused face + cell kernel (QGD_FUSED; qgd_kernels.hip fusedFaceCellKernel) ---------------------------------
// A block is up to 128 cells that sit together in space (consecutive cells of a Morton order of the cell centres: an 8x4x4 brick on
// a uniform box) and EVERY internal face of those cells -- the faces on the block's surface are computed by the block on either side.
// One workgroup stages the distinct cell records (the block's own cells first, then the cells across its surface) and vertex records
// of those faces in LDS, computes the five net fluxes of each face into LDS, and advances its own cells out of LDS: the fluxes of
// internal faces never reach HBM.  Fixed strides (capC, capV, capF, capE), lists padded with their last entry.
struct FusedBlocks {
    int32_t nBlocks = 0;
    int32_t nLayerBlocks = 0;                         // a shard: the first nLayerBlocks blocks hold the cells a neighbour waits for
    int32_t capC = 0, capV = 0, capF = 0, capE = 0;   // strides: staged cells, staged vertices, faces, face entries per own cell
    int32_t maxC = 0, maxV = 0, maxF = 0;             // what the largest block uses (sizes the LDS)
    RawVec<int32_t> hdr;      // 4 per block: own cells, staged cells, staged vertices, faces
    RawVec<int32_t> cells;    // capC per block
    RawVec<int32_t> verts;    // capV per block
    RawVec<int32_t> faceLabel;   // capF per block: the global face label
    // The block's LOCAL TOPOLOGY -- per face the positions of its two cells and four vertices in the staged lists, per own cell its face
    // entries, per vertex the positions of its cells -- is stored per TEMPLATE, and hdr2[1] names a block's template: the interior bricks of
    // a structured region are all alike (a 400^3 box: 500 000 blocks, a few hundred templates), so 12.4 of the 31.3 KB a block streamed per step
    // in round 5 stay in L2 instead.  Blocks with a patch face (its entry is the face's own label) and the blocks of a jittered, renumbered mesh
    // (no two bricks list their cells alike) have templates of their own: the same bytes as before, read through the template id.
    int32_t nTemplates = 0, templated = 0;
    RawVec<uint32_t> facePos;    // 3 per face, capF faces per template: lo | ln << 16, v0 | v1 << 16, v2 | v3 << 16 (positions in the staged lists)
    // the vertex values formed inside the block (volPointInterpolation's inverse-distance weights, pointCells order): the cells around the
    // block's vertices that are neither its own nor across one of its faces ("extra": edge and corner neighbours) are staged too, behind
    // the others in `cells`; hdr2[0] = all staged cells.  Per vertex: its cells' positions in `cells` and the weights, entry-major
    // (capPE x capV per block); count 0 = a patch point, whose value the patch-point kernel has put into the vertex records.
    int32_t capPE = 0, maxTot = 0, maxAll = 0;        // cells per vertex; staged cells incl. extras / without them, of the largest block
    int32_t maxLds = 0;                               // LDS bytes of the records of the block that needs most (the kernel lays each block out by its own counts)
    int32_t maxLdsImpl = 0;                           // ... in the layout of the implicitDiffusion branch's assembly (fvc::grad(U) staged too, eight flux planes)
    RawVec<int32_t> hdr2;     // 4 per block: all staged cells, template, 0, 0
    RawVec<uint8_t> vCount;   // capV per block
    RawVec<uint16_t> vPos;    // capPE x capV per template
    RawVec<double> vW;        // capPE x capV per block
    RawVec<uint8_t> nEntry;   // 128 per block: face entries of each own cell
    RawVec<int32_t> entry;    // capE x 128 per template, entry-major: (local face << 1) | (1: the cell is the neighbour, minus), or ~label of a boundary face
    int32_t brick[3] = {0, 0, 0};  // the lattice brick the blocks were cut from (8 x 4 x 4 unless another shape fills the blocks better; 0: count-based runs)
    double buildSeconds = 0.0;     // host time of buildFusedBlocks
    int64_t facesComputed = 0;     // over all blocks (a face between two blocks is computed by both)
    int64_t cellsStaged = 0, cellsStagedFull = 0, vertsStaged = 0;   // over all blocks: cell records staged (RecA), of which with RecB + centre; vertices formed
};
// LDS a block's records may take so that three blocks (+ 3 KB of parked face entries each) share a CU's 160 KB: what an 8x4x4 brick needs
constexpr int32_t kFusedLdsTarget = 48 * 360 + 32 * 288 + 72 * 225 + 24 * 288;
constexpr int32_t kFusedCells = 128, kFusedCapC = 320, kFusedCapV = 256, kFusedCapF = 512, kFusedCapTot = 384;   // kFusedCapC: own + across-a-face cells
FusedBlocks buildFusedBlocks(const StaticData& s);

}  // namespace qgd
