// qgd_mesh.hpp -- host-side polyMesh container of the MI355X QGD face-flux path.
//
// Flat arrays in OpenFOAM's own conventions (upper-triangular face order,
// boundary faces grouped per patch, int32 labels, fp64 scalars), plus the
// derived geometry the path needs: face area vectors/centres, cell
// centres/volumes, linear interpolation weights, delta coefficients and the
// CSR adjacency used by the kernels.  Nothing here touches the GPU.
//
// The geometry rules restate OpenFOAM v2312 (the release the reference pins in
// /root/reference/README.md:32-37); OpenFOAM itself is not part of the
// reference tree, so these are marked "L0 assumption" where they matter.
#pragma once
#include <cstdint>
#include <string>
#include <utility>
#include <vector>

namespace qgd {

// vector whose resize() leaves new elements uninitialised: multi-GB tables that a parallel loop fills completely are first touched by the
// threads that fill them (a value-initialising resize touches every page from one thread first: 4 s per GB in this kind of container)
template <class T>
struct DefaultInitAllocator : std::allocator<T> {
    template <class U> struct rebind { using other = DefaultInitAllocator<U>; };
    using std::allocator<T>::allocator;
    template <class U> void construct(U* p) noexcept(std::is_nothrow_default_constructible<U>::value) { ::new (static_cast<void*>(p)) U; }
    template <class U, class... Args> void construct(U* p, Args&&... args) { ::new (static_cast<void*>(p)) U(std::forward<Args>(args)...); }
};
template <class T> using RawVec = std::vector<T, DefaultInitAllocator<T>>;
// dst = src with every page of dst first touched by the thread that copies it
template <class T, class A>
inline void parallelCopy(RawVec<T>& dst, const std::vector<T, A>& src) {
    dst.resize(src.size());
    const int64_t n = (int64_t)src.size();
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) dst[i] = src[i];
}


struct Patch {
    std::string name;
    int32_t type = 0;   // QGD_PATCH_*
    int32_t start = 0;  // global label of first face
    int32_t size = 0;
    int32_t globalSize = -1;  // faces of this patch in the unsharded mesh (-1: this mesh is the whole mesh)
    // symmetryPlane patches: the ONE normal of the patch, the average of its faces' unit normals (L0 assumption:
    // symmetryPlanePolyPatch::calcGeometry, n_ = gAverage(faceNormals())), which symmetryPlaneFvPatchField uses on every face
    // instead of the face's own normal.  A shard inherits the normal of the whole patch (extractShard) so that cut and uncut
    // meshes agree bit for bit.
    double nHat[3] = {0.0, 0.0, 0.0};
    bool nHatInherited = false;
    bool nonEmptyGlobally() const { return (globalSize >= 0 ? globalSize : size) > 0; }
};

struct HostMesh {
    // ---- primitives (constant/polyMesh) -------------------------------------
    int32_t nPoints = 0, nFaces = 0, nInternalFaces = 0, nCells = 0;
    std::vector<double> points;        // 3*nPoints
    std::vector<int32_t> faceOffsets;  // nFaces+1
    std::vector<int32_t> facePoints;   // sum of face sizes
    std::vector<int32_t> owner;        // nFaces
    std::vector<int32_t> neighbour;    // nInternalFaces
    std::vector<Patch> patches;

    // ---- geometry -----------------------------------------------------------
    RawVec<double> Sf, Cf;        // 3*nFaces   (RawVec: the geometry is written in full by parallel loops, computeGeometry)
    RawVec<double> magSf;         // nFaces
    RawVec<double> C;             // 3*nCells
    RawVec<double> V;             // nCells
    RawVec<double> weights;       // nFaces  (boundary = 1)
    RawVec<double> deltaCoeffs;   // nFaces  1/|d|   (patch-normal delta on boundary)
    RawVec<double> nonOrthDeltaCoeffs;  // nFaces  1/max(n.d, 0.05|d|)
    int32_t geometricD[3] = {1, 1, 1}; // -1 for the direction of empty patches
    int32_t nGeometricD = 3;
    bool userGeometry = false;         // Sf, Cf, C, V came from the caller (qgd_mesh_set_geometry), not from the points

    // ---- cell-range sharding --------------------------------------------------
    // one slot per neighbouring shard: the local ghost cells refreshed from it and the local owned cells it needs,
    // both in ascending label order (box slabs: slot 0 = lower, slot 1 = upper neighbour, either may be empty)
    std::vector<std::vector<int32_t>> haloGhost, haloSend;
    std::vector<int32_t> haloPeer;     // rank behind each slot (-1 when the builder does not know it)
    // cyclic patch pairs served by ghost cells (unrollCyclic): slot k's ghosts are copies of THIS mesh's own cells, and the message they
    // expect is the one slot haloSelf[k] packs (empty, or -1 per slot, on an ordinary shard)
    std::vector<int32_t> haloSelf;
    std::vector<uint8_t> cellIsGhost;  // nCells (empty when unsharded)
    // labels in the unsharded mesh (filled by extractShard, empty otherwise); faceGlobal is -1-label for flipped faces
    std::vector<int32_t> cellGlobal, faceGlobal, pointGlobal;
    int32_t ownedBegin = 0, ownedEnd = 0;  // local label range of the owned cells of a shard (extractShard, makeBox slabs)
    int64_t cellGlobalOffset = 0;          // box slabs: global label = local label + offset (cellGlobal stays empty)
    // hQGDf of the faces of the halo patch(es) AS THE UNSHARDED MESH HAS IT (they are internal faces there), in patch order:
    // with it a ghost cell's hQGD -- an area-weighted mean over all its faces [QGDCoeffs.C L323-362] -- comes out right
    std::vector<double> haloFaceH;
    // the user's faceSet "degenerateStencilFaces" [leastSquaresStencil.C L63-128]: internal faces the leastSquares stencil
    // treats as degenerate (nf * snGrad) whatever their weights say
    std::vector<int32_t> degenerateFaces;

    int32_t nBoundaryFaces() const { return nFaces - nInternalFaces; }
    int32_t faceSize(int32_t f) const { return faceOffsets[f + 1] - faceOffsets[f]; }
    int32_t patchOfFace(int32_t f) const;  // -1 for internal faces

    // (re)compute Sf, Cf, C, V from points, then the derived coefficients
    void computeGeometry();
    // weights / deltaCoeffs / geometric directions from Sf, Cf, C
    void computeDerived();
    // sanity checks; returns empty string when consistent
    std::string check() const;
};

// CSR adjacency derived from the primitives
struct Csr {
    std::vector<int32_t> offsets;
    std::vector<int32_t> items;
    int32_t rowSize(int32_t r) const { return offsets[r + 1] - offsets[r]; }
};

// point -> cells, each row in ascending cell label (OpenFOAM pointCells order)
Csr buildPointCells(const HostMesh& m);
Csr buildPointCells(const HostMesh& m, const Csr& cellFaces);   // ... with buildCellFaces(m) at hand
// cell -> faces, ascending face label; used both for h_QGD and for the
// deterministic flux gather (== summation order of fvc::surfaceIntegrate)
Csr buildCellFaces(const HostMesh& m);
// cell -> faces in OpenFOAM cells() order: owned faces ascending, then
// neighbour faces ascending
Csr buildCellFacesFoamOrder(const HostMesh& m);

HostMesh makeBox(int32_t nx, int32_t ny, int32_t nzGlobal, int32_t kLo, int32_t kHi,
                 const double lo[3], const double hi[3], const int32_t patchTypes[6]);
HostMesh makeForwardStep(int32_t nx, int32_t ny, int32_t ixStep, int32_t iyStep,
                         double lx, double ly, double lz);
void jitterPoints(HostMesh& m, double amplitude, uint64_t seed);
void splitQuads(HostMesh& m, int32_t stride);
void splitEdges(HostMesh& m, int32_t stride);

// ---- renumbering and cell-range partitioning (qgd_partition.cpp) ---------------
// Relabel the cells (newOfOld[old] = new): owner/neighbour are swapped and the face reversed where needed, internal
// faces are re-sorted into upper-triangular order, boundary faces keep their order.  faceNewOfOld (optional, nFaces)
// receives new label, or -1-new when the face was reversed.
void renumberCells(HostMesh& m, const int32_t* newOfOld, int32_t* faceNewOfOld);
// reverse Cuthill-McKee order over the face-neighbour graph (bandwidth reduction before cell-range sharding)
std::vector<int32_t> cuthillMcKee(const HostMesh& m);
// Morton (Z-curve) order of the cell centres
std::vector<int32_t> mortonOrder(const HostMesh& m);
// The shard of rank `rank` when the cells are cut into the ranges cellStart[r] .. cellStart[r+1]: its owned cells plus
// one vertex-connected layer of ghost cells, every face of those cells (faces whose other cell is absent form a trailing
// QGD_PATCH_HALO patch), halo lists per neighbouring rank.
HostMesh extractShard(const HostMesh& g, int32_t nRanks, const int32_t* cellStart, int32_t rank);
// Translational cyclic patch pairs (patch indices {A, B}: face i of A is face i of B shifted) served by ghost cells: the real mesh followed by
// one vertex-connected layer of translated copies behind every half, the pairs' faces glued into internal faces, halo slots that refresh the
// copies from their originals on the same rank (haloSelf).  Rotational pairs are refused.
HostMesh unrollCyclic(const HostMesh& g, const std::vector<std::pair<int32_t, int32_t>>& pairs);

}  // namespace qgd
