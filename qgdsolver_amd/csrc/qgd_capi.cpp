// qgd_capi.cpp -- the C-ABI of include/qgd_amd.h (compiled with hipcc).
//
// One handle == one HIP device == one calling host thread.  Entries are
// synchronous at return unless documented otherwise.  No exceptions cross the
// boundary: every entry catches and converts to a status code, mirroring how the
// reference turns failures into FatalError exits [fvsc_8C_source.html L62,75-78].
// There is NO CPU fallback: without a HIP device the device/case entries fail
// with QGD_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>  // declarations only: RCCL is bound at run time (dlopen), the library does not link it

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/qgd_amd.h"
#include "qgd_device.hpp"
#include "qgd_mesh.hpp"
#include "qgd_setup.hpp"

using namespace qgd;

// ---------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------
static thread_local std::string g_lastError;
static int fail(int code, const std::string& msg) {
    g_lastError = msg;
    return code;
}
struct HipError : std::runtime_error {
    explicit HipError(const std::string& s) : std::runtime_error(s) {}
};
#define HIP_CHECK(expr)                                                                                     \
    do {                                                                                                    \
        hipError_t _e = (expr);                                                                             \
        if (_e != hipSuccess) {                                                                             \
            (void)hipGetLastError(); /* reported here, once: a later hipGetLastError() must not see it */   \
            throw HipError(std::string(#expr) + ": " + hipGetErrorString(_e) + " at " + __FILE__ + ":" +   \
                           std::to_string(__LINE__));                                                       \
        }                                                                                                   \
    } while (0)
#define QGD_TRY try {
#define QGD_CATCH                                                                 \
    }                                                                             \
    catch (const HipError& e) { return fail(QGD_ERR_HIP, e.what()); }             \
    catch (const std::invalid_argument& e) { return fail(QGD_ERR_INVALID, e.what()); } \
    catch (const std::exception& e) { return fail(QGD_ERR_INVALID, e.what()); }   \
    catch (...) { return fail(QGD_ERR_INVALID, "unknown exception"); }

// ---------------------------------------------------------------------------
// handle types
// ---------------------------------------------------------------------------
struct qgd_mesh_s {
    HostMesh m;
};

struct DeviceArena {
    std::vector<void*> ptrs;
    int64_t bytes = 0;
    void* raw(size_t nbytes) {
        void* d = nullptr;
        HIP_CHECK(hipMalloc(&d, nbytes));
        ptrs.push_back(d);
        bytes += (int64_t)nbytes;
        return d;
    }
    template <class T, class Al>
    T* upload(const std::vector<T, Al>& v) {
        if (v.empty()) return nullptr;
        void* d = raw(v.size() * sizeof(T));
        HIP_CHECK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
        return (T*)d;
    }
    template <class T>
    T* alloc(size_t n, bool zero = true) {
        if (n == 0) return nullptr;
        void* d = raw(n * sizeof(T));
        if (zero) { HIP_CHECK(hipMemset(d, 0, n * sizeof(T))); HIP_CHECK(hipStreamSynchronize(nullptr)); }   // see PressureSolver::alloc
        return (T*)d;
    }
    void release() {
        for (void* p : ptrs) (void)hipFree(p);
        ptrs.clear();
        bytes = 0;
    }
};

// Persistent per-device workspace of the host-pointer operator entries (qgd_fvsc_*, qgd_interpolate, qgd_qhd_*,
// qgd_species_flux): grow-only device buffers by slot, and two pinned staging chunks through which pageable caller memory
// is moved in a double-buffered pipeline (the host copy of chunk k+1 overlaps the DMA of chunk k).  Nothing is allocated
// or freed per call once the sizes have been seen.
struct Workspace {
    struct Buf { void* p = nullptr; size_t cap = 0; };
    std::vector<Buf> dev;
    void* pin[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    static constexpr size_t CHUNK = (size_t)32 << 20;
    int64_t bytes = 0;

    template <class T>
    T* get(size_t slot, size_t n) {
        if (dev.size() <= slot) dev.resize(slot + 1);
        const size_t need = std::max<size_t>(n, 1) * sizeof(T);
        Buf& b = dev[slot];
        if (b.cap < need) {
            if (b.p) { (void)hipFree(b.p); bytes -= (int64_t)b.cap; b.p = nullptr; b.cap = 0; }
            HIP_CHECK(hipMalloc(&b.p, need));
            b.cap = need; bytes += (int64_t)need;
        }
        return (T*)b.p;
    }
    void ensurePinned() {
        if (pin[0]) return;
        for (int i = 0; i < 2; ++i) {
            HIP_CHECK(hipHostMalloc(&pin[i], CHUNK, hipHostMallocDefault));
            HIP_CHECK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
        }
    }
    static bool devAccessible(const void* p) {
        hipPointerAttribute_t a;
        if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
        return a.type == hipMemoryTypeHost || a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged;
    }
    static void hostCopy(void* dst, const void* src, size_t n) {
        const int64_t blocks = (int64_t)((n + ((size_t)1 << 20) - 1) >> 20);
#pragma omp parallel for schedule(static) if (blocks > 4)
        for (int64_t b = 0; b < blocks; ++b) {
            const size_t off = (size_t)b << 20;
            std::memcpy((char*)dst + off, (const char*)src + off, std::min<size_t>((size_t)1 << 20, n - off));
        }
    }
    // host -> device, stream-ordered on s; returns when the source may be reused
    void h2d(void* dst, const void* src, size_t n, hipStream_t s) {
        if (!n) return;
        if (devAccessible(src)) { HIP_CHECK(hipMemcpyAsync(dst, src, n, hipMemcpyDefault, s)); HIP_CHECK(hipStreamSynchronize(s)); return; }
        ensurePinned();
        int k = 0;
        for (size_t off = 0; off < n; off += CHUNK, k ^= 1) {
            const size_t len = std::min(CHUNK, n - off);
            HIP_CHECK(hipEventSynchronize(ev[k]));
            hostCopy(pin[k], (const char*)src + off, len);
            HIP_CHECK(hipMemcpyAsync((char*)dst + off, pin[k], len, hipMemcpyHostToDevice, s));
            HIP_CHECK(hipEventRecord(ev[k], s));
        }
    }
    // device -> host, after everything queued on s; returns when dst holds the data
    void d2h(void* dst, const void* src, size_t n, hipStream_t s) {
        if (!n) return;
        if (devAccessible(dst)) { HIP_CHECK(hipMemcpyAsync(dst, src, n, hipMemcpyDefault, s)); HIP_CHECK(hipStreamSynchronize(s)); return; }
        ensurePinned();
        HIP_CHECK(hipEventSynchronize(ev[0]));
        HIP_CHECK(hipEventSynchronize(ev[1]));
        size_t prevOff = 0, prevLen = 0;
        int k = 0;
        for (size_t off = 0; off < n; off += CHUNK, k ^= 1) {
            const size_t len = std::min(CHUNK, n - off);
            HIP_CHECK(hipMemcpyAsync(pin[k], (const char*)src + off, len, hipMemcpyDeviceToHost, s));
            HIP_CHECK(hipEventRecord(ev[k], s));
            if (prevLen) { HIP_CHECK(hipEventSynchronize(ev[k ^ 1])); hostCopy((char*)dst + prevOff, pin[k ^ 1], prevLen); }
            prevOff = off; prevLen = len;
        }
        HIP_CHECK(hipEventSynchronize(ev[k ^ 1]));
        hostCopy((char*)dst + prevOff, pin[k ^ 1], prevLen);
    }
    void release() {
        for (Buf& b : dev) if (b.p) (void)hipFree(b.p);
        dev.clear();
        for (int i = 0; i < 2; ++i) {
            if (pin[i]) (void)hipHostFree(pin[i]);
            if (ev[i]) (void)hipEventDestroy(ev[i]);
            pin[i] = nullptr; ev[i] = nullptr;
        }
        bytes = 0;
    }
};

static int envChoice(const char* name, int dflt, const int* allowed, int nAllowed, int lo, int hi);

// ---- constraint patches in the resident cases -----------------------------------------------------------------------------------------
// OpenFOAM gives a field on a constraint patch the patch's own field type whatever the field file says (L0: fvPatchField<Type>::New,
// "patchFieldType overridden by the patch's constraint type"): empty patches and the cut planes of a shard carry nothing;
// symmetryPlane / symmetry reflect vectors (symmetryPlaneFvPatchField / basicSymmetryFvPatchField: evaluate = (pif +
// transform(I - 2 nn, pif))/2) and leave scalars zero-gradient.  The stencils know the same list
// [extendedFaceStencilScalarGrad.C L90-101, GaussVolPointBase3D.C L783-794].
static void constraintKinds(int ptype, int32_t& bcU, int32_t& bcT, int32_t& bcP) {
    if (ptype == QGD_PATCH_EMPTY || ptype == QGD_PATCH_HALO) { bcU = bcT = bcP = QGD_BC_NONE; }
    else if (ptype == QGD_PATCH_SYMMETRYPLANE || ptype == QGD_PATCH_SYMMETRY) { bcU = QGD_BC_SLIP; bcT = bcP = QGD_BC_ZEROGRADIENT; }
}
static void initPatchBC(PatchBCDev& b, const Patch& p) {
    std::memset(&b, 0, sizeof(b));
    b.ptype = p.type;
    b.bcU = b.bcT = b.bcP = QGD_BC_ZEROGRADIENT;
    constraintKinds(p.type, b.bcU, b.bcT, b.bcP);
    if (p.type == QGD_PATCH_SYMMETRYPLANE && (p.size > 0 || p.nHatInherited)) {
        b.planeN = 1;
        for (int k = 0; k < 3; ++k) b.nHat[k] = p.nHat[k];
    }
}
static void residentCasePatchCheck(const HostMesh& m, std::string& why, int& code) {
    why.clear(); code = 0;
    for (const Patch& p : m.patches) {
        if ((p.type == QGD_PATCH_CYCLIC || p.type == QGD_PATCH_WEDGE) && p.nonEmptyGlobally()) {
            why = std::string("patch '") + p.name + "' is a " + (p.type == QGD_PATCH_CYCLIC ? "cyclic" : "wedge") +
                  " patch: the resident cases do not serve coupled / wedge patch fields (use the fvsc operators, which do)";
            code = QGD_ERR_NOT_IMPLEMENTED;
            return;
        }
        if (p.type == QGD_PATCH_SYMMETRYPLANE) {
            for (int32_t f = p.start; f < p.start + p.size; ++f) {   // L0: symmetryPlanePolyPatch::calcGeometry, magSqr(n_ - nf) > SMALL is fatal
                double d2 = 0;
                for (int k = 0; k < 3; ++k) { const double x = p.nHat[k] - m.Sf[3 * (size_t)f + k] / m.magSf[f]; d2 += x * x; }
                if (d2 > 1e-15) { why = std::string("Symmetry plane '") + p.name + "' is not planar (use the symmetry patch type)"; code = QGD_ERR_INVALID; return; }
            }
        }
    }
}

struct qgd_device_s {
    int deviceId = 0;
    Workspace ws;
    double opMs[3] = {0, 0, 0};  // last host-pointer operator call: host->device, kernels, device->host (qgd_device_op_times)
    hipEvent_t opEv[2] = {nullptr, nullptr};
    DeviceArena arena;
    MeshView view{};
    int32_t nGeomD = 3;
    bool hasTri = false;
    bool wedgePrism = false;  // wedge patches + prism cells: GaussVolPoint is refused [fvsc_8C L65-82]
    int64_t fusedInfo[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // qgd_device_fused_blocks
    int64_t fusedFacesComputed = 0;   // internal faces the fused kernel's blocks compute per step, surface faces once per side (MeshView::fuBlocks > 0)
    int64_t fusedCellsStaged = 0, fusedCellsStagedFull = 0, fusedVertsStaged = 0;   // cell records / of which with RecB and centre / vertices its blocks stage per step
    std::vector<Patch> patches;
    // why a resident case (qgd_case_create / qgd_qhd_case_create) cannot run on this mesh, empty when it can: cyclic / wedge patches
    // with faces (their coupled / rotated patch fields are not served), a symmetryPlane that is not planar (fatal in OpenFOAM too)
    std::string caseRefusal; int caseRefusalCode = 0;
    std::vector<double> hf;  // host copy of hQGDf for the accessor
    // halo lists (device) and sizes, one entry per halo slot (neighbouring shard)
    struct HaloSlot {
        int32_t *ghost = nullptr, *send = nullptr, *ghostBF = nullptr, *sendBF = nullptr;
        int32_t nGhost = 0, nSend = 0, nGhostBF = 0, nSendBF = 0;
    };
    std::vector<HaloSlot> halo;
    // cyclic patch pairs served by ghost cells (qgd_mesh_unroll_cyclic): slot k unpacks what slot haloSelf[k] of the SAME case packs
    std::vector<int32_t> haloSelf;
    bool periodic() const { if (haloSelf.empty()) return false; for (int32_t x : haloSelf) if (x < 0) return false; return true; }
    bool sharded() const { for (const HaloSlot& h : halo) if (h.nGhost || h.nSend) return true; return false; }
    int32_t ownedBegin = 0, ownedEnd = 0;      // local labels of the owned cells (contiguous by construction of the shard builders)
    std::vector<int32_t> cellGlobal;           // extracted shards: label in the unsharded mesh per local cell (ascending)
    int64_t cellGlobalOffset = 0;              // box slabs: global label = local label + offset
    // local label of a cell of the unsharded mesh, -1 when this shard does not OWN it
    int32_t ownedLocalOf(int64_t globalCell) const {
        int64_t local = -1;
        if (!cellGlobal.empty()) {
            auto it = std::lower_bound(cellGlobal.begin(), cellGlobal.end(), (int32_t)globalCell);
            if (it != cellGlobal.end() && *it == globalCell) local = it - cellGlobal.begin();
        } else local = globalCell - cellGlobalOffset;
        return (local >= ownedBegin && local < ownedEnd) ? (int32_t)local : -1;
    }
    int32_t* sendAll = nullptr;     // send cells of every slot (each cell once)
    int32_t* sendBFAll = nullptr;   // their real-patch boundary faces
    int32_t nSendAll = 0, nSendBFAll = 0;
    hipStream_t stream = nullptr;
    int liveCases = 0;   // qgd_case_t / qgd_qhd_case_t created on this device and not freed yet (qgd_device_free refuses while > 0)
    ImplicitSolver* opSolver = nullptr;   // the linear solver of the stateless implicit operators (qgd_species_step_implicit), made on first use
    // MeshView::geoPos, built the first time a case that takes fvc::grad(U) per cell is created on this device (QGD_GEOPOS=0: never)
    void ensureFaceGeoPos() {
        static const int kOnOff[] = {0, 1};
        if (view.geoPos || view.nF == 0 || !envChoice("QGD_GEOPOS", 1, kOnOff, 2, 0, 0)) return;
        double4* g = arena.alloc<double4>((size_t)view.nF, false);
        (void)hipGetLastError();   // a stale error of an earlier, refused call must not be taken for this launch's
        launchFaceGeoPos(stream, view, g);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipStreamSynchronize(stream));
        view.geoPos = g;
    }
};

struct TimedLaunch {
    int k;
    hipEvent_t a, b;
};

struct qgd_case_s {
    qgd_device_s* dev = nullptr;
    qgd_case_options opt{};
    GasModel gas{};
    int stencil = ST_GVP3;      // device stencil kind
    // per-term fvsc entries (qgd_case_options::termStencil): mixB >= 0: the case walks its faces with the two stencils stencil < mixB, the
    // gradient components in mixMask by mixB (bit k: rho, Ux, Uy, Uz, p, e)
    int mixB = -1, mixMask = 0;
    int implXOrder = 0;         // QGD_IMPL_XEXTRAP: order of the start-value extrapolation of the implicit branch's solves (ImplView::have counts up to it)
    bool pRefresh = true;       // grad(p)'s word is GaussVolPoint: p's boundary conditions are re-evaluated inside it [GaussVolPointStencil_8C L73] (quirk B6)
    bool usesPoints = true;
    bool fused = false;         // qgd_case_step advances with fusedFaceCellKernel (QGD_FUSED)
    bool fusedAdj = false;      // adjustTimeStep: the blocks run up to their flux sums + Courant partials, cellFinishKernel advances once deltaT is known (QGD_FUSED_ADJUST)
    bool fusedImpl = false;     // implicitDiffusion: vertex values, QGD fluxes, tauMC and the U systems' rows are one launch on the same blocks (QGD_IMPL_FUSED)
    std::vector<double*> selfBuf;   // cyclic pairs served by ghost cells: one message buffer per halo slot (selfHaloExchange)
    bool ghostsCurrent = false;     // ... and whether the copies hold their originals' records (reset by set_fields / set_bc)
    bool hasQgdFlux = false;
    bool phiwRegistered = false;
    bool fieldsSet = false;
    std::vector<PatchBCDev> bc;
    PatchBCDev* bcDev = nullptr;
    DeviceArena arena;
    CaseView view{};
    double* dbgBuf = nullptr;
    ImplView impl{};            // implicitDiffusion branch: its face / cell work arrays
    ImplicitSolver* implSolver = nullptr;   // the two linear solves of the branch (device-scalar multi-right-hand-side PCG)
    int implSolveIndex = 0;                 // 0: the U solve is the one in flight, 1: the e solve
    bool reuseGradU = true;                 // QGD_IMPL_REUSE_GRADU (default 1)
    bool gradUValid = false;                // implicit branch, unsharded: fvc::grad(U) of phase 29 is still that of the records (phase 20 of the next step skips it)
    std::vector<double*> implSendBuf, implRecvBuf;   // native transport of the branch's own halo messages
    std::vector<double*> midSendBuf, midRecvBuf;     // the mid-assembly message (midExchangeOn)
    double* coef[4] = {nullptr, nullptr, nullptr, nullptr};  // device copies of non-uniform alphaQGD / ScQGD (cells, patch faces)
    double time = 0;
    int64_t steps = 0;
    // timing
    bool timing = false;
    std::vector<TimedLaunch> pending;
    std::vector<hipEvent_t> freeEvents;
    double totalMs[QGD_K_COUNT] = {};
    int64_t launches[QGD_K_COUNT] = {};
    hipEvent_t curStart = nullptr;
    // optional caller-owned stream (e.g. the one RCCL transfers are ordered on)
    hipStream_t userStream = nullptr;
    bool useUserStream = false;
    hipStream_t stream() const { return useUserStream ? userStream : dev->stream; }
    hipStream_t haloStream = nullptr;  // pack/unpack stream (defaults to stream())
    bool useHaloStream = false;
    // native halo transport (qgd_case_halo_exchange): message buffers per halo slot, events ordering the two streams
    std::vector<double*> sendBuf, recvBuf;
    hipStream_t ownHaloStream = nullptr;
    hipEvent_t evLayerDone = nullptr, evUnpacked = nullptr;
};

static hipEvent_t getEvent(qgd_case_s* c) {
    if (!c->freeEvents.empty()) {
        hipEvent_t e = c->freeEvents.back();
        c->freeEvents.pop_back();
        return e;
    }
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}
static void timingPre(void* ctx, int) {
    qgd_case_s* c = (qgd_case_s*)ctx;
    if (!c->timing) return;
    c->curStart = getEvent(c);
    if (c->curStart) (void)hipEventRecord(c->curStart, c->stream());
}
static void timingPost(void* ctx, int k) {
    qgd_case_s* c = (qgd_case_s*)ctx;
    if (!c->timing || !c->curStart) return;
    hipEvent_t e = getEvent(c);
    if (!e) return;
    (void)hipEventRecord(e, c->stream());
    c->pending.push_back(TimedLaunch{k, c->curStart, e});
    c->curStart = nullptr;
}
static void harvestTiming(qgd_case_s* c) {
    if (c->pending.empty()) return;
    (void)hipStreamSynchronize(c->stream());
    for (TimedLaunch& t : c->pending) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, t.a, t.b) == hipSuccess) {
            c->totalMs[t.k] += ms;
            c->launches[t.k]++;
        }
        c->freeEvents.push_back(t.a);
        c->freeEvents.push_back(t.b);
    }
    c->pending.clear();
}
static Launcher launcherOf(qgd_case_s* c) {
    Launcher L;
    L.stream = c->stream();
    L.pre = timingPre;
    L.post = timingPost;
    L.ctx = c;
    return L;
}

// The OpenMP runtime hipcc links spin-waits between parallel regions by default;
// on oversubscribed hosts that makes the (short) setup loops slower than serial.
__attribute__((constructor)) static void qgdInitOpenMP() { setenv("KMP_BLOCKTIME", "0", 0); }

static bool hasWedgeAndPrism(const HostMesh& m);
static int implHaloMove(qgd_case_s* c, int slot, int kind, double* buf, bool pack, hipStream_t stream);

// Tuning knobs of the measurement scripts (QGD_*): a value outside the supported set is an error, not a silent change of
// code path.  allowed == nullptr: any integer in [lo, hi].
static int envChoice(const char* name, int dflt, const int* allowed, int nAllowed, int lo = 0, int hi = 0) {
    const char* e = std::getenv(name);
    if (!e || !*e) return dflt;
    char* end = nullptr;
    const long v = std::strtol(e, &end, 10);
    bool ok = end && *end == '\0';
    if (ok && allowed) { ok = false; for (int i = 0; i < nAllowed; ++i) ok = ok || allowed[i] == v; }
    else if (ok) ok = v >= lo && v <= hi;
    if (!ok) throw std::invalid_argument(std::string(name) + "=" + e + " is not a supported value");
    return (int)v;
}

// ---------------------------------------------------------------------------
extern "C" {

const char* qgd_version(void) { return "qgdsolver_amd 0.1 (gfx950)"; }
const char* qgd_last_error(void) { return g_lastError.c_str(); }
int qgd_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ---- mesh -----------------------------------------------------------------------
int qgd_mesh_create(int32_t nPoints, const double* points, int32_t nFaces, const int32_t* faceOffsets, const int32_t* facePoints,
                    int32_t nInternalFaces, const int32_t* owner, const int32_t* neighbour, int32_t nCells, int32_t nPatches,
                    const int32_t* patchStart, const int32_t* patchSize, const int32_t* patchType, qgd_mesh_t* out) {
    QGD_TRY
    if (!points || !faceOffsets || !facePoints || !owner || (!neighbour && nInternalFaces > 0) || !out)
        return fail(QGD_ERR_INVALID, "qgd_mesh_create: null argument");
    if (nPoints <= 0 || nFaces <= 0 || nCells <= 0 || nInternalFaces < 0 || nInternalFaces > nFaces || nPatches < 0)
        return fail(QGD_ERR_INVALID, "qgd_mesh_create: bad sizes");
    qgd_mesh_s* h = new qgd_mesh_s();
    HostMesh& m = h->m;
    m.nPoints = nPoints; m.nFaces = nFaces; m.nInternalFaces = nInternalFaces; m.nCells = nCells;
    m.points.assign(points, points + 3 * (size_t)nPoints);
    m.faceOffsets.assign(faceOffsets, faceOffsets + nFaces + 1);
    m.facePoints.assign(facePoints, facePoints + faceOffsets[nFaces]);
    m.owner.assign(owner, owner + nFaces);
    if (nInternalFaces) m.neighbour.assign(neighbour, neighbour + nInternalFaces);
    for (int i = 0; i < nPatches; ++i) {
        Patch p;
        p.name = "patch" + std::to_string(i);
        p.type = patchType[i]; p.start = patchStart[i]; p.size = patchSize[i];
        m.patches.push_back(p);
    }
    std::string err = m.check();
    if (!err.empty()) { delete h; return fail(QGD_ERR_INVALID, "qgd_mesh_create: " + err); }
    m.computeGeometry();
    *out = h;
    return QGD_OK;
    QGD_CATCH
}

int qgd_mesh_box(int32_t nx, int32_t ny, int32_t nzGlobal, int32_t kLo, int32_t kHi, const double lo[3], const double hi[3],
                 const int32_t patchTypes[6], qgd_mesh_t* out) {
    QGD_TRY
    if (!lo || !hi || !out) return fail(QGD_ERR_INVALID, "qgd_mesh_box: null argument");
    qgd_mesh_s* h = new qgd_mesh_s();
    try { h->m = makeBox(nx, ny, nzGlobal, kLo, kHi, lo, hi, patchTypes); }
    catch (...) { delete h; throw; }
    *out = h;
    return QGD_OK;
    QGD_CATCH
}

int qgd_mesh_forward_step(int32_t nx, int32_t ny, int32_t ixStep, int32_t iyStep, double lx, double ly, double lz, qgd_mesh_t* out) {
    QGD_TRY
    if (!out) return fail(QGD_ERR_INVALID, "qgd_mesh_forward_step: null argument");
    qgd_mesh_s* h = new qgd_mesh_s();
    try { h->m = makeForwardStep(nx, ny, ixStep, iyStep, lx, ly, lz); }
    catch (...) { delete h; throw; }
    *out = h;
    return QGD_OK;
    QGD_CATCH
}
int qgd_mesh_jitter(qgd_mesh_t m, double amplitude, uint64_t seed) {
    QGD_TRY
    if (!m) return fail(QGD_ERR_INVALID, "null mesh");
    jitterPoints(m->m, amplitude, seed);
    m->m.haloFaceH.clear();   // the cut faces' hQGDf of the unsharded mesh no longer describes these points: back to the local rule
    return QGD_OK;
    QGD_CATCH
}
int qgd_mesh_split_quads(qgd_mesh_t m, int32_t stride) {
    QGD_TRY
    if (!m) return fail(QGD_ERR_INVALID, "null mesh");
    splitQuads(m->m, stride);
    m->m.haloFaceH.clear();   // face lists changed: the per-halo-face values would be misaligned
    return QGD_OK;
    QGD_CATCH
}
int qgd_mesh_split_edges(qgd_mesh_t m, int32_t stride) {
    QGD_TRY
    if (!m) return fail(QGD_ERR_INVALID, "null mesh");
    splitEdges(m->m, stride);
    m->m.haloFaceH.clear();
    return QGD_OK;
    QGD_CATCH
}
int qgd_mesh_set_geometry(qgd_mesh_t mh, const double* Sf, const double* Cf, const double* C, const double* V) {
    QGD_TRY
    if (!mh || !Sf || !Cf || !C || !V) return fail(QGD_ERR_INVALID, "qgd_mesh_set_geometry: null argument");
    HostMesh& m = mh->m;
    m.Sf.assign(Sf, Sf + 3 * (size_t)m.nFaces);
    m.Cf.assign(Cf, Cf + 3 * (size_t)m.nFaces);
    m.C.assign(C, C + 3 * (size_t)m.nCells);
    m.V.assign(V, V + (size_t)m.nCells);
    for (int32_t f = 0; f < m.nFaces; ++f) {
        const double* S = &m.Sf[3 * (size_t)f];
        m.magSf[f] = std::sqrt(S[0] * S[0] + S[1] * S[1] + S[2] * S[2]);
    }
    m.userGeometry = true;
    m.haloFaceH.clear();      // computed from the library's own centres; the caller's geometry rules now
    m.computeDerived();
    return QGD_OK;
    QGD_CATCH
}
int qgd_mesh_set_degenerate_faces(qgd_mesh_t mh, int32_t n, const int32_t* faces) {
    QGD_TRY
    if (!mh || n < 0 || (n > 0 && !faces)) return fail(QGD_ERR_INVALID, "qgd_mesh_set_degenerate_faces: bad argument");
    HostMesh& m = mh->m;
    for (int32_t i = 0; i < n; ++i)
        if (faces[i] < 0 || faces[i] >= m.nFaces) return fail(QGD_ERR_INVALID, "qgd_mesh_set_degenerate_faces: face label out of range");
    m.degenerateFaces.assign(faces, faces + n);
    std::sort(m.degenerateFaces.begin(), m.degenerateFaces.end());
    m.degenerateFaces.erase(std::unique(m.degenerateFaces.begin(), m.degenerateFaces.end()), m.degenerateFaces.end());
    return QGD_OK;
    QGD_CATCH
}
int qgd_mesh_free(qgd_mesh_t m) {
    delete m;
    return QGD_OK;
}
int qgd_mesh_sizes(qgd_mesh_t mh, int64_t sizes[7]) {
    if (!mh || !sizes) return fail(QGD_ERR_INVALID, "null argument");
    const HostMesh& m = mh->m;
    sizes[0] = m.nPoints; sizes[1] = m.nFaces; sizes[2] = m.nInternalFaces; sizes[3] = m.nCells;
    sizes[4] = (int64_t)m.patches.size(); sizes[5] = (int64_t)m.facePoints.size(); sizes[6] = m.nGeometricD;
    return QGD_OK;
}
// outBytes < 0: size query, *(int64_t*)out receives the array's size in bytes
int qgd_mesh_get(qgd_mesh_t mh, const char* name, void* out, int64_t outBytes) {
    QGD_TRY
    if (!mh || !name || !out) return fail(QGD_ERR_INVALID, "null argument");
    const HostMesh& m = mh->m;
    const std::string s(name);
    const void* src = nullptr;
    size_t bytes = 0;
    std::vector<int32_t> tmp;
    auto D = [&](const auto& v) { src = v.data(); bytes = v.size() * sizeof(double); };
    auto I = [&](const std::vector<int32_t>& v) { src = v.data(); bytes = v.size() * sizeof(int32_t); };
    if (s == "points") D(m.points);
    else if (s == "faceOffsets") I(m.faceOffsets);
    else if (s == "facePoints") I(m.facePoints);
    else if (s == "owner") I(m.owner);
    else if (s == "neighbour") I(m.neighbour);
    else if (s == "patchStart" || s == "patchSize" || s == "patchType") {
        for (const Patch& p : m.patches) tmp.push_back(s == "patchStart" ? p.start : (s == "patchSize" ? p.size : p.type));
        I(tmp);
    } else if (s.rfind("haloGhost", 0) == 0 || s.rfind("haloSend", 0) == 0) {
        const bool ghost = s[4] == 'G';
        const int slot = std::atoi(s.c_str() + (ghost ? 9 : 8));
        const auto& lists = ghost ? m.haloGhost : m.haloSend;
        if (slot >= 0 && slot < (int)lists.size()) I(lists[slot]); else I(tmp);
    } else if (s == "haloPeer") I(m.haloPeer);
    else if (s == "haloSelf") I(m.haloSelf);
    else if (s == "cellGlobal") I(m.cellGlobal);
    else if (s == "faceGlobal") I(m.faceGlobal);
    else if (s == "pointGlobal") I(m.pointGlobal);
    else if (s == "haloFaceH") D(m.haloFaceH);
    else if (s == "degenerateFaces") I(m.degenerateFaces);
    else if (s == "Sf") D(m.Sf);
    else if (s == "magSf") D(m.magSf);
    else if (s == "Cf") D(m.Cf);
    else if (s == "C") D(m.C);
    else if (s == "V") D(m.V);
    else if (s == "weights") D(m.weights);
    else if (s == "deltaCoeffs") D(m.deltaCoeffs);
    else if (s == "nonOrthDeltaCoeffs") D(m.nonOrthDeltaCoeffs);
    else return fail(QGD_ERR_UNKNOWN_NAME, "qgd_mesh_get: unknown array " + s);
    if (outBytes < 0) { *static_cast<int64_t*>(out) = (int64_t)bytes; return QGD_OK; }
    if ((int64_t)bytes > outBytes) return fail(QGD_ERR_INVALID, "qgd_mesh_get: output too small for " + s);
    if (bytes) std::memcpy(out, src, bytes);
    return QGD_OK;
    QGD_CATCH
}

int qgd_mesh_renumber(qgd_mesh_t mh, const int32_t* newOfOld, int32_t* faceNewOfOld) {
    QGD_TRY
    if (!mh || !newOfOld) return fail(QGD_ERR_INVALID, "qgd_mesh_renumber: null argument");
    if (!mh->m.haloGhost.empty()) return fail(QGD_ERR_INVALID, "qgd_mesh_renumber: renumber before sharding");
    renumberCells(mh->m, newOfOld, faceNewOfOld);
    const std::string err = mh->m.check();
    if (!err.empty()) return fail(QGD_ERR_INVALID, "qgd_mesh_renumber: " + err);
    return QGD_OK;
    QGD_CATCH
}
int qgd_mesh_rcm_order(qgd_mesh_t mh, int32_t* newOfOld) {
    QGD_TRY
    if (!mh || !newOfOld) return fail(QGD_ERR_INVALID, "qgd_mesh_rcm_order: null argument");
    const std::vector<int32_t> order = cuthillMcKee(mh->m);
    std::memcpy(newOfOld, order.data(), sizeof(int32_t) * order.size());
    return QGD_OK;
    QGD_CATCH
}
int qgd_mesh_morton_order(qgd_mesh_t mh, int32_t* newOfOld) {
    QGD_TRY
    if (!mh || !newOfOld) return fail(QGD_ERR_INVALID, "qgd_mesh_morton_order: null argument");
    const std::vector<int32_t> order = mortonOrder(mh->m);
    std::memcpy(newOfOld, order.data(), sizeof(int32_t) * order.size());
    return QGD_OK;
    QGD_CATCH
}
int qgd_mesh_shard(qgd_mesh_t global, int32_t nRanks, const int32_t* cellStart, int32_t rank, qgd_mesh_t* out) {
    QGD_TRY
    if (!global || !cellStart || !out) return fail(QGD_ERR_INVALID, "qgd_mesh_shard: null argument");
    if (nRanks < 1 || rank < 0 || rank >= nRanks) return fail(QGD_ERR_INVALID, "qgd_mesh_shard: rank out of range");
    if (!global->m.haloGhost.empty()) return fail(QGD_ERR_INVALID, "qgd_mesh_shard: the mesh is already a shard");
    if (cellStart[0] != 0 || cellStart[nRanks] != global->m.nCells) return fail(QGD_ERR_INVALID, "qgd_mesh_shard: cellStart must run from 0 to nCells");
    for (int r = 0; r < nRanks; ++r)
        if (cellStart[r + 1] <= cellStart[r]) return fail(QGD_ERR_INVALID, "qgd_mesh_shard: every rank needs at least one cell");
    qgd_mesh_s* h = new qgd_mesh_s();
    try {
        h->m = extractShard(global->m, nRanks, cellStart, rank);
        const std::string err = h->m.check();
        if (!err.empty()) { delete h; return fail(QGD_ERR_INVALID, "qgd_mesh_shard: " + err); }
    } catch (...) { delete h; throw; }
    *out = h;
    return QGD_OK;
    QGD_CATCH
}
int qgd_mesh_unroll_cyclic(qgd_mesh_t mesh, int32_t nPairs, const int32_t* pairs, qgd_mesh_t* out) {
    QGD_TRY
    if (!mesh || !out || (nPairs > 0 && !pairs)) return fail(QGD_ERR_INVALID, "qgd_mesh_unroll_cyclic: null argument");
    std::vector<std::pair<int32_t, int32_t>> pr;
    if (nPairs > 0) for (int32_t k = 0; k < nPairs; ++k) pr.push_back({pairs[2 * k], pairs[2 * k + 1]});
    else {   // consecutive cyclic patches with faces are the two halves of a pair (how blockMesh / createPatch write them)
        std::vector<int32_t> cyc;
        for (size_t i = 0; i < mesh->m.patches.size(); ++i) if (mesh->m.patches[i].type == QGD_PATCH_CYCLIC && mesh->m.patches[i].size > 0) cyc.push_back((int32_t)i);
        if (cyc.empty() || cyc.size() % 2) return fail(QGD_ERR_INVALID, "qgd_mesh_unroll_cyclic: the mesh has no cyclic patches, or an odd number of them");
        for (size_t i = 0; i < cyc.size(); i += 2) pr.push_back({cyc[i], cyc[i + 1]});
    }
    qgd_mesh_s* h = new qgd_mesh_s();
    try {
        h->m = unrollCyclic(mesh->m, pr);
        const std::string err = h->m.check();
        if (!err.empty()) { delete h; return fail(QGD_ERR_INVALID, "qgd_mesh_unroll_cyclic: " + err); }
    } catch (const std::invalid_argument& e) { delete h; return fail(QGD_ERR_NOT_IMPLEMENTED, e.what()); }
    catch (...) { delete h; throw; }
    *out = h;
    return QGD_OK;
    QGD_CATCH
}
int qgd_mesh_halo_slots(qgd_mesh_t mh, int32_t* nSlots) {
    if (!mh || !nSlots) return fail(QGD_ERR_INVALID, "null argument");
    *nSlots = (int32_t)mh->m.haloGhost.size();
    return QGD_OK;
}

// ---- device ----------------------------------------------------------------------
static int deviceCreate(qgd_mesh_t mh, int deviceId, int fusedChoice, qgd_device_t* out);
static double nowMs();
int qgd_device_create(qgd_mesh_t mh, int deviceId, qgd_device_t* out) { return deviceCreate(mh, deviceId, -1, out); }
int qgd_device_create_with(qgd_mesh_t mh, int deviceId, int32_t flags, qgd_device_t* out) {
    if (flags & ~(QGD_DEVICE_NO_FUSED_TABLES | QGD_DEVICE_FUSED_ANY_BLOCKS)) return fail(QGD_ERR_INVALID, "qgd_device_create_with: unknown flag");
    if ((flags & QGD_DEVICE_NO_FUSED_TABLES) && (flags & QGD_DEVICE_FUSED_ANY_BLOCKS))
        return fail(QGD_ERR_INVALID, "qgd_device_create_with: QGD_DEVICE_NO_FUSED_TABLES and QGD_DEVICE_FUSED_ANY_BLOCKS exclude each other");
    return deviceCreate(mh, deviceId, (flags & QGD_DEVICE_NO_FUSED_TABLES) ? 0 : ((flags & QGD_DEVICE_FUSED_ANY_BLOCKS) ? 2 : -1), out);
}
// fusedChoice: -1 = QGD_FUSED of the environment (default 1), 0 = no block tables, 2 = blocks of any size
static int deviceCreate(qgd_mesh_t mh, int deviceId, int fusedChoice, qgd_device_t* out) {
    QGD_TRY
    if (!mh || !out) return fail(QGD_ERR_INVALID, "qgd_device_create: null argument");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(QGD_ERR_NO_DEVICE, "qgd_device_create: no HIP device (this library has no CPU fallback)");
    if (deviceId < 0 || deviceId >= n) return fail(QGD_ERR_INVALID, "qgd_device_create: bad device id");
    HIP_CHECK(hipSetDevice(deviceId));
    const HostMesh& m = mh->m;
    if (m.patches.size() > QGD_MAX_PATCHES) return fail(QGD_ERR_INVALID, "too many patches");
    // QGD_SETUP_TIMING=1: where the device's set-up time goes, stage by stage, on stderr
    const bool stageTiming = std::getenv("QGD_SETUP_TIMING") && std::atoi(std::getenv("QGD_SETUP_TIMING")) != 0;
    double stageT0 = nowMs();
    auto stage = [&](const char* what) {
        if (!stageTiming) return;
        const double t = nowMs();
        std::fprintf(stderr, "qgd_device_create: %-44s %8.2f s\n", what, (t - stageT0) * 1e-3);
        stageT0 = t;
    };
    StaticData s = buildStaticData(m);
    stage("static tables on the host (buildStaticData)");
    qgd_device_s* d = new qgd_device_s();
    try {
        d->deviceId = deviceId;
        d->nGeomD = m.nGeometricD;
        d->hasTri = s.hasTri;
        d->wedgePrism = hasWedgeAndPrism(m);
        d->patches = m.patches;
        residentCasePatchCheck(m, d->caseRefusal, d->caseRefusalCode);
        d->hf.assign(s.hf.begin(), s.hf.end());
        const bool range = m.ownedEnd > m.ownedBegin;
        d->ownedBegin = range ? m.ownedBegin : 0;
        d->ownedEnd = range ? m.ownedEnd : m.nCells;
        d->cellGlobal = m.cellGlobal;
        d->cellGlobalOffset = m.cellGlobalOffset;
        for (int32_t c = 0; c < m.nCells && !m.cellIsGhost.empty(); ++c)
            if ((m.cellIsGhost[c] != 0) == (c >= d->ownedBegin && c < d->ownedEnd))
                throw std::invalid_argument("qgd_device_create: the owned cells of a shard must be one contiguous label range");
        HIP_CHECK(hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking));
        DeviceArena& a = d->arena;
        MeshView& v = d->view;
        v.nP = s.nP; v.nF = s.nF; v.nIF = s.nIF; v.nC = s.nC; v.nBF = s.nBF;
        v.ie1 = s.ie1; v.ie2 = s.ie2; v.ie3 = s.ie3;
        v.nGeomD = m.nGeometricD;
        for (int k = 0; k < 3; ++k) v.emptyDir[k] = m.geometricD[k] < 0 ? 1 : 0;
        static const int kBlocks[] = {64, 128, 256}, kWaves[] = {2, 3, 4};
        v.xcdRun = envChoice("QGD_XCD_RUN", 16, nullptr, 0, 0, 1 << 20);   // 0: one contiguous eighth of the tiles per XCD
        v.fuXcdRun = envChoice("QGD_FU_XCD_RUN", 64, nullptr, 0, 0, 1 << 20);   // (measured at 64 M cells: 4: 9.63, 16: 9.51-9.53, 64: 9.44, 256: 9.35-9.38 on one box; 16: 9.39, 64: 9.29-9.36, 1024: 9.38 on another)
        v.fblock = envChoice("QGD_FBLOCK", 128, kBlocks, 3);
        v.hasOther = 0;
        for (int64_t f = 0; f < s.nIF; ++f) if (s.fkind[f] == FK_OTHER) { v.hasOther = 1; break; }
        // Sf of a quadrilateral = (p3-p1) x (p4-p2) / 2 exactly (the triangle fan about any centre sums to it), and the
        // kernel holds those differences already: 24 B per face less to stream.  Not when the caller supplied its own Sf.
        static const int kOnOff[] = {0, 1};
        v.tileWaves = envChoice("QGD_FT_WAVES", 3, kWaves, 3);
        v.sGeo = (envChoice("QGD_SGEO", 1, kOnOff, 2) != 0 && !m.userGeometry && v.fblock == 128 && v.tileWaves == 3) ? 1 : 0;
        v.cblock = envChoice("QGD_CBLOCK", 256, kBlocks, 3);
        v.pblock = envChoice("QGD_PBLOCK", 256, kBlocks, 3);
        // upload + free each table in turn so the host peak stays at one table
        auto up = [&](auto& vec) { auto* p = a.upload(vec); std::decay_t<decltype(vec)>().swap(vec); return p; };
        {
            // face tiles of the LDS-staged 3-D GaussVolPoint kernel (QGD_FTILE=0: the gather kernel)
            const char* e = std::getenv("QGD_FTILE");
            FaceTiles t;
            if (!e || std::atoi(e) != 0) t = buildFaceTiles(s, v.fblock);
            const int64_t lds = ((int64_t)t.maxCells * 104 + (int64_t)t.maxVerts * 72 + 255) / 256 * 256;
            // a numbering under which a quarter of the tiles do not fit (reverse Cuthill-McKee levels, scrambled labels) is
            // better off with the gather kernel alone: 2.68 instead of 2.97 ms on the 16 M-cell irregular mesh in RCM order
            const int64_t nTiles = t.fb ? (s.nIF + t.fb - 1) / t.fb : 0;
            if (t.fb != 0 && lds <= 65536 && 4 * (int64_t)t.spill.size() <= nTiles) {
                v.tileLds = (int32_t)lds;
                v.tileMaxC = t.maxCells; v.tileMaxV = t.maxVerts;
                v.qhdTiles = envChoice("QGD_QHD_TILES", 1, kOnOff, 2);
                v.implTiles = envChoice("QGD_IMPL_TILES", 1, kOnOff, 2);
                if (v.fblock == 128 && v.tileWaves == 3 && envChoice("QGD_FTILE_FIXED", 1, kOnOff, 2) != 0 &&
                    (int64_t)nTiles * std::max(t.maxCells, t.maxVerts) < (int64_t)INT32_MAX) {
                    // the lists once more at a fixed stride (built and uploaded one after the other: the host peak stays at one table)
                    std::vector<uint8_t> flag((size_t)nTiles, 0);
                    for (int32_t tile : t.spill) flag[tile] = 1;
                    auto padded = [&](const std::vector<int32_t>& list, int which, int32_t stride) {
                        std::vector<int32_t> out((size_t)nTiles * stride, 0);
#pragma omp parallel for schedule(static)
                        for (int64_t tile = 0; tile < nTiles; ++tile) {
                            const int32_t b = t.off[2 * tile + which], e = t.off[2 * (tile + 1) + which];
                            if (e == b) continue;
                            int32_t* o = out.data() + (size_t)tile * stride;
                            for (int32_t i = 0; i < stride; ++i) o[i] = list[b + std::min(i, e - b - 1)];
                        }
                        return out;
                    };
                    { auto pc = padded(t.cells, 0, t.maxCells); v.tileCellsFix = up(pc); }
                    { auto pv = padded(t.verts, 1, t.maxVerts); v.tileVertsFix = up(pv); }
                    v.tileFlag = up(flag);
                }
                v.nTileSpill = (int32_t)t.spill.size(); v.tileSpill = up(t.spill);
                v.tileOff = up(t.off); v.tileCells = up(t.cells); v.tileVerts = up(t.verts);
                v.locC = up(t.locC); v.locV = reinterpret_cast<const uint2*>(up(t.locV));
            }
        }
        {
            // cell blocks of the fused face + cell kernel (QGD_FUSED, qgd_setup.hpp FusedBlocks): 3-D unsharded meshes whose tiles were built
            static const int kFusedModes[] = {0, 1, 2};
            // 2: whatever the blocks look like (tests, probes); the caller's explicit choice (qgd_device_create_with) wins over the environment
            const int fusedMode = fusedChoice >= 0 ? fusedChoice : envChoice("QGD_FUSED", 1, kFusedModes, 3);
            if (fusedMode != 0) {
                stage("face tiles (build + upload)");
                FusedBlocks fb = buildFusedBlocks(s);
                stage("cell blocks of the fused step (build)");
                // LDS per workgroup: RecA of every staged cell, RecB of the own + across-a-face cells, then vertex records + all coordinates,
                // later overwritten by the fluxes (FusedBlocks::maxLds); then the parked face entries of the own cells
                const int64_t ldsRec = fb.maxLds;   // the block that needs most; every block lays its records out by its own counts
                const int64_t ldsPark = (ldsRec + 15) / 16 * 16;
                const int64_t lds = (ldsPark + 6 * 128 * 4 + 255) / 256 * 256;
                // (a mesh whose blocks come out small -- under 88 cells on average: every block costs a workgroup two face passes whatever it
                // holds -- is better off with the three kernels: 5.0 against 4.4 ms per step on the 16 M-cell mesh of config 5 with 67-cell blocks)
                int64_t owned = 0;
                for (int64_t ci = 0; ci < s.nC; ++ci) owned += (s.ghost.empty() || s.ghost[ci] != 1) ? 1 : 0;
                // what one workgroup may ask for without raising the kernel's dynamic-LDS attribute: the device's own figure (64 KB on gfx950;
                // blocks of <= 32 cells are exempt from the builder's three-per-CU budget, so one fat polyhedral block could exceed it -- such
                // a mesh keeps the three kernels instead of failing in its first step)
                hipDeviceProp_t prop;
                HIP_CHECK(hipGetDeviceProperties(&prop, deviceId));
                const int64_t ldsLimit = std::min<int64_t>((int64_t)prop.sharedMemPerBlock, 64 * 1024);
                if (fb.nBlocks > 0 && (fusedMode == 2 || (int64_t)fb.nBlocks * 88 <= owned + 87) && fb.maxAll <= kFusedCapC && fb.maxTot <= kFusedCapTot &&
                    fb.capV <= kFusedCapV && fb.capF <= kFusedCapF && fb.capPE <= 255 && lds <= ldsLimit) {
                    v.fuBlocks = fb.nBlocks; v.fuLayerBlocks = fb.nLayerBlocks; v.fuCapC = fb.capC; v.fuCapV = fb.capV; v.fuCapF = fb.capF;
                    v.fuCapE = fb.capE; v.fuLds = (int32_t)lds; v.fuLdsCell = (int32_t)(ldsPark / 8);
                    v.fuCapPE = fb.capPE; v.fuMaxTot = fb.maxTot; v.fuMaxAll = fb.maxAll; v.fuMaxV = fb.maxV; v.fuMaxF = fb.maxF;
                    d->fusedFacesComputed = fb.facesComputed; d->fusedCellsStaged = fb.cellsStaged; d->fusedCellsStagedFull = fb.cellsStagedFull;
                    d->fusedVertsStaged = fb.vertsStaged;
                    v.fuHdr2 = reinterpret_cast<const int4*>(up(fb.hdr2)); v.fuVCount = up(fb.vCount); v.fuVPos = up(fb.vPos); v.fuVW = up(fb.vW);
                    v.fuHdr = reinterpret_cast<const int4*>(up(fb.hdr)); v.fuCells = up(fb.cells); v.fuVerts = up(fb.verts);
                    v.fuFaceLabel = up(fb.faceLabel); v.fuFacePos = up(fb.facePos); v.fuNEntry = up(fb.nEntry); v.fuEntry = up(fb.entry);
                    v.fuTemplates = fb.nTemplates;
                    {   // the implicitDiffusion branch's layout of the same blocks (qgd_kernels.hip fusedFaceCellKernel<..., IMPL>)
                        const int64_t parkI = ((int64_t)fb.maxLdsImpl + 15) / 16 * 16, ldsI = (parkI + 6 * 128 * 4 + 255) / 256 * 256;
                        v.fuLdsImpl = ldsI <= ldsLimit ? (int32_t)ldsI : 0;   // (on a box: the explicit step's figure, three blocks per CU)
                        v.fuLdsCellImpl = (int32_t)(parkI / 8);
                    }
                    d->fusedInfo[0] = fb.nBlocks; d->fusedInfo[1] = fb.nLayerBlocks; d->fusedInfo[2] = fb.nTemplates; d->fusedInfo[3] = lds;
                    // bytes a block streams per step out of its own lists / of its template's
                    d->fusedInfo[4] = 32 + 4 * (int64_t)fb.capC + 4 * (int64_t)fb.capV + 4 * (int64_t)fb.capF + fb.capV + 128 + 8 * (int64_t)fb.capPE * fb.capV;
                    d->fusedInfo[5] = 12 * (int64_t)fb.capF + 4 * (int64_t)fb.capE * 128 + 2 * (int64_t)fb.capPE * fb.capV;
                    d->fusedInfo[6] = (int64_t)(fb.buildSeconds * 1e3);
                    d->fusedInfo[7] = fb.brick[0] | fb.brick[1] << 8 | fb.brick[2] << 16;
                }
            }
        }
        stage("block tables (upload) / up to the static tables");
        v.own = up(s.own); v.nei = up(s.nei);
        v.verts = reinterpret_cast<const int4*>(up(s.verts));
        v.fkind = up(s.fkind);
        v.Sx = up(s.Sf[0]); v.Sy = up(s.Sf[1]); v.Sz = up(s.Sf[2]);
        v.magSf = up(s.magSf); v.w = up(s.w); v.hf = up(s.hf); v.dn = up(s.dn);
        v.X = up(s.X); v.Cc = up(s.Cc);
        v.bN = reinterpret_cast<const double4*>(up(s.bN)); v.bmvON = up(s.bmvON);
        v.ip13 = reinterpret_cast<const int2*>(up(s.ip13)); v.c2d = up(s.c2d);
        v.lsqSlice = up(s.lsqSlice); v.lsqCnt = up(s.lsqCnt); v.lsqCell = up(s.lsqCell);
        v.lsqGx = up(s.lsqGx); v.lsqGy = up(s.lsqGy); v.lsqGz = up(s.lsqGz); v.lsqDeg = up(s.lsqDeg);
        v.lsqBndZero = up(s.lsqBndZero);
        if (!s.bSymm.empty()) v.bSymm = up(s.bSymm);
        v.pcSlice = up(s.pcSlice); v.pcCount = up(s.pcCount); v.pcCell = up(s.pcCell); v.pcW = up(s.pcW);
        v.nBP = (int32_t)s.bpPoint.size();
        v.bpPoint = up(s.bpPoint); v.bpOff = up(s.bpOff); v.bpFace = up(s.bpFace); v.bpW = up(s.bpW);
        if (!s.cpOff.empty() && s.cpOff.back() > 0) { v.cpOff = up(s.cpOff); v.cpKind = up(s.cpKind); v.cpT = up(s.cpT); }
        v.cfSlice = up(s.cfSlice); v.cfCount = up(s.cfCount); v.cfItem = up(s.cfItem); v.cfNbr = up(s.cfNbr);
        v.fpos = up(s.fpos); v.cfPos = up(s.cfPos);
        v.V = up(s.V); v.hQGD = up(s.hQGD); v.ghost = up(s.ghost);
        v.bPatch = up(s.bPatch); v.hQGDb = up(s.hQGDb);
        {
            // a cell on a shard corner is needed by several neighbours: once in the list the update kernel walks
            std::vector<int32_t> sc, sf;
            for (size_t slot = 0; slot < s.haloSend.size(); ++slot) {
                sc.insert(sc.end(), s.haloSend[slot].begin(), s.haloSend[slot].end());
                sf.insert(sf.end(), s.haloSendBF[slot].begin(), s.haloSendBF[slot].end());
            }
            std::sort(sc.begin(), sc.end()); sc.erase(std::unique(sc.begin(), sc.end()), sc.end());
            std::sort(sf.begin(), sf.end()); sf.erase(std::unique(sf.begin(), sf.end()), sf.end());
            d->nSendAll = (int32_t)sc.size(); d->nSendBFAll = (int32_t)sf.size();
            d->sendAll = a.upload(sc); d->sendBFAll = a.upload(sf);
        }
        d->haloSelf = m.haloSelf;
        if (!d->haloSelf.empty() && d->haloSelf.size() != s.haloGhost.size()) throw std::invalid_argument("qgd_device_create: haloSelf does not match the halo slots");
        d->halo.resize(s.haloGhost.size());
        for (size_t slot = 0; slot < d->halo.size(); ++slot) {
            qgd_device_s::HaloSlot& h = d->halo[slot];
            h.nGhost = (int32_t)s.haloGhost[slot].size();
            h.nSend = (int32_t)s.haloSend[slot].size();
            h.nGhostBF = (int32_t)s.haloGhostBF[slot].size();
            h.nSendBF = (int32_t)s.haloSendBF[slot].size();
            h.ghost = a.upload(s.haloGhost[slot]);
            h.send = a.upload(s.haloSend[slot]);
            h.ghostBF = a.upload(s.haloGhostBF[slot]);
            h.sendBF = a.upload(s.haloSendBF[slot]);
        }
    } catch (...) {
        d->arena.release();
        if (d->stream) (void)hipStreamDestroy(d->stream);
        delete d;
        throw;
    }
    stage("static tables (upload), halo lists");
    *out = d;
    return QGD_OK;
    QGD_CATCH
}
int qgd_device_free(qgd_device_t d) {
    if (!d) return QGD_OK;
    // a case keeps a pointer to its device and runs on its stream: freed after the device it would read freed memory
    if (d->liveCases > 0)
        return fail(QGD_ERR_INVALID, "qgd_device_free: " + std::to_string(d->liveCases) + " case(s) created on this device are still open; free them first");
    (void)hipSetDevice(d->deviceId);
    if (d->opSolver) { (void)hipStreamSynchronize(d->stream); implicitSolverFree(d->opSolver); d->opSolver = nullptr; }
    d->ws.release();
    for (hipEvent_t e : d->opEv) if (e) (void)hipEventDestroy(e);
    d->arena.release();
    if (d->stream) (void)hipStreamDestroy(d->stream);
    delete d;
    return QGD_OK;
}

// prismMatcher + findIndices("wedge") of fvscOpName [fvsc_8C L65-82]: a wedge patch and at least one prism cell
// (five faces: two triangles and three quadrilaterals, six vertices)
static bool hasWedgeAndPrism(const HostMesh& m) {
    bool wedge = false;
    for (const Patch& p : m.patches) wedge = wedge || (p.type == QGD_PATCH_WEDGE && p.size > 0);
    if (!wedge) return false;
    std::vector<uint8_t> nTri((size_t)m.nCells, 0), nQuad((size_t)m.nCells, 0), nOther((size_t)m.nCells, 0);
    auto count = [&](int32_t c, int n) { if (n == 3) nTri[c]++; else if (n == 4) nQuad[c]++; else nOther[c]++; };
    for (int32_t f = 0; f < m.nFaces; ++f) {
        count(m.owner[f], m.faceSize(f));
        if (f < m.nInternalFaces) count(m.neighbour[f], m.faceSize(f));
    }
    for (int32_t c = 0; c < m.nCells; ++c)
        if (nTri[c] == 2 && nQuad[c] == 3 && nOther[c] == 0) return true;
    return false;
}
static const char* kWedgePrismMessage =
    "GaussVolPoint scheme does not support solving axisymmetric cases with wedge BC and prism cells. Try to set leastSquares scheme.";

// fvscOpName + fvscStencil::New [fvsc_8C L47-85, fvscStencil_8C L59-95]
static int stencilWordToId(int nGeomD, const std::string& w, int* id) {
    if ((w == "leastSquares" || w == "leastSquaresOpt") && nGeomD == 3)
        return fail(QGD_ERR_SCHEME, "Can't use leastSquares or leastSquaresOpt in 3D case.");
    if (w == "reduced") *id = QGD_FVSC_REDUCED;
    else if (w == "leastSquares" || w == "leastSquaresOpt") *id = QGD_FVSC_LEASTSQUARES;
    else if (w == "GaussVolPoint") *id = QGD_FVSC_GAUSSVOLPOINT;
    else return fail(QGD_ERR_UNKNOWN_NAME, "Unknown Model type " + w + "; valid: GaussVolPoint leastSquares leastSquaresOpt reduced");
    return QGD_OK;
}
static int deviceStencilRaw(int nGeomD, int id, int* st);
static int deviceStencil(const qgd_device_s* d, int id, int* st) {
    if (d->wedgePrism && id == QGD_FVSC_GAUSSVOLPOINT) return fail(QGD_ERR_SCHEME, kWedgePrismMessage);
    return deviceStencilRaw(d->nGeomD, id, st);
}
static int deviceStencilRaw(int nGeomD, int id, int* st) {
    if (id == QGD_FVSC_REDUCED) *st = ST_REDUCED;
    else if (id == QGD_FVSC_LEASTSQUARES) {
        if (nGeomD == 3) return fail(QGD_ERR_SCHEME, "Can't use leastSquares or leastSquaresOpt in 3D case.");
        *st = ST_LSQ;
    } else if (id == QGD_FVSC_GAUSSVOLPOINT) *st = (nGeomD == 3) ? ST_GVP3 : (nGeomD == 2 ? ST_GVP2 : ST_REDUCED);
    else return fail(QGD_ERR_UNKNOWN_NAME, "bad stencil id");
    return QGD_OK;
}
int qgd_stencil_lookup(qgd_device_t d, const char* word, int* stencilId) {
    if (!d || !word || !stencilId) return fail(QGD_ERR_INVALID, "null argument");
    if (d->wedgePrism && std::string(word) == "GaussVolPoint") return fail(QGD_ERR_SCHEME, kWedgePrismMessage);
    return stencilWordToId(d->nGeomD, word, stencilId);
}

enum WsSlot { WS_CELL = 0, WS_BND, WS_PT, WS_OUT, WS_A, WS_B, WS_C, WS_D, WS_E, WS_F, WS_G, WS_H };
static double nowMs() {
    timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return 1e3 * (double)t.tv_sec + 1e-6 * (double)t.tv_nsec;
}
static int fvscOp(qgd_device_t d, int stencilId, int op, int NC, const double* cell, const double* bnd, double* out) {
    QGD_TRY
    if (!d || !cell || !out || (!bnd && d->view.nBF > 0)) return fail(QGD_ERR_INVALID, "fvsc operator: null argument");
    int st = 0;
    int rc = deviceStencil(d, stencilId, &st);
    if (rc) return rc;
    HIP_CHECK(hipSetDevice(d->deviceId));
    const MeshView& v = d->view;
    const int NO = (op == 0) ? 3 * NC : NC / 3;
    Workspace& ws = d->ws;
    double* dc = ws.get<double>(WS_CELL, (size_t)v.nC * NC);
    double* db = ws.get<double>(WS_BND, (size_t)v.nBF * NC);
    double* dp = ws.get<double>(WS_PT, (size_t)v.nP * NC);
    double* dout = ws.get<double>(WS_OUT, (size_t)v.nF * NO);
    if (!d->opEv[0]) { HIP_CHECK(hipEventCreate(&d->opEv[0])); HIP_CHECK(hipEventCreate(&d->opEv[1])); }
    const double t0 = nowMs();
    ws.h2d(dc, cell, sizeof(double) * (size_t)v.nC * NC, d->stream);
    if (v.nBF) ws.h2d(db, bnd, sizeof(double) * (size_t)v.nBF * NC, d->stream);
    HIP_CHECK(hipMemsetAsync(dp, 0, sizeof(double) * (size_t)v.nP * NC, d->stream));
    HIP_CHECK(hipStreamSynchronize(d->stream));
    const double t1 = nowMs();
    (void)hipGetLastError();  // drop any stale sticky error so the check below is about our launches
    HIP_CHECK(hipEventRecord(d->opEv[0], d->stream));
    launchFvscOp(d->stream, st, op, NC, v, dc, db, dp, dout);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipEventRecord(d->opEv[1], d->stream));
    HIP_CHECK(hipEventSynchronize(d->opEv[1]));
    const double t2 = nowMs();
    ws.d2h(out, dout, sizeof(double) * (size_t)v.nF * NO, d->stream);
    HIP_CHECK(hipStreamSynchronize(d->stream));
    float kms = 0;
    (void)hipEventElapsedTime(&kms, d->opEv[0], d->opEv[1]);
    d->opMs[0] = t1 - t0; d->opMs[1] = kms; d->opMs[2] = nowMs() - t2;
    return QGD_OK;
    QGD_CATCH
}
int qgd_device_op_times(qgd_device_t d, double ms[3]) {
    if (!d || !ms) return fail(QGD_ERR_INVALID, "null argument");
    for (int k = 0; k < 3; ++k) ms[k] = d->opMs[k];
    return QGD_OK;
}
int qgd_device_face_tiles(qgd_device_t d, int64_t info[4]) {
    if (!d || !info) return fail(QGD_ERR_INVALID, "null argument");
    const MeshView& v = d->view;
    const bool on = v.tileOff != nullptr;
    info[0] = on ? v.fblock : 0;
    info[1] = on ? (v.nIF + v.fblock - 1) / v.fblock : 0;
    info[2] = on ? v.nTileSpill : 0;
    info[3] = on ? v.tileLds : 0;
    return QGD_OK;
}
int qgd_device_fused_blocks(qgd_device_t d, int64_t info[8]) {
    if (!d || !info) return fail(QGD_ERR_INVALID, "null argument");
    for (int k = 0; k < 8; ++k) info[k] = d->fusedInfo[k];
    return QGD_OK;
}
int qgd_fvsc_grad_s(qgd_device_t d, int id, const double* cell, const double* bnd, double* out) { return fvscOp(d, id, 0, 1, cell, bnd, out); }
int qgd_fvsc_grad_v(qgd_device_t d, int id, const double* cell, const double* bnd, double* out) { return fvscOp(d, id, 0, 3, cell, bnd, out); }
int qgd_fvsc_div_v(qgd_device_t d, int id, const double* cell, const double* bnd, double* out) { return fvscOp(d, id, 1, 3, cell, bnd, out); }
int qgd_fvsc_div_t(qgd_device_t d, int id, const double* cell, const double* bnd, double* out) { return fvscOp(d, id, 1, 9, cell, bnd, out); }

// ---- the same operators on DEVICE pointers: no staging, no PCIe, stream-ordered on the device handle's stream ----------------
// (a level-1 adapter whose fields already live in HBM -- another GPU library, a device-resident OpenFOAM build -- pays the
// kernels only: ~3 ms instead of ~80 ms per four gradients at 8 M cells)
static int fvscOpDev(qgd_device_t d, int stencilId, int op, int NC, const double* cell, const double* bnd, double* out) {
    QGD_TRY
    if (!d || !cell || !out || (!bnd && d->view.nBF > 0)) return fail(QGD_ERR_INVALID, "fvsc operator (device pointers): null argument");
    int st = 0;
    int rc = deviceStencil(d, stencilId, &st);
    if (rc) return rc;
    HIP_CHECK(hipSetDevice(d->deviceId));
    const MeshView& v = d->view;
    double* dp = d->ws.get<double>(WS_PT, (size_t)v.nP * NC);   // vertex values: the one work array (grow-only, reused)
    HIP_CHECK(hipMemsetAsync(dp, 0, sizeof(double) * (size_t)v.nP * NC, d->stream));
    (void)hipGetLastError();
    launchFvscOp(d->stream, st, op, NC, v, cell, bnd, dp, out);
    HIP_CHECK(hipGetLastError());
    return QGD_OK;
    QGD_CATCH
}
int qgd_fvsc_grad_s_dev(qgd_device_t d, int id, const double* cell, const double* bnd, double* out) { return fvscOpDev(d, id, 0, 1, cell, bnd, out); }
int qgd_fvsc_grad_v_dev(qgd_device_t d, int id, const double* cell, const double* bnd, double* out) { return fvscOpDev(d, id, 0, 3, cell, bnd, out); }
int qgd_fvsc_div_v_dev(qgd_device_t d, int id, const double* cell, const double* bnd, double* out) { return fvscOpDev(d, id, 1, 3, cell, bnd, out); }
int qgd_fvsc_div_t_dev(qgd_device_t d, int id, const double* cell, const double* bnd, double* out) { return fvscOpDev(d, id, 1, 9, cell, bnd, out); }
int qgd_interpolate_dev(qgd_device_t d, int32_t ncomp, const double* cell, const double* bnd, double* out) {
    QGD_TRY
    if (!d || !cell || !out || ncomp < 1 || ncomp > 9 || (!bnd && d->view.nBF > 0)) return fail(QGD_ERR_INVALID, "qgd_interpolate_dev: bad argument");
    HIP_CHECK(hipSetDevice(d->deviceId));
    (void)hipGetLastError();
    launchInterpolate(d->stream, ncomp, d->view, cell, bnd, out);
    HIP_CHECK(hipGetLastError());
    return QGD_OK;
    QGD_CATCH
}
int qgd_device_sync(qgd_device_t d) {
    QGD_TRY
    if (!d) return fail(QGD_ERR_INVALID, "null device");
    HIP_CHECK(hipSetDevice(d->deviceId));
    HIP_CHECK(hipStreamSynchronize(d->stream));
    return QGD_OK;
    QGD_CATCH
}
// the leastSquares stencil of an internal face as the device holds it (sliced ELL read back): cells in the order they are summed
int qgd_device_lsq_stencil(qgd_device_t d, int32_t face, int32_t* cells, int32_t cap, int32_t* count) {
    QGD_TRY
    if (!d || !count || (cap > 0 && !cells)) return fail(QGD_ERR_INVALID, "qgd_device_lsq_stencil: bad argument");
    const MeshView& m = d->view;
    if (!m.lsqCnt || !m.lsqSlice || !m.lsqCell) return fail(QGD_ERR_SCHEME, "qgd_device_lsq_stencil: the mesh has no leastSquares stencil (3-D)");
    if (face < 0 || face >= m.nIF) return fail(QGD_ERR_INVALID, "qgd_device_lsq_stencil: not an internal face");
    HIP_CHECK(hipSetDevice(d->deviceId));
    HIP_CHECK(hipStreamSynchronize(d->stream));
    uint8_t cnt8 = 0;
    int32_t slice = 0;
    HIP_CHECK(hipMemcpy(&cnt8, m.lsqCnt + face, sizeof(uint8_t), hipMemcpyDeviceToHost));
    const int32_t cnt = cnt8;
    HIP_CHECK(hipMemcpy(&slice, m.lsqSlice + (face >> 6), sizeof(int32_t), hipMemcpyDeviceToHost));
    for (int k = 0; k < cnt && k < cap; ++k)
        HIP_CHECK(hipMemcpy(cells + k, m.lsqCell + ((size_t)slice + k) * 64 + (face & 63), sizeof(int32_t), hipMemcpyDeviceToHost));
    *count = cnt;
    return QGD_OK;
    QGD_CATCH
}
int qgd_device_copy(qgd_device_t d, void* dst, const void* src, int64_t bytes, int toDevice) {
    QGD_TRY
    if (!d || !dst || !src || bytes < 0) return fail(QGD_ERR_INVALID, "qgd_device_copy: bad argument");
    HIP_CHECK(hipSetDevice(d->deviceId));
    // on the device's own stream and complete at return: that stream is non-blocking, and a hipMemcpy from pageable memory on the null
    // stream may return while its staged copy is still on the way -- a kernel queued on d->stream right after could read the old bytes
    if (bytes) HIP_CHECK(hipMemcpyAsync(dst, src, (size_t)bytes, toDevice ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost, d->stream));
    HIP_CHECK(hipStreamSynchronize(d->stream));
    return QGD_OK;
    QGD_CATCH
}
// qgd_qhd_fluxes on device pointers; outputs in the documented face-major layout
int qgd_qhd_fluxes_dev(qgd_device_t d, int stencilId, const qgd_qhd_inputs* in, qgd_qhd_outputs* out) {
    QGD_TRY
    if (!d || !in || !out) return fail(QGD_ERR_INVALID, "qgd_qhd_fluxes_dev: null argument");
    const MeshView& v = d->view;
    if (!in->U || !in->T || !in->rho || !in->tauQGDf || (v.nBF > 0 && (!in->Ub || !in->Tb || !in->rhob)))
        return fail(QGD_ERR_INVALID, "qgd_qhd_fluxes_dev: U, T, rho (cell + patch values) and tauQGDf are required");
    const bool haveP = in->p != nullptr;
    if (haveP && v.nBF > 0 && !in->pb) return fail(QGD_ERR_INVALID, "qgd_qhd_fluxes_dev: p given without its patch values");
    if ((out->gradPf || out->Wf || out->phiUf) && !haveP) return fail(QGD_ERR_INVALID, "qgd_qhd_fluxes_dev: gradPf/Wf/phiUf need p");
    if ((out->phiUf || out->phiTf) && !in->phi) return fail(QGD_ERR_INVALID, "qgd_qhd_fluxes_dev: phiUf/phiTf need phi");
    int st = 0;
    int rc = deviceStencil(d, stencilId, &st);
    if (rc) return rc;
    HIP_CHECK(hipSetDevice(d->deviceId));
    const size_t nC = (size_t)v.nC, nB = (size_t)v.nBF, nF = (size_t)v.nF, nP = (size_t)v.nP;
    Workspace& ws = d->ws;
    hipStream_t s_ = d->stream;
    double* dCell = ws.get<double>(WS_CELL, 5 * nC);
    double* dBnd = ws.get<double>(WS_BND, 5 * std::max<size_t>(nB, 1));
    double* dPt = ws.get<double>(WS_PT, 5 * nP);
    double* dOut = ws.get<double>(WS_OUT, (size_t)QHD_COUNT * nF);
    (void)hipGetLastError();
    launchPack5(s_, (int64_t)nC, in->U, in->T, in->p, dCell);
    launchPack5(s_, (int64_t)nB, in->Ub, in->Tb, haveP ? in->pb : nullptr, dBnd);
    HIP_CHECK(hipMemsetAsync(dPt, 0, sizeof(double) * std::max<size_t>(5 * nP, 1), s_));
    HIP_CHECK(hipMemsetAsync(dOut, 0, sizeof(double) * (size_t)QHD_COUNT * nF, s_));
    launchQhdFluxes(s_, st, v, dCell, dBnd, dPt, in->rho, in->rhob, in->tauQGDf, in->phi, in->beta, in->g[0], in->g[1], in->g[2], dOut);
    auto fetch = [&](double* dst, int first, int nc) { if (dst) launchSoaToAos(s_, (int64_t)nF, nc, dOut + (size_t)first * nF, dst); };
    fetch(out->gradUf, QHD_GRADU, 9); fetch(out->gradTf, QHD_GRADT, 3); fetch(out->phiu, QHD_PHIU, 1);
    fetch(out->phiwo, QHD_PHIWO, 1); fetch(out->taubyrhof, QHD_TAUBYRHO, 1); fetch(out->gradPf, QHD_GRADP, 3);
    fetch(out->Wf, QHD_WF, 3); fetch(out->phiUf, QHD_PHIUF, 3); fetch(out->phiTf, QHD_PHITF, 1);
    fetch(out->phiTauTReg, QHD_PHITAUT, 1);
    HIP_CHECK(hipGetLastError());
    return QGD_OK;
    QGD_CATCH
}
int qgd_species_flux_dev(qgd_device_t d, int stencilId, const double* Y, const double* Yb, const double* U, const double* Ub,
                         const double* phiJm, const double* phi, const double* tauQGDf, double* phiJmY, double* diffusiveFlux,
                         double* gradYf) {
    QGD_TRY
    if (!d || !Y || !U || !phiJm || !phi || !tauQGDf || !phiJmY || !diffusiveFlux)
        return fail(QGD_ERR_INVALID, "qgd_species_flux_dev: null argument");
    const MeshView& v = d->view;
    if (v.nBF > 0 && (!Yb || !Ub)) return fail(QGD_ERR_INVALID, "qgd_species_flux_dev: patch values of Y and U are required");
    int st = 0;
    int rc = deviceStencil(d, stencilId, &st);
    if (rc) return rc;
    HIP_CHECK(hipSetDevice(d->deviceId));
    const size_t nF = (size_t)v.nF, nP = (size_t)v.nP;
    hipStream_t s_ = d->stream;
    double* dPt = d->ws.get<double>(WS_PT, nP);
    double* dOut = d->ws.get<double>(WS_OUT, 5 * nF);
    HIP_CHECK(hipMemsetAsync(dPt, 0, sizeof(double) * std::max<size_t>(nP, 1), s_));
    (void)hipGetLastError();
    launchSpeciesFlux(s_, st, v, Y, Yb, dPt, U, Ub, phiJm, phi, tauQGDf, dOut);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemcpyAsync(phiJmY, dOut, sizeof(double) * nF, hipMemcpyDeviceToDevice, s_));
    HIP_CHECK(hipMemcpyAsync(diffusiveFlux, dOut + nF, sizeof(double) * nF, hipMemcpyDeviceToDevice, s_));
    if (gradYf) launchSoaToAos(s_, (int64_t)nF, 3, dOut + 2 * nF, gradYf);
    HIP_CHECK(hipGetLastError());
    return QGD_OK;
    QGD_CATCH
}

int qgd_species_step_dev(qgd_device_t d, const double* Y, const double* Yb, const double* rhoOld, const double* rho, const double* phiJmY,
                         const double* muf, double Sc, double deltaT, const double* Su, double* diffusiveFlux, double* Ynew) {
    QGD_TRY
    if (!d || !Y || !rhoOld || !rho || !phiJmY || !muf || !diffusiveFlux || !Ynew) return fail(QGD_ERR_INVALID, "qgd_species_step_dev: null argument");
    if (!(Sc > 0) || !(deltaT > 0)) return fail(QGD_ERR_INVALID, "qgd_species_step_dev: Sc and deltaT must be positive");
    const MeshView& v = d->view;
    if (v.nBF > 0 && !Yb) return fail(QGD_ERR_INVALID, "qgd_species_step_dev: patch values of Y are required");
    HIP_CHECK(hipSetDevice(d->deviceId));
    double* net = d->ws.get<double>(WS_OUT, (size_t)v.nF);
    (void)hipGetLastError();
    launchSpeciesStep(d->stream, v, Y, Yb, rhoOld, rho, phiJmY, muf, Sc, deltaT, Su, diffusiveFlux, net, Ynew);
    HIP_CHECK(hipGetLastError());
    return QGD_OK;
    QGD_CATCH
}
int qgd_species_step(qgd_device_t d, const double* Y, const double* Yb, const double* rhoOld, const double* rho, const double* phiJmY,
                     const double* muf, double Sc, double deltaT, const double* Su, double* diffusiveFlux, double* Ynew) {
    QGD_TRY
    if (!d || !Y || !rhoOld || !rho || !phiJmY || !muf || !diffusiveFlux || !Ynew) return fail(QGD_ERR_INVALID, "qgd_species_step: null argument");
    if (!(Sc > 0) || !(deltaT > 0)) return fail(QGD_ERR_INVALID, "qgd_species_step: Sc and deltaT must be positive");
    const MeshView& v = d->view;
    if (v.nBF > 0 && !Yb) return fail(QGD_ERR_INVALID, "qgd_species_step: patch values of Y are required");
    HIP_CHECK(hipSetDevice(d->deviceId));
    const size_t nC = (size_t)v.nC, nB = (size_t)v.nBF, nF = (size_t)v.nF;
    Workspace& ws = d->ws;
    hipStream_t st_ = d->stream;
    int nextSlot = WS_CELL;
    auto upD = [&](const double* src, size_t n) {
        double* dst = ws.get<double>((size_t)nextSlot++, n);
        if (src && n) ws.h2d(dst, src, sizeof(double) * n, st_);
        return dst;
    };
    double *dY = upD(Y, nC), *dYb = upD(Yb, nB), *dRo = upD(rhoOld, nC), *dR = upD(rho, nC), *dJ = upD(phiJmY, nF), *dMu = upD(muf, nF);
    double* dSu = Su ? upD(Su, nC) : nullptr;
    double *dDf = upD(diffusiveFlux, nF), *dNet = upD(nullptr, nF), *dNew = upD(nullptr, nC);
    (void)hipGetLastError();
    launchSpeciesStep(st_, v, dY, dYb, dRo, dR, dJ, dMu, Sc, deltaT, dSu, dDf, dNet, dNew);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipStreamSynchronize(st_));
    ws.d2h(diffusiveFlux, dDf, sizeof(double) * nF, st_);
    ws.d2h(Ynew, dNew, sizeof(double) * nC, st_);
    return QGD_OK;
    QGD_CATCH
}

// the implicitDiffusion branch of the species equation [QGDYEqn_8H L47-66] on device pointers; info (host) = {iterations, initial, final residual}
int qgd_species_step_implicit_dev(qgd_device_t d, const double* Y, const double* Yb, const uint8_t* fixedValueFace, const double* rhoOld,
                                  const double* rho, const double* phiJmY, const double* muf, double Sc, double deltaT, const double* Su, double tolerance,
                                  int32_t maxIter, double* diffusiveFlux, double* Ynew, double info[3]) {
    QGD_TRY
    if (!d || !Y || !rhoOld || !rho || !phiJmY || !muf || !diffusiveFlux || !Ynew || !info)
        return fail(QGD_ERR_INVALID, "qgd_species_step_implicit_dev: null argument");
    if (!(Sc > 0) || !(deltaT > 0) || !(tolerance > 0) || maxIter < 1)
        return fail(QGD_ERR_INVALID, "qgd_species_step_implicit_dev: Sc, deltaT, tolerance must be positive, maxIter >= 1");
    const MeshView& v = d->view;
    if (v.nBF > 0 && !Yb) return fail(QGD_ERR_INVALID, "qgd_species_step_implicit_dev: patch values of Y are required");
    if (d->sharded()) return fail(QGD_ERR_NOT_IMPLEMENTED, "qgd_species_step_implicit: the stateless operator solves on one device (no reductions across shards)");
    for (const Patch& pt : d->patches)   // fvm::laplacian couples the two sides of a cyclic patch inside the matrix; this one has no such rows
        if (pt.type == QGD_PATCH_CYCLIC && pt.nonEmptyGlobally())
            return fail(QGD_ERR_NOT_IMPLEMENTED, "qgd_species_step_implicit: patch '" + pt.name + "' is cyclic: the implicit laplacian across coupled patches is not served");
    HIP_CHECK(hipSetDevice(d->deviceId));
    if (!d->opSolver) d->opSolver = implicitSolverCreate(d->stream, v, d->ownedBegin, d->ownedEnd);
    double* work = d->ws.get<double>(WS_H, 3 * (size_t)v.nC + (size_t)v.nF);   // the last slot: the host-pointer entry fills the first ten
    (void)hipGetLastError();
    launchSpeciesStepImplicit(d->opSolver, v, Y, Yb, fixedValueFace, rhoOld, rho, phiJmY, muf, Sc, deltaT, Su, tolerance, maxIter, work, diffusiveFlux, Ynew, info);
    HIP_CHECK(hipGetLastError());
    return QGD_OK;
    QGD_CATCH
}
int qgd_species_step_implicit(qgd_device_t d, const double* Y, const double* Yb, const uint8_t* fixedValueFace, const double* rhoOld, const double* rho,
                              const double* phiJmY, const double* muf, double Sc, double deltaT, const double* Su, double tolerance, int32_t maxIter,
                              double* diffusiveFlux, double* Ynew, double info[3]) {
    QGD_TRY
    if (!d || !Y || !rhoOld || !rho || !phiJmY || !muf || !diffusiveFlux || !Ynew || !info)
        return fail(QGD_ERR_INVALID, "qgd_species_step_implicit: null argument");
    const MeshView& v = d->view;
    if (v.nBF > 0 && !Yb) return fail(QGD_ERR_INVALID, "qgd_species_step_implicit: patch values of Y are required");
    HIP_CHECK(hipSetDevice(d->deviceId));
    const size_t nC = (size_t)v.nC, nB = (size_t)v.nBF, nF = (size_t)v.nF;
    Workspace& ws = d->ws;
    hipStream_t st_ = d->stream;
    int nextSlot = WS_CELL;
    auto upD = [&](const double* src, size_t n) {
        double* dst = ws.get<double>((size_t)nextSlot++, n);
        if (src && n) ws.h2d(dst, src, sizeof(double) * n, st_);
        return dst;
    };
    double *dY = upD(Y, nC), *dYb = upD(Yb, nB), *dRo = upD(rhoOld, nC), *dR = upD(rho, nC), *dJ = upD(phiJmY, nF), *dMu = upD(muf, nF);
    double* dSu = Su ? upD(Su, nC) : nullptr;
    double *dDf = upD(diffusiveFlux, nF), *dNew = upD(nullptr, nC);
    uint8_t* dFix = nullptr;
    if (fixedValueFace && nB) {
        dFix = reinterpret_cast<uint8_t*>(ws.get<double>((size_t)nextSlot++, (nB + 7) / 8));
        ws.h2d(dFix, fixedValueFace, nB, st_);
    }
    const int rc = qgd_species_step_implicit_dev(d, dY, dYb, dFix, dRo, dR, dJ, dMu, Sc, deltaT, dSu, tolerance, maxIter, dDf, dNew, info);
    if (rc) return rc;
    HIP_CHECK(hipStreamSynchronize(st_));
    ws.d2h(diffusiveFlux, dDf, sizeof(double) * nF, st_);
    ws.d2h(Ynew, dNew, sizeof(double) * nC, st_);
    return QGD_OK;
    QGD_CATCH
}

int qgd_interpolate(qgd_device_t d, int32_t ncomp, const double* cell, const double* bnd, double* out) {
    QGD_TRY
    if (!d || !cell || !out || ncomp < 1 || ncomp > 9 || (!bnd && d->view.nBF > 0)) return fail(QGD_ERR_INVALID, "qgd_interpolate: bad argument");
    HIP_CHECK(hipSetDevice(d->deviceId));
    const MeshView& v = d->view;
    Workspace& ws = d->ws;
    double* dc = ws.get<double>(WS_CELL, (size_t)v.nC * ncomp);
    double* db = ws.get<double>(WS_BND, (size_t)v.nBF * ncomp);
    double* dout = ws.get<double>(WS_OUT, (size_t)v.nF * ncomp);
    ws.h2d(dc, cell, sizeof(double) * (size_t)v.nC * ncomp, d->stream);
    if (v.nBF) ws.h2d(db, bnd, sizeof(double) * (size_t)v.nBF * ncomp, d->stream);
    (void)hipGetLastError();
    launchInterpolate(d->stream, ncomp, v, dc, db, dout);
    HIP_CHECK(hipGetLastError());
    ws.d2h(out, dout, sizeof(double) * (size_t)v.nF * ncomp, d->stream);
    HIP_CHECK(hipStreamSynchronize(d->stream));
    return QGD_OK;
    QGD_CATCH
}
int qgd_flux(qgd_device_t d, int32_t ncomp, const double* flux, const double* psif, double* out) {
    if (!d || !flux || !psif || !out || ncomp < 1) return fail(QGD_ERR_INVALID, "qgd_flux: bad argument");
    const int64_t nF = d->view.nF;
    for (int64_t f = 0; f < nF; ++f)
        for (int k = 0; k < ncomp; ++k) out[f * ncomp + k] = flux[f] * psif[f * ncomp + k];  // flux*psif [QGDInterpolate_8H L104]
    return QGD_OK;
}
int qgd_flux_upwind(qgd_device_t d, int32_t ncomp, const double* flux, const double* cell, const double* bnd, double* out) {
    QGD_TRY
    if (!d || !flux || !cell || !out || ncomp < 1 || ncomp > 9 || (!bnd && d->view.nBF > 0)) return fail(QGD_ERR_INVALID, "qgd_flux_upwind: bad argument");
    HIP_CHECK(hipSetDevice(d->deviceId));
    const MeshView& v = d->view;
    Workspace& ws = d->ws;
    double* dc = ws.get<double>(WS_CELL, (size_t)v.nC * ncomp);
    double* db = ws.get<double>(WS_BND, (size_t)v.nBF * ncomp);
    double* df = ws.get<double>(WS_A, (size_t)v.nF);
    double* dout = ws.get<double>(WS_OUT, (size_t)v.nF * ncomp);
    ws.h2d(dc, cell, sizeof(double) * (size_t)v.nC * ncomp, d->stream);
    if (v.nBF) ws.h2d(db, bnd, sizeof(double) * (size_t)v.nBF * ncomp, d->stream);
    ws.h2d(df, flux, sizeof(double) * (size_t)v.nF, d->stream);
    (void)hipGetLastError();
    launchFluxUpwind(d->stream, ncomp, v, df, dc, db, dout);
    HIP_CHECK(hipGetLastError());
    ws.d2h(out, dout, sizeof(double) * (size_t)v.nF * ncomp, d->stream);
    HIP_CHECK(hipStreamSynchronize(d->stream));
    return QGD_OK;
    QGD_CATCH
}
int qgd_device_get(qgd_device_t d, const char* name, double* out, int64_t outDoubles) {
    QGD_TRY
    if (!d || !name || !out) return fail(QGD_ERR_INVALID, "qgd_device_get: null argument");
    HIP_CHECK(hipSetDevice(d->deviceId));
    const MeshView& v = d->view;
    const std::string s(name);
    if (s == "hQGDf") {
        if ((int64_t)d->hf.size() > outDoubles) return fail(QGD_ERR_INVALID, "output too small");
        std::copy(d->hf.begin(), d->hf.end(), out);
    } else if (s == "hQGD") {
        if (v.nC > outDoubles) return fail(QGD_ERR_INVALID, "output too small");
        HIP_CHECK(hipMemcpy(out, v.hQGD, sizeof(double) * (size_t)v.nC, hipMemcpyDeviceToHost));
    } else if (s == "hQGD.boundary") {
        if (v.nBF > outDoubles) return fail(QGD_ERR_INVALID, "output too small");
        if (v.nBF) HIP_CHECK(hipMemcpy(out, v.hQGDb, sizeof(double) * (size_t)v.nBF, hipMemcpyDeviceToHost));
    } else return fail(QGD_ERR_UNKNOWN_NAME, "qgd_device_get: unknown name " + s);
    return QGD_OK;
    QGD_CATCH
}

// ---- QHDFoam face fluxes -------------------------------------------------------------
int qgd_qhd_fluxes(qgd_device_t d, int stencilId, const qgd_qhd_inputs* in, qgd_qhd_outputs* out) {
    QGD_TRY
    if (!d || !in || !out) return fail(QGD_ERR_INVALID, "qgd_qhd_fluxes: null argument");
    const MeshView& v = d->view;
    if (!in->U || !in->T || !in->rho || !in->tauQGDf || (v.nBF > 0 && (!in->Ub || !in->Tb || !in->rhob)))
        return fail(QGD_ERR_INVALID, "qgd_qhd_fluxes: U, T, rho (cell + patch values) and tauQGDf are required");
    const bool haveP = in->p != nullptr;
    if (haveP && v.nBF > 0 && !in->pb) return fail(QGD_ERR_INVALID, "qgd_qhd_fluxes: p given without its patch values");
    if ((out->gradPf || out->Wf || out->phiUf) && !haveP) return fail(QGD_ERR_INVALID, "qgd_qhd_fluxes: gradPf/Wf/phiUf need p");
    if ((out->phiUf || out->phiTf) && !in->phi) return fail(QGD_ERR_INVALID, "qgd_qhd_fluxes: phiUf/phiTf need phi");
    int st = 0;
    int rc = deviceStencil(d, stencilId, &st);
    if (rc) return rc;
    HIP_CHECK(hipSetDevice(d->deviceId));
    const size_t nC = (size_t)v.nC, nB = (size_t)v.nBF, nF = (size_t)v.nF, nP = (size_t)v.nP;
    // host-side packing of the 5-component records {Ux,Uy,Uz,T,p}
    std::vector<double> cell5(5 * nC), bnd5(5 * std::max<size_t>(nB, 1), 0.0);
    for (size_t c = 0; c < nC; ++c) {
        for (int k = 0; k < 3; ++k) cell5[5 * c + k] = in->U[3 * c + k];
        cell5[5 * c + 3] = in->T[c];
        cell5[5 * c + 4] = haveP ? in->p[c] : 0.0;
    }
    for (size_t b = 0; b < nB; ++b) {
        for (int k = 0; k < 3; ++k) bnd5[5 * b + k] = in->Ub[3 * b + k];
        bnd5[5 * b + 3] = in->Tb[b];
        bnd5[5 * b + 4] = haveP ? in->pb[b] : 0.0;
    }
    Workspace& ws = d->ws;
    hipStream_t st_ = d->stream;
    auto up = [&](int slot, const double* src, size_t n) {
        double* dst = ws.get<double>((size_t)slot, n);
        if (src && n) ws.h2d(dst, src, sizeof(double) * n, st_);
        return dst;
    };
    double* dCell = up(WS_CELL, cell5.data(), cell5.size());
    double* dBnd = up(WS_BND, bnd5.data(), bnd5.size());
    double* dPt = ws.get<double>(WS_PT, 5 * nP);
    HIP_CHECK(hipMemsetAsync(dPt, 0, sizeof(double) * std::max<size_t>(5 * nP, 1), st_));
    double* dRho = up(WS_A, in->rho, nC);
    double* dRhob = up(WS_B, in->rhob, nB);
    double* dTau = up(WS_C, in->tauQGDf, nF);
    double* dPhi = in->phi ? up(WS_D, in->phi, nF) : nullptr;
    double* dOut = ws.get<double>(WS_OUT, (size_t)QHD_COUNT * nF);
    HIP_CHECK(hipMemsetAsync(dOut, 0, sizeof(double) * (size_t)QHD_COUNT * nF, st_));
    (void)hipGetLastError();
    launchQhdFluxes(st_, st, v, dCell, dBnd, dPt, dRho, dRhob, dTau, dPhi, in->beta, in->g[0], in->g[1], in->g[2], dOut);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipStreamSynchronize(st_));
    std::vector<double> slot(nF);
    auto fetch = [&](double* dst, int first, int nc) {
        if (!dst) return;
        if (nc == 1) { ws.d2h(dst, dOut + (size_t)first * nF, sizeof(double) * nF, st_); return; }
        for (int k = 0; k < nc; ++k) {
            ws.d2h(slot.data(), dOut + (size_t)(first + k) * nF, sizeof(double) * nF, st_);
#pragma omp parallel for schedule(static)
            for (int64_t f = 0; f < (int64_t)nF; ++f) dst[(size_t)f * nc + k] = slot[(size_t)f];
        }
    };
    fetch(out->gradUf, QHD_GRADU, 9); fetch(out->gradTf, QHD_GRADT, 3); fetch(out->phiu, QHD_PHIU, 1);
    fetch(out->phiwo, QHD_PHIWO, 1); fetch(out->taubyrhof, QHD_TAUBYRHO, 1); fetch(out->gradPf, QHD_GRADP, 3);
    fetch(out->Wf, QHD_WF, 3); fetch(out->phiUf, QHD_PHIUF, 3); fetch(out->phiTf, QHD_PHITF, 1);
    fetch(out->phiTauTReg, QHD_PHITAUT, 1);
    return QGD_OK;
    QGD_CATCH
}

int qgd_species_flux(qgd_device_t d, int stencilId, const double* Y, const double* Yb, const double* U, const double* Ub,
                     const double* phiJm, const double* phi, const double* tauQGDf, double* phiJmY, double* diffusiveFlux,
                     double* gradYf) {
    QGD_TRY
    if (!d || !Y || !U || !phiJm || !phi || !tauQGDf || !phiJmY || !diffusiveFlux)
        return fail(QGD_ERR_INVALID, "qgd_species_flux: null argument");
    const MeshView& v = d->view;
    if (v.nBF > 0 && (!Yb || !Ub)) return fail(QGD_ERR_INVALID, "qgd_species_flux: patch values of Y and U are required");
    int st = 0;
    int rc = deviceStencil(d, stencilId, &st);
    if (rc) return rc;
    HIP_CHECK(hipSetDevice(d->deviceId));
    const size_t nC = (size_t)v.nC, nB = (size_t)v.nBF, nF = (size_t)v.nF, nP = (size_t)v.nP;
    Workspace& ws = d->ws;
    hipStream_t st_ = d->stream;
    int nextSlot = WS_CELL;
    auto upD = [&](const double* src, size_t n) {
        double* dst = ws.get<double>((size_t)nextSlot++, n);
        if (src && n) ws.h2d(dst, src, sizeof(double) * n, st_);
        return dst;
    };
    double *dY = upD(Y, nC), *dYb = upD(Yb, nB), *dU = upD(U, 3 * nC), *dUb = upD(Ub, 3 * nB);
    double *dJm = upD(phiJm, nF), *dPhi = upD(phi, nF), *dTau = upD(tauQGDf, nF);
    double* dPt = upD(nullptr, nP);
    HIP_CHECK(hipMemsetAsync(dPt, 0, sizeof(double) * std::max<size_t>(nP, 1), st_));
    double* dOut = upD(nullptr, 5 * nF);
    (void)hipGetLastError();
    launchSpeciesFlux(st_, st, v, dY, dYb, dPt, dU, dUb, dJm, dPhi, dTau, dOut);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipStreamSynchronize(st_));
    ws.d2h(phiJmY, dOut, sizeof(double) * nF, st_);
    ws.d2h(diffusiveFlux, dOut + nF, sizeof(double) * nF, st_);
    if (gradYf) {
        std::vector<double> slot(nF);
        for (int k = 0; k < 3; ++k) {
            ws.d2h(slot.data(), dOut + (size_t)(2 + k) * nF, sizeof(double) * nF, st_);
            for (size_t f = 0; f < nF; ++f) gradYf[3 * f + k] = slot[f];
        }
    }
    return QGD_OK;
    QGD_CATCH
}

int qgd_poisson_control_default(qgd_poisson_control* c) {
    if (!c) return fail(QGD_ERR_INVALID, "null argument");
    c->tolerance = 1e-6; c->relTol = 0.0; c->maxIter = 1000; c->pRefCell = 0; c->pRefValue = 0.0;
    return QGD_OK;
}

int qgd_qhd_pressure(qgd_device_t d, const double* phiu, const double* phiwo, const double* taubyrhof, const int32_t* patchKind,
                     const double* pb, const double* gradb, const qgd_poisson_control* ctl, double* p, double* phi, double info[3]) {
    QGD_TRY
    if (!d || !phiu || !phiwo || !taubyrhof || !patchKind || !ctl || !p || !phi)
        return fail(QGD_ERR_INVALID, "qgd_qhd_pressure: null argument");
    const MeshView& v = d->view;
    const size_t nC = (size_t)v.nC, nB = (size_t)v.nBF, nF = (size_t)v.nF;
    if (d->sharded()) return fail(QGD_ERR_NOT_IMPLEMENTED, "qgd_qhd_pressure: the pressure equation is not distributed");
    if (!(ctl->tolerance >= 0) || ctl->maxIter < 0) return fail(QGD_ERR_INVALID, "qgd_qhd_pressure: bad solver controls");
    // per boundary face: 0 nothing to add (zeroGradient, constraint patches), 1 fixedValue, 2 fixedGradient
    std::vector<uint8_t> bKind(std::max<size_t>(nB, 1), 0);
    bool anyFixedValue = false;
    for (size_t ip = 0; ip < d->patches.size(); ++ip) {
        const Patch& pt = d->patches[ip];
        int kind = patchKind[ip];
        if (pt.type != QGD_PATCH_GENERIC) kind = QGD_BC_NONE;  // constraint patches carry no matrix contribution here
        uint8_t k = 0;
        if (kind == QGD_BC_FIXEDVALUE) { k = 1; anyFixedValue = anyFixedValue || pt.size > 0; if (!pb) return fail(QGD_ERR_INVALID, "qgd_qhd_pressure: fixedValue patch without pb"); }
        else if (kind == QGD_BC_QGDFLUX) { k = 2; if (!gradb) return fail(QGD_ERR_INVALID, "qgd_qhd_pressure: fixedGradient patch without gradb"); }
        else if (kind != QGD_BC_ZEROGRADIENT && kind != QGD_BC_NONE) return fail(QGD_ERR_INVALID, "qgd_qhd_pressure: unsupported patch kind");
        for (int32_t f = pt.start; f < pt.start + pt.size; ++f) bKind[(size_t)(f - v.nIF)] = k;
    }
    // fvMatrix::setReference acts only when the field needs a reference level (no fixedValue patch), L0
    int refCell = (!anyFixedValue && ctl->pRefCell >= 0) ? ctl->pRefCell : -1;
    if (refCell >= (int)nC) return fail(QGD_ERR_INVALID, "qgd_qhd_pressure: pRefCell out of range");
    HIP_CHECK(hipSetDevice(d->deviceId));
    DeviceArena tmp;
    try {
        auto upD = [&](const double* src, size_t n) {
            double* dst = tmp.alloc<double>(std::max<size_t>(n, 1), false);
            if (src && n) HIP_CHECK(hipMemcpy(dst, src, sizeof(double) * n, hipMemcpyHostToDevice));
            else { HIP_CHECK(hipMemset(dst, 0, sizeof(double) * std::max<size_t>(n, 1))); HIP_CHECK(hipStreamSynchronize(nullptr)); }
            return dst;
        };
        double* dGamma = upD(taubyrhof, nF);
        double* dPhiu = upD(phiu, nF);
        double* dPhiwo = upD(phiwo, nF);
        double* dPb = upD(pb, nB);
        double* dGb = upD(gradb, nB);
        double* dP = upD(p, nC);
        double* dPhi = tmp.alloc<double>(nF, false);
        uint8_t* dKind = tmp.upload(bKind);
        const size_t nBlocks = (nC + 255) / 256;
        double* work = tmp.alloc<double>(8 * nC + nF + std::max<size_t>(nB, 1) + 3 * nBlocks + 8, false);
        double res[2] = {0, 0};
        (void)hipGetLastError();
        const int iters = solveQhdPressure(d->stream, v, dGamma, dPhiu, dPhiwo, dKind, dPb, dGb, refCell, ctl->pRefValue, ctl->tolerance,
                                           ctl->relTol, ctl->maxIter, dP, dPhi, work, res);
        HIP_CHECK(hipMemcpy(p, dP, sizeof(double) * nC, hipMemcpyDeviceToHost));
        HIP_CHECK(hipMemcpy(phi, dPhi, sizeof(double) * nF, hipMemcpyDeviceToHost));
        if (info) { info[0] = iters; info[1] = res[0]; info[2] = res[1]; }
    } catch (...) { tmp.release(); throw; }
    tmp.release();
    return QGD_OK;
    QGD_CATCH
}

// ---- case ------------------------------------------------------------------------
int qgd_case_options_default(qgd_case_options* o) {
    if (!o) return fail(QGD_ERR_INVALID, "null argument");
    std::memset(o, 0, sizeof(*o));
    o->stencil = QGD_FVSC_GAUSSVOLPOINT;
    o->implicitDiffusion = 0;
    o->adjustTimeStep = 0;
    o->R = 1.0 / 1.4; o->Cv = (1.0 / 1.4) / 0.4;  // gamma = 1.4, c = 1 at T = 1
    o->mu = 0.0; o->Pr = 1.0;
    o->ScQGD = 1.0; o->PrQGD = 1.0; o->alphaQGD = 0.5;
    o->deltaT = 1e-4; o->maxCo = 0.5; o->maxDeltaT = 1.0; o->cTau = 0.75;
    o->implicitTol = 1e-10; o->implicitMaxIter = 1000;
    return QGD_OK;
}

int qgd_case_create(qgd_device_t d, const qgd_case_options* opt, qgd_case_t* out) {
    QGD_TRY
    if (!d || !opt || !out) return fail(QGD_ERR_INVALID, "qgd_case_create: null argument");
    if (opt->implicitDiffusion && (!(opt->implicitTol >= 0) || opt->implicitMaxIter < 0))
        return fail(QGD_ERR_INVALID, "qgd_case_create: bad implicitTol / implicitMaxIter");
    if (!(opt->R > 0) || !(opt->Cv > 0) || !(opt->Pr > 0) || !(opt->PrQGD > 0) || !(opt->deltaT > 0))
        return fail(QGD_ERR_INVALID, "qgd_case_create: R, Cv, Pr, PrQGD, deltaT must be positive");
    if ((opt->fluxSchemeU != QGD_FLUX_LINEAR && opt->fluxSchemeU != QGD_FLUX_UPWIND) || (opt->fluxSchemeH != QGD_FLUX_LINEAR && opt->fluxSchemeH != QGD_FLUX_UPWIND))
        return fail(QGD_ERR_INVALID, "qgd_case_create: fluxSchemeU / fluxSchemeH must be QGD_FLUX_LINEAR or QGD_FLUX_UPWIND");
    if (!d->caseRefusal.empty()) return fail(d->caseRefusalCode, "qgd_case_create: " + d->caseRefusal);
    int st = 0;
    int rc = deviceStencil(d, opt->stencil, &st);
    if (rc) return rc;
    HIP_CHECK(hipSetDevice(d->deviceId));
    qgd_case_s* c = new qgd_case_s();
    try {
        c->dev = d; c->opt = *opt; c->stencil = st;
        c->usesPoints = (st == ST_GVP3 || st == ST_GVP2);
        c->pRefresh = c->usesPoints;
        {   // per-term entries [fvsc_8C L51-58]: grad(U), grad(e), grad(rho), grad(p) -> components (Ux,Uy,Uz), e, rho, p of the face gradient
            static const int kBits[4] = {0x0e, 0x20, 0x01, 0x10};
            int term[4], lo = st, hi = st;
            for (int t = 0; t < 4; ++t) {
                term[t] = st;
                if (opt->termStencil[t] != 0) {
                    const int r2 = deviceStencil(d, opt->termStencil[t] - 1, &term[t]);   // the checks of fvscOpName apply per term
                    if (r2) { delete c; return r2; }
                }
                lo = std::min(lo, term[t]); hi = std::max(hi, term[t]);
            }
            for (int t = 0; t < 4; ++t)
                if (term[t] != lo && term[t] != hi) { delete c; return fail(QGD_ERR_NOT_IMPLEMENTED, "qgd_case_create: more than two distinct fvsc stencils in one case"); }
            if (lo != hi) {
                c->stencil = lo; c->mixB = hi; c->mixMask = 0;
                for (int t = 0; t < 4; ++t) if (term[t] == hi) c->mixMask |= kBits[t];
                c->usesPoints = (hi == ST_GVP3 || hi == ST_GVP2 || lo == ST_GVP3 || lo == ST_GVP2);
            } else c->stencil = lo;
            c->pRefresh = (term[3] == ST_GVP3 || term[3] == ST_GVP2);
        }
        GasModel& g = c->gas;
        g.R = opt->R; g.Cv = opt->Cv; g.mu0 = opt->mu; g.Pr = opt->Pr; g.ScQGD = opt->ScQGD; g.PrQGD = opt->PrQGD; g.rPrQGD = 1.0 / opt->PrQGD;
        g.alphaQGD = opt->alphaQGD;
        g.consistentEnergy = opt->consistentEnergy ? 1 : 0;
        g.implicitDiffusion = opt->implicitDiffusion ? 1 : 0;
        g.upwindU = opt->fluxSchemeU == QGD_FLUX_UPWIND ? 1 : 0;
        g.upwindH = opt->fluxSchemeH == QGD_FLUX_UPWIND ? 1 : 0;
        const double Cp = opt->Cv + opt->R;
        g.gamma = Cp / opt->Cv;
        const double rPr = 1.0 / opt->Pr;
        g.alphah0 = (Cp * opt->mu * rPr) / Cp;
        const MeshView& v = d->view;
        DeviceArena& a = c->arena;
        CaseView& cv = c->view;
        cv.A = a.alloc<RecA>(v.nC); cv.B = a.alloc<RecB>(v.nC); cv.rE = a.alloc<double>(v.nC);
        cv.P = a.alloc<RecA>(v.nP);
        // the fused step (fusedFaceCellKernel): uniform 3-D GaussVolPoint, explicit, fixed deltaT; `Gauss upwind` fluxes are an instantiation
        c->fused = v.fuBlocks > 0 && c->stencil == ST_GVP3 && c->mixB < 0 && !opt->implicitDiffusion && !opt->adjustTimeStep;
        if (c->fused) { cv.A2 = a.alloc<RecA>(v.nC); cv.B2 = a.alloc<RecB>(v.nC); }
        {   // the same blocks under Courant-number control: two launches (blocks up to their sums, then the cells) instead of three kernels
            static const int kOnOff[] = {0, 1};
            c->fusedAdj = v.fuBlocks > 0 && c->stencil == ST_GVP3 && c->mixB < 0 && !opt->implicitDiffusion && opt->adjustTimeStep &&
                          envChoice("QGD_FUSED_ADJUST", 1, kOnOff, 2) != 0;
            if (c->fusedAdj) cv.cellSum = a.alloc<double>(5 * (size_t)v.nC);
        }
        {   // the reference's default branch on the same blocks (unsharded, fixed deltaT, one 3-D GaussVolPoint stencil)
            static const int kOnOff[] = {0, 1};
            c->fusedImpl = v.fuBlocks > 0 && v.fuLdsImpl > 0 && c->stencil == ST_GVP3 && c->mixB < 0 && opt->implicitDiffusion && !opt->adjustTimeStep &&
                           !d->sharded() && envChoice("QGD_IMPL_FUSED", 1, kOnOff, 2) != 0 && fusedImplUPrepare(v, c->gas);
        }
        cv.bA = a.alloc<RecA>(v.nBF); cv.bB = a.alloc<RecB>(v.nBF);
        cv.bG = a.alloc<double>(v.nBF); cv.bPhiw = a.alloc<double>(v.nBF); cv.bPmid = a.alloc<double>(v.nBF);
        cv.bRhoLag = a.alloc<double>(v.nBF);
        cv.nBlkFace = faceBlocks(v) + bfaceBlocks(v);
        cv.nBlkCell = std::max(cellBlocks(v), v.fuBlocks) + (d->nSendAll + 63) / 64;   // (the fused kernel monitors min(rho), min(e) per block)
        cv.blkFace = a.alloc<double>(2 * (size_t)std::max(1, cv.nBlkFace));
        cv.blkFace2 = a.alloc<double>(2 * (size_t)QGD_FACE_REDUCE_PARTIALS);
        cv.blkCell = a.alloc<double>(2 * (size_t)std::max(1, cv.nBlkCell));
        cv.flux = a.alloc<double>(5 * (size_t)v.nF);
        cv.red = a.alloc<double>(8);
        cv.dt = a.alloc<double>(8);
        cv.dbg = nullptr;
        if (opt->implicitDiffusion) {
            ImplView& iv = c->impl;
            const size_t nC = (size_t)v.nC, nF = (size_t)v.nF;
            d->ensureFaceGeoPos();
            iv.gUc = a.alloc<double>(9 * nC);
            iv.phiTau = a.alloc<double>(3 * nF); iv.UfS = a.alloc<double>(3 * nF);
            iv.sTau = a.alloc<double>(nF); iv.mufS = a.alloc<double>(nF); iv.aU = a.alloc<double>(nF); iv.aE = a.alloc<double>(nF);
            iv.phiSig = a.alloc<double>(nF);
            iv.rhoNew = a.alloc<double>(nC);
            iv.xU = a.alloc<double>(3 * nC); iv.diagU = a.alloc<double>(3 * nC); iv.rhsU = a.alloc<double>(3 * nC);
            iv.xE = a.alloc<double>(nC); iv.diagE = a.alloc<double>(nC); iv.rhsE = a.alloc<double>(nC);
            c->implSolver = implicitSolverCreate(d->stream, v, d->ownedBegin, d->ownedEnd);
            { static const int kOnOff[] = {0, 1}; c->reuseGradU = envChoice("QGD_IMPL_REUSE_GRADU", 1, kOnOff, 2) != 0; }
            {   // start values of the two solves (qgd_implicit.hip "start values"): 0 = OpenFOAM's (the predictor), 1..3 = + the extrapolated correction
                static const int kOrders[] = {0, 1, 2, 3, 4};
                c->implXOrder = envChoice("QGD_IMPL_XEXTRAP", 3, kOrders, 5);
                if (c->implXOrder > 0) {
                    iv.pred = a.alloc<double>(4 * nC);
                    for (int j = 0; j < c->implXOrder; ++j) iv.dh[j] = a.alloc<double>(4 * nC);
                }
                iv.have = 0; iv.order = c->implXOrder;
                iv.w = nullptr; iv.dtHist = nullptr;
                if (c->implXOrder > 0 && opt->adjustTimeStep) { iv.w = a.alloc<double>(4); iv.dtHist = a.alloc<double>(8); }   // (zero-filled)
            }
        }
        c->bc.resize(d->patches.size());
        for (size_t i = 0; i < d->patches.size(); ++i) initPatchBC(c->bc[i], d->patches[i]);
        c->bcDev = a.alloc<PatchBCDev>(std::max<size_t>(1, c->bc.size()));
        const double dt0[8] = {opt->deltaT, 0, 0, 0, 0, 0, 0, 0};
        HIP_CHECK(hipMemcpy(cv.dt, dt0, sizeof(dt0), hipMemcpyHostToDevice));
    } catch (...) { c->arena.release(); delete c; throw; }
    d->liveCases++;
    *out = c;
    return QGD_OK;
    QGD_CATCH
}
int qgd_case_free(qgd_case_t c) {
    if (!c) return QGD_OK;
    (void)hipSetDevice(c->dev->deviceId);
    harvestTiming(c);
    for (hipEvent_t e : c->freeEvents) (void)hipEventDestroy(e);
    if (c->implSolver) { (void)hipStreamSynchronize(c->stream()); implicitSolverFree(c->implSolver); }
    if (c->ownHaloStream) { (void)hipStreamSynchronize(c->ownHaloStream); (void)hipStreamDestroy(c->ownHaloStream); }
    if (c->evLayerDone) (void)hipEventDestroy(c->evLayerDone);
    if (c->evUnpacked) (void)hipEventDestroy(c->evUnpacked);
    c->arena.release();
    c->dev->liveCases--;
    delete c;
    return QGD_OK;
}

int qgd_case_set_bc(qgd_case_t c, int32_t patch, int32_t bcU, const double* valueU, int32_t bcT, double valueT, int32_t bcP, double valueP) {
    QGD_TRY
    if (!c) return fail(QGD_ERR_INVALID, "null case");
    if (patch < 0 || patch >= (int32_t)c->bc.size()) return fail(QGD_ERR_INVALID, "qgd_case_set_bc: patch out of range");
    auto okU = [](int k) { return k == QGD_BC_ZEROGRADIENT || k == QGD_BC_FIXEDVALUE || k == QGD_BC_SLIP || k == QGD_BC_NONE; };
    auto okT = [](int k) { return k == QGD_BC_ZEROGRADIENT || k == QGD_BC_FIXEDVALUE || k == QGD_BC_NONE; };
    auto okP = [](int k) { return k == QGD_BC_ZEROGRADIENT || k == QGD_BC_FIXEDVALUE || k == QGD_BC_QGDFLUX || k == QGD_BC_NONE; };
    if (!okU(bcU) || !okT(bcT) || !okP(bcP)) return fail(QGD_ERR_INVALID, "qgd_case_set_bc: unsupported boundary-condition kind");
    PatchBCDev& b = c->bc[patch];
    constraintKinds(b.ptype, bcU, bcT, bcP);   // a constraint patch keeps its own field type whatever the caller asks for
    b.bcU = bcU; b.bcT = bcT; b.bcP = bcP; b.vT = valueT; b.vP = valueP;
    if (valueU) for (int k = 0; k < 3; ++k) b.vU[k] = valueU[k];
    c->fieldsSet = false;
    c->gradUValid = false;
    return QGD_OK;
    QGD_CATCH
}

int qgd_case_set_qgd_coeffs(qgd_case_t c, const double* alphaQGD, const double* alphaQGDb, const double* ScQGD, const double* ScQGDb) {
    QGD_TRY
    if (!c) return fail(QGD_ERR_INVALID, "null case");
    if (c->dev->view.nBF > 0 && ((alphaQGD && !alphaQGDb) || (ScQGD && !ScQGDb)))
        return fail(QGD_ERR_INVALID, "qgd_case_set_qgd_coeffs: a field needs both its cell and its patch values");
    HIP_CHECK(hipSetDevice(c->dev->deviceId));
    const MeshView& m = c->dev->view;
    auto put = [&](const double* src, size_t n, double*& slot) -> const double* {
        if (!src) return nullptr;
        if (!slot) slot = c->arena.alloc<double>(std::max<size_t>(n, 1), false);
        if (n) HIP_CHECK(hipMemcpy(slot, src, sizeof(double) * n, hipMemcpyHostToDevice));
        return slot;
    };
    c->view.aQ = put(alphaQGD, (size_t)m.nC, c->coef[0]);
    c->view.aQb = put(alphaQGD ? alphaQGDb : nullptr, (size_t)m.nBF, c->coef[1]);
    c->view.sc = put(ScQGD, (size_t)m.nC, c->coef[2]);
    c->view.scb = put(ScQGD ? ScQGDb : nullptr, (size_t)m.nBF, c->coef[3]);
    c->fieldsSet = false;
    c->gradUValid = false;
    return QGD_OK;
    QGD_CATCH
}

// one flux-assembly pass (updateFields.H + updateFluxes.H) on the current state; part 0 = all of it, 1 = up to p's mid-step boundary
// conditions, 2 = the rest.  A shard whose GaussVolPoint stencil meets a qgdFlux wall exchanges the mid-step patch pressure of the
// boundary layer's patch faces between 1 and 2 (midExchangeNeeded): a ghost cell's patch face forms it from an incomplete stencil, and the
// vertex values of p on the wall carry it into the stencil of owned faces.
static void assembleFluxes(qgd_case_s* c, bool adjust, int part = 0, bool internalFaces = true) {
    // internalFaces = false: the fused step computes them itself (stepFused)
    (void)hipGetLastError();  // drop any stale sticky error: the callers check after their launches
    const Launcher L = launcherOf(c);
    const MeshView& m = c->dev->view;
    const CaseView& v = c->view;
    const bool mid = c->pRefresh && c->hasQgdFlux;
    auto bface = [&](int phiwOnly, bool adj) {
        if (c->mixB >= 0) launchBoundaryFaceFluxMixed(L, c->stencil, c->mixB, c->mixMask, m, v, c->gas, c->bcDev, phiwOnly, adj);
        else launchBoundaryFaceFlux(L, c->stencil, m, v, c->gas, c->bcDev, phiwOnly, adj);
    };
    if (part != 2 && c->usesPoints) {
        if (internalFaces) launchPointInterp(L, m, v);   // (the fused kernel forms the vertex values of its blocks itself; patch points below)
        launchBoundaryPoints(L, m, v, false);
        // fvsc::grad(p) under GaussVolPoint re-runs p's BCs after phiwStar was refreshed
        // [QGDFoam/updateFluxes.H L63-65, GaussVolPointStencil_8C L73]
        if (mid) bface(1, false);  // + the mid-step pressure itself
    }
    if (part == 1) return;
    if (mid) launchBoundaryPoints(L, m, v, true);
    if (!internalFaces) {}
    else if (c->mixB >= 0) launchFaceFluxMixed(L, c->stencil, c->mixB, c->mixMask, m, v, c->gas, adjust);
    else launchFaceFlux(L, c->stencil, m, v, c->gas, adjust);
    bface(mid ? 2 : 0, adjust);
    // Courant-number control on the cell blocks: every block up to its cells' flux sums and its faces' Courant partials (the patch faces'
    // fluxes are in place now); the cells advance in stepAdvance, once deltaT is known
    if (c->fusedAdj && !internalFaces && !c->view.dbg) launchFusedAdjust(L, m, v, c->gas);
}
static bool midExchangeNeeded(const qgd_case_s* c) { return c->dev->sharded() && c->pRefresh && c->hasQgdFlux; }

int qgd_case_set_fields(qgd_case_t c, const double* U, const double* T, const double* p) {
    QGD_TRY
    if (!c || !U || !T || !p) return fail(QGD_ERR_INVALID, "qgd_case_set_fields: null argument");
    HIP_CHECK(hipSetDevice(c->dev->deviceId));
    const MeshView& m = c->dev->view;
    c->hasQgdFlux = false;
    for (const PatchBCDev& b : c->bc) if (b.bcP == QGD_BC_QGDFLUX) c->hasQgdFlux = true;
    if (!c->bc.empty()) HIP_CHECK(hipMemcpy(c->bcDev, c->bc.data(), sizeof(PatchBCDev) * c->bc.size(), hipMemcpyHostToDevice));
    double *dU = nullptr, *dT = nullptr, *dp = nullptr;
    auto cleanup = [&]() { (void)hipFree(dU); (void)hipFree(dT); (void)hipFree(dp); };
    try {
        HIP_CHECK(hipMalloc((void**)&dU, sizeof(double) * 3 * (size_t)m.nC));
        HIP_CHECK(hipMalloc((void**)&dT, sizeof(double) * (size_t)m.nC));
        HIP_CHECK(hipMalloc((void**)&dp, sizeof(double) * (size_t)m.nC));
        HIP_CHECK(hipMemcpy(dU, U, sizeof(double) * 3 * (size_t)m.nC, hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(dT, T, sizeof(double) * (size_t)m.nC, hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(dp, p, sizeof(double) * (size_t)m.nC, hipMemcpyHostToDevice));
        Launcher L = launcherOf(c);
        L.pre = nullptr; L.post = nullptr;
        (void)hipGetLastError();
        launchCellInit(L, m, c->view, c->gas, dU, dT, dp);
        launchBoundaryUpdate(L, m, c->view, c->gas, c->bcDev, true, false, 0, nullptr, 0);
        launchResetReductions(L, c->view);
        if (c->fused) {   // a shard's ghost records are written by the halo exchange only: both buffers start with the initial ones
            HIP_CHECK(hipMemcpyAsync(c->view.A2, c->view.A, sizeof(RecA) * (size_t)m.nC, hipMemcpyDeviceToDevice, c->stream()));
            HIP_CHECK(hipMemcpyAsync(c->view.B2, c->view.B, sizeof(RecB) * (size_t)m.nC, hipMemcpyDeviceToDevice, c->stream()));
        }
        const double dt0[3] = {c->opt.deltaT, 0.0, 0.0};
        HIP_CHECK(hipMemcpyAsync(c->view.dt, dt0, sizeof(dt0), hipMemcpyHostToDevice, c->stream()));
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipStreamSynchronize(c->stream()));
    } catch (...) { cleanup(); throw; }
    cleanup();
    c->phiwRegistered = true;  // createFaceFluxes.H registers "phiwStar" before the loop starts
    c->ghostsCurrent = false;  // (cyclic pairs served by ghost cells: the copies take their originals' records before the first step)
    c->fieldsSet = true;
    c->gradUValid = false;
    c->impl.have = 0;          // the start values of the implicit solves begin without a history
    c->time = 0; c->steps = 0;
    if (c->implSolver) { implicitSolverSetStream(c->implSolver, c->stream()); implicitStatsReset(c->implSolver); HIP_CHECK(hipStreamSynchronize(c->stream())); }
    return QGD_OK;
    QGD_CATCH
}

int qgd_case_update_fluxes(qgd_case_t c) {
    QGD_TRY
    if (!c) return fail(QGD_ERR_INVALID, "null case");
    if (!c->fieldsSet) return fail(QGD_ERR_INVALID, "qgd_case_update_fluxes: call qgd_case_set_fields first");
    HIP_CHECK(hipSetDevice(c->dev->deviceId));
    const MeshView& m = c->dev->view;
    if (!c->dbgBuf) c->dbgBuf = c->arena.alloc<double>((size_t)DBG_COUNT * m.nF);
    HIP_CHECK(hipMemsetAsync(c->dbgBuf, 0, sizeof(double) * (size_t)DBG_COUNT * m.nF, c->stream()));
    c->view.dbg = c->dbgBuf;
    assembleFluxes(c, false);
    c->view.dbg = nullptr;
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipStreamSynchronize(c->stream()));
    return QGD_OK;
    QGD_CATCH
}

// phase 0: flux assembly (+ the shard's max Cof / min tauQGDf into red[0], -red[1]);
// phase 1: deltaT, cell update, boundary refresh
static void stepAssemble(qgd_case_s* c, int part = 0) {
    const bool adjust = c->opt.adjustTimeStep != 0;
    assembleFluxes(c, adjust, part, !(c->fused || c->fusedImpl || c->fusedAdj));   // a fused case computes its internal faces inside the advance (stepAdvance / phase 21)
    if (adjust && part != 1) launchFaceReduce(launcherOf(c), c->view);
}
// ---- the implicitDiffusion branch [QGDUEqn.H L54-75, QGDEEqn.H L53-64] as stream-ordered phases ---------------------------------
//   20  deltaT, fvc::grad(U) of the old state                                -> message kind 1
//   21  tauMC / phiTauMC, rho, rhoU, the three U systems and their start values -> message kind 4
//   22  solver phase 0 (A x, r)                                              -> reduce control slots 0..2
//   23, 24  solver phases 1, 2                                               -> reduce slot 3 | slot 4 + message kind 3
//   25, 26, 27  one PCG iteration (solver phases 3, 4, 5)                     -> reduce slot 5 | slots 6..7 | message kind 3
//   28  U into the records, its boundary conditions                          -> message kind 2
//   29  fvc::grad(U) of the new velocity                                     -> message kind 1
//   30  phiSigmaDotU, the energy equation's explicit part, the e system       -> message kind 4; then 22, 23, 24, (25, 26, 27)* for e
//   35  rhoE, thermo, p, boundary refresh                                    -> the state message (qgd_case_halo_*)
// No host synchronisation anywhere inside.
static void implicitPhase(qgd_case_s* c, int phase) {
    const qgd_device_s* d = c->dev;
    const MeshView& m = d->view;
    ImplicitSolver* S = c->implSolver;
    hipStream_t st = c->stream();
    implicitSolverSetStream(S, st);
    const double tol = c->opt.implicitTol;
    const int maxIter = c->opt.implicitMaxIter;
    (void)hipGetLastError();
    switch (phase) {
        case 20: {
            const bool adjust = c->opt.adjustTimeStep != 0;
            if (adjust) launchDeltaT(launcherOf(c), c->view, c->opt.maxCo, c->opt.maxDeltaT, c->opt.cTau);
            if (adjust) launchImplicitStartWeights(st, c->view, c->impl);   // the start values' Lagrange weights from the deltaT ratios
            c->steps++;
            if (!adjust) c->time += c->opt.deltaT;
            implicitStepMark(S, true);
            // fvc::grad(U) of the state before the step IS the gradient phase 29 of the step before formed from the new velocity and its
            // patch values: nothing has touched either since (unsharded; any field, BC or coefficient upload resets the flag)
            if (!(c->reuseGradU && c->gradUValid && !d->sharded())) launchImplicitPart(st, m, c->view, c->impl, c->gas, c->bcDev, S, tol, maxIter, 0);
            c->gradUValid = false;
            break;
        }
        case 21: c->implSolveIndex = 0; launchImplicitPart(st, m, c->view, c->impl, c->gas, c->bcDev, S, tol, maxIter, 1, c->fusedImpl); break;
        case 22: case 23: case 24: case 25: case 26: case 27: implicitSolvePhase(S, phase - 22); break;
        case 28:
            implicitSolveEnd(S, 0);
            launchImplicitPart(st, m, c->view, c->impl, c->gas, c->bcDev, S, tol, maxIter, 2);
            break;
        case 29: launchImplicitPart(st, m, c->view, c->impl, c->gas, c->bcDev, S, tol, maxIter, 3); c->gradUValid = true; break;
        case 30: c->implSolveIndex = 1; launchImplicitPart(st, m, c->view, c->impl, c->gas, c->bcDev, S, tol, maxIter, 4); break;
        case 35:
            implicitSolveEnd(S, 1);
            launchImplicitPart(st, m, c->view, c->impl, c->gas, c->bcDev, S, tol, maxIter, 5);
            if (c->impl.pred) {   // this step's corrections sit in the oldest slot: it becomes the newest
                ImplView& iv = c->impl;
                double* t = iv.dh[iv.order - 1];
                for (int j = iv.order - 1; j >= 1; --j) iv.dh[j] = iv.dh[j - 1];
                iv.dh[0] = t;
                iv.have = std::min(iv.have + 1, iv.order);
            }
            implicitStepMark(S, false);
            launchBoundaryUpdate(launcherOf(c), m, c->view, c->gas, c->bcDev, false, c->phiwRegistered, 0, nullptr, 0);
            break;
        default: throw std::invalid_argument("implicit branch: phase must be 20..30 or 35");
    }
    HIP_CHECK(hipGetLastError());
}
// the whole advance with the transport behind the hooks (nullptr: one rank); haloImpl(kind) exchanges message kind 1 / 2
static void implicitAdvanceWith(qgd_case_s* c, const SolveHooks* hooks, const std::function<void(int)>& haloImpl) {
    ImplicitSolver* S = c->implSolver;
    implicitPhase(c, 20);
    if (haloImpl) haloImpl(1);
    implicitPhase(c, 21);
    implicitSolveRun(S, hooks);     // message kind 4, phase 0, ... through the hooks
    implicitPhase(c, 28);
    if (haloImpl) haloImpl(2);
    implicitPhase(c, 29);
    if (haloImpl) haloImpl(1);
    implicitPhase(c, 30);
    implicitSolveRun(S, hooks);
    implicitPhase(c, 35);
}

// part 0 = everything; part 1 = deltaT + the shard's boundary layer (cells a neighbour needs, and their patch faces);
// part 2 = the remaining owned cells/faces.  Ghost cells and their patch faces are only ever written by halo_unpack.
static void stepAdvance(qgd_case_s* c, int part) {
    const Launcher L = launcherOf(c);
    const qgd_device_s* d = c->dev;
    const MeshView& m = d->view;
    const bool adjust = c->opt.adjustTimeStep != 0;
    if (c->opt.implicitDiffusion) {
        implicitAdvanceWith(c, nullptr, nullptr);   // deltaT and the step count are advanced inside (phase 20)
        return;
    }
    if (part != 2) {
        if (adjust) launchDeltaT(L, c->view, c->opt.maxCo, c->opt.maxDeltaT, c->opt.cTau);
        c->steps++;
        if (!adjust) c->time += c->opt.deltaT;
    }
    if (c->fusedAdj) {
        launchCellFinish(L, m, c->view, c->gas, part, part == 1 ? d->sendAll : nullptr, part == 1 ? d->nSendAll : 0);
        launchBoundaryUpdate(L, m, c->view, c->gas, c->bcDev, false, c->phiwRegistered, part,
                             part == 1 ? d->sendBFAll : nullptr, part == 1 ? d->nSendBFAll : 0);
        return;
    }
    if (c->fused) {
        // The fused face + cell kernel writes the new records to A2 / B2 (its blocks read their neighbours' OLD records), which then become
        // A / B.  On a shard the boundary-layer blocks go first (part 1) and the swap follows them at once, so that the halo pack and the
        // patch faces of those cells see the new records; the remaining blocks (part 2) then read the old records where the swap left
        // them -- A2 / B2 -- and write where the new ones belong.
        if (part == 0) {
            launchFusedFaceCell(L, m, c->view, c->gas, 0, m.fuBlocks);
            std::swap(c->view.A, c->view.A2);
            std::swap(c->view.B, c->view.B2);
        } else if (part == 1) {
            launchFusedFaceCell(L, m, c->view, c->gas, 0, m.fuLayerBlocks);
            std::swap(c->view.A, c->view.A2);
            std::swap(c->view.B, c->view.B2);
        } else {
            CaseView back = c->view;
            std::swap(back.A, back.A2);
            std::swap(back.B, back.B2);
            launchFusedFaceCell(L, m, back, c->gas, m.fuLayerBlocks, m.fuBlocks - m.fuLayerBlocks);
        }
        launchBoundaryUpdate(L, m, c->view, c->gas, c->bcDev, false, c->phiwRegistered, part,
                             part == 1 ? d->sendBFAll : nullptr, part == 1 ? d->nSendBFAll : 0);
        return;
    }
    if (part == 0) {
        launchCellUpdate(L, m, c->view, c->gas, 0, nullptr, 0);
        launchBoundaryUpdate(L, m, c->view, c->gas, c->bcDev, false, c->phiwRegistered, 0, nullptr, 0);
    } else if (part == 1) {
        launchCellUpdate(L, m, c->view, c->gas, 1, d->sendAll, d->nSendAll);
        launchBoundaryUpdate(L, m, c->view, c->gas, c->bcDev, false, c->phiwRegistered, 1, d->sendBFAll, d->nSendBFAll);
    } else {
        launchCellUpdate(L, m, c->view, c->gas, 2, nullptr, 0);
        launchBoundaryUpdate(L, m, c->view, c->gas, c->bcDev, false, c->phiwRegistered, 2, nullptr, 0);
    }
}

// cyclic pairs served by ghost cells: every slot packs into the case's own buffers, every slot unpacks what its partner slot packed
// (mid = the 2-double message in the middle of the assembly, see assembleFluxes)
static void selfHaloExchange(qgd_case_s* c, bool mid) {
    qgd_device_s* d = c->dev;
    const size_t nSlots = d->halo.size();
    if (c->selfBuf.size() != nSlots) {
        c->selfBuf.assign(nSlots, nullptr);
        for (size_t k = 0; k < nSlots; ++k) {
            const qgd_device_s::HaloSlot& h = d->halo[k];
            const size_t n = std::max<size_t>(1, QGD_HALO_CELL_DOUBLES_HOST * (size_t)h.nSend + 12 * (size_t)h.nSendBF);
            c->selfBuf[k] = c->arena.alloc<double>(n);
        }
    }
    Launcher L = launcherOf(c);
    L.pre = nullptr; L.post = nullptr;
    (void)hipGetLastError();
    for (size_t k = 0; k < nSlots; ++k) {
        const qgd_device_s::HaloSlot& h = d->halo[k];
        if (mid) { if (h.nSendBF) launchMidHalo(L.stream, c->view, h.sendBF, h.nSendBF, c->selfBuf[k], true); }
        else if (h.nSend) launchHaloPack(L, c->view, c->gas, h.send, h.nSend, h.sendBF, h.nSendBF, c->selfBuf[k], true);
    }
    for (size_t k = 0; k < nSlots; ++k) {
        const qgd_device_s::HaloSlot& h = d->halo[k];
        const qgd_device_s::HaloSlot& from = d->halo[(size_t)d->haloSelf[k]];
        if (from.nSend != h.nGhost || from.nSendBF != h.nGhostBF) throw std::runtime_error("cyclic self-exchange: a slot's ghosts do not match its partner's message");
        double* buf = c->selfBuf[(size_t)d->haloSelf[k]];
        if (mid) { if (h.nGhostBF) launchMidHalo(L.stream, c->view, h.ghostBF, h.nGhostBF, buf, false); }
        else if (h.nGhost) launchHaloPack(L, c->view, c->gas, h.ghost, h.nGhost, h.ghostBF, h.nGhostBF, buf, false);
    }
    HIP_CHECK(hipGetLastError());
}

int qgd_case_step(qgd_case_t c, int32_t nSteps) {
    QGD_TRY
    if (!c) return fail(QGD_ERR_INVALID, "null case");
    if (!c->fieldsSet) return fail(QGD_ERR_INVALID, "qgd_case_step: call qgd_case_set_fields first");
    if (c->dev->sharded() && !c->dev->periodic())
        return fail(QGD_ERR_INVALID, "qgd_case_step: sharded mesh, drive it with qgd_case_step_phase + halo exchange");
    HIP_CHECK(hipSetDevice(c->dev->deviceId));
    if (c->dev->periodic()) {
        // cyclic pairs served by ghost cells: the copies are refreshed from their originals after every step (and in the middle of the
        // assembly where a shard would exchange there), all on the case's stream -- what a rank does with its neighbours, with itself
        if (c->opt.implicitDiffusion)
            return fail(QGD_ERR_NOT_IMPLEMENTED, "qgd_case_step: the implicitDiffusion branch on a mesh with cyclic patches (qgd_mesh_unroll_cyclic) is not served: "
                                                 "its solves would need the coupled rows; use the explicit branch");
        if (!c->ghostsCurrent) { selfHaloExchange(c, false); c->ghostsCurrent = true; }
        for (int i = 0; i < nSteps; ++i) {
            if (midExchangeNeeded(c)) { stepAssemble(c, 1); selfHaloExchange(c, true); stepAssemble(c, 2); }
            else stepAssemble(c);
            if (c->opt.adjustTimeStep) {}   // (one rank: the shard's own maximum IS the global one)
            stepAdvance(c, 0);
            selfHaloExchange(c, false);
        }
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipStreamSynchronize(c->stream()));
        return QGD_OK;
    }
    for (int i = 0; i < nSteps; ++i) { stepAssemble(c); stepAdvance(c, 0); }
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipStreamSynchronize(c->stream()));
    return QGD_OK;
    QGD_CATCH
}

int qgd_case_step_phase(qgd_case_t c, int phase) {
    QGD_TRY
    if (!c) return fail(QGD_ERR_INVALID, "null case");
    if (!c->fieldsSet) return fail(QGD_ERR_INVALID, "qgd_case_step_phase: call qgd_case_set_fields first");
    HIP_CHECK(hipSetDevice(c->dev->deviceId));
    if (c->opt.implicitDiffusion && (phase == 10 || phase == 11))
        return fail(QGD_ERR_NOT_IMPLEMENTED, "qgd_case_step_phase: the implicitDiffusion branch has no boundary-layer-first order (its linear "
                                            "solves span all cells); use phases 0 and 1, or 0 and 20..30, 35 on a shard");
    if (phase >= 20 && phase <= 35) {
        if (!c->opt.implicitDiffusion) return fail(QGD_ERR_INVALID, "qgd_case_step_phase: phases 20..35 belong to the implicitDiffusion branch");
        if ((phase > 30 && phase < 35)) return fail(QGD_ERR_INVALID, "qgd_case_step_phase: phase must be 0, 1, 2, 10, 11, 20..30 or 35");
        implicitPhase(c, phase);
        return QGD_OK;
    }
    if (c->opt.implicitDiffusion && phase == 1 && c->dev->sharded())
        return fail(QGD_ERR_INVALID, "qgd_case_step_phase: on a shard the implicitDiffusion branch advances through phases 20..35 with the "
                                     "exchanges between them");
    if (phase == 0) stepAssemble(c);
    else if (phase == 5) stepAssemble(c, 1);
    else if (phase == 6) stepAssemble(c, 2);
    else if (phase == 1) stepAdvance(c, 0);
    else if (phase == 10) stepAdvance(c, 1);
    else if (phase == 11) stepAdvance(c, 2);
    else if (phase == 3) {
        if (c->dev->sharded()) return fail(QGD_ERR_INVALID, "qgd_case_step_phase: phase 3 (one whole step, no exchange) is for unsharded meshes");
        stepAssemble(c);
        stepAdvance(c, 0);
    } else if (phase != 2) return fail(QGD_ERR_INVALID, "qgd_case_step_phase: phase must be 0, 1, 2, 3, 5, 6, 10 or 11");
    HIP_CHECK(hipGetLastError());
    return QGD_OK;  // asynchronous: qgd_case_stream_sync waits
    QGD_CATCH
}
int qgd_case_set_halo_stream(qgd_case_t c, void* hipStream) {
    if (!c) return fail(QGD_ERR_INVALID, "null case");
    c->haloStream = (hipStream_t)hipStream;
    c->useHaloStream = true;
    return QGD_OK;
}
int qgd_case_reduction_ptr(qgd_case_t c, void** devicePtr) {
    if (!c || !devicePtr) return fail(QGD_ERR_INVALID, "null argument");
    *devicePtr = c->view.red;
    return QGD_OK;
}
int qgd_case_set_stream(qgd_case_t c, void* hipStream) {
    if (!c) return fail(QGD_ERR_INVALID, "null case");
    (void)hipSetDevice(c->dev->deviceId);
    harvestTiming(c);
    (void)hipStreamSynchronize(c->stream());
    c->userStream = (hipStream_t)hipStream;
    c->useUserStream = true;
    return QGD_OK;
}
int qgd_case_stream_sync(qgd_case_t c) {
    QGD_TRY
    if (!c) return fail(QGD_ERR_INVALID, "null case");
    HIP_CHECK(hipSetDevice(c->dev->deviceId));
    HIP_CHECK(hipStreamSynchronize(c->stream()));
    return QGD_OK;
    QGD_CATCH
}

int qgd_device_alloc(qgd_device_t d, int64_t bytes, void** devicePtr) {
    QGD_TRY
    if (!d || !devicePtr || bytes <= 0) return fail(QGD_ERR_INVALID, "qgd_device_alloc: bad argument");
    HIP_CHECK(hipSetDevice(d->deviceId));
    HIP_CHECK(hipMalloc(devicePtr, (size_t)bytes));
    HIP_CHECK(hipMemset(*devicePtr, 0, (size_t)bytes));
    HIP_CHECK(hipStreamSynchronize(nullptr));   // the zero-fill is done before the caller's first copy on the device's (non-blocking) stream
    return QGD_OK;
    QGD_CATCH
}
int qgd_device_release(qgd_device_t d, void* devicePtr) {
    QGD_TRY
    if (!d) return fail(QGD_ERR_INVALID, "null device");
    HIP_CHECK(hipSetDevice(d->deviceId));
    HIP_CHECK(hipFree(devicePtr));
    return QGD_OK;
    QGD_CATCH
}

// halo message layout: 10 doubles per cell (RecA, RecB), 12 per boundary face (RecA, RecB, p gradient, lagged rho)
int qgd_case_halo_count(qgd_case_t c, int slot, int64_t* count) {
    if (!c || !count || slot < 0) return fail(QGD_ERR_INVALID, "bad argument");
    *count = 0;
    if (slot >= (int)c->dev->halo.size()) return QGD_OK;  // an unsharded mesh has no slots: nothing to exchange
    *count = QGD_HALO_CELL_DOUBLES_HOST * (int64_t)c->dev->halo[slot].nSend + 12 * (int64_t)c->dev->halo[slot].nSendBF;
    return QGD_OK;
}
int qgd_case_halo_recv_count(qgd_case_t c, int slot, int64_t* count) {
    if (!c || !count || slot < 0) return fail(QGD_ERR_INVALID, "bad argument");
    *count = 0;
    if (slot >= (int)c->dev->halo.size()) return QGD_OK;
    *count = QGD_HALO_CELL_DOUBLES_HOST * (int64_t)c->dev->halo[slot].nGhost + 12 * (int64_t)c->dev->halo[slot].nGhostBF;
    return QGD_OK;
}
int qgd_case_halo_pack(qgd_case_t c, int slot, double* sendBufDevice) {
    QGD_TRY
    if (!c || slot < 0) return fail(QGD_ERR_INVALID, "bad argument");
    qgd_device_s* d = c->dev;
    if (slot >= (int)d->halo.size() || !d->halo[slot].nSend) return QGD_OK;
    if (!sendBufDevice) return fail(QGD_ERR_INVALID, "null buffer");
    const qgd_device_s::HaloSlot& h = d->halo[slot];
    HIP_CHECK(hipSetDevice(d->deviceId));
    Launcher L = launcherOf(c);
    L.pre = nullptr; L.post = nullptr;
    if (c->useHaloStream) L.stream = c->haloStream;
    (void)hipGetLastError();
    launchHaloPack(L, c->view, c->gas, h.send, h.nSend, h.sendBF, h.nSendBF, sendBufDevice, true);
    HIP_CHECK(hipGetLastError());
    return QGD_OK;
    QGD_CATCH
}
int qgd_case_halo_unpack(qgd_case_t c, int slot, const double* recvBufDevice) {
    QGD_TRY
    if (!c || slot < 0) return fail(QGD_ERR_INVALID, "bad argument");
    qgd_device_s* d = c->dev;
    if (slot >= (int)d->halo.size() || !d->halo[slot].nGhost) return QGD_OK;
    if (!recvBufDevice) return fail(QGD_ERR_INVALID, "null buffer");
    const qgd_device_s::HaloSlot& h = d->halo[slot];
    HIP_CHECK(hipSetDevice(d->deviceId));
    Launcher L = launcherOf(c);
    L.pre = nullptr; L.post = nullptr;
    if (c->useHaloStream) L.stream = c->haloStream;
    (void)hipGetLastError();
    launchHaloPack(L, c->view, c->gas, h.ghost, h.nGhost, h.ghostBF, h.nGhostBF, const_cast<double*>(recvBufDevice), false);
    HIP_CHECK(hipGetLastError());
    return QGD_OK;
    QGD_CATCH
}

// the message between step phases 5 and 6 (see assembleFluxes): 2 doubles per patch face of the slot's boundary-layer cells
int qgd_case_mid_exchange_needed(qgd_case_t c, int32_t* needed) {
    if (!c || !needed) return fail(QGD_ERR_INVALID, "bad argument");
    *needed = midExchangeNeeded(c) ? 1 : 0;
    return QGD_OK;
}
int qgd_case_mid_halo_count(qgd_case_t c, int slot, int64_t* sendCount, int64_t* recvCount) {
    if (!c || slot < 0 || !sendCount || !recvCount) return fail(QGD_ERR_INVALID, "bad argument");
    *sendCount = *recvCount = 0;
    if (slot >= (int)c->dev->halo.size()) return QGD_OK;
    *sendCount = 2 * (int64_t)c->dev->halo[slot].nSendBF;
    *recvCount = 2 * (int64_t)c->dev->halo[slot].nGhostBF;
    return QGD_OK;
}
static int midHaloMove(qgd_case_s* c, int slot, double* buf, bool pack, hipStream_t stream) {
    qgd_device_s* d = c->dev;
    if (slot >= (int)d->halo.size()) return QGD_OK;
    const qgd_device_s::HaloSlot& h = d->halo[slot];
    const int32_t n = pack ? h.nSendBF : h.nGhostBF;
    if (n == 0) return QGD_OK;
    if (!buf) return fail(QGD_ERR_INVALID, "null buffer");
    (void)hipGetLastError();
    launchMidHalo(stream, c->view, pack ? h.sendBF : h.ghostBF, n, buf, pack);
    HIP_CHECK(hipGetLastError());
    return QGD_OK;
}
int qgd_case_mid_halo_pack(qgd_case_t c, int slot, double* sendBufDevice) {
    QGD_TRY
    if (!c || slot < 0) return fail(QGD_ERR_INVALID, "bad argument");
    HIP_CHECK(hipSetDevice(c->dev->deviceId));
    return midHaloMove(c, slot, sendBufDevice, true, c->stream());
    QGD_CATCH
}
int qgd_case_mid_halo_unpack(qgd_case_t c, int slot, const double* recvBufDevice) {
    QGD_TRY
    if (!c || slot < 0) return fail(QGD_ERR_INVALID, "bad argument");
    HIP_CHECK(hipSetDevice(c->dev->deviceId));
    return midHaloMove(c, slot, const_cast<double*>(recvBufDevice), false, c->stream());
    QGD_CATCH
}

// ---- QHDFoam case resident on the device -------------------------------------------------------------------------------
struct qgd_qhd_case_s {
    qgd_device_s* dev = nullptr;
    qgd_qhd_options opt{};
    int stencil = ST_GVP3;
    bool usesPoints = true, fieldsSet = false;
    double* pHist[4] = {nullptr, nullptr, nullptr, nullptr}; int pOrder = 0, pPrevHave = 0;   // QGD_QHD_PEXTRAP: p of the steps before (start value of the pressure solve, qgd_qhd.hip)
    int implXOrder = 0;         // QGD_IMPL_XEXTRAP (QhdView::xHave counts up to it)   // QGD_QHD_PEXTRAP: p of the step before (start value of the pressure solve, qgd_qhd.hip)
    std::vector<PatchBCDev> bc;
    PatchBCDev* bcDev = nullptr;
    DeviceArena arena;
    QhdView view{};
    double* c4b = nullptr;      // QGD_QHD_FUSED: the second {U,T} array of the block-fused U / T equations (qgd_qhd.hip qhdFusedAdvanceKernel); swapped with view.c4 every step
    double *tauF = nullptr, *tbr = nullptr, *scratch = nullptr;
    uint8_t* bKind = nullptr;
    PressureSolver* solver = nullptr;
    ImplicitSolver* implSolver = nullptr;  // implicitDiffusion: the four systems {Ux, Uy, Uz, T} as one multi-right-hand-side solve
    int implMask = 15;                     // components that are solved (validComponents: not those along empty directions)
    bool needRef = false;
    int localRefCell = -1;                 // local label of pRefCell when this shard owns it
    std::vector<int32_t> bcPRequested;     // p kinds as the caller set them (a box slab's cut plane hides the patch's own kind)
    std::vector<double*> sendBuf, recvBuf; // native transport: message buffers per halo slot (sized for the largest kind)
    double time = 0, lastIter = 0, lastRes0 = 0, lastRes = 0, lastSolveMs = 0;
    int64_t steps = 0;
};

int qgd_qhd_options_default(qgd_qhd_options* o) {
    if (!o) return fail(QGD_ERR_INVALID, "null argument");
    std::memset(o, 0, sizeof(*o));
    o->stencil = QGD_FVSC_GAUSSVOLPOINT;
    o->tauModel = 2; o->pRefCell = 0; o->pMaxIter = 1000; o->precond = 1;
    o->rho0 = 1.0; o->mu = 1e-3; o->Pr = 0.71; o->beta = 3e-3; o->g[1] = -9.81; o->deltaT = 1e-3;
    o->Tau = 1e-3; o->aQGD = 0.5; o->UQHD = 1.0; o->T0 = 1.0; o->Gr = 1e3; o->pTol = 1e-8; o->pRelTol = 0.0; o->pRefValue = 0.0;
    o->implicitTol = 1e-10; o->implicitMaxIter = 1000;
    return QGD_OK;
}
int qgd_qhd_case_create(qgd_device_t d, const qgd_qhd_options* opt, qgd_qhd_case_t* out) {
    QGD_TRY
    if (!d || !opt || !out) return fail(QGD_ERR_INVALID, "qgd_qhd_case_create: null argument");
    if (d->periodic())
        return fail(QGD_ERR_NOT_IMPLEMENTED, "qgd_qhd_case_create: a mesh whose cyclic patches were unrolled into ghost cells (qgd_mesh_unroll_cyclic) is served by "
                                             "QGDFoam's explicit branch only: the pressure equation would need the coupled rows");
    if (opt->implicitDiffusion && (!(opt->implicitTol > 0) || opt->implicitMaxIter < 1))
        return fail(QGD_ERR_INVALID, "qgd_qhd_case_create: implicitDiffusion needs implicitTol > 0 and implicitMaxIter >= 1");
    if (!(opt->rho0 > 0) || !(opt->Pr > 0) || !(opt->deltaT > 0) || opt->tauModel < 0 || opt->tauModel > 3)
        return fail(QGD_ERR_INVALID, "qgd_qhd_case_create: rho0, Pr, deltaT must be positive, tauModel in 0..3");
    if ((opt->fluxSchemeU != QGD_FLUX_LINEAR && opt->fluxSchemeU != QGD_FLUX_UPWIND) || (opt->fluxSchemeT != QGD_FLUX_LINEAR && opt->fluxSchemeT != QGD_FLUX_UPWIND))
        return fail(QGD_ERR_INVALID, "qgd_qhd_case_create: fluxSchemeU / fluxSchemeT must be QGD_FLUX_LINEAR or QGD_FLUX_UPWIND");
    if (!d->caseRefusal.empty()) return fail(d->caseRefusalCode, "qgd_qhd_case_create: " + d->caseRefusal);
    int st = 0;
    int rc = deviceStencil(d, opt->stencil, &st);
    if (rc) return rc;
    HIP_CHECK(hipSetDevice(d->deviceId));
    qgd_qhd_case_s* c = new qgd_qhd_case_s();
    try {
        c->dev = d; c->opt = *opt; c->stencil = st;
        c->usesPoints = (st == ST_GVP3 || st == ST_GVP2);
        d->ensureFaceGeoPos();
        const MeshView& v = d->view;
        if (!d->sharded() && opt->pRefCell >= v.nC) { delete c; return fail(QGD_ERR_INVALID, "qgd_qhd_case_create: pRefCell out of range"); }
        DeviceArena& a = c->arena;
        QhdView& q = c->view;
        const size_t nC = (size_t)v.nC, nB = (size_t)std::max(v.nBF, 1), nF = (size_t)v.nF, nP = (size_t)std::max(v.nP, 1);
        q.c4 = a.alloc<double>(4 * nC); q.b4 = a.alloc<double>(4 * nB); q.pt4 = a.alloc<double>(4 * nP);
        q.p = a.alloc<double>(nC); q.pb = a.alloc<double>(nB); q.pgb = a.alloc<double>(nB); q.ptp = a.alloc<double>(nP);
        c->tauF = a.alloc<double>(nF); c->tbr = a.alloc<double>(nF); q.tauF = c->tauF;
        q.phiu = a.alloc<double>(nF); q.phiwo = a.alloc<double>(nF); q.phi = a.alloc<double>(nF); q.phitr = a.alloc<double>(nF);
        q.ugu = a.alloc<double>(3 * nF); q.gUc = a.alloc<double>(9 * nC); q.F = a.alloc<double>(4 * nF);
        c->scratch = a.alloc<double>(8);
        q.implicit = opt->implicitDiffusion ? 1 : 0;
        q.upwindU = opt->fluxSchemeU == QGD_FLUX_UPWIND ? 1 : 0;
        q.upwindT = opt->fluxSchemeT == QGD_FLUX_UPWIND ? 1 : 0;
        static const int kFusedOnOff[] = {0, 1};
        // (off by default: measured at 8 M cells the one launch takes what the three kernels take -- profiles/r06_ab_qhd_fused_advance.txt)
        if (envChoice("QGD_QHD_FUSED", 0, kFusedOnOff, 2) != 0 && qhdFusedAdvanceEligible(st, v, q, nullptr, nullptr)) c->c4b = a.alloc<double>(4 * nC);
        if (q.implicit) {
            q.aG = a.alloc<double>(nF); q.diag4 = a.alloc<double>(4 * nC); q.rhs4 = a.alloc<double>(4 * nC); q.x4 = a.alloc<double>(4 * nC);
            c->implSolver = implicitSolverCreate(d->stream, v, d->ownedBegin, d->ownedEnd);
            c->implMask = 8;
            for (int k = 0; k < 3; ++k) if (!(v.nGeomD < 3 && v.emptyDir[k])) c->implMask |= 1 << k;   // validComponents (L0)
            {   // start values of the four systems: 0 = the current fields (OpenFOAM's), 1..3 = + the extrapolated time increment
                static const int kOrders[] = {0, 1, 2, 3, 4};
                c->implXOrder = envChoice("QGD_IMPL_XEXTRAP", 3, kOrders, 5);
                for (int j = 0; j < c->implXOrder; ++j) q.xd[j] = a.alloc<double>(4 * nC);
                q.xHave = 0; q.xOrder = c->implXOrder;
            }
        }
        q.rho0 = opt->rho0; q.nu = opt->mu / opt->rho0; q.Hi = (opt->mu / opt->Pr) / opt->rho0; q.beta = opt->beta;
        for (int k = 0; k < 3; ++k) q.g[k] = opt->g[k];
        q.dt = opt->deltaT; q.tauModel = opt->tauModel; q.Tau = opt->Tau; q.aQGD = opt->aQGD; q.UQHD = opt->UQHD; q.T0 = opt->T0; q.Gr = opt->Gr;
        c->bc.resize(d->patches.size());
        c->bcPRequested.assign(d->patches.size(), QGD_BC_ZEROGRADIENT);
        for (size_t i = 0; i < d->patches.size(); ++i) initPatchBC(c->bc[i], d->patches[i]);
        c->bcDev = a.alloc<PatchBCDev>(std::max<size_t>(1, c->bc.size()));
        c->bKind = a.alloc<uint8_t>(nB);
        {
            static const int kModes[] = {0, 1, 2, 3, 4};
            c->pOrder = envChoice("QGD_QHD_PEXTRAP", 4, kModes, 5);   // 0: OpenFOAM's start value p^n; k: extrapolation of order k in time (qgd_qhd.hip)
            for (int j = 0; j < c->pOrder; ++j) c->pHist[j] = a.alloc<double>(nC);
        }
    } catch (...) { if (c->implSolver) implicitSolverFree(c->implSolver); c->arena.release(); delete c; throw; }
    d->liveCases++;
    *out = c;
    return QGD_OK;
    QGD_CATCH
}
int qgd_qhd_case_free(qgd_qhd_case_t c) {
    if (!c) return QGD_OK;
    (void)hipSetDevice(c->dev->deviceId);
    if (c->solver) pressureSolverFree(c->solver);
    if (c->implSolver) { (void)hipStreamSynchronize(c->dev->stream); implicitSolverFree(c->implSolver); }
    c->arena.release();
    c->dev->liveCases--;
    delete c;
    return QGD_OK;
}
int qgd_qhd_case_set_bc(qgd_qhd_case_t c, int32_t patch, int32_t bcU, const double* valueU, int32_t bcT, double valueT, int32_t bcP,
                        double valueP) {
    QGD_TRY
    if (!c) return fail(QGD_ERR_INVALID, "null case");
    if (patch < 0 || patch >= (int32_t)c->bc.size()) return fail(QGD_ERR_INVALID, "qgd_qhd_case_set_bc: patch out of range");
    auto okU = [](int k) { return k == QGD_BC_ZEROGRADIENT || k == QGD_BC_FIXEDVALUE || k == QGD_BC_SLIP || k == QGD_BC_NONE; };
    auto okT = [](int k) { return k == QGD_BC_ZEROGRADIENT || k == QGD_BC_FIXEDVALUE || k == QGD_BC_NONE; };
    auto okP = [](int k) { return k == QGD_BC_ZEROGRADIENT || k == QGD_BC_FIXEDVALUE || k == QGD_BC_QGDFLUX || k == QGD_BC_QHDFLUX || k == QGD_BC_NONE; };
    if (!okU(bcU) || !okT(bcT) || !okP(bcP)) return fail(QGD_ERR_INVALID, "qgd_qhd_case_set_bc: unsupported boundary-condition kind");
    PatchBCDev& b = c->bc[patch];
    c->bcPRequested[patch] = bcP;
    constraintKinds(b.ptype, bcU, bcT, bcP);   // a constraint patch keeps its own field type whatever the caller asks for
    b.bcU = bcU; b.bcT = bcT; b.bcP = bcP; b.vT = valueT; b.vP = valueP;
    if (valueU) for (int k = 0; k < 3; ++k) b.vU[k] = valueU[k];
    c->fieldsSet = false;
    return QGD_OK;
    QGD_CATCH
}
int qgd_qhd_case_set_fields(qgd_qhd_case_t c, const double* U, const double* T, const double* p) {
    QGD_TRY
    if (!c || !U || !T || !p) return fail(QGD_ERR_INVALID, "qgd_qhd_case_set_fields: null argument");
    qgd_device_s* d = c->dev;
    HIP_CHECK(hipSetDevice(d->deviceId));
    const MeshView& m = d->view;
    if (!c->bc.empty()) HIP_CHECK(hipMemcpy(c->bcDev, c->bc.data(), sizeof(PatchBCDev) * c->bc.size(), hipMemcpyHostToDevice));
    // matrix kinds of the pressure equation per boundary face; fvMatrix::setReference acts only without a fixedValue patch
    std::vector<uint8_t> kind((size_t)std::max(m.nBF, 1), 0);
    bool anyFixed = false;
    for (size_t ip = 0; ip < d->patches.size(); ++ip) {
        const Patch& pt = d->patches[ip];
        int k = c->bc[ip].bcP;
        if (pt.type != QGD_PATCH_GENERIC) k = QGD_BC_NONE;
        const uint8_t kk = k == QGD_BC_FIXEDVALUE ? 1 : ((k == QGD_BC_QGDFLUX || k == QGD_BC_QHDFLUX) ? 2 : 0);
        // "does p need a reference level" is a property of the WHOLE mesh: a fixedValue patch counts when it has faces anywhere
        // (a shard may hold none of them; on an inner box slab the patch is the cut plane and only the caller's request is known)
        const bool genericGlobally = pt.type == QGD_PATCH_GENERIC || (pt.type == QGD_PATCH_HALO && pt.globalSize >= 0 && d->cellGlobal.empty());
        if (c->bcPRequested[ip] == QGD_BC_FIXEDVALUE && genericGlobally && pt.nonEmptyGlobally()) anyFixed = true;
        for (int32_t f = pt.start; f < pt.start + pt.size; ++f) kind[(size_t)(f - m.nIF)] = kk;
    }
    HIP_CHECK(hipMemcpy(c->bKind, kind.data(), kind.size(), hipMemcpyHostToDevice));
    c->needRef = !anyFixed && c->opt.pRefCell >= 0;
    c->localRefCell = c->needRef ? (d->sharded() ? d->ownedLocalOf(c->opt.pRefCell) : c->opt.pRefCell) : -1;
    Workspace& ws = d->ws;
    double* dU = ws.get<double>(WS_CELL, 3 * (size_t)m.nC);
    double* dT = ws.get<double>(WS_A, (size_t)m.nC);
    double* dp = ws.get<double>(WS_B, (size_t)m.nC);
    ws.h2d(dU, U, sizeof(double) * 3 * (size_t)m.nC, d->stream);
    ws.h2d(dT, T, sizeof(double) * (size_t)m.nC, d->stream);
    ws.h2d(dp, p, sizeof(double) * (size_t)m.nC, d->stream);
    (void)hipGetLastError();
    HIP_CHECK(hipMemsetAsync(c->view.phiwo, 0, sizeof(double) * (size_t)m.nF, d->stream));
    launchQhdInit(d->stream, m, c->view, c->bcDev, dU, dT, dp, c->tauF, c->tbr);
    if (c->view.implicit) {   // the matrix of the U and T equations: constant in time (thermo is not corrected in the loop, deltaT is fixed)
        launchQhdImplicitMatrix(d->stream, m, c->view, c->bcDev);
        implicitStatsReset(c->implSolver);
    }
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipStreamSynchronize(d->stream));
    if (c->solver) { pressureSolverFree(c->solver); c->solver = nullptr; }
    c->solver = pressureSolverCreate(d->stream, m, c->tbr, c->bKind, c->localRefCell, c->opt.precond, d->ownedBegin, d->ownedEnd,
                                     d->cellGlobal.empty() ? nullptr : d->cellGlobal.data(), d->cellGlobalOffset, d->sharded());
    c->fieldsSet = true;
    c->pPrevHave = 0;
    c->view.xHave = 0;
    c->time = 0; c->steps = 0;
    return QGD_OK;
    QGD_CATCH
}
// One step as stream-ordered phases (the exchanges of a sharded run go between them, see include/qgd_amd.h):
//   0 flux assembly + rhs + first residual | 1 normFactor | 2 first preconditioned residual | 3, 4, 5 one PCG iteration |
//   6 p's boundary conditions after the solve | 7 phi, the U and T equations | 8 reference level of p
static void qhdPhase(qgd_qhd_case_s* c, int phase) {
    qgd_device_s* d = c->dev;
    const MeshView& m = d->view;
    (void)hipGetLastError();
    switch (phase) {
        case 0:
            launchQhdAssemble(d->stream, c->stencil, c->usesPoints, m, c->view, c->bcDev);
            if (c->pOrder > 0) { launchQhdExtrapolateP(d->stream, m.nC, c->view.p, c->pHist, c->pPrevHave, c->pOrder); c->pPrevHave = std::min(c->pPrevHave + 1, c->pOrder); }
            pressureSolveBegin(c->solver, c->view.phiu, c->view.phiwo, c->view.pb, c->view.pgb, c->opt.pTol, c->opt.pRelTol, c->opt.pMaxIter, c->view.p);
            break;
        case 1: case 2: case 3: case 4: case 5: pressureSolvePhase(c->solver, phase); break;
        case 6: launchQhdPostSolve(d->stream, m, c->view, c->bcDev); break;
        case 7:
            pressureSolveFlux(c->solver, c->view.phi);
            if (!c->view.implicit)
            {
                if (launchQhdAdvance(d->stream, c->stencil, c->usesPoints, m, c->view, c->bcDev, c->needRef, c->localRefCell, c->opt.pRefValue,
                                     pressureSolverCtl(c->solver) + 8, c->c4b))
                    std::swap(c->view.c4, c->c4b);   // the blocks wrote the new {U,T} into the second array
            } else {
                // implicitDiffusion: face pass 2 without the laplacians, the right-hand sides; the solve (phases 10..15) and phase 16 follow
                implicitStepMark(c->implSolver, true);
                launchQhdImplicitAdvance(d->stream, c->stencil, c->usesPoints, m, c->view, c->bcDev, 0, c->implMask, false, -1, 0.0, nullptr);
                if (c->view.xOrder > 0) {   // the right-hand-side kernel put the current fields into the oldest slot: it becomes the newest
                    QhdView& q = c->view;
                    double* t = q.xd[q.xOrder - 1];
                    for (int j = q.xOrder - 1; j >= 1; --j) q.xd[j] = q.xd[j - 1];
                    q.xd[0] = t;
                    q.xHave = std::min(q.xHave + 1, q.xOrder);
                }
                const double gamma[4] = {c->view.nu, c->view.nu, c->view.nu, c->view.Hi};
                implicitSolveSetup(c->implSolver, 4, c->implMask, c->view.aG, c->view.diag4, c->view.rhs4, c->view.x4, c->opt.implicitTol,
                                   c->opt.implicitMaxIter, gamma);
            }
            break;
        case 10: case 11: case 12: case 13: case 14: case 15:
            if (!c->view.implicit) throw std::invalid_argument("qgd_qhd_case_step_phase: phases 10..16 belong to the implicitDiffusion branch");
            implicitSolvePhase(c->implSolver, phase - 10);
            break;
        case 16:
            if (!c->view.implicit) throw std::invalid_argument("qgd_qhd_case_step_phase: phases 10..16 belong to the implicitDiffusion branch");
            implicitSolveEnd(c->implSolver, 0);
            implicitStepMark(c->implSolver, false);
            launchQhdImplicitAdvance(d->stream, c->stencil, c->usesPoints, m, c->view, c->bcDev, 1, c->implMask, c->needRef, c->localRefCell,
                                     c->opt.pRefValue, pressureSolverCtl(c->solver) + 8);
            break;
        case 8:
            launchQhdFinish(d->stream, m, c->view, c->needRef, pressureSolverCtl(c->solver) + 8);
            c->time += c->opt.deltaT;
            c->steps++;
            break;
        case 9: pressureSolveContinue(c->solver); break;   // after the collective qgd_qhd_case_pending asked for
        default: throw std::invalid_argument("qgd_qhd_case_step_phase: phase must be 0..9 (10..16: implicitDiffusion)");
    }
    HIP_CHECK(hipGetLastError());
}
static void qhdReadStatus(qgd_qhd_case_s* c) {
    double st[4];
    pressureSolveStatus(c->solver, st);
    c->lastIter = st[1]; c->lastRes0 = st[2]; c->lastRes = st[3];
}
// a whole step with the transport behind `hooks` (nullptr: one rank); haloState(kind) exchanges message kind 0 or 1
static void qhdStepWith(qgd_qhd_case_s* c, const SolveHooks* hooks, const std::function<void(int)>& haloState) {
    qhdPhase(c, 0);
    HIP_CHECK(hipStreamSynchronize(c->dev->stream));   // one wait per step, so that lastSolveMs is the solve alone
    const double t0 = nowMs();
    double res[2];
    c->lastIter = pressureSolveRun(c->solver, hooks, res);
    c->lastRes0 = res[0]; c->lastRes = res[1];
    c->lastSolveMs = nowMs() - t0;
    qhdPhase(c, 6);
    if (haloState) haloState(1);
    qhdPhase(c, 7);
    if (c->view.implicit) {
        // the four systems of QHDUEqn.H L48-64 / QHDTEqn.H L71-79 as one solve; on shards its reductions and the ghost entries of
        // the iterate go through the hooks (message kind 4)
        SolveHooks ih;
        if (hooks) { ih.allreduce = hooks->allreduce; ih.allreduceBuf = hooks->allreduceBuf; }
        if (haloState) ih.haloDirection = [&]() { haloState(4); };
        implicitSolveRun(c->implSolver, &ih);
        qhdPhase(c, 16);
    }
    if (c->needRef && hooks && hooks->allreduce) hooks->allreduce(pressureSolverCtl(c->solver) + 8, 1);
    qhdPhase(c, 8);
    if (haloState) haloState(0);
}
int qgd_qhd_case_step(qgd_qhd_case_t c, int32_t nSteps) {
    QGD_TRY
    if (!c) return fail(QGD_ERR_INVALID, "null case");
    if (!c->fieldsSet) return fail(QGD_ERR_INVALID, "qgd_qhd_case_step: call qgd_qhd_case_set_fields first");
    qgd_device_s* d = c->dev;
    if (d->sharded())
        return fail(QGD_ERR_INVALID, "qgd_qhd_case_step: sharded mesh, drive it with qgd_qhd_case_step_phase + the exchanges, or qgd_qhd_case_step_sharded");
    HIP_CHECK(hipSetDevice(d->deviceId));
    for (int i = 0; i < nSteps; ++i) qhdStepWith(c, nullptr, nullptr);
    HIP_CHECK(hipStreamSynchronize(d->stream));
    return QGD_OK;
    QGD_CATCH
}
int qgd_qhd_case_step_phase(qgd_qhd_case_t c, int phase) {
    QGD_TRY
    if (!c) return fail(QGD_ERR_INVALID, "null case");
    if (!c->fieldsSet) return fail(QGD_ERR_INVALID, "qgd_qhd_case_step_phase: call qgd_qhd_case_set_fields first");
    if (phase < 0 || phase > 16) return fail(QGD_ERR_INVALID, "qgd_qhd_case_step_phase: phase must be 0..9 (10..16: implicitDiffusion)");
    HIP_CHECK(hipSetDevice(c->dev->deviceId));
    qhdPhase(c, phase);
    return QGD_OK;   // stream-ordered: qgd_qhd_case_solve_status / qgd_qhd_case_sync wait
    QGD_CATCH
}
int qgd_qhd_case_control_ptr(qgd_qhd_case_t c, void** devicePtr) {
    if (!c || !devicePtr) return fail(QGD_ERR_INVALID, "null argument");
    if (!c->solver) return fail(QGD_ERR_INVALID, "qgd_qhd_case_control_ptr: call qgd_qhd_case_set_fields first");
    *devicePtr = pressureSolverCtl(c->solver);
    return QGD_OK;
}
int qgd_qhd_case_sweep_time(qgd_qhd_case_t c, int reps, double info[4]) {
    QGD_TRY
    if (!c || !info || reps <= 0) return fail(QGD_ERR_INVALID, "bad argument");
    if (!c->solver) return fail(QGD_ERR_INVALID, "qgd_qhd_case_sweep_time: call qgd_qhd_case_set_fields first");
    HIP_CHECK(hipSetDevice(c->dev->deviceId));
    int rows = 0;
    double width = 0;
    info[0] = pressureSolverSweepMs(c->solver, reps, &rows, &width);
    info[1] = rows; info[2] = width; info[3] = pressureSolverSinglePrecisionCycle(c->solver) ? 4.0 : 8.0;
    return QGD_OK;
    QGD_CATCH
}
int qgd_qhd_case_control(qgd_qhd_case_t c, double control[16], int set) {
    QGD_TRY
    if (!c || !control) return fail(QGD_ERR_INVALID, "null argument");
    if (!c->solver) return fail(QGD_ERR_INVALID, "qgd_qhd_case_control: call qgd_qhd_case_set_fields first");
    HIP_CHECK(hipSetDevice(c->dev->deviceId));
    hipStream_t st = c->dev->stream;
    if (set) HIP_CHECK(hipMemcpyAsync(pressureSolverCtl(c->solver), control, 16 * sizeof(double), hipMemcpyHostToDevice, st));
    else HIP_CHECK(hipMemcpyAsync(control, pressureSolverCtl(c->solver), 16 * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    return QGD_OK;
    QGD_CATCH
}
// the control block (68 doubles) and the state of the implicitDiffusion solve in flight
int qgd_qhd_case_implicit_control(qgd_qhd_case_t c, double control[68], int set) {
    QGD_TRY
    if (!c || !control || !c->implSolver) return fail(QGD_ERR_INVALID, "qgd_qhd_case_implicit_control: an implicitDiffusion case is needed");
    HIP_CHECK(hipSetDevice(c->dev->deviceId));
    hipStream_t st = c->dev->stream;
    if (set) HIP_CHECK(hipMemcpyAsync(implicitSolverCtl(c->implSolver), control, 68 * sizeof(double), hipMemcpyHostToDevice, st));
    else HIP_CHECK(hipMemcpyAsync(control, implicitSolverCtl(c->implSolver), 68 * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    return QGD_OK;
    QGD_CATCH
}
int qgd_qhd_case_implicit_control_ptr(qgd_qhd_case_t c, void** devicePtr) {
    if (!c || !devicePtr || !c->implSolver) return fail(QGD_ERR_INVALID, "qgd_qhd_case_implicit_control_ptr: an implicitDiffusion case is needed");
    *devicePtr = implicitSolverCtl(c->implSolver);
    return QGD_OK;
}
int qgd_qhd_case_implicit_info(qgd_qhd_case_t c, double info[16]) {
    QGD_TRY
    if (!c || !info) return fail(QGD_ERR_INVALID, "bad argument");
    for (int i = 0; i < 16; ++i) info[i] = 0.0;
    if (!c->implSolver) return QGD_OK;
    HIP_CHECK(hipSetDevice(c->dev->deviceId));
    double allDone = 0;
    int it[4];
    double r0[4], r1[4];
    implicitSolveStatus4(c->implSolver, &allDone, it, r0, r1);
    for (int k = 0; k < 4; ++k) { info[k] = it[k]; info[4 + k] = r0[k]; info[8 + k] = r1[k]; }
    info[12] = implicitSolverUnconverged(c->implSolver, &info[14]);
    info[13] = implicitSolverChebyshev(c->implSolver) ? 2.0 : 1.0;
    return QGD_OK;
    QGD_CATCH
}
int qgd_qhd_case_implicit_solve_status(qgd_qhd_case_t c, double status[2]) {
    QGD_TRY
    if (!c || !status || !c->implSolver) return fail(QGD_ERR_INVALID, "qgd_qhd_case_implicit_solve_status: an implicitDiffusion case is needed");
    HIP_CHECK(hipSetDevice(c->dev->deviceId));
    int it[4];
    double r0[4], r1[4];
    implicitSolveStatus4(c->implSolver, &status[0], it, r0, r1);
    status[1] = implicitSolverRhs(c->implSolver);
    return QGD_OK;
    QGD_CATCH
}
int qgd_qhd_case_solve_status(qgd_qhd_case_t c, double status[4]) {
    QGD_TRY
    if (!c || !status) return fail(QGD_ERR_INVALID, "null argument");
    if (!c->solver) return fail(QGD_ERR_INVALID, "qgd_qhd_case_solve_status: call qgd_qhd_case_set_fields first");
    HIP_CHECK(hipSetDevice(c->dev->deviceId));
    pressureSolveStatus(c->solver, status);
    c->lastIter = status[1]; c->lastRes0 = status[2]; c->lastRes = status[3];
    return QGD_OK;
    QGD_CATCH
}
int qgd_qhd_case_sync(qgd_qhd_case_t c) {
    QGD_TRY
    if (!c) return fail(QGD_ERR_INVALID, "null case");
    HIP_CHECK(hipSetDevice(c->dev->deviceId));
    HIP_CHECK(hipStreamSynchronize(c->dev->stream));
    return QGD_OK;
    QGD_CATCH
}
// halo messages: doubles per listed cell / per listed patch face of message kind 0 (state), 1 (p), 2 (search direction), 3 (the iterate of
// the multigrid level that spans the ranks)
static void qhdHaloWidths(int kind, int& perCell, int& perFace) {
    perCell = kind == 0 ? 4 : (kind == 1 ? 10 : (kind == 4 ? 4 : 1));   // kind 4: the iterate of the implicit solve, {Ux, Uy, Uz, T}
    perFace = kind == 0 ? 4 : (kind == 1 ? 2 : 0);
}
// what the phase in flight waits for before qgd_qhd_case_step_phase(c, 9): see include/qgd_amd.h
int qgd_qhd_case_pending(qgd_qhd_case_t c, int32_t* action, void** devicePtr, int64_t* count) {
    if (!c || !action) return fail(QGD_ERR_INVALID, "bad argument");
    double* buf = nullptr;
    int64_t n = 0;
    *action = c->solver ? pressureSolvePending(c->solver, &buf, &n) : 0;
    if (devicePtr) *devicePtr = buf;
    if (count) *count = n;
    return QGD_OK;
}
int qgd_qhd_case_halo_count(qgd_qhd_case_t c, int slot, int kind, int64_t* sendCount, int64_t* recvCount) {
    if (!c || slot < 0 || kind < 0 || kind > 4 || !sendCount || !recvCount) return fail(QGD_ERR_INVALID, "bad argument");
    *sendCount = *recvCount = 0;
    if (slot >= (int)c->dev->halo.size()) return QGD_OK;
    int pc, pf;
    qhdHaloWidths(kind, pc, pf);
    const qgd_device_s::HaloSlot& h = c->dev->halo[slot];
    *sendCount = (int64_t)pc * h.nSend + (int64_t)pf * h.nSendBF;
    *recvCount = (int64_t)pc * h.nGhost + (int64_t)pf * h.nGhostBF;
    return QGD_OK;
}
static int qhdHaloMove(qgd_qhd_case_t c, int slot, int kind, double* buf, bool pack, hipStream_t stream) {
    qgd_device_s* d = c->dev;
    if (slot >= (int)d->halo.size()) return QGD_OK;
    const qgd_device_s::HaloSlot& h = d->halo[slot];
    const int32_t nCells = pack ? h.nSend : h.nGhost, nFaces = pack ? h.nSendBF : h.nGhostBF;
    if (nCells + nFaces == 0) return QGD_OK;
    if (!buf) return fail(QGD_ERR_INVALID, "null buffer");
    if (kind >= 2 && !c->solver) return fail(QGD_ERR_INVALID, "no solve in flight");
    (void)hipGetLastError();
    if (kind == 4) {
        if (!c->implSolver) return fail(QGD_ERR_INVALID, "message kind 4 belongs to the implicitDiffusion branch");
        launchSolverHalo(stream, c->implSolver, pack ? h.send : h.ghost, nCells, buf, pack);
        HIP_CHECK(hipGetLastError());
        return QGD_OK;
    }
    if (kind == 3) {
        float* vec = pressureSolverMgHaloVec(c->solver);
        if (!vec) return fail(QGD_ERR_INVALID, "message kind 3: no multigrid iterate is waiting for its ghost entries (qgd_qhd_case_pending)");
        launchQhdHaloFloat(stream, vec, pack ? h.send : h.ghost, nCells, buf, pack);
        HIP_CHECK(hipGetLastError());
        return QGD_OK;
    }
    launchQhdHalo(stream, c->view, c->solver ? pressureSolverDirection(c->solver) : nullptr, kind, pack ? h.send : h.ghost, nCells,
                  pack ? h.sendBF : h.ghostBF, nFaces, buf, pack);
    HIP_CHECK(hipGetLastError());
    return QGD_OK;
}
int qgd_qhd_case_halo_pack(qgd_qhd_case_t c, int slot, int kind, double* sendBufDevice) {
    QGD_TRY
    if (!c || slot < 0 || kind < 0 || kind > 4) return fail(QGD_ERR_INVALID, "bad argument");
    HIP_CHECK(hipSetDevice(c->dev->deviceId));
    return qhdHaloMove(c, slot, kind, sendBufDevice, true, c->dev->stream);
    QGD_CATCH
}
int qgd_qhd_case_halo_unpack(qgd_qhd_case_t c, int slot, int kind, const double* recvBufDevice) {
    QGD_TRY
    if (!c || slot < 0 || kind < 0 || kind > 4) return fail(QGD_ERR_INVALID, "bad argument");
    HIP_CHECK(hipSetDevice(c->dev->deviceId));
    return qhdHaloMove(c, slot, kind, const_cast<double*>(recvBufDevice), false, c->dev->stream);
    QGD_CATCH
}
int qgd_qhd_case_get_field(qgd_qhd_case_t c, const char* name, double* out, int64_t outDoubles) {
    QGD_TRY
    if (!c || !name || !out) return fail(QGD_ERR_INVALID, "qgd_qhd_case_get_field: null argument");
    qgd_device_s* d = c->dev;
    HIP_CHECK(hipSetDevice(d->deviceId));
    HIP_CHECK(hipStreamSynchronize(d->stream));
    const MeshView& m = d->view;
    std::string s(name);
    const std::string suffix = ".boundary";
    bool bnd = false;
    if (s.size() > suffix.size() && s.compare(s.size() - suffix.size(), suffix.size(), suffix) == 0) { bnd = true; s = s.substr(0, s.size() - suffix.size()); }
    const int64_t n = bnd ? m.nBF : m.nC;
    const QhdView& q = c->view;
    const double* direct = nullptr;
    int64_t count = 0;
    if (s == "p") { direct = bnd ? q.pb : q.p; count = n; }
    else if (!bnd && s == "phi") { direct = q.phi; count = m.nF; }
    else if (!bnd && s == "phiu") { direct = q.phiu; count = m.nF; }
    else if (!bnd && s == "phiwo") { direct = q.phiwo; count = m.nF; }
    else if (!bnd && s == "tauQGDf") { direct = c->tauF; count = m.nF; }
    if (direct) {
        if (count > outDoubles) return fail(QGD_ERR_INVALID, "output too small");
        if (count) d->ws.d2h(out, direct, sizeof(double) * (size_t)count, d->stream);
        return QGD_OK;
    }
    if (s != "U" && s != "T") return fail(QGD_ERR_UNKNOWN_NAME, "qgd_qhd_case_get_field: unknown field " + s);
    const int nc = s == "U" ? 3 : 1;
    if (n * nc > outDoubles) return fail(QGD_ERR_INVALID, "output too small");
    if (n == 0) return QGD_OK;
    double* tmp = d->ws.get<double>(WS_OUT, (size_t)(n * nc));
    (void)hipGetLastError();
    launchQhdExtract(d->stream, n, bnd ? q.b4 : q.c4, s == "U" ? 0 : 1, tmp);
    HIP_CHECK(hipGetLastError());
    d->ws.d2h(out, tmp, sizeof(double) * (size_t)(n * nc), d->stream);
    return QGD_OK;
    QGD_CATCH
}
int qgd_qhd_case_info(qgd_qhd_case_t c, double info[8]) {
    if (!c || !info) return fail(QGD_ERR_INVALID, "null argument");
    int sizes[16];
    info[0] = c->time; info[1] = c->opt.deltaT; info[2] = c->lastIter; info[3] = c->lastRes0; info[4] = c->lastRes; info[5] = (double)c->steps;
    info[6] = c->solver ? (double)pressureSolverLevels(c->solver, sizes, 16) : 0.0;
    info[7] = c->lastSolveMs;
    return QGD_OK;
}

int qgd_qhd_case_fused_info(qgd_qhd_case_t c, int64_t info[4]) {
    if (!c || !info) return fail(QGD_ERR_INVALID, "null argument");
    info[0] = info[1] = info[2] = info[3] = 0;
    int lds = 0;
    if (c->c4b && qhdFusedAdvanceEligible(c->stencil, c->dev->view, c->view, &lds, nullptr)) { info[0] |= 1; info[1] = c->dev->view.fuBlocks; info[2] = lds; }
    return QGD_OK;
}

// ---- native halo transport: RCCL send/recv inside the library -----------------------------------------------------------
// Replaces, for a C++/MPI host, what the reference does per gradient call with PstreamBuffers
// [extendedFaceStencilScalarGrad_8C L145-233] and with processor-patch evaluation [GaussVolPointStencil_8C L73]:
// ONE grouped send/recv pair per neighbouring rank per step, device buffers, stream-ordered, no host synchronisation.
extern "C++" {
namespace {
struct Rccl {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) getUniqueId = nullptr;
    decltype(&ncclCommInitRank) commInitRank = nullptr;
    decltype(&ncclCommDestroy) commDestroy = nullptr;
    decltype(&ncclSend) send = nullptr;
    decltype(&ncclRecv) recv = nullptr;
    decltype(&ncclAllReduce) allReduce = nullptr;
    decltype(&ncclGroupStart) groupStart = nullptr;
    decltype(&ncclGroupEnd) groupEnd = nullptr;
    decltype(&ncclGetErrorString) errorString = nullptr;
    decltype(&ncclCommCount) commCount = nullptr;
    decltype(&ncclCommUserRank) commUserRank = nullptr;
    decltype(&ncclCommCuDevice) commCuDevice = nullptr;
    std::string why;
};
Rccl& rcclRef() {
    static Rccl r;
    static bool tried = false;
    if (tried) return r;
    tried = true;
    // an RCCL already in the process (e.g. the one torch.distributed brought) is reused; otherwise the ROCm one is loaded
    const char* env = std::getenv("QGD_RCCL_LIB");
    const char* names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* n : names) {
        if (!n) continue;
        r.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
        if (r.lib) break;
    }
    for (const char* n : names) {
        if (r.lib) break;
        if (!n) continue;
        r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    }
    if (!r.lib) { r.why = "RCCL not found (librccl.so; set QGD_RCCL_LIB)"; return r; }
    auto sym = [&](const char* name) { void* p = dlsym(r.lib, name); if (!p && r.why.empty()) r.why = std::string("RCCL symbol missing: ") + name; return p; };
    r.getUniqueId = (decltype(r.getUniqueId))sym("ncclGetUniqueId");
    r.commInitRank = (decltype(r.commInitRank))sym("ncclCommInitRank");
    r.commDestroy = (decltype(r.commDestroy))sym("ncclCommDestroy");
    r.send = (decltype(r.send))sym("ncclSend");
    r.recv = (decltype(r.recv))sym("ncclRecv");
    r.allReduce = (decltype(r.allReduce))sym("ncclAllReduce");
    r.groupStart = (decltype(r.groupStart))sym("ncclGroupStart");
    r.groupEnd = (decltype(r.groupEnd))sym("ncclGroupEnd");
    r.errorString = (decltype(r.errorString))sym("ncclGetErrorString");
    r.commCount = (decltype(r.commCount))sym("ncclCommCount");
    r.commUserRank = (decltype(r.commUserRank))sym("ncclCommUserRank");
    r.commCuDevice = (decltype(r.commCuDevice))sym("ncclCommCuDevice");
    return r;
}
}  // namespace
}  // extern "C++"
struct qgd_comm_s {
    ncclComm_t comm = nullptr;
    int rank = 0, nRanks = 1, deviceId = 0;
};
#define RCCL_CHECK(expr)                                                                                          \
    do {                                                                                                          \
        ncclResult_t _r = (expr);                                                                                 \
        if (_r != ncclSuccess)                                                                                    \
            throw HipError(std::string(#expr) + ": " + (rcclRef().errorString ? rcclRef().errorString(_r) : "RCCL error")); \
    } while (0)

int qgd_comm_unique_id(void* id128) {
    QGD_TRY
    if (!id128) return fail(QGD_ERR_INVALID, "qgd_comm_unique_id: null argument");
    if (!rcclRef().why.empty()) return fail(QGD_ERR_NOT_IMPLEMENTED, rcclRef().why);
    ncclUniqueId id;
    RCCL_CHECK(rcclRef().getUniqueId(&id));
    static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
    std::memcpy(id128, &id, sizeof(id));
    return QGD_OK;
    QGD_CATCH
}
int qgd_comm_create(int deviceId, int rank, int nRanks, const void* id128, qgd_comm_t* out) {
    QGD_TRY
    if (!out || !id128 || nRanks < 1 || rank < 0 || rank >= nRanks) return fail(QGD_ERR_INVALID, "qgd_comm_create: bad argument");
    if (!rcclRef().why.empty()) return fail(QGD_ERR_NOT_IMPLEMENTED, rcclRef().why);
    HIP_CHECK(hipSetDevice(deviceId));
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    qgd_comm_s* c = new qgd_comm_s();
    c->rank = rank; c->nRanks = nRanks; c->deviceId = deviceId;
    try { RCCL_CHECK(rcclRef().commInitRank(&c->comm, nRanks, id, rank)); }
    catch (...) { delete c; throw; }
    *out = c;
    return QGD_OK;
    QGD_CATCH
}
int qgd_comm_free(qgd_comm_t c) {
    if (!c) return QGD_OK;
    ncclResult_t r = ncclSuccess;
    if (c->comm && rcclRef().commDestroy) r = rcclRef().commDestroy(c->comm);
    delete c;
    if (r != ncclSuccess)
        return fail(QGD_ERR_HIP, std::string("qgd_comm_free: ncclCommDestroy: ") + (rcclRef().errorString ? rcclRef().errorString(r) : "RCCL error"));
    return QGD_OK;
}

int qgd_comm_info(qgd_comm_t c, int32_t info[3]) {
    QGD_TRY
    if (!c || !info) return fail(QGD_ERR_INVALID, "qgd_comm_info: null argument");
    int rank = -1, count = -1, device = -1;
    RCCL_CHECK(rcclRef().commUserRank(c->comm, &rank));
    RCCL_CHECK(rcclRef().commCount(c->comm, &count));
    RCCL_CHECK(rcclRef().commCuDevice(c->comm, &device));
    info[0] = rank; info[1] = count; info[2] = device;
    return QGD_OK;
    QGD_CATCH
}

static void ensureHaloBuffers(qgd_case_s* c) {
    qgd_device_s* d = c->dev;
    if (c->sendBuf.size() == d->halo.size()) return;
    c->sendBuf.assign(d->halo.size(), nullptr);
    c->recvBuf.assign(d->halo.size(), nullptr);
    for (size_t s = 0; s < d->halo.size(); ++s) {
        const qgd_device_s::HaloSlot& h = d->halo[s];
        c->sendBuf[s] = c->arena.alloc<double>(QGD_HALO_CELL_DOUBLES_HOST * (size_t)h.nSend + 12 * (size_t)h.nSendBF);
        c->recvBuf[s] = c->arena.alloc<double>(QGD_HALO_CELL_DOUBLES_HOST * (size_t)h.nGhost + 12 * (size_t)h.nGhostBF);
    }
}
// RCCL matches the messages of one peer in issue order, and both sides issue in their own slot order: two slots towards
// the same rank (two ranks on a periodic cut, two disjoint interfaces with one neighbour) would land in each other's
// ghost lists.  Refused; a rank exchanging with itself (the one-GPU test of this path) is the one ordered exception.
// Called by every exchange AND at the top of the sharded step entries, before the first message of a step is issued
// (the mid-assembly message and the messages inside the implicit solves come before the state message).
static void checkHaloPeers(const qgd_comm_s* comm, const int32_t* peers, int n) {
    for (int a = 0; a < n; ++a)
        for (int b = a + 1; b < n; ++b)
            if (peers[a] >= 0 && peers[a] == peers[b] && peers[a] != comm->rank)
                throw std::invalid_argument("halo exchange: rank " + std::to_string(peers[a]) + " is the peer of two halo slots; "
                                            "one slot per neighbouring rank (qgd_mesh_shard builds them that way)");
    for (int s = 0; s < n; ++s)
        if (peers[s] >= comm->nRanks) throw std::invalid_argument("halo exchange: peer rank out of range");
}
// pack -> grouped send/recv -> unpack on `stream`
static void haloExchangeOn(qgd_case_s* c, qgd_comm_s* comm, const int32_t* peers, int nSlots, hipStream_t stream) {
    qgd_device_s* d = c->dev;
    const int n = std::min<int>(nSlots, (int)d->halo.size());
    checkHaloPeers(comm, peers, n);
    ensureHaloBuffers(c);
    Launcher L = launcherOf(c);
    L.pre = nullptr; L.post = nullptr; L.stream = stream;
    (void)hipGetLastError();
    for (int s = 0; s < n; ++s) {
        const qgd_device_s::HaloSlot& h = d->halo[s];
        if (peers[s] < 0 || !h.nSend) continue;
        launchHaloPack(L, c->view, c->gas, h.send, h.nSend, h.sendBF, h.nSendBF, c->sendBuf[s], true);
    }
    HIP_CHECK(hipGetLastError());
    RCCL_CHECK(rcclRef().groupStart());
    try {
        for (int s = 0; s < n; ++s) {
            const qgd_device_s::HaloSlot& h = d->halo[s];
            if (peers[s] < 0) continue;
            const size_t ns = QGD_HALO_CELL_DOUBLES_HOST * (size_t)h.nSend + 12 * (size_t)h.nSendBF, nr = QGD_HALO_CELL_DOUBLES_HOST * (size_t)h.nGhost + 12 * (size_t)h.nGhostBF;
            if (ns) RCCL_CHECK(rcclRef().send(c->sendBuf[s], ns, ncclFloat64, peers[s], comm->comm, stream));
            if (nr) RCCL_CHECK(rcclRef().recv(c->recvBuf[s], nr, ncclFloat64, peers[s], comm->comm, stream));
        }
    } catch (...) {
        (void)rcclRef().groupEnd();   // never leave the group open behind a failed call: the next collective would join it
        throw;
    }
    RCCL_CHECK(rcclRef().groupEnd());
    for (int s = 0; s < n; ++s) {
        const qgd_device_s::HaloSlot& h = d->halo[s];
        if (peers[s] < 0 || !h.nGhost) continue;
        launchHaloPack(L, c->view, c->gas, h.ghost, h.nGhost, h.ghostBF, h.nGhostBF, c->recvBuf[s], false);
    }
    HIP_CHECK(hipGetLastError());
}
int qgd_case_halo_exchange(qgd_case_t c, qgd_comm_t comm, const int32_t* peers, int nSlots) {
    QGD_TRY
    if (!c) return fail(QGD_ERR_INVALID, "null case");
    if (c->dev->halo.empty() || nSlots <= 0) return QGD_OK;  // unsharded: nothing to exchange
    if (!comm || !peers) return fail(QGD_ERR_INVALID, "qgd_case_halo_exchange: null argument");
    HIP_CHECK(hipSetDevice(c->dev->deviceId));
    haloExchangeOn(c, comm, peers, nSlots, c->stream());
    return QGD_OK;  // stream-ordered on the case's stream
    QGD_CATCH
}
int qgd_case_allreduce_max(qgd_case_t c, qgd_comm_t comm) {
    QGD_TRY
    if (!c) return fail(QGD_ERR_INVALID, "null case");
    if (!comm || comm->nRanks == 1) return QGD_OK;
    HIP_CHECK(hipSetDevice(c->dev->deviceId));
    RCCL_CHECK(rcclRef().allReduce(c->view.red, c->view.red, 2, ncclFloat64, ncclMax, comm->comm, c->stream()));
    return QGD_OK;
    QGD_CATCH
}
// the implicit branch's own messages (kind 1 grad U, 2 U, 3 search direction) over the library's transport, on the case's stream
static void implHaloExchangeOn(qgd_case_s* c, qgd_comm_s* comm, const int32_t* peers, int nSlots, int kind) {
    qgd_device_s* d = c->dev;
    const int n = std::min<int>(nSlots, (int)d->halo.size());
    checkHaloPeers(comm, peers, n);
    if (c->implSendBuf.size() != d->halo.size()) {
        c->implSendBuf.assign(d->halo.size(), nullptr);
        c->implRecvBuf.assign(d->halo.size(), nullptr);
        for (size_t s2 = 0; s2 < d->halo.size(); ++s2) {
            c->implSendBuf[s2] = c->arena.alloc<double>(9 * (size_t)d->halo[s2].nSend);
            c->implRecvBuf[s2] = c->arena.alloc<double>(9 * (size_t)d->halo[s2].nGhost);
        }
    }
    hipStream_t stream = c->stream();
    const size_t w = (size_t)implicitHaloWidth(c->implSolver, kind);
    for (int s2 = 0; s2 < n; ++s2)
        if (peers[s2] >= 0 && implHaloMove(c, s2, kind, c->implSendBuf[s2], true, stream) != QGD_OK) throw std::invalid_argument(g_lastError);
    RCCL_CHECK(rcclRef().groupStart());
    try {
        for (int s2 = 0; s2 < n; ++s2) {
            const qgd_device_s::HaloSlot& h = d->halo[s2];
            if (peers[s2] < 0) continue;
            if (h.nSend) RCCL_CHECK(rcclRef().send(c->implSendBuf[s2], w * h.nSend, ncclFloat64, peers[s2], comm->comm, stream));
            if (h.nGhost) RCCL_CHECK(rcclRef().recv(c->implRecvBuf[s2], w * h.nGhost, ncclFloat64, peers[s2], comm->comm, stream));
        }
    } catch (...) { (void)rcclRef().groupEnd(); throw; }
    RCCL_CHECK(rcclRef().groupEnd());
    for (int s2 = 0; s2 < n; ++s2)
        if (peers[s2] >= 0 && implHaloMove(c, s2, kind, c->implRecvBuf[s2], false, stream) != QGD_OK) throw std::invalid_argument(g_lastError);
}
// the mid-assembly message over the library's transport, on the case's stream
static void midExchangeOn(qgd_case_s* c, qgd_comm_s* comm, const int32_t* peers, int nSlots) {
    qgd_device_s* d = c->dev;
    const int n = std::min<int>(nSlots, (int)d->halo.size());
    checkHaloPeers(comm, peers, n);
    if (c->midSendBuf.size() != d->halo.size()) {
        c->midSendBuf.assign(d->halo.size(), nullptr);
        c->midRecvBuf.assign(d->halo.size(), nullptr);
        for (size_t s2 = 0; s2 < d->halo.size(); ++s2) {
            c->midSendBuf[s2] = c->arena.alloc<double>(2 * (size_t)d->halo[s2].nSendBF);
            c->midRecvBuf[s2] = c->arena.alloc<double>(2 * (size_t)d->halo[s2].nGhostBF);
        }
    }
    hipStream_t stream = c->stream();
    for (int s2 = 0; s2 < n; ++s2)
        if (peers[s2] >= 0 && midHaloMove(c, s2, c->midSendBuf[s2], true, stream) != QGD_OK) throw std::invalid_argument(g_lastError);
    RCCL_CHECK(rcclRef().groupStart());
    try {
        for (int s2 = 0; s2 < n; ++s2) {
            const qgd_device_s::HaloSlot& h = d->halo[s2];
            if (peers[s2] < 0) continue;
            if (h.nSendBF) RCCL_CHECK(rcclRef().send(c->midSendBuf[s2], 2 * (size_t)h.nSendBF, ncclFloat64, peers[s2], comm->comm, stream));
            if (h.nGhostBF) RCCL_CHECK(rcclRef().recv(c->midRecvBuf[s2], 2 * (size_t)h.nGhostBF, ncclFloat64, peers[s2], comm->comm, stream));
        }
    } catch (...) { (void)rcclRef().groupEnd(); throw; }
    RCCL_CHECK(rcclRef().groupEnd());
    for (int s2 = 0; s2 < n; ++s2)
        if (peers[s2] >= 0 && midHaloMove(c, s2, c->midRecvBuf[s2], false, stream) != QGD_OK) throw std::invalid_argument(g_lastError);
}
int qgd_case_step_sharded(qgd_case_t c, qgd_comm_t comm, const int32_t* peers, int nSlots, int overlapped) {
    QGD_TRY
    if (!c) return fail(QGD_ERR_INVALID, "null case");
    if (!c->fieldsSet) return fail(QGD_ERR_INVALID, "qgd_case_step_sharded: call qgd_case_set_fields first");
    const bool sharded = !c->dev->halo.empty() && nSlots > 0;
    if (sharded && (!comm || !peers)) return fail(QGD_ERR_INVALID, "qgd_case_step_sharded: null argument");
    HIP_CHECK(hipSetDevice(c->dev->deviceId));
    const bool adjust = c->opt.adjustTimeStep != 0;
    if (sharded) checkHaloPeers(comm, peers, std::min<int>(nSlots, (int)c->dev->halo.size()));   // before the first message of the step
    if (sharded && midExchangeNeeded(c)) {
        stepAssemble(c, 1);
        midExchangeOn(c, comm, peers, nSlots);
        stepAssemble(c, 2);
    } else stepAssemble(c);
    if (adjust && comm && comm->nRanks > 1)
        RCCL_CHECK(rcclRef().allReduce(c->view.red, c->view.red, 2, ncclFloat64, ncclMax, comm->comm, c->stream()));
    if (!sharded) { stepAdvance(c, 0); HIP_CHECK(hipGetLastError()); return QGD_OK; }
    if (c->opt.implicitDiffusion) {
        // the reference's default branch on shards: the dot products of its four solves are ncclAllReduce of the control block, the
        // gradients, the new velocity and the search directions travel as grouped send/recv pairs, then the state message as usual
        SolveHooks hooks;
        hipStream_t st = c->stream();
        if (comm->nRanks > 1) {
            hooks.allreduce = [&](double* ptr, int n) { RCCL_CHECK(rcclRef().allReduce(ptr, ptr, (size_t)n, ncclFloat64, ncclSum, comm->comm, st)); };
            // the Chebyshev solves' spectral bound: MAX over the ranks (op 3; 2 = SUM)
            hooks.allreduceBuf = [&](double* ptr, int64_t n, int op) {
                RCCL_CHECK(rcclRef().allReduce(ptr, ptr, (size_t)n, ncclFloat64, op == 3 ? ncclMax : ncclSum, comm->comm, st));
            };
        }
        hooks.haloDirection = [&]() { implHaloExchangeOn(c, comm, peers, nSlots, 3); };
        hooks.haloGuess = [&]() { implHaloExchangeOn(c, comm, peers, nSlots, 4); };
        implicitAdvanceWith(c, &hooks, [&](int kind) { implHaloExchangeOn(c, comm, peers, nSlots, kind); });
        haloExchangeOn(c, comm, peers, nSlots, st);
        HIP_CHECK(hipGetLastError());
        return QGD_OK;
    }
    if (!overlapped) {
        stepAdvance(c, 0);
        haloExchangeOn(c, comm, peers, nSlots, c->stream());
    } else {
        // boundary layer first; pack / send / recv / unpack on the library's halo stream while the compute stream
        // updates the remaining cells; the next assembly waits for the unpack
        if (!c->ownHaloStream) {
            HIP_CHECK(hipStreamCreateWithFlags(&c->ownHaloStream, hipStreamNonBlocking));
            HIP_CHECK(hipEventCreateWithFlags(&c->evLayerDone, hipEventDisableTiming));
            HIP_CHECK(hipEventCreateWithFlags(&c->evUnpacked, hipEventDisableTiming));
        }
        stepAdvance(c, 1);
        HIP_CHECK(hipEventRecord(c->evLayerDone, c->stream()));
        HIP_CHECK(hipStreamWaitEvent(c->ownHaloStream, c->evLayerDone, 0));
        haloExchangeOn(c, comm, peers, nSlots, c->ownHaloStream);
        HIP_CHECK(hipEventRecord(c->evUnpacked, c->ownHaloStream));
        stepAdvance(c, 2);
        HIP_CHECK(hipStreamWaitEvent(c->stream(), c->evUnpacked, 0));
    }
    HIP_CHECK(hipGetLastError());
    return QGD_OK;  // stream-ordered: qgd_case_stream_sync waits
    QGD_CATCH
}

// ---- the QHD case over the library's own transport ----------------------------------------------------------------------
// pack -> grouped send/recv -> unpack of message kind 0 (state), 1 (p) or 2 (search direction of the pressure solve) on the
// device's stream; buffers owned by the case, sized once for the widest kind
static void qhdHaloExchangeOn(qgd_qhd_case_s* c, qgd_comm_s* comm, const int32_t* peers, int nSlots, int kind) {
    qgd_device_s* d = c->dev;
    const int n = std::min<int>(nSlots, (int)d->halo.size());
    checkHaloPeers(comm, peers, n);
    if (c->sendBuf.size() != d->halo.size()) {
        c->sendBuf.assign(d->halo.size(), nullptr);
        c->recvBuf.assign(d->halo.size(), nullptr);
        for (size_t s2 = 0; s2 < d->halo.size(); ++s2) {
            const qgd_device_s::HaloSlot& h = d->halo[s2];
            c->sendBuf[s2] = c->arena.alloc<double>(10 * (size_t)h.nSend + 4 * (size_t)h.nSendBF);
            c->recvBuf[s2] = c->arena.alloc<double>(10 * (size_t)h.nGhost + 4 * (size_t)h.nGhostBF);
        }
    }
    int pc, pf;
    qhdHaloWidths(kind, pc, pf);
    hipStream_t stream = d->stream;
    for (int s2 = 0; s2 < n; ++s2)
        if (peers[s2] >= 0 && qhdHaloMove(c, s2, kind, c->sendBuf[s2], true, stream) != QGD_OK) throw std::invalid_argument(g_lastError);
    RCCL_CHECK(rcclRef().groupStart());
    try {
        for (int s2 = 0; s2 < n; ++s2) {
            const qgd_device_s::HaloSlot& h = d->halo[s2];
            if (peers[s2] < 0) continue;
            const size_t ns = (size_t)pc * h.nSend + (size_t)pf * h.nSendBF, nr = (size_t)pc * h.nGhost + (size_t)pf * h.nGhostBF;
            if (ns) RCCL_CHECK(rcclRef().send(c->sendBuf[s2], ns, ncclFloat64, peers[s2], comm->comm, stream));
            if (nr) RCCL_CHECK(rcclRef().recv(c->recvBuf[s2], nr, ncclFloat64, peers[s2], comm->comm, stream));
        }
    } catch (...) { (void)rcclRef().groupEnd(); throw; }
    RCCL_CHECK(rcclRef().groupEnd());
    for (int s2 = 0; s2 < n; ++s2)
        if (peers[s2] >= 0 && qhdHaloMove(c, s2, kind, c->recvBuf[s2], false, stream) != QGD_OK) throw std::invalid_argument(g_lastError);
}
int qgd_qhd_case_halo_exchange(qgd_qhd_case_t c, qgd_comm_t comm, const int32_t* peers, int nSlots, int kind) {
    QGD_TRY
    if (!c || kind < 0 || kind > 4) return fail(QGD_ERR_INVALID, "bad argument");
    if (c->dev->halo.empty() || nSlots <= 0) return QGD_OK;
    if (!comm || !peers) return fail(QGD_ERR_INVALID, "qgd_qhd_case_halo_exchange: null argument");
    HIP_CHECK(hipSetDevice(c->dev->deviceId));
    qhdHaloExchangeOn(c, comm, peers, nSlots, kind);
    return QGD_OK;
    QGD_CATCH
}
// one QHDFoam step of a sharded case [QHDFoam_8C L83-139]: the dot products of the pressure solve are ncclAllReduce of the
// control block, its search direction and the states travel as grouped send/recv pairs; the only host waits are the
// run-ahead checks of the solve (pressureSolveRun)
int qgd_qhd_case_step_sharded(qgd_qhd_case_t c, qgd_comm_t comm, const int32_t* peers, int nSlots, int32_t nSteps) {
    QGD_TRY
    if (!c) return fail(QGD_ERR_INVALID, "null case");
    if (!c->fieldsSet) return fail(QGD_ERR_INVALID, "qgd_qhd_case_step_sharded: call qgd_qhd_case_set_fields first");
    qgd_device_s* d = c->dev;
    const bool sharded = !d->halo.empty() && nSlots > 0;
    if (sharded && (!comm || !peers)) return fail(QGD_ERR_INVALID, "qgd_qhd_case_step_sharded: null argument");
    HIP_CHECK(hipSetDevice(d->deviceId));
    if (sharded) checkHaloPeers(comm, peers, std::min<int>(nSlots, (int)d->halo.size()));
    SolveHooks hooks;
    if (comm && comm->nRanks > 1)
        hooks.allreduce = [&](double* ptr, int n) { RCCL_CHECK(rcclRef().allReduce(ptr, ptr, (size_t)n, ncclFloat64, ncclSum, comm->comm, d->stream)); };
    if (sharded) hooks.haloDirection = [&]() { qhdHaloExchangeOn(c, comm, peers, nSlots, 2); };
    if (sharded) hooks.haloMg = [&]() { qhdHaloExchangeOn(c, comm, peers, nSlots, 3); };
    if (comm && comm->nRanks > 1)
        hooks.allreduceBuf = [&](double* ptr, int64_t n, int op) {
            RCCL_CHECK(rcclRef().allReduce(ptr, ptr, (size_t)n, ncclFloat64, op == 3 ? ncclMax : ncclSum, comm->comm, d->stream));
        };
    std::function<void(int)> haloState;
    if (sharded) haloState = [&](int kind) { qhdHaloExchangeOn(c, comm, peers, nSlots, kind); };
    for (int i = 0; i < nSteps; ++i) qhdStepWith(c, &hooks, haloState);
    HIP_CHECK(hipStreamSynchronize(d->stream));
    return QGD_OK;
    QGD_CATCH
}

// ---- accessors --------------------------------------------------------------------
int qgd_case_get_field(qgd_case_t c, const char* name, double* out, int64_t outDoubles) {
    QGD_TRY
    if (!c || !name || !out) return fail(QGD_ERR_INVALID, "qgd_case_get_field: null argument");
    HIP_CHECK(hipSetDevice(c->dev->deviceId));
    HIP_CHECK(hipStreamSynchronize(c->stream()));
    const MeshView& m = c->dev->view;
    const GasModel& g = c->gas;
    std::string s(name);
    bool bnd = false;
    const std::string suffix = ".boundary";
    if (s.size() > suffix.size() && s.compare(s.size() - suffix.size(), suffix.size(), suffix) == 0) {
        bnd = true;
        s = s.substr(0, s.size() - suffix.size());
    }
    // face fields from the debug buffer
    static const std::map<std::string, std::pair<int, int>> faceSlots = {
        {"phiJm", {DBG_PHIJM, 1}}, {"phiJmU", {DBG_PHIJMU, 3}}, {"phiP", {DBG_PHIP, 3}}, {"phiPi", {DBG_PHIPI, 3}},
        {"phiJmH", {DBG_PHIJMH, 1}}, {"phiQ", {DBG_PHIQ, 1}}, {"phiPiU", {DBG_PHIPIU, 1}}, {"phiwStar", {DBG_PHIW, 1}},
        {"phi", {DBG_PHI, 1}}, {"tauQGDf", {DBG_TAU, 1}}, {"gradUf", {DBG_GRADU, 9}}, {"gradef", {DBG_GRADE, 3}},
        {"gradRhof", {DBG_GRADRHO, 3}}, {"gradPf", {DBG_GRADP, 3}}};
    if (!bnd && s == "hQGDf") {
        if ((int64_t)c->dev->hf.size() > outDoubles) return fail(QGD_ERR_INVALID, "output too small");
        std::copy(c->dev->hf.begin(), c->dev->hf.end(), out);
        return QGD_OK;
    }
    // face fields of the implicitDiffusion branch, as the last step left them [updateFluxes.H L107-111, QGDUEqn.H L72-74]
    if (!bnd && (s == "phiTauMC" || s == "phiSigmaDotU")) {
        if (!c->impl.phiTau) return fail(QGD_ERR_INVALID, "qgd_case_get_field: " + s + " exists with implicitDiffusion true only");
        const int nc = s == "phiTauMC" ? 3 : 1;
        if ((int64_t)m.nF * nc > outDoubles) return fail(QGD_ERR_INVALID, "output too small");
        HIP_CHECK(hipStreamSynchronize(c->stream()));
        // stored at the faces' slot-major positions (qgd_implicit.hip implFaceKernel): internal face f at fpos[f], patch faces at their label
        std::vector<double> tmp((size_t)m.nF);
        std::vector<int32_t> fpos((size_t)m.nIF);
        if (m.nIF) HIP_CHECK(hipMemcpy(fpos.data(), m.fpos, sizeof(int32_t) * (size_t)m.nIF, hipMemcpyDeviceToHost));
        for (int k = 0; k < nc; ++k) {
            const double* src = nc == 3 ? c->impl.phiTau + (size_t)k * m.nF : c->impl.phiSig;
            HIP_CHECK(hipMemcpy(tmp.data(), src, sizeof(double) * (size_t)m.nF, hipMemcpyDeviceToHost));
            for (int64_t f = 0; f < m.nF; ++f) out[f * nc + k] = tmp[f < m.nIF ? (size_t)fpos[f] : (size_t)f];
        }
        return QGD_OK;
    }
    auto fs = faceSlots.find(s);
    if (!bnd && fs != faceSlots.end()) {
        if (!c->dbgBuf) return fail(QGD_ERR_INVALID, "face fields are materialised by qgd_case_update_fluxes; call it first");
        const int slot = fs->second.first, nc = fs->second.second;
        if ((int64_t)m.nF * nc > outDoubles) return fail(QGD_ERR_INVALID, "output too small");
        std::vector<double> tmp((size_t)m.nF);
        for (int k = 0; k < nc; ++k) {
            HIP_CHECK(hipMemcpy(tmp.data(), c->dbgBuf + (size_t)(slot + k) * m.nF, sizeof(double) * (size_t)m.nF, hipMemcpyDeviceToHost));
            for (int64_t f = 0; f < m.nF; ++f) out[f * nc + k] = tmp[f];
        }
        return QGD_OK;
    }
    // cell / boundary fields: extracted from the records on the device, one dense copy back
    static const std::map<std::string, int> cellFields = {
        {"rho", XF_RHO}, {"U", XF_U}, {"p", XF_P}, {"e", XF_E}, {"T", XF_T}, {"rhoU", XF_RHOU}, {"rhoE", XF_RHOE}, {"c", XF_C},
        {"psi", XF_PSI}, {"mu", XF_MU}, {"alphau", XF_ALPHAU}, {"tauQGD", XF_TAUQGD}, {"muQGD", XF_MUQGD},
        {"alphauQGD", XF_ALPHAUQGD}, {"hQGD", XF_HQGD}, {"H", XF_H}, {"gamma", XF_GAMMA}};
    auto cf = cellFields.find(s);
    if (cf == cellFields.end()) return fail(QGD_ERR_UNKNOWN_NAME, "qgd_case_get_field: unknown field " + s);
    const int64_t n = bnd ? m.nBF : m.nC;
    const int nc = (cf->second == XF_U || cf->second == XF_RHOU) ? 3 : 1;
    if (n * nc > outDoubles) return fail(QGD_ERR_INVALID, "output too small");
    if (n == 0) return QGD_OK;
    double* tmp = nullptr;
    HIP_CHECK(hipMalloc((void**)&tmp, sizeof(double) * (size_t)(n * nc)));
    try {
        (void)hipGetLastError();
        launchExtractField(c->stream(), bnd ? c->view.bA : c->view.A, bnd ? c->view.bB : c->view.B, bnd ? nullptr : c->view.rE,
                           bnd ? m.hQGDb : m.hQGD, bnd ? c->view.aQb : c->view.aQ, n, g, cf->second, tmp);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipStreamSynchronize(c->stream()));
        HIP_CHECK(hipMemcpy(out, tmp, sizeof(double) * (size_t)(n * nc), hipMemcpyDeviceToHost));
    } catch (...) { (void)hipFree(tmp); throw; }
    (void)hipFree(tmp);
    return QGD_OK;
    QGD_CATCH
}

int qgd_case_info(qgd_case_t c, double info[6]) {
    QGD_TRY
    if (!c || !info) return fail(QGD_ERR_INVALID, "null argument");
    HIP_CHECK(hipSetDevice(c->dev->deviceId));
    Launcher L = launcherOf(c);
    L.pre = nullptr; L.post = nullptr;
    (void)hipGetLastError();
    launchCellMinReduce(L, c->view);  // min(rho), min(e) since the previous query [QGDFoam_8C L142]
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipStreamSynchronize(c->stream()));
    double red[4], dt[3];
    HIP_CHECK(hipMemcpy(red, c->view.red, sizeof(red), hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(dt, c->view.dt, sizeof(dt), hipMemcpyDeviceToHost));
    info[0] = c->opt.adjustTimeStep ? dt[1] : c->time;
    info[1] = dt[0];
    info[2] = dt[2];
    info[3] = red[2];
    info[4] = red[3];
    info[5] = (double)c->steps;
    return QGD_OK;
    QGD_CATCH
}

int qgd_case_fused_info(qgd_case_t c, int64_t info[8]) {
    if (!c || !info) return fail(QGD_ERR_INVALID, "null argument");
    const MeshView& v = c->dev->view;
    info[0] = c->fused ? 1 : (c->fusedImpl ? 2 : (c->fusedAdj ? 3 : 0));   // 2: the implicitDiffusion branch's block-fused assembly of the U systems; 3: Courant-number control (blocks up to their sums + a cell kernel)
    info[1] = (c->fused || c->fusedImpl || c->fusedAdj) ? v.fuBlocks : 0;
    info[2] = c->fused ? c->dev->fusedFacesComputed : 0;
    info[3] = (c->fused || c->fusedAdj) ? v.fuLds : (c->fusedImpl ? v.fuLdsImpl : 0);
    info[4] = c->fused ? c->dev->fusedCellsStaged : 0;
    info[5] = c->fused ? c->dev->fusedCellsStagedFull : 0;
    info[6] = c->fused ? c->dev->fusedVertsStaged : 0;
    info[7] = c->fused ? v.fuLayerBlocks : 0;
    return QGD_OK;
}

int qgd_case_implicit_info(qgd_case_t c, double info[16]) {
    QGD_TRY
    if (!c || !info) return fail(QGD_ERR_INVALID, "null argument");
    for (int k = 0; k < 16; ++k) info[k] = 0.0;
    info[13] = c->opt.implicitDiffusion ? 1.0 : 0.0;
    if (!c->implSolver) return QGD_OK;
    info[13] = implicitSolverChebyshev(c->implSolver) ? 2.0 : 1.0;   // 1: conjugate gradients (QGD_IMPL_SOLVER=pcg), 2: Chebyshev iteration
    HIP_CHECK(hipSetDevice(c->dev->deviceId));
    implicitSolverSetStream(c->implSolver, c->stream());
    int it[4]; double r0[4], r1[4], bad = 0, stalled = 0;
    implicitSolverInfo(c->implSolver, it, r0, r1, &bad, &stalled);
    for (int k = 0; k < 4; ++k) { info[k] = it[k]; info[4 + k] = r0[k]; info[8 + k] = r1[k]; }
    info[12] = bad; info[14] = stalled;
    return QGD_OK;
    QGD_CATCH
}
int qgd_case_implicit_apply_time(qgd_case_t c, int reps, double info[2]) {
    QGD_TRY
    if (!c || !info || reps <= 0) return fail(QGD_ERR_INVALID, "bad argument");
    if (!c->implSolver || !c->fieldsSet) return fail(QGD_ERR_INVALID, "qgd_case_implicit_apply_time: an implicitDiffusion case after qgd_case_set_fields");
    HIP_CHECK(hipSetDevice(c->dev->deviceId));
    implicitSolverSetStream(c->implSolver, c->stream());
    HIP_CHECK(hipStreamSynchronize(c->stream()));
    (void)hipGetLastError();
    int rows = 0;
    info[0] = implicitApplyMs(c->implSolver, c->impl, reps, &rows);
    info[1] = rows;
    return QGD_OK;
    QGD_CATCH
}
// the branch's own halo messages and control block on a shard (kinds 1 grad U, 2 U, 3 search direction; the state message is
// qgd_case_halo_*): counts in doubles
int qgd_case_implicit_halo_count(qgd_case_t c, int slot, int kind, int64_t* sendCount, int64_t* recvCount) {
    if (!c || slot < 0 || kind < 1 || kind > 4 || !sendCount || !recvCount) return fail(QGD_ERR_INVALID, "bad argument");
    *sendCount = *recvCount = 0;
    if (!c->implSolver) return fail(QGD_ERR_INVALID, "qgd_case_implicit_halo_count: not an implicitDiffusion case");
    if (slot >= (int)c->dev->halo.size()) return QGD_OK;
    const int w = kind >= 3 ? 3 : implicitHaloWidth(c->implSolver, kind);   // kinds 3, 4: room for the widest solve
    *sendCount = (int64_t)w * c->dev->halo[slot].nSend;
    *recvCount = (int64_t)w * c->dev->halo[slot].nGhost;
    return QGD_OK;
}
static int implHaloMove(qgd_case_s* c, int slot, int kind, double* buf, bool pack, hipStream_t stream) {
    qgd_device_s* d = c->dev;
    if (slot >= (int)d->halo.size()) return QGD_OK;
    const qgd_device_s::HaloSlot& h = d->halo[slot];
    const int32_t n = pack ? h.nSend : h.nGhost;
    if (n == 0) return QGD_OK;
    if (!buf) return fail(QGD_ERR_INVALID, "null buffer");
    launchImplicitHalo(stream, d->view, c->view, c->impl, c->implSolver, kind, pack ? h.send : h.ghost, n, buf, pack);
    return QGD_OK;
}
int qgd_case_implicit_halo_pack(qgd_case_t c, int slot, int kind, double* sendBufDevice) {
    QGD_TRY
    if (!c || slot < 0 || kind < 1 || kind > 4 || !c->implSolver) return fail(QGD_ERR_INVALID, "bad argument");
    HIP_CHECK(hipSetDevice(c->dev->deviceId));
    return implHaloMove(c, slot, kind, sendBufDevice, true, c->stream());
    QGD_CATCH
}
int qgd_case_implicit_halo_unpack(qgd_case_t c, int slot, int kind, const double* recvBufDevice) {
    QGD_TRY
    if (!c || slot < 0 || kind < 1 || kind > 4 || !c->implSolver) return fail(QGD_ERR_INVALID, "bad argument");
    HIP_CHECK(hipSetDevice(c->dev->deviceId));
    return implHaloMove(c, slot, kind, const_cast<double*>(recvBufDevice), false, c->stream());
    QGD_CATCH
}
// control block of the solve in flight: 68 doubles, slot-major (control[slot * 4 + component]); a sharded run SUM-reduces
// control[0..12), [12..16), [16..20), [20..24), [24..32) after solver phases 0, 1, 2, 3, 4.  status: {all components done,
// right-hand sides of the solve in flight}
int qgd_case_implicit_control(qgd_case_t c, double control[68], int set) {
    QGD_TRY
    if (!c || !control || !c->implSolver) return fail(QGD_ERR_INVALID, "bad argument");
    HIP_CHECK(hipSetDevice(c->dev->deviceId));
    hipStream_t st = c->stream();
    if (set) HIP_CHECK(hipMemcpyAsync(implicitSolverCtl(c->implSolver), control, 68 * sizeof(double), hipMemcpyHostToDevice, st));
    else HIP_CHECK(hipMemcpyAsync(control, implicitSolverCtl(c->implSolver), 68 * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    return QGD_OK;
    QGD_CATCH
}
int qgd_case_implicit_control_ptr(qgd_case_t c, void** devicePtr) {
    if (!c || !devicePtr || !c->implSolver) return fail(QGD_ERR_INVALID, "bad argument");
    *devicePtr = implicitSolverCtl(c->implSolver);
    return QGD_OK;
}
int qgd_case_implicit_solve_status(qgd_case_t c, double status[2]) {
    QGD_TRY
    if (!c || !status || !c->implSolver) return fail(QGD_ERR_INVALID, "bad argument");
    HIP_CHECK(hipSetDevice(c->dev->deviceId));
    implicitSolverSetStream(c->implSolver, c->stream());
    int it[3]; double r0[3], r1[3];
    implicitSolveStatus(c->implSolver, &status[0], it, r0, r1);
    status[1] = implicitSolverRhs(c->implSolver);
    return QGD_OK;
    QGD_CATCH
}

int qgd_struct_sizes(int64_t sizes[4]) {
    if (!sizes) return fail(QGD_ERR_INVALID, "null argument");
    sizes[0] = (int64_t)sizeof(qgd_case_options); sizes[1] = (int64_t)sizeof(qgd_qhd_options);
    sizes[2] = (int64_t)sizeof(qgd_poisson_control); sizes[3] = QGD_ABI_VERSION;
    return QGD_OK;
}

// ---- measurement -------------------------------------------------------------------
int qgd_case_timing(qgd_case_t c, int enable) {
    if (!c) return fail(QGD_ERR_INVALID, "null case");
    if (!enable) harvestTiming(c);
    c->timing = enable != 0;
    return QGD_OK;
}
int qgd_case_kernel_time(qgd_case_t c, int k, double* totalMs, int64_t* launches) {
    if (!c || k < 0 || k >= QGD_K_COUNT || !totalMs || !launches) return fail(QGD_ERR_INVALID, "bad argument");
    (void)hipSetDevice(c->dev->deviceId);
    harvestTiming(c);
    *totalMs = c->totalMs[k];
    *launches = c->launches[k];
    return QGD_OK;
}
int qgd_case_timing_reset(qgd_case_t c) {
    if (!c) return fail(QGD_ERR_INVALID, "null case");
    (void)hipSetDevice(c->dev->deviceId);
    harvestTiming(c);
    for (int k = 0; k < QGD_K_COUNT; ++k) { c->totalMs[k] = 0; c->launches[k] = 0; }
    return QGD_OK;
}
int qgd_case_device_bytes(qgd_case_t c, int64_t* bytes) {
    if (!c || !bytes) return fail(QGD_ERR_INVALID, "null argument");
    *bytes = c->arena.bytes + c->dev->arena.bytes;
    return QGD_OK;
}

}  // extern "C"
