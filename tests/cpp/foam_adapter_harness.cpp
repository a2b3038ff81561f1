// Control-flow harness of the OpenFOAM adapter qgdsolver_amd/foam/hipStencil.{H,C}, compiled against tests/cpp/foam_stub
// (NOT OpenFOAM -- see foamStub.H) and against a recording mock of the eight C-ABI entries the adapter calls.  It asserts the
// three behaviours VERDICT r03 asked for and nothing about numbers:
//   1. only the GaussVolPoint word re-evaluates the boundary conditions of its input
//      [GaussVolPointStencil_8C L73/91/103/121 vs reducedFaceNormalStencil_8C L69-108, leastSquaresStencil_8C L145-275];
//   2. Pstream::parRun() is refused with FatalError;
//   3. a processor patch that has faces is refused with FatalError.
// Prints one line per check and exits non-zero on the first failure.
#include <cstdio>
#include <cstring>
#include <string>

#include "hipStencil.H"

// ---- definitions of the stub's externs ------------------------------------------------------------------------------
namespace Foam
{
const scalar pTraits<scalar>::zero = 0.0;
const vector pTraits<vector>::zero = {{0, 0, 0}};
const tensor pTraits<tensor>::zero = {{0, 0, 0, 0, 0, 0, 0, 0, 0}};
const dimensionSet dimLength;
errorStream FatalError;
const newline nl = newline();
bool Pstream::parRun_ = false;
}

// ---- recording mock of the C-ABI (the real library needs a GPU; this harness is about the adapter's control flow) -----
static int g_meshCreates = 0, g_lastPatchTypes[16], g_nPatchTypes = 0, g_opCalls = 0;
static std::string g_lookupWord;
extern "C" {
const char* qgd_last_error(void) { return "mock"; }
int qgd_mesh_create(int32_t, const double*, int32_t, const int32_t*, const int32_t*, int32_t, const int32_t*, const int32_t*, int32_t,
                    int32_t nPatches, const int32_t*, const int32_t*, const int32_t* patchType, qgd_mesh_t* out) {
    ++g_meshCreates;
    g_nPatchTypes = nPatches;
    for (int i = 0; i < nPatches && i < 16; ++i) g_lastPatchTypes[i] = patchType[i];
    *out = reinterpret_cast<qgd_mesh_t>(0x1);
    return QGD_OK;
}
int qgd_mesh_set_geometry(qgd_mesh_t, const double*, const double*, const double*, const double*) { return QGD_OK; }
int qgd_mesh_free(qgd_mesh_t) { return QGD_OK; }
int qgd_device_create(qgd_mesh_t, int, qgd_device_t* out) { *out = reinterpret_cast<qgd_device_t>(0x2); return QGD_OK; }
int qgd_device_free(qgd_device_t) { return QGD_OK; }
int qgd_stencil_lookup(qgd_device_t, const char* w, int* id) { g_lookupWord = w; *id = 7; return QGD_OK; }
int qgd_fvsc_grad_s(qgd_device_t, int, const double*, const double*, double*) { ++g_opCalls; return QGD_OK; }
int qgd_fvsc_grad_v(qgd_device_t, int, const double*, const double*, double*) { ++g_opCalls; return QGD_OK; }
int qgd_fvsc_div_v(qgd_device_t, int, const double*, const double*, double*) { ++g_opCalls; return QGD_OK; }
int qgd_fvsc_div_t(qgd_device_t, int, const double*, const double*, double*) { ++g_opCalls; return QGD_OK; }
}

using namespace Foam;

// two hexahedra side by side: 12 points, 11 faces (1 internal), patches {walls: 8 faces, frontBack (empty): 2, [proc: 0 or 1]}
static void twoCells(fvMesh& m, bool withProcessorFaces) {
    m.points_.setSize(12);
    m.nCells_ = 2;
    const label nF = 11;
    m.faces_.setSize(nF);
    for (label f = 0; f < nF; ++f) { m.faces_[f] = face(4); for (label k = 0; k < 4; ++k) m.faces_[f][k] = (f + k) % 12; }
    m.owner_.setSize(nF); m.owner_ = 0;
    m.neighbour_.setSize(1); m.neighbour_[0] = 1;
    m.Sf_.setSize(nF); m.Cf_.setSize(nF); m.C_.setSize(2); m.V_.setSize(2);
    const label nProc = withProcessorFaces ? 1 : 0;
    m.pbm_.append(polyPatch("walls", 1, 8 - nProc));
    m.pbm_.append(polyPatch("frontBack", 9 - nProc, 2));
    m.pbm_.append(polyPatch("procBoundary0to1", 11 - nProc, nProc));
    m.boundary_.append(new fvPatch("walls", 8 - nProc));
    m.boundary_.append(new emptyFvPatch("frontBack", 2));
    m.boundary_.append(new processorFvPatch("procBoundary0to1", nProc));
}

static int failures = 0;
static void check(bool ok, const char* what) {
    std::printf("%s %s\n", ok ? "ok  " : "FAIL", what);
    if (!ok) ++failures;
}

int main() {
    // 1. the BC side effect, word by word, for all four operators
    const char* words[3] = {"hipReduced", "hipLeastSquares", "hipGaussVolPoint"};
    const char* targets[3] = {"reduced", "leastSquares", "GaussVolPoint"};
    for (int w = 0; w < 3; ++w) {
        fvMesh mesh;
        twoCells(mesh, false);
        IOobject io(words[w], "0", mesh);
        fvsc::fvscStencil* st = fvsc::tableEntry::New(words[w], io);   // what fvscStencil::New does with fvSchemes' word
        check(st != nullptr, (std::string("run-time selection finds ") + words[w]).c_str());
        if (!st) continue;
        check(g_lookupWord == targets[w], (std::string(words[w]) + " asks the library for '" + targets[w] + "'").c_str());
        check(g_nPatchTypes == 3 && g_lastPatchTypes[0] == QGD_PATCH_GENERIC && g_lastPatchTypes[1] == QGD_PATCH_EMPTY &&
                  g_lastPatchTypes[2] != QGD_PATCH_HALO,
              "patch kinds: generic, empty, and a face-less processor patch is NOT announced as a halo patch");
        dimensioned<scalar> zs("0", dimensionSet(), 0.0);
        dimensioned<vector> zv("0", dimensionSet(), pTraits<vector>::zero);
        dimensioned<tensor> zt("0", dimensionSet(), pTraits<tensor>::zero);
        volScalarField p(IOobject("p", "0", mesh), mesh, zs);
        volVectorField U(IOobject("U", "0", mesh), mesh, zv);
        volTensorField T(IOobject("T", "0", mesh), mesh, zt);
        const int ops0 = g_opCalls;
        tmp<surfaceVectorField> g1 = st->Grad(p);
        tmp<surfaceTensorField> g2 = st->Grad(U);
        tmp<surfaceScalarField> d1 = st->Div(U);
        tmp<surfaceVectorField> d2 = st->Div(T);
        check(g_opCalls - ops0 == 4, "the four virtuals each reach their C-ABI entry once");
        const int expect = (w == 2) ? 1 : 0;
        check(p.nCorrectBCs == expect && T.nCorrectBCs == expect && U.nCorrectBCs == 2 * expect,
              (std::string(words[w]) + (expect ? ": correctBoundaryConditions() before every operator (GaussVolPointStencil_8C L73-121)"
                                               : ": inputs' boundary conditions left as they are (reducedFaceNormalStencil_8C L69-108)"))
                  .c_str());
        check(g1().primitiveField().size() == 1 && g1().boundaryField().size() == 3 && g1().boundaryField()[1].size() == 0,
              "result is a surface field over internal faces + patches (empty patch: no slots)");
        delete st;
    }
    // 2. decomposed run
    {
        fvMesh mesh;
        twoCells(mesh, false);
        Pstream::parRun_ = true;
        const int before = g_meshCreates;
        bool thrown = false;
        std::string msg;
        try { fvsc::hipGaussVolPoint st(IOobject("x", "0", mesh)); } catch (const FoamFatal& e) { thrown = true; msg = e.what(); }
        Pstream::parRun_ = false;
        check(thrown && msg.find("parRun") != std::string::npos && g_meshCreates == before, "Pstream::parRun() -> FatalError before anything is uploaded");
    }
    // 3. processor patch with faces in a serial run (reconstructed-by-hand case directories exist)
    {
        fvMesh mesh;
        twoCells(mesh, true);
        const int before = g_meshCreates;
        bool thrown = false;
        std::string msg;
        try { fvsc::hipReduced st(IOobject("x", "0", mesh)); } catch (const FoamFatal& e) { thrown = true; msg = e.what(); }
        check(thrown && msg.find("procBoundary0to1") != std::string::npos && g_meshCreates == before, "processor patch with faces -> FatalError naming the patch");
    }
    std::printf("%s\n", failures ? "HARNESS FAILED" : "HARNESS OK");
    return failures ? 1 : 0;
}
