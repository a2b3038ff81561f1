// C++ host-side mirror over the C-ABI: reads like a reference call site (fvsc::grad(p), fvsc::div(U)).
// Linear field on a box: the GaussVolPoint gradient must be exact on interior faces.
#include <cmath>
#include <cstdio>
#include <vector>

#include "qgd_amd_fvsc.hpp"

using namespace qgd_amd;

int main() {
    if (qgd_device_count() < 1) { std::printf("no device\n"); return 77; }
    const double lo[3] = {0, 0, 0}, hi[3] = {1, 1, 1};
    qgd_mesh_t m = nullptr;
    check(qgd_mesh_box(8, 8, 8, 0, 8, lo, hi, nullptr, &m), "qgd_mesh_box");
    int64_t sz[7];
    check(qgd_mesh_sizes(m, sz), "sizes");
    fvMesh mesh;
    mesh.nFaces = sz[1]; mesh.nInternalFaces = sz[2]; mesh.nCells = sz[3];
    check(qgd_device_create(m, 0, &mesh.device), "qgd_device_create");
    std::vector<double> C(3 * sz[3]), Cf(3 * sz[1]);
    check(qgd_mesh_get(m, "C", C.data(), (int64_t)C.size() * 8), "C");
    check(qgd_mesh_get(m, "Cf", Cf.data(), (int64_t)Cf.size() * 8), "Cf");
    const double g[3] = {1.3, -0.7, 0.4};
    volField p{"p", 1, {}, {}};
    for (int64_t c = 0; c < sz[3]; ++c) p.internal.push_back(2.0 + g[0] * C[3 * c] + g[1] * C[3 * c + 1] + g[2] * C[3 * c + 2]);
    for (int64_t f = sz[2]; f < sz[1]; ++f) p.boundary.push_back(2.0 + g[0] * Cf[3 * f] + g[1] * Cf[3 * f + 1] + g[2] * Cf[3 * f + 2]);
    surfaceField gp = fvsc::grad(mesh, p);
    // faces whose centre is at least 2 cells from the boundary
    double worst = 0;
    int counted = 0;
    for (int64_t f = 0; f < sz[2]; ++f) {
        bool inner = true;
        for (int k = 0; k < 3; ++k) inner = inner && Cf[3 * f + k] > 0.26 && Cf[3 * f + k] < 0.74;
        if (!inner) continue;
        ++counted;
        for (int k = 0; k < 3; ++k) worst = std::fmax(worst, std::fabs(gp.values[3 * f + k] - g[k]));
    }
    std::printf("interior faces %d, max |grad - g| = %.3e\n", counted, worst);
    if (!(counted > 50) || !(worst < 1e-12)) return 1;
    // scheme checks behave like fvscOpName: leastSquares is fatal on a 3-D mesh, unknown words are fatal
    mesh.fvscSchemes["grad(T)"] = "leastSquares";
    volField T = p; T.name = "T";
    try { fvsc::grad(mesh, T); return 2; } catch (const FatalError& e) { if (e.status != QGD_ERR_SCHEME) return 3; }
    mesh.fvscSchemes["grad(T)"] = "noSuchStencil";
    try { fvsc::grad(mesh, T); return 4; } catch (const FatalError& e) { if (e.status != QGD_ERR_UNKNOWN_NAME) return 5; }
    if (mesh.registry.size() != 1) return 6;  // lookupOrNew cached exactly the GaussVolPoint stencil
    // qgdInterpolate / qgdFlux read like the reference's call sites [QGDFoam/updateFields.H L45, updateFluxes.H L78]
    surfaceField pf = qgdInterpolate(mesh, p);
    std::vector<double> w(sz[1]);
    std::vector<int32_t> own(sz[1]), nei(sz[2]);
    check(qgd_mesh_get(m, "weights", w.data(), (int64_t)w.size() * 8), "weights");
    check(qgd_mesh_get(m, "owner", own.data(), (int64_t)own.size() * 4), "owner");
    check(qgd_mesh_get(m, "neighbour", nei.data(), (int64_t)nei.size() * 4), "neighbour");
    for (int64_t f = 0; f < sz[2]; ++f) {
        const double ref = w[f] * (p.internal[own[f]] - p.internal[nei[f]]) + p.internal[nei[f]];
        if (std::fabs(pf.values[f] - ref) > 1e-15 * std::fabs(ref)) return 7;
    }
    for (int64_t f = sz[2]; f < sz[1]; ++f) if (pf.values[f] != p.boundary[f - sz[2]]) return 8;
    surfaceField phi{1, std::vector<double>(sz[1], 2.0)};
    surfaceField phiP = qgdFlux(mesh, phi, "phiJm", p, pf);
    for (int64_t f = 0; f < sz[1]; ++f) if (phiP.values[f] != 2.0 * pf.values[f]) return 9;
    mesh.interpolationSchemes["default"] = "none";          // `none` still means linearInterpolate [L56-63]
    if (qgdInterpolate(mesh, p).values[0] != pf.values[0]) return 10;
    // `default linear` (what nearly every fvSchemes file carries) and `interpolate(p) linear` go to fvc::interpolate in the reference
    // [L42-65] = the same weights, the same numbers: served by the library, bit for bit
    mesh.interpolationSchemes["default"] = "linear";
    { surfaceField r = qgdInterpolate(mesh, p); for (int64_t f = 0; f < sz[1]; ++f) if (r.values[f] != pf.values[f]) return 15; }
    mesh.interpolationSchemes["interpolate(p)"] = "linear";
    { surfaceField r = qgdInterpolate(mesh, p); for (int64_t f = 0; f < sz[1]; ++f) if (r.values[f] != pf.values[f]) return 16; }
    mesh.interpolationSchemes["interpolate(p)"] = "vanLeer";  // any other scheme is OpenFOAM's business: fatal here
    try { qgdInterpolate(mesh, p); return 11; } catch (const FatalError& e) { if (e.status != QGD_ERR_NOT_IMPLEMENTED) return 12; }
    mesh.interpolationSchemes.erase("interpolate(p)");
    mesh.interpolationSchemes["default"] = "cubic";
    try { qgdInterpolate(mesh, p); return 17; } catch (const FatalError& e) { if (e.status != QGD_ERR_NOT_IMPLEMENTED) return 18; }
    mesh.interpolationSchemes["default"] = "linear";
    // qgdFlux with a divSchemes entry of the flux's own name [L86-104]: Gauss linear = flux * linear(psi), Gauss upwind = flux * the upwind cell
    mesh.divSchemes["div(phiJm,p)"] = "Gauss linear";
    { surfaceField r = qgdFlux(mesh, phi, "phiJm", p, pf); for (int64_t f = 0; f < sz[1]; ++f) if (r.values[f] != 2.0 * pf.values[f]) return 13; }
    mesh.divSchemes["div(phiJm,p)"] = "Gauss upwind";
    for (int64_t f = 0; f < sz[1]; ++f) phi.values[f] = (f % 3 == 0) ? -1.5 : 2.0;   // both flux directions
    {
        surfaceField r = qgdFlux(mesh, phi, "phiJm", p, pf);
        for (int64_t f = 0; f < sz[2]; ++f) {
            const double lambda = phi.values[f] >= 0.0 ? 1.0 : 0.0;
            const double up = lambda * (p.internal[own[f]] - p.internal[nei[f]]) + p.internal[nei[f]];
            if (r.values[f] != phi.values[f] * up) return 14;
        }
        for (int64_t f = sz[2]; f < sz[1]; ++f) if (r.values[f] != phi.values[f] * p.boundary[f - sz[2]]) return 19;
    }
    mesh.divSchemes["div(phiJm,p)"] = "Gauss limitedLinear 1";
    try { qgdFlux(mesh, phi, "phiJm", p, pf); return 20; } catch (const FatalError& e) { if (e.status != QGD_ERR_NOT_IMPLEMENTED) return 21; }
    mesh.registry.clear();
    qgd_device_free(mesh.device);
    qgd_mesh_free(m);
    std::printf("ok\n");
    return 0;
}
