// qgd_amd_fvsc.hpp -- header-only C++ mirror of the reference's fvsc interface over the C-ABI of qgd_amd.h.
//
// Same names, argument meaning and error behaviour as the reference so call sites read the same:
//   fvsc::grad(vf) / fvsc::div(vf)                 fvsc.H L46-68, fvsc.C L87-167
//   fvscStencil::New / lookupOrNew, TypeName words fvscStencil.H L46-137, fvscStencil.C L59-118
//   scheme word from fvSchemes.fvsc[term] else default, leastSquares refused in 3-D   fvsc.C L47-63
//   qgdInterpolate(psi), qgdFlux(flux, psi, psif[, fluxName])                          QGDInterpolate.H L38-131
// Fields are plain std::vector<double> containers (OpenFOAM itself is not required); the OpenFOAM adapter in
// INTEGRATION.md wraps the same C entries with tmp<surface*Field> results.
#pragma once
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "qgd_amd.h"

namespace qgd_amd {

// FatalError equivalent: the reference exits the process; a library user gets an exception carrying the status
struct FatalError : std::runtime_error {
    int status;
    FatalError(int s, const std::string& where) : std::runtime_error(where + ": " + qgd_last_error()), status(s) {}
};
inline void check(int status, const char* where) {
    if (status != QGD_OK) throw FatalError(status, where);
}

// vol<Type>Field / surface<Type>Field stand-ins: ncomp interleaved components
struct volField {
    std::string name;
    int ncomp;
    std::vector<double> internal;  // nCells*ncomp
    std::vector<double> boundary;  // nBoundaryFaces*ncomp (patch values)
};
struct surfaceField {
    int ncomp;
    std::vector<double> values;  // nFaces*ncomp, internal faces then patches in mesh order
};

// fvMesh stand-in: device-resident mesh + the fvSchemes.fvsc dictionary + the object registry of stencils
class fvscStencil;
struct fvMesh {
    qgd_device_t device = nullptr;
    int64_t nCells = 0, nFaces = 0, nInternalFaces = 0;
    std::map<std::string, std::string> fvscSchemes{{"default", "GaussVolPoint"}};
    // fvSchemes.interpolationSchemes / divSchemes as qgdInterpolate / qgdFlux consult them [QGDInterpolate.H L42-66, L86]
    std::map<std::string, std::string> interpolationSchemes;
    std::map<std::string, std::string> divSchemes;
    std::map<std::string, std::shared_ptr<fvscStencil>> registry;
};

namespace fvsc {

class fvscStencil {
protected:
    fvMesh& mesh_;
    int stencilId_;
    std::string word_;

public:
    fvscStencil(const std::string& word, fvMesh& mesh) : mesh_(mesh), stencilId_(-1), word_(word) {
        check(qgd_stencil_lookup(mesh.device, word.c_str(), &stencilId_), "fvscStencil::New");
    }
    virtual ~fvscStencil() {}

    // fvscStencil::New (fvscStencil.C L59-95): unknown words and leastSquares-in-3-D are fatal
    static std::shared_ptr<fvscStencil> New(const std::string& word, fvMesh& mesh) {
        return std::make_shared<fvscStencil>(word, mesh);
    }
    // fvscStencil::lookupOrNew (fvscStencil.C L98-118): one stencil per word, cached in the mesh registry
    static fvscStencil& lookupOrNew(const std::string& word, fvMesh& mesh);

    virtual surfaceField Grad(const volField& vf) {
        surfaceField r;
        if (vf.ncomp == 1) {
            r.ncomp = 3;
            r.values.resize((size_t)mesh_.nFaces * 3);
            check(qgd_fvsc_grad_s(mesh_.device, stencilId_, vf.internal.data(), vf.boundary.data(), r.values.data()), "Grad(volScalarField)");
        } else if (vf.ncomp == 3) {
            r.ncomp = 9;
            r.values.resize((size_t)mesh_.nFaces * 9);
            check(qgd_fvsc_grad_v(mesh_.device, stencilId_, vf.internal.data(), vf.boundary.data(), r.values.data()), "Grad(volVectorField)");
        } else {
            throw FatalError(QGD_ERR_NOT_IMPLEMENTED, "Grad: not implemented for this field type");  // notImplemented(...)
        }
        return r;
    }
    virtual surfaceField Div(const volField& vf) {
        surfaceField r;
        if (vf.ncomp == 3) {
            r.ncomp = 1;
            r.values.resize((size_t)mesh_.nFaces);
            check(qgd_fvsc_div_v(mesh_.device, stencilId_, vf.internal.data(), vf.boundary.data(), r.values.data()), "Div(volVectorField)");
        } else if (vf.ncomp == 9) {
            r.ncomp = 3;
            r.values.resize((size_t)mesh_.nFaces * 3);
            check(qgd_fvsc_div_t(mesh_.device, stencilId_, vf.internal.data(), vf.boundary.data(), r.values.data()), "Div(volTensorField)");
        } else {
            throw FatalError(QGD_ERR_NOT_IMPLEMENTED, "Div: not implemented for this field type");
        }
        return r;
    }
};

// fvscOpName (fvsc.C L47-58)
inline std::string fvscOpName(const fvMesh& mesh, const std::string& termName) {
    auto it = mesh.fvscSchemes.find(termName);
    if (it != mesh.fvscSchemes.end()) return it->second;
    return mesh.fvscSchemes.at("default");
}
inline surfaceField grad(fvMesh& mesh, const volField& vf) {
    return fvscStencil::lookupOrNew(fvscOpName(mesh, "grad(" + vf.name + ")"), mesh).Grad(vf);
}
inline surfaceField div(fvMesh& mesh, const volField& vf) {
    return fvscStencil::lookupOrNew(fvscOpName(mesh, "div(" + vf.name + ")"), mesh).Div(vf);
}

}  // namespace fvsc

// qgdInterpolate [QGDInterpolate.H L38-67]: linearInterpolate(psi) unless interpolationSchemes names a scheme for
// "interpolate(<psi>)" or a default other than `none`; those branches hand the field to fvc::interpolate.  With the word
// `linear` (what nearly every fvSchemes file carries as its default) fvc::interpolate IS linearInterpolate -- the same weights,
// the same numbers -- and takes the library path; any other scheme lives in OpenFOAM's scheme library, which is not behind
// this boundary: fatal, like a missing run-time table entry.
inline surfaceField qgdInterpolate(fvMesh& mesh, const volField& psi) {
    const auto& is = mesh.interpolationSchemes;
    const auto own = is.find("interpolate(" + psi.name + ")");
    const auto def = is.find("default");
    if (own != is.end()) {
        if (own->second != "linear")
            throw FatalError(QGD_ERR_NOT_IMPLEMENTED, "qgdInterpolate(" + psi.name + "): interpolationSchemes{interpolate(" + psi.name + ") " + own->second +
                                                          ";} -- fvc::interpolate with a scheme other than linear stays in OpenFOAM");
    } else if (def != is.end() && def->second != "none" && def->second != "linear") {
        throw FatalError(QGD_ERR_NOT_IMPLEMENTED, "qgdInterpolate(" + psi.name + "): interpolationSchemes{default " + def->second +
                                                      ";} -- fvc::interpolate with a scheme other than linear stays in OpenFOAM");
    }
    surfaceField r;
    r.ncomp = psi.ncomp;
    r.values.resize((size_t)mesh.nFaces * psi.ncomp);
    check(qgd_interpolate(mesh.device, psi.ncomp, psi.internal.data(), psi.boundary.data(), r.values.data()), "qgdInterpolate");
    return r;
}
// qgdFlux [QGDInterpolate.H L76-118]: flux*psif unless divSchemes holds an entry for the flux name; then fvc::flux(flux, psi, name):
// `Gauss linear` = flux * linear(psi), the same numbers; `Gauss upwind` = flux * the upwind cell's psi (qgd_flux_upwind); limited
// schemes stay in OpenFOAM (fatal).  divSchemes.default is not consulted (`found(fluxName)`, L86).
inline surfaceField qgdFlux(fvMesh& mesh, const surfaceField& flux, const volField& psi, const surfaceField& psif,
                            const std::string& fluxName) {
    surfaceField r;
    r.ncomp = psif.ncomp;
    r.values.resize(psif.values.size());
    const auto it = mesh.divSchemes.find(fluxName);
    if (it != mesh.divSchemes.end() && it->second == "Gauss upwind") {
        check(qgd_flux_upwind(mesh.device, psi.ncomp, flux.values.data(), psi.internal.data(), psi.boundary.data(), r.values.data()), "qgdFlux");
        return r;
    }
    if (it != mesh.divSchemes.end() && it->second != "Gauss linear")
        throw FatalError(QGD_ERR_NOT_IMPLEMENTED, "qgdFlux(" + fluxName + "): divSchemes{" + fluxName + " " + it->second +
                                                      ";} -- fvc::flux with a scheme other than Gauss linear / Gauss upwind stays in OpenFOAM");
    check(qgd_flux(mesh.device, psif.ncomp, flux.values.data(), psif.values.data(), r.values.data()), "qgdFlux");
    return r;
}
inline surfaceField qgdFlux(fvMesh& mesh, const surfaceField& flux, const std::string& fluxFieldName, const volField& psi,
                            const surfaceField& psif) {
    return qgdFlux(mesh, flux, psi, psif, "div(" + fluxFieldName + "," + psi.name + ")");  // [L116]
}

class fvscStencil : public fvsc::fvscStencil {
    using fvsc::fvscStencil::fvscStencil;
};

inline fvsc::fvscStencil& fvsc::fvscStencil::lookupOrNew(const std::string& word, fvMesh& mesh) {
    auto it = mesh.registry.find(word);
    if (it == mesh.registry.end()) {
        auto s = std::make_shared<qgd_amd::fvscStencil>(word, mesh);
        it = mesh.registry.emplace(word, s).first;
    }
    return *it->second;
}

}  // namespace qgd_amd
