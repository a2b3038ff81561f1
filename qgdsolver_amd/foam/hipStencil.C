/*---------------------------------------------------------------------------*\
    hipStencil.C -- see hipStencil.H.  NOT COMPILED HERE (no OpenFOAM in the image).
\*---------------------------------------------------------------------------*/
#include "hipStencil.H"
#include "addToRunTimeSelectionTable.H"
#include "emptyFvPatch.H"
#include "wedgeFvPatch.H"
#include "symmetryFvPatch.H"
#include "symmetryPlaneFvPatch.H"
#include "processorFvPatch.H"
#include "coupledFvPatch.H"

namespace Foam
{
namespace fvsc
{
    defineTypeNameAndDebug(hipStencil, 0);
    defineTypeNameAndDebug(hipReduced, 0);
    defineTypeNameAndDebug(hipLeastSquares, 0);
    defineTypeNameAndDebug(hipGaussVolPoint, 0);
    addToRunTimeSelectionTable(fvscStencil, hipReduced, components);
    addToRunTimeSelectionTable(fvscStencil, hipLeastSquares, components);
    addToRunTimeSelectionTable(fvscStencil, hipGaussVolPoint, components);
}
}

static void qgdCheck(int status, const char* where)
{
    if (status != QGD_OK)
    {
        // every failure of the library becomes the reference's own failure mode [fvsc_8C L62, L75-78]
        FatalErrorIn(where) << qgd_last_error() << Foam::nl << Foam::exit(Foam::FatalError);
    }
}

Foam::fvsc::hipStencil::hipStencil(const IOobject& io, const word& targetWord)
:
    fvscStencil(io), hmesh_(nullptr), hdev_(nullptr), stencilId_(-1), correctBCs_(targetWord == "GaussVolPoint")
{
    uploadMesh(targetWord);
}

Foam::fvsc::hipStencil::~hipStencil()
{
    qgd_device_free(hdev_);
    qgd_mesh_free(hmesh_);
}

void Foam::fvsc::hipStencil::uploadMesh(const word& targetWord)
{
    const fvMesh& mesh = mesh_;
    // The reference's stencils serve processor patches (patchNeighbourField: GaussVolPointBase3D_8C L588, L688, L785, L882;
    // extendedFaceStencilScalarGrad_8C L116-268).  This adapter does not yet: QGD_PATCH_HALO means "ghost cells behind the
    // patch", and a decomposed OpenFOAM mesh has none.  Refuse instead of answering wrongly under mpirun.
    if (Pstream::parRun())
    {
        FatalErrorIn("hipStencil::uploadMesh")
            << "hip* fvsc stencils do not serve decomposed cases yet (Pstream::parRun()): run undecomposed, or shard "
            << "inside the library (qgd_mesh_shard / qgd_case_step_sharded, INTEGRATION.md)" << nl << exit(FatalError);
    }
    forAll(mesh.boundary(), patchi)
    {
        if (isA<processorFvPatch>(mesh.boundary()[patchi]) && mesh.boundary()[patchi].size() > 0)
        {
            FatalErrorIn("hipStencil::uploadMesh")
                << "patch " << mesh.boundary()[patchi].name() << " is a processor patch with faces; hip* fvsc stencils "
                << "do not serve processor patches yet" << nl << exit(FatalError);
        }
    }
    const faceList& faces = mesh.faces();
    labelList offsets(faces.size() + 1, 0);
    forAll(faces, f) { offsets[f + 1] = offsets[f] + faces[f].size(); }
    labelList facePoints(offsets[faces.size()]);
    forAll(faces, f) { forAll(faces[f], k) { facePoints[offsets[f] + k] = faces[f][k]; } }

    const polyBoundaryMesh& pbm = mesh.boundaryMesh();
    labelList pStart(pbm.size()), pSize(pbm.size()), pType(pbm.size());
    forAll(pbm, i)
    {
        const fvPatch& fvp = mesh.boundary()[i];
        pStart[i] = pbm[i].start();
        pSize[i]  = pbm[i].size();
        pType[i]  =
            isA<emptyFvPatch>(fvp)         ? QGD_PATCH_EMPTY
          : isA<symmetryPlaneFvPatch>(fvp) ? QGD_PATCH_SYMMETRYPLANE
          : isA<symmetryFvPatch>(fvp)      ? QGD_PATCH_SYMMETRY
          : isA<wedgeFvPatch>(fvp)         ? QGD_PATCH_WEDGE
          : isA<coupledFvPatch>(fvp)       ? QGD_PATCH_CYCLIC
          :                                  QGD_PATCH_GENERIC;
    }
    // label is int32 and scalar is double in the default build (WM_LABEL_SIZE=32, WM_PRECISION_OPTION=DP)
    qgdCheck
    (
        qgd_mesh_create
        (
            mesh.nPoints(), reinterpret_cast<const double*>(mesh.points().cdata()),
            mesh.nFaces(), offsets.cdata(), facePoints.cdata(),
            mesh.nInternalFaces(), mesh.faceOwner().cdata(), mesh.faceNeighbour().cdata(),
            mesh.nCells(), pbm.size(), pStart.cdata(), pSize.cdata(), pType.cdata(), &hmesh_
        ),
        "hipStencil::uploadMesh"
    );
    // OpenFOAM's own geometry, so that nothing depends on our restatement of its decomposition rules
    qgdCheck
    (
        qgd_mesh_set_geometry
        (
            hmesh_,
            reinterpret_cast<const double*>(mesh.faceAreas().cdata()),
            reinterpret_cast<const double*>(mesh.faceCentres().cdata()),
            reinterpret_cast<const double*>(mesh.cellCentres().cdata()),
            mesh.cellVolumes().cdata()
        ),
        "hipStencil::uploadMesh"
    );
    qgdCheck(qgd_device_create(hmesh_, 0, &hdev_), "hipStencil::uploadMesh");
    qgdCheck(qgd_stencil_lookup(hdev_, targetWord.c_str(), &stencilId_), "hipStencil::uploadMesh");
}

template<class Type>
void Foam::fvsc::hipStencil::flattenBoundary
(
    const GeometricField<Type, fvPatchField, volMesh>& vf, List<scalar>& out
) const
{
    const label nc = pTraits<Type>::nComponents;
    out.setSize((mesh_.nFaces() - mesh_.nInternalFaces())*nc);
    out = 0.0;
    forAll(vf.boundaryField(), patchi)
    {
        const fvPatchField<Type>& pf = vf.boundaryField()[patchi];   // size 0 on empty patches
        const label start = mesh_.boundaryMesh()[patchi].start() - mesh_.nInternalFaces();
        forAll(pf, i)
        {
            for (direction d = 0; d < nc; ++d) { out[(start + i)*nc + d] = component(pf[i], d); }
        }
    }
}

template<class Type>
Foam::tmp<Foam::GeometricField<Type, Foam::fvsPatchField, Foam::surfaceMesh>>
Foam::fvsc::hipStencil::wrap(const word& name, const dimensionSet& dims, const List<scalar>& flat) const
{
    typedef GeometricField<Type, fvsPatchField, surfaceMesh> SurfType;
    tmp<SurfType> tres
    (
        new SurfType
        (
            IOobject(name, mesh_.time().timeName(), mesh_, IOobject::NO_READ, IOobject::NO_WRITE),
            mesh_, dimensioned<Type>("0", dims, pTraits<Type>::zero)
        )
    );
    SurfType& res = tres.ref();
    const label nc = pTraits<Type>::nComponents;
    forAll(res.primitiveField(), f)
    {
        for (direction d = 0; d < nc; ++d) { setComponent(res.primitiveFieldRef()[f], d) = flat[f*nc + d]; }
    }
    forAll(res.boundaryField(), patchi)
    {
        fvsPatchField<Type>& pf = res.boundaryFieldRef()[patchi];
        const label start = mesh_.boundaryMesh()[patchi].start();
        forAll(pf, i)
        {
            for (direction d = 0; d < nc; ++d) { setComponent(pf[i], d) = flat[(start + i)*nc + d]; }
        }
    }
    return tres;
}

Foam::tmp<Foam::surfaceVectorField> Foam::fvsc::hipStencil::Grad(const volScalarField& vF)
{
    // GaussVolPoint re-evaluates the BCs of its input first [GaussVolPointStencil_8C L73]; keep that side effect for
    // that word only (reduced and leastSquares never do: reducedFaceNormalStencil_8C L69-108)
    refreshPatches(vF);
    List<scalar> bnd, out(mesh_.nFaces()*3);
    flattenBoundary(vF, bnd);
    qgdCheck(qgd_fvsc_grad_s(hdev_, stencilId_, vF.primitiveField().cdata(), bnd.cdata(), out.data()), "hipStencil::Grad");
    return wrap<vector>("grad(" + vF.name() + ")", vF.dimensions()/dimLength, out);
}

Foam::tmp<Foam::surfaceTensorField> Foam::fvsc::hipStencil::Grad(const volVectorField& iVF)
{
    refreshPatches(iVF);
    List<scalar> bnd, out(mesh_.nFaces()*9);
    flattenBoundary(iVF, bnd);
    qgdCheck
    (
        qgd_fvsc_grad_v(hdev_, stencilId_, reinterpret_cast<const double*>(iVF.primitiveField().cdata()), bnd.cdata(), out.data()),
        "hipStencil::Grad"
    );
    return wrap<tensor>("grad(" + iVF.name() + ")", iVF.dimensions()/dimLength, out);
}

Foam::tmp<Foam::surfaceScalarField> Foam::fvsc::hipStencil::Div(const volVectorField& iVF)
{
    refreshPatches(iVF);
    List<scalar> bnd, out(mesh_.nFaces());
    flattenBoundary(iVF, bnd);
    qgdCheck
    (
        qgd_fvsc_div_v(hdev_, stencilId_, reinterpret_cast<const double*>(iVF.primitiveField().cdata()), bnd.cdata(), out.data()),
        "hipStencil::Div"
    );
    return wrap<scalar>("div(" + iVF.name() + ")", iVF.dimensions()/dimLength, out);
}

Foam::tmp<Foam::surfaceVectorField> Foam::fvsc::hipStencil::Div(const volTensorField& iTF)
{
    refreshPatches(iTF);
    List<scalar> bnd, out(mesh_.nFaces()*3);
    flattenBoundary(iTF, bnd);
    qgdCheck
    (
        qgd_fvsc_div_t(hdev_, stencilId_, reinterpret_cast<const double*>(iTF.primitiveField().cdata()), bnd.cdata(), out.data()),
        "hipStencil::Div"
    );
    return wrap<vector>("div(" + iTF.name() + ")", iTF.dimensions()/dimLength, out);
}
