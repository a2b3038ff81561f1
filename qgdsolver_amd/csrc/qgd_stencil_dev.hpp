// qgd_stencil_dev.hpp -- device functions shared by the kernel files (qgd_kernels.hip: QGDFoam path, qgd_qhd.hip: QHDFoam
// path): streamed loads, the fvsc face gradient of every stencil (reduced, leastSquares, GaussVolPoint 2-D / 3-D with its
// coefficients rebuilt from geometry) and the XCD-aware workgroup -> tile map.  Reference listings as cited per function.
#pragma once
#include "qgd_device.hpp"

namespace qgd {

#ifndef QGD_BLOCK
#define QGD_BLOCK 256
#endif

__device__ __forceinline__ double lerpf(double w, double a, double b) { return w * (a - b) + b; }

// The five fp64 divisions of a face -- 1/(6V) of the Gauss coefficients, p/rho/rho of the heat flux, muQGD/PrQGD of the two cells' alphaEff --
// as v_rcp_f64 + two Newton steps (<= 1-2 ulp) or a product with the precomputed 1/PrQGD instead of the IEEE division sequence
// (v_div_scale x2, v_rcp, 6 fma, v_div_fmas, v_div_fixup): -43 of 726 fp64 instructions of the staged face kernel, F 7.31 -> 7.12 ms at
// 64 M cells on one box (profiles/r04_ab_face_instruction_diet.txt).  A deliberate deviation of a few ulp from the listing's
// divisions.  ONE flavour everywhere a face kernel divides (QGD_RCP): the case's kernels (staged, gather, 2-D, boundary)
// -- which therefore stay bit-identical to each other -- and the stateless fvsc operators and field accessors.  (Operators and case
// still differ in the last bits on quadrilaterals for another reason: the operators form the listing's coefficients a_k/6 and 1/V, the
// case the difference form with 1/(6V); the parity tests hold both to the CPU restatement, <= 1e-11.)  -DQGD_F_DIET=0 builds the divisions back.
#ifndef QGD_F_DIET
#define QGD_F_DIET 1
#endif
__device__ __forceinline__ double rcpNewton(const double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    return r;
}
#if QGD_F_DIET
#define QGD_RCP(x) rcpNewton(x)
#else
#define QGD_RCP(x) (1.0 / (x))
#endif

// the normal a slip / symmetry / symmetryPlane patch reflects about on boundary face f: the face's own unit normal
// (basicSymmetryFvPatchField: patch().nf()), or the patch's one normal on a symmetryPlane (PatchBCDev::nHat)
__device__ __forceinline__ void symmNormal(const MeshView& m, const PatchBCDev& bc, const int f, double n[3]) {
    if (bc.planeN) { n[0] = bc.nHat[0]; n[1] = bc.nHat[1]; n[2] = bc.nHat[2]; return; }
    const double ms = m.magSf[f];
    n[0] = m.Sx[f] / ms; n[1] = m.Sy[f] / ms; n[2] = m.Sz[f] / ms;
}

// geometry records are packed triples (24 B): the face kernels pay for every byte their gathers pull in, padding included
__device__ __forceinline__ double4 ld3(const double* __restrict__ base, const int i) {
    const double* p = base + 3 * (size_t)i;
    double4 r;
    r.x = p[0]; r.y = p[1]; r.z = p[2]; r.w = 0.0;
    return r;
}

// Streamed-once data (per-face geometry, gather lists) is loaded non-temporally so it does not push the
// re-used cell/vertex records out of the 4 MiB L2 of the XCD.
#ifndef QGD_NT
#define QGD_NT 1
#endif
template <class T>
__device__ __forceinline__ T ldStream(const T* p) {
#if QGD_NT
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}
template <class T>
__device__ __forceinline__ void stStream(T* p, T v) {
#if QGD_NT
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}

template <int NC>
struct FaceVals {
    double o[NC];   // owner cell values
    double n[NC];   // neighbour cell (internal face) or patch value (boundary face)
    double sn[NC];  // boundary face: patch snGrad
};

// ---------------------------------------------------------------------------
// GaussVolPoint 3-D coefficients of one face from its geometry [GaussVolPointBase3D_8C L161-476].
// O/N: owner / neighbour cell centre (boundary: mirror point), x1..x4: face vertices in face order.
// quad: a[3d+0]=a0, a[3d+1]=a1, a[3d+2]=a5 (a2=-a0, a3=-a1, a4=-a5) [L353-389];  rV = 1/V [L346-350]
// tri : t[4d+0..2]=a0..a2 (vertices), t[4d+3]=a3 (neighbour), owner = -a3 [L193-229]; rV = 1/V [L186-190]
// ---------------------------------------------------------------------------
__device__ __forceinline__ void gvpQuadCoef(const double4 O, const double4 N, const double4 x1, const double4 x2,
                                            const double4 x3, const double4 x4, double a[9], double& rV) {
    const double sixth = (1.0 / 6.0);
    const double o[3] = {O.x, O.y, O.z}, n[3] = {N.x, N.y, N.z};
    const double p1[3] = {x1.x, x1.y, x1.z}, p2[3] = {x2.x, x2.y, x2.z}, p3[3] = {x3.x, x3.y, x3.z}, p4[3] = {x4.x, x4.y, x4.z};
    double d31[3], d42[3], on[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { d31[k] = p3[k] - p1[k]; d42[k] = p4[k] - p2[k]; on[k] = o[k] - n[k]; }
    const double cr[3] = {d42[1] * on[2] - d42[2] * on[1], d42[2] * on[0] - d42[0] * on[2], d42[0] * on[1] - d42[1] * on[0]};
    double vol = d31[0] * cr[0] + d31[1] * cr[1] + d31[2] * cr[2];
    vol *= sixth;
    rV = QGD_RCP(vol);
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const int u = (d + 1) % 3, v = (d + 2) % 3;
        a[3 * d + 0] = sixth * ((n[u] - o[u]) * (p2[v] - p4[v]) - (n[v] - o[v]) * (p2[u] - p4[u]));
        a[3 * d + 1] = sixth * ((n[u] - o[u]) * (p3[v] - p1[v]) - (n[v] - o[v]) * (p3[u] - p1[u]));
        a[3 * d + 2] = sixth * ((p1[u] - p3[u]) * (p2[v] - p4[v]) - (p1[v] - p3[v]) * (p2[u] - p4[u]));
    }
}
__device__ __forceinline__ void gvpTriCoef(const double4 O, const double4 N, const double4 x1, const double4 x2,
                                           const double4 x3, double t[12], double& rV) {
    const double sixth = (1.0 / 6.0);
    const double o[3] = {O.x, O.y, O.z}, n[3] = {N.x, N.y, N.z};
    const double p1[3] = {x1.x, x1.y, x1.z}, p2[3] = {x2.x, x2.y, x2.z}, p3[3] = {x3.x, x3.y, x3.z};
    double e21[3], e31[3], on[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { e21[k] = p2[k] - p1[k]; e31[k] = p3[k] - p1[k]; on[k] = o[k] - n[k]; }
    const double cr[3] = {e21[1] * e31[2] - e21[2] * e31[1], e21[2] * e31[0] - e21[0] * e31[2], e21[0] * e31[1] - e21[1] * e31[0]};
    double vol = cr[0] * on[0] + cr[1] * on[1] + cr[2] * on[2];
    vol *= sixth;
    rV = QGD_RCP(vol);
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const int u = (d + 1) % 3, v = (d + 2) % 3;
        t[4 * d + 0] = sixth * ((o[v] - n[v]) * (p2[u] - p3[u]) + (n[u] - o[u]) * (p2[v] - p3[v]));
        t[4 * d + 1] = sixth * ((n[u] - o[u]) * (p3[v] - p1[v]) + (o[v] - n[v]) * (p3[u] - p1[u]));
        t[4 * d + 2] = sixth * ((n[u] - o[u]) * (p1[v] - p2[v]) + (o[v] - n[v]) * (p1[u] - p2[u]));
        t[4 * d + 3] = sixth * (p1[v] * (p2[u] - p3[u]) + p2[v] * (p3[u] - p1[u]) + p3[v] * (p1[u] - p2[u]));
    }
}

// The quadrilateral / triangle forms of the GaussVolPoint face gradient from values already in registers (the generic walk below
// and the LDS-staged QHD kernels share it): o / psiN the owner and neighbour (boundary: mirror-point) values, q0..q3 the vertex
// values, geometry as in gvpQuadCoef / gvpTriCoef.  kind 0: quadrilateral, otherwise triangle (x3 / q3 unused).
template <int NC, int UOFF>
__device__ __forceinline__ void gvp3GradCore(const int kind, const bool internal, const double4 cO, const double4 cN, const double4 x0,
                                             const double4 x1, const double4 x2, const double4 x3, const double* __restrict__ o,
                                             const double* __restrict__ psiN, const double* __restrict__ q0, const double* __restrict__ q1,
                                             const double* __restrict__ q2, const double* __restrict__ q3, double* __restrict__ g) {
    double rV;
    if (kind == 0) {  // quad: a2=-a0, a3=-a1, a4(nei)=-a5(own) [3D.C L361-363]
        double a[9];
        gvpQuadCoef(cO, cN, x0, x1, x2, x3, a, rV);
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const double a0 = a[3 * d], a1 = a[3 * d + 1], a5 = a[3 * d + 2];
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                double s = psiN[k] * (-a5);
                s += o[k] * a5;
                s += q0[k] * a0;
                s += q1[k] * a1;
                s += q2[k] * (-a0);
                s += q3[k] * (-a1);
                g[d * NC + k] = s * rV;
            }
        }
    } else {  // triangle: slots a0,a1,a2 vertices, a3 neighbour, owner = -a3 [3D.C L193-229]
        double t[12];
        gvpTriCoef(cO, cN, x0, x1, x2, t, rV);
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const double a0 = t[4 * d], a1 = t[4 * d + 1], a2 = t[4 * d + 2], a3 = t[4 * d + 3];
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                double s = psiN[k] * a3;
                s += o[k] * (-a3);
                s += q0[k] * a0;
                s += q1[k] * a1;
                s += q2[k] * a2;
                g[d * NC + k] = s * rV;
            }
        }
        if (UOFF >= 0 && internal) {
            // interior triangles, vector field: every row i holds d_j U_j [3D.C L844-854]
            double dg[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) dg[j] = g[j * NC + UOFF + j];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) g[i * NC + UOFF + j] = dg[j];
        }
    }
}

// ---- staging the distinct records of a face tile through LDS (qgd_setup.hpp FaceTiles; QHD face passes, implicit face kernel) ----------
typedef double v2dTile __attribute__((ext_vector_type(2)));
// K rounds of pieces of type PT: lane q of round k takes piece tid + k * FB of the tile's list -- PPR consecutive pieces per record, taken
// from a record of STRIDE pieces starting at piece OFFSET (the whole record by default)
template <typename PT, int K, int PPR, int FB, int STRIDE = PPR, int OFFSET = 0>
struct TileStager {
    PT d[K];
    __device__ __forceinline__ void load(const PT* __restrict__ g, const int32_t* __restrict__ list, const int nU, const int tid) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int q = tid + k * FB, r = q / PPR;
            d[k] = g[(size_t)list[min(r, nU - 1)] * STRIDE + OFFSET + (q - r * PPR)];
        }
    }
    __device__ __forceinline__ void store(PT* __restrict__ s, const int nU, const int tid) const {
#pragma unroll
        for (int k = 0; k < K; ++k) { const int q = tid + k * FB; if (q < nU * PPR) s[q] = d[k]; }
    }
};

// ---------------------------------------------------------------------------
// fvsc face gradient of an NC-component field: g[i*NC + k] = d_i phi_k.
// cellF / ptF are AoS with stride NC (cell and vertex values).
// UOFF >= 0 marks three consecutive components as a vector so that the
// interior-triangle pattern of the reference's vector gradient
// [GaussVolPointBase3D_8C L844-854] is reproduced.
// ---------------------------------------------------------------------------
template <int ST, int NC, int UOFF>
__device__ __forceinline__ void faceGradient(const MeshView& m, const int f, const FaceVals<NC>& v,
                                             const double* __restrict__ cellF, const double* __restrict__ ptF,
                                             double* __restrict__ g) {
    const bool internal = f < m.nIF;
    const int b = f - m.nIF;
    const int kind = m.fkind[f];
#pragma unroll
    for (int i = 0; i < 3 * NC; ++i) g[i] = 0.0;
    if (kind == 3) return;  // FK_SKIP: empty patches carry no field

    auto reducedForm = [&]() {
        const double ms = m.magSf[f];
        const double nx = m.Sx[f] / ms, ny = m.Sy[f] / ms, nz = m.Sz[f] / ms;
        const double dn = m.dn[f];
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            const double s = internal ? dn * (v.n[k] - v.o[k]) : v.sn[k];
            g[0 * NC + k] = nx * s;
            g[1 * NC + k] = ny * s;
            g[2 * NC + k] = nz * s;
        }
    };

    if constexpr (ST == ST_REDUCED) {
        reducedForm();
        return;
    }
    if constexpr (ST == ST_GVP3) {
        if (kind == 2) { reducedForm(); return; }  // faces with > 4 vertices [3D.C L759-768]
        const int4 vt = m.verts[f];
        double psiN[NC];
        if (internal) {
#pragma unroll
            for (int k = 0; k < NC; ++k) psiN[k] = v.n[k];
        } else {
            const double hd = m.bmvON[b];
#pragma unroll
            for (int k = 0; k < NC; ++k) psiN[k] = v.n[k] + v.sn[k] * hd * 0.5;  // [3D.C L790-793]
        }
        const double4 cO = ld3(m.Cc, m.own[f]);
        const double4 cN = internal ? ld3(m.Cc, m.nei[f]) : m.bN[b];
        const int v3 = kind == 0 ? vt.w : vt.z;   // (a triangle has no fourth vertex: its slot is not read)
        const double* p0 = ptF + (size_t)vt.x * NC;
        const double* p1 = ptF + (size_t)vt.y * NC;
        const double* p2 = ptF + (size_t)vt.z * NC;
        const double* p3 = ptF + (size_t)v3 * NC;
        double q0[NC], q1[NC], q2[NC], q3[NC];
#pragma unroll
        for (int k = 0; k < NC; ++k) { q0[k] = p0[k]; q1[k] = p1[k]; q2[k] = p2[k]; q3[k] = p3[k]; }
        gvp3GradCore<NC, UOFF>(kind, internal, cO, cN, ld3(m.X, vt.x), ld3(m.X, vt.y), ld3(m.X, vt.z), ld3(m.X, v3), v.o, psiN, q0, q1, q2, q3, g);
        return;
    }
    if constexpr (ST == ST_GVP2) {
        const int2 ip = m.ip13[f];
        const size_t nF = (size_t)m.nF;
        const double c1 = m.c2d[0 * nF + f], c2 = m.c2d[1 * nF + f], c3 = m.c2d[2 * nF + f], c4 = m.c2d[3 * nF + f];
        const double mv42 = m.c2d[4 * nF + f], mv13 = m.c2d[5 * nF + f];
        const double* pa = ptF + (size_t)ip.x * NC;
        const double* pb = ptF + (size_t)ip.y * NC;
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            const double hi = internal ? v.n[k] : (v.n[k] + v.sn[k] * mv42 * 0.5);  // [2D.C L343-346]
            const double dfdn = (hi - v.o[k]) / mv42;
            const double dfdt = (pb[k] - pa[k]) / mv13;
            g[m.ie1 * NC + k] = (dfdn * c1 - dfdt * c2);
            g[m.ie2 * NC + k] = (dfdt * c3 - dfdn * c4);
        }
        return;
    }
    if constexpr (ST == ST_LSQ) {
        if (!internal) {
            if (m.lsqBndZero[b]) return;  // constraint patches stay zero [ScalarGrad.C L90-101]
            const double ms = m.magSf[f];
            const double nx = m.Sx[f] / ms, ny = m.Sy[f] / ms, nz = m.Sz[f] / ms;
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                g[0 * NC + k] = nx * v.sn[k];
                g[1 * NC + k] = ny * v.sn[k];
                g[2 * NC + k] = nz * v.sn[k];
            }
            return;
        }
        if (m.lsqDeg[f]) { reducedForm(); return; }  // [ScalarGrad.C L76-83]
        const double w = m.w[f];
        double pf[NC];
#pragma unroll
        for (int k = 0; k < NC; ++k) pf[k] = lerpf(w, v.o[k], v.n[k]);
        const int cnt = m.lsqCnt[f];
        const size_t base = (size_t)m.lsqSlice[f >> 6] * 64 + (f & 63);
        for (int i = 0; i < cnt; ++i) {
            const size_t e = base + (size_t)i * 64;
            const double* cv = cellF + (size_t)m.lsqCell[e] * NC;
            const double gx = m.lsqGx[e], gy = m.lsqGy[e], gz = m.lsqGz[e];
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                const double dphi = cv[k] - pf[k];
                g[0 * NC + k] = g[0 * NC + k] + gx * dphi;
                g[1 * NC + k] = g[1 * NC + k] + gy * dphi;
                g[2 * NC + k] = g[2 * NC + k] + gz * dphi;
            }
        }
        return;
    }
}

// XCD-aware tile order: consecutive workgroups are dealt round-robin to the 8
// XCDs (block b -> XCD b%8), each with a private L2.  Remap so that every XCD
// walks one contiguous eighth of the face range and neighbouring tiles (which
// share cell and vertex records) meet in the same L2.
__device__ __forceinline__ int xcdTile(int nTiles, int run = 0) {
    const int b = blockIdx.x;
    if (run <= 0) {
        const int per = nTiles >> 3;        // tiles per XCD (the tail past 8*per keeps identity order)
        if (b >= (per << 3)) return b;
        return (b & 7) * per + (b >> 3);
    }
    // runs of `run` consecutive tiles dealt round-robin to the XCDs: consecutive tiles still meet in one L2, and the
    // eight XCDs stay inside one window of 8*run tiles, so what one of them fetched from HBM is found by the others (one
    // k-plane later) in the shared Infinity Cache instead of each XCD keeping a plane-sized working set of its own
    const int span = run << 3;
    const int full = (nTiles / span) * span;
    if (b >= full) return b;
    const int xcd = b & 7, i = b >> 3;
    return ((i / run) * 8 + xcd) * run + (i % run);
}

// transform(T, v) = T & v and transform(T, A) = T & A & T^T (L0: transform.H), T row-major
__device__ __forceinline__ void transformVec(const double* __restrict__ T, const double v[3], double out[3]) {
#pragma unroll
    for (int i = 0; i < 3; ++i) out[i] = T[3 * i] * v[0] + T[3 * i + 1] * v[1] + T[3 * i + 2] * v[2];
}
__device__ __forceinline__ void transformTen(const double* __restrict__ T, const double A[9], double out[9]) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            double acc = 0.0;
#pragma unroll
            for (int l = 0; l < 3; ++l) acc += (T[3 * i] * A[l] + T[3 * i + 1] * A[3 + l] + T[3 * i + 2] * A[6 + l]) * T[3 * j + l];
            out[3 * i + j] = acc;
        }
}

// patch points: weighted mean of the surrounding boundary-face values.  Source
// and destination strides/offset are free so the qgdFlux pass can refresh the
// pressure component alone from the mid-step patch pressures.
// vecMode: what of the NC components is a vector / tensor and therefore subject to the mesh's point constraints
// (MeshView::cpOff; L0: pointConstraints::constrain at the end of volPointInterpolation::interpolateBoundaryField):
// -1 nothing (scalars, or a field the reference interpolates component by component: the 2-D gradient of a vector
// [GaussVolPointBase.C L79-87]), 0..NC-3 the offset of a vector, -2 the whole record is a tensor (NC = 9).
template <int NC>
__global__ __launch_bounds__(QGD_BLOCK) void boundaryPointKernel(const MeshView m, const double* __restrict__ bndF,
                                                                const int bndStride, double* __restrict__ ptF,
                                                                const int ptStride, const int ptOffset, const int vecMode) {
    const int i = blockIdx.x * QGD_BLOCK + threadIdx.x;
    if (i >= m.nBP) return;
    const int p = m.bpPoint[i];
    double acc[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) acc[k] = 0.0;
    for (int e = m.bpOff[i]; e < m.bpOff[i + 1]; ++e) {
        const double w = m.bpW[e];
        const double* bv = bndF + (size_t)m.bpFace[e] * bndStride;
#pragma unroll
        for (int k = 0; k < NC; ++k) acc[k] += w * bv[k];
    }
    if constexpr (NC >= 3) {
        if (vecMode != -1 && m.cpOff) {
            for (int e = m.cpOff[i]; e < m.cpOff[i + 1]; ++e) {
                const double* T = m.cpT + 9 * (size_t)e;
                const bool mean = m.cpKind[e] == 0;
                if constexpr (NC == 9) {
                    if (vecMode == -2) {
                        double tA[9];
                        transformTen(T, acc, tA);
#pragma unroll
                        for (int k = 0; k < 9; ++k) acc[k] = mean ? (acc[k] + tA[k]) / 2.0 : tA[k];
                        continue;
                    }
                }
                if (vecMode >= 0 && vecMode + 3 <= NC) {
                    double v[3], tv[3];
                    for (int k = 0; k < 3; ++k) v[k] = acc[vecMode + k];
                    transformVec(T, v, tv);
                    for (int k = 0; k < 3; ++k) acc[vecMode + k] = mean ? (v[k] + tv[k]) / 2.0 : tv[k];
                }
            }
        }
    }
    double* o = ptF + (size_t)p * ptStride + ptOffset;
#pragma unroll
    for (int k = 0; k < NC; ++k) o[k] = acc[k];
}

// cell -> vertex interpolation of a plain NC-component field with the structure of pointInterpRecKernel (sliced-ELL list
// read as contiguous runs, eight gathers in flight before the ordered sum)
template <int NC>
__global__ __launch_bounds__(QGD_BLOCK) void pointInterpFastKernel(const MeshView m, const double* __restrict__ cellF,
                                                                  double* __restrict__ ptF) {
    const int p = xcdTile((int)gridDim.x, m.xcdRun) * QGD_BLOCK + threadIdx.x;
    if (p >= m.nP) return;
    const int n = m.pcCount[p];
    if (n == 0) return;
    const size_t base = (size_t)m.pcSlice[p >> 6] * 64 + (p & 63);
    double acc[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) acc[k] = 0.0;
    for (int i = 0; i < n; i += 8) {
        int id[8];
        double w[8], r[8][NC];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const bool on = i + q < n;
            id[q] = on ? m.pcCell[base + (size_t)(i + q) * 64] : 0;
            w[q] = on ? m.pcW[base + (size_t)(i + q) * 64] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
            for (int k = 0; k < NC; ++k) r[q][k] = (i + q < n) ? cellF[(size_t)id[q] * NC + k] : 0.0;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (i + q < n)
#pragma unroll
                for (int k = 0; k < NC; ++k) acc[k] += w[q] * r[q][k];
    }
#pragma unroll
    for (int k = 0; k < NC; ++k) ptF[(size_t)p * NC + k] = acc[k];
}

// Workgroup reduction (wave shuffles + 4-entry LDS): slot[0] = max(a), slot[1] = min(b).
// With `accumulate` the slot keeps the running extremum since it was last reset.
template <int BLOCK = QGD_BLOCK>
__device__ __forceinline__ void blockMaxMin(double a, double b, double* __restrict__ slot, const bool accumulate) {
    __shared__ double sa[BLOCK / 64], sb[BLOCK / 64];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        a = fmax(a, __shfl_down(a, off, 64));
        b = fmin(b, __shfl_down(b, off, 64));
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sa[wave] = a; sb[wave] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 1; i < BLOCK / 64; ++i) { a = fmax(a, sa[i]); b = fmin(b, sb[i]); }
        if (accumulate) { a = fmax(a, slot[0]); b = fmin(b, slot[1]); }
        slot[0] = a;
        slot[1] = b;
    }
}
// effective transport coefficients of a cell/patch value.  L0 assumption:
// laminar muEff = mut(0) + mu, alphaEff = gamma*(alpha + alphat(0)) for an
// internal-energy thermo; mu = mu0 + muQGD, alpha = alphah0 + muQGD/PrQGD
// [QGDThermo_8C L91-98, constScPrModel1_8C L106-115].
__device__ __forceinline__ double muEffOf(const GasModel& gm, double muQGD) { return 0.0 + (gm.mu0 + muQGD); }
__device__ __forceinline__ double alphaEffOf(const GasModel& gm, double muQGD) {
#if QGD_F_DIET
    return gm.gamma * ((gm.alphah0 + muQGD * gm.rPrQGD) + 0.0);
#else
    return gm.gamma * ((gm.alphah0 + muQGD / gm.PrQGD) + 0.0);
#endif
}

// fvc::grad(U), Gauss linear (L0), of one cell: gather in ascending face order -- the cell's own velocity once, per face the neighbour
// cell's (cfNbr) or the patch value, the weight and Sf; same operations in the same order as walking owner and neighbour face by
// face.  U of cell c at Ucell[c * CS + CO ...], of patch face b at Ubnd[b * BS + BO ...].  A wavefront of hexahedra takes the six-face
// pass with every load in flight before the first use (the loop is one dependent chain per face: 1.08 ms at 8 M cells; this: see
// DESIGN.md); wavefronts with other cells take the loop as a whole.
template <int CS, int CO, int BS, int BO>
__device__ __forceinline__ void cellGradGauss(const MeshView& m, const int ci, const double* __restrict__ Ucell, const double* __restrict__ Ubnd,
                                              double* __restrict__ gOut) {
    const int n = m.cfCount[ci];
    const size_t base = (size_t)m.cfSlice[ci >> 6] * 64 + (ci & 63);
    const double Uc[3] = {Ucell[(size_t)ci * CS + CO], Ucell[(size_t)ci * CS + CO + 1], Ucell[(size_t)ci * CS + CO + 2]};
    double G[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    auto add = [&](const int it, const int kind, const double (&Uf)[3], const double (&S)[3]) {
        if (kind == 3) return;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int j = 0; j < 3; ++j) G[3 * a + j] = it >= 0 ? G[3 * a + j] + S[a] * Uf[j] : G[3 * a + j] - S[a] * Uf[j];
    };
    if (m.geoPos && __ballot(n != 6) == 0) {
        // the faces' {w, Sf} at their slot-major positions: one contiguous 2-KB run per slot and wavefront (by label, on a hexahedral
        // box, every third double of twelve lines per array: 0.30 of this kernel's 0.69 ms at 8 M cells)
        int ps[6], nb[6];
        double4 ge[6];
        double Un[6][3];
#pragma unroll
        for (int q = 0; q < 6; ++q) { ps[q] = m.cfPos[base + (size_t)q * 64]; nb[q] = m.cfNbr[base + (size_t)q * 64]; }
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const int pos = ps[q] >= 0 ? ps[q] : ~ps[q];
            ge[q] = m.geoPos[pos];
            const double* __restrict__ src = nb[q] >= 0 ? Ucell + (size_t)nb[q] * CS + CO : Ubnd + (size_t)(pos - m.nIF) * BS + BO;
            Un[q][0] = src[0]; Un[q][1] = src[1]; Un[q][2] = src[2];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const double w = ge[q].x, S[3] = {ge[q].y, ge[q].z, ge[q].w};
            double Uf[3];
#pragma unroll
            for (int k = 0; k < 3; ++k)
                Uf[k] = nb[q] < 0 ? Un[q][k] : (ps[q] >= 0 ? lerpf(w, Uc[k], Un[q][k]) : lerpf(w, Un[q][k], Uc[k]));   // lerp(w, owner, neighbour)
            add(ps[q], w < 0.0 ? 3 : 0, Uf, S);
        }
    } else if (__ballot(n != 6) == 0) {
        int it[6], nb[6], kind[6];
        double w[6], S[6][3], Un[6][3];
#pragma unroll
        for (int q = 0; q < 6; ++q) { it[q] = m.cfItem[base + (size_t)q * 64]; nb[q] = m.cfNbr[base + (size_t)q * 64]; }
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const int f = it[q] >= 0 ? it[q] : ~it[q];
            kind[q] = m.fkind[f];
            w[q] = m.w[f];
            S[q][0] = m.Sx[f]; S[q][1] = m.Sy[f]; S[q][2] = m.Sz[f];
            const double* __restrict__ src = nb[q] >= 0 ? Ucell + (size_t)nb[q] * CS + CO : Ubnd + (size_t)(f - m.nIF) * BS + BO;
            Un[q][0] = src[0]; Un[q][1] = src[1]; Un[q][2] = src[2];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            double Uf[3];
#pragma unroll
            for (int k = 0; k < 3; ++k)
                Uf[k] = nb[q] < 0 ? Un[q][k] : (it[q] >= 0 ? lerpf(w[q], Uc[k], Un[q][k]) : lerpf(w[q], Un[q][k], Uc[k]));   // lerp(w, owner, neighbour)
            add(it[q], kind[q], Uf, S[q]);
        }
    } else if (m.geoPos) {
        // any cell shapes (config 5: every seventh quadrilateral split, hardly a wavefront of six-faced cells): eight faces per pass, their
        // positions and neighbours first, then geometry and values in flight before the ordered sums (the plain loop below is one
        // dependent chain per face: 2.66 ms at 16 M irregular cells)
        for (int i0 = 0; i0 < n; i0 += 8) {
            int ps[8], nb[8];
            double4 ge[8];
            double Un[8][3];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const bool on = i0 + u < n;
                ps[u] = on ? m.cfPos[base + (size_t)(i0 + u) * 64] : 0;
                nb[u] = on ? m.cfNbr[base + (size_t)(i0 + u) * 64] : -1;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const bool on = i0 + u < n;
                const int pos = ps[u] >= 0 ? ps[u] : ~ps[u];
                ge[u] = on ? m.geoPos[pos] : make_double4(-1.0, 0.0, 0.0, 0.0);   // w < 0: skipped like the face of an empty patch
                const double* __restrict__ src = nb[u] >= 0 ? Ucell + (size_t)nb[u] * CS + CO : (on ? Ubnd + (size_t)(pos - m.nIF) * BS + BO : Ucell + (size_t)ci * CS + CO);
                Un[u][0] = src[0]; Un[u][1] = src[1]; Un[u][2] = src[2];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const double w = ge[u].x, S[3] = {ge[u].y, ge[u].z, ge[u].w};
                double Uf[3];
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    Uf[k] = nb[u] < 0 ? Un[u][k] : (ps[u] >= 0 ? lerpf(w, Uc[k], Un[u][k]) : lerpf(w, Un[u][k], Uc[k]));
                add(ps[u], w < 0.0 ? 3 : 0, Uf, S);
            }
        }
    } else {
        for (int i = 0; i < n; ++i) {
            const int it = m.cfItem[base + (size_t)i * 64];
            const int nb = m.cfNbr[base + (size_t)i * 64];
            const int f = it >= 0 ? it : ~it;
            const int kind = m.fkind[f];
            if (kind == 3) continue;
            double Uf[3];
            if (nb >= 0) {
                const double w = m.w[f];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const double un = Ucell[(size_t)nb * CS + CO + k];
                    Uf[k] = it >= 0 ? lerpf(w, Uc[k], un) : lerpf(w, un, Uc[k]);
                }
            } else {
#pragma unroll
                for (int k = 0; k < 3; ++k) Uf[k] = Ubnd[(size_t)(f - m.nIF) * BS + BO + k];
            }
            const double S[3] = {m.Sx[f], m.Sy[f], m.Sz[f]};
            add(it, kind, Uf, S);
        }
    }
    const double V = m.V[ci];
#pragma unroll
    for (int k = 0; k < 9; ++k) gOut[(size_t)ci * 9 + k] = G[k] / V;
}

}  // namespace qgd
