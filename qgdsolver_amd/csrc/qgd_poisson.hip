// qgd_poisson.hip -- the pressure equation of QHDFoam on the device.
//
//   QHDpEqn.H L35-47:   fvc::div(phiu) - fvc::div(phiwo) - fvm::laplacian(taubyrhof, p) == 0,   phi = phiu - phiwo + pEqn.flux()
//
// In OpenFOAM the matrix assembly and the linear solver behind fvm::laplacian / fvScalarMatrix::solve are the
// framework's (L0: Gauss laplacian with the uncorrected surface-normal gradient, PCG).  Here: the face coefficients
// a_f = Gamma_f |S_f| delta_f, a cell-gather (owner/neighbour lists in ascending face order, no atomics) for diagonal,
// source and the matrix-vector product, and a Jacobi-preconditioned conjugate-gradient loop whose dot products are
// two-level block sums in a fixed order -- so a solve is reproducible bit for bit.  Convergence is judged the way
// OpenFOAM's lduMatrix solvers do: sum|b - A x| / normFactor, normFactor = sum(|A x - A xbar| + |b - A xbar|).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>

#include "qgd_device.hpp"

namespace qgd {

namespace {

struct PoissonView {
    const double* a;      // nF   Gamma |S| delta (boundary faces: with the patch deltaCoeff)
    const double* gs;     // nBF  Gamma |S| on boundary faces
    const uint8_t* bKind; // nBF  0 none (zeroGradient, empty, halo), 1 fixedValue, 2 fixedGradient
    const double* pb;     // nBF
    const double* gb;     // nBF
    double* diag;         // nC
    double* rhs;          // nC
};

#define PB 256

// Control block of a solve (device doubles, PressureSolver::ctl): the sums that have to be global in a sharded run sit in
// the front so that a caller all-reduces them in place between the phases; the rest is the loop's own state.  Once
// C_DONE is set every kernel of the loop returns at once: the host may run ahead of the device without reading anything back.
enum CtlSlot : int { C_ABSR = 0, C_SUMX = 1, C_N = 2, C_NORM = 3, C_RZ = 4, C_DQ = 5, C_ABSR2 = 6, C_RZNEW = 7, C_SHIFT = 8,
                     C_RES = 9, C_RES0 = 10, C_DONE = 11, C_ITER = 12, C_ALPHA = 13, C_BETA = 14, C_NORMF = 15, C_COUNT = 16 };
__device__ __forceinline__ bool solveDone(const double* __restrict__ ctl) { return ctl != nullptr && ctl[C_DONE] != 0.0; }


__device__ __forceinline__ double blockSum(double v) {
    __shared__ double s[PB / 64];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < PB / 64; ++i) t += s[i];
    }
    __syncthreads();
    return t;  // valid in thread 0
}

__global__ __launch_bounds__(PB) void coeffKernel(const MeshView m, const double* __restrict__ gamma, double* __restrict__ a,
                                                   double* __restrict__ gs) {
    const int f = blockIdx.x * PB + threadIdx.x;
    if (f >= m.nF) return;
    const double g = gamma[f] * m.magSf[f];
    a[f] = g * m.dn[f];
    if (f >= m.nIF) gs[f - m.nIF] = g;
}

// diagonal and source of one cell: its faces in ascending label order
__global__ __launch_bounds__(PB) void assembleKernel(const MeshView m, const PoissonView v, const double* __restrict__ phiu,
                                                      const double* __restrict__ phiwo, const int refCell, double refValue,
                                                      const double* __restrict__ refFrom, const int rowBegin = 0, const int rowEnd = -1) {
    const int c = rowBegin + blockIdx.x * PB + threadIdx.x;
    if (c >= (rowEnd < 0 ? m.nC : rowEnd)) return;
    const int n = m.cfCount[c];
    const size_t base = (size_t)m.cfSlice[c >> 6] * 64 + (c & 63);
    double diag = 0, rhs = 0;
    for (int i = 0; i < n; ++i) {
        const int it = m.cfItem[base + (size_t)i * 64];
        const int f = it >= 0 ? it : ~it;
        if (m.fkind[f] == 3) continue;  // empty patches
        const double flux = phiu[f] - phiwo[f];
        rhs = it >= 0 ? rhs - flux : rhs + flux;  // -(fvc::div(phiu) - fvc::div(phiwo)) V
        if (f < m.nIF) diag += v.a[f];
        else {
            const int b = f - m.nIF;
            if (v.bKind[b] == 1) { diag += v.a[f]; rhs += v.a[f] * v.pb[b]; }
            else if (v.bKind[b] == 2) rhs += v.gs[b] * v.gb[b];
        }
    }
    if (c == refCell) {  // fvMatrix::setReference (L0): source += diag*value, diag += diag
        if (refFrom) refValue = refFrom[refCell];  // setReference(pRefCell, getRefCellValue(p, pRefCell)) [QHDpEqn.H L43]
        rhs += diag * refValue;
        diag += diag;
    }
    v.diag[c] = diag;
    v.rhs[c] = rhs;
}

// y = A x, optionally the block partial sums of x.y (for p.Ap)
__global__ __launch_bounds__(PB) void applyKernel(const MeshView m, const double* __restrict__ a, const double* __restrict__ diag,
                                                   const double* __restrict__ x, double* __restrict__ y, double* __restrict__ part,
                                                   const int rowBegin = 0, const int rowEnd = -1, const double* __restrict__ ctl = nullptr) {
    // rows [rowBegin, rowEnd) (the owned cells of a shard; columns may be ghost cells); ctl: the solve's control block
    if (solveDone(ctl)) return;
    const int c = rowBegin + blockIdx.x * PB + threadIdx.x;
    double xy = 0;
    if (c < (rowEnd < 0 ? m.nC : rowEnd)) {
        const int n = m.cfCount[c];
        const size_t base = (size_t)m.cfSlice[c >> 6] * 64 + (c & 63);
        const double xc = x[c];
        double s = diag[c] * xc;
        // eight entries per pass, their face labels, neighbour cells, coefficients and x values in flight before the ordered sum
        for (int i0 = 0; i0 < n; i0 += 8) {
            int nbv[8];
            double av[8], xv[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const bool on = i0 + q < n;
                const int it = on ? m.cfItem[base + (size_t)(i0 + q) * 64] : 0;
                nbv[q] = on ? m.cfNbr[base + (size_t)(i0 + q) * 64] : -1;
                const int f = it >= 0 ? it : ~it;
                av[q] = nbv[q] >= 0 ? a[f] : 0.0;
                xv[q] = nbv[q] >= 0 ? x[nbv[q]] : 0.0;
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) if (nbv[q] >= 0) s -= av[q] * xv[q];
        }
        y[c] = s;
        xy = xc * s;
    }
    if (part) {
        const double t = blockSum(xy);
        if (threadIdx.x == 0) part[blockIdx.x] = t;
    }
}

// mode 0: r = b - Ax (Ax in q), z = r/diag, p = z;      partial sums: [r.z, |r|, x]
// mode 1: x += alpha p, r -= alpha q, z = r/diag;       partial sums: [r.z, |r|, x]
__global__ __launch_bounds__(PB) void updateKernel(const int n, const int mode, const double alpha, const double* __restrict__ diag,
                                                    const double* __restrict__ b, double* __restrict__ x, double* __restrict__ r,
                                                    double* __restrict__ z, double* __restrict__ p, const double* __restrict__ q,
                                                    double* __restrict__ part, const int nBlocks) {
    const int c = blockIdx.x * PB + threadIdx.x;
    double rz = 0, ar = 0, xs = 0;
    if (c < n) {
        double rc, xc = x[c];
        if (mode == 0) rc = b[c] - q[c];
        else { xc += alpha * p[c]; x[c] = xc; rc = r[c] - alpha * q[c]; }
        const double zc = rc / diag[c];
        r[c] = rc;
        z[c] = zc;
        if (mode == 0) p[c] = zc;
        rz = rc * zc; ar = fabs(rc); xs = xc;
    }
    const double t0 = blockSum(rz), t1 = blockSum(ar), t2 = blockSum(xs);
    if (threadIdx.x == 0) { part[blockIdx.x] = t0; part[nBlocks + blockIdx.x] = t1; part[2 * nBlocks + blockIdx.x] = t2; }
}

__global__ __launch_bounds__(PB) void directionKernel(const int n, const double beta, const double* __restrict__ z, double* __restrict__ p) {
    const int c = blockIdx.x * PB + threadIdx.x;
    if (c < n) p[c] = z[c] + beta * p[c];
}

// normFactor pieces: sum(|Ax - xbar*A1| + |b - xbar*A1|)
__global__ __launch_bounds__(PB) void normFactorKernel(const int n, const double xbar, const double* __restrict__ Ax,
                                                        const double* __restrict__ A1, const double* __restrict__ b, double* __restrict__ part) {
    const int c = blockIdx.x * PB + threadIdx.x;
    double v = 0;
    if (c < n) { const double ref = xbar * A1[c]; v = fabs(Ax[c] - ref) + fabs(b[c] - ref); }
    const double t = blockSum(v);
    if (threadIdx.x == 0) part[blockIdx.x] = t;
}

__global__ __launch_bounds__(PB) void fillKernel(const int n, const double v, double* __restrict__ x) {
    const int c = blockIdx.x * PB + threadIdx.x;
    if (c < n) x[c] = v;
}

// second level of the sums: one workgroup folds `count` rows of nBlocks partials in ascending order
__global__ __launch_bounds__(PB) void foldKernel(const double* __restrict__ part, const int nBlocks, const int count, double* __restrict__ out) {
    for (int k = 0; k < count; ++k) {
        // four independent chains per thread (the partials of 8 M cells are 31 250: 122 dependent adds per thread otherwise)
        const double* __restrict__ p = part + (size_t)k * nBlocks;
        double v0 = 0, v1 = 0, v2 = 0, v3 = 0;
        int i = threadIdx.x;
        for (; i + 3 * PB < nBlocks; i += 4 * PB) { v0 += p[i]; v1 += p[i + PB]; v2 += p[i + 2 * PB]; v3 += p[i + 3 * PB]; }
        for (; i < nBlocks; i += PB) v0 += p[i];
        const double t = blockSum((v0 + v1) + (v2 + v3));
        if (threadIdx.x == 0) out[k] = t;
    }
}

// phi = phiu - phiwo + pEqn.flux():   -a_f (p_N - p_O) inside, -a_b (p_b - p_O) / -Gamma|S| g_b / 0 on patches
__global__ __launch_bounds__(PB) void fluxKernel(const MeshView m, const PoissonView v, const double* __restrict__ phiu,
                                                  const double* __restrict__ phiwo, const double* __restrict__ p, double* __restrict__ phi) {
    const int f = blockIdx.x * PB + threadIdx.x;
    if (f >= m.nF) return;
    double corr = 0;
    if (f < m.nIF) corr = -v.a[f] * (p[m.nei[f]] - p[m.own[f]]);
    else {
        const int b = f - m.nIF;
        if (v.bKind[b] == 1) corr = -v.a[f] * (v.pb[b] - p[m.own[f]]);
        else if (v.bKind[b] == 2) corr = -v.gs[b] * v.gb[b];
    }
    phi[f] = (m.fkind[f] == 3) ? 0.0 : (phiu[f] - phiwo[f]) + corr;
}

inline int blocksOf(int64_t n) { return (int)((n + PB - 1) / PB); }

#define PCHECK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) throw std::runtime_error(std::string("HIP: ") + hipGetErrorString(e_)); \
    } while (0)

}  // namespace

// Device buffers are the caller's (all on the device of `stream`): gamma, phiu, phiwo [nF]; bKind, pb, gb [nBF];
// p [nC] in/out; phi [nF] out; work = 8*nC + nF + nBF + 3*blocks + 8 doubles.  Returns the iteration count;
// residuals[0..1] = initial and final normalised residual.
int solveQhdPressure(hipStream_t stream, const MeshView& m, const double* gamma, const double* phiu, const double* phiwo,
                     const uint8_t* bKind, const double* pb, const double* gb, int refCell, double refValue, double tolerance,
                     double relTol, int maxIter, double* p, double* phi, double* work, double residuals[2]) {
    const int nC = m.nC, nF = m.nF;
    const int nb = blocksOf(nC);
    double* a = work;
    double* gs = a + nF;
    double* diag = gs + (m.nBF > 0 ? m.nBF : 1);
    double* rhs = diag + nC;
    double* r = rhs + nC;
    double* z = r + nC;
    double* d = z + nC;
    double* q = d + nC;
    double* A1 = q + nC;
    double* ones = A1 + nC;
    double* part = ones + nC;
    double* scal = part + 3 * (size_t)nb;
    PoissonView v{a, gs, bKind, pb, gb, diag, rhs};
    double h[4];

    coeffKernel<<<blocksOf(nF), PB, 0, stream>>>(m, gamma, a, gs);
    assembleKernel<<<nb, PB, 0, stream>>>(m, v, phiu, phiwo, refCell, refValue, nullptr);
    // normFactor (L0: lduMatrix::solver::normFactor)
    fillKernel<<<nb, PB, 0, stream>>>(nC, 1.0, ones);
    applyKernel<<<nb, PB, 0, stream>>>(m, a, diag, ones, A1, nullptr);
    applyKernel<<<nb, PB, 0, stream>>>(m, a, diag, p, q, nullptr);
    updateKernel<<<nb, PB, 0, stream>>>(nC, 0, 0.0, diag, rhs, p, r, z, d, q, part, nb);
    foldKernel<<<1, PB, 0, stream>>>(part, nb, 3, scal);
    PCHECK(hipMemcpyAsync(h, scal, 3 * sizeof(double), hipMemcpyDeviceToHost, stream));
    PCHECK(hipStreamSynchronize(stream));
    double rz = h[0];
    const double sumAbsR = h[1], xbar = h[2] / nC;
    normFactorKernel<<<nb, PB, 0, stream>>>(nC, xbar, q, A1, rhs, part);
    foldKernel<<<1, PB, 0, stream>>>(part, nb, 1, scal);
    PCHECK(hipMemcpyAsync(h, scal, sizeof(double), hipMemcpyDeviceToHost, stream));
    PCHECK(hipStreamSynchronize(stream));
    const double normFactor = h[0] + 1e-20;
    double res = sumAbsR / normFactor;
    residuals[0] = res;
    int it = 0;
    while (it < maxIter && !(res < tolerance || (relTol > 0 && res < relTol * residuals[0]))) {
        applyKernel<<<nb, PB, 0, stream>>>(m, a, diag, d, q, part);
        foldKernel<<<1, PB, 0, stream>>>(part, nb, 1, scal);
        PCHECK(hipMemcpyAsync(h, scal, sizeof(double), hipMemcpyDeviceToHost, stream));
        PCHECK(hipStreamSynchronize(stream));
        const double dq = h[0];
        if (!(dq > 0) || !(rz > 0)) break;  // converged to round-off (or a singular system without reference)
        const double alpha = rz / dq;
        updateKernel<<<nb, PB, 0, stream>>>(nC, 1, alpha, diag, rhs, p, r, z, d, q, part, nb);
        foldKernel<<<1, PB, 0, stream>>>(part, nb, 3, scal);
        PCHECK(hipMemcpyAsync(h, scal, 3 * sizeof(double), hipMemcpyDeviceToHost, stream));
        PCHECK(hipStreamSynchronize(stream));
        const double rzNew = h[0];
        res = h[1] / normFactor;
        directionKernel<<<nb, PB, 0, stream>>>(nC, rzNew / rz, z, d);
        rz = rzNew;
        ++it;
    }
    residuals[1] = res;
    fluxKernel<<<blocksOf(nF), PB, 0, stream>>>(m, v, phiu, phiwo, p, phi);
    PCHECK(hipGetLastError());
    PCHECK(hipStreamSynchronize(stream));
    return it;
}


// Jacobi-preconditioned conjugate gradients on  y_c = diag_c x_c - sum_{internal faces of c} a_f x_nb  (a symmetric positive
// definite "diagonal + Gauss laplacian" system: the implicit-diffusion solves of QGDUEqn.H L56-68 / QGDEEqn.H L55-61), with
// the reproducible two-level sums and OpenFOAM's normalised residual.  work: 6*nC + 3*blocks + 8 doubles.  Returns iterations.
int diagLaplacianPcg(hipStream_t stream, const MeshView& m, const double* a, const double* diag, const double* rhs, double* x,
                     double* work, double tolerance, int maxIter, double residuals[2]) {
    const int nC = m.nC, nb = blocksOf(nC);
    double* r = work;
    double* z = r + nC;
    double* d = z + nC;
    double* q = d + nC;
    double* A1 = q + nC;
    double* ones = A1 + nC;
    double* part = ones + nC;
    double* scal = part + 3 * (size_t)nb;
    double h[4];
    fillKernel<<<nb, PB, 0, stream>>>(nC, 1.0, ones);
    applyKernel<<<nb, PB, 0, stream>>>(m, a, diag, ones, A1, nullptr);
    applyKernel<<<nb, PB, 0, stream>>>(m, a, diag, x, q, nullptr);
    updateKernel<<<nb, PB, 0, stream>>>(nC, 0, 0.0, diag, rhs, x, r, z, d, q, part, nb);
    foldKernel<<<1, PB, 0, stream>>>(part, nb, 3, scal);
    PCHECK(hipMemcpyAsync(h, scal, 3 * sizeof(double), hipMemcpyDeviceToHost, stream));
    PCHECK(hipStreamSynchronize(stream));
    double rz = h[0];
    const double sumAbsR = h[1], xbar = h[2] / nC;
    normFactorKernel<<<nb, PB, 0, stream>>>(nC, xbar, q, A1, rhs, part);
    foldKernel<<<1, PB, 0, stream>>>(part, nb, 1, scal);
    PCHECK(hipMemcpyAsync(h, scal, sizeof(double), hipMemcpyDeviceToHost, stream));
    PCHECK(hipStreamSynchronize(stream));
    const double normFactor = h[0] + 1e-20;
    double res = sumAbsR / normFactor;
    residuals[0] = res;
    int it = 0;
    while (it < maxIter && !(res < tolerance)) {
        applyKernel<<<nb, PB, 0, stream>>>(m, a, diag, d, q, part);
        foldKernel<<<1, PB, 0, stream>>>(part, nb, 1, scal);
        PCHECK(hipMemcpyAsync(h, scal, sizeof(double), hipMemcpyDeviceToHost, stream));
        PCHECK(hipStreamSynchronize(stream));
        const double dq = h[0];
        if (!(dq > 0) || !(rz > 0)) break;
        const double alpha = rz / dq;
        updateKernel<<<nb, PB, 0, stream>>>(nC, 1, alpha, diag, rhs, x, r, z, d, q, part, nb);
        foldKernel<<<1, PB, 0, stream>>>(part, nb, 3, scal);
        PCHECK(hipMemcpyAsync(h, scal, 3 * sizeof(double), hipMemcpyDeviceToHost, stream));
        PCHECK(hipStreamSynchronize(stream));
        const double rzNew = h[0];
        res = h[1] / normFactor;
        directionKernel<<<nb, PB, 0, stream>>>(nC, rzNew / rz, z, d);
        rz = rzNew;
        ++it;
    }
    residuals[1] = res;
    PCHECK(hipGetLastError());
    return it;
}

// ---------------------------------------------------------------------------------------------------------------------
// Aggregation multigrid as the preconditioner of the same conjugate-gradient loop.
//
// The Jacobi-PCG above needs 772 / 1174 iterations at 2 M / 8 M cells; QHDFoam solves this equation every step
// [QHDpEqn.H L36-47], and in QHDFoam its matrix never changes (taubyrhof is fixed after start-up), so a hierarchy built once
// pays for itself at the first step.  Built on the host from the face coefficients: pairwise matching along the strongest
// connection, two passes per level (aggregates of about four cells; three passes -- the eight-cell aggregates of OpenFOAM's
// faceAreaPair agglomeration -- need twice the iterations here), Galerkin coarse operators with piecewise-constant prolongation (coarse face coefficient = sum of the fine
// ones between two aggregates).  One V-cycle = nu damped-Jacobi sweeps before and after the coarse-grid correction, which is
// over-weighted (x += oc * P e_c, oc = 1.8: plain aggregation under-estimates smooth corrections); with equal pre- and
// post-smoothing the cycle is a symmetric positive definite operator, as CG needs.  Every sum is a gather in a fixed order:
// a solve is reproducible bit for bit, like the Jacobi variant.
// ---------------------------------------------------------------------------------------------------------------------
namespace {

template <typename T>
struct MgLevelT {
    int n = 0, width = 0;
    T* diag = nullptr;          // n
    int* col = nullptr;         // width*n, column-major ELL, -1 = padding
    T* val = nullptr;           // width*n: a_ij > 0 (A_ij = -a_ij)
    int* agg = nullptr;         // n: aggregate of each node in the next level
    int* aggStart = nullptr;    // nNext+1
    int* aggItems = nullptr;    // n
    T *x = nullptr, *x2 = nullptr, *b = nullptr, *r = nullptr;
};
using MgLevelDev = MgLevelT<double>;   // the cycle in double; MgLevelT<float>: the same cycle as a single-precision preconditioner

// xout = xin + omega (b - A xin)/diag   (xin == nullptr: from zero, xout = omega b/diag);  rout (optional) = b - A xin
template <typename T>
__global__ __launch_bounds__(PB) void mgSmoothKernel(const MgLevelT<T> L, const T omega, const T* __restrict__ b,
                                                     const T* __restrict__ xin, T* __restrict__ xout, T* __restrict__ rout,
                                                     const double* __restrict__ ctl = nullptr) {
    const int i = blockIdx.x * PB + threadIdx.x;
    if (i >= L.n || solveDone(ctl)) return;
    const T d = L.diag[i];
    if (!xin) { xout[i] = omega * b[i] / d; return; }
    const T xi = xin[i];
    T s = d * xi;
    for (int k = 0; k < L.width; ++k) {
        const int c = L.col[(size_t)k * L.n + i];
        if (c >= 0) s -= L.val[(size_t)k * L.n + i] * xin[c];
    }
    const T r = b[i] - s;
    if (rout) rout[i] = r;
    if (xout) xout[i] = xi + omega * r / d;
}
// vectors between the double-precision CG and a single-precision cycle
template <typename A, typename B>
__global__ __launch_bounds__(PB) void mgConvertKernel(const int n, const A* __restrict__ in, B* __restrict__ out, const double* __restrict__ ctl = nullptr) {
    const int i = blockIdx.x * PB + threadIdx.x;
    if (i < n && !solveDone(ctl)) out[i] = (B)in[i];
}
// y = A x through the ELL rows of level 0 (the same matrix as applyKernel walks face by face: 150 instead of 266 us at 8 M rows),
// block partial sums of x.y
__global__ __launch_bounds__(PB) void mgApplyKernel(const MgLevelDev L, const double* __restrict__ x, double* __restrict__ y,
                                                    double* __restrict__ part, const double* __restrict__ ctl = nullptr) {
    const int i = blockIdx.x * PB + threadIdx.x;
    if (solveDone(ctl)) return;
    double xy = 0;
    if (i < L.n) {
        const double xi = x[i];
        double s = L.diag[i] * xi;
        for (int k = 0; k < L.width; ++k) {
            const int c = L.col[(size_t)k * L.n + i];
            if (c >= 0) s -= L.val[(size_t)k * L.n + i] * x[c];
        }
        y[i] = s;
        xy = xi * s;
    }
    const double t = blockSum(xy);
    if (threadIdx.x == 0) part[blockIdx.x] = t;
}
template <typename T>
__global__ __launch_bounds__(PB) void mgRestrictKernel(const int nCoarse, const int* __restrict__ aggStart, const int* __restrict__ aggItems,
                                                       const T* __restrict__ r, T* __restrict__ rc, const double* __restrict__ ctl = nullptr) {
    const int I = blockIdx.x * PB + threadIdx.x;
    if (I >= nCoarse || solveDone(ctl)) return;
    T s = 0;
    for (int k = aggStart[I]; k < aggStart[I + 1]; ++k) s += r[aggItems[k]];
    rc[I] = s;
}
template <typename T>
__global__ __launch_bounds__(PB) void mgProlongKernel(const int n, const int* __restrict__ agg, const T oc, const T* __restrict__ ec,
                                                      T* __restrict__ x, const double* __restrict__ ctl = nullptr) {
    const int i = blockIdx.x * PB + threadIdx.x;
    if (i < n && !solveDone(ctl)) x[i] += oc * ec[agg[i]];
}
// coarsest level: `sweeps` Jacobi sweeps by one workgroup (n <= MG_COARSE_MAX), the iterate in LDS
#define MG_COARSE_MAX 1024
template <typename T>
__global__ __launch_bounds__(1024) void mgCoarseKernel(const MgLevelT<T> L, const T omega, const int sweeps, const T* __restrict__ b,
                                                        T* __restrict__ x, const double* __restrict__ ctl = nullptr) {
    __shared__ T xa[MG_COARSE_MAX], xb[MG_COARSE_MAX];
    if (solveDone(ctl)) return;
    const int i = threadIdx.x;
    const bool on = i < L.n;
    const T d = on ? L.diag[i] : (T)1, bi = on ? b[i] : (T)0;
    T* cur = xa;
    T* nxt = xb;
    // the row of this thread stays in registers across the sweeps (rows wider than MG_COARSE_ROW read the rest from memory)
    constexpr int MG_COARSE_ROW = 32;
    int cc[MG_COARSE_ROW];
    T vv[MG_COARSE_ROW];
#pragma unroll
    for (int k = 0; k < MG_COARSE_ROW; ++k) {
        const bool has = on && k < L.width;
        cc[k] = has ? L.col[(size_t)k * L.n + i] : -1;
        vv[k] = (has && cc[k] >= 0) ? L.val[(size_t)k * L.n + i] : (T)0;
        if (cc[k] < 0) cc[k] = i;   // value 0: reads its own entry
    }
    if (on) cur[i] = omega * bi / d;
    __syncthreads();
    for (int s = 1; s < sweeps; ++s) {
        if (on) {
            T t = d * cur[i];
#pragma unroll
            for (int k = 0; k < MG_COARSE_ROW; ++k) t -= vv[k] * cur[cc[k]];
            for (int k = MG_COARSE_ROW; k < L.width; ++k) {
                const int c = L.col[(size_t)k * L.n + i];
                if (c >= 0) t -= L.val[(size_t)k * L.n + i] * cur[c];
            }
            nxt[i] = cur[i] + omega * (bi - t) / d;
        }
        __syncthreads();
        T* tmp = cur; cur = nxt; nxt = tmp;
    }
    if (on) x[i] = cur[i];
}
// PCG pieces with a general preconditioner.  Vectors are indexed by local cell label; the rows of a solve are the owned cells
// [ob, ob + n) (a shard's ghost cells lie outside that range and only ever appear as columns of the matrix product).
__global__ __launch_bounds__(PB) void dotKernel(const int n, const double* __restrict__ a, const double* __restrict__ b, double* __restrict__ part,
                                                 const double* __restrict__ ctl) {
    if (solveDone(ctl)) return;
    const int c = blockIdx.x * PB + threadIdx.x;
    const double t = blockSum(c < n ? a[c] * b[c] : 0.0);
    if (threadIdx.x == 0) part[blockIdx.x] = t;
}
// x += alpha d, r -= alpha q with alpha from the control block; partial sums [|r|]
__global__ __launch_bounds__(PB) void axpyKernel(const int n, double* __restrict__ x, double* __restrict__ r,
                                                  const double* __restrict__ d, const double* __restrict__ q, double* __restrict__ part,
                                                  const double* __restrict__ ctl) {
    if (solveDone(ctl)) return;
    const double alpha = ctl[C_ALPHA];
    const int c = blockIdx.x * PB + threadIdx.x;
    double ar = 0;
    if (c < n) { x[c] += alpha * d[c]; const double rc = r[c] - alpha * q[c]; r[c] = rc; ar = fabs(rc); }
    const double t = blockSum(ar);
    if (threadIdx.x == 0) part[blockIdx.x] = t;
}
// r = b - q (q = A x); partial sums [|r|, x]
__global__ __launch_bounds__(PB) void residual0Kernel(const int n, const double* __restrict__ b, const double* __restrict__ q,
                                                       const double* __restrict__ x, double* __restrict__ r, double* __restrict__ part, const int nBlocks) {
    const int c = blockIdx.x * PB + threadIdx.x;
    double ar = 0, xs = 0;
    if (c < n) { const double rc = b[c] - q[c]; r[c] = rc; ar = fabs(rc); xs = x[c]; }
    const double t0 = blockSum(ar), t1 = blockSum(xs);
    if (threadIdx.x == 0) { part[blockIdx.x] = t0; part[nBlocks + blockIdx.x] = t1; }
}
// normFactor pieces with xbar = (global sum of x) / (global number of rows) taken from the control block
__global__ __launch_bounds__(PB) void normFactorCtlKernel(const int n, const double* __restrict__ ctl, const double* __restrict__ Ax,
                                                           const double* __restrict__ A1, const double* __restrict__ b, double* __restrict__ part) {
    const int c = blockIdx.x * PB + threadIdx.x;
    const double xbar = ctl[C_SUMX] / ctl[C_N];
    double v = 0;
    if (c < n) { const double ref = xbar * A1[c]; v = fabs(Ax[c] - ref) + fabs(b[c] - ref); }
    const double t = blockSum(v);
    if (threadIdx.x == 0) part[blockIdx.x] = t;
}
// d = z + beta d (beta from the control block; first = 1: d = z)
__global__ __launch_bounds__(PB) void directionCtlKernel(const int n, const int first, const double* __restrict__ z, double* __restrict__ d,
                                                          const double* __restrict__ ctl) {
    if (solveDone(ctl)) return;
    const int c = blockIdx.x * PB + threadIdx.x;
    if (c < n) d[c] = first ? z[c] : z[c] + ctl[C_BETA] * d[c];
}
// folds `count` rows of partials into ctl[first ...] (one workgroup per row, ascending order, four chains per thread)
__global__ __launch_bounds__(PB) void foldCtlKernel(const double* __restrict__ part, const int nBlocks, const int count, double* __restrict__ ctl,
                                                     const int first, const int always) {
    if (!always && solveDone(ctl)) return;
    const int k = blockIdx.x;
    const double* __restrict__ p = part + (size_t)k * nBlocks;
    double v0 = 0, v1 = 0, v2 = 0, v3 = 0;
    int i = threadIdx.x;
    for (; i + 3 * PB < nBlocks; i += 4 * PB) { v0 += p[i]; v1 += p[i + PB]; v2 += p[i + 2 * PB]; v3 += p[i + 3 * PB]; }
    for (; i < nBlocks; i += PB) v0 += p[i];
    const double t = blockSum((v0 + v1) + (v2 + v3));
    if (threadIdx.x == 0) ctl[first + k] = t;
}
// the loop's own bookkeeping, one thread each (what the host did between synchronisations before):
// stage 0: start of a solve (local row count);  1: after the global normFactor: first residual, done?;
// 2: after the global d.Ad: alpha or breakdown;  3: after the global |r|, r.z: residual, iteration count, done?, beta
__global__ void ctlKernel(double* __restrict__ ctl, const int stage, const double nRows, const double tol, const double relTol, const int maxIter) {
    if (stage == 0) {
        ctl[C_N] = nRows; ctl[C_DONE] = 0.0; ctl[C_ITER] = 0.0; ctl[C_ALPHA] = 0.0; ctl[C_BETA] = 0.0; ctl[C_SHIFT] = 0.0;
        ctl[C_RZ] = 0.0; ctl[C_DQ] = 0.0; ctl[C_ABSR2] = 0.0; ctl[C_RZNEW] = 0.0;
    } else if (stage == 1) {
        const double nf = ctl[C_NORM] + 1e-20;
        const double res = ctl[C_ABSR] / nf;
        ctl[C_NORMF] = nf; ctl[C_RES] = res; ctl[C_RES0] = res;
        if (res < tol || maxIter <= 0) ctl[C_DONE] = 1.0;
    } else if (ctl[C_DONE] == 0.0) {
        if (stage == 2) {
            const double dq = ctl[C_DQ], rz = ctl[C_RZ];
            if (!(dq > 0) || !(rz > 0)) ctl[C_DONE] = 2.0;   // converged to round-off (or a singular system without reference)
            else ctl[C_ALPHA] = rz / dq;
        } else {
            const double res = ctl[C_ABSR2] / ctl[C_NORMF];
            const double it = ctl[C_ITER] + 1.0;
            ctl[C_RES] = res; ctl[C_ITER] = it;
            if (res < tol || (relTol > 0 && res < relTol * ctl[C_RES0]) || it >= (double)maxIter) ctl[C_DONE] = 1.0;
            else { ctl[C_BETA] = ctl[C_RZNEW] / ctl[C_RZ]; ctl[C_RZ] = ctl[C_RZNEW]; }
        }
    }
}

}  // namespace

struct PressureSolver {
    MeshView m{};
    hipStream_t stream = nullptr;
    int refCell = -1, precond = 1;
    int ob = 0, oe = 0;             // rows of the system: the owned cells [ob, oe) of the (possibly sharded) mesh
    double omega = 0.8, oc = 1.8;   // measured (8 M cells / 16 M irregular): 0.67 -> 0.8 with 4:1 coarsening 46 -> 24 / 52 -> 25 iterations
    int nu = 2, coarseSweeps = 40;
    std::vector<void*> owned;
    std::vector<MgLevelDev> L;
    std::vector<MgLevelT<float>> Lf;   // single-precision copy of the hierarchy (QGD_MG_F32; level 0 of L stays for the CG's own A x)
    bool f32 = false;
    // finest-level vectors, indexed by local cell label (ghost entries of d are filled by the caller's halo exchange)
    double *a = nullptr, *gs = nullptr, *diag = nullptr, *rhs = nullptr, *r = nullptr, *z = nullptr, *d = nullptr, *q = nullptr, *A1 = nullptr,
           *ones = nullptr, *part = nullptr;
    double* ctl = nullptr;          // control block (CtlSlot)
    double* hostCtl = nullptr;      // pinned mirror: 4 slots of C_COUNT doubles for the run-ahead check + 1 for the final read
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    const uint8_t* bKind = nullptr;
    int64_t bytes = 0;
    // arguments of the solve in flight (set by phase 0)
    const double *phiu = nullptr, *phiwo = nullptr, *pb = nullptr, *gb = nullptr;
    double* p = nullptr;
    double tol = 0, relTol = 0;
    int maxIter = 0;
    // the V-cycle is a fixed sequence of ~75 small launches on fixed buffers (r -> z); with QGD_MG_GRAPH=1 it is captured once into
    // a hipGraph and replayed per CG iteration.  Measured: 5.11 -> 5.03 ms per step at 64^3, nothing at 128^3 / 200^3 (the
    // asynchronous launches were already hidden), and rocprofv3 crashes on the captured graph -- hence opt-in.
    hipGraphExec_t cycleGraph = nullptr;
    bool cycleGraphTried = false;

    template <class T>
    T* alloc(size_t n, const T* host = nullptr) {
        void* p = nullptr;
        const size_t nb = std::max<size_t>(n, 1) * sizeof(T);
        PCHECK(hipMalloc(&p, nb));
        owned.push_back(p);
        bytes += (int64_t)nb;
        if (host && n) PCHECK(hipMemcpy(p, host, n * sizeof(T), hipMemcpyHostToDevice));
        else PCHECK(hipMemset(p, 0, nb));
        return (T*)p;
    }
    ~PressureSolver() {
        if (cycleGraph) (void)hipGraphExecDestroy(cycleGraph);
        for (void* p : owned) (void)hipFree(p);
        if (hostCtl) (void)hipHostFree(hostCtl);
        for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e);
    }

    // z = M r on level l (b -> x), in the precision of the level arrays; every launch returns at once when the solve is done
    template <typename T>
    void vcycleT(std::vector<MgLevelT<T>>& Lv, size_t l, const T* b, T* x) {
        MgLevelT<T>& lv = Lv[l];
        const int nb = blocksOf(lv.n);
        const T om = (T)omega, over = (T)oc;
        const T* none = nullptr;
        T* noOut = nullptr;
        if (l + 1 == Lv.size()) {
            if (lv.n <= MG_COARSE_MAX) mgCoarseKernel<T><<<1, 1024, 0, stream>>>(lv, om, coarseSweeps, b, x, ctl);
            else {
                T* cur = x; T* nxt = lv.x2;
                mgSmoothKernel<T><<<nb, PB, 0, stream>>>(lv, om, b, none, cur, noOut, ctl);
                for (int s = 1; s < coarseSweeps; ++s) { mgSmoothKernel<T><<<nb, PB, 0, stream>>>(lv, om, b, cur, nxt, noOut, ctl); std::swap(cur, nxt); }
                if (cur != x) PCHECK(hipMemcpyAsync(x, cur, sizeof(T) * lv.n, hipMemcpyDeviceToDevice, stream));
            }
            return;
        }
        MgLevelT<T>& nx = Lv[l + 1];
        T* cur = x; T* nxt = lv.x2;
        mgSmoothKernel<T><<<nb, PB, 0, stream>>>(lv, om, b, none, cur, noOut, ctl);
        for (int s = 1; s < nu; ++s) { mgSmoothKernel<T><<<nb, PB, 0, stream>>>(lv, om, b, cur, nxt, noOut, ctl); std::swap(cur, nxt); }
        mgSmoothKernel<T><<<nb, PB, 0, stream>>>(lv, om, b, cur, noOut, lv.r, ctl);           // r = b - A x
        mgRestrictKernel<T><<<blocksOf(nx.n), PB, 0, stream>>>(nx.n, lv.aggStart, lv.aggItems, lv.r, nx.b, ctl);
        vcycleT<T>(Lv, l + 1, nx.b, nx.x);
        mgProlongKernel<T><<<nb, PB, 0, stream>>>(lv.n, lv.agg, over, nx.x, cur, ctl);
        for (int s = 0; s < nu; ++s) { mgSmoothKernel<T><<<nb, PB, 0, stream>>>(lv, om, b, cur, nxt, noOut, ctl); std::swap(cur, nxt); }
        if (cur != x) PCHECK(hipMemcpyAsync(x, cur, sizeof(T) * lv.n, hipMemcpyDeviceToDevice, stream));
    }
    void vcycle(size_t l, const double* b, double* x) {
        if (!Lf.empty() && l == 0) {
            // the cycle as a single-precision operator between double-precision CG vectors: half the bytes of every sweep
            const int nb = blocksOf(L[0].n);
            mgConvertKernel<double, float><<<nb, PB, 0, stream>>>(L[0].n, b, Lf[0].b, ctl);
            vcycleT<float>(Lf, 0, Lf[0].b, Lf[0].x);
            mgConvertKernel<float, double><<<nb, PB, 0, stream>>>(L[0].n, Lf[0].x, x, ctl);
        } else vcycleT<double>(L, l, b, x);
    }
    // z = M r over the owned rows (the multigrid hierarchy is built on the owned block: for a shard it is the additive-Schwarz
    // block of this rank, couplings to ghost cells stay in the diagonal only)
    void precondition() {
        const int n = oe - ob, nb = blocksOf(n);
        if (precond == 1 && !L.empty()) {
            if (!cycleGraphTried) {
                cycleGraphTried = true;
                hipGraph_t g = nullptr;
                if (std::getenv("QGD_MG_GRAPH") && hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
                    bool ok = true;
                    try { vcycle(0, r + ob, z + ob); } catch (...) { ok = false; }
                    if (hipStreamEndCapture(stream, &g) != hipSuccess || !ok || !g) { g = nullptr; (void)hipGetLastError(); }
                    if (g) {
                        if (hipGraphInstantiate(&cycleGraph, g, nullptr, nullptr, 0) != hipSuccess) { cycleGraph = nullptr; (void)hipGetLastError(); }
                        (void)hipGraphDestroy(g);
                    }
                }
            }
            if (cycleGraph) PCHECK(hipGraphLaunch(cycleGraph, stream));
            else vcycle(0, r + ob, z + ob);
        } else {
            MgLevelDev jl; jl.n = n; jl.diag = diag + ob;
            const double* none = nullptr; double* noOut = nullptr;
            mgSmoothKernel<double><<<nb, PB, 0, stream>>>(jl, 1.0, r + ob, none, z + ob, noOut, ctl);   // z = r/diag
        }
    }
};

namespace {
// one pairwise matching pass over a graph (n nodes, edges I<J with weights w): strongest unmatched neighbour in node order;
// nodes left alone join the aggregate of their strongest neighbour
int pairwisePass(int n, const std::vector<int>& I, const std::vector<int>& J, const std::vector<double>& w, std::vector<int>& agg) {
    const size_t E = I.size();
    std::vector<int64_t> off((size_t)n + 1, 0);
    for (size_t e = 0; e < E; ++e) { off[I[e] + 1]++; off[J[e] + 1]++; }
    for (int i = 0; i < n; ++i) off[i + 1] += off[i];
    std::vector<int> nb((size_t)off[n]);
    std::vector<double> nw((size_t)off[n]);
    std::vector<int64_t> fill(off.begin(), off.end() - 1);
    for (size_t e = 0; e < E; ++e) {
        nb[fill[I[e]]] = J[e]; nw[fill[I[e]]++] = w[e];
        nb[fill[J[e]]] = I[e]; nw[fill[J[e]]++] = w[e];
    }
    agg.assign((size_t)n, -1);
    int na = 0;
    for (int i = 0; i < n; ++i) {
        if (agg[i] >= 0) continue;
        int best = -1; double bw = -1.0;
        for (int64_t k = off[i]; k < off[i + 1]; ++k) if (agg[nb[k]] < 0 && nb[k] != i && nw[k] > bw) { bw = nw[k]; best = nb[k]; }
        if (best >= 0) { agg[i] = agg[best] = na++; }
    }
    for (int i = 0; i < n; ++i) {
        if (agg[i] >= 0) continue;
        int best = -1; double bw = -1.0;
        for (int64_t k = off[i]; k < off[i + 1]; ++k) if (agg[nb[k]] >= 0 && nw[k] > bw) { bw = nw[k]; best = nb[k]; }
        agg[i] = best >= 0 ? agg[best] : na++;
    }
    return na;
}
// Galerkin coarse graph for piecewise-constant prolongation
void coarsenGraph(int na, const std::vector<int>& agg, std::vector<int>& I, std::vector<int>& J, std::vector<double>& w, std::vector<double>& diag) {
    std::vector<double> dc((size_t)na, 0.0);
    for (size_t i = 0; i < diag.size(); ++i) dc[agg[i]] += diag[i];
    std::vector<std::pair<int64_t, double>> ed;
    ed.reserve(I.size());
    for (size_t e = 0; e < I.size(); ++e) {
        const int a = agg[I[e]], b = agg[J[e]];
        if (a == b) { dc[a] -= 2.0 * w[e]; continue; }
        ed.push_back({(int64_t)std::min(a, b) * na + std::max(a, b), w[e]});
    }
    std::sort(ed.begin(), ed.end(), [](const std::pair<int64_t, double>& x, const std::pair<int64_t, double>& y) { return x.first < y.first; });
    I.clear(); J.clear(); w.clear();
    for (size_t k = 0; k < ed.size();) {
        double s = 0; size_t j = k;
        while (j < ed.size() && ed[j].first == ed[k].first) s += ed[j++].second;
        I.push_back((int)(ed[k].first / na)); J.push_back((int)(ed[k].first % na)); w.push_back(s);
        k = j;
    }
    diag.swap(dc);
}
}  // namespace

static void mgUploadLevel(PressureSolver* S, int n, const std::vector<int>& I, const std::vector<int>& J, const std::vector<double>& w,
                          const std::vector<double>& diag) {
    std::vector<int> deg((size_t)n, 0);
    for (size_t e = 0; e < I.size(); ++e) { deg[I[e]]++; deg[J[e]]++; }
    int width = 0;
    for (int d : deg) width = std::max(width, d);
    std::vector<int> col((size_t)width * n, -1), fill((size_t)n, 0);
    std::vector<double> val((size_t)width * n, 0.0);
    for (size_t e = 0; e < I.size(); ++e) {
        const int i = I[e], j = J[e];
        col[(size_t)fill[i] * n + i] = j; val[(size_t)fill[i]++ * n + i] = w[e];
        col[(size_t)fill[j] * n + j] = i; val[(size_t)fill[j]++ * n + j] = w[e];
    }
    MgLevelDev lv;
    lv.n = n; lv.width = width;
    lv.diag = S->alloc<double>(n, diag.data());
    lv.col = S->alloc<int>(col.size(), col.data());
    lv.val = S->alloc<double>(val.size(), val.data());
    if (S->f32) {
        // the double-precision level keeps only what the CG's own matrix product needs (level 0) or nothing
        std::vector<float> vf(val.begin(), val.end()), df(diag.begin(), diag.end());
        MgLevelT<float> lf;
        lf.n = n; lf.width = width; lf.col = lv.col;
        lf.diag = S->alloc<float>(n, df.data());
        lf.val = S->alloc<float>(vf.size(), vf.data());
        lf.x = S->alloc<float>(n); lf.x2 = S->alloc<float>(n); lf.b = S->alloc<float>(n); lf.r = S->alloc<float>(n);
        S->Lf.push_back(lf);
    } else {
        lv.x = S->alloc<double>(n); lv.x2 = S->alloc<double>(n); lv.b = S->alloc<double>(n); lv.r = S->alloc<double>(n);
    }
    S->L.push_back(lv);
}

PressureSolver* pressureSolverCreate(hipStream_t stream, const MeshView& m, const double* taubyrho, const uint8_t* bKind, int refCell,
                                     int precond, int ownedBegin, int ownedEnd) {
    PressureSolver* S = new PressureSolver();
    try {
        S->m = m; S->stream = stream; S->refCell = refCell; S->precond = precond; S->bKind = bKind;
        S->ob = ownedBegin; S->oe = ownedEnd < 0 ? m.nC : ownedEnd;
        if (S->ob < 0 || S->oe > m.nC || S->ob >= S->oe) throw std::invalid_argument("pressureSolverCreate: bad owned range");
        { const char* e = std::getenv("QGD_MG_F32"); S->f32 = !e || std::atoi(e) != 0; }   // default: single-precision cycle
        // tuning knobs of the cycle (experiments; the defaults are what the tests and DESIGN.md's numbers use)
        // out-of-range values are refused (an omega of 0 would stall the smoother silently)
        auto knob = [](const char* name, double dflt, double lo, double hi) {
            const char* e = std::getenv(name);
            if (!e || !*e) return dflt;
            char* end = nullptr;
            const double v = std::strtod(e, &end);
            if (!end || *end != '\0' || !(v >= lo && v <= hi))
                throw std::invalid_argument(std::string(name) + "=" + e + " is outside [" + std::to_string(lo) + ", " + std::to_string(hi) + "]");
            return v;
        };
        S->nu = (int)knob("QGD_MG_NU", S->nu, 1, 8);
        S->oc = knob("QGD_MG_OC", S->oc, 0.5, 3.0);
        S->omega = knob("QGD_MG_OMEGA", S->omega, 0.1, 1.0);
        S->coarseSweeps = (int)knob("QGD_MG_COARSE_SWEEPS", S->coarseSweeps, 1, 1000);
        // pairwise matching passes per level: 2 = aggregates of ~4 cells (3 passes = ~8 cells need twice the iterations)
        const int passes = (int)knob("QGD_MG_PASSES", 2, 1, 4);
        const int nC = m.nC, nF = m.nF, ob = S->ob, oe = S->oe, nRows = oe - ob, nb = blocksOf(nRows);
        S->a = S->alloc<double>(nF); S->gs = S->alloc<double>(std::max(m.nBF, 1));
        S->diag = S->alloc<double>(nC); S->rhs = S->alloc<double>(nC); S->r = S->alloc<double>(nC); S->z = S->alloc<double>(nC);
        S->d = S->alloc<double>(nC); S->q = S->alloc<double>(nC); S->A1 = S->alloc<double>(nC); S->ones = S->alloc<double>(nC);
        S->part = S->alloc<double>(3 * (size_t)std::max(nb, 1)); S->ctl = S->alloc<double>(C_COUNT);
        PCHECK(hipHostMalloc((void**)&S->hostCtl, sizeof(double) * C_COUNT * 5, hipHostMallocDefault));
        for (hipEvent_t& e : S->ev) PCHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        coeffKernel<<<blocksOf(nF), PB, 0, stream>>>(m, taubyrho, S->a, S->gs);
        // static part of the matrix: diagonal (with the doubled reference row) from a zero-flux assembly
        PoissonView v{S->a, S->gs, bKind, S->rhs /*unused*/, S->rhs, S->diag, S->rhs};
        double* zeros = S->alloc<double>(nF);
        double* zb = S->alloc<double>(std::max(m.nBF, 1));
        v.pb = zb; v.gb = zb;
        fillKernel<<<blocksOf(nC), PB, 0, stream>>>(nC, 1.0, S->diag);   // rows outside the owned range: identity (never used)
        assembleKernel<<<nb, PB, 0, stream>>>(m, v, zeros, zeros, refCell, 0.0, nullptr, ob, oe);
        fillKernel<<<blocksOf(nC), PB, 0, stream>>>(nC, 1.0, S->ones);
        applyKernel<<<nb, PB, 0, stream>>>(m, S->a, S->diag, S->ones, S->A1, nullptr, ob, oe);
        PCHECK(hipStreamSynchronize(stream));
        if (precond == 1) {
            // the hierarchy of the owned block: couplings between two owned cells (a shard's couplings to ghost cells stay in
            // the diagonal: the block of an additive-Schwarz preconditioner)
            std::vector<int> own((size_t)m.nIF), nei((size_t)m.nIF);
            std::vector<double> wAll((size_t)m.nIF), dAll((size_t)nC);
            if (m.nIF) {
                PCHECK(hipMemcpy(own.data(), m.own, sizeof(int) * (size_t)m.nIF, hipMemcpyDeviceToHost));
                PCHECK(hipMemcpy(nei.data(), m.nei, sizeof(int) * (size_t)m.nIF, hipMemcpyDeviceToHost));
                PCHECK(hipMemcpy(wAll.data(), S->a, sizeof(double) * (size_t)m.nIF, hipMemcpyDeviceToHost));
            }
            PCHECK(hipMemcpy(dAll.data(), S->diag, sizeof(double) * (size_t)nC, hipMemcpyDeviceToHost));
            std::vector<int> I, J;
            std::vector<double> w, diag(dAll.begin() + ob, dAll.begin() + oe);
            I.reserve((size_t)m.nIF); J.reserve((size_t)m.nIF); w.reserve((size_t)m.nIF);
            for (int f = 0; f < m.nIF; ++f)
                if (own[f] >= ob && own[f] < oe && nei[f] >= ob && nei[f] < oe) { I.push_back(own[f] - ob); J.push_back(nei[f] - ob); w.push_back(wAll[f]); }
            std::vector<int>().swap(own); std::vector<int>().swap(nei); std::vector<double>().swap(wAll); std::vector<double>().swap(dAll);
            int n = nRows;
            mgUploadLevel(S, n, I, J, w, diag);
            while (n > 600 && S->L.size() < 12) {
                std::vector<int> total((size_t)n);
                for (int i = 0; i < n; ++i) total[i] = i;
                int cur = n;
                for (int pass = 0; pass < passes && cur > 64; ++pass) {
                    std::vector<int> agg;
                    const int na = pairwisePass(cur, I, J, w, agg);
                    coarsenGraph(na, agg, I, J, w, diag);
                    for (int i = 0; i < n; ++i) total[i] = agg[total[i]];
                    cur = na;
                }
                if (cur >= n) break;
                // aggregate lists of the level just finished (CSR by coarse node, members in ascending order)
                std::vector<int> start((size_t)cur + 1, 0), items((size_t)n);
                for (int i = 0; i < n; ++i) start[total[i] + 1]++;
                for (int k = 0; k < cur; ++k) start[k + 1] += start[k];
                std::vector<int> fill(start.begin(), start.end() - 1);
                for (int i = 0; i < n; ++i) items[fill[total[i]]++] = i;
                MgLevelDev& fine = S->L.back();
                fine.agg = S->alloc<int>(n, total.data());
                fine.aggStart = S->alloc<int>((size_t)cur + 1, start.data());
                fine.aggItems = S->alloc<int>(n, items.data());
                if (S->f32) { MgLevelT<float>& ff = S->Lf.back(); ff.agg = fine.agg; ff.aggStart = fine.aggStart; ff.aggItems = fine.aggItems; }
                n = cur;
                mgUploadLevel(S, n, I, J, w, diag);
            }
        }
    } catch (...) { delete S; throw; }
    return S;
}
void pressureSolverFree(PressureSolver* S) { delete S; }
int64_t pressureSolverBytes(const PressureSolver* S) { return S ? S->bytes : 0; }
int pressureSolverLevels(const PressureSolver* S, int* sizes, int cap) {
    if (!S) return 0;
    for (size_t l = 0; l < S->L.size() && (int)l < cap; ++l) sizes[l] = S->L[l].n;
    return (int)S->L.size();
}
double* pressureSolverCtl(PressureSolver* S) { return S->ctl; }
double* pressureSolverDirection(PressureSolver* S) { return S->d; }

// ---------------------------------------------------------------------------------------------------------------------
// One solve of QHDpEqn.H L36-47 with the persistent solver, as stream-ordered phases with NO host synchronisation inside:
// every scalar of the loop (alpha, beta, the residual, the iteration count, "done") lives in the control block on the device.
// Between the phases a sharded caller all-reduces (SUM) the named slots of the control block in place and exchanges the
// ghost entries of the search direction; a single rank just runs them back to back.
//   phase 0  rhs from the fluxes and the patch data (reference value read from p itself: setReference(pRefCell,
//            getRefCellValue(p, pRefCell))), q = A p, r = b - q              -> local {sum|r|, sum x, rows} in ctl[0..3)
//   phase 1  normFactor pieces with the global xbar (L0: lduMatrix::solver::normFactor)   -> ctl[3]
//   phase 2  first residual, done?; z = M r, d = z, r.z                     -> ctl[4];   then halo of d
//   phase 3  q = A d, d.q                                                    -> ctl[5]
//   phase 4  alpha (or breakdown), x += alpha d, r -= alpha q, z = M r, {sum|r|, r.z}   -> ctl[6..8)
//   phase 5  residual, iteration count, done?, beta, d = z + beta d;       then halo of d
// The preconditioner is applied before the convergence test of an iteration (one cycle more than the host-driven loop ran,
// at the last iteration only) so that |r| and r.z travel in ONE reduction.
// ---------------------------------------------------------------------------------------------------------------------
void pressureSolveBegin(PressureSolver* S, const double* phiu, const double* phiwo, const double* pb, const double* gb, double tolerance,
                        double relTol, int maxIter, double* p) {
    S->phiu = phiu; S->phiwo = phiwo; S->pb = pb; S->gb = gb; S->tol = tolerance; S->relTol = relTol; S->maxIter = maxIter; S->p = p;
    const MeshView& m = S->m;
    hipStream_t stream = S->stream;
    const int ob = S->ob, oe = S->oe, n = oe - ob, nb = blocksOf(n);
    PoissonView v{S->a, S->gs, S->bKind, pb, gb, S->diag, S->rhs};
    ctlKernel<<<1, 1, 0, stream>>>(S->ctl, 0, (double)n, tolerance, relTol, maxIter);
    assembleKernel<<<nb, PB, 0, stream>>>(m, v, phiu, phiwo, S->refCell, 0.0, p, ob, oe);
    applyKernel<<<nb, PB, 0, stream>>>(m, S->a, S->diag, p, S->q, nullptr, ob, oe);
    residual0Kernel<<<nb, PB, 0, stream>>>(n, S->rhs + ob, S->q + ob, p + ob, S->r + ob, S->part, nb);
    foldCtlKernel<<<2, PB, 0, stream>>>(S->part, nb, 2, S->ctl, C_ABSR, 1);
    PCHECK(hipGetLastError());
}
void pressureSolvePhase(PressureSolver* S, int phase) {
    const MeshView& m = S->m;
    hipStream_t stream = S->stream;
    const int ob = S->ob, oe = S->oe, n = oe - ob, nb = blocksOf(n);
    double* ctl = S->ctl;
    switch (phase) {
        case 1:
            normFactorCtlKernel<<<nb, PB, 0, stream>>>(n, ctl, S->q + ob, S->A1 + ob, S->rhs + ob, S->part);
            foldCtlKernel<<<1, PB, 0, stream>>>(S->part, nb, 1, ctl, C_NORM, 1);
            break;
        case 2:
            ctlKernel<<<1, 1, 0, stream>>>(ctl, 1, 0.0, S->tol, S->relTol, S->maxIter);
            S->precondition();
            directionCtlKernel<<<nb, PB, 0, stream>>>(n, 1, S->z + ob, S->d + ob, ctl);
            dotKernel<<<nb, PB, 0, stream>>>(n, S->r + ob, S->z + ob, S->part, ctl);
            foldCtlKernel<<<1, PB, 0, stream>>>(S->part, nb, 1, ctl, C_RZ, 0);
            break;
        case 3:
            // unsharded: the ELL rows of multigrid level 0 are the whole matrix (150 instead of 266 us at 8 M rows); a shard walks
            // its faces, whose neighbour columns include the ghost cells
            if (S->precond == 1 && !S->L.empty() && ob == 0 && oe == m.nC) mgApplyKernel<<<nb, PB, 0, stream>>>(S->L[0], S->d, S->q, S->part, ctl);
            else applyKernel<<<nb, PB, 0, stream>>>(m, S->a, S->diag, S->d, S->q, S->part, ob, oe, ctl);
            foldCtlKernel<<<1, PB, 0, stream>>>(S->part, nb, 1, ctl, C_DQ, 0);
            break;
        case 4:
            ctlKernel<<<1, 1, 0, stream>>>(ctl, 2, 0.0, S->tol, S->relTol, S->maxIter);
            axpyKernel<<<nb, PB, 0, stream>>>(n, S->p + ob, S->r + ob, S->d + ob, S->q + ob, S->part, ctl);
            S->precondition();
            dotKernel<<<nb, PB, 0, stream>>>(n, S->r + ob, S->z + ob, S->part + nb, ctl);
            foldCtlKernel<<<2, PB, 0, stream>>>(S->part, nb, 2, ctl, C_ABSR2, 0);
            break;
        case 5:
            ctlKernel<<<1, 1, 0, stream>>>(ctl, 3, 0.0, S->tol, S->relTol, S->maxIter);
            directionCtlKernel<<<nb, PB, 0, stream>>>(n, 0, S->z + ob, S->d + ob, ctl);
            break;
        default: throw std::invalid_argument("pressureSolvePhase: phase must be 1..5");
    }
    PCHECK(hipGetLastError());
}
// phi = phiu - phiwo + pEqn.flux() (after the ghost cells of p have been refreshed)
void pressureSolveFlux(PressureSolver* S, double* phi) {
    PoissonView v{S->a, S->gs, S->bKind, S->pb, S->gb, S->diag, S->rhs};
    fluxKernel<<<blocksOf(S->m.nF), PB, 0, S->stream>>>(S->m, v, S->phiu, S->phiwo, S->p, phi);
    PCHECK(hipGetLastError());
}
// {done (1 converged or out of iterations, 2 breakdown), iterations, initial, final normalised residual}; waits for the stream
void pressureSolveStatus(PressureSolver* S, double out[4]) {
    double* h = S->hostCtl + 4 * C_COUNT;
    PCHECK(hipMemcpyAsync(h, S->ctl, sizeof(double) * C_COUNT, hipMemcpyDeviceToHost, S->stream));
    PCHECK(hipStreamSynchronize(S->stream));
    out[0] = h[C_DONE]; out[1] = h[C_ITER]; out[2] = h[C_RES0]; out[3] = h[C_RES];
}
// The loop of a solve after pressureSolveBegin, for callers that own the transport through two hooks (nullptr on one rank):
// allreduce(ptr, n) sums n doubles at ptr over the ranks in place (stream-ordered), haloDirection() refreshes the ghost
// entries of the search direction.  The host runs at most two iterations ahead of the device: it reads the "done" flag of
// iteration i-2 before it queues iteration i, so the device never idles and at most two iterations of early-returning
// launches are wasted.
int pressureSolveRun(PressureSolver* S, const SolveHooks* hooks, double residuals[2]) {
    hipStream_t stream = S->stream;
    double* ctl = S->ctl;
    auto reduce = [&](int first, int count) { if (hooks && hooks->allreduce) hooks->allreduce(ctl + first, count); };
    auto halo = [&]() { if (hooks && hooks->haloDirection) hooks->haloDirection(); };
    reduce(C_ABSR, 3);
    pressureSolvePhase(S, 1);
    reduce(C_NORM, 1);
    pressureSolvePhase(S, 2);
    reduce(C_RZ, 1);
    halo();
    const int ahead = 2;
    for (int it = 0; it < S->maxIter; ++it) {
        if (it >= ahead) {
            const int slot = (it - ahead) & 3;
            PCHECK(hipEventSynchronize(S->ev[slot]));
            if (S->hostCtl[slot * C_COUNT + C_DONE] != 0.0) break;
        }
        pressureSolvePhase(S, 3);
        reduce(C_DQ, 1);
        pressureSolvePhase(S, 4);
        reduce(C_ABSR2, 2);
        pressureSolvePhase(S, 5);
        halo();
        const int slot = it & 3;
        PCHECK(hipMemcpyAsync(S->hostCtl + slot * C_COUNT, ctl, sizeof(double) * C_COUNT, hipMemcpyDeviceToHost, stream));
        PCHECK(hipEventRecord(S->ev[slot], stream));
    }
    double st[4];
    pressureSolveStatus(S, st);
    residuals[0] = st[2]; residuals[1] = st[3];
    return (int)st[1];
}
// measurement: `reps` full damped-Jacobi sweeps of multigrid level 0 (the kernel a solve spends most of its time in) between two
// HIP events on the solver's stream; returns the average milliseconds per sweep (0 without a hierarchy).  rows / width: the ELL
// shape of that level, for the byte model.
double pressureSolverSweepMs(PressureSolver* S, int reps, int* rows, int* width) {
    if (rows) *rows = 0;
    if (width) *width = 0;
    if (S->L.empty() || reps <= 0) return 0.0;
    hipEvent_t a, b;
    PCHECK(hipEventCreate(&a)); PCHECK(hipEventCreate(&b));
    float ms = 0;
    const int nb = blocksOf(S->L[0].n);
    if (rows) *rows = S->L[0].n;
    if (width) *width = S->L[0].width;
    if (!S->Lf.empty()) {
        MgLevelT<float>& lv = S->Lf[0];
        float* noOut = nullptr;
        mgSmoothKernel<float><<<nb, PB, 0, S->stream>>>(lv, (float)S->omega, lv.b, lv.x, lv.x2, noOut, nullptr);
        PCHECK(hipEventRecord(a, S->stream));
        for (int i = 0; i < reps; ++i) mgSmoothKernel<float><<<nb, PB, 0, S->stream>>>(lv, (float)S->omega, lv.b, (i & 1) ? lv.x2 : lv.x, (i & 1) ? lv.x : lv.x2, noOut, nullptr);
        PCHECK(hipEventRecord(b, S->stream));
    } else {
        MgLevelDev& lv = S->L[0];
        double* noOut = nullptr;
        mgSmoothKernel<double><<<nb, PB, 0, S->stream>>>(lv, S->omega, lv.b, lv.x, lv.x2, noOut, nullptr);
        PCHECK(hipEventRecord(a, S->stream));
        for (int i = 0; i < reps; ++i) mgSmoothKernel<double><<<nb, PB, 0, S->stream>>>(lv, S->omega, lv.b, (i & 1) ? lv.x2 : lv.x, (i & 1) ? lv.x : lv.x2, noOut, nullptr);
        PCHECK(hipEventRecord(b, S->stream));
    }
    PCHECK(hipEventSynchronize(b));
    PCHECK(hipEventElapsedTime(&ms, a, b));
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    return (double)ms / reps;
}
bool pressureSolverSinglePrecisionCycle(const PressureSolver* S) { return !S->Lf.empty(); }

// single rank, everything: rhs, solve, flux
int pressureSolve(PressureSolver* S, const double* phiu, const double* phiwo, const double* pb, const double* gb, double tolerance,
                  double relTol, int maxIter, double* p, double* phi, double residuals[2]) {
    pressureSolveBegin(S, phiu, phiwo, pb, gb, tolerance, relTol, maxIter, p);
    const int it = pressureSolveRun(S, nullptr, residuals);
    pressureSolveFlux(S, phi);
    return it;
}

}  // namespace qgd
