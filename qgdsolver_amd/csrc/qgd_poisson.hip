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
#include <cstdio>
#include <cstdlib>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>
#include <omp.h>

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
    // eight faces per pass: labels, then kind / fluxes / coefficient of every face in flight before the ordered sums
    for (int i0 = 0; i0 < n; i0 += 8) {
        int itv[8], kind[8];
        double fu[8], fw[8], av[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) itv[u] = i0 + u < n ? m.cfItem[base + (size_t)(i0 + u) * 64] : 0;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const bool on = i0 + u < n;
            const int f = itv[u] >= 0 ? itv[u] : ~itv[u];
            kind[u] = on ? m.fkind[f] : 3;
            fu[u] = on ? phiu[f] : 0.0;
            fw[u] = on ? phiwo[f] : 0.0;
            av[u] = on ? v.a[f] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (kind[u] == 3) continue;  // empty patches (and the entries past the row end)
            const int it = itv[u], f = it >= 0 ? it : ~it;
            const double flux = fu[u] - fw[u];
            rhs = it >= 0 ? rhs - flux : rhs + flux;  // -(fvc::div(phiu) - fvc::div(phiwo)) V
            if (f < m.nIF) diag += av[u];
            else {
                const int b = f - m.nIF;
                if (v.bKind[b] == 1) { diag += av[u]; rhs += av[u] * v.pb[b]; }
                else if (v.bKind[b] == 2) rhs += v.gs[b] * v.gb[b];
            }
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
        if (e_ != hipSuccess) { (void)hipGetLastError(); throw std::runtime_error(std::string("HIP: ") + hipGetErrorString(e_)); } \
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


// ---------------------------------------------------------------------------------------------------------------------
// Algebraic multigrid as the preconditioner of the same conjugate-gradient loop.
//
// The Jacobi-PCG above needs 772 / 1174 iterations at 2 M / 8 M cells; QHDFoam solves this equation every step
// [QHDpEqn.H L36-47], and in QHDFoam its matrix never changes (taubyrhof is fixed after start-up), so a hierarchy built once
// pays for itself at the first step.  Built on the host from the face coefficients, two kinds:
//   smoothed aggregation (default, see smoothedLevel below): root-and-neighbours aggregates of the strength graph, the prolongator
//     smoothed by one Jacobi step, Galerkin coarse operators, the last level (<= 2048 rows, QGD_MG_DENSE_MAX) solved exactly with its dense inverse;
//     8 M cells: levels of 8 M / 1 M / 55 k / 1.7 k rows, 6 CG iterations to QHDFoam's 1e-8;
//   plain aggregation (QGD_MG_SA=0, what round 2 shipped): pairwise matching along the strongest connection, two passes per level
//     (aggregates of ~4 cells), piecewise-constant prolongation with an over-weighted correction (x += 1.8 P e_c); 24 iterations.
// One V-cycle = nu damped-Jacobi sweeps before and after the coarse-grid correction; with equal pre- and post-smoothing the cycle is a
// symmetric positive definite operator, as CG needs.  Every sum is a gather in a fixed order: a solve is reproducible bit for bit,
// like the Jacobi variant.
// ---------------------------------------------------------------------------------------------------------------------
namespace {

template <typename T>
struct MgLevelT {
    int n = 0, width = 0;       // rows; the widest row
    int xrun = 0;               // row blocks per XCD run (xcdRunBlock), 0: plain order
    long long entries = 0;      // stored (padded) entries of the sliced ELL
    T* diag = nullptr;          // n
    // sliced ELL: 64 rows per slice, slice s holds sliceStart[s+1] - sliceStart[s] entry rows of 64 lanes; entry k of row i sits at
    // (sliceStart[i >> 6] + k) * 64 + (i & 63).  The rows of an aggregated level differ a lot in length (4-cell aggregates of
    // hexahedra: 6 to ~20 neighbours): padding every row to the widest made level 1 cost two thirds of level 0 with a quarter of the rows.
    const int* sliceStart = nullptr;   // n/64 + 2
    // or (rowStart != nullptr) plain CSR walked by one wavefront per row: the small levels of a smoothed-aggregation hierarchy have
    // 50-100 entries per row and too few rows to fill the chip with one lane per row (55 k rows x 88: 52 us per sweep, 11 us this way)
    const int* rowStart = nullptr;     // n + 1
    int* col = nullptr;         // -1 = padding (sliced ELL only)
    T* val = nullptr;           // a_ij (A_ij = -a_ij)
    const T* inverse = nullptr; // coarsest level: dense n x n inverse (row-major), or nullptr = Jacobi sweeps
    int* agg = nullptr;         // n: aggregate of each node in the next level
    int* aggStart = nullptr;    // nNext+1
    int* aggItems = nullptr;    // n
    // smoothed aggregation (pS != nullptr): the prolongator P (n rows, sliced ELL like the matrix: 4-10 entries per row) and its
    // transpose (nNext rows of 50-500 entries: CSR, one wavefront per row, ptS = row starts; above 300 k rows sliced ELL, ptSliced = 1)
    const int *pS = nullptr, *pCol = nullptr, *ptS = nullptr, *ptCol = nullptr;
    int ptSliced = 0;
    const T *pVal = nullptr, *ptVal = nullptr;
    T *x = nullptr, *x2 = nullptr, *b = nullptr, *r = nullptr;
};
using MgLevelDev = MgLevelT<double>;   // the cycle in double; MgLevelT<float>: the same cycle as a single-precision preconditioner

// (the order of the row blocks over the XCDs: xcdRunBlock in qgd_device.hpp.  Measured, profiles/r03_ab_row_xcd_run.txt: the double-precision
// matrix product of the implicit branch gains 3-4 % with runs of 16-64, the single-precision sweeps here lose 1 % -- their extra fetches
// are Infinity-Cache hits -- so this solver's default is the plain order, QGD_ROW_XCD_RUN = 0)
// s (+/-)= sum_k val[k] x[col[k]] over the w entries of one sliced-ELL row, in entry order.  The labels and coefficients of eight
// entries are requested before the first gather goes out: the plain loop (label, wait, gather, wait, per entry) left the level-0
// sweep latency-bound at 3.7 TB/s.  Padding (col < 0) adds 0 * 0, which changes no bit of s; w is uniform over the wavefront.
template <int SIGN, typename T>
__device__ __forceinline__ T ellRowAcc(T s, const int* __restrict__ col, const T* __restrict__ val, const T* __restrict__ x, const size_t e0,
                                       const int w) {
    constexpr int U = 8;
    for (int k0 = 0; k0 < w; k0 += U) {
        int c[U];
        T v[U], xv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool in = k0 + u < w;
            c[u] = in ? col[e0 + (size_t)(k0 + u) * 64] : -1;
            v[u] = in ? val[e0 + (size_t)(k0 + u) * 64] : (T)0;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) xv[u] = c[u] >= 0 ? x[c[u]] : (T)0;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (c[u] < 0) v[u] = (T)0;
            s = SIGN > 0 ? s + v[u] * xv[u] : s - v[u] * xv[u];
        }
    }
    return s;
}
// xout = xin + omega (b - A xin)/diag   (xin == nullptr: from zero, xout = omega b/diag);  rout (optional) = b - A xin
template <typename T>
__global__ __launch_bounds__(PB) void mgSmoothKernel(const MgLevelT<T> L, const T omega, const T* __restrict__ b,
                                                     const T* __restrict__ xin, T* xout, T* __restrict__ rout,
                                                     const double* __restrict__ ctl = nullptr, const T cx = 1, const T cm = 0) {
    const int i = xcdRunBlock(L.xrun) * PB + threadIdx.x;
    if (i >= L.n || solveDone(ctl)) return;
    const T d = L.diag[i];
    if (!xin) { xout[i] = omega * b[i] / d; return; }
    const T xi = xin[i];
    T s = d * xi;
    const int s0 = L.sliceStart[i >> 6], w = L.sliceStart[(i >> 6) + 1] - s0;
    const size_t e0 = (size_t)s0 * 64 + (i & 63);
    s = ellRowAcc<-1>(s, L.col, L.val, xin, e0, w);
    const T r = b[i] - s;
    if (rout) rout[i] = r;
    if (xout) {
        // cx = 1, cm = 0: a damped-Jacobi sweep;  otherwise one step of the Chebyshev recurrence, whose previous iterate sits in xout[i]
        T v = cx * xi + omega * r / d;
        if (cm != (T)0) v -= cm * xout[i];
        xout[i] = v;
    }
}
// The LAST post-smoothing sweep of level 0 in a single-precision cycle, fused with what used to follow it: the iterate goes out in double
// (z of the CG) and the block's share of r.z is summed on the way -- the copy into the level's x, the conversion pass and the dot-product
// pass (read 4 + write 8, read 8 + 8 bytes per row) are gone; r costs 8 bytes per row here.  Same sweep arithmetic as mgSmoothKernel,
// same block partials as dotKernel (block blk of PB rows -> part[blk]): not a bit of the solve changes.
__global__ __launch_bounds__(PB) void mgSmoothLastKernel(const MgLevelT<float> L, const float omega, const float* __restrict__ b,
                                                         const float* __restrict__ xin, const float* __restrict__ xprev, double* __restrict__ z,
                                                         const double* __restrict__ r, double* __restrict__ part,
                                                         const double* __restrict__ ctl, const float cx, const float cm) {
    const int blk = xcdRunBlock(L.xrun);
    const int i = blk * PB + threadIdx.x;
    if (solveDone(ctl)) return;
    double rz = 0;
    if (i < L.n) {
        const float d = L.diag[i];
        const float xi = xin[i];
        float s = d * xi;
        const int s0 = L.sliceStart[i >> 6], w = L.sliceStart[(i >> 6) + 1] - s0;
        const size_t e0 = (size_t)s0 * 64 + (i & 63);
        s = ellRowAcc<-1>(s, L.col, L.val, xin, e0, w);
        const float res = b[i] - s;
        float v = cx * xi + omega * res / d;
        if (cm != 0.0f) v -= cm * xprev[i];
        const double zi = (double)v;
        z[i] = zi;
        rz = r[i] * zi;
    }
    const double t = blockSum(rz);
    if (threadIdx.x == 0) part[blk] = t;
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
    const int blk = xcdRunBlock(L.xrun);   // the partial sums stay in block order: the fold adds them as before
    const int i = blk * PB + threadIdx.x;
    if (solveDone(ctl)) return;
    double xy = 0;
    if (i < L.n) {
        const double xi = x[i];
        double s = L.diag[i] * xi;
        const int s0 = L.sliceStart[i >> 6], w = L.sliceStart[(i >> 6) + 1] - s0;
        const size_t e0 = (size_t)s0 * 64 + (i & 63);
        s = ellRowAcc<-1>(s, L.col, L.val, x, e0, w);
        y[i] = s;
        xy = xi * s;
    }
    const double t = blockSum(xy);
    if (threadIdx.x == 0) part[blk] = t;
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
// sum over the 64 lanes of a wavefront, the same tree for every row (valid in lane 0)
template <typename T>
__device__ __forceinline__ T waveSum(T v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
// the sweep of mgSmoothKernel for a CSR level, one wavefront per row (PB / 64 rows per workgroup)
template <typename T>
__global__ __launch_bounds__(PB) void mgSmoothRowKernel(const MgLevelT<T> L, const T omega, const T* __restrict__ b, const T* __restrict__ xin,
                                                        T* xout, T* __restrict__ rout, const double* __restrict__ ctl = nullptr, const T cx = 1,
                                                        const T cm = 0) {
    const int i = blockIdx.x * (PB / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= L.n || solveDone(ctl)) return;
    if (!xin) { if (lane == 0) xout[i] = omega * b[i] / L.diag[i]; return; }
    T s = 0;
    for (int k = L.rowStart[i] + lane; k < L.rowStart[i + 1]; k += 64) s += L.val[k] * xin[L.col[k]];
    s = waveSum(s);
    if (lane != 0) return;
    const T d = L.diag[i], xi = xin[i];
    const T r = b[i] - (d * xi - s);
    if (rout) rout[i] = r;
    if (xout) {
        T v = cx * xi + omega * r / d;
        if (cm != (T)0) v -= cm * xout[i];
        xout[i] = v;
    }
}
// the same two transfers with a smoothed prolongator: rc = P^T r (one wavefront per coarse row), x += oc P ec
template <typename T>
__global__ __launch_bounds__(PB) void mgRestrictRowKernel(const int nCoarse, const int* __restrict__ rowStart, const int* __restrict__ col,
                                                          const T* __restrict__ val, const T* __restrict__ r, T* __restrict__ rc,
                                                          const double* __restrict__ ctl = nullptr) {
    const int I = blockIdx.x * (PB / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (I >= nCoarse || solveDone(ctl)) return;
    T s = 0;
    for (int k = rowStart[I] + lane; k < rowStart[I + 1]; k += 64) s += val[k] * r[col[k]];
    s = waveSum(s);
    if (lane == 0) rc[I] = s;
}
// rc = P^T r with P^T as sliced ELL, one coarse row per lane: level 0 -> 1 has 1 M rows of ~50 entries (105 us; 240 us by wavefronts)
template <typename T>
__global__ __launch_bounds__(PB) void mgRestrictEllKernel(const int nCoarse, const int* __restrict__ sliceStart, const int* __restrict__ col,
                                                          const T* __restrict__ val, const T* __restrict__ r, T* __restrict__ rc,
                                                          const double* __restrict__ ctl = nullptr) {
    const int I = blockIdx.x * PB + threadIdx.x;
    if (I >= nCoarse || solveDone(ctl)) return;
    const int s0 = sliceStart[I >> 6], w = sliceStart[(I >> 6) + 1] - s0;
    const size_t e0 = (size_t)s0 * 64 + (I & 63);
    T s = 0;
    s = ellRowAcc<1>(s, col, val, r, e0, w);
    rc[I] = s;
}
// ---- the distributed level 0 of a sharded solve (DistMg below): rows = the owned cells [ob, ob + L.n), every vector indexed by the
// LOCAL cell label (so that the ghost columns of a row read the ghost entries a halo exchange has just refreshed) ----
template <typename T>
__global__ __launch_bounds__(PB) void mgSmoothOwnedKernel(const MgLevelT<T> L, const int ob, const T omega, const T* __restrict__ b,
                                                          const T* __restrict__ xin, T* xout, T* __restrict__ rout,
                                                          const double* __restrict__ ctl, const T cx, const T cm) {
    const int r = blockIdx.x * PB + threadIdx.x;
    if (r >= L.n || solveDone(ctl)) return;
    const int i = r + ob;
    const T d = L.diag[i];
    if (!xin) { xout[i] = omega * b[i] / d; return; }
    const T xi = xin[i];
    T s = d * xi;
    const int s0 = L.sliceStart[r >> 6], w = L.sliceStart[(r >> 6) + 1] - s0;
    const size_t e0 = (size_t)s0 * 64 + (r & 63);
    s = ellRowAcc<-1>(s, L.col, L.val, xin, e0, w);
    const T res = b[i] - s;
    if (rout) rout[i] = res;
    if (xout) {
        T v = cx * xi + omega * res / d;
        if (cm != (T)0) v -= cm * xout[i];
        xout[i] = v;
    }
}
// this rank's share of P^T r for the coarse nodes its cells touch (one wavefront per touched node), as doubles for the all-reduce
template <typename T>
__global__ __launch_bounds__(PB) void mgRestrictPartialKernel(const int nRows, const int* __restrict__ rowNode, const int* __restrict__ rowStart,
                                                              const int* __restrict__ col, const T* __restrict__ val, const T* __restrict__ r,
                                                              double* __restrict__ out, const double* __restrict__ ctl) {
    const int k = blockIdx.x * (PB / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (k >= nRows || solveDone(ctl)) return;
    T s = 0;
    for (int e = rowStart[k] + lane; e < rowStart[k + 1]; e += 64) s += val[e] * r[col[e]];
    s = waveSum(s);
    if (lane == 0) out[rowNode[k]] = (double)s;
}
// coarsest level: x = A^-1 b with the dense inverse, one wavefront per row
template <typename T>
__global__ __launch_bounds__(PB) void mgDenseKernel(const int n, const T* __restrict__ inverse, const T* __restrict__ b, T* __restrict__ x,
                                                    const double* __restrict__ ctl = nullptr) {
    const int i = blockIdx.x * (PB / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= n || solveDone(ctl)) return;
    T s = 0;
    for (int k = lane; k < n; k += 64) s += inverse[(size_t)i * n + k] * b[k];
    s = waveSum(s);
    if (lane == 0) x[i] = s;
}
template <typename T>
__global__ __launch_bounds__(PB) void mgProlongEllKernel(const int n, const int* __restrict__ sliceStart, const int* __restrict__ col,
                                                         const T* __restrict__ val, const T oc, const T* __restrict__ ec, T* __restrict__ x,
                                                         const double* __restrict__ ctl = nullptr) {
    const int i = blockIdx.x * PB + threadIdx.x;
    if (i >= n || solveDone(ctl)) return;
    const int s0 = sliceStart[i >> 6], w = sliceStart[(i >> 6) + 1] - s0;
    const size_t e0 = (size_t)s0 * 64 + (i & 63);
    T s = 0;
    s = ellRowAcc<1>(s, col, val, ec, e0, w);
    x[i] += oc * s;
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
    const int s0 = on ? L.sliceStart[i >> 6] : 0, wRow = on ? L.sliceStart[(i >> 6) + 1] - s0 : 0;
    const size_t e0 = (size_t)s0 * 64 + (i & 63);
#pragma unroll
    for (int k = 0; k < MG_COARSE_ROW; ++k) {
        const bool has = on && k < wRow;
        cc[k] = has ? L.col[e0 + (size_t)k * 64] : -1;
        vv[k] = (has && cc[k] >= 0) ? L.val[e0 + (size_t)k * 64] : (T)0;
        if (cc[k] < 0) cc[k] = i;   // value 0: reads its own entry
    }
    if (on) cur[i] = omega * bi / d;
    __syncthreads();
    for (int s = 1; s < sweeps; ++s) {
        if (on) {
            T t = d * cur[i];
#pragma unroll
            for (int k = 0; k < MG_COARSE_ROW; ++k) t -= vv[k] * cur[cc[k]];
            for (int k = MG_COARSE_ROW; k < wRow; ++k) {
                const int c = L.col[e0 + (size_t)k * 64];
                if (c >= 0) t -= L.val[e0 + (size_t)k * 64] * cur[c];
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
// b0 != nullptr: the head of the single-precision cycle rides along -- its right-hand side b0 = (float) r and its first pre-smoothing step
// from the zero iterate x0 = om0 b0 / diag0 (what mgConvertKernel + the first mgSmoothKernel launch of the cycle did in two more passes)
__global__ __launch_bounds__(PB) void axpyKernel(const int n, double* __restrict__ x, double* __restrict__ r,
                                                  const double* __restrict__ d, const double* __restrict__ q, double* __restrict__ part,
                                                  const double* __restrict__ ctl, float* __restrict__ b0 = nullptr, float* __restrict__ x0 = nullptr,
                                                  const float* __restrict__ diag0 = nullptr, const float om0 = 0) {
    if (solveDone(ctl)) return;
    const double alpha = ctl[C_ALPHA];
    const int c = blockIdx.x * PB + threadIdx.x;
    double ar = 0;
    if (c < n) {
        x[c] += alpha * d[c]; const double rc = r[c] - alpha * q[c]; r[c] = rc; ar = fabs(rc);
        if (b0) { const float bf = (float)rc; b0[c] = bf; x0[c] = om0 * bf / diag0[c]; }
    }
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
// folds `count` rows of partials into ctl[first ...] (one workgroup per row, ascending order, eight chains per thread)
__global__ __launch_bounds__(PB) void foldCtlKernel(const double* __restrict__ part, const int nBlocks, const int count, double* __restrict__ ctl,
                                                     const int first, const int always) {
    if (!always && solveDone(ctl)) return;
    const int k = blockIdx.x;
    const double* __restrict__ p = part + (size_t)k * nBlocks;
    double v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int i = threadIdx.x;
    for (; i + 7 * PB < nBlocks; i += 8 * PB) {
        double x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = p[i + j * PB];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += x[j];
    }
    for (; i < nBlocks; i += PB) v[0] += p[i];
    const double t = blockSum(((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7])));
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
    } else if (stage == 4) {
        // unsharded solves only: the convergence test BEFORE the preconditioner is applied to the new residual (the sums are local, so
        // the early test costs one fold; a sharded solve keeps |r| and r.z in one all-reduce and pays one cycle at the last iteration)
        if (ctl[C_DONE] == 0.0) {
            const double res = ctl[C_ABSR2] / ctl[C_NORMF];
            const double it = ctl[C_ITER] + 1.0;
            if (res < tol || (relTol > 0 && res < relTol * ctl[C_RES0]) || it >= (double)maxIter) { ctl[C_RES] = res; ctl[C_ITER] = it; ctl[C_DONE] = 1.0; }
        }
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
    int nu0 = 0;                           // QGD_MG_NU0: sweeps before / after on level 0 only (0: as nu) -- the level whose passes cost
    // the smoother's steps: x <- (1 + cm) x + cr D^-1 r - cm x_previous.  Damped Jacobi: cr = omega, cm = 0.  With QGD_MG_CHEB=ratio > 1
    // the nu steps are the Chebyshev polynomial of D^-1 A for the eigenvalue interval [lmax / ratio, lmax] (lmax = 2: Gershgorin, the
    // rows of every level are weakly diagonally dominant) -- a fixed polynomial, the same before and after the coarse correction, so
    // the cycle stays a symmetric preconditioner
    double cr[8] = {0, 0, 0, 0, 0, 0, 0, 0}, cm[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    void smootherSetup(const double ratio, const double lmax) {
        for (int k = 0; k < 8; ++k) { cr[k] = omega; cm[k] = 0; }
        if (!(ratio > 1)) return;
        const double hi = lmax, lo = lmax / ratio, theta = 0.5 * (hi + lo), delta = 0.5 * (hi - lo), sigma = theta / delta;
        double rho = 1.0 / sigma;
        cr[0] = 1.0 / theta;
        for (int k = 1; k < 8; ++k) {
            const double rhoNew = 1.0 / (2.0 * sigma - rho);
            cm[k] = rhoNew * rho; cr[k] = 2.0 * rhoNew / delta;
            rho = rhoNew;
        }
    }
    std::vector<double> smootherScale;   // per level (smoothed aggregation: coarse operators are no M-matrices, lambda_max is estimated)
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
    bool earlyTest = false;   // pressureSolveRun without an all-reduce hook: test convergence before the last cycle instead of after it
    // knobs of the hierarchy builder (read once at creation; the distributed hierarchy is built later, at the first solve)
    bool sa = true;
    int passes = 2;
    double saTheta = 0.08;
    int denseMax = 2048;                   // QGD_MG_DENSE_MAX
    int64_t distMaxCells = 20000000;       // QGD_MG_DIST_MAX_CELLS: above it a sharded solve keeps the rank-local hierarchy (distSetupStep)
    int rowRun = 0;                        // QGD_ROW_XCD_RUN: row blocks per XCD run of the sliced-ELL sweeps (xcdRunBlock), 0: plain order (measured: runs of 16 make the f32 sweeps 1 % slower)
    // ---- a hierarchy that SPANS THE RANKS of a sharded solve (QGD_MG_DIST, default on; 0: the rank-local block hierarchy) ----------------
    // A rank-local hierarchy is block Jacobi: 7 -> 65 / 97 / 140 iterations on 2 / 4 / 8 shards of a 128^3 box.  Here level 0 stays
    // distributed (each rank smooths its own rows, the ghost entries of the iterate refreshed before every sweep), every level below it
    // is REPLICATED: all ranks hold the same global coarse hierarchy, all-reduce their shares of the level-1 right-hand side and run
    // the coarse part of the cycle redundantly (1/8 of the rows and less: cheaper than exchanging per level).  Set-up: the global
    // matrix is gathered by two all-reduces of a zero-padded buffer (sizes: MAX; diagonal and, per global cell, the couplings to its
    // higher-numbered neighbours: SUM), then every rank builds the same hierarchy.  The comm points are handed to the caller one at a
    // time: `pending` says what to do with `buf` / `haloVec` before pressureSolveContinue.
    struct Dist {
        bool wanted = false, built = false;
        int pending = 0;            // 0 nothing, 1 halo of haloVec (one value per cell), 2 SUM all-reduce of buf[0, bufN), 3 MAX all-reduce
        int setupStage = 0, pc = -1, resumePhase = -1, sweep = 0;
        double* buf = nullptr;
        int64_t bufN = 0, bufCap = 0;
        float* haloVec = nullptr;
        std::vector<int> own, nei;             // the local mesh's internal faces (host copies for the gather and level 0)
        std::vector<double> a, diag;
        std::vector<int32_t> cellGlobal;       // extracted shards; empty: global = local + cellGlobalOffset
        int64_t cellGlobalOffset = 0, nCg = 0;
        int K = 0, n1 = 0;
        // QGD_MG_DIST=2: level 0 is coarsened PER RANK (aggregates, prolongator and the rank's rows of the level-1 matrix from its own cells;
        // cells next to a cut keep their tentative prolongator row, so that no coarse row needs another rank's fine rows), only the level-1
        // matrix is gathered and replicated -- distSetupStep stages 10..15
        int mode = 1;
        struct Local0 {
            std::vector<int64_t> off; std::vector<int> nb; std::vector<double> nw; std::vector<uint8_t> strong, cutAdj;
            std::vector<double> diag;
            std::vector<int> cutI, cutJ; std::vector<double> cutW;      // cut faces: owned row (0-based), ghost cell (local label), coefficient
            std::vector<int> agg, gid, ghostGid;                        // aggregate of each owned cell; its dense global number; the same of the ghost cells
            int na = 0;
            std::vector<float> tmp;
            std::vector<int> e1I, e1J; std::vector<double> e1W, d1;     // this rank's share of the level-1 matrix (global numbers, lower number first)
            std::vector<int64_t> pOff; std::vector<int> pCol; std::vector<double> pVal;   // the owned cells' rows of P (columns: global level-1 numbers)
        };
        std::shared_ptr<Local0> l0;
        int nPt = 0;                           // coarse nodes this rank's cells touch, their rows of P^T (CSR over owned cells, local labels)
        int *ptNode = nullptr, *ptStart = nullptr, *ptCol = nullptr;
        float* ptVal = nullptr;
        float *cur = nullptr, *nxt = nullptr;
        const double* rhs = nullptr;           // r -> z of the application in flight
        double* out = nullptr;
        int64_t globalOf(int i) const { return cellGlobal.empty() ? (int64_t)i + cellGlobalOffset : (int64_t)cellGlobal[i]; }
    } dist;
    void distEnsureBuf(int64_t n) {
        if (n <= dist.bufCap) return;
        dist.buf = alloc<double>((size_t)n);      // the old one stays in `owned` until the solver goes (set-up only grows it twice)
        dist.bufCap = n;
    }
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
        else {
            // hipMemset runs on the NULL stream and returns before it is done; the solver's stream is non-blocking, so a copy or kernel
            // queued there right after could be overtaken by the zero-fill (seen: the gather buffers of DistMg zeroed AFTER they were filled)
            PCHECK(hipMemset(p, 0, nb));
            PCHECK(hipStreamSynchronize(nullptr));
        }
        return (T*)p;
    }
    ~PressureSolver() {
        if (cycleGraph) (void)hipGraphExecDestroy(cycleGraph);
        for (void* p : owned) (void)hipFree(p);
        if (hostCtl) (void)hipHostFree(hostCtl);
        for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e);
    }

    // z = M r on level l (b -> x), in the precision of the level arrays; every launch returns at once when the solve is done
    // Fused hand-over between the CG and the single-precision cycle (unsharded solves; QGD_MG_FUSE=0: the separate passes).  headDone:
    // b and the first pre-smoothing step of level 0 were written by axpyKernel; tailZ/tailR/tailPart: the last post-smoothing sweep of
    // level 0 writes z in double and the block partials of r.z (mgSmoothLastKernel)
    bool fuse = false, headDone = false;
    double* tailZ = nullptr; const double* tailR = nullptr; double* tailPart = nullptr;
    bool fusedCycle() const { return fuse && precond == 1 && Lf.size() >= 2 && !Lf[0].rowStart && nu0 >= 1 && !dist.wanted; }
    template <typename T>
    void vcycleT(std::vector<MgLevelT<T>>& Lv, size_t l, const T* b, T* x) {
        MgLevelT<T>& lv = Lv[l];
        const int nb = blocksOf(lv.n);
        const double sc = l < smootherScale.size() ? smootherScale[l] : 1.0;    // 2 / lambda_max(D^-1 A) of this level (1: the Gershgorin bound)
        const T om = (T)(omega * sc), over = (T)oc;
        const T* none = nullptr;
        T* noOut = nullptr;
        const int nbRow = (lv.n + PB / 64 - 1) / (PB / 64);
        // one sweep of the level in its own layout
        auto sweep = [&](T w, const T* rhs, const T* xin, T* xout, T* rout, T cx, T cmPrev) {
            if (lv.rowStart) mgSmoothRowKernel<T><<<nbRow, PB, 0, stream>>>(lv, w, rhs, xin, xout, rout, ctl, cx, cmPrev);
            else mgSmoothKernel<T><<<nb, PB, 0, stream>>>(lv, w, rhs, xin, xout, rout, ctl, cx, cmPrev);
        };
        if (l + 1 == Lv.size()) {
            if (lv.inverse) mgDenseKernel<T><<<nbRow, PB, 0, stream>>>(lv.n, lv.inverse, b, x, ctl);
            else if (lv.n <= MG_COARSE_MAX && !lv.rowStart) mgCoarseKernel<T><<<1, 1024, 0, stream>>>(lv, om, coarseSweeps, b, x, ctl);
            else {
                T* cur = x; T* nxt = lv.x2;
                sweep(om, b, none, cur, noOut, (T)1, (T)0);
                for (int s = 1; s < coarseSweeps; ++s) { sweep(om, b, cur, nxt, noOut, (T)1, (T)0); std::swap(cur, nxt); }
                if (cur != x) PCHECK(hipMemcpyAsync(x, cur, sizeof(T) * lv.n, hipMemcpyDeviceToDevice, stream));
            }
            return;
        }
        MgLevelT<T>& nx = Lv[l + 1];
        T* cur = x; T* nxt = lv.x2;
        const int nu = l == 0 ? nu0 : this->nu;   // level 0 may take fewer sweeps than the cheap coarse levels (QGD_MG_NU0)
        // pre-smoothing from a zero iterate: step 0 is cr[0] b/d, step 1 has no previous iterate to subtract
        if (!(l == 0 && headDone)) sweep((T)(cr[0] * sc), b, none, cur, noOut, (T)1, (T)0);
        for (int s = 1; s < nu; ++s) {
            sweep((T)(cr[s] * sc), b, cur, nxt, noOut, (T)(1.0 + cm[s]), (T)(s >= 2 ? cm[s] : 0.0));
            std::swap(cur, nxt);
        }
        sweep(om, b, cur, noOut, lv.r, (T)1, (T)0);           // r = b - A x
        if (lv.pS && lv.ptSliced) mgRestrictEllKernel<T><<<blocksOf(nx.n), PB, 0, stream>>>(nx.n, lv.ptS, lv.ptCol, lv.ptVal, lv.r, nx.b, ctl);
        else if (lv.pS) mgRestrictRowKernel<T><<<(nx.n + PB / 64 - 1) / (PB / 64), PB, 0, stream>>>(nx.n, lv.ptS, lv.ptCol, lv.ptVal, lv.r, nx.b, ctl);
        else mgRestrictKernel<T><<<blocksOf(nx.n), PB, 0, stream>>>(nx.n, lv.aggStart, lv.aggItems, lv.r, nx.b, ctl);
        vcycleT<T>(Lv, l + 1, nx.b, nx.x);
        if (lv.pS) mgProlongEllKernel<T><<<nb, PB, 0, stream>>>(lv.n, lv.pS, lv.pCol, lv.pVal, over, nx.x, cur, ctl);
        else mgProlongKernel<T><<<nb, PB, 0, stream>>>(lv.n, lv.agg, over, nx.x, cur, ctl);
        for (int s = 0; s < nu; ++s) {
            if constexpr (std::is_same<T, float>::value) {
                if (l == 0 && tailZ && s == nu - 1 && !lv.rowStart) {
                    mgSmoothLastKernel<<<nb, PB, 0, stream>>>(lv, (float)(cr[s] * sc), b, cur, nxt, tailZ, tailR, tailPart, ctl, (float)(1.0 + cm[s]), (float)cm[s]);
                    return;
                }
            }
            sweep((T)(cr[s] * sc), b, cur, nxt, noOut, (T)(1.0 + cm[s]), (T)cm[s]);
            std::swap(cur, nxt);
        }
        if (cur != x) PCHECK(hipMemcpyAsync(x, cur, sizeof(T) * lv.n, hipMemcpyDeviceToDevice, stream));
    }
    void vcycle(size_t l, const double* b, double* x) {
        if (!Lf.empty() && l == 0) {
            // the cycle as a single-precision operator between double-precision CG vectors: half the bytes of every sweep
            const int nb = blocksOf(L[0].n);
            if (!headDone) mgConvertKernel<double, float><<<nb, PB, 0, stream>>>(L[0].n, b, Lf[0].b, ctl);
            vcycleT<float>(Lf, 0, Lf[0].b, Lf[0].x);
            if (!tailZ) mgConvertKernel<float, double><<<nb, PB, 0, stream>>>(L[0].n, Lf[0].x, x, ctl);
        } else vcycleT<double>(L, l, b, x);
    }
    // z = M r over the owned rows (the multigrid hierarchy is built on the owned block: for a shard it is the additive-Schwarz
    // block of this rank, couplings to ghost cells stay in the diagonal only)
    // head: the caller's axpyKernel wrote the head of the cycle; rzPart != nullptr: the cycle's last sweep leaves the block partials of
    // r.z there (both only with `fuse`; returns whether the partials were written)
    bool precondition(bool head = false, double* rzPart = nullptr) {
        const int n = oe - ob, nb = blocksOf(n);
        const bool fused = fusedCycle();
        headDone = fused && head;
        tailZ = fused && rzPart ? z + ob : nullptr; tailR = r + ob; tailPart = rzPart;
        struct Reset { PressureSolver* s; ~Reset() { s->headDone = false; s->tailZ = nullptr; } } reset{this};
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
        return tailZ != nullptr;
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
        int best = -1; double bw = 0.0;   // only along a positive coupling (smoothed-aggregation levels have a few of the other sign)
        for (int64_t k = off[i]; k < off[i + 1]; ++k) if (agg[nb[k]] < 0 && nb[k] != i && nw[k] > bw) { bw = nw[k]; best = nb[k]; }
        if (best >= 0) { agg[i] = agg[best] = na++; }
    }
    for (int i = 0; i < n; ++i) {
        if (agg[i] >= 0) continue;
        int best = -1; double bw = -1e300;
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
// ---------------------------------------------------------------------------------------------------------------------
// Smoothed aggregation (QGD_MG_SA, the default).  With the piecewise-constant prolongator T of the aggregates the V-cycle contracts by
// ~0.6 per CG iteration (24 iterations for QHDFoam's 1e-8 at 8 M cells) whatever the smoother -- Chebyshev steps instead of Jacobi
// sweeps changed 24 to 23 -- because the coarse spaces cannot represent smooth corrections.  One Jacobi step on the prolongator,
//      P = (I - omegaP D^-1 A) T,     A_coarse = P^T A P,     omegaP = 4 / (3 lambda_max(D^-1 A)),
// fixes that at the price of wider coarse rows, so the aggregates are larger (three pairwise passes, ~8 cells).  Everything is built once
// on the host: P row by row, the triple product coarse row by coarse row (Gustavson with a marker array per thread; the order of
// every sum is fixed by the row orders, not by the thread count).
// ---------------------------------------------------------------------------------------------------------------------
struct HostCsr {
    std::vector<int64_t> off;
    std::vector<int> col;
    std::vector<double> val;
};
static void adjacencyOf(int n, const std::vector<int>& I, const std::vector<int>& J, const std::vector<double>& w, std::vector<int64_t>& off,
                        std::vector<int>& nb, std::vector<double>& nw) {
    const size_t E = I.size();
    off.assign((size_t)n + 1, 0);
    for (size_t e = 0; e < E; ++e) { off[I[e] + 1]++; off[J[e] + 1]++; }
    for (int i = 0; i < n; ++i) off[i + 1] += off[i];
    nb.resize((size_t)off[n]); nw.resize((size_t)off[n]);
    std::vector<int64_t> fill(off.begin(), off.end() - 1);
    for (size_t e = 0; e < E; ++e) {
        nb[fill[I[e]]] = J[e]; nw[fill[I[e]]++] = w[e];
        nb[fill[J[e]]] = I[e]; nw[fill[J[e]]++] = w[e];
    }
}
// largest eigenvalue of D^-1 A (A_ii = diag, A_ij = -w_ij) by power iteration from a fixed start vector, with a 10 % margin
static double lambdaMaxOf(int n, const std::vector<int64_t>& off, const std::vector<int>& nb, const std::vector<double>& nw,
                          const std::vector<double>& diag) {
    std::vector<double> x((size_t)n), y((size_t)n);
    uint32_t seed = 12345u;
    for (int i = 0; i < n; ++i) { seed = seed * 1664525u + 1013904223u; x[i] = ((seed >> 8) & 0xffff) / 65536.0 - 0.5; }
    double lambda = 0;
    for (int it = 0; it < 30; ++it) {
        double xx = 0, yy = 0;
#pragma omp parallel for schedule(static) reduction(+ : xx, yy) if (n > 20000)
        for (int i = 0; i < n; ++i) {
            double sum = diag[i] * x[i];
            for (int64_t k = off[i]; k < off[i + 1]; ++k) sum -= nw[k] * x[nb[k]];
            y[i] = sum / diag[i];
            xx += x[i] * x[i]; yy += y[i] * y[i];
        }
        lambda = std::max(lambda, std::sqrt(yy / std::max(xx, 1e-300)));
        const double inv = 1.0 / std::sqrt(std::max(yy, 1e-300));
        for (int i = 0; i < n; ++i) x[i] = y[i] * inv;
    }
    return 1.1 * lambda;
}
// strong couplings of the level: a_ij > theta sqrt(a_ii a_jj) (negative off-diagonal entries only)
static void strengthOf(int n, const std::vector<int64_t>& off, const std::vector<int>& nb, const std::vector<double>& nw,
                       const std::vector<double>& diag, double theta, std::vector<uint8_t>& strong) {
    strong.assign(nb.size(), 0);
#pragma omp parallel for schedule(static) if (n > 20000)
    for (int i = 0; i < n; ++i)
        for (int64_t k = off[i]; k < off[i + 1]; ++k) strong[k] = nw[k] > theta * std::sqrt(diag[i] * diag[nb[k]]) ? 1 : 0;
}
// Aggregates of the strength graph (Vanek, Mandel, Brezina 1996): 1. a node whose strong neighbours are all free becomes the root of
// {node} + neighbours;  2. the remaining nodes join the adjacent aggregate of pass 1 they are coupled to most strongly;  3. what is
// left (no aggregated strong neighbour) forms aggregates with its free neighbours, isolated nodes stay alone.  Ascending node order.
static int rootAggregates(int n, const std::vector<int64_t>& off, const std::vector<int>& nb, const std::vector<double>& nw,
                          const std::vector<uint8_t>& strong, std::vector<int>& agg) {
    agg.assign((size_t)n, -1);
    int na = 0;
    for (int i = 0; i < n; ++i) {
        if (agg[i] >= 0) continue;
        bool free = true, any = false;
        for (int64_t k = off[i]; k < off[i + 1] && free; ++k)
            if (strong[k]) { any = true; if (agg[nb[k]] >= 0) free = false; }
        if (!free || !any) continue;
        agg[i] = na;
        for (int64_t k = off[i]; k < off[i + 1]; ++k) if (strong[k]) agg[nb[k]] = na;
        ++na;
    }
    std::vector<int> first(agg);
    for (int i = 0; i < n; ++i) {
        if (first[i] >= 0) continue;
        int best = -1; double bw = -1e300;
        for (int64_t k = off[i]; k < off[i + 1]; ++k) if (strong[k] && first[nb[k]] >= 0 && nw[k] > bw) { bw = nw[k]; best = nb[k]; }
        if (best >= 0) agg[i] = first[best];
    }
    for (int i = 0; i < n; ++i) {
        if (agg[i] >= 0) continue;
        agg[i] = na;
        for (int64_t k = off[i]; k < off[i + 1]; ++k) if (strong[k] && agg[nb[k]] < 0) agg[nb[k]] = na;
        ++na;
    }
    return na;
}
// P = (I - omegaP DF^-1 AF) T with the filtered matrix AF (weak couplings lumped into the diagonal), A_coarse = P^T A P with the full A
static void smoothedLevel(int n, const std::vector<int64_t>& off, const std::vector<int>& nb, const std::vector<double>& nw,
                          const std::vector<uint8_t>& strong, const std::vector<double>& diag, const std::vector<int>& agg, int nc, double omegaP,
                          HostCsr& P, HostCsr& PT,
                          std::vector<int>& cI, std::vector<int>& cJ, std::vector<double>& cw, std::vector<double>& cdiag,
                          const std::vector<uint8_t>* noSmooth = nullptr) {
    // noSmooth[i] != 0: row i keeps its tentative entry (1 at its aggregate) -- the cells next to a cut of a per-rank coarsening (distSetupStep)
    // ---- P, row by row (entries in ascending column order) ----
    auto rowOfP = [&](int i, std::vector<int>& cols, std::vector<double>& vals) {
        cols.clear(); vals.clear();
        if (noSmooth && (*noSmooth)[i]) { cols.push_back(agg[i]); vals.push_back(1.0); return; }
        double dF = diag[i];
        for (int64_t k = off[i]; k < off[i + 1]; ++k) if (!strong[k]) dF -= nw[k];
        if (!(dF > 0.1 * diag[i])) dF = diag[i];
        cols.push_back(agg[i]); vals.push_back(1.0 - omegaP);
        const double s = omegaP / dF;
        for (int64_t k = off[i]; k < off[i + 1]; ++k) {
            if (!strong[k]) continue;
            const int a = agg[nb[k]];
            size_t q = 0;
            while (q < cols.size() && cols[q] != a) ++q;
            if (q == cols.size()) { cols.push_back(a); vals.push_back(0.0); }
            vals[q] += s * nw[k];
        }
        for (size_t a = 1; a < cols.size(); ++a)
            for (size_t b = a; b > 0 && cols[b - 1] > cols[b]; --b) { std::swap(cols[b - 1], cols[b]); std::swap(vals[b - 1], vals[b]); }
    };
    P.off.assign((size_t)n + 1, 0);
#pragma omp parallel
    {
        std::vector<int> cols; std::vector<double> vals;
#pragma omp for schedule(static)
        for (int i = 0; i < n; ++i) { rowOfP(i, cols, vals); P.off[i + 1] = (int64_t)cols.size(); }
    }
    for (int i = 0; i < n; ++i) P.off[i + 1] += P.off[i];
    P.col.resize((size_t)P.off[n]); P.val.resize((size_t)P.off[n]);
#pragma omp parallel
    {
        std::vector<int> cols; std::vector<double> vals;
#pragma omp for schedule(static)
        for (int i = 0; i < n; ++i) {
            rowOfP(i, cols, vals);
            for (size_t q = 0; q < cols.size(); ++q) { P.col[P.off[i] + q] = cols[q]; P.val[P.off[i] + q] = vals[q]; }
        }
    }
    // ---- P^T (rows in ascending fine index) ----
    PT.off.assign((size_t)nc + 1, 0);
    for (int c : P.col) PT.off[c + 1]++;
    for (int c = 0; c < nc; ++c) PT.off[c + 1] += PT.off[c];
    PT.col.resize(P.col.size()); PT.val.resize(P.col.size());
    {
        std::vector<int64_t> fill(PT.off.begin(), PT.off.end() - 1);
        for (int i = 0; i < n; ++i)
            for (int64_t q = P.off[i]; q < P.off[i + 1]; ++q) { const int64_t at = fill[P.col[q]]++; PT.col[at] = i; PT.val[at] = P.val[q]; }
    }
    // ---- A_coarse = P^T A P: row I gathers p_iI * A_ik * p_kJ over the fine rows i of P^T's row I ----
    cdiag.assign((size_t)nc, 0.0);
    int nThreads = 1;
#pragma omp parallel
    {
#pragma omp single
        nThreads = omp_get_num_threads();
    }
    std::vector<std::vector<int>> tI((size_t)nThreads), tJ((size_t)nThreads);
    std::vector<std::vector<double>> tw((size_t)nThreads);
#pragma omp parallel num_threads(nThreads)
    {
        const int t = omp_get_thread_num();
        const int lo = (int)((int64_t)nc * t / nThreads), hi = (int)((int64_t)nc * (t + 1) / nThreads);
        std::vector<int> where((size_t)nc, -1), touched;
        std::vector<double> acc;
        for (int Ic = lo; Ic < hi; ++Ic) {
            touched.clear(); acc.clear();
            auto add = [&](int k, double f) {
                for (int64_t q = P.off[k]; q < P.off[k + 1]; ++q) {
                    const int Jc = P.col[q];
                    int at = where[Jc];
                    if (at < 0) { at = where[Jc] = (int)touched.size(); touched.push_back(Jc); acc.push_back(0.0); }
                    acc[at] += f * P.val[q];
                }
            };
            for (int64_t e = PT.off[Ic]; e < PT.off[Ic + 1]; ++e) {
                const int i = PT.col[e];
                const double pi = PT.val[e];
                add(i, pi * diag[i]);
                for (int64_t k = off[i]; k < off[i + 1]; ++k) add(nb[k], -pi * nw[k]);
            }
            const double dI = where[Ic] >= 0 ? acc[where[Ic]] : 0.0;
            cdiag[Ic] = dI;
            std::vector<int> order(touched);
            std::sort(order.begin(), order.end());
            for (int Jc : order) {
                const double v = acc[where[Jc]];
                if (Jc > Ic && std::fabs(v) > 1e-13 * std::fabs(dI)) { tI[t].push_back(Ic); tJ[t].push_back(Jc); tw[t].push_back(-v); }
            }
            for (int Jc : touched) where[Jc] = -1;
        }
    }
    cI.clear(); cJ.clear(); cw.clear();
    for (int t = 0; t < nThreads; ++t) {
        cI.insert(cI.end(), tI[t].begin(), tI[t].end());
        cJ.insert(cJ.end(), tJ[t].begin(), tJ[t].end());
        cw.insert(cw.end(), tw[t].begin(), tw[t].end());
    }
}
}  // namespace

// dense inverse of a small symmetric positive definite level (Cholesky in double); false when a pivot collapses (a singular block)
static bool denseInverse(int n, const std::vector<int>& I, const std::vector<int>& J, const std::vector<double>& w, const std::vector<double>& diag,
                         std::vector<double>& inv) {
    std::vector<double> A((size_t)n * n, 0.0);
    double dmax = 0;
    for (int i = 0; i < n; ++i) { A[(size_t)i * n + i] = diag[i]; dmax = std::max(dmax, diag[i]); }
    for (size_t e = 0; e < I.size(); ++e) { A[(size_t)I[e] * n + J[e]] -= w[e]; A[(size_t)J[e] * n + I[e]] -= w[e]; }
    // A = L L^T, L in the lower triangle
    for (int j = 0; j < n; ++j) {
        double d = A[(size_t)j * n + j];
        for (int k = 0; k < j; ++k) d -= A[(size_t)j * n + k] * A[(size_t)j * n + k];
        if (!(d > 1e-12 * dmax)) return false;
        const double ljj = std::sqrt(d);
        A[(size_t)j * n + j] = ljj;
#pragma omp parallel for schedule(static) if (n - j > 256)
        for (int i = j + 1; i < n; ++i) {
            double v = A[(size_t)i * n + j];
            for (int k = 0; k < j; ++k) v -= A[(size_t)i * n + k] * A[(size_t)j * n + k];
            A[(size_t)i * n + j] = v / ljj;
        }
    }
    inv.assign((size_t)n * n, 0.0);
#pragma omp parallel for schedule(dynamic, 8)
    for (int c = 0; c < n; ++c) {
        std::vector<double> y((size_t)n, 0.0);
        for (int i = c; i < n; ++i) {            // L y = e_c
            double v = i == c ? 1.0 : 0.0;
            for (int k = c; k < i; ++k) v -= A[(size_t)i * n + k] * y[k];
            y[i] = v / A[(size_t)i * n + i];
        }
        for (int i = n - 1; i >= 0; --i) {       // L^T x = y
            double v = y[i];
            for (int k = i + 1; k < n; ++k) v -= A[(size_t)k * n + i] * y[k];
            y[i] = v / A[(size_t)i * n + i];
        }
        for (int i = 0; i < n; ++i) inv[(size_t)i * n + c] = y[i];
    }
    return true;
}
#define MG_DENSE_MAX 2048   // a level of at most this many rows is the last one and is solved exactly (default of QGD_MG_DENSE_MAX)
static void mgUploadLevel(PressureSolver* S, int n, const std::vector<int>& I, const std::vector<int>& J, const std::vector<double>& w,
                          const std::vector<double>& diag, bool last) {
    std::vector<int> deg((size_t)n, 0);
    for (size_t e = 0; e < I.size(); ++e) { deg[I[e]]++; deg[J[e]]++; }
    int width = 0;
    for (int d : deg) width = std::max(width, d);
    // layout: one lane per row (sliced ELL) for the large levels, one wavefront per row (CSR) for small levels with long rows
    const bool rows = !S->L.empty() && n <= 300000 && 2.0 * (double)I.size() >= 24.0 * n;
    std::vector<int> start, col;
    std::vector<double> val;
    long long stored = 0;
    if (rows) {
        start.assign((size_t)n + 1, 0);
        for (int i = 0; i < n; ++i) start[i + 1] = start[i] + deg[i];
        stored = start[n];
        col.assign(std::max<size_t>((size_t)stored, 1), 0); val.assign(std::max<size_t>((size_t)stored, 1), 0.0);
        std::vector<int> fill(start.begin(), start.end() - 1);
        for (size_t e = 0; e < I.size(); ++e) {
            const int i = I[e], j = J[e];
            col[fill[i]] = j; val[fill[i]++] = w[e];
            col[fill[j]] = i; val[fill[j]++] = w[e];
        }
    } else {
        const int nSlices = (n + 63) / 64;
        start.assign((size_t)nSlices + 2, 0);
        for (int sl = 0; sl < nSlices; ++sl) {
            int wmax = 0;
            for (int i = sl * 64; i < std::min(n, sl * 64 + 64); ++i) wmax = std::max(wmax, deg[i]);
            start[sl + 1] = start[sl] + wmax;
        }
        start[nSlices + 1] = start[nSlices];
        stored = (long long)start[nSlices] * 64;
        col.assign(std::max<size_t>((size_t)stored, 1), -1); val.assign(std::max<size_t>((size_t)stored, 1), 0.0);
        std::vector<int> fill((size_t)n, 0);
        auto slot = [&](int i, int k) { return ((size_t)start[i >> 6] + k) * 64 + (i & 63); };
        for (size_t e = 0; e < I.size(); ++e) {
            const int i = I[e], j = J[e];
            col[slot(i, fill[i])] = j; val[slot(i, fill[i]++)] = w[e];
            col[slot(j, fill[j])] = i; val[slot(j, fill[j]++)] = w[e];
        }
    }
    std::vector<double> inv;
    const bool dense = last && !S->L.empty() && n <= S->denseMax && denseInverse(n, I, J, w, diag, inv);
    MgLevelDev lv;
    lv.n = n; lv.width = width; lv.entries = stored; lv.xrun = S->rowRun;
    lv.diag = S->alloc<double>(n, diag.data());
    const int* startDev = S->alloc<int>(start.size(), start.data());
    if (rows) lv.rowStart = startDev; else lv.sliceStart = startDev;
    lv.col = S->alloc<int>(col.size(), col.data());
    lv.val = S->alloc<double>(val.size(), val.data());
    if (S->f32) {
        // the double-precision level keeps only what the CG's own matrix product needs (level 0) or nothing
        std::vector<float> vf(val.begin(), val.end()), df(diag.begin(), diag.end());
        MgLevelT<float> lf;
        lf.n = n; lf.width = width; lf.entries = lv.entries; lf.xrun = lv.xrun; lf.col = lv.col; lf.sliceStart = lv.sliceStart; lf.rowStart = lv.rowStart;
        lf.diag = S->alloc<float>(n, df.data());
        lf.val = S->alloc<float>(vf.size(), vf.data());
        lf.x = S->alloc<float>(n); lf.x2 = S->alloc<float>(n); lf.b = S->alloc<float>(n); lf.r = S->alloc<float>(n);
        if (dense) { std::vector<float> invf(inv.begin(), inv.end()); lf.inverse = S->alloc<float>(invf.size(), invf.data()); }
        S->Lf.push_back(lf);
    } else {
        lv.x = S->alloc<double>(n); lv.x2 = S->alloc<double>(n); lv.b = S->alloc<double>(n); lv.r = S->alloc<double>(n);
        if (dense) lv.inverse = S->alloc<double>(inv.size(), inv.data());
    }
    S->L.push_back(lv);
}

// a host CSR as sliced ELL on the device (values in T)
template <typename T>
static void mgUploadEll(PressureSolver* S, int nRows, const HostCsr& M, const int** sliceStartOut, const int** colOut, const T** valOut) {
    const int nSlices = (nRows + 63) / 64;
    std::vector<int> sliceStart((size_t)nSlices + 2, 0);
    for (int sl = 0; sl < nSlices; ++sl) {
        int64_t w = 0;
        for (int i = sl * 64; i < std::min(nRows, sl * 64 + 64); ++i) w = std::max(w, M.off[i + 1] - M.off[i]);
        sliceStart[sl + 1] = sliceStart[sl] + (int)w;
    }
    sliceStart[nSlices + 1] = sliceStart[nSlices];
    const size_t nEntries = std::max<size_t>((size_t)sliceStart[nSlices] * 64, 1);
    std::vector<int> col(nEntries, -1);
    std::vector<T> val(nEntries, (T)0);
    for (int i = 0; i < nRows; ++i)
        for (int64_t q = M.off[i]; q < M.off[i + 1]; ++q) {
            const size_t at = ((size_t)sliceStart[i >> 6] + (size_t)(q - M.off[i])) * 64 + (i & 63);
            col[at] = M.col[q]; val[at] = (T)M.val[q];
        }
    *sliceStartOut = S->alloc<int>(sliceStart.size(), sliceStart.data());
    *colOut = S->alloc<int>(col.size(), col.data());
    *valOut = S->alloc<T>(val.size(), val.data());
}

template <typename T>
static void mgUploadCsr(PressureSolver* S, int nRows, const HostCsr& M, const int** rowStartOut, const int** colOut, const T** valOut) {
    if (M.off[nRows] > 0x7fffffffLL) throw std::runtime_error("multigrid transfer operator has more than 2^31 entries");
    std::vector<int> start((size_t)nRows + 1);
    for (int i = 0; i <= nRows; ++i) start[i] = (int)M.off[i];
    std::vector<T> val(M.val.begin(), M.val.end());
    *rowStartOut = S->alloc<int>(start.size(), start.data());
    *colOut = S->alloc<int>(M.col.size(), M.col.data());
    *valOut = S->alloc<T>(val.size(), val.data());
}

// The levels below level 0 of graph (n, I, J, w, diag): S->L.back() is level 0 already (uploaded by the caller: the whole matrix,
// the owned block of a shard, or the distributed level 0 of DistMg).  firstTransfer (optional) receives the prolongator of level 0 and
// the size of level 1 INSTEAD of the default upload of P and P^T (the distributed level 0 keeps only its own rows of them).
static void mgBuildHierarchy(PressureSolver* S, int n, std::vector<int>& I, std::vector<int>& J, std::vector<double>& w, std::vector<double>& diag,
                     const std::function<void(const HostCsr& P, int nCoarse)>& firstTransfer, bool resume = false) {
    // resume: the levels so far (and their smoother scales) stand -- the per-rank coarsening of level 0 (distSetupStep) continues with the
    // replicated levels from level 1 on
    const bool sa = S->sa;
    const int passes = S->passes;
    const double saTheta = S->saTheta;
    if (!resume) S->smootherScale.assign(1, 1.0);
    while (n > (sa ? S->denseMax : 600) && S->L.size() < 12) {
        std::vector<int> total((size_t)n);
        for (int i = 0; i < n; ++i) total[i] = i;
        int cur = n;
        if (sa) {
            std::vector<int64_t> off;
            std::vector<int> nbr;
            std::vector<double> nw;
            std::vector<uint8_t> strong;
            adjacencyOf(n, I, J, w, off, nbr, nw);
            strengthOf(n, off, nbr, nw, diag, saTheta * std::pow(0.5, (double)(S->L.size() - 1)), strong);
            cur = rootAggregates(n, off, nbr, nw, strong, total);
            if (cur >= n || cur < 1) break;
            // lambda_max(D^-1 A): 2 on level 0 (Gershgorin; the rows are weakly diagonally dominant), estimated below it
            const double lmax = 2.0 / S->smootherScale.back();
            HostCsr P, PT;
            std::vector<int> cI, cJ;
            std::vector<double> cw, cdiag;
            smoothedLevel(n, off, nbr, nw, strong, diag, total, cur, (4.0 / 3.0) / lmax, P, PT, cI, cJ, cw, cdiag);
            MgLevelDev& fine = S->L.back();
            if (firstTransfer && S->L.size() == 1) {
                firstTransfer(P, cur);
                fine.pS = S->Lf.back().pS;
            } else if (S->f32) {
                MgLevelT<float>& ff = S->Lf.back();
                mgUploadEll<float>(S, n, P, &ff.pS, &ff.pCol, &ff.pVal);
                if (cur > 300000) { mgUploadEll<float>(S, cur, PT, &ff.ptS, &ff.ptCol, &ff.ptVal); ff.ptSliced = 1; }
                else mgUploadCsr<float>(S, cur, PT, &ff.ptS, &ff.ptCol, &ff.ptVal);
                fine.pS = ff.pS;   // marks the level; the double arrays of the levels below 0 are not used with the f32 cycle
            } else {
                mgUploadEll<double>(S, n, P, &fine.pS, &fine.pCol, &fine.pVal);
                if (cur > 300000) { mgUploadEll<double>(S, cur, PT, &fine.ptS, &fine.ptCol, &fine.ptVal); fine.ptSliced = 1; }
                else mgUploadCsr<double>(S, cur, PT, &fine.ptS, &fine.ptCol, &fine.ptVal);
            }
            I.swap(cI); J.swap(cJ); w.swap(cw); diag.swap(cdiag);
            n = cur;
            adjacencyOf(n, I, J, w, off, nbr, nw);
            S->smootherScale.push_back(2.0 / lambdaMaxOf(n, off, nbr, nw, diag));
            mgUploadLevel(S, n, I, J, w, diag, n <= S->denseMax);
            continue;
        }
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
        mgUploadLevel(S, n, I, J, w, diag, n <= 600);
    }
}

PressureSolver* pressureSolverCreate(hipStream_t stream, const MeshView& m, const double* taubyrho, const uint8_t* bKind, int refCell,
                                     int precond, int ownedBegin, int ownedEnd, const int32_t* cellGlobal, int64_t cellGlobalOffset, bool sharded) {
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
        S->nu0 = (int)knob("QGD_MG_NU0", 0, 0, 8);
        S->fuse = knob("QGD_MG_FUSE", 1, 0, 1) != 0 && !std::getenv("QGD_MG_GRAPH");
        if (S->nu0 == 0) S->nu0 = S->nu;
        S->oc = knob("QGD_MG_OC", S->oc, 0.5, 3.0);
        S->omega = knob("QGD_MG_OMEGA", S->omega, 0.1, 1.0);
        S->coarseSweeps = (int)knob("QGD_MG_COARSE_SWEEPS", S->coarseSweeps, 1, 1000);
        S->smootherSetup(knob("QGD_MG_CHEB", 0, 0, 1000), knob("QGD_MG_CHEB_LMAX", 2.0, 0.5, 4.0));
        // pairwise matching passes per level: 2 = aggregates of ~4 cells (3 passes = ~8 cells need twice the iterations)
        // smoothed aggregation with ~8-cell aggregates and no over-weighting, or (QGD_MG_SA=0) the plain aggregation of round 2
        const bool sa = S->sa = knob("QGD_MG_SA", 1, 0, 1) != 0;
        S->passes = (int)knob("QGD_MG_PASSES", 2, 1, 4);                   // plain aggregation: pairwise matching passes per level
        S->saTheta = knob("QGD_MG_SA_THETA", 0.08, 0.0, 0.9);             // strength threshold on level 0, halved per level
        S->rowRun = (int)knob("QGD_ROW_XCD_RUN", 0, 0, 4096);
        S->denseMax = (int)knob("QGD_MG_DENSE_MAX", MG_DENSE_MAX, 64, 8192);  // the last level (solved exactly) has at most this many rows
        if (sa) S->oc = knob("QGD_MG_OC", 1.0, 0.5, 3.0);
        S->distMaxCells = (int64_t)knob("QGD_MG_DIST_MAX_CELLS", (double)S->distMaxCells, 0, 2.0e9);
        // a shard with smoothed aggregation and the single-precision cycle builds the hierarchy that spans the ranks, at its first solve
        const int distMode = (int)knob("QGD_MG_DIST", 1, 0, 2);   // 1: the global level-0 matrix replicated on every rank; 2: level 0 coarsened per rank
        const bool distWanted = sharded && precond == 1 && sa && S->f32 && distMode != 0;
        S->dist.mode = distMode;
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
            if (distWanted) {
                S->dist.wanted = true;
                S->dist.own.swap(own); S->dist.nei.swap(nei); S->dist.a.swap(wAll); S->dist.diag.swap(dAll);
                if (cellGlobal) S->dist.cellGlobal.assign(cellGlobal, cellGlobal + nC);
                S->dist.cellGlobalOffset = cellGlobalOffset;
                return S;
            }
            std::vector<int> I, J;
            std::vector<double> w, diag(dAll.begin() + ob, dAll.begin() + oe);
            I.reserve((size_t)m.nIF); J.reserve((size_t)m.nIF); w.reserve((size_t)m.nIF);
            for (int f = 0; f < m.nIF; ++f)
                if (own[f] >= ob && own[f] < oe && nei[f] >= ob && nei[f] < oe) { I.push_back(own[f] - ob); J.push_back(nei[f] - ob); w.push_back(wAll[f]); }
            std::vector<int>().swap(own); std::vector<int>().swap(nei); std::vector<double>().swap(wAll); std::vector<double>().swap(dAll);
            int n = nRows;
            mgUploadLevel(S, n, I, J, w, diag, false);
            mgBuildHierarchy(S, n, I, J, w, diag, nullptr);
        }
        if (std::getenv("QGD_MG_VERBOSE"))
            for (size_t l = 0; l < S->L.size(); ++l)
                std::fprintf(stderr, "[qgd mg] level %zu: %d rows, widest %d, %lld stored entries (%.1f per row), smoother scale %.3f%s\n", l, S->L[l].n,
                             S->L[l].width, S->L[l].entries, (double)S->L[l].entries / std::max(S->L[l].n, 1),
                             l < S->smootherScale.size() ? S->smootherScale[l] : 1.0, S->L[l].pS ? ", smoothed prolongator" : "");
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
// ---- DistMg: set-up ------------------------------------------------------------------------------------------------------------
static void distBlockFallback(PressureSolver* S) {
    // too small for two levels (or a single rank): the rank-local hierarchy after all
    PressureSolver::Dist& D = S->dist;
    const int ob = S->ob, oe = S->oe;
    std::vector<int> I, J;
    std::vector<double> w, diag(D.diag.begin() + ob, D.diag.begin() + oe);
    for (size_t f = 0; f < D.own.size(); ++f)
        if (D.own[f] >= ob && D.own[f] < oe && D.nei[f] >= ob && D.nei[f] < oe) { I.push_back(D.own[f] - ob); J.push_back(D.nei[f] - ob); w.push_back(D.a[f]); }
    mgUploadLevel(S, oe - ob, I, J, w, diag, false);
    mgBuildHierarchy(S, oe - ob, I, J, w, diag, nullptr);
    D.wanted = false;
}
// level 0 of a hierarchy that spans the ranks: the owned rows with their ghost columns as a sliced ELL, every vector by local cell label
static void distUploadLevel0(PressureSolver* S) {
    PressureSolver::Dist& D = S->dist;
    const int ob = S->ob, oe = S->oe, nOwned = oe - ob;
    const int nC = S->m.nC;
    {
        std::vector<int> deg((size_t)nOwned, 0);
        for (size_t f = 0; f < D.own.size(); ++f) {
            if (D.own[f] >= ob && D.own[f] < oe) deg[D.own[f] - ob]++;
            if (D.nei[f] >= ob && D.nei[f] < oe) deg[D.nei[f] - ob]++;
        }
        const int nSlices = (nOwned + 63) / 64;
        std::vector<int> start((size_t)nSlices + 2, 0);
        int width = 0;
        for (int sl = 0; sl < nSlices; ++sl) {
            int wmax = 0;
            for (int r = sl * 64; r < std::min(nOwned, sl * 64 + 64); ++r) wmax = std::max(wmax, deg[r]);
            start[sl + 1] = start[sl] + wmax;
            width = std::max(width, wmax);
        }
        start[nSlices + 1] = start[nSlices];
        const size_t stored = std::max<size_t>((size_t)start[nSlices] * 64, 1);
        std::vector<int> col(stored, -1), fill((size_t)nOwned, 0);
        std::vector<float> val(stored, 0.0f);
        auto put = [&](int rowCell, int colCell, double a) {
            const int r = rowCell - ob;
            const size_t at = ((size_t)start[r >> 6] + fill[r]++) * 64 + (r & 63);
            col[at] = colCell; val[at] = (float)a;
        };
        for (size_t f = 0; f < D.own.size(); ++f) {
            if (D.own[f] >= ob && D.own[f] < oe) put(D.own[f], D.nei[f], D.a[f]);
            if (D.nei[f] >= ob && D.nei[f] < oe) put(D.nei[f], D.own[f], D.a[f]);
        }
        std::vector<float> df(D.diag.begin(), D.diag.end());
        MgLevelDev lv;        // the double-precision twin only marks the level (sizes for the log, pS as the flag)
        lv.n = nOwned; lv.width = width; lv.entries = (long long)stored;
        MgLevelT<float> lf;
        lf.n = nOwned; lf.width = width; lf.entries = (long long)stored;
        lf.sliceStart = S->alloc<int>(start.size(), start.data());
        lf.col = S->alloc<int>(col.size(), col.data());
        lf.val = S->alloc<float>(val.size(), val.data());
        lf.diag = S->alloc<float>(nC, df.data());
        lf.x = S->alloc<float>(nC); lf.x2 = S->alloc<float>(nC); lf.b = S->alloc<float>(nC); lf.r = S->alloc<float>(nC);
        lv.sliceStart = lf.sliceStart;
        S->L.push_back(lv); S->Lf.push_back(lf);
    }
}
// the transfer between the distributed level 0 and the replicated level 1 from the owned cells' rows of P (columns: level-1 numbers):
// P as a sliced ELL over the owned cells and, for every coarse node they touch, its row of P^T over them (ascending cell)
static void distUploadTransfer(PressureSolver* S, const HostCsr& Pl, int nCoarse) {
    PressureSolver::Dist& D = S->dist;
    const int ob = S->ob, nOwned = S->oe - S->ob;
    D.n1 = nCoarse;
    std::vector<std::pair<int64_t, int64_t>> trip;   // (node, position in Pl) -> rows of P^T over this rank's cells
    trip.reserve(Pl.col.size());
    std::vector<int> rowOfPos(Pl.col.size());
    for (int r = 0; r < nOwned; ++r)
        for (int64_t at = Pl.off[r]; at < Pl.off[r + 1]; ++at) { trip.push_back({(int64_t)Pl.col[at], at}); rowOfPos[(size_t)at] = r; }
    MgLevelT<float>& ff = S->Lf.back();
    mgUploadEll<float>(S, nOwned, Pl, &ff.pS, &ff.pCol, &ff.pVal);
    std::sort(trip.begin(), trip.end());
    std::vector<int> node, start(1, 0), colr;
    std::vector<float> valr;
    for (size_t t = 0; t < trip.size(); ++t) {
        if (t == 0 || trip[t].first != trip[t - 1].first) { if (t) start.push_back((int)colr.size()); node.push_back((int)trip[t].first); }
        colr.push_back(rowOfPos[(size_t)trip[t].second] + ob);      // local cell label: r is indexed like every level-0 vector
        valr.push_back((float)Pl.val[(size_t)trip[t].second]);
    }
    start.push_back((int)colr.size());
    D.nPt = (int)node.size();
    D.ptNode = S->alloc<int>(node.size(), node.data());
    D.ptStart = S->alloc<int>(start.size(), start.data());
    D.ptCol = S->alloc<int>(colr.size(), colr.data());
    D.ptVal = S->alloc<float>(valr.size(), valr.data());
}
static bool distSetupLocal0(PressureSolver* S);
// one stage of the set-up; returns true when the hierarchy stands (or was abandoned), false when a collective is pending
static bool distSetupStep(PressureSolver* S) {
    PressureSolver::Dist& D = S->dist;
    const int ob = S->ob, oe = S->oe, nOwned = oe - ob;
    hipStream_t stream = S->stream;
    if (D.setupStage == 0) {
        // sizes: the number of cells of the unsharded mesh, the most couplings any cell has to higher-numbered neighbours
        std::vector<int> cnt((size_t)S->m.nC, 0);
        int64_t gmax = 0;
        int K = 0;
        for (int i = ob; i < oe; ++i) gmax = std::max(gmax, D.globalOf(i) + 1);
        for (size_t f = 0; f < D.own.size(); ++f) {
            const int lo = D.globalOf(D.own[f]) < D.globalOf(D.nei[f]) ? D.own[f] : D.nei[f];
            if (lo >= ob && lo < oe) K = std::max(K, ++cnt[lo]);
        }
        S->distEnsureBuf(2);
        const double v[2] = {(double)gmax, (double)K};
        PCHECK(hipMemcpyAsync(D.buf, v, sizeof(v), hipMemcpyHostToDevice, stream));
        PCHECK(hipStreamSynchronize(stream));
        D.bufN = 2; D.pending = 3; D.setupStage = 1;
        return false;
    }
    if (D.setupStage == 1) {
        double v[2];
        PCHECK(hipMemcpyAsync(v, D.buf, sizeof(v), hipMemcpyDeviceToHost, stream));
        PCHECK(hipStreamSynchronize(stream));
        D.nCg = (int64_t)v[0]; D.K = (int)v[1];
        // The hierarchy that spans the ranks REPLICATES the global matrix: every rank gathers nCg (1 + 2K) doubles (host and device),
        // rebuilds the global rows and coarsens them on its own host -- set-up time and memory grow with the GLOBAL cell count, on every
        // rank of the node at once (64 M cells, K = 3: 3.6 GB per buffer per rank, a 64 M-row host coarsening times eight).  Above
        // distMaxCells (default 20 M: config 5's 16 M cells fit, 1.2 GB per buffer) the rank-local hierarchy is kept and said so.
        // (QGD_MG_DIST=2 gathers one flag per global cell and the level-1 matrix only: its limit is eight times the replicated set-up's, and a
        // mesh too large for the replicated set-up takes it by itself instead of falling back to rank-local hierarchies)
        if (D.mode == 1 && D.nCg > S->distMaxCells && D.nCg <= 8 * S->distMaxCells) {
            D.mode = 2;
            std::fprintf(stderr, "[qgd mg] %lld cells in all exceed QGD_MG_DIST_MAX_CELLS = %lld: level 0 of the multigrid hierarchy is coarsened per rank "
                                 "(QGD_MG_DIST=2), only the level-1 matrix is gathered\n", (long long)D.nCg, (long long)S->distMaxCells);
        }
        const bool tooLarge = D.nCg > (D.mode == 2 ? 8 * S->distMaxCells : S->distMaxCells);
        if (D.nCg <= S->denseMax || D.nCg >= 0x7fffffffLL || D.nCg == nOwned || tooLarge) {
            if (tooLarge)
                std::fprintf(stderr, "[qgd mg] %lld cells in all exceed QGD_MG_DIST_MAX_CELLS = %lld: the multigrid hierarchy stays rank-local "
                                     "(block Jacobi across the shards; expect more pressure iterations)\n", (long long)D.nCg, (long long)S->distMaxCells);
            else if (std::getenv("QGD_MG_VERBOSE")) std::fprintf(stderr, "[qgd mg] rank-local hierarchy: %lld cells in all, %d here\n", (long long)D.nCg, nOwned);
            distBlockFallback(S); D.built = true; return true;
        }
        if (D.mode == 2) return distSetupLocal0(S);
        // this rank's share of the global matrix: the diagonal of its cells and, per cell, its couplings to higher-numbered neighbours
        const int64_t n = D.nCg, K = D.K;
        std::vector<double> h((size_t)(n * (1 + 2 * K)), 0.0);
        std::vector<int> cnt((size_t)S->m.nC, 0);
        for (int i = ob; i < oe; ++i) h[(size_t)D.globalOf(i)] = D.diag[i];
        for (size_t f = 0; f < D.own.size(); ++f) {
            const int64_t go = D.globalOf(D.own[f]), gn = D.globalOf(D.nei[f]);
            const int lo = go < gn ? D.own[f] : D.nei[f];
            if (lo < ob || lo >= oe) continue;
            const int64_t glo = std::min(go, gn), ghi = std::max(go, gn);
            const int k = cnt[lo]++;
            if (k >= K || glo >= n || ghi >= n) throw std::runtime_error("distributed multigrid: the ranks disagree about the size of the global matrix");
            h[(size_t)(n + glo * K + k)] = (double)(ghi + 1);
            h[(size_t)(n + n * K + glo * K + k)] = D.a[f];
        }
        S->distEnsureBuf((int64_t)h.size());
        PCHECK(hipMemcpyAsync(D.buf, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice, stream));
        PCHECK(hipStreamSynchronize(stream));
        D.bufN = (int64_t)h.size(); D.pending = 2; D.setupStage = 2;
        return false;
    }
    if (D.setupStage >= 10) return distSetupLocal0(S);
    // stage 2: every rank holds the whole matrix now and builds the same hierarchy from it
    const int64_t n = D.nCg, K = D.K;
    std::vector<int> I, J;
    std::vector<double> w, diag((size_t)n);
    {
        std::vector<double> h((size_t)D.bufN);
        PCHECK(hipMemcpyAsync(h.data(), D.buf, sizeof(double) * h.size(), hipMemcpyDeviceToHost, stream));
        PCHECK(hipStreamSynchronize(stream));
        for (int64_t g = 0; g < n; ++g) {
            diag[(size_t)g] = h[(size_t)g];
            for (int k = 0; k < K; ++k) {
                const double nb1 = h[(size_t)(n + g * K + k)];
                if (nb1 > 0) { I.push_back((int)g); J.push_back((int)(nb1 - 1.0)); w.push_back(h[(size_t)(n + n * K + g * K + k)]); }
            }
        }
    }
    // a cell nobody contributed (a caller without the all-reduce: one rank driving a shard on its own) -> the rank-local hierarchy
    bool complete = true;
    for (int64_t g = 0; g < n && complete; ++g) complete = diag[(size_t)g] > 0.0;
    if (!complete) {
        if (std::getenv("QGD_MG_VERBOSE")) std::fprintf(stderr, "[qgd mg] rank-local hierarchy: the gathered matrix has rows nobody contributed\n");
        distBlockFallback(S); D.built = true; return true;
    }
    distUploadLevel0(S);
    auto firstTransfer = [&](const HostCsr& P, int nCoarse) {
        // the owned cells' rows of the global P (columns: nodes of the replicated level 1)
        HostCsr Pl;
        Pl.off.assign((size_t)nOwned + 1, 0);
        for (int r = 0; r < nOwned; ++r) { const int64_t g = D.globalOf(r + ob); Pl.off[r + 1] = Pl.off[r] + (P.off[g + 1] - P.off[g]); }
        Pl.col.resize((size_t)Pl.off[nOwned]); Pl.val.resize((size_t)Pl.off[nOwned]);
        for (int r = 0; r < nOwned; ++r) {
            const int64_t g = D.globalOf(r + ob);
            for (int64_t q = P.off[g], at = Pl.off[r]; q < P.off[g + 1]; ++q, ++at) { Pl.col[at] = P.col[q]; Pl.val[at] = P.val[q]; }
        }
        distUploadTransfer(S, Pl, nCoarse);
    };
    mgBuildHierarchy(S, (int)n, I, J, w, diag, firstTransfer);
    if (S->L.size() < 2) throw std::runtime_error("distributed multigrid: the global hierarchy has a single level");
    S->distEnsureBuf(D.n1);
    std::vector<int>().swap(D.own); std::vector<int>().swap(D.nei); std::vector<double>().swap(D.a); std::vector<double>().swap(D.diag);
    if (std::getenv("QGD_MG_VERBOSE"))
        for (size_t l = 0; l < S->L.size(); ++l)
            std::fprintf(stderr, "[qgd mg, spanning the ranks] level %zu: %d rows%s, %.1f stored entries per row\n", l, S->L[l].n,
                         l == 0 ? " of this rank" : " (replicated)", (double)S->L[l].entries / std::max(S->L[l].n, 1));
    D.built = true;
    return true;
}
// ---- QGD_MG_DIST=2: level 0 coarsened PER RANK ----------------------------------------------------------------------------------------
// The replicated set-up above gathers the GLOBAL level-0 matrix on every rank and coarsens it there: memory and host time grow with the
// global cell count on every rank at once.  Here each rank aggregates its OWN cells (couplings across a cut count as weak), smooths the
// prolongator of its own rows -- cells with a coupling across a cut keep their tentative row, so a coarse node's row of P^T A P needs the
// fine rows of its owner only, and the two ranks of a cut face compute the coupling of their two coarse nodes from the same number --, and
// contributes its coarse rows to the level-1 matrix, which is gathered (1/8 of the rows) and replicated with everything below it as before.
// What the ranks exchange: the aggregates' global numbers (a flag per global cell at each aggregate's lowest-numbered member: SUM, then a
// prefix sum), the numbers of the ghost cells' aggregates (two halo messages: the high and the low twelve bits as floats), the widest coarse
// row (MAX), the level-1 matrix (SUM of a zero-padded buffer, as for level 0 above).  Stages 10..15 of the set-up.
static bool distSetupLocal0(PressureSolver* S) {
    PressureSolver::Dist& D = S->dist;
    const int ob = S->ob, oe = S->oe, nOwned = oe - ob, nC = S->m.nC;
    hipStream_t stream = S->stream;
    if (!D.l0) D.l0 = std::make_shared<PressureSolver::Dist::Local0>();
    PressureSolver::Dist::Local0& Z = *D.l0;
    auto download = [&](std::vector<double>& h, int64_t n) {
        h.resize((size_t)n);
        PCHECK(hipMemcpyAsync(h.data(), D.buf, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, stream));
        PCHECK(hipStreamSynchronize(stream));
    };
    auto upload = [&](const std::vector<double>& h) {
        S->distEnsureBuf((int64_t)h.size());
        PCHECK(hipMemcpyAsync(D.buf, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice, stream));
        PCHECK(hipStreamSynchronize(stream));
        D.bufN = (int64_t)h.size();
    };
    auto sendPerCell = [&](int shift) {   // twelve bits of every owned cell's aggregate number, as a float, to the neighbours' ghost copies
        Z.tmp.assign((size_t)nC, 0.0f);
        for (int i = 0; i < nOwned; ++i) Z.tmp[(size_t)(ob + i)] = (float)((Z.gid[Z.agg[i]] >> shift) & 4095);
        MgLevelT<float>& L0 = S->Lf[0];
        PCHECK(hipMemcpyAsync(L0.x, Z.tmp.data(), sizeof(float) * (size_t)nC, hipMemcpyHostToDevice, stream));
        PCHECK(hipStreamSynchronize(stream));
        D.haloVec = L0.x; D.pending = 1;
    };
    auto receivePerCell = [&](int shift) {
        MgLevelT<float>& L0 = S->Lf[0];
        Z.tmp.resize((size_t)nC);
        PCHECK(hipMemcpyAsync(Z.tmp.data(), L0.x, sizeof(float) * (size_t)nC, hipMemcpyDeviceToHost, stream));
        PCHECK(hipStreamSynchronize(stream));
        for (int c = 0; c < nC; ++c) if (c < ob || c >= oe) Z.ghostGid[c] |= ((int)Z.tmp[(size_t)c]) << shift;
    };
    if (D.setupStage == 1) {
        // the owned block's graph, the couplings across the cuts, the aggregates
        std::vector<int> I, J;
        std::vector<double> w;
        for (size_t f = 0; f < D.own.size(); ++f) {
            const bool o = D.own[f] >= ob && D.own[f] < oe, n = D.nei[f] >= ob && D.nei[f] < oe;
            if (o && n) { I.push_back(D.own[f] - ob); J.push_back(D.nei[f] - ob); w.push_back(D.a[f]); }
            else if (o != n) { Z.cutI.push_back((o ? D.own[f] : D.nei[f]) - ob); Z.cutJ.push_back(o ? D.nei[f] : D.own[f]); Z.cutW.push_back(D.a[f]); }
        }
        Z.diag.assign(D.diag.begin() + ob, D.diag.begin() + oe);
        adjacencyOf(nOwned, I, J, w, Z.off, Z.nb, Z.nw);
        strengthOf(nOwned, Z.off, Z.nb, Z.nw, Z.diag, S->saTheta, Z.strong);
        Z.cutAdj.assign((size_t)nOwned, 0);
        for (int i : Z.cutI) Z.cutAdj[(size_t)i] = 1;
        Z.na = rootAggregates(nOwned, Z.off, Z.nb, Z.nw, Z.strong, Z.agg);
        if (Z.na < 1 || Z.na >= nOwned) { distBlockFallback(S); D.built = true; return true; }
        // a flag at the lowest global cell number of every aggregate; summed over the ranks, its prefix sums number the aggregates globally
        std::vector<int64_t> repOf((size_t)Z.na, INT64_MAX);
        for (int i = 0; i < nOwned; ++i) repOf[(size_t)Z.agg[i]] = std::min(repOf[(size_t)Z.agg[i]], D.globalOf(ob + i));
        std::vector<double> h((size_t)D.nCg, 0.0);
        for (int a = 0; a < Z.na; ++a) h[(size_t)repOf[(size_t)a]] = 1.0;
        Z.gid.resize((size_t)Z.na);
        for (int a = 0; a < Z.na; ++a) Z.gid[(size_t)a] = (int)repOf[(size_t)a];   // (the representative for now; its prefix count in stage 10)
        upload(h);
        D.pending = 2; D.setupStage = 10;
        return false;
    }
    if (D.setupStage == 10) {
        std::vector<double> h;
        download(h, D.nCg);
        std::vector<int> order((size_t)Z.na);
        for (int a = 0; a < Z.na; ++a) order[(size_t)a] = a;
        std::sort(order.begin(), order.end(), [&](int a, int b) { return Z.gid[(size_t)a] < Z.gid[(size_t)b]; });
        int64_t count = 0, g = 0;
        for (int a : order) {
            const int64_t rep = Z.gid[(size_t)a];
            for (; g < rep; ++g) count += h[(size_t)g] != 0.0 ? 1 : 0;
            Z.gid[(size_t)a] = (int)count;
        }
        for (; g < D.nCg; ++g) count += h[(size_t)g] != 0.0 ? 1 : 0;
        if (count >= (1 << 24) || count < 2) { distBlockFallback(S); D.built = true; return true; }   // (numbers travel as two twelve-bit floats)
        D.n1 = (int)count;
        distUploadLevel0(S);     // the distributed level 0 (its vectors carry the two halo messages below)
        Z.ghostGid.assign((size_t)nC, 0);
        sendPerCell(12);
        D.setupStage = 11;
        return false;
    }
    if (D.setupStage == 11) { receivePerCell(12); sendPerCell(0); D.setupStage = 12; return false; }
    if (D.setupStage == 12) {
        receivePerCell(0);
        // the prolongator of the owned rows (tentative next to a cut) and this rank's rows of P^T A P, in local aggregate numbers ...
        HostCsr P, PT;
        std::vector<int> cI, cJ;
        std::vector<double> cw, cdiag;
        smoothedLevel(nOwned, Z.off, Z.nb, Z.nw, Z.strong, Z.diag, Z.agg, Z.na, (4.0 / 3.0) / 2.0, P, PT, cI, cJ, cw, cdiag, &Z.cutAdj);
        // ... then in global numbers, with the couplings across the cuts: a cut face between the aggregates I (here) and J (there) adds its
        // coefficient to their coupling; the owner of the lower number contributes the pair (the other side computes the same number)
        std::map<std::pair<int, int>, double> pairs;
        for (size_t e = 0; e < cI.size(); ++e) {
            const int a = Z.gid[(size_t)cI[e]], b = Z.gid[(size_t)cJ[e]];
            pairs[{std::min(a, b), std::max(a, b)}] += cw[e];
        }
        for (size_t e = 0; e < Z.cutI.size(); ++e) {
            const int a = Z.gid[(size_t)Z.agg[(size_t)Z.cutI[e]]], b = Z.ghostGid[(size_t)Z.cutJ[e]];
            if (a == b || b < 0 || b >= D.n1) throw std::runtime_error("distributed multigrid: a ghost cell's aggregate number is out of range");
            if (a < b) pairs[{a, b}] += Z.cutW[e];
        }
        Z.e1I.clear(); Z.e1J.clear(); Z.e1W.clear();
        std::vector<int> cnt((size_t)D.n1, 0);
        int K1 = 0;
        for (const auto& kv : pairs) {
            Z.e1I.push_back(kv.first.first); Z.e1J.push_back(kv.first.second); Z.e1W.push_back(kv.second);
            K1 = std::max(K1, ++cnt[(size_t)kv.first.first]);
        }
        Z.d1.assign((size_t)Z.na, 0.0);
        for (int a = 0; a < Z.na; ++a) Z.d1[(size_t)a] = cdiag[(size_t)a];
        Z.pOff = P.off; Z.pCol.resize(P.col.size()); Z.pVal = P.val;
        for (size_t q = 0; q < P.col.size(); ++q) Z.pCol[q] = Z.gid[(size_t)P.col[q]];
        std::vector<int64_t>().swap(Z.off); std::vector<int>().swap(Z.nb); std::vector<double>().swap(Z.nw); std::vector<uint8_t>().swap(Z.strong);
        upload(std::vector<double>{(double)K1});
        D.pending = 3; D.setupStage = 13;
        return false;
    }
    if (D.setupStage == 13) {
        std::vector<double> v;
        download(v, 1);
        D.K = (int)v[0];
        const int64_t n = D.n1, K = D.K;
        std::vector<double> h((size_t)(n * (1 + 2 * K)), 0.0);
        for (int a = 0; a < Z.na; ++a) h[(size_t)Z.gid[(size_t)a]] = Z.d1[(size_t)a];
        std::vector<int> cnt((size_t)n, 0);
        for (size_t e = 0; e < Z.e1I.size(); ++e) {
            const int64_t lo = Z.e1I[e];
            const int k = cnt[(size_t)lo]++;
            if (k >= K) throw std::runtime_error("distributed multigrid: the ranks disagree about the width of the level-1 matrix");
            h[(size_t)(n + lo * K + k)] = (double)(Z.e1J[e] + 1);
            h[(size_t)(n + n * K + lo * K + k)] = Z.e1W[e];
        }
        upload(h);
        D.pending = 2; D.setupStage = 14;
        return false;
    }
    // stage 14: every rank holds the level-1 matrix; the levels from 1 on are built from it, identically everywhere
    const int64_t n = D.n1, K = D.K;
    std::vector<int> I, J;
    std::vector<double> w, diag((size_t)n);
    {
        std::vector<double> h;
        download(h, D.bufN);
        for (int64_t g = 0; g < n; ++g) {
            diag[(size_t)g] = h[(size_t)g];
            for (int k = 0; k < K; ++k) {
                const double nb1 = h[(size_t)(n + g * K + k)];
                if (nb1 > 0) { I.push_back((int)g); J.push_back((int)(nb1 - 1.0)); w.push_back(h[(size_t)(n + n * K + g * K + k)]); }
            }
        }
    }
    for (int64_t g = 0; g < n; ++g)
        if (!(diag[(size_t)g] > 0.0)) throw std::runtime_error("distributed multigrid: a level-1 row nobody contributed (did every rank reduce?)");
    {
        HostCsr Pl;
        Pl.off = Z.pOff; Pl.col = Z.pCol; Pl.val = Z.pVal;
        distUploadTransfer(S, Pl, (int)n);
        S->L.back().pS = S->Lf.back().pS;
    }
    std::vector<int64_t> off;
    std::vector<int> nbr;
    std::vector<double> nw;
    adjacencyOf((int)n, I, J, w, off, nbr, nw);
    S->smootherScale.assign(1, 1.0);
    S->smootherScale.push_back(2.0 / lambdaMaxOf((int)n, off, nbr, nw, diag));
    mgUploadLevel(S, (int)n, I, J, w, diag, n <= S->denseMax);
    mgBuildHierarchy(S, (int)n, I, J, w, diag, nullptr, true);
    S->distEnsureBuf(D.n1);
    std::vector<int>().swap(D.own); std::vector<int>().swap(D.nei); std::vector<double>().swap(D.a); std::vector<double>().swap(D.diag);
    D.l0.reset();
    if (std::getenv("QGD_MG_VERBOSE"))
        for (size_t l = 0; l < S->L.size(); ++l)
            std::fprintf(stderr, "[qgd mg, spanning the ranks, level 0 coarsened per rank] level %zu: %d rows%s, %.1f stored entries per row\n", l, S->L[l].n,
                         l == 0 ? " of this rank" : " (replicated)", (double)S->L[l].entries / std::max(S->L[l].n, 1));
    D.built = true;
    return true;
}
// ---- DistMg: one application z = M r, from comm point to comm point; returns true when z stands ----
static bool distApplyStep(PressureSolver* S) {
    PressureSolver::Dist& D = S->dist;
    hipStream_t stream = S->stream;
    const int ob = S->ob, nOwned = S->oe - S->ob, nb = blocksOf(nOwned), nu = S->nu0;   // level 0 of the spanning hierarchy
    MgLevelT<float>& L0 = S->Lf[0];
    MgLevelT<float>& L1 = S->Lf[1];
    const double* ctl = S->ctl;
    const float* none = nullptr;
    float* noOut = nullptr;
    auto sweep = [&](double w, const float* xin, float* xout, float* rout, double cx, double cmPrev) {
        mgSmoothOwnedKernel<float><<<nb, PB, 0, stream>>>(L0, ob, (float)w, L0.b, xin, xout, rout, ctl, (float)cx, (float)cmPrev);
    };
    auto halo = [&](float* v) { D.haloVec = v; D.pending = 1; };
    // pc: 0 start | 1 .. nu-1 pre-sweeps | 100 residual + restriction | 101 coarse levels + prolongation | 200 + s post-sweeps
    for (;;) {
        if (D.pc == 0) {
            mgConvertKernel<double, float><<<nb, PB, 0, stream>>>(nOwned, D.rhs + ob, L0.b + ob, ctl);
            D.cur = L0.x; D.nxt = L0.x2;
            sweep(S->cr[0], none, D.cur, noOut, 1.0, 0.0);
            D.pc = nu > 1 ? 1 : 100;
            halo(D.cur);
            return false;
        }
        if (D.pc >= 1 && D.pc < 100) {
            const int s2 = D.pc;
            sweep(S->cr[s2], D.cur, D.nxt, noOut, 1.0 + S->cm[s2], s2 >= 2 ? S->cm[s2] : 0.0);
            std::swap(D.cur, D.nxt);
            D.pc = s2 + 1 < nu ? s2 + 1 : 100;
            halo(D.cur);
            return false;
        }
        if (D.pc == 100) {
            sweep(S->omega, D.cur, noOut, L0.r, 1.0, 0.0);     // r = b - A x over the owned rows
            PCHECK(hipMemsetAsync(D.buf, 0, sizeof(double) * (size_t)D.n1, stream));
            if (D.nPt > 0)
                mgRestrictPartialKernel<float><<<(D.nPt + PB / 64 - 1) / (PB / 64), PB, 0, stream>>>(D.nPt, D.ptNode, D.ptStart, D.ptCol, D.ptVal, L0.r,
                                                                                                   D.buf, ctl);
            D.bufN = D.n1; D.pending = 2; D.pc = 101;
            return false;
        }
        if (D.pc == 101) {
            mgConvertKernel<double, float><<<blocksOf(D.n1), PB, 0, stream>>>(D.n1, D.buf, L1.b, ctl);
            S->vcycleT<float>(S->Lf, 1, L1.b, L1.x);
            mgProlongEllKernel<float><<<nb, PB, 0, stream>>>(nOwned, L0.pS, L0.pCol, L0.pVal, (float)S->oc, L1.x, D.cur + ob, ctl);
            D.pc = 200;
            halo(D.cur);
            return false;
        }
        const int s2 = D.pc - 200;
        sweep(S->cr[s2], D.cur, D.nxt, noOut, 1.0 + S->cm[s2], S->cm[s2]);
        std::swap(D.cur, D.nxt);
        if (s2 + 1 < nu) { D.pc = 200 + s2 + 1; halo(D.cur); return false; }
        mgConvertKernel<float, double><<<nb, PB, 0, stream>>>(nOwned, D.cur + ob, D.out + ob, ctl);
        D.pc = -1;
        PCHECK(hipGetLastError());
        return true;
    }
}
static void pressurePhaseTail(PressureSolver* S, int phase, bool rzDone = false);
// runs the preconditioner of phase `phase`; with the hierarchy that spans the ranks it stops at every comm point (pending != 0)
static void pressurePrecondition(PressureSolver* S, int phase) {
    PressureSolver::Dist& D = S->dist;
    if (!(D.wanted && D.built && S->L.size() >= 2 && S->Lf.size() >= 2 && S->Lf[0].pS && D.n1 > 0)) {
        // (unsharded: phase 4's axpyKernel wrote the head of this cycle, the cycle's last sweep leaves the partials of r.z)
        const bool rzDone = S->precondition(phase == 4, phase == 2 ? S->part : S->part + blocksOf(S->oe - S->ob));
        pressurePhaseTail(S, phase, rzDone);
        return;
    }
    D.rhs = S->r; D.out = S->z; D.pc = 0; D.resumePhase = phase;
    if (distApplyStep(S)) { D.resumePhase = -1; pressurePhaseTail(S, phase); }
}
int pressureSolvePending(PressureSolver* S, double** buf, int64_t* n) {
    if (buf) *buf = S->dist.buf;
    if (n) *n = S->dist.bufN;
    return S->dist.pending;
}
float* pressureSolverMgHaloVec(PressureSolver* S) { return S->dist.haloVec; }
static void pressureBeginBody(PressureSolver* S);
void pressureSolveContinue(PressureSolver* S) {
    PressureSolver::Dist& D = S->dist;
    if (D.pending == 0) return;
    D.pending = 0;
    if (D.resumePhase == 0) {            // set-up inside pressureSolveBegin
        if (!distSetupStep(S)) return;
        D.resumePhase = -1;
        pressureBeginBody(S);
        return;
    }
    if (distApplyStep(S)) { const int ph = D.resumePhase; D.resumePhase = -1; pressurePhaseTail(S, ph); }
}
void pressureSolveBegin(PressureSolver* S, const double* phiu, const double* phiwo, const double* pb, const double* gb, double tolerance,
                        double relTol, int maxIter, double* p) {
    S->earlyTest = false;
    S->phiu = phiu; S->phiwo = phiwo; S->pb = pb; S->gb = gb; S->tol = tolerance; S->relTol = relTol; S->maxIter = maxIter; S->p = p;
    if (S->dist.wanted && !S->dist.built) {
        // the first solve of a shard builds the hierarchy that spans the ranks: two collectives (pressureSolvePending), then the rest
        S->dist.resumePhase = 0;
        if (!distSetupStep(S)) return;
        S->dist.resumePhase = -1;
    }
    pressureBeginBody(S);
}
static void pressureBeginBody(PressureSolver* S) {
    const MeshView& m = S->m;
    hipStream_t stream = S->stream;
    const int ob = S->ob, oe = S->oe, n = oe - ob, nb = blocksOf(n);
    PoissonView v{S->a, S->gs, S->bKind, S->pb, S->gb, S->diag, S->rhs};
    ctlKernel<<<1, 1, 0, stream>>>(S->ctl, 0, (double)n, S->tol, S->relTol, S->maxIter);
    assembleKernel<<<nb, PB, 0, stream>>>(m, v, S->phiu, S->phiwo, S->refCell, 0.0, S->p, ob, oe);
    applyKernel<<<nb, PB, 0, stream>>>(m, S->a, S->diag, S->p, S->q, nullptr, ob, oe);
    residual0Kernel<<<nb, PB, 0, stream>>>(n, S->rhs + ob, S->q + ob, S->p + ob, S->r + ob, S->part, nb);
    foldCtlKernel<<<2, PB, 0, stream>>>(S->part, nb, 2, S->ctl, C_ABSR, 1);
    PCHECK(hipGetLastError());
}
// what follows the preconditioner in phases 2 and 4
static void pressurePhaseTail(PressureSolver* S, int phase, bool rzDone) {
    // rzDone: the block partials of r.z are in place already (mgSmoothLastKernel)
    hipStream_t stream = S->stream;
    const int ob = S->ob, n = S->oe - ob, nb = blocksOf(n);
    double* ctl = S->ctl;
    if (phase == 2) {
        directionCtlKernel<<<nb, PB, 0, stream>>>(n, 1, S->z + ob, S->d + ob, ctl);
        if (!rzDone) dotKernel<<<nb, PB, 0, stream>>>(n, S->r + ob, S->z + ob, S->part, ctl);
        foldCtlKernel<<<1, PB, 0, stream>>>(S->part, nb, 1, ctl, C_RZ, 0);
    } else {
        if (!rzDone) dotKernel<<<nb, PB, 0, stream>>>(n, S->r + ob, S->z + ob, S->part + nb, ctl);
        foldCtlKernel<<<2, PB, 0, stream>>>(S->part, nb, 2, ctl, C_ABSR2, 0);
    }
    PCHECK(hipGetLastError());
}
void pressureSolvePhase(PressureSolver* S, int phase) {
    const MeshView& m = S->m;
    hipStream_t stream = S->stream;
    const int ob = S->ob, oe = S->oe, n = oe - ob, nb = blocksOf(n);
    double* ctl = S->ctl;
    if (S->dist.pending) throw std::logic_error("pressureSolvePhase: a collective of the phase in flight is pending (pressureSolvePending)");
    switch (phase) {
        case 1:
            normFactorCtlKernel<<<nb, PB, 0, stream>>>(n, ctl, S->q + ob, S->A1 + ob, S->rhs + ob, S->part);
            foldCtlKernel<<<1, PB, 0, stream>>>(S->part, nb, 1, ctl, C_NORM, 1);
            break;
        case 2:
            ctlKernel<<<1, 1, 0, stream>>>(ctl, 1, 0.0, S->tol, S->relTol, S->maxIter);
            pressurePrecondition(S, 2);       // + the tail of the phase, now or after the pending collectives
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
            if (S->fusedCycle()) {
                const double sc0 = S->smootherScale.empty() ? 1.0 : S->smootherScale[0];
                axpyKernel<<<nb, PB, 0, stream>>>(n, S->p + ob, S->r + ob, S->d + ob, S->q + ob, S->part, ctl, S->Lf[0].b, S->Lf[0].x, S->Lf[0].diag,
                                                  (float)(S->cr[0] * sc0));
            } else axpyKernel<<<nb, PB, 0, stream>>>(n, S->p + ob, S->r + ob, S->d + ob, S->q + ob, S->part, ctl);
            if (S->earlyTest) {
                foldCtlKernel<<<1, PB, 0, stream>>>(S->part, nb, 1, ctl, C_ABSR2, 0);
                ctlKernel<<<1, 1, 0, stream>>>(ctl, 4, 0.0, S->tol, S->relTol, S->maxIter);
            }
            pressurePrecondition(S, 4);
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
    S->earlyTest = !(hooks && hooks->allreduce);
    // the comm points of the hierarchy that spans the ranks (set-up inside pressureSolveBegin, the cycle inside phases 2 and 4)
    auto drain = [&]() {
        double* buf; int64_t nbuf;
        for (int what; (what = pressureSolvePending(S, &buf, &nbuf)) != 0;) {
            if (what == 1) { if (hooks && hooks->haloMg) hooks->haloMg(); }
            else if (hooks && hooks->allreduceBuf) hooks->allreduceBuf(buf, nbuf, what);
            pressureSolveContinue(S);
        }
    };
    drain();
    reduce(C_ABSR, 3);
    pressureSolvePhase(S, 1);
    reduce(C_NORM, 1);
    pressureSolvePhase(S, 2);
    drain();
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
        drain();
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
double pressureSolverSweepMs(PressureSolver* S, int reps, int* rows, double* width) {
    if (rows) *rows = 0;
    if (width) *width = 0.0;
    if (S->L.empty() || reps <= 0 || S->dist.wanted) return 0.0;
    hipEvent_t a, b;
    PCHECK(hipEventCreate(&a)); PCHECK(hipEventCreate(&b));
    float ms = 0;
    const int nb = blocksOf(S->L[0].n);
    if (rows) *rows = S->L[0].n;
    if (width) *width = (double)S->L[0].entries / std::max(1, S->L[0].n);   // stored entries per row
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
