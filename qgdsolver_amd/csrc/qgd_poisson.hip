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

#include <cmath>
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
                                                      const double* __restrict__ phiwo, const int refCell, const double refValue) {
    const int c = blockIdx.x * PB + threadIdx.x;
    if (c >= m.nC) return;
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
        rhs += diag * refValue;
        diag += diag;
    }
    v.diag[c] = diag;
    v.rhs[c] = rhs;
}

// y = A x, optionally the block partial sums of x.y (for p.Ap)
__global__ __launch_bounds__(PB) void applyKernel(const MeshView m, const double* __restrict__ a, const double* __restrict__ diag,
                                                   const double* __restrict__ x, double* __restrict__ y, double* __restrict__ part) {
    const int c = blockIdx.x * PB + threadIdx.x;
    double xy = 0;
    if (c < m.nC) {
        const int n = m.cfCount[c];
        const size_t base = (size_t)m.cfSlice[c >> 6] * 64 + (c & 63);
        const double xc = x[c];
        double s = diag[c] * xc;
        for (int i = 0; i < n; ++i) {
            const int it = m.cfItem[base + (size_t)i * 64];
            const int f = it >= 0 ? it : ~it;
            if (f >= m.nIF) continue;
            const int nb = it >= 0 ? m.nei[f] : m.own[f];
            s -= a[f] * x[nb];
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
        double v = 0;
        for (int i = threadIdx.x; i < nBlocks; i += PB) v += part[(size_t)k * nBlocks + i];
        const double t = blockSum(v);
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
    assembleKernel<<<nb, PB, 0, stream>>>(m, v, phiu, phiwo, refCell, refValue);
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

}  // namespace qgd
