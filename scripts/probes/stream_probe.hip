// How does achieved HBM bandwidth on one MI355X depend on the NUMBER of concurrent sequential streams a kernel reads and
// writes?  Thread i reads a[k][i] for k < NR (8 B each), writes b[k][i] for k < NW.  (Design probe for the face kernel, which
// reads ~27 and writes 15 streams.)   build: hipcc --offload-arch=gfx950 -O3 stream_probe.hip -o stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
struct Ptrs { const double* r[64]; double* w[16]; };
template <int NR, int NW>
__global__ __launch_bounds__(256) void k(Ptrs p, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double v[NR];
#pragma unroll
    for (int q = 0; q < NR; ++q) v[q] = __builtin_nontemporal_load(p.r[q] + i);
    double s = 0;
#pragma unroll
    for (int q = 0; q < NR; ++q) s += v[q];
    if (NW == 0) { if (s == 1.2345e300) p.w[0][i] = s; }
#pragma unroll
    for (int q = 0; q < NW; ++q) p.w[q][i] = s + q;
}
template <int NR, int NW>
int run(Ptrs p, size_t n) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int grid = (int)((n + 255) / 256);
    k<NR, NW><<<grid, 256>>>(p, n);
    CK(hipEventRecord(a));
    for (int it = 0; it < 5; ++it) k<NR, NW><<<grid, 256>>>(p, n);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 5;
    printf("reads %2d writes %2d: %.3f ms  %.2f TB/s\n", NR, NW, ms, (NR + NW) * 8.0 * n / ms / 1e9);
    return 0;
}
int main() {
    const size_t n = 24u << 20;   // 24 Mi elements = the faces of a 200^3 box, 192 MiB per stream
    Ptrs p;
    for (int q = 0; q < 64; ++q) { double* d; CK(hipMalloc(&d, n * 8 + 4096 * q)); CK(hipMemset(d, 0, n * 8)); p.r[q] = d; }
    for (int q = 0; q < 16; ++q) { double* d; CK(hipMalloc(&d, n * 8 + 4096 * q)); p.w[q] = d; }
    run<1, 0>(p, n); run<2, 0>(p, n); run<4, 0>(p, n); run<8, 0>(p, n); run<12, 0>(p, n); run<16, 0>(p, n); run<24, 0>(p, n); run<32, 0>(p, n);
    run<48, 0>(p, n); run<64, 0>(p, n);
    run<8, 1>(p, n); run<8, 5>(p, n); run<8, 15>(p, n); run<24, 5>(p, n); run<24, 15>(p, n); run<32, 15>(p, n);
    run<0 + 1, 15>(p, n); run<1, 5>(p, n);
    return 0;
}
