// Where does global_load_lds put its data?  One wave, every lane loads SIZE bytes from src + lane * SIZE (float i at word i) into
// LDS at a wave-uniform base; the LDS words are then dumped.  Also with only the even lanes active (exec mask).
// build: hipcc --offload-arch=gfx950 -O2 scripts/probes/lds_dma_probe.hip -o gpurun_out/lds_dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void* ldsPtr;
typedef const __attribute__((address_space(1))) void* gblPtr;
template <int SIZE, bool EVEN>
__global__ void probe(const float* src, float* out) {
    __shared__ float buf[512];
    for (int i = threadIdx.x; i < 512; i += 64) buf[i] = -1.0f;
    __syncthreads();
    const int lane = threadIdx.x;
    gblPtr g = (gblPtr)(reinterpret_cast<const char*>(src) + lane * SIZE);
    if (!EVEN || (lane & 1) == 0) {
        if (SIZE == 4) __builtin_amdgcn_global_load_lds(g, (ldsPtr)buf, 4, 0, 0);
        else if (SIZE == 12) __builtin_amdgcn_global_load_lds(g, (ldsPtr)buf, 12, 0, 0);
        else __builtin_amdgcn_global_load_lds(g, (ldsPtr)buf, 16, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 64) out[i] = buf[i];
}
template <int SIZE, bool EVEN>
static void run(const float* src, float* out) {
    probe<SIZE, EVEN><<<1, 64>>>(src, out);
    std::vector<float> h(512);
    (void)hipMemcpy(h.data(), out, 512 * 4, hipMemcpyDeviceToHost);
    std::printf("size %2d %s:", SIZE, EVEN ? "even lanes" : "all lanes ");
    for (int i = 0; i < 40; ++i) std::printf(" %g", h[i]);
    int last = -1;
    for (int i = 0; i < 512; ++i) if (h[i] != -1.0f) last = i;
    int holes = 0;
    for (int i = 0; i <= last; ++i) if (h[i] == -1.0f) ++holes;
    std::printf(" ... last written word %d, holes below it %d\n", last, holes);
}
int main() {
    std::vector<float> h(1024);
    for (int i = 0; i < 1024; ++i) h[i] = (float)i;
    float *src, *out;
    (void)hipMalloc(&src, 4096); (void)hipMalloc(&out, 2048);
    (void)hipMemcpy(src, h.data(), 4096, hipMemcpyHostToDevice);
    run<4, false>(src, out); run<12, false>(src, out); run<16, false>(src, out);
    run<4, true>(src, out); run<12, true>(src, out); run<16, true>(src, out);
    return 0;
}
