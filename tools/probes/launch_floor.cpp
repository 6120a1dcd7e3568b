// Probe (tools/ only): what does ONE dependent kernel node of a hipGraph cost when the kernel itself does (almost) nothing?
// The B <= 4 UNet programs are chains of ~700 dependent launches of 8-25 us; this sizes the part of that which no kernel
// change can remove (dispatch + drain + first-wave start), for several grid shapes, and the cost of a launch whose waves
// each do one dependent memory round trip.
// Build: hipcc --offload-arch=gfx950 -O2 -std=c++17 tools/probes/launch_floor.cpp -o tools/build/launch_floor
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(2);                                                                       \
        }                                                                                  \
    } while (0)

__global__ void k_empty(int* p) {
    if (p == (int*)1) *p = 0;
}
extern __shared__ int dyn_lds[];
__global__ void k_lds(int* p) {
    if (p == (int*)1) *p = dyn_lds[threadIdx.x];
}
// every thread: load 16 B written by the previous launch, store 16 B for the next one (one memory round trip per wave)
__global__ void k_chain(const uint4* __restrict__ in, uint4* __restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    uint4 v = in[i];
    v.x += 1;
    out[i] = v;
}
// `hops` dependent round trips per wave (pointer-chase through the buffer the previous launch wrote)
__global__ void k_hops(const unsigned* __restrict__ in, unsigned* __restrict__ out, int hops, unsigned mask) {
    unsigned i = (blockIdx.x * blockDim.x + threadIdx.x) & mask;
    for (int h = 0; h < hops; ++h) i = in[i] & mask;
    out[(blockIdx.x * blockDim.x + threadIdx.x) & mask] = i;
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 700, reps = 20;
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const size_t bytes = 64u << 20;
    uint4 *a, *b;
    CK(hipMalloc(&a, bytes));
    CK(hipMalloc(&b, bytes));
    CK(hipMemset(a, 0, bytes));
    CK(hipMemset(b, 0, bytes));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_lds), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));

    struct Case { const char* name; int kind, grid, block, lds, hops; };
    const Case cases[] = {
        {"empty 1 x 64", 0, 1, 64, 0, 0},
        {"empty 160 x 256", 0, 160, 256, 0, 0},
        {"empty 256 x 512", 0, 256, 512, 0, 0},
        {"empty 1024 x 256", 0, 1024, 256, 0, 0},
        {"lds 48K 160 x 256", 1, 160, 256, 48 * 1024, 0},
        {"lds 150K 256 x 512", 1, 256, 512, 150 * 1024, 0},
        {"chain 16B/thread 160 x 256", 2, 160, 256, 0, 0},
        {"chain 16B/thread 2048 x 256 (8 MB)", 2, 2048, 256, 0, 0},
        {"hops 1 160 x 256", 3, 160, 256, 0, 1},
        {"hops 4 160 x 256", 3, 160, 256, 0, 4},
        {"hops 10 160 x 256", 3, 160, 256, 0, 10},
        {"hops 20 160 x 256", 3, 160, 256, 0, 20},
    };
    printf("# hipGraph of %d DEPENDENT kernel nodes (stream capture), %d replays; us per node\n", N, reps);
    for (const Case& c : cases) {
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < N; ++i) {
            uint4* in = (i & 1) ? b : a;
            uint4* out = (i & 1) ? a : b;
            if (c.kind == 0) hipLaunchKernelGGL(k_empty, dim3(c.grid), dim3(c.block), 0, s, (int*)nullptr);
            else if (c.kind == 1) hipLaunchKernelGGL(k_lds, dim3(c.grid), dim3(c.block), c.lds, s, (int*)nullptr);
            else if (c.kind == 2) hipLaunchKernelGGL(k_chain, dim3(c.grid), dim3(c.block), 0, s, in, out);
            else hipLaunchKernelGGL(k_hops, dim3(c.grid), dim3(c.block), 0, s, (const unsigned*)in, (unsigned*)out, c.hops, (1u << 22) - 1);
        }
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("   %-40s %7.2f us per node\n", c.name, ms * 1e3f / reps / N);
        fflush(stdout);
        CK(hipGraphExecDestroy(ge));
        CK(hipGraphDestroy(g));
    }
    return 0;
}
