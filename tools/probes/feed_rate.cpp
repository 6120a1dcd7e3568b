// How fast can a CU be FED?  (tools/ only: a measurement, nothing of the product links against it.)
// Every GEMM / conv main loop of this repository ends up at 30-46 GB/s of operand delivery per CU whatever its structure
// (lock-step 8 waves, ping-pong, one wave per SIMD; profiles/r04_*), and so does the vendor library.  This probe takes the
// arithmetic away: 256 blocks (one per CU, 512 threads) stream 16 KiB "half-tiles" (128 rows x 128 B, row stride LD bytes,
// the K position advancing by 128 B per tile - the GEMM A / W access pattern) into an LDS ring by LDS-DMA with DEPTH
// half-tiles in flight, wait with counted vmcnt, and do nothing else.  Sources:
//   same      every block reads the same panel                      (L1 / L2 hits)
//   xcd       the 32 blocks of an XCD share 4 panels                (what a grouped tile order gives)
//   own       every block its own panel, 4 GiB footprint            (HBM)
// Modes: glds = global_load_lds_dwordx4 (64-bit per-lane address), buf = buffer_load_dwordx4 ... offen lds (32-bit offset +
// scalar K offset), reg = global_load_dwordx4 into VGPRs + ds_write_b128 (the vendor's register staging).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/feed_rate.cpp -o tools/build/feed_rate
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                                       \
    do {                                                                                            \
        hipError_t e_ = (x);                                                                        \
        if (e_ != hipSuccess) {                                                                     \
            fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));  \
            exit(2);                                                                                \
        }                                                                                           \
    } while (0)

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// MODE 0 glds, 1 buffer..lds, 2 registers + ds_write.  One "tile" = one 16 KiB half-tile; the block's panel has 128 rows.
template <int DEPTH, int MODE>
__global__ void __launch_bounds__(512) feed_kernel(const char* base, long panel_stride, int panels_mod, int panel_div, long ld, int ntiles,
                                                   float* sink, int gm = 0, int gn = 0) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int NS = 8;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane >> 3, cl = (lane & 7) ^ (lr & 7);
    int panel = (blockIdx.x / panel_div) % panels_mod, panel_b = panel;
    if (gm > 0) {                                // GEMM-like: the XCD's 32 blocks form a gm x gn patch; even half-tiles come from the block's A
        const int x = blockIdx.x % 8, i = blockIdx.x / 8;      // panel (shared by gn blocks), odd ones from its W panel (shared by gm blocks)
        panel = x * gm + i % gm;
        panel_b = 8 * gm + x * gn + (i / gm) % gn;
    }
    const char* pbase = base + (long)panel * panel_stride;
    const long b_delta = (long)(panel_b - panel) * panel_stride;
    const int tiles_per_row = (int)(ld / 128);
    unsigned off[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) off[n] = (unsigned)((long)(n * 64 + wave * 8 + lr) * ld + cl * 16);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(pbase), 0, 0x7fffffff, 0x00020000);
    f32x4 keep = {0.f, 0.f, 0.f, 0.f};
    f32x4 regs[DEPTH][2];
    auto issue = [&](int t, int slot_reg) {
        const int kt = gm > 0 ? t >> 1 : t;                                // (GEMM-like: A and W half-tiles alternate, same K position)
        const long koff = (long)(kt % tiles_per_row) * 128 + (long)(kt / tiles_per_row) * (128 * ld) + ((gm > 0 && (t & 1)) ? b_delta : 0);
        char* dst = lds + (t % NS) * 16384 + wave * 8 * 128;
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            if (MODE == 0) __builtin_amdgcn_global_load_lds((gptr_t)(pbase + koff + (size_t)off[n]), (lptr_t)(dst + n * 8192), 16, 0, 0);
            else if (MODE == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)(dst + n * 8192), 16, off[n], (int)koff, 0, 0);
            else regs[slot_reg][n] = *reinterpret_cast<const f32x4*>(pbase + koff + (size_t)off[n]);
        }
    };
    if constexpr (MODE != 2) {
#pragma unroll
        for (int t = 0; t < DEPTH - 1; ++t) issue(t, 0);
        for (int t = 0; t < ntiles; ++t) {
            issue(t + DEPTH - 1, 0);
            wait_vm<2 * (DEPTH - 1)>();                               // half-tile t has landed (this wave's part)
            asm volatile("s_barrier" ::: "memory");
            if ((t & 63) == 63) keep += *reinterpret_cast<const f32x4*>(lds + (t % NS) * 16384 + tid * 16);     // (keeps the stream observable)
        }
        wait_vm<0>();
    } else {
        // register ring: DEPTH half-tiles in flight in VGPRs, written to LDS when they land (compiler-counted vmcnt)
#pragma unroll
        for (int t = 0; t < DEPTH - 1; ++t) issue(t, t);
        for (int t0 = 0; t0 < ntiles; t0 += DEPTH) {
#pragma unroll
            for (int s = 0; s < DEPTH; ++s) {
                const int t = t0 + s;
                issue(t + DEPTH - 1, (s + DEPTH - 1) % DEPTH);
                char* dst = lds + (t % NS) * 16384 + (wave * 8 + lr) * 128 + ((lane & 7) * 16);
#pragma unroll
                for (int n = 0; n < 2; ++n) *reinterpret_cast<f32x4*>(dst + n * 8192) = regs[s][n];
                asm volatile("s_barrier" ::: "memory");
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        keep += *reinterpret_cast<const f32x4*>(lds + tid * 16);
    }
    if (keep[0] == 123.456f) sink[blockIdx.x] = keep[0] + keep[1] + keep[2] + keep[3];
}

template <int DEPTH, int MODE>
static float run(const char* base, long panel_stride, int panels_mod, int panel_div, long ld, int ntiles, float* sink, hipStream_t st, int gm = 0, int gn = 0) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(feed_kernel<DEPTH, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    float best = 1e30f;
    for (int r = 0; r < 4; ++r) {
        CK(hipEventRecord(e0, st));
        hipLaunchKernelGGL((feed_kernel<DEPTH, MODE>), dim3(256), dim3(512), 131072, st, base, panel_stride, panels_mod, panel_div, ld, ntiles, sink, gm, gn);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (r > 0 && ms < best) best = ms;
    }
    return best;
}

int main(int argc, char** argv) {
    const int ntiles = 1024;                     // half-tiles per block: 16 MiB per block, 4 GiB per launch
    const long panel_bytes = 16l << 20;          // 16 MiB per panel
    const int npanels = 256 + 16;                // > 4 GiB
    char* buf;
    CK(hipMalloc(&buf, panel_bytes * npanels + (64 << 20)));
    CK(hipMemset(buf, 1, panel_bytes * npanels + (64 << 20)));
    float* sink;
    CK(hipMalloc(&sink, 4096));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    printf("# operand delivery per CU, no arithmetic: 256 blocks x 512 threads, 16 KiB half-tiles (128 rows x 128 B), %d half-tiles per block\n", ntiles);
    if (argc > 1 && argv[1][0] == 'm') {         // part 1: staging mode x depth at the power-of-two row stride
        const long ld = 16384;
        const char* src_name[3] = {"same", "xcd (4 panels per XCD)", "own (4 GiB)"};
        const char* mode_name[3] = {"glds", "buffer..lds", "regs + ds_write"};
        for (int src = 0; src < 3; ++src)
            for (int mode = 0; mode < 3; ++mode) {
                printf("ld %6ld %-24s %-16s", ld, src_name[src], mode_name[mode]);
                for (int depth = 2; depth <= 8; depth += 2) {
                    const int panels_mod = src == 0 ? 1 : (src == 1 ? 32 : 256);
                    float ms = 0;
#define RUN(D, M) ms = run<D, M>(buf, panel_bytes, panels_mod, 1, ld, ntiles, sink, st)
                    if (mode == 0) { if (depth == 2) RUN(2, 0); else if (depth == 4) RUN(4, 0); else if (depth == 6) RUN(6, 0); else RUN(8, 0); }
                    if (mode == 1) { if (depth == 2) RUN(2, 1); else if (depth == 4) RUN(4, 1); else if (depth == 6) RUN(6, 1); else RUN(8, 1); }
                    if (mode == 2) { if (depth == 2) RUN(2, 2); else if (depth == 4) RUN(4, 2); else if (depth == 6) RUN(6, 2); else RUN(8, 2); }
                    const double gbs = 16384.0 * ntiles / (ms * 1e-3) / 1e9;
                    printf("  depth %d: %6.1f GB/s/CU (%5.2f TB/s)", depth, gbs, gbs * 256 / 1e3);
                }
                printf("\n");
                fflush(stdout);
            }
        return 0;
    }
    // part 2: row stride (= K of the GEMM) x sharing pattern, glds, 6 half-tiles in flight
    const long lds_[] = {1280, 2560, 5120, 10240, 16384, 16384 + 256};
    struct Src { const char* name; int panels_mod, gm, gn; };
    const Src srcs[] = {{"same panel everywhere", 1, 0, 0}, {"4 panels per XCD", 32, 0, 0}, {"GEMM 1 x 32 per XCD", 0, 1, 32}, {"GEMM 2 x 16", 0, 2, 16},
                        {"GEMM 4 x 8", 0, 4, 8}, {"GEMM 8 x 4", 0, 8, 4}, {"own panel (HBM)", 256, 0, 0}};
    for (long ld : lds_)
        for (const Src& sc : srcs) {
            const float ms = run<6, 0>(buf, panel_bytes, sc.panels_mod ? sc.panels_mod : 1, 1, ld, ntiles, sink, st, sc.gm, sc.gn);
            const double gbs = 16384.0 * ntiles / (ms * 1e-3) / 1e9;
            printf("row stride %6ld B  %-24s %6.1f GB/s/CU (%5.2f TB/s)  -> a 256 x 256 x 64 tile every %.2f us = %.0f TFLOP/s\n", ld, sc.name, gbs, gbs * 256 / 1e3,
                   65536.0 / gbs / 1e3, 256.0 * 2 * 256 * 256 * 64 / (65536.0 / gbs / 1e3) / 1e6);
            fflush(stdout);
        }
    return 0;
}
