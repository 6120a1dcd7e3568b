// Torch-free GEMM harness for the MI355X box (tools/ only - a DIAGNOSTIC, never linked into the product):
//   * correctness of a forced lb_gemm_f16 variant against the library's automatic choice on the same operands
//     (the main loops are bit-identical by construction: same K order per accumulator), with bias / residual / GEGLU
//     epilogues as in the UNet programs;
//   * cold-weight timing (weights rotated through more copies than the Infinity Cache holds, as in the model: 5.1 GB of
//     weights per UNet forward), variants interleaved in ONE process, median of rounds (hipEvents on the launch stream);
//   * the vendor library (rocBLAS gemm_ex, fp16 in / fp32 accumulate) on the same operands, to size the head-room.
// Build (cross-compiles here):  hipcc --offload-arch=gfx950 -O2 -std=c++17 tools/gemm_bench.cpp -I include
//        -L latentblending_amd/hip -llbhip -lrocblas -Wl,-rpath,'$ORIGIN/../../latentblending_amd/hip' -o tools/build/gemm_bench
// Usage: tools/build/gemm_bench [set] [rounds]      set = b17 | b2 | big | all | ksweep
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <rocblas/rocblas.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "lb_hip.h"

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(2);                                                                       \
        }                                                                                  \
    } while (0)

// GB_PREFETCH study: touch a weight matrix from a second stream while the previous GEMM runs (does a weight that already sits in the
// Infinity Cache / an L2 make the launch-bound small-M GEMMs faster?).  16 B per lane, the sum goes nowhere.
__global__ void prefetch_touch_kernel(const uint4* __restrict__ w, long n16, unsigned* sink) {
    unsigned acc = 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long)gridDim.x * blockDim.x) {
        const uint4 v = w[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345679u) *sink = acc;
}

struct Shape { int M, N, K; int geglu; int epi; const char* what; };     // epi: 0 none, 1 bias, 2 bias + residual

static const Shape B17[] = {
    {4352, 10240, 1280, 1, 1, "ff GEGLU (60 / forward)"},
    {4352, 1280, 1280, 0, 2, "attn out / cross q (192)"},
    {4352, 1280, 5120, 0, 2, "ff out (60)"},
    {4352, 3840, 1280, 0, 0, "self qkv (60)"},
    {17408, 5120, 640, 1, 1, "ff GEGLU 640 (10)"},
    {17408, 640, 640, 0, 2, "attn out 640 (40)"},
    {17408, 640, 2560, 0, 2, "ff out 640 (10)"},
    {17408, 1920, 640, 0, 0, "self qkv 640 (10)"},
    {4352, 1280, 2560, 0, 1, "proj / skip 1x1"},
    {69632, 320, 640, 0, 1, "skip 1x1 320"},
};
static const Shape B2[] = {
    {512, 1280, 1280, 0, 2, "attn out / cross q (192)"},
    {512, 1280, 5120, 0, 2, "ff out (60)"},
    {512, 10240, 1280, 1, 1, "ff GEGLU (60)"},
    {512, 3840, 1280, 0, 0, "self qkv (60)"},
    {2048, 640, 640, 0, 2, "attn out 640 (40)"},
    {2048, 5120, 640, 1, 1, "ff GEGLU 640"},
};
// K sweep of the most frequent projection (no epilogue operands): per-K-tile slope and fixed cost of a launch, ours and the vendor's
static const Shape KSWEEP[] = {
    {4352, 1280, 640, 0, 0, "K sweep: 10 K-tiles"},
    {4352, 1280, 1152, 0, 0, "K sweep: 18 K-tiles"},
    {4352, 1280, 1280, 0, 0, "K sweep: 20 K-tiles"},
    {4352, 1280, 1920, 0, 0, "K sweep: 30 K-tiles"},
    {4352, 1280, 2560, 0, 0, "K sweep: 40 K-tiles"},
};
static const Shape BIG[] = {
    {8192, 8192, 8192, 0, 0, "8192^3"},
    {4096, 4096, 4096, 0, 0, "4096^3"},
    {8192, 8192, 1280, 0, 0, "8192^2 x 1280"},
};

static uint32_t rng_state = 12345u;
static inline float frand() {               // uniform [-1, 1)
    rng_state = rng_state * 1664525u + 1013904223u;
    return (float)(int32_t)rng_state * (1.0f / 2147483648.0f);
}
static void fill_half(std::vector<__half>& v, float scale) {
    for (auto& x : v) x = __float2half(frand() * scale);
}

struct Variant { const char* name; int tile; int splitk; int group; int lean = 1; int noepi = 0; };    // lean: lb_gemm_set_lean_epilogue; noepi: the launch WITHOUT bias / residual (what the rocBLAS row computes: alpha A W^T, beta = 0)

int main(int argc, char** argv) {
    const std::string set = argc > 1 ? argv[1] : "b17";
    const int rounds = argc > 2 ? atoi(argv[2]) : 5;
    std::vector<Shape> shapes;
    if (set == "b17" || set == "all") shapes.insert(shapes.end(), std::begin(B17), std::end(B17));
    if (set == "b2" || set == "all") shapes.insert(shapes.end(), std::begin(B2), std::end(B2));
    if (set == "big" || set == "all") shapes.insert(shapes.end(), std::begin(BIG), std::end(BIG));
    if (set == "ksweep") shapes.insert(shapes.end(), std::begin(KSWEEP), std::end(KSWEEP));
    const Variant all_variants[] = {{"auto", 0, 0, 8}, {"auto-rowepi", 0, 0, 8, 0}, {"auto-noepi", 0, 0, 8, 1, 1}, {"pp", 9, 1, 8}, {"pp-g0", 9, 1, 0}, {"pp-g4", 9, 1, 4}, {"t5", 5, 1, 8}, {"t4", 4, 1, 8}, {"t7", 7, 1, 8}, {"t10", 10, 1, 8}, {"t11", 11, 0, 8}, {"t3", 3, 0, 8}, {"t11-s1", 11, 1, 8}, {"t3-s1", 3, 1, 8}, {"t1", 1, 1, 8}, {"t11-s2", 11, 2, 8}, {"t3-s2", 3, 2, 8}, {"t3-s8", 3, 8, 8}, {"t2", 2, 0, 8}, {"t2-s2", 2, 2, 8}, {"t2-s4", 2, 4, 8}, {"t2-s8", 2, 8, 8}, {"t1-s4", 1, 4, 8}, {"t1-s8", 1, 8, 8}, {"t10-s2", 10, 2, 8}};
    // GB_VARIANTS=auto,pp-m1 selects (the first one is the reference of the bit-identity check); GB_NOCHECK / GB_NOROCBLAS = 1 skip those parts
    std::vector<Variant> variants;
    {
        const char* sel = getenv("GB_VARIANTS");
        std::string want = sel ? std::string(",") + sel + "," : "";
        for (const Variant& v : all_variants)
            if (want.empty() || want.find(std::string(",") + v.name + ",") != std::string::npos) variants.push_back(v);
    }
    const int NV = (int)variants.size();
    const bool nocheck = getenv("GB_NOCHECK") != nullptr, norocblas = getenv("GB_NOROCBLAS") != nullptr;
    // GB_WARM=1: every launch reads the SAME weight copy (resident in L2 / Infinity Cache: the bound a perfect prefetcher could reach);
    // GB_PREFETCH=<blocks>: weights stay cold, but copy i + 1 is touched by <blocks> x 256 threads on a second stream while GEMM i runs
    const bool warm = getenv("GB_WARM") != nullptr;
    const int prefetch_blocks = getenv("GB_PREFETCH") ? atoi(getenv("GB_PREFETCH")) : 0;
    hipStream_t pstream;
    CK(hipStreamCreate(&pstream));
    std::vector<hipEvent_t> pev(64);
    for (auto& e : pev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    unsigned* psink;
    CK(hipMalloc(&psink, 4));
    setvbuf(stdout, nullptr, _IOLBF, 0);

    hipStream_t stream;
    CK(hipStreamCreate(&stream));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    rocblas_handle rb;
    rocblas_create_handle(&rb);
    rocblas_set_stream(rb, stream);
    void* zero_page;
    CK(hipMalloc(&zero_page, 256));
    CK(hipMemset(zero_page, 0, 256));
    printf("# lb_gemm_f16 variants vs the automatic choice (bit-identity) and rocBLAS gemm_ex; %s weights%s, median of %d rounds\n",
           warm ? "WARM (one copy)" : "cold", prefetch_blocks > 0 ? " + next copy touched from a second stream" : "", rounds);

    for (const Shape& s : shapes) {
        const long wbytes = (long)s.N * s.K * 2;
        int nw = (int)std::max(2l, std::min(48l, (long)(400e6 / wbytes)));
        if ((long)s.M * s.N * s.K > 200e9) nw = 2;                    // the big squares are their own cache-busters
        const int n_out = s.geglu ? s.N / 2 : s.N;
        std::vector<__half> hA((size_t)s.M * s.K), hW((size_t)s.N * s.K), hR((size_t)s.M * n_out);
        std::vector<float> hB(s.N);
        fill_half(hA, 1.0f);
        fill_half(hW, 1.0f / std::sqrt((float)s.K) * 1.7f);
        fill_half(hR, 1.0f);
        for (auto& b : hB) b = frand();
        __half *dA, *dR, *dC, *dCref;
        float* dB;
        std::vector<__half*> dW(nw);
        CK(hipMalloc(&dA, hA.size() * 2));
        CK(hipMalloc(&dR, hR.size() * 2));
        CK(hipMalloc(&dC, (size_t)s.M * s.N * 2));          // (rocBLAS writes the full M x N also for the GEGLU shapes)
        CK(hipMalloc(&dCref, (size_t)s.M * s.N * 2));
        CK(hipMalloc(&dB, hB.size() * 4));
        float* dWs;
        CK(hipMalloc(&dWs, (size_t)lb_gemm_workspace_bytes(s.M, s.N)));
        CK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(dR, hR.data(), hR.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
        for (int i = 0; i < nw; ++i) {
            CK(hipMalloc(&dW[i], wbytes));
            if (i == 0) CK(hipMemcpy(dW[0], hW.data(), wbytes, hipMemcpyHostToDevice));
            else CK(hipMemcpy(dW[i], dW[0], wbytes, hipMemcpyDeviceToDevice));
        }
        auto params = [&](int wi, __half* out, bool noepi = false) {
            LbGemmParams p;
            memset(&p, 0, sizeof(p));
            p.A = (const lb_half*)dA; p.W = (const lb_half*)dW[wi]; p.C = out;
            p.M = s.M; p.N = s.N; p.K = s.K; p.lda = s.K; p.ldw = s.K; p.ldc = n_out; p.ldr = n_out;
            p.rows_per_batch = s.M; p.alpha = 1.f; p.zero_page = zero_page; p.partial = dWs;
            if (s.geglu) p.flags |= LB_GEMM_GEGLU;
            if (s.epi >= 1 && !noepi) p.bias = dB;
            if (s.epi >= 2 && !s.geglu && !noepi) p.residual = dR;
            return p;
        };
        auto run = [&](const Variant& v, int wi, __half* out) {
            lb_gemm_set_tuning(v.tile, v.splitk);
            lb_gemm_pp_set_group(v.group);
            lb_gemm_set_lean_epilogue(v.lean);
            LbGemmParams p = params(wi, out, v.noepi != 0);
            const int rc = lb_gemm_f16(&p, stream);
            if (rc) { fprintf(stderr, "lb_gemm_f16 failed: %s\n", lb_last_error_string()); exit(3); }
        };
        printf("M=%6d N=%6d K=%5d %s%s  [%s]\n", s.M, s.N, s.K, s.geglu ? "GEGLU " : "", s.epi == 2 ? "bias+res" : (s.epi ? "bias" : "plain"), s.what);
        // ---- correctness: every variant against the first one ----
        if (!nocheck) {
        run(variants[0], 0, dCref);
        CK(hipStreamSynchronize(stream));
        std::vector<__half> hC(hR.size()), hCref(hR.size());
        CK(hipMemcpy(hCref.data(), dCref, hCref.size() * 2, hipMemcpyDeviceToHost));
        double ref_abs = 0;
        for (auto& x : hCref) ref_abs = std::max(ref_abs, (double)std::fabs(__half2float(x)));
        for (int v = 1; v < NV; ++v) {
            if (variants[v].noepi) continue;                           // (a different function: timing row only)
            CK(hipMemsetAsync(dC, 0xff, hC.size() * 2, stream));
            run(variants[v], 0, dC);
            CK(hipStreamSynchronize(stream));
            CK(hipMemcpy(hC.data(), dC, hC.size() * 2, hipMemcpyDeviceToHost));
            long bad = 0;
            double worst = 0;
            for (size_t i = 0; i < hC.size(); ++i) {
                const float a = __half2float(hC[i]), b = __half2float(hCref[i]);
                if (memcmp(&hC[i], &hCref[i], 2) != 0) {
                    ++bad;
                    const double d = std::isfinite(a) ? std::fabs((double)a - b) : 1e30;
                    worst = std::max(worst, d);
                }
            }
            printf("   check %-12s vs the first: %ld / %zu elements differ, max |diff| %.4g (max |ref| %.3g)%s\n", variants[v].name, bad, hC.size(),
                   worst, ref_abs, bad == 0 ? "  BIT-IDENTICAL" : (worst <= 2e-3 * ref_abs ? "  (rounding-level)" : "  ** MISMATCH **"));
        }
        }
        // ---- timing ----
        const double flops = 2.0 * s.M * s.N * s.K;
        std::vector<std::vector<float>> us(NV + 1);
        const float alpha = 1.f, beta = 0.f;
        for (int r = 0; r < rounds + 1; ++r) {
            for (int v = 0; v <= NV; ++v) {
                if (v == NV && norocblas) continue;
                CK(hipEventRecord(e0, stream));
                for (int i = 0; i < nw; ++i) {
                    if (prefetch_blocks > 0 && v < NV) {       // copy i + 1 is touched while GEMM i runs (never more than one launch ahead)
                        if (i > 0) CK(hipStreamWaitEvent(pstream, pev[(i - 1) % 64], 0));
                        hipLaunchKernelGGL(prefetch_touch_kernel, dim3(prefetch_blocks), dim3(256), 0, pstream,
                                           (const uint4*)dW[(i + 1) % nw], wbytes / 16, psink);
                    }
                    if (v < NV) {
                        run(variants[v], warm ? 0 : i, dC);
                        if (prefetch_blocks > 0) CK(hipEventRecord(pev[i % 64], stream));
                    } else
                        rocblas_gemm_ex(rb, rocblas_operation_transpose, rocblas_operation_none, s.N, s.M, s.K, &alpha, dW[warm ? 0 : i], rocblas_datatype_f16_r,
                                        s.K, dA, rocblas_datatype_f16_r, s.K, &beta, dC, rocblas_datatype_f16_r, s.N, dC, rocblas_datatype_f16_r, s.N,
                                        rocblas_datatype_f32_r, rocblas_gemm_algo_standard, 0, 0);
                    if (false)
                        rocblas_gemm_ex(rb, rocblas_operation_transpose, rocblas_operation_none, s.N, s.M, s.K, &alpha, dW[i], rocblas_datatype_f16_r,
                                        s.K, dA, rocblas_datatype_f16_r, s.K, &beta, dC, rocblas_datatype_f16_r, s.N, dC, rocblas_datatype_f16_r, s.N,
                                        rocblas_datatype_f32_r, rocblas_gemm_algo_standard, 0, 0);
                }
                CK(hipEventRecord(e1, stream));
                CK(hipEventSynchronize(e1));
                if (prefetch_blocks > 0) CK(hipStreamSynchronize(pstream));
                float ms = 0;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (r > 0) us[v].push_back(ms * 1e3f / nw);           // (round 0 = warm-up)
            }
        }
        for (int v = 0; v <= NV; ++v) {
            if (us[v].empty()) continue;
            std::sort(us[v].begin(), us[v].end());
            const float med = us[v][us[v].size() / 2], best = us[v][0];
            printf("   %-14s %8.1f us median (%7.1f TF/s)   best %8.1f us\n", v < NV ? variants[v].name : "rocBLAS (plain)", med, flops / med / 1e6, best);
        }
        fflush(stdout);
        lb_gemm_set_tuning(0, 0);
        for (auto w : dW) CK(hipFree(w));
        CK(hipFree(dWs));
        CK(hipFree(dA)); CK(hipFree(dR)); CK(hipFree(dC)); CK(hipFree(dCref)); CK(hipFree(dB));
    }
    return 0;
}
