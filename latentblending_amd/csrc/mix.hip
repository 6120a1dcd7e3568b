// Latent-mixing primitives for gfx950: whole-tensor slerp (float64 reductions), conditioning
// lerp, scheduler input scaling, CFG combine + Euler / Euler-ancestral update.
// All are HBM/L2-bound elementwise kernels: 16 B per lane, coalesced, wave-shuffle reductions.
//
// Reference behaviour (paths relative to /root/reference):
//   slerp           latentblending/utils.py:29-71      (called at blending_engine.py:449 and
//                                                        diffusers_holder.py:324)
//   lerp            latentblending/utils.py:97          (blending_engine.py:650)
//   scale / step    diffusers_holder.py:330, 347-349, 356 (diffusers Euler schedulers)
#include "lb_common.h"
#include <vector>

#define LB_MAX_PAIRS 16

struct SlerpPairs {
    const void* p0[LB_MAX_PAIRS];
    const void* p1[LB_MAX_PAIRS];
    void* out[LB_MAX_PAIRS];
    double fract[LB_MAX_PAIRS];
};

template <typename T> struct OutOf { typedef float type; };
template <> struct OutOf<f16> { typedef f16 type; };

template <typename O> __device__ __forceinline__ O lb_from_f64(double x);
template <> __device__ __forceinline__ f16 lb_from_f64<f16>(double x) { return lb_f64_to_f16(x); }
template <> __device__ __forceinline__ float lb_from_f64<float>(double x) { return (float)x; }

// Block-wide sum of three doubles; result valid in every thread.
__device__ __forceinline__ void block_sum3(double& a, double& b, double& c, double* red) {
    a = lb_wave_sum(a); b = lb_wave_sum(b); c = lb_wave_sum(c);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    if (lane == 0) { red[wave * 3 + 0] = a; red[wave * 3 + 1] = b; red[wave * 3 + 2] = c; }
    __syncthreads();
    if (wave == 0) {
        double x = lane < nw ? red[lane * 3 + 0] : 0.0;
        double y = lane < nw ? red[lane * 3 + 1] : 0.0;
        double z = lane < nw ? red[lane * 3 + 2] : 0.0;
        x = lb_wave_sum(x); y = lb_wave_sum(y); z = lb_wave_sum(z);
        if (lane == 0) { red[48] = x; red[49] = y; red[50] = z; }
    }
    __syncthreads();
    a = red[48]; b = red[49]; c = red[50];
}

// weights of p0 / p1 from <p0,p1>, |p0|^2, |p1|^2 — the reference's formula, in float64.
__device__ __forceinline__ void slerp_weights(double dot, double n0sq, double n1sq, double fract,
                                              double& w0, double& w1) {
    double c = dot / (sqrt(n0sq) * sqrt(n1sq));
    const double lim = 1.0 - 1e-7;
    c = c > lim ? lim : (c < -lim ? -lim : c);   // NaN passes through, as torch.clamp does
    const double theta = acos(c);
    const double st = sin(theta);
    const double tt = theta * fract;
    w0 = sin(theta - tt) / st;
    w1 = sin(tt) / st;
}

// The weights are a few hundred dependent float64 instructions (sqrt, acos, three sin, divisions).  Wave 0 evaluates them and
// publishes them through LDS while the other waves of the block sleep at the barrier (the CU's VALU stays free for the other
// resident blocks); its lanes 0, 1, 2 take ONE of the three sines each - same functions on the same arguments as the
// sequential chain (slerp_weights), so the weights are the same bits, two sine latencies sooner.
__device__ __forceinline__ void slerp_weights_block(double dot, double n0sq, double n1sq, double fract,
                                                    double& w0, double& w1, double* red) {
    if (threadIdx.x < LB_WAVE) {
        double c = dot / (sqrt(n0sq) * sqrt(n1sq));
        const double lim = 1.0 - 1e-7;
        c = c > lim ? lim : (c < -lim ? -lim : c);   // NaN passes through, as torch.clamp does
        const double theta = acos(c);
        const double tt = theta * fract;
        const int l = threadIdx.x;
        const double s = sin(l == 0 ? theta : (l == 1 ? theta - tt : tt));
        const double st = __shfl(s, 0, LB_WAVE), sa = __shfl(s, 1, LB_WAVE), sb = __shfl(s, 2, LB_WAVE);
        if (l == 0) {
            red[52] = sa / st;
            red[53] = sb / st;
        }
    }
    __syncthreads();
    w0 = red[52];
    w1 = red[53];
}

template <typename T, int VEC>
__device__ __forceinline__ void slerp_body(const T* __restrict__ p0, const T* __restrict__ p1,
                                           typename OutOf<T>::type* __restrict__ out, long n,
                                           double fract, T* stage0, T* stage1, bool staged,
                                           double* red) {
    typedef typename OutOf<T>::type O;
    typedef T VT __attribute__((ext_vector_type(VEC)));
    typedef O VO __attribute__((ext_vector_type(VEC)));
    const long nvec = n / VEC;
    double dot = 0, s0 = 0, s1 = 0;
    for (long i = threadIdx.x; i < nvec; i += blockDim.x) {
        VT a = *reinterpret_cast<const VT*>(p0 + i * VEC);
        VT b = *reinterpret_cast<const VT*>(p1 + i * VEC);
        if (staged) {
            *reinterpret_cast<VT*>(stage0 + i * VEC) = a;
            *reinterpret_cast<VT*>(stage1 + i * VEC) = b;
        }
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const double x = (double)a[j], y = (double)b[j];
            dot += x * y; s0 += x * x; s1 += y * y;
        }
    }
    for (long i = nvec * VEC + threadIdx.x; i < n; i += blockDim.x) {
        const double x = (double)p0[i], y = (double)p1[i];
        dot += x * y; s0 += x * x; s1 += y * y;
    }
    block_sum3(dot, s0, s1, red);
    double w0, w1;
    slerp_weights_block(dot, s0, s1, fract, w0, w1, red);
    for (long i = threadIdx.x; i < nvec; i += blockDim.x) {
        VT a, b;
        if (staged) {
            a = *reinterpret_cast<const VT*>(stage0 + i * VEC);
            b = *reinterpret_cast<const VT*>(stage1 + i * VEC);
        } else {
            a = *reinterpret_cast<const VT*>(p0 + i * VEC);
            b = *reinterpret_cast<const VT*>(p1 + i * VEC);
        }
        VO r;
#pragma unroll
        for (int j = 0; j < VEC; ++j)
            r[j] = lb_from_f64<O>(__dadd_rn(__dmul_rn((double)a[j], w0), __dmul_rn((double)b[j], w1)));
        *reinterpret_cast<VO*>(out + i * VEC) = r;
    }
    for (long i = nvec * VEC + threadIdx.x; i < n; i += blockDim.x)
        out[i] = lb_from_f64<O>(__dadd_rn(__dmul_rn((double)p0[i], w0), __dmul_rn((double)p1[i], w1)));
}

template <typename T, int VEC>
__global__ void __launch_bounds__(512) slerp_pairs_kernel(SlerpPairs args, long n) {
    __shared__ double red[64];
    const int b = blockIdx.x;
    slerp_body<T, VEC>((const T*)args.p0[b], (const T*)args.p1[b],
                       (typename OutOf<T>::type*)args.out[b], n, args.fract[b], nullptr, nullptr,
                       false, red);
}

// Contiguous batch [npairs][n]; inputs optionally staged in LDS so HBM is read exactly once
// (6 B/element for fp16: read p0, read p1, write out).
template <typename T, int VEC>
__global__ void __launch_bounds__(512) slerp_batched_kernel(const T* __restrict__ p0,
                                                             const T* __restrict__ p1,
                                                             typename OutOf<T>::type* __restrict__ out,
                                                             const double* __restrict__ fracts,
                                                             long n, int staged) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* red = reinterpret_cast<double*>(smem);            // 64 doubles = 512 B
    T* stage0 = reinterpret_cast<T*>(smem + 512);
    T* stage1 = stage0 + n;
    const long b = blockIdx.x;
    slerp_body<T, VEC>(p0 + b * n, p1 + b * n, out + b * n, n, fracts[b], stage0, stage1,
                       staged != 0, red);
}

// Batch of pairs with element strides between consecutive pairs (0 = every pair reads the same tensor: the
// parental mix of ONE pair of anchors at many fractions) and device-side fractions.  Each thread keeps its
// (An fp32 shortcut for the weighted sum - r = fma32(a, w0f, b w1f), accepted when half(r - e) == half(r + e) for the
// error bound e = 1.0625 * 2^-22 (|a w0f| + |b w1f|), else the float64 chain - is equal to the chain by construction and
// was verified bit-for-bit on 42 M elements (tests, numpy emulation).  Measured SLOWER, 2.7 vs 3.55 TB/s
// (profiles/r02_slerp_study.txt): ~0.2 % of the elements need the chain, i.e. one wave-iteration in eight, and the
// compiler keeps both paths' operands live.  Not kept; the float64 chain below is the product path.)
// VPT 16-byte vectors of both inputs in REGISTERS between the reduction and the weighted sum: HBM is read exactly
// once (6 B / element) without an LDS round trip and the block needs 0.5 KiB of LDS.  Only the RAW halves stay live across the
// reduction: the weighted sum converts them to float64 again (an opaque asm between the two phases stops the compiler from keeping
// the 64 converted doubles = 128 VGPRs of the first phase; rounds 1-3 shipped that form: 172 VGPRs at VPT = 4, ONE block per
// CU, its load / reduce / weights / sum / store phases strictly one after the other - 3.06 TB/s by rocprofv3).  Now <= 80
// VGPRs: three 512-thread blocks share a CU and one block's loads and stores overlap another's float64 arithmetic -
// 4.34 TB/s (profiles/r04_mixing_rocprof.json).  Forcing 64 VGPRs (four blocks per CU, one spilled register) measured
// no better (4.1-4.3 TB/s): what is left is the ~28 VALU instructions per element of the float64 chain.
// STUDY (only instantiated with -DLB_STUDY_BUILD, tools/slerp_study.py): 0 = product; 1 = lerp weights instead of the
// acos / sin chain; 2 = fp32 weighted sum (NOT exact)
template <int VPT, int STUDY = 0>
__global__ void __launch_bounds__(512) slerp_strided_kernel(const f16* __restrict__ p0, long stride0,
                                                             const f16* __restrict__ p1, long stride1,
                                                             f16* __restrict__ out, const double* __restrict__ fracts,
                                                             long n) {
    __shared__ double red[64];
    const long b = blockIdx.x;
    const f16* a0 = p0 + b * stride0;
    const f16* b0 = p1 + b * stride1;
    f16* o = out + b * n;
    const long nvec = n >> 3;
    f16x8 va[VPT], vb[VPT];
    const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
        const long i = threadIdx.x + (long)j * 512;
        va[j] = i < nvec ? *reinterpret_cast<const f16x8*>(a0 + i * 8) : zero8;
        vb[j] = i < nvec ? *reinterpret_cast<const f16x8*>(b0 + i * 8) : zero8;
    }
    double dot = 0, s0 = 0, s1 = 0;
#pragma unroll
    for (int j = 0; j < VPT; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const double x = (double)va[j][e], y = (double)vb[j][e];
            dot += x * y; s0 += x * x; s1 += y * y;
        }
    block_sum3(dot, s0, s1, red);
    double w0, w1;
    if (STUDY == 1) { w1 = fracts[b] + 1e-30 * dot; w0 = 1.0 - w1; }
    else slerp_weights_block(dot, s0, s1, fracts[b], w0, w1, red);
#pragma unroll
    for (int j = 0; j < VPT; ++j) asm volatile("" : "+v"(va[j]), "+v"(vb[j]));       // (see above: re-convert, do not keep)
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
        const long i = threadIdx.x + (long)j * 512;
        if (i < nvec) {
            f16x8 r;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (STUDY == 2) { r[e] = (f16)((float)va[j][e] * (float)w0 + (float)vb[j][e] * (float)w1); continue; }
                r[e] = lb_f64_to_f16(__dadd_rn(__dmul_rn((double)va[j][e], w0), __dmul_rn((double)vb[j][e], w1)));
            }
            *reinterpret_cast<f16x8*>(o + i * 8) = r;
        }
    }
}

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

struct SlerpChunk { SlerpPairs args; int cnt; bool vec_ok; };

template <typename T>
static int slerp_chunks_impl(const std::vector<SlerpChunk>& chunks, long n, hipStream_t stream) {
    constexpr int VEC = 16 / sizeof(T) > 8 ? 8 : 16 / sizeof(T);
    for (const SlerpChunk& c : chunks) {
        if (c.vec_ok)
            hipLaunchKernelGGL((slerp_pairs_kernel<T, VEC>), dim3(c.cnt), dim3(512), 0, stream, c.args, n);
        else
            hipLaunchKernelGGL((slerp_pairs_kernel<T, 1>), dim3(c.cnt), dim3(512), 0, stream, c.args, n);
        int rc = lb_check_launch("lb_slerp_pairs");
        if (rc) return rc;
    }
    return 0;
}

static std::vector<SlerpChunk> slerp_make_chunks(const void* const* p0, const void* const* p1,
                                                 void* const* out, const double* fracts, int npairs) {
    std::vector<SlerpChunk> chunks;
    for (int base = 0; base < npairs; base += LB_MAX_PAIRS) {
        SlerpChunk c;
        c.cnt = npairs - base < LB_MAX_PAIRS ? npairs - base : LB_MAX_PAIRS;
        c.vec_ok = true;
        for (int i = 0; i < c.cnt; ++i) {
            c.args.p0[i] = p0[base + i]; c.args.p1[i] = p1[base + i]; c.args.out[i] = out[base + i];
            c.args.fract[i] = fracts[base + i];
            c.vec_ok = c.vec_ok && aligned16(c.args.p0[i]) && aligned16(c.args.p1[i]) && aligned16(c.args.out[i]);
        }
        chunks.push_back(c);
    }
    return chunks;
}

extern "C" int lb_slerp_pairs_f16(const void* const* p0, const void* const* p1, void* const* out,
                                  const double* fracts, int npairs, long n, void* stream) {
    LB_REQUIRE(npairs >= 0 && n > 0, "lb_slerp_pairs_f16: bad sizes");
    const std::vector<SlerpChunk> chunks = slerp_make_chunks(p0, p1, out, fracts, npairs);
    LB_DISPATCH("lb_slerp_pairs_f16", slerp_chunks_impl<f16>(chunks, n, s));
}

extern "C" int lb_slerp_pairs_f32(const void* const* p0, const void* const* p1, void* const* out,
                                  const double* fracts, int npairs, long n, void* stream) {
    LB_REQUIRE(npairs >= 0 && n > 0, "lb_slerp_pairs_f32: bad sizes");
    const std::vector<SlerpChunk> chunks = slerp_make_chunks(p0, p1, out, fracts, npairs);
    LB_DISPATCH("lb_slerp_pairs_f32", slerp_chunks_impl<float>(chunks, n, s));
}

extern "C" int lb_slerp_pairs_f64(const void* const* p0, const void* const* p1, void* const* out,
                                  const double* fracts, int npairs, long n, void* stream) {
    LB_REQUIRE(npairs >= 0 && n > 0, "lb_slerp_pairs_f64: bad sizes");
    const std::vector<SlerpChunk> chunks = slerp_make_chunks(p0, p1, out, fracts, npairs);
    LB_DISPATCH("lb_slerp_pairs_f64", slerp_chunks_impl<double>(chunks, n, s));
}

static int slerp_strided_impl(const void* p0, long stride0, const void* p1, long stride1, void* out,
                              const double* fracts_dev, long npairs, long n, hipStream_t stream);

static int slerp_batched_impl(const void* p0, const void* p1, void* out, const double* fracts_dev,
                              long npairs, long n, hipStream_t stream) {
    if (n <= 512 * 8 * 8) return slerp_strided_impl(p0, n, p1, n, out, fracts_dev, npairs, n, stream);
    const size_t stage_bytes = (size_t)n * 2 * sizeof(f16);
    const int staged = stage_bytes <= 128 * 1024 ? 1 : 0;
    const size_t smem = 512 + (staged ? stage_bytes : 0);
    auto kern = slerp_batched_kernel<f16, 8>;
    static unsigned long long seen = 0;
    LB_ONCE_PER_DEVICE(seen)
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 512 + 128 * 1024);
    hipLaunchKernelGGL(kern, dim3((unsigned)npairs), dim3(512), smem, stream,
                       (const f16*)p0, (const f16*)p1, (f16*)out, fracts_dev, n, staged);
    return lb_check_launch("lb_slerp_batched_f16");
}

#ifdef LB_STUDY_BUILD
static int g_slerp_study = 0;
extern "C" void lb_slerp_set_study(int v) { g_slerp_study = v; }
#endif

// n > 32768 elements per pair (latents beyond 90 x 90): the pair does not fit the register-staged kernel; two passes over
// global memory (the second one hits L2: a pair is <= a few hundred KiB), same arithmetic.
__global__ void __launch_bounds__(512) slerp_strided_big_kernel(const f16* __restrict__ p0, long stride0,
                                                                 const f16* __restrict__ p1, long stride1,
                                                                 f16* __restrict__ out, const double* __restrict__ fracts,
                                                                 long n) {
    __shared__ double red[64];
    const long b = blockIdx.x;
    slerp_body<f16, 8>(p0 + b * stride0, p1 + b * stride1, out + b * n, n, fracts[b], nullptr, nullptr, false, red);
}

static int slerp_strided_impl(const void* p0, long stride0, const void* p1, long stride1, void* out,
                              const double* fracts_dev, long npairs, long n, hipStream_t stream) {
    const long nvec = n >> 3;
    const dim3 grid((unsigned)npairs), block(512);
    if (nvec > 512 * 8) {
        hipLaunchKernelGGL(slerp_strided_big_kernel, grid, block, 0, stream, (const f16*)p0, stride0, (const f16*)p1, stride1,
                           (f16*)out, fracts_dev, n);
        return lb_check_launch("lb_slerp_strided_f16(two-pass)");
    }
#ifdef LB_STUDY_BUILD
    if (g_slerp_study && nvec > 1024 && nvec <= 2048) {      // bottleneck studies on the L = 64 latent size only
        if (g_slerp_study == 1)
            hipLaunchKernelGGL((slerp_strided_kernel<4, 1>), grid, block, 0, stream, (const f16*)p0, stride0, (const f16*)p1, stride1,
                               (f16*)out, fracts_dev, n);
        else
            hipLaunchKernelGGL((slerp_strided_kernel<4, 2>), grid, block, 0, stream, (const f16*)p0, stride0, (const f16*)p1, stride1,
                               (f16*)out, fracts_dev, n);
        return lb_check_launch("lb_slerp_strided_f16(study)");
    }
#endif
#define LB_SLERP_STRIDED(V) hipLaunchKernelGGL((slerp_strided_kernel<V>), grid, block, 0, stream, (const f16*)p0, stride0, \
                                               (const f16*)p1, stride1, (f16*)out, fracts_dev, n)
    if (nvec <= 512) LB_SLERP_STRIDED(1);
    else if (nvec <= 1024) LB_SLERP_STRIDED(2);
    else if (nvec <= 2048) LB_SLERP_STRIDED(4);
    else LB_SLERP_STRIDED(8);
#undef LB_SLERP_STRIDED
    return lb_check_launch("lb_slerp_strided_f16");
}

extern "C" int lb_slerp_strided_f16(const void* p0, long stride0, const void* p1, long stride1, void* out,
                                    const double* fracts_dev, long npairs, long n, void* stream) {
    LB_REQUIRE(npairs > 0 && n > 0 && n % 8 == 0, "lb_slerp_strided_f16: n must be a positive multiple of 8");
    LB_REQUIRE(stride0 % 8 == 0 && stride1 % 8 == 0 && stride0 >= 0 && stride1 >= 0, "lb_slerp_strided_f16: strides multiples of 8");
    LB_REQUIRE(aligned16(p0) && aligned16(p1) && aligned16(out), "lb_slerp_strided_f16: 16-B alignment");
    LB_DISPATCH("lb_slerp_strided_f16", slerp_strided_impl(p0, stride0, p1, stride1, out, fracts_dev, npairs, n, s));
}

extern "C" int lb_slerp_batched_f16(const void* p0, const void* p1, void* out,
                                    const double* fracts_dev, long npairs, long n, void* stream) {
    LB_REQUIRE(npairs > 0 && n > 0 && n % 8 == 0, "lb_slerp_batched_f16: n must be a multiple of 8");
    LB_REQUIRE(aligned16(p0) && aligned16(p1) && aligned16(out), "lb_slerp_batched_f16: 16-B alignment");
    LB_DISPATCH("lb_slerp_batched_f16", slerp_batched_impl(p0, p1, out, fracts_dev, npairs, n, s));
}

// ------------------------------------------------------------------------------------------
// lerp:  out = half(half(wa*p0) + half(wb*p1))  — the rounding sequence of
// "(1 - f) * p0 + f * p1" on fp16 tensors with fp32 op-math (utils.py:97).
// ------------------------------------------------------------------------------------------
__global__ void lerp_f16_kernel(const f16* __restrict__ p0, const f16* __restrict__ p1,
                                f16* __restrict__ out, long n, float wa, float wb) {
    const long nvec = n >> 3;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
        const f16x8 a = *reinterpret_cast<const f16x8*>(p0 + i * 8);
        const f16x8 b = *reinterpret_cast<const f16x8*>(p1 + i * 8);
        f16x8 r;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const f16 ta = (f16)((float)a[j] * wa);
            const f16 tb = (f16)((float)b[j] * wb);
            r[j] = (f16)((float)ta + (float)tb);
        }
        *reinterpret_cast<f16x8*>(out + i * 8) = r;
    }
    for (long i = (nvec << 3) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const f16 ta = (f16)((float)p0[i] * wa);
        const f16 tb = (f16)((float)p1[i] * wb);
        out[i] = (f16)((float)ta + (float)tb);
    }
}

__global__ void lerp_f32_kernel(const float* __restrict__ p0, const float* __restrict__ p1,
                                float* __restrict__ out, long n, float wa, float wb) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        out[i] = __fadd_rn(__fmul_rn(p0[i], wa), __fmul_rn(p1[i], wb));
}

static unsigned ew_grid(long work_items, int block) {
    long g = (work_items + block - 1) / block;
    if (g < 1) g = 1;
    if (g > 2048) g = 2048;   // grid-stride above 8 blocks/CU x 256 CUs
    return (unsigned)g;
}

extern "C" int lb_lerp_f16(const void* p0, const void* p1, void* out, long n, double fract,
                           void* stream) {
    LB_REQUIRE(n > 0, "lb_lerp_f16: n");
    LB_REQUIRE(aligned16(p0) && aligned16(p1) && aligned16(out), "lb_lerp_f16: 16-B alignment");
    LB_DISPATCH_STMT("lb_lerp_f16", hipLaunchKernelGGL(lerp_f16_kernel, dim3(ew_grid((n + 7) / 8, 256)), dim3(256), 0, s,
                                                  (const f16*)p0, (const f16*)p1, (f16*)out, n,
                                                  (float)(1.0 - fract), (float)fract));
}

extern "C" int lb_lerp_f32(const void* p0, const void* p1, void* out, long n, double fract,
                           void* stream) {
    LB_REQUIRE(n > 0, "lb_lerp_f32: n");
    LB_DISPATCH_STMT("lb_lerp_f32", hipLaunchKernelGGL(lerp_f32_kernel, dim3(ew_grid(n, 256)), dim3(256), 0, s,
                                                  (const float*)p0, (const float*)p1, (float*)out, n,
                                                  (float)(1.0 - fract), (float)fract));
}

// ------------------------------------------------------------------------------------------
// Scheduler kernels.  Per-sample parameters live in device memory so that a captured hipGraph
// can be replayed with new sigmas / guidance without re-instantiation:
//   params[b*8 + 0] sigma_from   [1] sigma_to (Euler) or sigma_down (ancestral)   [2] sigma_up
//   params[b*8 + 3] guidance scale   [4] dt = fp32(sigma_next - sigma_from), computed in double
//   on the host exactly like the Python-float arithmetic of the diffusers schedulers.
// ------------------------------------------------------------------------------------------
#define LB_STEP_STRIDE 8

// Both kernels: 16 B per lane (8 halves), blockIdx.y walks the samples so the per-sample scalars are loaded once per
// block (no per-element division / parameter fetch); a scalar tail covers per_sample % 8 and unaligned buffers.

// x_in = half(x / sqrt(sigma^2 + 1)); with `dup` the batch is written twice (CFG: [uncond | cond]).
template <bool VEC>
__global__ void __launch_bounds__(256) scale_input_kernel(const f16* __restrict__ x, f16* __restrict__ out,
                                                           const float* __restrict__ params, long per_sample, int batch,
                                                           int dup) {
    const long total = per_sample * batch;
    for (int b = blockIdx.y; b < batch; b += gridDim.y) {
        const float s = params[b * LB_STEP_STRIDE + 0];
        const float denom = sqrtf(s * s + 1.0f);
        const f16* xs = x + (long)b * per_sample;
        f16* os = out + (long)b * per_sample;
        const long nvec = VEC ? per_sample >> 3 : 0;
        const long stride = (long)gridDim.x * blockDim.x;
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
            const f16x8 v = *reinterpret_cast<const f16x8*>(xs + i * 8);
            f16x8 r;
#pragma unroll
            for (int j = 0; j < 8; ++j) r[j] = (f16)__fdiv_rn((float)v[j], denom);
            *reinterpret_cast<f16x8*>(os + i * 8) = r;
            if (dup) *reinterpret_cast<f16x8*>(os + total + i * 8) = r;
        }
        for (long i = (nvec << 3) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < per_sample; i += stride) {
            const f16 v = (f16)__fdiv_rn((float)xs[i], denom);
            os[i] = v;
            if (dup) os[i + total] = v;
        }
    }
}

static dim3 sample_grid(long per_sample, int batch) {
    long gx = ((per_sample + 7) / 8 + 255) / 256;
    if (gx < 1) gx = 1;
    if (gx > 64) gx = 64;
    return dim3((unsigned)gx, (unsigned)(batch < 32768 ? batch : 32768), 1);
}

extern "C" int lb_scale_model_input_f16(const void* x, void* out, const float* params_dev,
                                        long per_sample, int batch, int dup_for_cfg, void* stream) {
    LB_REQUIRE(per_sample > 0 && batch > 0, "lb_scale_model_input_f16: sizes");
    const bool vec = per_sample % 8 == 0 && aligned16(x) && aligned16(out);
    LB_DISPATCH_STMT("lb_scale_model_input_f16",
                     if (vec) hipLaunchKernelGGL(scale_input_kernel<true>, sample_grid(per_sample, batch), dim3(256), 0, s,
                                                 (const f16*)x, (f16*)out, params_dev, per_sample, batch, dup_for_cfg);
                     else hipLaunchKernelGGL(scale_input_kernel<false>, sample_grid(per_sample, batch), dim3(256), 0, s,
                                             (const f16*)x, (f16*)out, params_dev, per_sample, batch, dup_for_cfg));
}

// eps layout: cfg == 0: eps[b] ; cfg == 1: eps[0..B) = uncond, eps[B..2B) = text.
// CFG combine reproduces the fp16 tensor arithmetic of diffusers_holder.py:348-349
// (sub, scalar mul, add — each rounded to fp16); the update itself is fp32, rounded once.
__device__ __forceinline__ f16 euler_one(f16 xh, f16 eu, f16 et, f16 nz, float s_from, float dt, float s_up, float g,
                                         bool cfg, bool ancestral) {
    f16 e = eu;
    if (cfg) {
        const f16 diff = (f16)((float)et - (float)eu);
        const f16 sc = (f16)(g * (float)diff);
        e = (f16)((float)eu + (float)sc);
    }
    const float xf = (float)xh;
    const float x0 = __fsub_rn(xf, __fmul_rn(s_from, (float)e));
    const float d = __fdiv_rn(__fsub_rn(xf, x0), s_from);
    float nxt = __fadd_rn(xf, __fmul_rn(d, dt));
    if (ancestral) nxt = __fadd_rn(nxt, __fmul_rn((float)nz, s_up));
    return (f16)nxt;
}

template <bool VEC, bool CFG, bool ANC>
__global__ void __launch_bounds__(256) euler_step_kernel(const f16* __restrict__ x, const f16* __restrict__ eps,
                                                          const f16* __restrict__ noise, f16* __restrict__ out,
                                                          const float* __restrict__ params, long per_sample, int batch) {
    const long total = per_sample * batch;
    for (int b = blockIdx.y; b < batch; b += gridDim.y) {
        const float s_from = params[b * LB_STEP_STRIDE + 0];
        const float dt = params[b * LB_STEP_STRIDE + 4];
        const float s_up = params[b * LB_STEP_STRIDE + 2];
        const float g = params[b * LB_STEP_STRIDE + 3];
        const long base = (long)b * per_sample;
        const long nvec = VEC ? per_sample >> 3 : 0;
        const long stride = (long)gridDim.x * blockDim.x;
        const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
            const long o = base + i * 8;
            const f16x8 xv = *reinterpret_cast<const f16x8*>(x + o);
            const f16x8 eu = *reinterpret_cast<const f16x8*>(eps + o);
            const f16x8 et = CFG ? *reinterpret_cast<const f16x8*>(eps + total + o) : zero8;
            const f16x8 nz = ANC ? *reinterpret_cast<const f16x8*>(noise + o) : zero8;
            f16x8 r;
#pragma unroll
            for (int j = 0; j < 8; ++j) r[j] = euler_one(xv[j], eu[j], et[j], nz[j], s_from, dt, s_up, g, CFG, ANC);
            *reinterpret_cast<f16x8*>(out + o) = r;
        }
        for (long i = (nvec << 3) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < per_sample; i += stride) {
            const long o = base + i;
            out[o] = euler_one(x[o], eps[o], CFG ? eps[o + total] : (f16)0.f, ANC ? noise[o] : (f16)0.f, s_from, dt, s_up, g, CFG, ANC);
        }
    }
}

template <bool VEC>
static void euler_launch(const void* x, const void* eps, const void* noise, void* out, const float* params_dev,
                         long per_sample, int batch, int cfg, int ancestral, hipStream_t s) {
    const dim3 grid = sample_grid(per_sample, batch), block(256);
#define LB_EULER(C, A) hipLaunchKernelGGL((euler_step_kernel<VEC, C, A>), grid, block, 0, s, (const f16*)x, (const f16*)eps, \
                                          (const f16*)noise, (f16*)out, params_dev, per_sample, batch)
    if (cfg && ancestral) LB_EULER(true, true);
    else if (cfg) LB_EULER(true, false);
    else if (ancestral) LB_EULER(false, true);
    else LB_EULER(false, false);
#undef LB_EULER
}

extern "C" int lb_euler_step_f16(const void* x, const void* eps, const void* noise, void* out,
                                 const float* params_dev, long per_sample, int batch, int cfg,
                                 int ancestral, void* stream) {
    LB_REQUIRE(per_sample > 0 && batch > 0, "lb_euler_step_f16: sizes");
    LB_REQUIRE(!ancestral || noise != nullptr, "lb_euler_step_f16: ancestral step needs noise");
    const bool vec = per_sample % 8 == 0 && aligned16(x) && aligned16(eps) && aligned16(out) && (!ancestral || aligned16(noise));
    LB_DISPATCH_STMT("lb_euler_step_f16",
                     if (vec) euler_launch<true>(x, eps, noise, out, params_dev, per_sample, batch, cfg, ancestral, s);
                     else euler_launch<false>(x, eps, noise, out, params_dev, per_sample, batch, cfg, ancestral, s));
}

// ------------------------------------------------------------------------------------------------
// DDIM step (eta = 0, epsilon prediction; diffusers DDIMScheduler.step as SD / SDXL configure it: clip_sample = False,
// set_alpha_to_one = False, leading spacing with steps_offset = 1).  Third party, reached from
// /root/reference/latentblending/diffusers_holder.py:356 when the pipe carries a DDIM scheduler (the reference's own SDXL
// path constructs Euler schedulers, :42; north_star names "the Euler/DDIM step").
//   params row: {0 (sigma of lb_scale_model_input_f16: DDIM's scale_model_input is the identity, x / sqrt(0 + 1) = x),
//                sqrt(abar_t), sqrt(abar_prev), guidance, sqrt(1 - abar_t), sqrt(1 - abar_prev), 1 / sqrt(abar_t) [fp32], -}
// diffusers does NOT upcast here (Euler does): sample and model_output are fp16 tensors, the coefficients 0-dim fp32 tensors,
// so EVERY tensor operation rounds to fp16 -
//   x0   = (sample - sqrt(1 - abar_t) * eps) / sqrt(abar_t)          [mul, sub, division: three roundings.  The divisor is a 0-dim HOST
//                                                                     tensor: the device library's true-division kernel then computes
//                                                                     a * (1 / b) with the fp32 reciprocal formed once - so does this
//                                                                     kernel (slot 6 of the row), not a correctly rounded a / b]
//   dir  = sqrt(1 - abar_prev) * eps                                 [one]
//   prev = sqrt(abar_prev) * x0 + dir                                [mul, add: two]
// and this kernel rounds in the same six places (CFG combine as in euler_one).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ f16 ddim_one(f16 xh, f16 eu, f16 et, float inv_sa_t, float sa_p, float sb_t, float sb_p, float g, bool cfg) {
    f16 e = eu;
    if (cfg) {
        const f16 diff = (f16)((float)et - (float)eu);
        const f16 sc = (f16)(g * (float)diff);
        e = (f16)((float)eu + (float)sc);
    }
    // every fp32 result passes through an opaque register before it is rounded to fp16: hipcc otherwise fuses "multiply, then
    // round" into v_fma_mix{lo,hi}_f16, which rounds the exact product ONCE - a tensor library (and the oracle) rounds twice,
    // fp32 then fp16 (measured on the device: differences of a few fp16 ulps where the final sum cancels)
    auto r16 = [](float v) {
        asm volatile("" : "+v"(v));
        return (f16)v;
    };
    const f16 t1 = r16(__fmul_rn(sb_t, (float)e));
    const f16 t2 = r16(__fsub_rn((float)xh, (float)t1));
    const f16 x0 = r16(__fmul_rn((float)t2, inv_sa_t));
    const f16 dir = r16(__fmul_rn(sb_p, (float)e));
    const f16 t3 = r16(__fmul_rn(sa_p, (float)x0));
    return r16(__fadd_rn((float)t3, (float)dir));
}

template <bool VEC, bool CFG>
__global__ void __launch_bounds__(256) ddim_step_kernel(const f16* __restrict__ x, const f16* __restrict__ eps, f16* __restrict__ out,
                                                         const float* __restrict__ params, long per_sample, int batch) {
    const long total = per_sample * batch;
    for (int b = blockIdx.y; b < batch; b += gridDim.y) {
        const float sa_t = params[b * LB_STEP_STRIDE + 6], sa_p = params[b * LB_STEP_STRIDE + 2], g = params[b * LB_STEP_STRIDE + 3];      // (sa_t: the RECIPROCAL 1 / sqrt(abar_t))
        const float sb_t = params[b * LB_STEP_STRIDE + 4], sb_p = params[b * LB_STEP_STRIDE + 5];
        const long base = (long)b * per_sample;
        const long nvec = VEC ? per_sample >> 3 : 0;
        const long stride = (long)gridDim.x * blockDim.x;
        const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
            const long o = base + i * 8;
            const f16x8 xv = *reinterpret_cast<const f16x8*>(x + o);
            const f16x8 eu = *reinterpret_cast<const f16x8*>(eps + o);
            const f16x8 et = CFG ? *reinterpret_cast<const f16x8*>(eps + total + o) : zero8;
            f16x8 r;
#pragma unroll
            for (int j = 0; j < 8; ++j) r[j] = ddim_one(xv[j], eu[j], et[j], sa_t, sa_p, sb_t, sb_p, g, CFG);
            *reinterpret_cast<f16x8*>(out + o) = r;
        }
        for (long i = (nvec << 3) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < per_sample; i += stride) {
            const long o = base + i;
            out[o] = ddim_one(x[o], eps[o], CFG ? eps[o + total] : (f16)0.f, sa_t, sa_p, sb_t, sb_p, g, CFG);
        }
    }
}

extern "C" int lb_ddim_step_f16(const void* x, const void* eps, void* out, const float* params_dev, long per_sample, int batch,
                                int cfg, void* stream) {
    LB_REQUIRE(per_sample > 0 && batch > 0, "lb_ddim_step_f16: sizes");
    const bool vec = per_sample % 8 == 0 && aligned16(x) && aligned16(eps) && aligned16(out);
    const dim3 grid = sample_grid(per_sample, batch), block(256);
#define LB_DDIM(V, C) hipLaunchKernelGGL((ddim_step_kernel<V, C>), grid, block, 0, s, (const f16*)x, (const f16*)eps, (f16*)out, params_dev, per_sample, batch)
    LB_DISPATCH_STMT("lb_ddim_step_f16",
                     if (vec && cfg) LB_DDIM(true, true); else if (vec) LB_DDIM(true, false);
                     else if (cfg) LB_DDIM(false, true); else LB_DDIM(false, false));
#undef LB_DDIM
}

