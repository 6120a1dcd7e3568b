// GroupNorm(+SiLU) on NHWC activations and row LayerNorm for gfx950.
//
// GroupNorm is two launches with a deterministic reduction tree (no float atomics):
//   1. gn_partial: grid (pixel chunks, B).  Every thread owns a FIXED 8-channel vector and walks
//      the chunk's pixels with coalesced 16-B loads; per-channel sums meet in LDS and one thread
//      per group folds them in float64 -> partial[b][chunk][group] = (sum, sumsq).
//   2. gn_apply:   grid (blocks, B).  Prologue folds the chunk partials of its sample into
//      mean / rstd (float64), then a coalesced normalise * gamma + beta (+SiLU) pass writes fp16.
// Input may be fp16 (UNet) or fp32 (VAE residual stream).
//
// Replaces torch.nn.GroupNorm / LayerNorm inside diffusers' ResnetBlock2D, Transformer2DModel,
// BasicTransformerBlock and the VAE decoder (reached from
// /root/reference/latentblending/diffusers_holder.py:336 and :135).
#include "lb_common.h"

#define GN_MAX_GROUPS 32
#define GN_MAX_CHUNKS 256

template <typename T> struct Vec8 { typedef T type __attribute__((ext_vector_type(8))); };

template <typename T>
__device__ __forceinline__ void load8(const T* p, float (&v)[8]);
template <>
__device__ __forceinline__ void load8<f16>(const f16* p, float (&v)[8]) {
    const f16x8 x = *reinterpret_cast<const f16x8*>(p);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (float)x[j];
}
template <>
__device__ __forceinline__ void load8<float>(const float* p, float (&v)[8]) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p);
    const f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[j] = a[j]; v[4 + j] = b[j]; }
}

// x: [B][HW][ldx]; partial: [B][nchunk][groups][2] doubles
template <typename T, int VPT>
__global__ void __launch_bounds__(256) gn_partial_kernel(const T* __restrict__ x,
                                                         double* __restrict__ partial, int HW, int C,
                                                         int ldx, int groups, int chunk_px) {
    extern __shared__ __attribute__((aligned(16))) float gn_lds[];   // [2][rows][C]
    const int vecs = C >> 3;
    const int rows = VPT == 1 ? 256 / vecs : 1;
    const int tid = threadIdx.x;
    const int chunk = blockIdx.x, b = blockIdx.y, nchunk = gridDim.x;
    const int px_begin = chunk * chunk_px;
    int px_end = px_begin + chunk_px;
    if (px_end > HW) px_end = HW;
    const int r = VPT == 1 ? tid / vecs : 0;
    const bool active = VPT == 1 ? tid < rows * vecs : true;
    float* lsum = gn_lds;
    float* lsq = gn_lds + rows * C;

#pragma unroll
    for (int j = 0; j < VPT; ++j) {
        const int v = VPT == 1 ? tid % vecs : tid + j * 256;
        float s[8] = {0, 0, 0, 0, 0, 0, 0, 0}, q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (active && v < vecs) {
            const T* base = x + ((long)b * HW) * ldx + v * 8;
            int px = px_begin + r;
            for (; px + 3 * rows < px_end; px += 4 * rows) {     // four independent 16-B requests in flight
                float v0[8], v1[8], v2[8], v3[8];
                load8<T>(base + (long)px * ldx, v0);
                load8<T>(base + (long)(px + rows) * ldx, v1);
                load8<T>(base + (long)(px + 2 * rows) * ldx, v2);
                load8<T>(base + (long)(px + 3 * rows) * ldx, v3);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    s[e] += v0[e]; q[e] += v0[e] * v0[e];
                    s[e] += v1[e]; q[e] += v1[e] * v1[e];
                    s[e] += v2[e]; q[e] += v2[e] * v2[e];
                    s[e] += v3[e]; q[e] += v3[e] * v3[e];
                }
            }
            for (; px < px_end; px += rows) {
                float val[8];
                load8<T>(base + (long)px * ldx, val);
#pragma unroll
                for (int e = 0; e < 8; ++e) { s[e] += val[e]; q[e] += val[e] * val[e]; }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                lsum[r * C + v * 8 + e] = s[e];
                lsq[r * C + v * 8 + e] = q[e];
            }
        }
    }
    __syncthreads();
    if (tid < groups) {
        const int cpg = C / groups;
        double s = 0, q = 0;
        for (int rr = 0; rr < rows; ++rr)
            for (int c = tid * cpg; c < (tid + 1) * cpg; ++c) {
                s += (double)lsum[rr * C + c];
                q += (double)lsq[rr * C + c];
            }
        double* out = partial + (((long)b * nchunk + chunk) * groups + tid) * 2;
        out[0] = s;
        out[1] = q;
    }
}

// Apply pass.  Like the statistics pass every thread owns a FIXED 8-channel vector (v = tid % vecs, pixel row
// r = tid / vecs) and walks the pixels of its chunk: the affine transform of its channels is folded ONCE into
// y = x * scale + shift (scale = rstd * gamma, shift = beta - mean * scale; 16 registers), so the streaming loop is
// one 16-B load, 8 FMAs (+ SiLU) and one 16-B store per vector.  (Round 1 re-read gamma / beta / mean / rstd per
// vector: four extra VMEM requests and 16 LDS reads per 16 bytes of payload - the pass ran at 2.4 TB/s.)
template <typename T>
__global__ void __launch_bounds__(256) gn_apply_kernel(const T* __restrict__ x,
                                                       const double* __restrict__ partial,
                                                       const float* __restrict__ gamma,
                                                       const float* __restrict__ beta,
                                                       f16* __restrict__ y, int HW, int C, int ldx,
                                                       int ldy, int groups, int nchunk, float eps,
                                                       int silu, int chunk_px) {
    __shared__ float mean_s[GN_MAX_GROUPS], rstd_s[GN_MAX_GROUPS];
    __shared__ double fold_s[8][GN_MAX_GROUPS], fold_q[8][GN_MAX_GROUPS];
    const int b = blockIdx.y, tid = threadIdx.x;
    const int cpg = C / groups;
    {   // fold the chunk partials of this sample: 8 interleaved subsets per group, then a fixed-order sum
        const int grp = tid & 31, sub = tid >> 5;
        if (grp < groups) {
            double s = 0, q = 0;
            for (int c = sub; c < nchunk; c += 8) {
                const double* pp = partial + (((long)b * nchunk + c) * groups + grp) * 2;
                s += pp[0];
                q += pp[1];
            }
            fold_s[sub][grp] = s;
            fold_q[sub][grp] = q;
        }
    }
    __syncthreads();
    if (tid < groups) {
        double s = 0, q = 0;
#pragma unroll
        for (int sub = 0; sub < 8; ++sub) { s += fold_s[sub][tid]; q += fold_q[sub][tid]; }
        const double cnt = (double)HW * cpg;
        const double mean = s / cnt;
        double var = q / cnt - mean * mean;
        if (var < 0) var = 0;
        mean_s[tid] = (float)mean;
        rstd_s[tid] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
    const int vecs = C >> 3;
    const int rows = vecs <= 256 ? 256 / vecs : 1;
    const int px_begin = blockIdx.x * chunk_px;
    int px_end = px_begin + chunk_px;
    if (px_end > HW) px_end = HW;
    const int r = vecs <= 256 ? tid / vecs : 0;
    if (vecs <= 256 && tid >= rows * vecs) return;
    for (int v = vecs <= 256 ? tid % vecs : tid; v < vecs; v += 256) {
        const int c0 = v * 8;
        float sc[8], sh[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int grp = (c0 + e) / cpg;
            const float a = rstd_s[grp] * gamma[c0 + e];
            sc[e] = a;
            sh[e] = beta[c0 + e] - mean_s[grp] * a;
        }
        const T* src = x + ((long)b * HW) * ldx + c0;
        f16* dst = y + ((long)b * HW) * ldy + c0;
        auto finish = [&](const float (&val)[8], int px) {
            f16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float t = __builtin_fmaf(val[e], sc[e], sh[e]);
                if (silu) t = lb_silu(t);
                o[e] = (f16)t;
            }
            *reinterpret_cast<f16x8*>(dst + (long)px * ldy) = o;
        };
        int px = px_begin + r;
        for (; px + 3 * rows < px_end; px += 4 * rows) {          // four independent 16-B requests in flight
            float v0[8], v1[8], v2[8], v3[8];
            load8<T>(src + (long)px * ldx, v0);
            load8<T>(src + (long)(px + rows) * ldx, v1);
            load8<T>(src + (long)(px + 2 * rows) * ldx, v2);
            load8<T>(src + (long)(px + 3 * rows) * ldx, v3);
            finish(v0, px);
            finish(v1, px + rows);
            finish(v2, px + 2 * rows);
            finish(v3, px + 3 * rows);
        }
        for (; px < px_end; px += rows) {
            float val[8];
            load8<T>(src + (long)px * ldx, val);
            finish(val, px);
        }
        if (vecs <= 256) break;
    }
}

extern "C" long lb_groupnorm_workspace_bytes(int B, int groups) {
    return (long)B * GN_MAX_CHUNKS * groups * 2 * (long)sizeof(double);
}

// x: [B][HW][ldx] (fp16, or fp32 when x_is_f32), y: [B][HW][ldy] fp16, gamma/beta fp32 [C]
static int groupnorm_impl(const void* x, void* y, const float* gamma, const float* beta,
                          void* workspace, int B, int HW, int C, int ldx, int ldy, int groups,
                          float eps, int silu, int x_is_f32, hipStream_t stream) {
    LB_REQUIRE(C % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0, "lb_groupnorm_nhwc: C/ld multiple of 8");
    LB_REQUIRE(groups > 0 && groups <= GN_MAX_GROUPS && C % groups == 0, "lb_groupnorm_nhwc: groups");
    const int vecs = C / 8;
    // ~2048 statistic blocks per launch (8 per CU), at least 32 pixels and at most GN_MAX_CHUNKS chunks per sample
    int want = (2048 + B - 1) / B;
    if (want > GN_MAX_CHUNKS) want = GN_MAX_CHUNKS;
    if (want < 16) want = 16;
    int chunk_px = (HW + want - 1) / want;
    if (chunk_px < 32) chunk_px = 32;
    const int nchunk = (HW + chunk_px - 1) / chunk_px;
    double* partial = (double*)workspace;
    const int vpt = vecs <= 256 ? 1 : 2;
    const int rows = vpt == 1 ? 256 / vecs : 1;
    const size_t lds = (size_t)2 * rows * C * sizeof(float);
    dim3 grid1(nchunk, B);
#define GN_PART(T, V) hipLaunchKernelGGL((gn_partial_kernel<T, V>), grid1, dim3(256), lds, stream, \
                                         (const T*)x, partial, HW, C, ldx, groups, chunk_px)
    if (x_is_f32) { if (vpt == 1) GN_PART(float, 1); else GN_PART(float, 2); }
    else          { if (vpt == 1) GN_PART(f16, 1);   else GN_PART(f16, 2); }
#undef GN_PART
    int rc = lb_check_launch("lb_groupnorm_nhwc(partial)");
    if (rc) return rc;
    // apply pass: ~4096 blocks per launch (finer chunks than the statistics pass: no partials to fold per chunk)
    int want2 = (4096 + B - 1) / B;
    int apx = (HW + want2 - 1) / want2;
    const int min_px = 4 * rows;
    if (apx < min_px) apx = min_px;
    const int achunks = (HW + apx - 1) / apx;
    dim3 grid2((unsigned)achunks, B);
    if (x_is_f32)
        hipLaunchKernelGGL((gn_apply_kernel<float>), grid2, dim3(256), 0, stream, (const float*)x,
                           partial, gamma, beta, (f16*)y, HW, C, ldx, ldy, groups, nchunk, eps, silu, apx);
    else
        hipLaunchKernelGGL((gn_apply_kernel<f16>), grid2, dim3(256), 0, stream, (const f16*)x,
                           partial, gamma, beta, (f16*)y, HW, C, ldx, ldy, groups, nchunk, eps, silu, apx);
    return lb_check_launch("lb_groupnorm_nhwc(apply)");
}

// Samples per statistics+apply pair (experiment, OFF by default).  The apply pass re-reads what the statistics pass just
// read; the idea was to walk a big batch (VAE decode at 512^2) in sample groups that fit the 256 MiB Infinity Cache so
// that the second read never reaches HBM.  Measured (profiles/r02_groupnorm_l3.txt): every group size from 32 to 192 MiB
// is SLOWER than one pair over the whole batch (17x512x512x128: 647 us whole batch, 784-1045 us grouped) - the smaller
// launches lose more memory-level parallelism than the cache returns.  0 = whole batch.
static long g_gn_l3_bytes = 0;
extern "C" void lb_groupnorm_set_l3_chunk(long bytes) { g_gn_l3_bytes = bytes; }

static int groupnorm_chunked(const void* x, void* y, const float* gamma, const float* beta, void* workspace,
                             int B, int HW, int C, int ldx, int ldy, int groups, float eps, int silu,
                             int x_is_f32, hipStream_t stream) {
    const long elt = x_is_f32 ? 4 : 2;
    const long sample_bytes = (long)HW * ldx * elt;
    long nb = B;
    if (g_gn_l3_bytes > 0 && sample_bytes * B > g_gn_l3_bytes) {
        nb = g_gn_l3_bytes / sample_bytes;
        if (nb < 1) nb = 1;
    }
    for (long b0 = 0; b0 < B; b0 += nb) {
        const int bn = (int)(B - b0 < nb ? B - b0 : nb);
        const int rc = groupnorm_impl((const char*)x + b0 * sample_bytes, (f16*)y + b0 * (long)HW * ldy, gamma, beta,
                                      (double*)workspace + b0 * GN_MAX_CHUNKS * groups * 2, bn, HW, C, ldx, ldy,
                                      groups, eps, silu, x_is_f32, stream);
        if (rc) return rc;
    }
    return 0;
}

// ------------------------------------------------------------------------------------------
// Round 6: ONE-launch GroupNorm for feature maps whose (sample, channel slab) fits the registers of one block - the UNet's 16^2 and
// 32^2 levels (33 of its 46 GroupNorms; 1.3 ms per B = 17 forward and 0.8 ms per B = 2 forward as partial + apply launches).  A block owns
// one sample and a slab of SL = lcm(8, C / groups) channels (whole groups AND whole 16-byte vectors: 40, 80 or 120 channels), i.e.
// vps = SL / 8 vectors per pixel; thread (r = tid / vps, slot = tid % vps) keeps the vectors of pixels r, r + rows, ... (at most NPX of
// them, raw fp16) in registers.  Every request of the slab - and gamma / beta - is in flight at once; the statistics go per thread in fp32
// (<= NPX values per channel), then in float64 through LDS (rows, then the channels of a group), the same E[x^2] - E[x]^2 form in double
// as the two-launch path; the apply pass runs out of the registers: x is read ONCE, nothing but y is written.
// ------------------------------------------------------------------------------------------
template <int NPX>
__global__ void __launch_bounds__(256) gn_fused_kernel(const f16* __restrict__ x, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, f16* __restrict__ y, int HW, int C, int ldx,
                                                       int ldy, int cpg, int SL, float eps, int silu) {
    extern __shared__ __attribute__((aligned(16))) double gnf_lds[];       // [2][rows][SL] doubles, then [2][SL], then mean / rstd per group
    const int vps = SL >> 3, rows = 256 / vps;
    const int tid = threadIdx.x, b = blockIdx.y, c_slab = blockIdx.x * SL;
    const bool active = tid < rows * vps;
    const int r = active ? tid / vps : 0, slot = active ? tid - r * vps : 0;
    const int c0 = c_slab + slot * 8;
    const f16* src = x + ((long)b * HW) * ldx + c0;
    f16x8 raw[NPX];
#pragma unroll
    for (int k = 0; k < NPX; ++k) {             // (pixels past HW re-read the last one: weighted out of the sums, stored nowhere)
        const int px = min(r + k * rows, HW - 1);
        raw[k] = *reinterpret_cast<const f16x8*>(src + (long)px * ldx);
    }
    __builtin_amdgcn_sched_barrier(0);
    const f32x4 g0 = *reinterpret_cast<const f32x4*>(gamma + c0), g1 = *reinterpret_cast<const f32x4*>(gamma + c0 + 4);
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(beta + c0), b1 = *reinterpret_cast<const f32x4*>(beta + c0 + 4);
    __builtin_amdgcn_sched_barrier(0);
    float sm[8] = {0, 0, 0, 0, 0, 0, 0, 0}, sq[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < NPX; ++k) {
        const bool ok = active && r + k * rows < HW;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v = ok ? (float)raw[k][e] : 0.f;
            sm[e] += v;
            sq[e] += v * v;
        }
    }
    double* lsum = gnf_lds;                     // [rows][SL]
    double* lsq = gnf_lds + rows * SL;
    if (active) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            lsum[r * SL + slot * 8 + e] = (double)sm[e];
            lsq[r * SL + slot * 8 + e] = (double)sq[e];
        }
    }
    __syncthreads();
    double* csum = gnf_lds + 2 * rows * SL;     // [2][SL]: per channel, over the rows (fixed order)
    if (tid < 2 * SL) {
        const double* col = (tid < SL ? lsum : lsq) + (tid < SL ? tid : tid - SL);
        double a = 0;
        for (int rr = 0; rr < rows; ++rr) a += col[rr * SL];
        csum[tid] = a;
    }
    __syncthreads();
    const int ngrp = SL / cpg;                  // groups of this slab (1, 2 or 4)
    float* stat = reinterpret_cast<float*>(csum + 2 * SL);      // [ngrp][2]: mean, rstd
    if (tid < ngrp) {
        double a = 0, q = 0;
        for (int c = tid * cpg; c < (tid + 1) * cpg; ++c) { a += csum[c]; q += csum[SL + c]; }
        const double cnt = (double)HW * cpg;
        const double mean = a / cnt;
        double var = q / cnt - mean * mean;
        if (var < 0) var = 0;
        stat[2 * tid] = (float)mean;
        stat[2 * tid + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
    if (!active) return;
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int grp = (slot * 8 + e) / cpg;
        const float gam = e < 4 ? g0[e & 3] : g1[e & 3], bet = e < 4 ? b0[e & 3] : b1[e & 3];
        const float a = stat[2 * grp + 1] * gam;
        sc[e] = a;
        sh[e] = bet - stat[2 * grp] * a;
    }
    f16* dst = y + ((long)b * HW) * ldy + c0;
#pragma unroll
    for (int k = 0; k < NPX; ++k) {
        const int px = r + k * rows;
        f16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float t = __builtin_fmaf((float)raw[k][e], sc[e], sh[e]);
            if (silu) t = lb_silu(t);
            o[e] = (f16)t;
        }
        if (px < HW) *reinterpret_cast<f16x8*>(dst + (long)px * ldy) = o;
    }
}

static int g_gn_fused = 1;      // 1 = the one-launch form wherever a (sample, slab) fits a block's registers (round 6), 0 = always partial + apply
extern "C" void lb_groupnorm_set_fused(int on) { g_gn_fused = on; }

// slab width (channels) and pixels per thread of the one-launch form; 0 = not eligible
static int gn_fused_plan(int HW, int C, int groups, int x_is_f32, int* npx_out) {
    if (!g_gn_fused || x_is_f32) return 0;
    const int cpg = C / groups;
    int SL = cpg;
    while (SL % 8) SL += cpg;                   // lcm(8, cpg)
    if (SL > 120 || C % SL) return 0;
    const int rows = 256 / (SL / 8);
    const int npx = (HW + rows - 1) / rows;
    if (npx > 24) return 0;
    *npx_out = npx <= 8 ? 8 : (npx <= 16 ? 16 : 24);
    return SL;
}

extern "C" int lb_groupnorm_plan(int HW, int C, int groups, int x_is_f32) {       // 1 = lb_groupnorm_nhwc takes the one-launch form for this shape
    int npx = 0;
    return (groups > 0 && C > 0 && C % groups == 0 && C % 8 == 0 && gn_fused_plan(HW, C, groups, x_is_f32, &npx)) ? 1 : 0;
}

static int groupnorm_fused(const void* x, void* y, const float* gamma, const float* beta, int B, int HW, int C, int ldx, int ldy,
                           int groups, float eps, int silu, int SL, int npx, hipStream_t s) {
    const int cpg = C / groups, rows = 256 / (SL / 8);
    const size_t smem = (size_t)(2 * rows * SL + 2 * SL) * sizeof(double) + 16 * sizeof(float);
    const dim3 grid(C / SL, B);
#define LB_GNF_ARGS (const f16*)x, gamma, beta, (f16*)y, HW, C, ldx, ldy, cpg, SL, eps, silu
    if (npx == 8) hipLaunchKernelGGL(gn_fused_kernel<8>, grid, dim3(256), smem, s, LB_GNF_ARGS);
    else if (npx == 16) hipLaunchKernelGGL(gn_fused_kernel<16>, grid, dim3(256), smem, s, LB_GNF_ARGS);
    else hipLaunchKernelGGL(gn_fused_kernel<24>, grid, dim3(256), smem, s, LB_GNF_ARGS);
#undef LB_GNF_ARGS
    return lb_check_launch("lb_groupnorm_nhwc(fused)");
}

extern "C" int lb_groupnorm_nhwc(const void* x, void* y, const float* gamma, const float* beta,
                                 void* workspace, int B, int HW, int C, int ldx, int ldy, int groups,
                                 float eps, int silu, int x_is_f32, void* stream) {
    LB_REQUIRE(B > 0 && HW > 0 && C > 0, "lb_groupnorm_nhwc: sizes");
    LB_REQUIRE(C % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0, "lb_groupnorm_nhwc: C/ld multiple of 8");
    LB_REQUIRE(groups > 0 && groups <= GN_MAX_GROUPS && C % groups == 0, "lb_groupnorm_nhwc: groups");
    LB_REQUIRE(C <= 4096, "lb_groupnorm_nhwc: C <= 4096");
    int npx = 0;
    const int SL = gn_fused_plan(HW, C, groups, x_is_f32, &npx);
    if (SL) LB_DISPATCH("lb_groupnorm_nhwc", groupnorm_fused(x, y, gamma, beta, B, HW, C, ldx, ldy, groups, eps, silu, SL, npx, s));
    LB_DISPATCH("lb_groupnorm_nhwc", groupnorm_chunked(x, y, gamma, beta, workspace, B, HW, C, ldx, ldy, groups,
                                                       eps, silu, x_is_f32, s));
}

// ------------------------------------------------------------------------------------------
// GroupNorm from statistics the PRODUCING conv left behind (LB_GEMM_CH_STATS, conv3_halo.hip): ch_stats holds, per
// (channel, 64-pixel row block) - channel-major, so a group's channels are contiguous runs - (sum, sum of squares) of the
// stored values.  gn_fold_stats folds them per (sample, group) in float64 - thread t takes rows t, t + 256, ... of every
// channel of the group, then a fixed-order block reduction - into
// the SAME partial[b][0][group] slot layout gn_apply_kernel consumes (nchunk = 1): x is read once, by the apply pass.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) gn_fold_stats_kernel(const float2* __restrict__ ch_stats, double* __restrict__ partial,
                                                            int C, int groups, int rows_per_sample, long total_rows) {
    __shared__ double red_s[256], red_q[256];
    const int grp = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int cpg = C / groups;
    // channel-major buffer [C][total_rows]: channel c of sample b is the contiguous run [c][b * rows_per_sample ..]
    double s = 0, q = 0;
    for (int c = 0; c < cpg; ++c) {
        const float2* run = ch_stats + (long)(grp * cpg + c) * total_rows + (long)b * rows_per_sample;
        for (int r = tid; r < rows_per_sample; r += 256) {
            const float2 v = run[r];
            s += (double)v.x;
            q += (double)v.y;
        }
    }
    red_s[tid] = s;
    red_q[tid] = q;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (tid < w) { red_s[tid] += red_s[tid + w]; red_q[tid] += red_q[tid + w]; }
        __syncthreads();
    }
    if (tid == 0) {
        double* out = partial + ((long)b * groups + grp) * 2;
        out[0] = red_s[0];
        out[1] = red_q[0];
    }
}

static int groupnorm_from_stats_impl(const void* x, void* y, const float* gamma, const float* beta, const float* ch_stats,
                                     void* workspace, int B, int HW, int C, int ldx, int ldy, int groups, float eps, int silu,
                                     int x_is_f32, int rows_per_sample, hipStream_t stream) {
    double* partial = (double*)workspace;
    hipLaunchKernelGGL(gn_fold_stats_kernel, dim3(groups, B), dim3(256), 0, stream, (const float2*)ch_stats, partial, C, groups,
                       rows_per_sample, (long)B * rows_per_sample);
    int rc = lb_check_launch("lb_groupnorm_from_stats(fold)");
    if (rc) return rc;
    const int vecs = C / 8;
    const int rows = vecs <= 256 ? 256 / vecs : 1;
    int want2 = (4096 + B - 1) / B;
    int apx = (HW + want2 - 1) / want2;
    const int min_px = 4 * rows;
    if (apx < min_px) apx = min_px;
    const int achunks = (HW + apx - 1) / apx;
    dim3 grid2((unsigned)achunks, B);
    if (x_is_f32)
        hipLaunchKernelGGL((gn_apply_kernel<float>), grid2, dim3(256), 0, stream, (const float*)x,
                           partial, gamma, beta, (f16*)y, HW, C, ldx, ldy, groups, 1, eps, silu, apx);
    else
        hipLaunchKernelGGL((gn_apply_kernel<f16>), grid2, dim3(256), 0, stream, (const f16*)x,
                           partial, gamma, beta, (f16*)y, HW, C, ldx, ldy, groups, 1, eps, silu, apx);
    return lb_check_launch("lb_groupnorm_from_stats(apply)");
}

extern "C" int lb_groupnorm_from_stats(const void* x, void* y, const float* gamma, const float* beta, const float* ch_stats,
                                       void* workspace, int B, int HW, int C, int ldx, int ldy, int groups, float eps,
                                       int silu, int x_is_f32, int stat_rows_per_sample, void* stream) {
    LB_REQUIRE(B > 0 && HW > 0 && C > 0 && stat_rows_per_sample > 0 && ch_stats != nullptr, "lb_groupnorm_from_stats: sizes");
    LB_REQUIRE(C % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0 && C <= 4096, "lb_groupnorm_from_stats: C/ld multiple of 8, C <= 4096");
    LB_REQUIRE(groups > 0 && groups <= GN_MAX_GROUPS && C % groups == 0, "lb_groupnorm_from_stats: groups");
    LB_DISPATCH("lb_groupnorm_from_stats", groupnorm_from_stats_impl(x, y, gamma, beta, ch_stats, workspace, B, HW, C, ldx, ldy,
                                                                     groups, eps, silu, x_is_f32, stat_rows_per_sample, s));
}

// ------------------------------------------------------------------------------------------
// LayerNorm over the last dimension: one wave per row, row kept in registers (C <= 2048).
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) layernorm_kernel(const f16* __restrict__ x,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta,
                                                        f16* __restrict__ y, int M, int C, int ldx,
                                                        int ldy, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= M) return;
    const int vecs = C >> 3;
    float val[4][8];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int v = lane + j * 64;
        if (v < vecs) {
            load8<f16>(x + (long)row * ldx + v * 8, val[j]);
#pragma unroll
            for (int e = 0; e < 8; ++e) s += val[j][e];
        }
    }
    const float mean = lb_wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int v = lane + j * 64;
        if (v < vecs) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = val[j][e] - mean; q += d * d; }
        }
    }
    const float rstd = rsqrtf(lb_wave_sum(q) / (float)C + eps);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int v = lane + j * 64;
        if (v < vecs) {
            const int c0 = v * 8;
            f16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e)
                o[e] = (f16)((val[j][e] - mean) * rstd * gamma[c0 + e] + beta[c0 + e]);
            *reinterpret_cast<f16x8*>(y + (long)row * ldy + c0) = o;
        }
    }
}

// Round 6: the form above waits for every 16-byte load of a row before it issues the next one (each sits in its own
// exec-masked branch: hipcc drains vmcnt(0) per branch), fetches gamma / beta only after both reductions, and reduces through 12
// ds_bpermute round trips - five to six dependent memory latencies per row, 11 us for the 22 MB of a B = 17 hidden state.  Here
// every load of the row AND its gamma / beta vectors are requested up front (vector slots past C read slot 0 and are weighted
// out), and the two reductions run on permlane / DPP exchanges.  Same per-lane summation order, same butterfly: the outputs are
// bit-identical to layernorm_kernel's (tests/test_kernels_gpu.py compares the two).
template <int NV>
__global__ void __launch_bounds__(256) layernorm_rows_kernel(const f16* __restrict__ x, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, f16* __restrict__ y, int M, int C,
                                                             int ldx, int ldy, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= M) return;                                   // (wave-uniform)
    const int vecs = C >> 3;
    f16x8 raw[NV];
    f32x4 gm[NV][2], bt[NV][2];
    bool ok[NV];
    int vc[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {          // the row itself first: requests retire in order, the reductions wait for these only
        const int v = lane + j * 64;
        ok[j] = v < vecs;
        vc[j] = ok[j] ? v : 0;
        raw[j] = *reinterpret_cast<const f16x8*>(x + (long)row * ldx + vc[j] * 8);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        gm[j][0] = *reinterpret_cast<const f32x4*>(gamma + vc[j] * 8);
        gm[j][1] = *reinterpret_cast<const f32x4*>(gamma + vc[j] * 8 + 4);
        bt[j][0] = *reinterpret_cast<const f32x4*>(beta + vc[j] * 8);
        bt[j][1] = *reinterpret_cast<const f32x4*>(beta + vc[j] * 8 + 4);
    }
    // (no branch anywhere below - a vector slot past C recomputes and re-stores vector 0 of the row with the same bits - so
    //  hipcc has no predicated block to sink the gamma / beta requests into; the sched_barrier keeps them above the arithmetic)
    __builtin_amdgcn_sched_barrier(0);
    float val[NV][8];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            val[j][e] = (float)raw[j][e];
            s += ok[j] ? val[j][e] : 0.f;
        }
    const float mean = lb_wave_sum_dpp(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float d = val[j][e] - mean;
            q = ok[j] ? __builtin_fmaf(d, d, q) : q;      // (the fused form layernorm_kernel's "q += d * d" contracts to: same bits)
        }
    const float rstd = rsqrtf(lb_wave_sum_dpp(q) / (float)C + eps);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        f16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e)
            o[e] = (f16)((val[j][e] - mean) * rstd * gm[j][e >> 2][e & 3] + bt[j][e >> 2][e & 3]);
        *reinterpret_cast<f16x8*>(y + (long)row * ldy + vc[j] * 8) = o;
    }
}

static int g_ln_form = 1;       // 1 = layernorm_rows_kernel (round 6), 0 = the round-1 kernel (A/B: tools/ln_bench.py; tests compare the two)
extern "C" void lb_layernorm_set_form(int form) { g_ln_form = form; }

extern "C" int lb_layernorm_f16(const void* x, void* y, const float* gamma, const float* beta, int M,
                                int C, int ldx, int ldy, float eps, void* stream) {
    LB_REQUIRE(M > 0 && C > 0 && C % 8 == 0 && C <= 2048, "lb_layernorm_f16: C multiple of 8, <= 2048");
    LB_REQUIRE(ldx % 8 == 0 && ldy % 8 == 0, "lb_layernorm_f16: ld multiple of 8");
    const int form = g_ln_form;
    const int nv = (C / 8 + 63) / 64;
    const dim3 grid((M + 3) / 4), block(256);
#define LB_LN_ARGS (const f16*)x, gamma, beta, (f16*)y, M, C, ldx, ldy, eps
    if (form == 0) LB_DISPATCH_STMT("lb_layernorm_f16", hipLaunchKernelGGL(layernorm_kernel, grid, block, 0, s, LB_LN_ARGS));
    if (nv == 1) LB_DISPATCH_STMT("lb_layernorm_f16", hipLaunchKernelGGL(layernorm_rows_kernel<1>, grid, block, 0, s, LB_LN_ARGS));
    if (nv == 2) LB_DISPATCH_STMT("lb_layernorm_f16", hipLaunchKernelGGL(layernorm_rows_kernel<2>, grid, block, 0, s, LB_LN_ARGS));
    if (nv == 3) LB_DISPATCH_STMT("lb_layernorm_f16", hipLaunchKernelGGL(layernorm_rows_kernel<3>, grid, block, 0, s, LB_LN_ARGS));
    LB_DISPATCH_STMT("lb_layernorm_f16", hipLaunchKernelGGL(layernorm_rows_kernel<4>, grid, block, 0, s, LB_LN_ARGS));
#undef LB_LN_ARGS
}
