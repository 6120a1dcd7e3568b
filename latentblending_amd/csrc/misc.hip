// Small coalesced elementwise / gather kernels of the UNet, VAE and LPIPS paths on gfx950:
// sinusoidal embeddings, strided column copies (skip-connection concat), casts, VAE output
// quantisation, LPIPS input scaling, 3x3/2 max-pool and the LPIPS distance reduction.
//
// Replaces (third party, reached from the reference call sites in parentheses):
//   Timesteps / get_timestep_embedding           (diffusers_holder.py:336)
//   torch.cat([hidden, skip], dim=1)             (diffusers_holder.py:336)
//   VaeImageProcessor.postprocess                (diffusers_holder.py:141)
//   lpips ScalingLayer / max_pool2d / normalize_tensor / spatial_average (blending_engine.py:756)
#include "lb_common.h"

static unsigned grid_for(long items, int block = 256, long cap = 2048) {
    long g = (items + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (unsigned)g;
}

// out[r][col_off + j*dim + k] = (k < dim/2 ? cos : sin)(vals[r*per_row + j] * 10000^(-k'/half))
__global__ void sinusoid_kernel(const float* __restrict__ vals, int rows, int per_row, int dim,
                                f16* __restrict__ out, int ld_out, int col_off, int val_stride) {
    const int half = dim >> 1;
    const long total = (long)rows * per_row * dim;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int k = (int)(i % dim);
        const long rj = i / dim;
        const int j = (int)(rj % per_row), r = (int)(rj / per_row);
        const int kk = k < half ? k : k - half;
        const float freq = expf(-9.21034037197618f * (float)kk / (float)half);   // ln(10000)
        const float ang = vals[(long)r * val_stride + j] * freq;
        out[(long)r * ld_out + col_off + j * dim + k] = (f16)(k < half ? cosf(ang) : sinf(ang));
    }
}

extern "C" int lb_sinusoid_f16(const float* vals_dev, int rows, int per_row, int val_stride, int dim,
                               void* out, int ld_out, int col_off, void* stream) {
    LB_REQUIRE(rows > 0 && per_row > 0 && dim > 0 && dim % 2 == 0, "lb_sinusoid_f16: sizes");
    LB_DISPATCH_STMT("lb_sinusoid_f16", hipLaunchKernelGGL(sinusoid_kernel, dim3(grid_for((long)rows * per_row * dim)), dim3(256), 0,
                       s, vals_dev, rows, per_row, dim, (f16*)out, ld_out, col_off,
                       val_stride));
}

// dst[r][dst_off + c] = src[r][c], c < cols (multiple of 8), 16-B vectors
__global__ void copy_cols_kernel(const f16* __restrict__ src, f16* __restrict__ dst, long rows, int cols,
                                 int ld_src, int ld_dst, int dst_off) {
    const int vecs = cols >> 3;
    const long items = rows * vecs;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < items; i += (long)gridDim.x * blockDim.x) {
        const long r = i / vecs;
        const int v = (int)(i - r * vecs);
        *reinterpret_cast<f16x8*>(dst + r * ld_dst + dst_off + v * 8) =
            *reinterpret_cast<const f16x8*>(src + r * ld_src + v * 8);
    }
}

extern "C" int lb_copy_cols_f16(const void* src, void* dst, long rows, int cols, int ld_src, int ld_dst,
                                int dst_off, void* stream) {
    LB_REQUIRE(rows > 0 && cols > 0 && cols % 8 == 0 && ld_src % 8 == 0 && ld_dst % 8 == 0 && dst_off % 8 == 0,
               "lb_copy_cols_f16: cols / ld / offset multiples of 8");
    LB_DISPATCH_STMT("lb_copy_cols_f16", hipLaunchKernelGGL(copy_cols_kernel, dim3(grid_for(rows * (cols / 8))), dim3(256), 0, s,
                       (const f16*)src, (f16*)dst, rows, cols, ld_src, ld_dst, dst_off));
}

__global__ void cast_f16_f32_kernel(const f16* __restrict__ x, float* __restrict__ y, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) y[i] = (float)x[i];
}
__global__ void cast_f32_f16_kernel(const float* __restrict__ x, f16* __restrict__ y, long n, float mul) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        y[i] = (f16)fminf(fmaxf(x[i] * mul, -65504.f), 65504.f);
}
extern "C" int lb_cast_f16_to_f32(const void* x, void* y, long n, void* stream) {
    LB_DISPATCH_STMT("lb_cast_f16_to_f32", hipLaunchKernelGGL(cast_f16_f32_kernel, dim3(grid_for(n)), dim3(256), 0, s, (const f16*)x, (float*)y, n));
}
// y = fp16(saturate(x * mul)); mul = 0 is read as 1 (a power-of-two down-scale keeps the VAE's
// fp32 residual stream inside fp16 range for the next conv, whose epilogue multiplies back)
extern "C" int lb_cast_f32_to_f16(const void* x, void* y, long n, float mul, void* stream) {
    const float m = mul == 0.f ? 1.f : mul;
    LB_DISPATCH_STMT("lb_cast_f32_to_f16", hipLaunchKernelGGL(cast_f32_f16_kernel, dim3(grid_for(n)), dim3(256), 0, s, (const float*)x, (f16*)y, n, m));
}

// NCHW fp16 latent [B][C][HW] <-> NHWC fp16 [B][HW][ld] (C small: 4), with an optional scalar
// multiply on the way in (latents / scaling_factor before the VAE, diffusers_holder.py:135).
__global__ void nchw_to_nhwc_kernel(const f16* __restrict__ x, f16* __restrict__ y, int B, int C, int HW,
                                    int ld, float mul) {
    const long total = (long)B * HW * ld;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % ld);
        const long bp = i / ld;
        const int px = (int)(bp % HW), b = (int)(bp / HW);
        y[i] = c < C ? (f16)((float)x[((long)b * C + c) * HW + px] * mul) : (f16)0.f;
    }
}
__global__ void nhwc_to_nchw_kernel(const f16* __restrict__ x, f16* __restrict__ y, int B, int C, int HW, int ld) {
    const long total = (long)B * C * HW;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int px = (int)(i % HW);
        const long bc = i / HW;
        const int c = (int)(bc % C), b = (int)(bc / C);
        y[i] = x[((long)b * HW + px) * ld + c];
    }
}
extern "C" int lb_nchw_to_nhwc_f16(const void* x, void* y, int B, int C, int HW, int ld, float mul, void* stream) {
    LB_REQUIRE(B > 0 && C > 0 && HW > 0 && ld >= C, "lb_nchw_to_nhwc_f16: sizes");
    LB_DISPATCH_STMT("lb_nchw_to_nhwc_f16", hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(grid_for((long)B * HW * ld)), dim3(256), 0, s,
                       (const f16*)x, (f16*)y, B, C, HW, ld, mul == 0.f ? 1.f : mul));
}
extern "C" int lb_nhwc_to_nchw_f16(const void* x, void* y, int B, int C, int HW, int ld, void* stream) {
    LB_REQUIRE(B > 0 && C > 0 && HW > 0 && ld >= C, "lb_nhwc_to_nchw_f16: sizes");
    LB_DISPATCH_STMT("lb_nhwc_to_nchw_f16", hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(grid_for((long)B * C * HW)), dim3(256), 0, s,
                       (const f16*)x, (f16*)y, B, C, HW, ld));
}

// VAE output [B][HW][ld] (fp32 or fp16, channels 0..2) -> uint8 [B][HW][3]:
// round_half_even(clamp(x/2 + 0.5, 0, 1) * 255)
template <typename T>
__global__ void postprocess_u8_kernel(const T* __restrict__ x, uint8_t* __restrict__ out, long pixels, int ld) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < pixels * 3; i += (long)gridDim.x * blockDim.x) {
        const long px = i / 3;
        const int c = (int)(i - px * 3);
        float v = (float)x[px * ld + c] / 2.f + 0.5f;
        v = fminf(fmaxf(v, 0.f), 1.f);
        out[i] = (uint8_t)rintf(v * 255.f);
    }
}
static int postprocess_impl(const void* x, void* out_u8, long pixels, int ld, int x_is_f32, hipStream_t s) {
    if (x_is_f32)
        hipLaunchKernelGGL((postprocess_u8_kernel<float>), dim3(grid_for(pixels * 3)), dim3(256), 0, s,
                           (const float*)x, (uint8_t*)out_u8, pixels, ld);
    else
        hipLaunchKernelGGL((postprocess_u8_kernel<f16>), dim3(grid_for(pixels * 3)), dim3(256), 0, s,
                           (const f16*)x, (uint8_t*)out_u8, pixels, ld);
    return lb_check_launch("lb_postprocess_u8");
}
extern "C" int lb_postprocess_u8(const void* x, void* out_u8, long pixels, int ld, int x_is_f32, void* stream) {
    LB_REQUIRE(pixels > 0 && ld >= 3, "lb_postprocess_u8: sizes");
    LB_DISPATCH("lb_postprocess_u8", postprocess_impl(x, out_u8, pixels, ld, x_is_f32, s));
}

// ---- LPIPS helpers ------------------------------------------------------------------------
// uint8 frame [N][HW][3] -> fp16 NHWC [N][HW][8]: ((2*x/255 - 1) - shift_c) / scale_c, channels 3..7 zero
__global__ void lpips_prep_kernel(const uint8_t* __restrict__ img, f16* __restrict__ out, long pixels) {
    const float shift[3] = {-.030f, -.088f, -.188f}, scale[3] = {.458f, .448f, .450f};
    for (long px = (long)blockIdx.x * blockDim.x + threadIdx.x; px < pixels; px += (long)gridDim.x * blockDim.x) {
        f16x8 o = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v = 2.f * (float)img[px * 3 + c] / 255.f - 1.f;
            o[c] = (f16)((v - shift[c]) / scale[c]);
        }
        *reinterpret_cast<f16x8*>(out + px * 8) = o;
    }
}
extern "C" int lb_lpips_prep_u8(const void* img_u8, void* out_f16, long pixels, void* stream) {
    LB_DISPATCH_STMT("lb_lpips_prep_u8", hipLaunchKernelGGL(lpips_prep_kernel, dim3(grid_for(pixels)), dim3(256), 0, s,
                       (const uint8_t*)img_u8, (f16*)out_f16, pixels));
}

// NHWC max-pool k3 s2 (no padding): [N][H][W][C] -> [N][Ho][Wo][C], Ho = (H-3)/2+1
__global__ void maxpool3s2_kernel(const f16* __restrict__ x, f16* __restrict__ y, int N, int H, int W, int C,
                                  int Ho, int Wo) {
    const int vecs = C >> 3;
    const long items = (long)N * Ho * Wo * vecs;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < items; i += (long)gridDim.x * blockDim.x) {
        const int v = (int)(i % vecs);
        long r = i / vecs;
        const int ox = (int)(r % Wo); r /= Wo;
        const int oy = (int)(r % Ho);
        const int n = (int)(r / Ho);
        f16x8 m;
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = (f16)-65504.f;
        for (int dy = 0; dy < 3; ++dy)
            for (int dx = 0; dx < 3; ++dx) {
                const f16x8 t = *reinterpret_cast<const f16x8*>(x + (((long)n * H + oy * 2 + dy) * W + ox * 2 + dx) * C + v * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) m[e] = t[e] > m[e] ? t[e] : m[e];
            }
        *reinterpret_cast<f16x8*>(y + (((long)n * Ho + oy) * Wo + ox) * C + v * 8) = m;
    }
}
extern "C" int lb_maxpool3s2_nhwc_f16(const void* x, void* y, int N, int H, int W, int C, void* stream) {
    LB_REQUIRE(N > 0 && H >= 3 && W >= 3 && C % 8 == 0, "lb_maxpool3s2_nhwc_f16: sizes");
    const int Ho = (H - 3) / 2 + 1, Wo = (W - 3) / 2 + 1;
    LB_DISPATCH_STMT("lb_maxpool3s2_nhwc_f16", hipLaunchKernelGGL(maxpool3s2_kernel, dim3(grid_for((long)N * Ho * Wo * (C / 8))), dim3(256), 0, s,
                       (const f16*)x, (f16*)y, N, H, W, C, Ho, Wo));
}

// One LPIPS tap for up to 16 frame pairs: features fa/fb [HW][C] per pair (pointers by value);
//   d = mean_px sum_c lin[c] * (fa/(|fa|+eps) - fb/(|fb|+eps))^2 ;  acc[pair] += d
// One wave per pixel; per-block partial sums are folded by one thread in fixed order
// (deterministic, no float atomics).  acc must be zeroed before the first tap.
#define LPIPS_MAX_PAIRS 16
struct LpipsPairs { const f16* a[LPIPS_MAX_PAIRS]; const f16* b[LPIPS_MAX_PAIRS]; };

__global__ void __launch_bounds__(256) lpips_tap_kernel(LpipsPairs pp, const float* __restrict__ lin,
                                                        float* __restrict__ block_part, int HW, int C) {
    __shared__ float red[4];
    const int pair = blockIdx.y;
    const f16* fa = pp.a[pair];
    const f16* fb = pp.b[pair];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc = 0.f;
    for (int px = blockIdx.x * 4 + wave; px < HW; px += gridDim.x * 4) {
        float sa = 0.f, sb = 0.f;
        for (int c = lane; c < C; c += 64) {
            const float a = (float)fa[(long)px * C + c], b = (float)fb[(long)px * C + c];
            sa += a * a; sb += b * b;
        }
        sa = lb_wave_sum(sa); sb = lb_wave_sum(sb);
        const float ia = 1.f / (sqrtf(sa) + 1e-10f), ib = 1.f / (sqrtf(sb) + 1e-10f);
        float d = 0.f;
        for (int c = lane; c < C; c += 64) {
            const float t = (float)fa[(long)px * C + c] * ia - (float)fb[(long)px * C + c] * ib;
            d += lin[c] * t * t;
        }
        acc += lb_wave_sum(d);
    }
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) block_part[(long)pair * gridDim.x + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ void lpips_fold_kernel(const float* __restrict__ block_part, float* __restrict__ acc, int nblk, float inv_hw) {
    const int pair = blockIdx.x;
    if (threadIdx.x == 0) {
        double s = 0;
        for (int i = 0; i < nblk; ++i) s += (double)block_part[(long)pair * nblk + i];
        acc[pair] += (float)(s * inv_hw);
    }
}
static int lpips_tap_impl(LpipsPairs pp, const float* lin, float* acc, float* workspace, int npairs, int HW, int C,
                          hipStream_t s) {
    int nblk = (HW + 3) / 4;
    if (nblk > 128) nblk = 128;
    hipLaunchKernelGGL(lpips_tap_kernel, dim3(nblk, npairs), dim3(256), 0, s, pp, lin, workspace, HW, C);
    int rc = lb_check_launch("lb_lpips_tap");
    if (rc) return rc;
    hipLaunchKernelGGL(lpips_fold_kernel, dim3(npairs), dim3(64), 0, s, workspace, acc, nblk, 1.f / (float)HW);
    return lb_check_launch("lb_lpips_tap(fold)");
}
// feats_a / feats_b: HOST arrays of npairs device pointers (<= 16); workspace: 16*128 floats
extern "C" int lb_lpips_tap(const void* const* feats_a, const void* const* feats_b, const float* lin, float* acc,
                            float* workspace, int npairs, int HW, int C, void* stream) {
    LB_REQUIRE(npairs > 0 && npairs <= LPIPS_MAX_PAIRS && HW > 0 && C > 0, "lb_lpips_tap: sizes (<= 16 pairs)");
    LpipsPairs pp;
    for (int i = 0; i < npairs; ++i) { pp.a[i] = (const f16*)feats_a[i]; pp.b[i] = (const f16*)feats_b[i]; }
    LB_DISPATCH("lb_lpips_tap", lpips_tap_impl(pp, lin, acc, workspace, npairs, HW, C, s));
}

// generic fill (zero LPIPS accumulators etc. without a runtime memset node)
__global__ void fill_f32_kernel(float* x, long n, float v) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) x[i] = v;
}
extern "C" int lb_fill_f32(void* x, long n, float v, void* stream) {
    LB_DISPATCH_STMT("lb_fill_f32", hipLaunchKernelGGL(fill_f32_kernel, dim3(grid_for(n)), dim3(256), 0, s, (float*)x, n, v));
}

// ------------------------------------------------------------------------------------------
// CLIP text towers (encode_prompt, /root/reference/latentblending/diffusers_holder.py:79-96):
// out[r][:] = token_embedding[ids[r]][:] + position_embedding[r % seq][:]   (CLIPTextEmbeddings)
// ------------------------------------------------------------------------------------------
__global__ void embed_tokens_kernel(const int* __restrict__ ids, const f16* __restrict__ tok,
                                    const f16* __restrict__ pos, f16* __restrict__ out, int rows, int seq, int C, int vocab) {
    const int vecs = C >> 3;
    const long total = (long)rows * vecs;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int r = (int)(i / vecs), v = (int)(i - (long)r * vecs);
        int id = ids[r];
        id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
        const f16x8 a = *reinterpret_cast<const f16x8*>(tok + (long)id * C + v * 8);
        const f16x8 b = *reinterpret_cast<const f16x8*>(pos + (long)(r % seq) * C + v * 8);
        f16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (f16)((float)a[e] + (float)b[e]);
        *reinterpret_cast<f16x8*>(out + (long)r * C + v * 8) = o;
    }
}

extern "C" int lb_embed_tokens_f16(const int* ids_dev, const void* tok_emb, const void* pos_emb, void* out, int rows,
                                   int seq, int C, int vocab, void* stream) {
    LB_REQUIRE(rows > 0 && seq > 0 && C > 0 && C % 8 == 0 && vocab > 0, "lb_embed_tokens_f16: sizes (C multiple of 8)");
    LB_DISPATCH_STMT("lb_embed_tokens_f16", hipLaunchKernelGGL(embed_tokens_kernel, dim3(grid_for((long)rows * (C / 8))), dim3(256), 0, s,
                       ids_dev, (const f16*)tok_emb, (const f16*)pos_emb, (f16*)out, rows, seq, C, vocab));
}

// out[i][:] = src[rows_idx[i]][:]  (the EOS-token row of every prompt: pooled CLIP output)
__global__ void gather_rows_kernel(const f16* __restrict__ src, const int* __restrict__ idx, f16* __restrict__ out, int n, int C,
                                   int ld_src) {
    const int vecs = C >> 3;
    const long total = (long)n * vecs;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int r = (int)(i / vecs), v = (int)(i - (long)r * vecs);
        *reinterpret_cast<f16x8*>(out + (long)r * C + v * 8) = *reinterpret_cast<const f16x8*>(src + (long)idx[r] * ld_src + v * 8);
    }
}

extern "C" int lb_gather_rows_f16(const void* src, const int* rows_idx_dev, void* out, int n, int C, int ld_src, void* stream) {
    LB_REQUIRE(n > 0 && C > 0 && C % 8 == 0 && ld_src % 8 == 0, "lb_gather_rows_f16: sizes (C, ld multiples of 8)");
    LB_DISPATCH_STMT("lb_gather_rows_f16", hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for((long)n * (C / 8))), dim3(256), 0, s,
                       (const f16*)src, rows_idx_dev, (f16*)out, n, C, ld_src));
}


// ------------------------------------------------------------------------------------------
// Frame in-betweening for the transition movie (reference utils.py:166-176 -> interpolate_linear :97 on float32 copies
// of the uint8 key frames with a float64 weight): out[k] = uint8( (1 - w[k]) * frame[left[k]] + w[k] * frame[left[k]+1] ),
// the products and the sum rounded to float64 as numpy evaluates them (NumPy >= 2 promotes the float32 array with the
// float64 scalar to float64), the cast truncating.  16 bytes per thread.
// ------------------------------------------------------------------------------------------
typedef unsigned char u8x16 __attribute__((ext_vector_type(16)));

__global__ void __launch_bounds__(256) frames_lerp_u8_kernel(const unsigned char* __restrict__ frames, const int* __restrict__ left,
                                                             const double* __restrict__ w, unsigned char* __restrict__ out,
                                                             long frame_bytes) {
    const long k = blockIdx.y;
    const unsigned char* a = frames + (long)left[k] * frame_bytes;
    const unsigned char* b = a + frame_bytes;
    unsigned char* o = out + k * frame_bytes;
    const double wk = w[k], w0 = 1.0 - wk;
    const long nvec = frame_bytes >> 4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
        const u8x16 va = *reinterpret_cast<const u8x16*>(a + i * 16), vb = *reinterpret_cast<const u8x16*>(b + i * 16);
        u8x16 r;
#pragma unroll
        for (int e = 0; e < 16; ++e)
            r[e] = (unsigned char)(int)__dadd_rn(__dmul_rn(w0, (double)va[e]), __dmul_rn(wk, (double)vb[e]));
        *reinterpret_cast<u8x16*>(o + i * 16) = r;
    }
    for (long i = (nvec << 4) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < frame_bytes; i += (long)gridDim.x * blockDim.x)
        o[i] = (unsigned char)(int)__dadd_rn(__dmul_rn(w0, (double)a[i]), __dmul_rn(wk, (double)b[i]));
}

extern "C" int lb_frames_lerp_u8(const void* frames, const int* left_dev, const double* w_dev, void* out, long n_out,
                                 long frame_bytes, void* stream) {
    LB_REQUIRE(n_out > 0 && n_out <= 65535 && frame_bytes > 0, "lb_frames_lerp_u8: 1..65535 output frames");
    LB_REQUIRE(((uintptr_t)frames & 15) == 0 && ((uintptr_t)out & 15) == 0 && frame_bytes % 16 == 0,
               "lb_frames_lerp_u8: 16-byte aligned buffers, frame size a multiple of 16 bytes");
    long bx = (frame_bytes / 16 + 255) / 256;
    if (bx > 512) bx = 512;
    LB_DISPATCH_STMT("lb_frames_lerp_u8", hipLaunchKernelGGL(frames_lerp_u8_kernel, dim3((unsigned)bx, (unsigned)n_out), dim3(256), 0, s,
                       (const unsigned char*)frames, left_dev, w_dev, (unsigned char*)out, frame_bytes));
}
