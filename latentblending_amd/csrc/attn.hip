// Fused attention forward for head_dim 64 on gfx950 (UNet self- and cross-attention), plus a row
// softmax used by the VAE mid-block attention (head_dim 512, done as GEMM -> softmax -> GEMM).
//
//   O[b,q,h,:] = softmax_k( Q[b,q,h,:] . K[b,k,h,:] * scale ) . V[b,k,h,:]
//
// Block = 4 waves, 64 query rows (16 per wave); K and V^T tiles of 64 keys are staged in LDS
// (XOR-swizzled 128-B rows) and shared by the four waves; online softmax in registers.
// Both products run on v_mfma_f32_16x16x32_f16 with the operands arranged so that NO cross-lane
// data movement is needed between them:
//   S^T = K . Q^T   -> lane (q = lane&15, g = lane>>4) holds S for keys {16t + 4g + r}
//   O^T = V^T . P^T -> the MFMA k-slot (g, j) is mapped to key 32s + 16(j>>2) + 4g + (j&3), which is
//                      exactly what the lane already holds; V^T rows are read from LDS with the
//                      same permutation (two 8-byte reads per fragment).
// V is consumed transposed ([H*64][B*Skv], produced directly by a swapped-operand GEMM), K/Q in
// the fused-QKV token layout.  Keys >= Skv_valid are masked (cross-attention pads 77 -> 80).
//
// Replaces diffusers' AttnProcessor2_0 / F.scaled_dot_product_attention inside the UNet call at
// /root/reference/latentblending/diffusers_holder.py:336.
#include "lb_common.h"
#include "../../include/lb_hip.h"

#define ATT_D 64
#define ATT_KV 64

__global__ void __launch_bounds__(256) attn_fwd_d64_kernel(const LbAttnParams p) {
    __shared__ __attribute__((aligned(16))) f16 lds[2 * 2 * ATT_KV * ATT_D];   // [buf][K | Vt][64][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, l16 = lane & 15;
    const int b = blockIdx.z, h = blockIdx.y;
    const int q_row = blockIdx.x * 64 + wave * 16 + l16;          // this lane's query (as b-operand col)
    const f16* Q = reinterpret_cast<const f16*>(p.Q);
    const f16* K = reinterpret_cast<const f16*>(p.K);
    const f16* Vt = reinterpret_cast<const f16*>(p.Vt);
    f16* O = reinterpret_cast<f16*>(p.O);
    const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

    // Q fragments (b operand): k = d = 32s + 8g .. +8
    f16x8 qf[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        qf[s] = zero8;
        if (q_row < p.Sq)
            qf[s] = *reinterpret_cast<const f16x8*>(Q + ((long)b * p.Sq + q_row) * p.ldq + h * ATT_D + s * 32 + g * 8);
    }

    // staging: thread stages two 16-B chunks of K and two of V^T per tile
    const int slot = tid & 7, row0 = tid >> 3;                     // rows row0, row0 + 32
    f16x8 k_reg[2], v_reg[2];
    const int nt = (p.Skv + ATT_KV - 1) / ATT_KV;

    auto load_tile = [&](int t) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = row0 + i * 32;
            const int key = t * ATT_KV + r;
            k_reg[i] = zero8;
            if (key < p.Skv)
                k_reg[i] = *reinterpret_cast<const f16x8*>(K + ((long)b * p.Skv + key) * p.ldk + h * ATT_D + slot * 8);
            const int kcol = t * ATT_KV + slot * 8;                // V^T: row = d, 8 consecutive keys
            v_reg[i] = zero8;
            if (kcol < p.Skv)
                v_reg[i] = *reinterpret_cast<const f16x8*>(Vt + ((long)h * ATT_D + r) * p.ldvt + (long)b * p.Skv + kcol);
        }
    };
    auto store_tile = [&](int buf) {
        f16* Ks = lds + buf * 2 * ATT_KV * ATT_D;
        f16* Vs = Ks + ATT_KV * ATT_D;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = row0 + i * 32;
            *reinterpret_cast<f16x8*>(Ks + r * 64 + ((slot ^ (r & 7)) << 3)) = k_reg[i];
            *reinterpret_cast<f16x8*>(Vs + r * 64 + ((slot ^ (r & 7)) << 3)) = v_reg[i];
        }
    };

    f32x4 ot[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) ot[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;
    const float sc = p.scale * 1.44269504088896340736f;           // fold log2(e): exp2 domain

    load_tile(0);
    store_tile(0);
    __syncthreads();

    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        if (t + 1 < nt) load_tile(t + 1);
        const f16* Ks = lds + buf * 2 * ATT_KV * ATT_D;
        const f16* Vs = Ks + ATT_KV * ATT_D;

        // ---- S^T = K . Q^T ----
        f32x4 st[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            st[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const int r = kt * 16 + l16;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const f16x8 kf = *reinterpret_cast<const f16x8*>(Ks + r * 64 + (((s * 4 + g) ^ (r & 7)) << 3));
                st[kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[s], st[kt], 0, 0, 0);
            }
        }
        // ---- online softmax (this lane: one query, 16 of the tile's 64 keys) ----
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = t * ATT_KV + kt * 16 + 4 * g + r;
                const float v = key < p.Skv_valid ? st[kt][r] * sc : -INFINITY;
                st[kt][r] = v;
                mx = fmaxf(mx, v);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16, LB_WAVE));
        mx = fmaxf(mx, __shfl_xor(mx, 32, LB_WAVE));
        const float m_new = fmaxf(m_run, mx);
        const float m_use = m_new == -INFINITY ? 0.f : m_new;      // fully masked tile: keep zeros
        const float alpha = exp2f(m_run - m_use);                  // m_run = -inf -> 0
        float psum = 0.f;
        f16x8 pf[2];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = exp2f(st[kt][r] - m_use);
                psum += e;
                pf[kt >> 1][(kt & 1) * 4 + r] = (f16)e;
            }
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) ot[dt][r] *= alpha;

        // ---- O^T += V^T . P^T ----
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const int r = dt * 16 + l16;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                // keys 32s + 4g + {0..3} and 32s + 16 + 4g + {0..3}
                const int c_lo = s * 4 + (g >> 1), c_hi = c_lo + 2;
                const f16x4 lo = *reinterpret_cast<const f16x4*>(Vs + r * 64 + ((c_lo ^ (r & 7)) << 3) + (g & 1) * 4);
                const f16x4 hi = *reinterpret_cast<const f16x4*>(Vs + r * 64 + ((c_hi ^ (r & 7)) << 3) + (g & 1) * 4);
                const f16x8 vf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                ot[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf[s], ot[dt], 0, 0, 0);
            }
        }
        if (t + 1 < nt) store_tile(buf ^ 1);
        __syncthreads();
    }

    l_run += __shfl_xor(l_run, 16, LB_WAVE);
    l_run += __shfl_xor(l_run, 32, LB_WAVE);
    const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
    if (q_row < p.Sq) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const f16x4 o = {(f16)(ot[dt][0] * inv), (f16)(ot[dt][1] * inv), (f16)(ot[dt][2] * inv),
                             (f16)(ot[dt][3] * inv)};
            *reinterpret_cast<f16x4*>(O + ((long)b * p.Sq + q_row) * p.ldo + h * ATT_D + dt * 16 + 4 * g) = o;
        }
    }
}

extern "C" int lb_attn_fwd_d64(const LbAttnParams* pp, void* stream) {
    const LbAttnParams p = *pp;
    LB_REQUIRE(p.B > 0 && p.H > 0 && p.Sq > 0 && p.Skv > 0, "lb_attn_fwd_d64: sizes");
    LB_REQUIRE(p.Skv % 8 == 0 && p.Skv_valid > 0 && p.Skv_valid <= p.Skv, "lb_attn_fwd_d64: Skv % 8, valid");
    LB_REQUIRE(p.ldq % 8 == 0 && p.ldk % 8 == 0 && p.ldvt % 8 == 0 && p.ldo % 4 == 0, "lb_attn_fwd_d64: ld alignment");
    dim3 grid((p.Sq + 63) / 64, p.H, p.B);
    LB_DISPATCH_STMT("lb_attn_fwd_d64", hipLaunchKernelGGL(attn_fwd_d64_kernel, grid, dim3(256), 0, s, p));
}

// ------------------------------------------------------------------------------------------
// in-place row softmax: x[M][ld] fp16, softmax over the first N columns of (x * scale)
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) softmax_rows_kernel(f16* __restrict__ x, int N, int ld, float scale) {
    __shared__ float red[8];
    f16* row = x + (long)blockIdx.x * ld;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float sc = scale * 1.44269504088896340736f;
    float mx = -INFINITY;
    for (int i = tid * 8; i < N; i += 256 * 8) {
        const f16x8 v = *reinterpret_cast<const f16x8*>(row + i);
#pragma unroll
        for (int e = 0; e < 8; ++e) mx = fmaxf(mx, (float)v[e] * sc);
    }
    mx = lb_wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    for (int i = tid * 8; i < N; i += 256 * 8) {
        const f16x8 v = *reinterpret_cast<const f16x8*>(row + i);
#pragma unroll
        for (int e = 0; e < 8; ++e) sum += exp2f((float)v[e] * sc - mx);
    }
    sum = lb_wave_sum(sum);
    if (lane == 0) red[4 + wave] = sum;
    __syncthreads();
    const float inv = 1.f / (red[4] + red[5] + red[6] + red[7]);
    for (int i = tid * 8; i < N; i += 256 * 8) {
        const f16x8 v = *reinterpret_cast<const f16x8*>(row + i);
        f16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (f16)(exp2f((float)v[e] * sc - mx) * inv);
        *reinterpret_cast<f16x8*>(row + i) = o;
    }
}

extern "C" int lb_softmax_rows_f16(void* x, int M, int N, int ld, float scale, void* stream) {
    LB_REQUIRE(M > 0 && N > 0 && N % 8 == 0 && ld % 8 == 0, "lb_softmax_rows_f16: N, ld multiples of 8");
    LB_DISPATCH_STMT("lb_softmax_rows_f16", hipLaunchKernelGGL(softmax_rows_kernel, dim3(M), dim3(256), 0, s, (f16*)x, N, ld, scale));
}
