// Fused attention forward for head_dim 512 on gfx950: the VAE decoder's mid-block attention (one head of 512 channels over
// (H/8)^2 tokens: S = 4,096 at 512^2, 16,384 at 1024^2).
//
//   O[b,q,h,:] = softmax_k( Q[b,q,h,:] . K[b,k,h,:] * scale ) . V[b,k,h,:]
//
// Why: the three-launch form (scores GEMM -> lb_softmax_rows_f16 -> PV GEMM, native/vae.py) writes an S x S fp16 score
// matrix per sample - 32 MB at 512^2, 512 MB at 1024^2 - reads it twice and rounds the scores to fp16 before the
// softmax.  Here the scores stay in fp32 registers and nothing of size S x S exists.
//
// Same conventions as attn.hip (d = 64), re-dimensioned for 1 KB rows:
//   * block = 4 waves, every wave owns 16 query rows (64 per block); its Q rows live in registers as 16 MFMA b-fragments,
//     its O^T accumulators are 32 blocks of 16 d x 16 queries (128 VGPRs) plus the all-ones block that accumulates the
//     softmax denominators on the matrix pipe;
//   * K and V tiles of KT = 32 keys (32 KiB each) travel global -> LDS with `global_load_lds_dwordx4` into a 2-stage ring
//     (128 KiB): tile t+1 is in flight while tile t is consumed; one wave instruction fills one key row (64 chunks of 16 B);
//   * S^T = K . Q^T (lane (q = lane & 15, g = lane >> 4) holds keys 16 kb + 4 g + r), online softmax in the exp2 domain with
//     the deferred rescale of attn.hip, O^T += V^T . P^T with the V^T operand fetched from the row-major V tile through
//     `ds_read_b64_tr_b16`, so P never moves between lanes;
//   * the 16-B chunks of a row are XOR-swizzled inside each group of 16 chunks (256 B = all 64 banks), applied to the global
//     SOURCE address of the LDS-DMA.  The two operands are read by different patterns and use different keys:
//       K (ds_read_b128: 16 lanes = 16 keys, one logical chunk per g):            chunk ^= key & 15
//       V (transpose read: 32 lanes = 8 keys x (2 chunks x 2 halves)):            chunk ^= 2 (key & 7) + ((key >> 3) & 1)
//     (with the K key the V pattern would be 2-way conflicted: keys k and k ^ 1 swap the chunk pair; with the V key the
//     b128 lane groups - {0-3, 12-15, 20-27}, ... - would be).
// Per 32-key tile and wave: 32 + 32 MFMAs (+1 for the row sums) against 32 ds_read_b128 + 64 ds_read_b64_tr_b16, i.e. the
// LDS array is as busy as the matrix pipe (every wave reads the whole K and V tile: there is no reuse across waves at 16
// query rows per wave, and 32 rows would need 256 accumulator registers): LDS-bound by design, which still beats the
// three-launch form because the S x S traffic is gone.
//
// Replaces: diffusers' Attention / AttnProcessor2_0 inside AutoencoderKL.decode's mid block, reached from
// /root/reference/latentblending/diffusers_holder.py:135.
#include "lb_common.h"
#include "../../include/lb_hip.h"

#define A5_D 512
#define A5_KT 32
#define A5_NS 2
#define A5_DEFER 8.0f

typedef __attribute__((address_space(1))) const void* a5_gptr_t;
typedef __attribute__((address_space(3))) void* a5_lptr_t;

template <int N> struct A5Int { static constexpr int value = N; };

template <int OFF_BYTES> __device__ __forceinline__ f16x4 a5_tr_read(unsigned lds_byte_addr) {
    f16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(lds_byte_addr), "n"(OFF_BYTES) : "memory");
    return v;
}
template <int N> __device__ __forceinline__ void a5_tr_wait() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}
// the two transpose reads of d-block DT: keys 4 g + (m >> 2) and the same + 16; (DT >> 3) selects the 16-chunk group (+ 256 B),
// (DT & 7) the precomputed swizzled offset
template <int DT> __device__ __forceinline__ void a5_tr_pair(unsigned vb, const int (&voff)[8], f16x4& lo, f16x4& hi) {
    lo = a5_tr_read<(DT >> 3) * 256>(vb + voff[DT & 7] * 2);
    hi = a5_tr_read<(DT >> 3) * 256 + 16 * A5_D * 2>(vb + voff[DT & 7] * 2);
}
__device__ __forceinline__ unsigned a5_lds_addr(const void* p) { return (unsigned)(unsigned long)(a5_lptr_t)p; }
__device__ __forceinline__ int a5_swz_k(int key) { return key & 15; }
__device__ __forceinline__ int a5_swz_v(int key) { return 2 * (key & 7) + ((key >> 3) & 1); }

__global__ void __launch_bounds__(256) attn_fwd_d512_kernel(const LbAttnParams p) {
    constexpr int KT = A5_KT, D = A5_D;
    constexpr int TILE = KT * D;                // halves per operand tile
    constexpr int STAGE = 2 * TILE;             // K tile, then V tile
    constexpr int ROWS_PER_WAVE = KT / 4;       // 8 wave instructions per operand and tile
    extern __shared__ __attribute__((aligned(16))) f16 a5_lds[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, l16 = lane & 15;
    const int b = blockIdx.z, h = blockIdx.y;
    const int q0 = blockIdx.x * 64 + wave * 16;
    const f16* Q = reinterpret_cast<const f16*>(p.Q);
    const f16* K = reinterpret_cast<const f16*>(p.K);
    const f16* V = reinterpret_cast<const f16*>(p.V);
    f16* O = reinterpret_cast<f16*>(p.O);
    const f16* zero = reinterpret_cast<const f16*>(p.zero_page);
    const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

    // ---- loader: wave instruction i of wave w fills tile row r = 8 w + i; lane l is PHYSICAL chunk l of that row and fetches
    //      logical chunk (l & ~15) | ((l & 15) ^ swz(r)) ----
    const f16* kbase = K + (long)b * p.Skv * p.ldk + h * D;
    const f16* vbase = V + (long)b * p.Skv * p.ldv + h * D;
    auto issue_tile = [&](int t, int st) {
        f16* base = a5_lds + st * STAGE;
#pragma unroll
        for (int i = 0; i < ROWS_PER_WAVE; ++i) {
            const int r = wave * ROWS_PER_WAVE + i;
            const int key = t * KT + r;
            const int lc = (lane & ~15) | ((lane & 15) ^ a5_swz_k(r));
            const f16* src = key < p.Skv ? kbase + (long)key * p.ldk + lc * 8 : zero;
            __builtin_amdgcn_global_load_lds((a5_gptr_t)src, (a5_lptr_t)(base + r * D), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < ROWS_PER_WAVE; ++i) {
            const int r = wave * ROWS_PER_WAVE + i;
            const int key = t * KT + r;
            const int lc = (lane & ~15) | ((lane & 15) ^ a5_swz_v(r));
            const f16* src = key < p.Skv ? vbase + (long)key * p.ldv + lc * 8 : zero;
            __builtin_amdgcn_global_load_lds((a5_gptr_t)src, (a5_lptr_t)(base + TILE + r * D), 16, 0, 0);
        }
    };

    const int nt = (p.Skv + KT - 1) / KT;
    // ---- prologue: Q fragments first (b operand: query l16, k = d = 32 s + 8 g .. + 8), then tile 0 ----
    f16x8 qf[16];
    {
        const int q_row = q0 + l16;
        const f16* qp = Q + ((long)b * p.Sq + (q_row < p.Sq ? q_row : 0)) * p.ldq + h * D + g * 8;
#pragma unroll
        for (int s = 0; s < 16; ++s) qf[s] = q_row < p.Sq ? *reinterpret_cast<const f16x8*>(qp + s * 32) : zero8;
    }
    __builtin_amdgcn_sched_barrier(0);
    issue_tile(0, 0);

    // ---- loop-invariant LDS offsets (halves, relative to the stage base) ----
    // K fragment (a operand) of key block kb, d-step s: row 16 kb + l16, logical chunk 4 s + g = 16 (s >> 2) + (4 (s & 3) + g)
    int koff[4];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) koff[s4] = l16 * D + (((4 * s4 + g) ^ a5_swz_k(l16)) << 3);
    // V^T fragment of d-block dt through the transpose read: lane m = l16 supplies the address of key 4 g + (m >> 2) (+ 16 for the
    // upper half of the k-slots), d = 16 dt + 4 (m & 3): logical chunk 2 dt + ((m & 3) >> 1) = 16 (dt >> 3) + (2 (dt & 7) + ...),
    // 8-byte half (m & 1); keys k and k + 16 share the swizzle key
    const int vrow = 4 * g + (l16 >> 2);
    int voff[8];
#pragma unroll
    for (int d8 = 0; d8 < 8; ++d8)
        voff[d8] = TILE + vrow * D + (((2 * d8 + ((l16 & 3) >> 1)) ^ a5_swz_v(vrow)) << 3) + (l16 & 1) * 4;

    f32x4 ot[32], lt = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dt = 0; dt < 32; ++dt) ot[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY;
    const f16x8 ones8 = {(f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f};
    const float sc = p.scale * 1.44269504088896340736f;

    for (int t = 0; t < nt; ++t) {
        const int st = t & 1;
        // tile t has landed in every wave's share; every wave is done with tile t-1 (stage st ^ 1): refill it
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        issue_tile(t + 1, st ^ 1);                 // (masked to the zero page past the end)
        const f16* Ks = a5_lds + st * STAGE;

        // ---- S^T = K . Q^T: two key blocks, 16 d-steps each (the two accumulation chains are interleaved) ----
        // (fragments are requested one group of four - two d-steps x two key blocks - ahead of the MFMAs that consume them)
        f32x4 sacc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        {
            f16x8 kf[2][4];
            auto k_request = [&](int grp, f16x8 (&dst)[4]) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int s = 2 * grp + (j >> 1), kb = j & 1;
                    dst[j] = *reinterpret_cast<const f16x8*>(Ks + kb * 16 * D + (s >> 2) * 128 + koff[s & 3]);
                }
            };
            k_request(0, kf[0]);
#pragma unroll
            for (int grp = 0; grp < 8; ++grp) {
                if (grp + 1 < 8) k_request(grp + 1, kf[(grp + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    sacc[j & 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[grp & 1][j], qf[2 * grp + (j >> 1)], sacc[j & 1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- online softmax (this lane: query l16, keys 16 kb + 4 g + r of the tile) ----
        if ((t + 1) * KT > p.Skv_valid) {          // wave-uniform: only the last tile masks
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (t * KT + kb * 16 + 4 * g + r >= p.Skv_valid) sacc[kb][r] = -INFINITY;
        }
        float mx = fmaxf(fmaxf(fmaxf(sacc[0][0], sacc[0][1]), fmaxf(sacc[0][2], sacc[0][3])),
                         fmaxf(fmaxf(sacc[1][0], sacc[1][1]), fmaxf(sacc[1][2], sacc[1][3])));
        mx = fmaxf(mx, __shfl_xor(mx, 16, LB_WAVE));
        mx = fmaxf(mx, __shfl_xor(mx, 32, LB_WAVE));
        const float m_cand = mx * sc;
        if (__any(m_cand > m_run + A5_DEFER)) {
            const float m_new = fmaxf(m_run, m_cand);
            const float m_fin = m_new == -INFINITY ? 0.f : m_new;
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_fin);
            m_run = m_new;
#pragma unroll
            for (int dt = 0; dt < 32; ++dt)
#pragma unroll
                for (int r = 0; r < 4; ++r) ot[dt][r] *= alpha;
#pragma unroll
            for (int r = 0; r < 4; ++r) lt[r] *= alpha;
        }
        const float m_use = m_run == -INFINITY ? 0.f : m_run;
        f16x8 pf;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                pf[kb * 4 + r] = (f16)__builtin_amdgcn_exp2f(__builtin_fmaf(sacc[kb][r], sc, -m_use));

        // ---- O^T += V^T . P^T: 32 d-blocks in 8 groups of 4; group n+1 is requested before group n is multiplied ----
        {
            const unsigned vb = a5_lds_addr(Ks);
            f16x4 vlo[2][4], vhi[2][4];
            auto request = [&](auto grp_c, int slot) {
                constexpr int grp = decltype(grp_c)::value;          // d-blocks 4 grp .. 4 grp + 3
                a5_tr_pair<4 * grp + 0>(vb, voff, vlo[slot][0], vhi[slot][0]);
                a5_tr_pair<4 * grp + 1>(vb, voff, vlo[slot][1], vhi[slot][1]);
                a5_tr_pair<4 * grp + 2>(vb, voff, vlo[slot][2], vhi[slot][2]);
                a5_tr_pair<4 * grp + 3>(vb, voff, vlo[slot][3], vhi[slot][3]);
            };
            auto multiply = [&](auto grp_c, int slot) {
                constexpr int grp = decltype(grp_c)::value;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f16x8 vf = {vlo[slot][j][0], vlo[slot][j][1], vlo[slot][j][2], vlo[slot][j][3],
                                      vhi[slot][j][0], vhi[slot][j][1], vhi[slot][j][2], vhi[slot][j][3]};
                    ot[4 * grp + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf, ot[4 * grp + j], 0, 0, 0);
                }
            };
            request(A5Int<0>{}, 0);
            request(A5Int<1>{}, 1); a5_tr_wait<8>(); multiply(A5Int<0>{}, 0);
            request(A5Int<2>{}, 0); a5_tr_wait<8>(); multiply(A5Int<1>{}, 1);
            request(A5Int<3>{}, 1); a5_tr_wait<8>(); multiply(A5Int<2>{}, 0);
            request(A5Int<4>{}, 0); a5_tr_wait<8>(); multiply(A5Int<3>{}, 1);
            request(A5Int<5>{}, 1); a5_tr_wait<8>(); multiply(A5Int<4>{}, 0);
            request(A5Int<6>{}, 0); a5_tr_wait<8>(); multiply(A5Int<5>{}, 1);
            request(A5Int<7>{}, 1); a5_tr_wait<8>(); multiply(A5Int<6>{}, 0);
            a5_tr_wait<0>(); multiply(A5Int<7>{}, 1);
            lt = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones8, pf, lt, 0, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the masked tail request still targets this block's LDS

    const float l = lt[0];
    const float inv = l > 0.f ? 1.f / l : 0.f;
    const int q_row = q0 + l16;
    if (q_row < p.Sq) {
        f16* op = O + ((long)b * p.Sq + q_row) * p.ldo + h * D + 4 * g;
#pragma unroll
        for (int dt = 0; dt < 32; ++dt) {
            const f16x4 o = {(f16)(ot[dt][0] * inv), (f16)(ot[dt][1] * inv), (f16)(ot[dt][2] * inv), (f16)(ot[dt][3] * inv)};
            *reinterpret_cast<f16x4*>(op + dt * 16) = o;
        }
    }
}

extern "C" int lb_attn_fwd_d512(const LbAttnParams* pp, void* stream) {
    const LbAttnParams p = *pp;
    LB_REQUIRE(p.B > 0 && p.H > 0 && p.Sq > 0 && p.Skv > 0, "lb_attn_fwd_d512: sizes");
    LB_REQUIRE(p.Skv_valid > 0 && p.Skv_valid <= p.Skv, "lb_attn_fwd_d512: 0 < Skv_valid <= Skv");
    LB_REQUIRE(p.ldq % 8 == 0 && p.ldk % 8 == 0 && p.ldv % 8 == 0 && p.ldo % 4 == 0, "lb_attn_fwd_d512: ld alignment");
    LB_REQUIRE(p.zero_page != nullptr, "lb_attn_fwd_d512: zero_page (>= 16 zero bytes) is required");
    LB_REQUIRE(!p.causal, "lb_attn_fwd_d512: no causal form (the VAE attention is bidirectional)");
    LB_REQUIRE(p.B < 65536 && p.H < 65536, "lb_attn_fwd_d512: grid limits");
    constexpr int SMEM = A5_NS * 2 * A5_KT * A5_D * (int)sizeof(f16);          // 128 KiB
    static unsigned long long seen = 0;
    LB_ONCE_PER_DEVICE(seen)                  // (first call on a device happens at record time, outside any capture)
        hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_d512_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    LB_DISPATCH_STMT("lb_attn_fwd_d512",
                     hipLaunchKernelGGL(attn_fwd_d512_kernel, dim3((p.Sq + 63) / 64, p.H, p.B), dim3(256), SMEM, s, p));
}
