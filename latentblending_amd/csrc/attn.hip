// Fused attention forward for head_dim 64 on gfx950 (UNet self- and cross-attention), plus a row
// softmax used by the VAE mid-block attention (head_dim 512, done as GEMM -> softmax -> GEMM).
//
//   O[b,q,h,:] = softmax_k( Q[b,q,h,:] . K[b,k,h,:] * scale ) . V[b,k,h,:]
//
// Q, K and V are read in their natural token-major layout straight out of the fused QKV projection
// ([B*S][3C], one GEMM) or, for cross-attention, out of the conditioning program's K|V projection of the
// text context — no V^T copy exists anywhere.
//
// Block = 4 waves; every wave owns 16*QG query rows (QG = 2: 128 rows per block, QG = 1: 64 rows for grids
// that would not fill the chip).  K and V tiles of KT keys travel global -> LDS with
// `global_load_lds_dwordx4` (no VGPR round trip, no ds_write pass) through an NS-stage ring with counted
// `vmcnt` + one raw `s_barrier` per tile, exactly like the GEMM (gemm_glds.hip): tile t+NS-1 is in flight
// while tile t is consumed.  LDS images are [key][64] fp16 rows of 128 B whose 16-B chunks are XOR-swizzled
// by (key & 7); the swizzle is applied to the per-lane GLOBAL source address (the LDS side of the DMA is
// lane-linear).
//
// Both products run on v_mfma_f32_16x16x32_f16 with operands arranged so that P never moves between lanes:
//   S^T = K . Q^T   -> lane (q = lane&15, g = lane>>4) holds S for keys {16 kb + 4g + r}; K fragments are
//                      conflict-free ds_read_b128 of the swizzled rows; a K/V fragment feeds QG MFMAs.
//   O^T = V^T . P^T -> MFMA k-slot (g, j) is mapped to key 32 ks + 16 (j>>2) + 4g + (j&3): exactly the keys whose
//                      probabilities the lane already holds.  The V^T operand comes from the row-major V tile
//                      through the LDS transpose read `ds_read_b64_tr_b16` (each 16-lane group fetches a
//                      [4 keys][16 d] block; lane m receives column m), two reads per fragment.
// Online softmax in the exp2 domain: e = v_exp_f32(fma(s, scale*log2e, -m)) with a DEFERRED rescale (the running
// maximum moves only when a query outgrows it by 2^8), row sums on the matrix pipe (an all-ones V^T block);
// masking (context padding 77 -> 80, ragged last tile) only in the last tile, behind a wave-uniform branch.
// Built with -fno-honor-nans (no canonicalising v_max around every fmaxf) and the VGPR form of the MFMAs
// (accumulators are VALU operands of the softmax: no v_accvgpr_read / _write traffic).  KT = 64 streams long sequences;
// KT = 96 holds a whole short sequence (cross-attention: 80 keys) in ONE tile: no second, mostly empty tile.
//
// Replaces diffusers' AttnProcessor2_0 / F.scaled_dot_product_attention inside the UNet call at
// /root/reference/latentblending/diffusers_holder.py:336.
#include "lb_common.h"
#include "../../include/lb_hip.h"

#define ATT_D 64
#define ATT_DEFER 8.0f      // log2 units: the running max is only raised when a score exceeds it by more than this

typedef __attribute__((address_space(1))) const void* att_gptr_t;
typedef __attribute__((address_space(3))) void* att_lptr_t;
typedef __fp16 att_h4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef __attribute__((address_space(3))) att_h4* att_h4_lptr;

template <int N> struct AttInt { static constexpr int value = N; };

template <int N> __device__ __forceinline__ void att_wait_vm_barrier() {
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}

// [4 keys][16 d] block of a row-major tile -> lane m of the 16-lane group gets column m (4 keys) (gfx950 LDS transpose
// read).  Inline asm on purpose: through the builtin the compiler treats the read as a possible alias of the
// in-flight LDS-DMA of the NEXT tile and drains it with `s_waitcnt vmcnt(0)` every tile.  The reads are therefore
// invisible to the compiler's wait-count bookkeeping: att_tr_wait<N>() + sched_barrier order them by hand.
template <int OFF_BYTES> __device__ __forceinline__ f16x4 att_tr_read(unsigned lds_byte_addr) {
    f16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(lds_byte_addr), "n"(OFF_BYTES) : "memory");
    return v;
}
template <int N> __device__ __forceinline__ void att_tr_wait() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);      // (MFMAs are register-only: keep them below the wait)
}
__device__ __forceinline__ unsigned att_lds_addr(const void* p) { return (unsigned)(unsigned long)(att_lptr_t)p; }

typedef _Float16 att_h2 __attribute__((ext_vector_type(2)));

// Block -> (query block, head, sample).  The hardware places consecutive workgroup ids on consecutive XCDs (id % 8), each with its own
// L2: under the 3-D grid of rounds 1-5 (query block fastest) the blocks that stream the SAME K / V of one (sample, head) sat on
// different XCDs, and every XCD fetched that K / V through the fabric for itself - 8 x the bytes at S = 1024 (348 MB per launch at
// B = 17: ~4.5 TB/s of fabric traffic, which is what bounded every form of the kernel at ~80 us).  1-D grid: XCD k serves the (sample,
// head) pairs 8 g + k, all query blocks of a pair back to back on that XCD; a tail of < 8 pairs is laid out linearly.  Bijective.
// p.reserved_ bit 1 = the old order (A/B).
struct AttBlock { int qb, h, b; };
__device__ __forceinline__ AttBlock att_block(const LbAttnParams& p, int rows_per_block) {
    const int nqb = (p.Sq + rows_per_block - 1) / rows_per_block;
    const int id = blockIdx.x;
    int qb, pair;
    if (p.reserved_ & 2) {
        qb = id % nqb;
        pair = id / nqb;
    } else {
        const int n_pairs = p.H * p.B, full = (n_pairs >> 3) * 8 * nqb;      // blocks of the complete groups of 8 pairs
        if (id < full) {
            const int xcd = id & 7, slot = id >> 3;
            qb = slot % nqb;
            pair = (slot / nqb) * 8 + xcd;
        } else {
            const int tail = id - full;
            qb = tail % nqb;
            pair = (n_pairs >> 3) * 8 + tail / nqb;
        }
    }
    AttBlock r;
    r.qb = qb;
    r.h = pair % p.H;
    r.b = pair / p.H;
    return r;
}

template <int KT, int QG, int NS>
__global__ void __launch_bounds__(256) attn_fwd_d64_kernel(const LbAttnParams p) {
    constexpr int NKB = KT / 16;            // 16-key blocks of S^T per tile
    constexpr int NKS = KT / 32;            // 32-key MFMA k-steps of the PV product per tile
    constexpr int STAGE = 2 * KT * ATT_D;   // halves per ring stage: K tile, then V tile
    constexpr int NLK = KT * 8 / 256;       // 16-B chunks per thread and operand and tile
    constexpr int NL = 2 * NLK;             // VMEM->LDS requests per thread and tile
    extern __shared__ __attribute__((aligned(16))) f16 att_lds[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, l16 = lane & 15;
    const AttBlock blk = att_block(p, 64 * QG);
    const int b = blk.b, h = blk.h;
    const int q0 = blk.qb * (64 * QG) + wave * (16 * QG);
    const f16* Q = reinterpret_cast<const f16*>(p.Q);
    const f16* K = reinterpret_cast<const f16*>(p.K);
    const f16* V = reinterpret_cast<const f16*>(p.V);
    f16* O = reinterpret_cast<f16*>(p.O);
    const f16* zero = reinterpret_cast<const f16*>(p.zero_page);
    const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

    // ---- loader: thread (r0 = tid>>3 (+32 i), c = tid&7) owns physical chunk c of tile row r0 + 32 i and fetches
    //      logical chunk c ^ (row & 7); (row & 7) == (r0 & 7) for every i
    const int r0 = tid >> 3;
    const int cl = (tid & 7) ^ (r0 & 7);
    const f16* kbase = K + (long)b * p.Skv * p.ldk + h * ATT_D + cl * 8;
    const f16* vbase = V + (long)b * p.Skv * p.ldv + h * ATT_D + cl * 8;
    auto issue_tile = [&](int t, int st) {
        f16* base = att_lds + st * STAGE;
#pragma unroll
        for (int i = 0; i < NLK; ++i) {
            const int key = t * KT + r0 + 32 * i;
            const f16* src = key < p.Skv ? kbase + (long)key * p.ldk : zero;
            __builtin_amdgcn_global_load_lds((att_gptr_t)src, (att_lptr_t)(base + (wave * 8 + i * 32) * ATT_D), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NLK; ++i) {
            const int key = t * KT + r0 + 32 * i;
            const f16* src = key < p.Skv ? vbase + (long)key * p.ldv : zero;
            __builtin_amdgcn_global_load_lds((att_gptr_t)src, (att_lptr_t)(base + KT * ATT_D + (wave * 8 + i * 32) * ATT_D), 16, 0, 0);
        }
    };

    const int nt = (p.Skv + KT - 1) / KT;
    // ---- prologue: the Q fragments (b operand: k = d = 32 s + 8 g .. +8) are requested FIRST (requests retire in
    //      order: the counted waits of the loop then never wait for a younger request than they mean), then tiles
    //      0 .. NS-2 go in flight ----
    // (rows past Sq re-read the last row - their results are never stored: no exec-masked block per load, behind each of which
    //  hipcc waits for vmcnt(0), i.e. four serialised memory round trips in front of the first tile request until round 5)
    f16x8 qf[QG][2];
#pragma unroll
    for (int qg = 0; qg < QG; ++qg) {
        const int q_row = min(q0 + qg * 16 + l16, p.Sq - 1);
#pragma unroll
        for (int s = 0; s < 2; ++s)
            qf[qg][s] = *reinterpret_cast<const f16x8*>(Q + ((long)b * p.Sq + q_row) * p.ldq + h * ATT_D + s * 32 + g * 8);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < (NS > 1 ? NS - 1 : 1); ++s) issue_tile(s, s);       // (NS == 1: the single-tile form, its one tile)

    // ---- loop-invariant LDS offsets (halves, relative to the stage base) ----
    // K fragment (a operand): row kb*16 + l16, logical chunk s*4 + g
    int koff[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) koff[s] = l16 * ATT_D + (((s * 4 + g) ^ (l16 & 7)) << 3);
    // V^T fragment through the transpose read: lane m = l16 supplies the address of key 4g + (m>>2) (+ 16 hh + 32 ks),
    // d = 16 dt + 4 (m&3): logical chunk 2 dt + ((m&3)>>1), 8-byte half (m&1); swizzle (key & 7) = (4 (g&1) + (m>>2))
    const int vrow = 4 * g + (l16 >> 2);
    const int vsw = vrow & 7;
    int voff[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
        voff[dt] = KT * ATT_D + vrow * ATT_D + (((2 * dt + ((l16 & 3) >> 1)) ^ vsw) << 3) + (l16 & 1) * 4;

    // O^T accumulators plus ONE extra 16-row block whose V^T operand is all ones: its rows accumulate the softmax
    // denominators sum_k P[q][k] on the (under-used) matrix pipe instead of 32 VALU adds per tile, from the very
    // fp16-rounded probabilities the numerator uses, already summed over all keys (no cross-lane reduction at the end)
    f32x4 ot[QG][4], lt[QG];
    float m_run[QG];
#pragma unroll
    for (int qg = 0; qg < QG; ++qg) {
        m_run[qg] = -INFINITY;
        lt[qg] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) ot[qg][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const f16x8 ones8 = {(f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f};
    const float sc = p.scale * 1.44269504088896340736f;           // fold log2(e): exp2 domain

    int st = 0;
    for (int t = 0; t < nt; ++t) {
        att_wait_vm_barrier<(NS > 1 ? NS - 2 : 0) * NL>();     // tile t landed everywhere; stage (t-1) % NS is free everywhere
        if constexpr (NS > 1) {
            int refill = st - 1;
            if (refill < 0) refill += NS;
            issue_tile(t + NS - 1, refill);       // (masked to the zero page past the end: the counts stay constant)
        }
        const f16* Ks = att_lds + st * STAGE;

        // ---- S^T = K . Q^T ----
        f32x4 sacc[QG][NKB];
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            if (KT > 64 && kb * 16 >= p.Skv) {    // single-tile form: key blocks beyond the sequence are never computed
#pragma unroll
                for (int qg = 0; qg < QG; ++qg) sacc[qg][kb] = (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
                continue;
            }
            const f16x8 kf0 = *reinterpret_cast<const f16x8*>(Ks + kb * 16 * ATT_D + koff[0]);
            const f16x8 kf1 = *reinterpret_cast<const f16x8*>(Ks + kb * 16 * ATT_D + koff[1]);
#pragma unroll
            for (int qg = 0; qg < QG; ++qg) {
                f32x4 a = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf0, qf[qg][0], (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                sacc[qg][kb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf1, qf[qg][1], a, 0, 0, 0);
            }
        }
        // ---- online softmax (this lane: one query per group, keys 16 kb + 4 g + r of the tile) ----
        // wave-uniform: only the last tile(s) mask; a causal wave masks from the tile holding its first query on
        const bool ragged = (t + 1) * KT > p.Skv_valid || (p.causal && (t + 1) * KT > q0);
        f16x8 pf[QG][NKS];
#pragma unroll
        for (int qg = 0; qg < QG; ++qg) {
            if (ragged) {
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                    {
                        const int key = t * KT + kb * 16 + 4 * g + r;
                        if (key >= p.Skv_valid || (p.causal && key > q0 + qg * 16 + l16)) sacc[qg][kb][r] = -INFINITY;
                    }
            }
            float mx = -INFINITY;
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
                mx = fmaxf(fmaxf(mx, fmaxf(sacc[qg][kb][0], sacc[qg][kb][1])), fmaxf(sacc[qg][kb][2], sacc[qg][kb][3]));
            mx = fmaxf(mx, __shfl_xor(mx, 16, LB_WAVE));
            mx = fmaxf(mx, __shfl_xor(mx, 32, LB_WAVE));
            // deferred rescale: the running maximum only moves (and O, l are only rescaled) when some query of the wave
            // outgrew it by more than 2^ATT_DEFER; otherwise probabilities are taken against the old maximum and stay
            // <= 2^ATT_DEFER (exact in the fp32 accumulators, the same relative precision in fp16)
            const float m_cand = mx * sc;                          // sc > 0: max commutes with the scaling
            if (__any(m_cand > m_run[qg] + ATT_DEFER)) {
                const float m_new = fmaxf(m_run[qg], m_cand);
                const float m_fin = m_new == -INFINITY ? 0.f : m_new;          // fully masked so far: keep zeros
                const float alpha = __builtin_amdgcn_exp2f(m_run[qg] - m_fin); // m_run = -inf -> 0
                m_run[qg] = m_new;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) ot[qg][dt][r] *= alpha;
#pragma unroll
                for (int r = 0; r < 4; ++r) lt[qg][r] *= alpha;
            }
            const float m_use = m_run[qg] == -INFINITY ? 0.f : m_run[qg];
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    pf[qg][kb >> 1][(kb & 1) * 4 + r] = (f16)__builtin_amdgcn_exp2f(__builtin_fmaf(sacc[qg][kb][r], sc, -m_use));
        }
        // ---- O^T += V^T . P^T ----  (V^T fragments of k-step ks+1 are requested before the MFMAs of k-step ks)
        {
            const unsigned vb = att_lds_addr(Ks);
            f16x4 vlo[NKS][4], vhi[NKS][4];
            auto request = [&](auto ks_c) {
                constexpr int ks = decltype(ks_c)::value;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    vlo[ks][dt] = att_tr_read<(32 * ks) * ATT_D * 2>(vb + voff[dt] * 2);
                    vhi[ks][dt] = att_tr_read<(32 * ks + 16) * ATT_D * 2>(vb + voff[dt] * 2);
                }
            };
            auto multiply = [&](auto ks_c) {
                constexpr int ks = decltype(ks_c)::value;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const f16x8 vf = {vlo[ks][dt][0], vlo[ks][dt][1], vlo[ks][dt][2], vlo[ks][dt][3],
                                      vhi[ks][dt][0], vhi[ks][dt][1], vhi[ks][dt][2], vhi[ks][dt][3]};
#pragma unroll
                    for (int qg = 0; qg < QG; ++qg)
                        ot[qg][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf[qg][ks], ot[qg][dt], 0, 0, 0);
                }
#pragma unroll
                for (int qg = 0; qg < QG; ++qg)
                    lt[qg] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones8, pf[qg][ks], lt[qg], 0, 0, 0);
            };
            request(AttInt<0>{});
            request(AttInt<1>{});
            att_tr_wait<8>();
            multiply(AttInt<0>{});
            if constexpr (NKS == 3) {
                request(AttInt<2>{});        // (keys 64..95: zero rows / zero probabilities when the sequence is shorter)
                att_tr_wait<8>();
                multiply(AttInt<1>{});
                att_tr_wait<0>();
                multiply(AttInt<2>{});
            } else {
                att_tr_wait<0>();
                multiply(AttInt<1>{});
            }
        }
        st = st + 1 == NS ? 0 : st + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the masked tail requests before the block may exit

    // Output: lane (g, l16) holds O[q = l16][d = 16 dt + 4 g + r].  WIDE form (p.reserved_ & 1: ldo % 8 == 0, O 16-byte aligned): the
    // quads of dt and dt + 1 are paired with the neighbouring 16-lane row through v_permlane16_swap (the exchange of the GEMM
    // epilogue, lb_gemm.h), so that every lane owns 8 CONSECUTIVE halves: two 16-byte stores per query group instead of four 8-byte ones.
    const bool wide = (p.reserved_ & 1) != 0;
#pragma unroll
    for (int qg = 0; qg < QG; ++qg) {
        const float l = lt[qg][0];                                 // (every row of the ones-block holds the same sum)
        const float inv = l > 0.f ? 1.f / l : 0.f;
        const int q_row = q0 + qg * 16 + l16;
        f16* orow = O + ((long)b * p.Sq + (q_row < p.Sq ? q_row : p.Sq - 1)) * p.ldo + h * ATT_D;
        if (wide) {                                                // (wave-uniform; every lane takes part in the exchange)
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                unsigned u[2][2];
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const f32x4 a = ot[qg][2 * pr + k];
                    u[k][0] = __builtin_bit_cast(unsigned, (att_h2){(f16)(a[0] * inv), (f16)(a[1] * inv)});
                    u[k][1] = __builtin_bit_cast(unsigned, (att_h2){(f16)(a[2] * inv), (f16)(a[3] * inv)});
                }
                const auto r0 = __builtin_amdgcn_permlane16_swap(u[0][0], u[1][0], false, false);
                const auto r1 = __builtin_amdgcn_permlane16_swap(u[0][1], u[1][1], false, false);
                const int n = (2 * pr + 1) * 16 + 4 * g;
                const int nst = (g & 1) ? n - 4 : n - 16;
                typedef unsigned att_u4 __attribute__((ext_vector_type(4)));
                if (q_row < p.Sq) *reinterpret_cast<att_u4*>(orow + nst) = (att_u4){r0[0], r1[0], r0[1], r1[1]};
            }
            continue;
        }
        if (q_row < p.Sq) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const f16x4 o = {(f16)(ot[qg][dt][0] * inv), (f16)(ot[qg][dt][1] * inv), (f16)(ot[qg][dt][2] * inv),
                                 (f16)(ot[qg][dt][3] * inv)};
                *reinterpret_cast<f16x4*>(orow + dt * 16 + 4 * g) = o;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Round 6: the streaming form (64-key tiles, 3-stage ring) with the VALU work per tile halved.  PMC of the kernel above on the UNet's
// self-attention (profiles/r04_attention_pmc.json): 233 VALU instructions per (wave, tile) = 1,063 issue cycles (a wave64 VALU
// instruction occupies its SIMD for 4 cycles, v_exp_f32 for 8) against 576 cycles of MFMA - with three waves per SIMD the VALU
// pipe is 75 % busy and the matrix pipe 40 %: the kernel is VALU-bound, not matrix-bound.  What went:
//   * the 32 v_fma (score * scale * log2e - max) per tile: Q is multiplied by scale * log2e ONCE in the prologue (one extra fp16
//     rounding of Q, of the size of the rounding the projection already applied), and -max enters as the ACCUMULATOR INPUT of the
//     first QK^T MFMA, so the matrix pipe returns (score - max) in the exp2 domain and the probabilities are v_exp_f32 of the
//     accumulators as they are;
//   * the two ds_bpermute round trips of every row-max: v_permlane16_swap / v_permlane32_swap (VALU, no LDS crossbar);
//   * ~50 instructions of per-tile bookkeeping: 64-bit address arithmetic behind exec-masked branches for every direct-to-LDS
//     request (now a uniform base + a 32-bit running offset clamped to the last row - rows past the sequence re-read the last row,
//     their probabilities are exactly 0), the key indices of the mask (now inside the ragged branch only);
//   * the K fragments of a tile are read into registers up front (the budget of three waves per SIMD is 168 VGPRs; the old kernel
//     used 133 and exposed one LDS latency per 16-key block).
// The running maximum is kept in the same deferred form (it moves when a score outgrows it by 2^ATT_DEFER); a move re-bases the
// scores already computed against the old maximum inside the (rare) branch.
// ------------------------------------------------------------------------------------------
template <int QG>
__global__ void __launch_bounds__(256) attn_fwd_d64_stream_kernel(const LbAttnParams p) {
    constexpr int KT = 64, NS = 3, NKB = KT / 16, NKS = KT / 32, NL = 4;
    constexpr int STAGE = 2 * KT * ATT_D;   // halves per ring stage: K tile, then V tile
    extern __shared__ __attribute__((aligned(16))) f16 att_lds[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, l16 = lane & 15;
    const AttBlock blk = att_block(p, 64 * QG);
    const int b = blk.b, h = blk.h;
    const int q0 = __builtin_amdgcn_readfirstlane(blk.qb * (64 * QG) + wave * (16 * QG));      // (wave-uniform: keep it scalar)
    const f16* Q = reinterpret_cast<const f16*>(p.Q);
    f16* O = reinterpret_cast<f16*>(p.O);
    const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    const int lane_k = 4 * g, lane_c = 4 * g - l16;     // masks compare these lane constants with per-(tile, block, r) scalars

    // ---- loader: thread (r0 = tid>>3 (+32), c = tid&7) owns physical chunk c of tile rows r0, r0 + 32 and fetches logical chunk
    //      c ^ (row & 7).  Uniform base pointer + 32-bit element offset, advanced by one tile per request and clamped to the last row.
    const int r0 = tid >> 3;
    const int cl = (tid & 7) ^ (r0 & 7);
    const f16* kbase = reinterpret_cast<const f16*>(p.K) + (long)b * p.Skv * p.ldk + h * ATT_D;
    const f16* vbase = reinterpret_cast<const f16*>(p.V) + (long)b * p.Skv * p.ldv + h * ATT_D;
    const unsigned kmax = (unsigned)(p.Skv - 1) * p.ldk + cl * 8, vmax = (unsigned)(p.Skv - 1) * p.ldv + cl * 8;
    unsigned ko[2], vo[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        ko[i] = min((unsigned)(r0 + 32 * i) * p.ldk + cl * 8, kmax);
        vo[i] = min((unsigned)(r0 + 32 * i) * p.ldv + cl * 8, vmax);
    }
    const unsigned kstep = (unsigned)KT * p.ldk, vstep = (unsigned)KT * p.ldv;
    auto issue_tile = [&](int st) {
        f16* base = att_lds + st * STAGE;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            __builtin_amdgcn_global_load_lds((att_gptr_t)(kbase + ko[i]), (att_lptr_t)(base + (wave * 8 + i * 32) * ATT_D), 16, 0, 0);
            ko[i] = min(ko[i] + kstep, kmax);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            __builtin_amdgcn_global_load_lds((att_gptr_t)(vbase + vo[i]), (att_lptr_t)(base + KT * ATT_D + (wave * 8 + i * 32) * ATT_D), 16, 0, 0);
            vo[i] = min(vo[i] + vstep, vmax);
        }
    };

    const int nt = (p.Skv + KT - 1) / KT;
    // ---- prologue: Q fragments (b operand: k = d = 32 s + 8 g .. +8) first, then tiles 0 .. NS-2 in flight; Q is carried into the
    //      exp2 domain here: q * (scale * log2 e), rounded to fp16 once ----
    const float sc = p.scale * 1.44269504088896340736f;
    f16x8 qraw[QG][2];      // (rows past Sq re-read the last row - never stored: no exec-masked load blocks, see attn_fwd_d64_kernel)
#pragma unroll
    for (int qg = 0; qg < QG; ++qg) {
        const int q_row = min(q0 + qg * 16 + l16, p.Sq - 1);
#pragma unroll
        for (int s = 0; s < 2; ++s)
            qraw[qg][s] = *reinterpret_cast<const f16x8*>(Q + ((long)b * p.Sq + q_row) * p.ldq + h * ATT_D + s * 32 + g * 8);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) issue_tile(s);
    f16x8 qf[QG][2];
#pragma unroll
    for (int qg = 0; qg < QG; ++qg)
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int e = 0; e < 8; ++e) qf[qg][s][e] = (f16)((float)qraw[qg][s][e] * sc);

    // ---- loop-invariant LDS offsets (halves, relative to the stage base): as in attn_fwd_d64_kernel ----
    int koff[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) koff[s] = l16 * ATT_D + (((s * 4 + g) ^ (l16 & 7)) << 3);
    const int vrow = 4 * g + (l16 >> 2);
    const int vsw = vrow & 7;
    int voff[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
        voff[dt] = KT * ATT_D + vrow * ATT_D + (((2 * dt + ((l16 & 3) >> 1)) ^ vsw) << 3) + (l16 & 1) * 4;

    f32x4 ot[QG][4], lt[QG], negm[QG];      // negm: -running maximum (exp2 domain), the accumulator input of QK^T
    float m_use[QG];
#pragma unroll
    for (int qg = 0; qg < QG; ++qg) {
        m_use[qg] = 0.f;
        negm[qg] = (f32x4){0.f, 0.f, 0.f, 0.f};
        lt[qg] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) ot[qg][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const f16x8 ones8 = {(f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f};

    int st = 0;
    for (int t = 0; t < nt; ++t) {
        att_wait_vm_barrier<(NS - 2) * NL>();     // tile t landed everywhere; stage (t-1) % NS is free everywhere
        {
            int refill = st - 1;
            if (refill < 0) refill += NS;
            issue_tile(refill);                   // (clamped to the last row past the end: the counts stay constant)
        }
        const f16* Ks = att_lds + st * STAGE;

        // ---- S^T - max = K . Q^T + (-max): every K fragment of the tile is requested before the first MFMA ----
        f16x8 kf[NKB][2];
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int s = 0; s < 2; ++s) kf[kb][s] = *reinterpret_cast<const f16x8*>(Ks + kb * 16 * ATT_D + koff[s]);
        f32x4 sacc[QG][NKB];
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int qg = 0; qg < QG; ++qg) {
                const f32x4 a = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[kb][0], qf[qg][0], negm[qg], 0, 0, 0);
                sacc[qg][kb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[kb][1], qf[qg][1], a, 0, 0, 0);
            }
        // ---- online softmax (this lane: one query per group, keys 16 kb + 4 g + r of the tile) ----
        const bool ragged = (t + 1) * KT > p.Skv_valid || (p.causal && (t + 1) * KT > q0);      // wave-uniform
        f16x8 pf[QG][NKS];
#pragma unroll
        for (int qg = 0; qg < QG; ++qg) {
            if (ragged) {       // key = t KT + 16 kb + 4 g + r is masked when key >= Skv_valid, or (causal) key > q0 + 16 qg + l16
                const int vrel = p.Skv_valid - t * KT, crel = q0 + qg * 16 - t * KT;
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (lane_k >= vrel - kb * 16 - r || (p.causal && lane_c > crel - kb * 16 - r)) sacc[qg][kb][r] = -INFINITY;
            }
            float mx = fmaxf(fmaxf(fmaxf(sacc[qg][0][0], sacc[qg][0][1]), sacc[qg][0][2]), sacc[qg][0][3]);
#pragma unroll
            for (int kb = 1; kb < NKB; ++kb)      // (a chain: two v_max3 per key block)
                mx = fmaxf(fmaxf(fmaxf(fmaxf(mx, sacc[qg][kb][0]), sacc[qg][kb][1]), sacc[qg][kb][2]), sacc[qg][kb][3]);
            {   // the query's other three 16-lane rows: permlane swaps instead of ds_bpermute round trips
                const unsigned u = __float_as_uint(mx);
                const auto r16 = __builtin_amdgcn_permlane16_swap(u, u, false, false);
                const unsigned v = __float_as_uint(fmaxf(__uint_as_float(r16[0]), __uint_as_float(r16[1])));
                const auto r32 = __builtin_amdgcn_permlane32_swap(v, v, false, false);
                mx = fmaxf(__uint_as_float(r32[0]), __uint_as_float(r32[1]));
            }
            // mx is RELATIVE to the running maximum.  Deferred rescale: the maximum moves (and O, l are rescaled, and the scores
            // of this tile re-based) only when some query of the wave outgrew it by more than 2^ATT_DEFER; the first tile sets it
            // (every query sees at least key 0 there: Skv_valid > 0, causal k <= q).
            if (t == 0 || __any(mx > ATT_DEFER)) {
                const float m_abs = mx + m_use[qg];
                const float m_new = t == 0 ? m_abs : fmaxf(m_use[qg], m_abs);
                const float delta = m_new - m_use[qg];
                const float alpha = t == 0 ? 1.f : __builtin_amdgcn_exp2f(-delta);
                m_use[qg] = m_new;
                negm[qg] = (f32x4){-m_new, -m_new, -m_new, -m_new};
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) sacc[qg][kb][r] -= delta;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) ot[qg][dt][r] *= alpha;
#pragma unroll
                for (int r = 0; r < 4; ++r) lt[qg][r] *= alpha;
            }
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    pf[qg][kb >> 1][(kb & 1) * 4 + r] = (f16)__builtin_amdgcn_exp2f(sacc[qg][kb][r]);
        }
        // ---- O^T += V^T . P^T ----  (V^T fragments of both k-steps are requested before the first MFMA)
        {
            const unsigned vb = att_lds_addr(Ks);
            f16x4 vlo[NKS][4], vhi[NKS][4];
            auto request = [&](auto ks_c) {
                constexpr int ks = decltype(ks_c)::value;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    vlo[ks][dt] = att_tr_read<(32 * ks) * ATT_D * 2>(vb + voff[dt] * 2);
                    vhi[ks][dt] = att_tr_read<(32 * ks + 16) * ATT_D * 2>(vb + voff[dt] * 2);
                }
            };
            auto multiply = [&](auto ks_c) {
                constexpr int ks = decltype(ks_c)::value;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const f16x8 vf = {vlo[ks][dt][0], vlo[ks][dt][1], vlo[ks][dt][2], vlo[ks][dt][3],
                                      vhi[ks][dt][0], vhi[ks][dt][1], vhi[ks][dt][2], vhi[ks][dt][3]};
#pragma unroll
                    for (int qg = 0; qg < QG; ++qg)
                        ot[qg][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf[qg][ks], ot[qg][dt], 0, 0, 0);
                }
#pragma unroll
                for (int qg = 0; qg < QG; ++qg)
                    lt[qg] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones8, pf[qg][ks], lt[qg], 0, 0, 0);
            };
            request(AttInt<0>{});
            request(AttInt<1>{});
            att_tr_wait<8>();
            multiply(AttInt<0>{});
            att_tr_wait<0>();
            multiply(AttInt<1>{});
        }
        st = st + 1 == NS ? 0 : st + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the tail requests before the block may exit

    // Output: as attn_fwd_d64_kernel (lane (g, l16) holds O[q = l16][d = 16 dt + 4 g + r]; paired 16-byte stores in the wide form)
    const bool wide = (p.reserved_ & 1) != 0;
#pragma unroll
    for (int qg = 0; qg < QG; ++qg) {
        const float l = lt[qg][0];
        const float inv = l > 0.f ? 1.f / l : 0.f;
        const int q_row = q0 + qg * 16 + l16;
        f16* orow = O + ((long)b * p.Sq + (q_row < p.Sq ? q_row : p.Sq - 1)) * p.ldo + h * ATT_D;
        if (wide) {
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                unsigned u[2][2];
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const f32x4 a = ot[qg][2 * pr + k];
                    u[k][0] = __builtin_bit_cast(unsigned, (att_h2){(f16)(a[0] * inv), (f16)(a[1] * inv)});
                    u[k][1] = __builtin_bit_cast(unsigned, (att_h2){(f16)(a[2] * inv), (f16)(a[3] * inv)});
                }
                const auto r0_ = __builtin_amdgcn_permlane16_swap(u[0][0], u[1][0], false, false);
                const auto r1_ = __builtin_amdgcn_permlane16_swap(u[0][1], u[1][1], false, false);
                const int n = (2 * pr + 1) * 16 + 4 * g;
                const int nst = (g & 1) ? n - 4 : n - 16;
                typedef unsigned att_u4 __attribute__((ext_vector_type(4)));
                if (q_row < p.Sq) *reinterpret_cast<att_u4*>(orow + nst) = (att_u4){r0_[0], r1_[0], r0_[1], r1_[1]};
            }
            continue;
        }
        if (q_row < p.Sq) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const f16x4 o = {(f16)(ot[qg][dt][0] * inv), (f16)(ot[qg][dt][1] * inv), (f16)(ot[qg][dt][2] * inv),
                                 (f16)(ot[qg][dt][3] * inv)};
                *reinterpret_cast<f16x4*>(orow + dt * 16 + 4 * g) = o;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Round 6: PING-PONG form of the streaming kernel.  The three co-resident blocks of attn_fwd_d64_stream_kernel run the same
// instruction stream from the same start: their waves meet in the same phase on a SIMD - all of them multiplying (matrix pipe
// contended, VALU idle), then all of them exponentiating (VALU contended, matrix pipe idle) - and a tile costs a SIMD the SUM of
// both phases (measured: 1,890 cycles per (wave, tile) for 576 of MFMA + ~650 of VALU).  Here a block has 8 waves = two groups
// of four; waves w and w + 4 share a SIMD, and group B runs exactly one phase behind group A, one s_barrier per phase:
//        phase      0       1        2         3         4
//        group A   QK(0)   SM(0)   PV(0)+QK(1) SM(1)   PV(1)+QK(2) ...          SM = softmax (VALU), PV / QK = MFMA
//        group B    -      QK(0)   SM(0)     PV(0)+QK(1) SM(1)     ...
// so that a SIMD always holds one wave in its matrix phase and one in its VALU phase.  Both groups stream the SAME 64-key tiles
// (the block owns 128 QG consecutive queries of one head): every thread requests one 16-byte chunk of K and one of V per tile
// (512 threads = 64 rows x 8 chunks), in the even phases; a 4-stage ring: stage t % 4 is read in phases 2t .. 2t + 3 (K_t by QK(t)
// of both groups, V_t by PV(t) of both groups), refilled with tile t + 3's successor in phase 2t + 2 ... i.e. tile t + 3 is
// requested in phase 2t + 2 into the stage tile t - 1 left in phase 2t + 1, and must have landed by phase 2t + 6: every wave waits
// vmcnt(2) (one younger tile may still be in flight) at the end of the odd phases.  The arithmetic per wave is that of
// attn_fwd_d64_stream_kernel (same instruction order inside QK / SM / PV): bit-identical outputs.
// ------------------------------------------------------------------------------------------
template <int QG>
__global__ void __launch_bounds__(512, QG == 1 ? 4 : 2) attn_fwd_d64_pp_kernel(const LbAttnParams p) {
    constexpr int KT = 64, NS = 4, NKB = KT / 16, NKS = KT / 32;
    constexpr int STAGE = 2 * KT * ATT_D;   // halves per ring stage: K tile, then V tile
    extern __shared__ __attribute__((aligned(16))) f16 att_lds[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, l16 = lane & 15;
    const AttBlock blk = att_block(p, 128 * QG);
    const int b = blk.b, h = blk.h;
    const bool grpB = __builtin_amdgcn_readfirstlane(wave >> 2) != 0;
    const int q0 = __builtin_amdgcn_readfirstlane(blk.qb * (128 * QG) + wave * (16 * QG));
    const f16* Q = reinterpret_cast<const f16*>(p.Q);
    f16* O = reinterpret_cast<f16*>(p.O);
    const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    const int lane_k = 4 * g, lane_c = 4 * g - l16;

    // ---- loader: thread (row = tid>>3, c = tid&7) owns physical chunk c of tile row `row` (K and V) and fetches logical chunk c ^ (row & 7)
    const int r0 = tid >> 3;
    const int cl = (tid & 7) ^ (r0 & 7);
    const f16* kbase = reinterpret_cast<const f16*>(p.K) + (long)b * p.Skv * p.ldk + h * ATT_D;
    const f16* vbase = reinterpret_cast<const f16*>(p.V) + (long)b * p.Skv * p.ldv + h * ATT_D;
    const unsigned kmax = (unsigned)(p.Skv - 1) * p.ldk + cl * 8, vmax = (unsigned)(p.Skv - 1) * p.ldv + cl * 8;
    unsigned ko = min((unsigned)r0 * p.ldk + cl * 8, kmax), vo = min((unsigned)r0 * p.ldv + cl * 8, vmax);
    const unsigned kstep = (unsigned)KT * p.ldk, vstep = (unsigned)KT * p.ldv;
    auto issue_tile = [&](int st) {
        f16* base = att_lds + st * STAGE;
        __builtin_amdgcn_global_load_lds((att_gptr_t)(kbase + ko), (att_lptr_t)(base + (wave * 8) * ATT_D), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((att_gptr_t)(vbase + vo), (att_lptr_t)(base + KT * ATT_D + (wave * 8) * ATT_D), 16, 0, 0);
        ko = min(ko + kstep, kmax);
        vo = min(vo + vstep, vmax);
    };

    const int nt = (p.Skv + KT - 1) / KT;
    const float sc = p.scale * 1.44269504088896340736f;
    f16x8 qraw[QG][2];      // (rows past Sq re-read the last row - never stored: no exec-masked load blocks, see attn_fwd_d64_kernel)
#pragma unroll
    for (int qg = 0; qg < QG; ++qg) {
        const int q_row = min(q0 + qg * 16 + l16, p.Sq - 1);
#pragma unroll
        for (int s = 0; s < 2; ++s)
            qraw[qg][s] = *reinterpret_cast<const f16x8*>(Q + ((long)b * p.Sq + q_row) * p.ldq + h * ATT_D + s * 32 + g * 8);
    }
    __builtin_amdgcn_sched_barrier(0);
    issue_tile(0);      // (tiles 1 and 2 follow behind the first K-fragment reads: hipcc drains every direct-to-LDS request still in
    //                     flight in front of the first ds_read of the straight-line code that issued it)
    f16x8 qf[QG][2];
#pragma unroll
    for (int qg = 0; qg < QG; ++qg)
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int e = 0; e < 8; ++e) qf[qg][s][e] = (f16)((float)qraw[qg][s][e] * sc);

    int koff[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) koff[s] = l16 * ATT_D + (((s * 4 + g) ^ (l16 & 7)) << 3);
    const int vrow = 4 * g + (l16 >> 2);
    const int vsw = vrow & 7;
    int voff[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
        voff[dt] = KT * ATT_D + vrow * ATT_D + (((2 * dt + ((l16 & 3) >> 1)) ^ vsw) << 3) + (l16 & 1) * 4;

    f32x4 ot[QG][4], lt[QG], negm[QG];
    float m_use[QG];
#pragma unroll
    for (int qg = 0; qg < QG; ++qg) {
        m_use[qg] = 0.f;
        negm[qg] = (f32x4){0.f, 0.f, 0.f, 0.f};
        lt[qg] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) ot[qg][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const f16x8 ones8 = {(f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f};

    f32x4 sacc[QG][NKB];
    f16x8 kf[NKB][2];
    auto read_k = [&](int st) {         // every K fragment of a tile (conflict-free ds_read_b128 of the swizzled rows)
        const f16* Ks = att_lds + st * STAGE;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int s = 0; s < 2; ++s) kf[kb][s] = *reinterpret_cast<const f16x8*>(Ks + kb * 16 * ATT_D + koff[s]);
    };
    auto qk = [&]() {                   // S^T - max = K . Q^T + (-max)
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int qg = 0; qg < QG; ++qg) {
                const f32x4 a = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[kb][0], qf[qg][0], negm[qg], 0, 0, 0);
                sacc[qg][kb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[kb][1], qf[qg][1], a, 0, 0, 0);
            }
    };

    // ---- phase 0 (group A) / phase 1 (group B): QK(0) ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // tile 0 (this thread's part)
    if (grpB) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_barrier();                         // ... everybody's part
    read_k(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    issue_tile(1);
    issue_tile(2);
    qk();
    if (grpB) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");      // (end of an odd phase: tile 1)

    int st = 0;                         // ring stage of tile t
    for (int t = 0; t < nt; ++t) {
        // ================= SM(t): VALU phase =================
        __builtin_amdgcn_s_barrier();
        if (grpB) issue_tile((st + 3) & 3);               // (even phase: tile t + 3 into the stage tile t - 1 has left)
        const bool ragged = (t + 1) * KT > p.Skv_valid || (p.causal && (t + 1) * KT > q0);      // wave-uniform
        f16x8 pf[QG][NKS];
#pragma unroll
        for (int qg = 0; qg < QG; ++qg) {
            if (ragged) {
                const int vrel = p.Skv_valid - t * KT, crel = q0 + qg * 16 - t * KT;
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (lane_k >= vrel - kb * 16 - r || (p.causal && lane_c > crel - kb * 16 - r)) sacc[qg][kb][r] = -INFINITY;
            }
            float mx = fmaxf(fmaxf(fmaxf(sacc[qg][0][0], sacc[qg][0][1]), sacc[qg][0][2]), sacc[qg][0][3]);
#pragma unroll
            for (int kb = 1; kb < NKB; ++kb)
                mx = fmaxf(fmaxf(fmaxf(fmaxf(mx, sacc[qg][kb][0]), sacc[qg][kb][1]), sacc[qg][kb][2]), sacc[qg][kb][3]);
            {
                const unsigned u = __float_as_uint(mx);
                const auto r16 = __builtin_amdgcn_permlane16_swap(u, u, false, false);
                const unsigned v = __float_as_uint(fmaxf(__uint_as_float(r16[0]), __uint_as_float(r16[1])));
                const auto r32 = __builtin_amdgcn_permlane32_swap(v, v, false, false);
                mx = fmaxf(__uint_as_float(r32[0]), __uint_as_float(r32[1]));
            }
            if (t == 0 || __any(mx > ATT_DEFER)) {
                const float m_abs = mx + m_use[qg];
                const float m_new = t == 0 ? m_abs : fmaxf(m_use[qg], m_abs);
                const float delta = m_new - m_use[qg];
                const float alpha = t == 0 ? 1.f : __builtin_amdgcn_exp2f(-delta);
                m_use[qg] = m_new;
                negm[qg] = (f32x4){-m_new, -m_new, -m_new, -m_new};
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) sacc[qg][kb][r] -= delta;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) ot[qg][dt][r] *= alpha;
#pragma unroll
                for (int r = 0; r < 4; ++r) lt[qg][r] *= alpha;
            }
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    pf[qg][kb >> 1][(kb & 1) * 4 + r] = (f16)__builtin_amdgcn_exp2f(sacc[qg][kb][r]);
        }
        if (!grpB) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");     // (end of an odd phase: tile t + 1)
        // ================= PV(t) + QK(t + 1): matrix phase =================
        __builtin_amdgcn_s_barrier();
        if (!grpB) issue_tile((st + 3) & 3);              // (even phase)
        read_k((st + 1) & 3);           // K_{t+1}, requested ahead of the V^T fragments (past the end: the clamped tile, scores unused)
        {
            const unsigned vb = att_lds_addr(att_lds + st * STAGE);
            f16x4 vlo[NKS][4], vhi[NKS][4];
            auto request = [&](auto ks_c) {
                constexpr int ks = decltype(ks_c)::value;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    vlo[ks][dt] = att_tr_read<(32 * ks) * ATT_D * 2>(vb + voff[dt] * 2);
                    vhi[ks][dt] = att_tr_read<(32 * ks + 16) * ATT_D * 2>(vb + voff[dt] * 2);
                }
            };
            auto multiply = [&](auto ks_c) {
                constexpr int ks = decltype(ks_c)::value;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const f16x8 vf = {vlo[ks][dt][0], vlo[ks][dt][1], vlo[ks][dt][2], vlo[ks][dt][3],
                                      vhi[ks][dt][0], vhi[ks][dt][1], vhi[ks][dt][2], vhi[ks][dt][3]};
#pragma unroll
                    for (int qg = 0; qg < QG; ++qg)
                        ot[qg][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf[qg][ks], ot[qg][dt], 0, 0, 0);
                }
#pragma unroll
                for (int qg = 0; qg < QG; ++qg)
                    lt[qg] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones8, pf[qg][ks], lt[qg], 0, 0, 0);
            };
            request(AttInt<0>{});
            request(AttInt<1>{});
            att_tr_wait<8>();           // (LDS returns in order: the 8 K fragments and the first 8 V^T reads have arrived)
            multiply(AttInt<0>{});
            att_tr_wait<0>();
            multiply(AttInt<1>{});
        }
        st = (st + 1) & 3;
        qk();
        if (grpB) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");      // (end of an odd phase: tile t + 2)
    }
    if (!grpB) __builtin_amdgcn_s_barrier();              // (group B's last phase starts behind this barrier)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // drain the tail requests before the block may exit

    const bool wide = (p.reserved_ & 1) != 0;
#pragma unroll
    for (int qg = 0; qg < QG; ++qg) {
        const float l = lt[qg][0];
        const float inv = l > 0.f ? 1.f / l : 0.f;
        const int q_row = q0 + qg * 16 + l16;
        f16* orow = O + ((long)b * p.Sq + (q_row < p.Sq ? q_row : p.Sq - 1)) * p.ldo + h * ATT_D;
        if (wide) {
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                unsigned u[2][2];
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const f32x4 a = ot[qg][2 * pr + k];
                    u[k][0] = __builtin_bit_cast(unsigned, (att_h2){(f16)(a[0] * inv), (f16)(a[1] * inv)});
                    u[k][1] = __builtin_bit_cast(unsigned, (att_h2){(f16)(a[2] * inv), (f16)(a[3] * inv)});
                }
                const auto r0_ = __builtin_amdgcn_permlane16_swap(u[0][0], u[1][0], false, false);
                const auto r1_ = __builtin_amdgcn_permlane16_swap(u[0][1], u[1][1], false, false);
                const int n = (2 * pr + 1) * 16 + 4 * g;
                const int nst = (g & 1) ? n - 4 : n - 16;
                typedef unsigned att_u4 __attribute__((ext_vector_type(4)));
                if (q_row < p.Sq) *reinterpret_cast<att_u4*>(orow + nst) = (att_u4){r0_[0], r1_[0], r0_[1], r1_[1]};
            }
            continue;
        }
        if (q_row < p.Sq) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const f16x4 o = {(f16)(ot[qg][dt][0] * inv), (f16)(ot[qg][dt][1] * inv), (f16)(ot[qg][dt][2] * inv),
                                 (f16)(ot[qg][dt][3] * inv)};
                *reinterpret_cast<f16x4*>(orow + dt * 16 + 4 * g) = o;
            }
        }
    }
}

template <int QG>
static void attn_launch_pp(const LbAttnParams& p, hipStream_t s) {
    constexpr size_t smem = (size_t)4 * 2 * 64 * ATT_D * sizeof(f16);       // 64 KiB (QG = 1: 126 VGPRs, two blocks per CU; QG = 2: one)
    const dim3 grid((unsigned)((p.Sq + 128 * QG - 1) / (128 * QG)) * p.H * p.B);
    hipLaunchKernelGGL((attn_fwd_d64_pp_kernel<QG>), grid, dim3(512), smem, s, p);
}

template <int QG>
static void attn_launch_stream(const LbAttnParams& p, hipStream_t s) {
    const size_t smem = (size_t)3 * 2 * 64 * ATT_D * sizeof(f16);
    const dim3 grid((unsigned)((p.Sq + 64 * QG - 1) / (64 * QG)) * p.H * p.B);
    hipLaunchKernelGGL((attn_fwd_d64_stream_kernel<QG>), grid, dim3(256), smem, s, p);
}

template <int KT, int QG, int NS>
static void attn_launch(const LbAttnParams& p, hipStream_t s) {
    const size_t smem = (size_t)NS * 2 * KT * ATT_D * sizeof(f16);
    if constexpr (NS * 2 * KT * ATT_D * sizeof(f16) > 64 * 1024) {      // (beyond the default dynamic-LDS limit: the 5-stage A/B form)
        static unsigned long long seen = 0;
        LB_ONCE_PER_DEVICE(seen)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_d64_kernel<KT, QG, NS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    }
    const dim3 grid((unsigned)((p.Sq + 64 * QG - 1) / (64 * QG)) * p.H * p.B);
    hipLaunchKernelGGL((attn_fwd_d64_kernel<KT, QG, NS>), grid, dim3(256), smem, s, p);
}

// variant (testing): 0 = by shape, else bit 0..1 QG (1 / 2), bit 4 forces the 64-key streaming tile, bit 5 = a 5-stage ring for the
// streaming form (80 KiB: a sequence of <= 256 keys is then requested whole in the prologue; A/B knob, not a default), bit 8 = the
// streaming kernel of rounds 1-5 (attn_fwd_d64_kernel<64, QG, 3>) instead of attn_fwd_d64_stream_kernel, bit 9 = the 8-wave ping-pong
// form (attn_fwd_d64_pp_kernel), bit 11 = the block order of rounds 1-5 instead of the XCD-aware one
static int g_attn_force = 0;
extern "C" void lb_attn_set_tuning(int force) { g_attn_force = force; }

static int attn_dispatch(const LbAttnParams& pin, int force, hipStream_t s) {
    LbAttnParams p = pin;
    p.reserved_ = (!(force & 128) && p.ldo % 8 == 0 && (reinterpret_cast<uintptr_t>(p.O) & 15) == 0) ? 1 : 0;    // 16-byte output stores
    if (force & 2048) p.reserved_ |= 2;      // block order of rounds 1-5 (query block fastest: the sharers of a K / V on different XCDs)
    // short sequences (cross-attention: 80 context rows) sit in ONE 96-key tile; long ones stream 64-key tiles
    const bool single = p.Skv <= 96 && !(force & 16);
    const long blocks128 = (long)((p.Sq + 127) / 128) * p.H * p.B;
    int qg = force & 3;
    if (qg == 0) qg = blocks128 >= 384 ? 2 : 1;   // 128-row blocks once they fill the chip
    // The one-tile form needs no ring: ONE stage (24 KiB of LDS, nothing but the tile itself requested - the two-stage form
    // spent a tile's worth of zero-page requests and LDS on a refill that never comes).  Beyond ~4 rounds of 128-row blocks the
    // 64-row blocks (81 VGPRs: five blocks per CU instead of three) quantise better.  MI355X, B = 17 (profiles/r04_attn_ab.txt):
    // cross S = 1024: 18.1 -> 15.2 us, cross S = 256: 11.0 -> 10.2 us.  Bit 6 of `force` = the former two-stage form (A/B).
    if (single && !(force & 3) && blocks128 >= 1024) qg = 1;
    if (single && !(force & 64)) { if (qg == 2) attn_launch<96, 2, 1>(p, s); else attn_launch<96, 1, 1>(p, s); }
    else if (single) { if (qg == 2) attn_launch<96, 2, 2>(p, s); else attn_launch<96, 1, 2>(p, s); }
    else if (force & 32) { if (qg == 2) attn_launch<64, 2, 5>(p, s); else attn_launch<64, 1, 5>(p, s); }
    else if (force & 256) { if (qg == 2) attn_launch<64, 2, 3>(p, s); else attn_launch<64, 1, 3>(p, s); }      // (rounds 1-5 streaming kernel: A/B)
    else if (force & 512) { if (qg == 2) attn_launch_pp<2>(p, s); else attn_launch_pp<1>(p, s); }      // (ping-pong form, forced)
    else        { if (qg == 2) attn_launch_stream<2>(p, s); else attn_launch_stream<1>(p, s); }
    return lb_check_launch("lb_attn_fwd_d64");
}

extern "C" int lb_attn_fwd_d64(const LbAttnParams* pp, void* stream) {
    const LbAttnParams p = *pp;
    LB_REQUIRE(p.B > 0 && p.H > 0 && p.Sq > 0 && p.Skv > 0, "lb_attn_fwd_d64: sizes");
    LB_REQUIRE(p.Skv_valid > 0 && p.Skv_valid <= p.Skv, "lb_attn_fwd_d64: 0 < Skv_valid <= Skv");
    LB_REQUIRE(p.ldq % 8 == 0 && p.ldk % 8 == 0 && p.ldv % 8 == 0 && p.ldo % 4 == 0, "lb_attn_fwd_d64: ld alignment");
    LB_REQUIRE(p.zero_page != nullptr, "lb_attn_fwd_d64: zero_page (>= 16 zero bytes) is required");
    LB_REQUIRE(!p.causal || p.Sq == p.Skv, "lb_attn_fwd_d64: causal attention needs Sq == Skv");
    const int force = g_attn_force;
    LB_DISPATCH("lb_attn_fwd_d64", attn_dispatch(p, force, s));
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
