// PROBE (round 4, not part of liblbhip.so): built as tile code 10 at commit "gemm_w4.hip: one-wave-per-SIMD ...", verified
// bit-identical to the shipped kernels on every shape, measured 0.94-1.05 PFLOP/s on the large squares and 0.24-0.86 on the
// B = 17 shapes (profiles/r04_gemm_bench_call3.txt: below the ping-pong kernel everywhere - its requests are only one K-tile
// deep, and 30 % of its wave cycles are waits for them), then taken out of the library.  Kept here for the technique: 256
// accumulators under fixed AGPR names behind inline-asm MFMAs (hipcc's own allocation of 256 loop-carried accumulators
// rotates them through VGPRs and 1 KiB of scratch per lane), audited by tools/probes/check_w4_asm.py.
// One-wave-per-SIMD MFMA GEMM for gfx950: 256 x 256 x 64 block tile, FOUR waves (2 x 2), each owning a 128 x 128 output tile
// (256 fp32 accumulator registers per lane - the whole accumulator half of the 512-entry register file), the K loop
// software-pipelined INSIDE every wave's instruction stream: while the 64 MFMAs of one 32-deep k-step execute, the same wave
// issues the 16 ds_read_b128 of the next k-step's fragments and (every other k-step) its 16 LDS-DMA requests of the K-tile
// after next.  Same contract and epilogues as gemm.hip / gemm_glds.hip / gemm_pp.hip (plain and GEGLU GEMMs; lb_gemm_f16
// routes here, tile code 10); bit-identical results (same K order per accumulator).
//
// Why.  profiles/r04_gemm_bench_call2.txt + r04_pmc_sq.json: the 8-wave tiles keep the matrix pipe 45-50 % busy in lock-step
// (gemm_glds.hip) and 59 % busy as two staggered groups (gemm_pp.hip: a wave's load segment - 12 LDS reads plus two LDS-DMA
// issues of ~100 cycles each - is longer than its partner's 256-cycle compute segment, and every phase ends in a
// workgroup-wide rendezvous), while the vendor library reaches 1.39-1.46 PFLOP/s on the same operands.  With ONE wave per
// SIMD nothing has to be handed over: a 128 x 128 wave tile needs 0.25 ds_read_b128 and 0.125 LDS-DMA issues per MFMA
// (64 x 128: 0.375 / 0.125 per wave but twice the waves), all of which fit into the 12 free issue cycles behind every
// 16-cycle MFMA, and the only synchronisation is ONE barrier per K-tile.
//
// Ring: two 64 KiB stages (A rows 0..255 then W rows 0..255, [rows][64] halves, 16-B chunks XOR-swizzled by row & 7 on the
// per-lane SOURCE address as in the other kernels).  Per K-tile t, per wave:
//     block 0:  64 MFMAs on F0 = (t, k 0..31)      | 16 ds_read_b128 -> F1 = (t, k 32..63)
//     s_waitcnt lgkmcnt(0) vmcnt(0) ; s_barrier    | all reads of stage t returned; K-tile t+1 landed everywhere
//     block 1:  64 MFMAs on F1                     | 16 ds_read_b128 -> F0 = (t+1, k 0..31) ; 16 LDS-DMA requests of K-tile t+2
// The requests of K-tile t+2 have the whole of block 0 of K-tile t+1 (>= 1024 cycles) to land before they are waited for,
// and the wait (`vmcnt(0)`) never covers a request issued less than one block earlier.
//
// Replaces (third party, reached from /root/reference/latentblending/diffusers_holder.py:336): the torch.nn.Linear layers
// of the SDXL UNet's transformer blocks at batch >= 8.
#include "lb_common.h"
#include "lb_gemm.h"

typedef __attribute__((address_space(3))) void* lptr_t;

#define W4_BK 64
#define W4_STAGE_H 32768         // halves per ring stage: (256 + 256) rows x 64 halves = 64 KiB
#define W4_LDS_BYTES (2 * W4_STAGE_H * 2)

template <int V> struct W4Int { static constexpr int value = V; };
typedef W4Int<0> C0;
typedef W4Int<1> C1;

// The 64 accumulator tiles live in a[0:255] under fixed names: acc tile (i, j) = a[4 (8 i + j) .. + 3].  hipcc's own allocation
// of 256 loop-carried MFMA accumulators rotates them through VGPRs and scratch (every MFMA gets a different destination and
// a v_accvgpr copy chain; 1 KiB of scratch per lane), so the MFMAs are inline asm on literal accumulator registers and the
// compiler never sees an accumulator value: one statement at kernel entry clobbers a0..a255 (which makes the kernel descriptor
// allocate the accumulator file), the epilogue reads the tiles back eight at a time.  Audit after every edit of this file
// (tools/probes/check_w4_asm.py): no compiler-generated v_accvgpr_* and no scratch access anywhere in the kernel.
template <int IDX> __device__ __forceinline__ void w4_mfma(const f16x8& b, const f16x8& a) {
    asm volatile("v_mfma_f32_16x16x32_f16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(b), "v"(a), "n"(4 * IDX), "n"(4 * IDX + 3) : "memory");
}
template <int IDX> __device__ __forceinline__ void w4_acc_zero() {
    asm volatile("v_accvgpr_write_b32 a[%c0], 0\n\tv_accvgpr_write_b32 a[%c1], 0\n\tv_accvgpr_write_b32 a[%c2], 0\n\tv_accvgpr_write_b32 a[%c3], 0" ::"n"(4 * IDX),
                 "n"(4 * IDX + 1), "n"(4 * IDX + 2), "n"(4 * IDX + 3));
}
template <int IDX> __device__ __forceinline__ f32x4 w4_acc_read() {
    f32x4 v;
    asm volatile("v_accvgpr_read_b32 %0, a[%c4]\n\tv_accvgpr_read_b32 %1, a[%c5]\n\tv_accvgpr_read_b32 %2, a[%c6]\n\tv_accvgpr_read_b32 %3, a[%c7]"
                 : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3])
                 : "n"(4 * IDX), "n"(4 * IDX + 1), "n"(4 * IDX + 2), "n"(4 * IDX + 3));
    return v;
}
#define W4_UNROLL16(F) F(W4Int<0>{}); F(W4Int<1>{}); F(W4Int<2>{}); F(W4Int<3>{}); F(W4Int<4>{}); F(W4Int<5>{}); F(W4Int<6>{}); F(W4Int<7>{}); \
                       F(W4Int<8>{}); F(W4Int<9>{}); F(W4Int<10>{}); F(W4Int<11>{}); F(W4Int<12>{}); F(W4Int<13>{}); F(W4Int<14>{}); F(W4Int<15>{})
#define W4_UNROLL8(F) F(W4Int<0>{}); F(W4Int<1>{}); F(W4Int<2>{}); F(W4Int<3>{}); F(W4Int<4>{}); F(W4Int<5>{}); F(W4Int<6>{}); F(W4Int<7>{})

template <bool GEGLU>
__global__ void __launch_bounds__(256) gemm_f16_w4_kernel(const LbGemmParams p) {
    constexpr int BM = 256, BN = 256;
    extern __shared__ __attribute__((aligned(16))) f16 lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int g = lane >> 4, l16 = lane & 15;

    // ---- block -> tile (XCD-aware bijective remap, then the operand with more bytes is the shared one) ----
    const int n_eff = GEGLU ? p.N / 2 : p.N;
    constexpr int BN_OUT = GEGLU ? BN / 2 : BN;
    const int n_blocks = (n_eff + BN_OUT - 1) / BN_OUT;
    const int m_blocks = (p.M + BM - 1) / BM;
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x;
        const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const bool w_dominant = n_eff > p.M;
    const int block_n = w_dominant ? bid / m_blocks : bid % n_blocks;
    const int block_m = w_dominant ? bid % m_blocks : bid / n_blocks;
    const int m0 = block_m * BM;
    const int n0 = block_n * BN_OUT;

    const int k_tiles_total = p.K / W4_BK;                       // (launcher: K % 64 == 0)
    const int tiles_per_split = (k_tiles_total + p.splitk - 1) / p.splitk;
    const int kt_begin = blockIdx.z * tiles_per_split;
    int kt_end = kt_begin + tiles_per_split;
    if (kt_end > k_tiles_total) kt_end = k_tiles_total;
    const int nkt = kt_end - kt_begin;

    // ---- staging: a K-tile = 512 LDS rows of 128 B = 64 wave instructions = 16 per wave (rows n*32 + wave*8 + lr) ----
    // `buffer_load_dwordx4 ... offen lds`: per-lane state is one 32-bit byte offset per staged row, the K position is the
    // scalar offset of the instruction.  No masking: rows beyond M / N re-read the last valid row (their accumulators are
    // never stored), requests beyond the K range re-read the last K-tile (into a stage nobody reads again).
    const int lr = lane >> 3;
    const int cl = (lane & 7) ^ (lr & 7);
    unsigned a_off[8], w_off[8];
#pragma unroll
    for (int n = 0; n < 8; ++n) {
        int m = m0 + n * 32 + wave * 8 + lr;
        m = m < p.M ? m : p.M - 1;
        a_off[n] = (unsigned)(((long)m * p.lda + cl * 8) * 2);
        const int r = n * 32 + wave * 8 + lr;                    // LDS row of the W half: wave column r >> 7, fragment (r >> 4) & 7, row r & 15
        long wrow;
        if (GEGLU) {                 // fragments (2 jp, 2 jp + 1) = (h, gate) of output columns n0 + ((r >> 7) * 4 + jp) * 16 + 0..15
            int oc = n0 + ((r >> 7) * 4 + ((r >> 5) & 3)) * 16 + (r & 15);
            oc = oc < n_eff ? oc : n_eff - 1;
            wrow = (long)((r >> 4) & 1) * n_eff + oc;
        } else {
            const int col = n0 + r;
            wrow = col < p.N ? col : p.N - 1;
        }
        w_off[n] = (unsigned)((wrow * p.ldw + cl * 8) * 2);
    }
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<lb_half*>(p.A), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<lb_half*>(p.W), 0, 0x7fffffff, 0x00020000);

    // request instruction n (0..7: A rows, 8..15: W rows) of K-tile `tile` into ring stage tile & 1
    auto stage_one = [&](int n, int tile) {
        const int kt = tile < nkt ? tile : nkt - 1;
        const int koff = (kt_begin + kt) * (W4_BK * 2);              // scalar byte offset of the K-tile
        f16* base = lds + (tile & 1) * W4_STAGE_H + (n * 32 + wave * 8) * W4_BK;     // (n >= 8: the W half follows the 256 A rows)
        if (n < 8) __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (lptr_t)base, 16, a_off[n & 7], koff, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lptr_t)base, 16, w_off[n & 7], koff, 0, 0);
    };

    // ---- fragment reads: lane (g, l16) reads row (16 i + l16) at 16-B chunk (4 ks + g) ^ (row & 7) ----
    const f16* a_rd[2];
    const f16* b_rd[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int ch = ((ks * 4 + g) ^ (l16 & 7)) << 3;
        a_rd[ks] = lds + (wr * 128 + l16) * W4_BK + ch;
        b_rd[ks] = lds + (256 + wc * 128 + l16) * W4_BK + ch;
    }
    f16x8 fa[2][8], fb[2][8];        // [k-step][fragment]
    asm volatile("" ::: "a0", "a255");       // (reserves the accumulator file in the kernel descriptor; every a[..] below is ours)
    {
        auto zero4 = [&](auto qc) {
            constexpr int Q = decltype(qc)::value;
            w4_acc_zero<4 * Q>(); w4_acc_zero<4 * Q + 1>(); w4_acc_zero<4 * Q + 2>(); w4_acc_zero<4 * Q + 3>();
        };
        W4_UNROLL16(zero4);
    }

    // One k-step: 64 MFMAs on fragment set CUR; the 16 fragment reads of the NEXT k-step (from the stage at `rd_off`
    // halves) and, when STAGE, the 16 requests of K-tile `st_tile` are spread over them in SOURCE order - the MFMA
    // statements are volatile and clobber memory, so nothing moves across them: 16 groups of 4 MFMAs, the first eight followed
    // by 2 LDS reads each, every group of a STAGE block by 1 request.
    auto block = [&](auto curc, auto stc, int rd_off, int st_tile) {
        constexpr int CUR = decltype(curc)::value, NXT = CUR ^ 1;
        constexpr bool STAGE = decltype(stc)::value != 0;
        auto group = [&](auto qc) {
            constexpr int Q = decltype(qc)::value, I = Q >> 1, J0 = (Q & 1) * 4;
            w4_mfma<I * 8 + J0 + 0>(fb[CUR][J0 + 0], fa[CUR][I]);
            w4_mfma<I * 8 + J0 + 1>(fb[CUR][J0 + 1], fa[CUR][I]);
            w4_mfma<I * 8 + J0 + 2>(fb[CUR][J0 + 2], fa[CUR][I]);
            w4_mfma<I * 8 + J0 + 3>(fb[CUR][J0 + 3], fa[CUR][I]);
            if constexpr (Q < 8) {       // (all 16 reads in the first half of the k-step: none is younger than 32 MFMAs when it is waited for)
                fb[NXT][Q] = *reinterpret_cast<const f16x8*>(b_rd[NXT] + rd_off + Q * 16 * W4_BK);
                fa[NXT][Q] = *reinterpret_cast<const f16x8*>(a_rd[NXT] + rd_off + Q * 16 * W4_BK);
            }
            if constexpr (STAGE) stage_one(Q, st_tile);
        };
        W4_UNROLL16(group);
    };

    // ---- prologue: K-tiles 0 and 1 requested, K-tile 0 landed, F0 = (0, k 0..31) ----
#pragma unroll
    for (int n = 0; n < 16; ++n) stage_one(n, 0);
#pragma unroll
    for (int n = 0; n < 16; ++n) stage_one(n, 1);
    asm volatile("s_waitcnt vmcnt(16)\n\ts_barrier" ::: "memory");
#pragma unroll
    for (int j = 0; j < 8; ++j) fb[0][j] = *reinterpret_cast<const f16x8*>(b_rd[0] + j * 16 * W4_BK);
#pragma unroll
    for (int i = 0; i < 8; ++i) fa[0][i] = *reinterpret_cast<const f16x8*>(a_rd[0] + i * 16 * W4_BK);

    for (int t = 0; t < nkt; ++t) {
        const int cur_off = (t & 1) * W4_STAGE_H, nxt_off = cur_off ^ W4_STAGE_H;
        block(C0{}, C0{}, cur_off, 0);                               // MFMAs (t, k 0..31) | reads (t, k 32..63)
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        block(C1{}, C1{}, nxt_off, t + 2);                           // MFMAs (t, k 32..63) | reads (t+1, k 0..31) | requests K-tile t+2
        __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // the tail requests (re-reads of the last K-tile) drained

    // ---- epilogue (shared with the other GEMM kernels), one 16-row fragment row of the wave tile at a time ----
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");               // (the last MFMAs' results are in the accumulator file before it is read)
    auto finish_row = [&](auto ic) {
        constexpr int I = decltype(ic)::value;
        f32x4 acc[1][8];
        acc[0][0] = w4_acc_read<I * 8 + 0>(); acc[0][1] = w4_acc_read<I * 8 + 1>(); acc[0][2] = w4_acc_read<I * 8 + 2>();
        acc[0][3] = w4_acc_read<I * 8 + 3>(); acc[0][4] = w4_acc_read<I * 8 + 4>(); acc[0][5] = w4_acc_read<I * 8 + 5>();
        acc[0][6] = w4_acc_read<I * 8 + 6>(); acc[0][7] = w4_acc_read<I * 8 + 7>();
        const int m = m0 + wr * 128 + I * 16 + l16;
        if (p.splitk > 1) {
            float* slab = p.partial + (long)blockIdx.z * p.M * p.N;
            if (m < p.M) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int n = n0 + wc * 128 + j * 16 + 4 * g;
                    if (n < p.N) *reinterpret_cast<f32x4*>(slab + (long)m * p.N + n) = acc[0][j];
                }
            }
        } else {
            lb_gemm_tile_epilogue<1, 8, GEGLU>(p, acc, m, n0 + wc * 128 + 4 * g, n0 + wc * 64 + 4 * g);
        }
    };
    W4_UNROLL8(finish_row);
}

int lb_gemm_w4_eligible(const LbGemmParams& p) {
    return !p.conv && p.K % W4_BK == 0 && !(p.flags & (LB_GEMM_LN_A | LB_GEMM_CH_STATS)) && p.lda % 8 == 0 && p.ldw % 8 == 0 &&
           (long)p.M * p.lda * 2 < (1l << 31) && (long)p.N * p.ldw * 2 < (1l << 31);       // 32-bit buffer offsets
}

template <bool GEGLU>
static void w4_launch(const LbGemmParams& p, dim3 grid, hipStream_t stream) {
    static unsigned long long seen = 0;
    if (lb_first_call_on_device(seen))
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f16_w4_kernel<GEGLU>), hipFuncAttributeMaxDynamicSharedMemorySize, W4_LDS_BYTES);
    hipLaunchKernelGGL((gemm_f16_w4_kernel<GEGLU>), grid, dim3(256), W4_LDS_BYTES, stream, p);
}

int lb_gemm_launch_w4(const LbGemmParams& p, dim3 grid, hipStream_t stream) {
    if (p.flags & LB_GEMM_GEGLU) w4_launch<true>(p, grid, stream);
    else w4_launch<false>(p, grid, stream);
    return 0;
}
