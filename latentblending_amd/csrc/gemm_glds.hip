// Direct-to-LDS variant of the MFMA GEMM / implicit-GEMM conv (same contract as gemm.hip).
//
// Staging uses `global_load_lds_dwordx4`: each wave instruction moves 64 x 16 B straight from
// global memory into 1 KiB of LDS (no VGPR round trip, no ds_write pass - the register->LDS path is
// the narrowest pipe of the CU).  The LDS image of a tile is the same XOR-swizzled [rows][64] fp16
// layout the fragment reads expect; because the LDS destination of the instruction is lane-linear,
// the swizzle is applied to the per-lane GLOBAL source address instead (lane (r, s) fetches logical
// chunk s ^ (r & 7) of row r - the 8 lanes of a row still cover one 128-B line).  Out-of-range
// chunks (conv padding, M / N / K tails) fetch from a 16-byte zero page.
//
// Pipeline: S-stage LDS ring, tile t+S-1 in flight while tile t is multiplied.  Per K-tile:
//   s_waitcnt vmcnt((S-2) * loads_per_tile)   <- this thread's part of tile t has landed
//   s_barrier                                 <- everybody's part has landed AND everybody finished
//                                                reading stage (t-1) % S, which is refilled next
//   issue tile t+S-1 ; ds_read + MFMA on stage t % S
// i.e. ONE barrier per K-tile and never a vmcnt(0) in the steady state.
#include <type_traits>
#include "lb_common.h"
#include "lb_gemm.h"

#define BK 64
#ifndef LB_GLDS_LEAN        // round 6: lean request addressing (see the kernel); 0 = the forms of rounds 1-5 (A/B builds)
#define LB_GLDS_LEAN 1
#endif
// Ablation builds for tools/gemm_ablate.py ONLY (never the shipped library): 1 = no global->LDS
// requests, 2 = no LDS reads / MFMAs, 3 = MFMAs on register operands (no LDS reads).
#ifndef LB_ABLATE
#define LB_ABLATE 0
#endif
#ifndef LB_BURST            // 1 = the older loop body: all requests right after the barrier (A/B studies)
#define LB_BURST 0
#endif
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int N> __device__ __forceinline__ void wait_vm_barrier() {
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}

// Row statistics for a LayerNorm fused into the A operand (LB_GEMM_LN_A): the lane's fragment of K-half s holds 8
// consecutive k-values of row (16 i + l16) for every i < TM; over the whole K loop the four lanes g = 0..3 of a row
// see every column exactly once.  Sums run on the (idle) VALU as v_dot2_f32_f16: x.x and x.1 of two halves per
// instruction, fp32 accumulation.
typedef _Float16 lb_h2 __attribute__((ext_vector_type(2)));
template <int TM>
__device__ __forceinline__ void ln_accumulate(const f16x8 (&af)[TM], float (&sum)[TM], float (&sq)[TM]) {
    const lb_h2 one2 = {(f16)1.f, (f16)1.f};
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const lb_h2 x = {af[i][2 * e], af[i][2 * e + 1]};
            sum[i] = __builtin_amdgcn_fdot2(x, one2, sum[i], false);
            sq[i] = __builtin_amdgcn_fdot2(x, x, sq[i], false);
        }
}

// WMW = waves along M (2 -> 4 waves / 256 threads, 3 -> 6 waves / 384 threads, 4 -> 8 waves / 512 threads); always 2
// waves along N.  The 6-wave 192x128 tile exists for wave quantisation: M = 4352 (17 samples x 256 tokens) makes
// 170 tiles of 256x128 (2/3 of the chip, one round) but 230 tiles of 192x128 at 3/4 of the work each.  Its weight
// tile (128 rows) is not a multiple of the 48 rows one round of its 384 threads stages: the third round is half
// masked (rows >= BN fetch the zero page into 16 padding rows of the stage).
// LNA: 0 none; 1 LayerNorm of A folded in, row statistics accumulated from the A fragments inside the K loop (used by the
// launch-bound B = 2 programs).  (A second form - statistics written by the PRODUCING GEMM's epilogue and only applied here -
// was built in round 2, measured slower than the stand-alone LayerNorm at every batch size and removed in round 3:
// profiles/r02_ln_stats_ab.txt, git history.)
// KG (round 6): K-groups per block.  KG = 2: the block has TWO sets of 2 WMW waves; set g multiplies the K-tiles g, g + 2, ... of the
// block's output tile out of its own LDS ring, and the two partial accumulators are added through LDS before the epilogue.  For the
// launch-bound programs (B <= 4: M = 512 rows -> 160 blocks of 64x64 on 256 CUs, ONE wave per SIMD): a lone wave runs its K-tile as one
// dependent chain - wait, barrier, fragment reads, 8 MFMAs, four requests - ~950 cycles for 128 cycles of matrix work, and nothing else
// is resident to fill the gaps; a second wave per SIMD working on the other half of K doubles what is in flight per CU without
// another launch, another slab round trip or more blocks.  Not bit-identical to KG = 1 (two partial sums per output instead of one chain).
template <int BM, int BN, bool CONV, bool GEGLU, int S, int WMW, int LNA = 0, int KG = 1>
__global__ void __launch_bounds__(WMW * 128 * KG) gemm_f16_glds_kernel(const LbGemmParams p) {
    static_assert(KG == 1 || !CONV, "K-groups: plain / GEGLU GEMMs only");
    constexpr int NT = WMW * 128;           // threads per K-group (= per block when KG == 1)
    constexpr int RPI = NT / 8;             // tile rows covered by one round of wave instructions
    constexpr int AI = BM * 8 / NT;         // wave instructions (16-B chunks per thread) of A per K-tile
    constexpr int WI = (BN * 8 + NT - 1) / NT;
    constexpr int NL = AI + WI;             // VMEM loads per thread per K-tile
    constexpr int WROWS = BM / WMW;         // rows of the block tile owned by one wave
    constexpr int WPAD = WI * RPI;          // weight rows staged per K-tile (>= BN)
    static_assert(BM * 8 % NT == 0, "A tile rows must be a whole number of staging rounds");
    constexpr int TM = WROWS / 16, TN = BN / 32;
    constexpr int STAGE = (BM + WPAD) * BK; // halves per ring stage
    extern __shared__ __attribute__((aligned(16))) f16 lds_all[];

    const int kgrp = KG > 1 ? __builtin_amdgcn_readfirstlane((int)threadIdx.x / NT) : 0;      // (wave-uniform)
    const int tid = KG > 1 ? (int)threadIdx.x - kgrp * NT : (int)threadIdx.x;
    f16* const lds = lds_all + kgrp * (S * STAGE);
    const int lane = tid & 63;
    // (wave-uniform by construction: through readfirstlane, so that the LDS destination of every direct-to-LDS request (M0) is scalar
    //  arithmetic instead of a VALU chain + v_readfirstlane per request - round 6, LB_GLDS_LEAN)
    const int wave = LB_GLDS_LEAN ? __builtin_amdgcn_readfirstlane(tid >> 6) : tid >> 6;
    const int wave_m = wave >> 1, wave_n = wave & 1;
    const int g = lane >> 4, l16 = lane & 15;

    const int n_blocks = GEGLU ? (p.N / 2 + BN / 2 - 1) / (BN / 2) : (p.N + BN - 1) / BN;
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x;
        const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int m_blocks = (p.M + BM - 1) / BM;
    const bool w_dominant = (GEGLU ? p.N / 2 : p.N) > p.M;
    const int block_n = w_dominant ? bid / m_blocks : bid % n_blocks;
    const int block_m = w_dominant ? bid % m_blocks : bid / n_blocks;
    const int m0 = block_m * BM;
    const int n0 = GEGLU ? block_n * (BN / 2) : block_n * BN;

    const int k_tiles_total = (p.K + BK - 1) / BK;
    const int tiles_per_split = (k_tiles_total + p.splitk - 1) / p.splitk;
    const int kt_begin = blockIdx.z * tiles_per_split;
    int kt_end = kt_begin + tiles_per_split;
    if (kt_end > k_tiles_total) kt_end = k_tiles_total;
    const int nkt = kt_end - kt_begin;
    const int k_end = kt_end * BK < p.K ? kt_end * BK : p.K;

    // thread (r = tid>>3 (+32 i), s = tid&7) owns physical chunk s of tile row r and fetches logical
    // chunk s ^ (r & 7); (r & 7) == (lane >> 3) for every i because row groups start at multiples of 8
    const int row0 = tid >> 3;
    const int cl = (tid & 7) ^ ((tid >> 3) & 7);     // logical chunk fetched by this lane

    long a_off[AI];
    int a_iy[AI], a_ix[AI];
    bool a_ok[AI];
    int ci = 0, ky = 0, kx = 0;
    int kcur = (kt_begin + kgrp) * BK + cl * 8;
#pragma unroll
    for (int i = 0; i < AI; ++i) {
        const int m = m0 + row0 + i * RPI;
        a_ok[i] = m < p.M;
        if (CONV) {
            const int hw = p.Hout * p.Wout;
            const int b = m / hw, rem = m - b * hw;
            const int oy = rem / p.Wout, ox = rem - oy * p.Wout;
            a_off[i] = (long)b * p.Hin * p.Win * p.ldx;
            a_iy[i] = oy * p.stride - (p.scatter ? 1 - p.sc_py : p.pad);
            a_ix[i] = ox * p.stride - (p.scatter ? 1 - p.sc_px : p.pad);
        } else {
            a_off[i] = (long)m * p.lda;
            a_iy[i] = a_ix[i] = 0;
        }
    }
    if (CONV) {
        const int tap = kcur / p.Cin;
        ci = kcur - tap * p.Cin;
        ky = tap / p.KW;
        kx = tap - ky * p.KW;
    }
    long w_off[WI];
    bool w_ok[WI];
#pragma unroll
    for (int i = 0; i < WI; ++i) {
        const int tr = row0 + i * RPI;
        int n;
        if (GEGLU) {
            const int sub = tr >> 4;
            n = (sub & 1) * (p.N / 2) + n0 + (sub >> 1) * 16 + (tr & 15);
            w_ok[i] = tr < BN && (n0 + (sub >> 1) * 16 + (tr & 15)) < p.N / 2;
        } else {
            n = n0 + tr;
            w_ok[i] = tr < BN && n < p.N;
        }
        w_off[i] = w_ok[i] ? (long)n * p.ldw : 0;
    }
    const lb_half* zero = reinterpret_cast<const lb_half*>(p.zero_page);
    const int hin_eff = p.Hin << p.ups, win_eff = p.Win << p.ups;

    // ---- round 6: LEAN requests for plain / GEGLU GEMMs whose K is a whole number of K-tiles (every Linear of the UNet / CLIP programs) ----
    // A request used to cost a 64-bit address, two selects against the zero page and a VALU chain for M0: 75 instructions around the 8 MFMAs
    // of a 64x64 K-tile, 110 around the 24 of the 192x128 one - and the launch-bound B <= 4 GEMMs run their K-tiles as ONE dependent chain per
    // wave.  Lean form (gemm_pp.hip's): a 32-bit BYTE offset per lane and request on a wave-uniform base that carries the K position (scalar),
    // no masks - rows past M / N re-read the last row (their accumulator rows / columns are never stored: every output element has its own
    // accumulator), tiles requested past the block's K range re-read its last K-tile (into ring stages nobody reads again).  Not when a
    // K-group would have to MULTIPLY a tile past the end (KG > 1 with an odd tile count: those must be zero tiles), not for convolutions
    // (padding must read zeros), not for a ragged K.  Same bytes staged for everything that is stored: bit-identical.
    constexpr bool LEAN_OK = LB_GLDS_LEAN && !CONV && !(BM == 256 && BN == 256);
    const bool lean = LEAN_OK && p.K % BK == 0 && nkt >= 1 && (KG == 1 || nkt % KG == 0) &&
                      (long)p.M * p.lda * 2 < (1l << 32) && (long)p.N * p.ldw * 2 < (1l << 32);
    unsigned a_off32[AI], w_off32[WI];
    if constexpr (LEAN_OK) {
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            int m = m0 + row0 + i * RPI;
            m = m < p.M ? m : p.M - 1;
            a_off32[i] = (unsigned)(((long)m * p.lda + cl * 8) * 2);
        }
#pragma unroll
        for (int i = 0; i < WI; ++i) {
            const int tr = row0 + i * RPI;
            int n;
            if (GEGLU) {
                const int sub = tr >> 4, half = p.N / 2;
                int oc = n0 + (sub >> 1) * 16 + (tr & 15);
                oc = oc < half ? oc : half - 1;
                n = (sub & 1) * half + oc;
            } else {
                n = n0 + tr;
            }
            n = n < p.N ? n : p.N - 1;              // (also the staging rows past BN of a partial round: padding rows of the stage)
            w_off32[i] = (unsigned)(((long)n * p.ldw + cl * 8) * 2);
        }
    }
    // K-tile of this K-group the NEXT issue_tile stages, and the last one that exists (scalar)
    int lean_kt = kt_begin + kgrp;
    const int lean_kt_last = kt_begin + kgrp + ((nkt - 1 - kgrp) / KG) * KG;

    // request number idx (0..NL-1: AI rows of A, then WI rows of W) of the current K-tile into the ring
    // stage at `base` (branch-free: masked chunks read the zero page)
    auto issue_one = [&](auto leanc, int idx, f16* base, bool k_ok) {
        const lb_half* src;
        f16* dst;
        if constexpr (decltype(leanc)::value) {
            const long kbyte = (long)(lean_kt < lean_kt_last ? lean_kt : lean_kt_last) * (BK * 2);        // scalar
            if (idx < AI) {
                src = reinterpret_cast<const lb_half*>(reinterpret_cast<const char*>(p.A) + kbyte + (size_t)a_off32[idx < AI ? idx : 0]);
                dst = base + (wave * 8 + idx * RPI) * BK;
            } else {
                const int i = idx - AI;
                src = reinterpret_cast<const lb_half*>(reinterpret_cast<const char*>(p.W) + kbyte + (size_t)w_off32[i < WI && i >= 0 ? i : 0]);
                dst = base + BM * BK + (wave * 8 + i * RPI) * BK;
            }
            if (LB_ABLATE != 1) __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 0, 0);
            return;
        }
        if (idx < AI) {
            const int i = idx;
            if (CONV) {
                const int iy = a_iy[i] + ky, ix = a_ix[i] + kx;
                const bool ok = a_ok[i] && k_ok && iy >= 0 && iy < hin_eff && ix >= 0 && ix < win_eff;
                src = ok ? p.A + a_off[i] + ((long)(iy >> p.ups) * p.Win + (ix >> p.ups)) * p.ldx + ci : zero;
            } else {
                src = (a_ok[i] && k_ok) ? p.A + a_off[i] + kcur : zero;
            }
            // wave-uniform LDS base of this instruction's 8 rows; hardware adds lane * 16 B
            dst = base + (wave * 8 + i * RPI) * BK;
        } else {
            const int i = idx - AI;
            src = (w_ok[i] && k_ok) ? p.W + w_off[i] + kcur : zero;
            dst = base + BM * BK + (wave * 8 + i * RPI) * BK;
        }
        if (LB_ABLATE != 1) __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 0, 0);
    };
    // advance this thread's K position (and conv tap state) by one K-tile
    auto advance_k = [&]() {
        kcur += BK * KG;
        lean_kt += KG;
        if (CONV) {
            ci += BK;
            if (p.Cin >= BK) {
                const bool wrap = ci >= p.Cin;
                ci -= wrap ? p.Cin : 0;
                kx += wrap ? 1 : 0;
                const bool wrapx = kx == p.KW;
                kx = wrapx ? 0 : kx;
                ky += wrapx ? 1 : 0;
            } else {
                while (ci >= p.Cin) {
                    ci -= p.Cin;
                    if (++kx == p.KW) { kx = 0; ++ky; }
                }
            }
        }
    };
    auto issue_tile = [&](auto leanc, int st) {
        f16* base = lds + st * STAGE;
        const bool k_ok = kcur < k_end;
#pragma unroll
        for (int idx = 0; idx < NL; ++idx) issue_one(leanc, idx, base, k_ok);
        advance_k();
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float ln_sum[TM], ln_sq[TM];            // LNA: running sum / sum of squares of this lane's k-slice of its rows
#pragma unroll
    for (int i = 0; i < TM; ++i) ln_sum[i] = ln_sq[i] = 0.f;

    // operand fragments of K-half s of the tile in stage st
    auto read_frags = [&](int st, int s, f16x8 (&af)[TM], f16x8 (&wf)[TN]) {
        const f16* Ab = lds + st * STAGE + (wave_m * WROWS) * BK;
        const f16* Wb = lds + st * STAGE + BM * BK + (wave_n * (BN / 2)) * BK;
        const int chunk = s * 4 + g;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int r = i * 16 + l16;
            if (LB_ABLATE == 3) { af[i] = (f16x8){(f16)lane, 1, 2, 3, 4, 5, 6, (f16)st}; continue; }
            af[i] = *reinterpret_cast<const f16x8*>(Ab + r * BK + ((chunk ^ (r & 7)) << 3));
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int r = j * 16 + l16;
            if (LB_ABLATE == 3) { wf[j] = (f16x8){(f16)st, 1, 2, 3, 4, 5, 6, (f16)lane}; continue; }
            wf[j] = *reinterpret_cast<const f16x8*>(Wb + r * BK + ((chunk ^ (r & 7)) << 3));
        }
    };
    auto mma_rows = [&](const f16x8 (&af)[TM], const f16x8 (&wf)[TN], int i_lo, int i_hi) {
#pragma unroll
        for (int i = i_lo; i < i_hi; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[j], af[i], acc[i][j], 0, 0, 0);
    };

    // the prologue and the K loop, once per request form: `lean` is a block-uniform run-time fact, the two forms are separate loops (as
    // one loop with the form tested per request hipcc computes BOTH addresses and selects)
    auto main_loop = [&](auto leanc) {
    // ---- prologue: tiles 0 .. S-2 in flight ----
#pragma unroll
    for (int s = 0; s < S - 1; ++s) issue_tile(leanc, s);

    // One K-tile per iteration.  The NL requests of tile t+S-1 are NOT issued in one burst after the
    // barrier (all waves of a block would then sit in the VMEM issue queue together and start their
    // MFMAs together): they are spread over the four MFMA groups of tile t, so a wave blocked on a
    // full memory pipeline has MFMAs of its own still executing.  sched_barrier pins the order.
    // (tools/gemm_ablate.py: 5-13 % on the plain / GEGLU contractions, first > 1 PFLOP/s at 8192^3.)
    constexpr int HM = TM > 1 ? TM / 2 : 1;
    int st = 0;                               // ring stage of tile t
    const int nloop = KG > 1 ? (nkt + KG - 1) / KG : nkt;      // (every K-group runs the same number of barriers; tiles past k_end are zero tiles)
    for (int t = 0; t < nloop; ++t) {
        wait_vm_barrier<(S - 2) * NL>();      // tile t landed everywhere; stage (t-1)%S free everywhere
        int refill = st - 1;
        if (refill < 0) refill += S;
        if (LB_ABLATE == 2) {
            issue_tile(leanc, refill);
        } else if (LB_BURST || (CONV && WMW == 2)) {   // 4-wave conv tiles measured 3-6 % faster with the burst
            issue_tile(leanc, refill);               // tile t+S-1 (masked to zeros past the end)
            f16x8 af[TM], wf[TN];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                read_frags(st, s, af, wf);
                if (LNA == 1) ln_accumulate<TM>(af, ln_sum, ln_sq);
                mma_rows(af, wf, 0, TM);
            }
        } else {
            f16* base = lds + refill * STAGE;
            const bool k_ok = kcur < k_end;
            f16x8 a0[TM], w0[TN], a1[TM], w1[TN];
            read_frags(st, 0, a0, w0);
#pragma unroll
            for (int idx = 0; idx < NL; ++idx)
                if (idx * 4 / NL == 0) issue_one(leanc, idx, base, k_ok);
            __builtin_amdgcn_sched_barrier(0);
            mma_rows(a0, w0, 0, HM);
            read_frags(st, 1, a1, w1);
            if (LNA == 1) ln_accumulate<TM>(a0, ln_sum, ln_sq);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int idx = 0; idx < NL; ++idx)
                if (idx * 4 / NL == 1) issue_one(leanc, idx, base, k_ok);
            __builtin_amdgcn_sched_barrier(0);
            mma_rows(a0, w0, HM, TM);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int idx = 0; idx < NL; ++idx)
                if (idx * 4 / NL == 2) issue_one(leanc, idx, base, k_ok);
            __builtin_amdgcn_sched_barrier(0);
            mma_rows(a1, w1, 0, HM);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int idx = 0; idx < NL; ++idx)
                if (idx * 4 / NL == 3) issue_one(leanc, idx, base, k_ok);
            advance_k();
            if (LNA == 1) ln_accumulate<TM>(a1, ln_sum, ln_sq);
            __builtin_amdgcn_sched_barrier(0);
            mma_rows(a1, w1, HM, TM);
        }
        st = st + 1 == S ? 0 : st + 1;
    }
    };      // main_loop
    if constexpr (LEAN_OK) {
        if (lean) main_loop(std::true_type{});
        else main_loop(std::false_type{});
    } else {
        main_loop(std::false_type{});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the masked tail requests before exit/epilogue

    if constexpr (KG > 1) {
        // partial accumulators (and LayerNorm row sums) of K-groups 1.. -> LDS -> added by K-group 0 in group order, which alone
        // runs the epilogue.  Slot layout [wave][i][j][lane] x 16 B: every lane re-reads exactly the slot its twin wrote.
        __syncthreads();                                            // every group is done with its ring
        float* red = reinterpret_cast<float*>(lds_all);
        constexpr int SLOT = (TM * TN * 4 + 2 * TM) * 64;           // floats per wave
        if (kgrp > 0) {
            float* mine = red + ((kgrp - 1) * (NT / 64) + wave) * SLOT;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) *reinterpret_cast<f32x4*>(mine + ((i * TN + j) * 64 + lane) * 4) = acc[i][j];
            if (LNA) {
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    mine[TM * TN * 256 + (2 * i) * 64 + lane] = ln_sum[i];
                    mine[TM * TN * 256 + (2 * i + 1) * 64 + lane] = ln_sq[i];
                }
            }
        }
        __syncthreads();
        if (kgrp > 0) return;
#pragma unroll
        for (int gk = 1; gk < KG; ++gk) {
            const float* theirs = red + ((gk - 1) * (NT / 64) + wave) * SLOT;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const f32x4 o = *reinterpret_cast<const f32x4*>(theirs + ((i * TN + j) * 64 + lane) * 4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[i][j][r] += o[r];
                }
            if (LNA) {
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    ln_sum[i] += theirs[TM * TN * 256 + (2 * i) * 64 + lane];
                    ln_sq[i] += theirs[TM * TN * 256 + (2 * i + 1) * 64 + lane];
                }
            }
        }
    }

    // ---- epilogue (identical to gemm.hip) ----
    if (p.splitk > 1) {
        float* slab = p.partial + (long)blockIdx.z * p.M * p.N;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + wave_m * WROWS + i * 16 + l16;
            if (m >= p.M) continue;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wave_n * (BN / 2) + j * 16 + 4 * g;
                if (n < p.N) *reinterpret_cast<f32x4*>(slab + (long)m * p.N + n) = acc[i][j];
            }
        }
        return;
    }
    if (LNA) {
        // every lane ends with the statistics of its own TM output rows: fold the four k-slices (g = 0..3) of a row
        LbLnRows<TM> ln;
        const float inv_k = 1.f / (float)p.K;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            float su = ln_sum[i], sq = ln_sq[i];
            su += __shfl_xor(su, 16, LB_WAVE); sq += __shfl_xor(sq, 16, LB_WAVE);
            su += __shfl_xor(su, 32, LB_WAVE); sq += __shfl_xor(sq, 32, LB_WAVE);
            const float mean = su * inv_k;
            const float var = fmaxf(sq * inv_k - mean * mean, 0.f);
            ln.mean[i] = mean;
            ln.rstd[i] = rsqrtf(var + p.ln_eps);
        }
        const int row0 = m0 + wave_m * WROWS + l16;
        lb_gemm_tile_epilogue_rows_ln<TM, TN, GEGLU, true>(p, acc, [row0](int i) { return row0 + i * 16; },
                                                           n0 + wave_n * (BN / 2) + 4 * g,
                                                           n0 + wave_n * (BN / 64) * 16 + 4 * g, &ln);
        return;
    }
    if constexpr (!GEGLU) {     // the one-round-trip form wherever it applies (lb_gemm.h); else the general epilogue below
        const int row_lo = m0 + wave_m * WROWS, row_hi = row_lo + 16 * TM;
        int batch = -1;
        if (p.rowvec) {
            const int last = (row_hi < p.M ? row_hi : p.M) - 1;
            const int b0 = row_lo / p.rows_per_batch;
            batch = (last >= row_lo && last / p.rows_per_batch == b0) ? b0 : -1;
        }
        const int r0 = row_lo + l16;
        if (lb_gemm_tile_epilogue_lean<TM, TN, false, false>(p, acc, [r0](int i) { return r0 + i * 16; }, row_lo, row_hi,
                                                             [l16](int i) { return l16 + i * 16; }, (long)row_lo, batch,
                                                             n0 + wave_n * (BN / 2)))
            return;
    }
    lb_gemm_tile_epilogue<TM, TN, GEGLU>(p, acc, m0 + wave_m * WROWS + l16, n0 + wave_n * (BN / 2) + 4 * g,
                                         n0 + wave_n * (BN / 64) * 16 + 4 * g);
}

template <int BM, int BN, int WMW>
constexpr int glds_stage_rows() {                       // A rows + weight rows incl. the padding of a partial staging round
    constexpr int NT = WMW * 128, RPI = NT / 8, WI = (BN * 8 + NT - 1) / NT;
    return BM + WI * RPI;
}

// K-group form (KG = 2) of a plain / GEGLU / LayerNorm-folded GEMM tile
template <int BM, int BN, int S, int WMW, int KG>
static int launch_glds_kgroups(const LbGemmParams& p, dim3 grid, hipStream_t stream) {
    const size_t smem = (size_t)KG * S * glds_stage_rows<BM, BN, WMW>() * BK * sizeof(f16);
    const bool geglu = (p.flags & LB_GEMM_GEGLU) != 0, lna = (p.flags & LB_GEMM_LN_A) != 0;
    const dim3 block(WMW * 128 * KG);
    if (geglu && lna) hipLaunchKernelGGL((gemm_f16_glds_kernel<BM, BN, false, true, S, WMW, 1, KG>), grid, block, smem, stream, p);
    else if (geglu) hipLaunchKernelGGL((gemm_f16_glds_kernel<BM, BN, false, true, S, WMW, 0, KG>), grid, block, smem, stream, p);
    else if (lna) hipLaunchKernelGGL((gemm_f16_glds_kernel<BM, BN, false, false, S, WMW, 1, KG>), grid, block, smem, stream, p);
    else hipLaunchKernelGGL((gemm_f16_glds_kernel<BM, BN, false, false, S, WMW, 0, KG>), grid, block, smem, stream, p);
    return 0;
}

template <int BM, int BN, int S, int WMW = 2>
static int launch_glds_variant(const LbGemmParams& p, dim3 grid, hipStream_t stream) {
    const size_t smem = (size_t)S * glds_stage_rows<BM, BN, WMW>() * BK * sizeof(f16);
    const bool geglu = (p.flags & LB_GEMM_GEGLU) != 0;
    const dim3 block(WMW * 128);
    const bool lna = (p.flags & LB_GEMM_LN_A) != 0;
    if (p.conv) hipLaunchKernelGGL((gemm_f16_glds_kernel<BM, BN, true, false, S, WMW>), grid, block, smem, stream, p);
    else if (geglu && lna) hipLaunchKernelGGL((gemm_f16_glds_kernel<BM, BN, false, true, S, WMW, 1>), grid, block, smem, stream, p);
    else if (geglu) hipLaunchKernelGGL((gemm_f16_glds_kernel<BM, BN, false, true, S, WMW>), grid, block, smem, stream, p);
    else if (lna) hipLaunchKernelGGL((gemm_f16_glds_kernel<BM, BN, false, false, S, WMW, 1>), grid, block, smem, stream, p);
    else hipLaunchKernelGGL((gemm_f16_glds_kernel<BM, BN, false, false, S, WMW>), grid, block, smem, stream, p);
    return 0;
}

// dynamic LDS above 64 KiB needs an opt-in per kernel; done once, outside of any stream capture
template <int BM, int BN, int S, int WMW = 2>
static void allow_lds() {
    const int smem = S * glds_stage_rows<BM, BN, WMW>() * BK * (int)sizeof(f16);
    hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f16_glds_kernel<BM, BN, true, false, S, WMW>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f16_glds_kernel<BM, BN, false, true, S, WMW>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f16_glds_kernel<BM, BN, false, false, S, WMW>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f16_glds_kernel<BM, BN, false, true, S, WMW, 1>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f16_glds_kernel<BM, BN, false, false, S, WMW, 1>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, smem);
}

void lb_gemm_glds_init() {
    static unsigned long long seen = 0;
    LbFirstCallOnDevice once(seen);         // (held until every attribute below is set: a second thread waits here)
    if (!once.first) return;
    allow_lds<128, 128, 2, 2>(); allow_lds<128, 128, 3>(); allow_lds<128, 128, 4>();
    allow_lds<128, 64, 2, 2>(); allow_lds<128, 64, 3>(); allow_lds<128, 64, 4>();
    allow_lds<64, 64, 2>(); allow_lds<64, 64, 3, 2>(); allow_lds<64, 64, 4>();
    allow_lds<256, 128, 2, 4>(); allow_lds<256, 128, 3, 4>();
    allow_lds<256, 256, 2, 4>();
    allow_lds<192, 128, 3, 3>();
    allow_lds<192, 128, 3, 4>();
    {   // 64x64, two K-groups: 2 x 3 x 16 KiB
        const int smem = 2 * 3 * glds_stage_rows<64, 64, 2>() * BK * (int)sizeof(f16);
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f16_glds_kernel<64, 64, false, true, 3, 2, 1, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f16_glds_kernel<64, 64, false, true, 3, 2, 0, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f16_glds_kernel<64, 64, false, false, 3, 2, 1, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f16_glds_kernel<64, 64, false, false, 3, 2, 0, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    }
}

// tile: 1 = 128x128, 2 = 128x64, 3 = 64x64, 4 = 256x128 (8 waves), 5 = 256x256 (8 waves, 64x128 per wave),
// 7 = 192x128 (6 waves, 3-stage ring); stages: 2..4 (0 = default for the tile).
// Round-3 experiments that were built, verified and measured, and are NOT in this file (git history: commits 1e5bf26 ..
// "Narrow-N conv kernel"; records in profiles/): deeper single-step rings (S = 6 / 8) and a double-step form (two K-tiles
// per barrier) for the 64x64 / 128x64 tiles - no gain at any depth (profiles/r03_small_m_sweep*.txt: the B = 2 GEMMs
// stay at ~10 us whatever the loop structure); an extra L2-prefetch WAVE per block for the 6- / 8-wave tiles, running 5
// K-tiles ahead of the direct-to-LDS requests - +-0..4 % (profiles/r03_prefetch_wave_ab.txt): neither the memory latency
// under the ring depth nor the per-barrier episode is what bounds these loops.
int lb_gemm_launch_glds(const LbGemmParams& p, int tile, int stages, dim3 grid, hipStream_t stream) {
    if (tile == 7) return launch_glds_variant<192, 128, 3, 3>(p, grid, stream);            // 3 x 42 KiB
    if (tile == 10) return launch_glds_variant<192, 128, 3, 4>(p, grid, stream);           // 8 waves x (48 x 64): 3 x 40 KiB
    if (tile == 11) {                                                                      // 64x64, two K-groups of 4 waves (no conv form)
        if (p.conv) return launch_glds_variant<64, 64, 3, 2>(p, grid, stream);
        return launch_glds_kgroups<64, 64, 3, 2, 2>(p, grid, stream);
    }
    if (tile == 5) return launch_glds_variant<256, 256, 2, 4>(p, grid, stream);     // 128 KiB: two stages only
    if (tile == 4) {
        if (stages == 3) return launch_glds_variant<256, 128, 3, 4>(p, grid, stream);
        return launch_glds_variant<256, 128, 2, 4>(p, grid, stream);
    }
    if (tile == 1) {
        if (stages == 2) return launch_glds_variant<128, 128, 2, 2>(p, grid, stream);
        if (stages == 4) return launch_glds_variant<128, 128, 4>(p, grid, stream);
        return launch_glds_variant<128, 128, 3>(p, grid, stream);
    }
    if (tile == 2) {
        if (stages == 2) return launch_glds_variant<128, 64, 2, 2>(p, grid, stream);
        if (stages == 4) return launch_glds_variant<128, 64, 4>(p, grid, stream);
        return launch_glds_variant<128, 64, 3>(p, grid, stream);
    }
    if (stages == 2) return launch_glds_variant<64, 64, 2>(p, grid, stream);
    if (stages == 3) return launch_glds_variant<64, 64, 3, 2>(p, grid, stream);
    return launch_glds_variant<64, 64, 4>(p, grid, stream);
}
