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
#include "lb_common.h"
#include "lb_gemm.h"

#define BK 64
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
// LNA: 0 none; 1 LayerNorm of A folded in, row statistics accumulated from the A fragments inside the K loop; 2 folded
// in with the statistics read from p.row_stats (left there by the epilogue of the GEMM that produced A, STATS).
// STATS: this GEMM's epilogue writes the row statistics of its output (LB_GEMM_ROW_STATS).
// PF > 0: one EXTRA wave per block (threads NT .. NT+63) is an L2 prefetcher: while the compute waves multiply K-tile t it
// touches one byte of every 128-B line of K-tile t + PF (BM rows of A, BN rows of W), never waits for the data and takes
// part in the per-tile barrier only.  The compute waves' direct-to-LDS requests of that tile (issued S-1 tiles ahead) then
// hit the XCD's L2 instead of paying Infinity-Cache / HBM latency with only S-1 tiles (96 KiB) in flight per CU: the ring
// depth no longer has to cover the memory latency, only the L2 latency.  (A prefetch issued by the compute waves
// themselves would not do: vmcnt retires in order, so a slow prefetch would hold back the wait for the tile behind it.)
// KD = 2 ("double step", 4-wave tiles on small grids): TWO K-tiles per barrier.  A 64x64 tile gives a wave only 8 MFMAs
// (128 cycles of matrix work) per K-tile against ~700 cycles of fixed cost per barrier episode (counted wait + barrier,
// the LDS round trip of the fragment reads, the request issue): the B = 2 anchor programs' GEMMs ran at 0.4 us per
// K-tile whatever the ring depth (profiles/r03_small_m_sweep.txt).  Here one episode waits for tiles t and t+1, reads all
// their fragments in one burst (one exposed LDS latency) and issues 2 x 16 MFMAs; the ring holds S (even) tiles, S-2 in
// flight.  K-tiles past the end of an odd-length K range are zero tiles (masked requests), as in the prologue.
template <int BM, int BN, bool CONV, bool GEGLU, int S, int WMW, int LNA = 0, bool STATS = false, int PF = 0, int KD = 1>
__global__ void __launch_bounds__(WMW * 128 + (PF ? 64 : 0)) gemm_f16_glds_kernel(const LbGemmParams p) {
    static_assert(KD == 1 || (KD == 2 && S % 2 == 0 && S >= 4 && PF == 0), "double-step form: even ring of >= 4 stages");
    constexpr int NT = WMW * 128;           // COMPUTE threads per block
    constexpr int RPI = NT / 8;             // tile rows covered by one round of wave instructions
    constexpr int AI = BM * 8 / NT;         // wave instructions (16-B chunks per thread) of A per K-tile
    constexpr int WI = (BN * 8 + NT - 1) / NT;
    constexpr int NL = AI + WI;             // VMEM loads per thread per K-tile
    constexpr int WROWS = BM / WMW;         // rows of the block tile owned by one wave
    constexpr int WPAD = WI * RPI;          // weight rows staged per K-tile (>= BN)
    static_assert(BM * 8 % NT == 0, "A tile rows must be a whole number of staging rounds");
    constexpr int TM = WROWS / 16, TN = BN / 32;
    constexpr int STAGE = (BM + WPAD) * BK; // halves per ring stage
    extern __shared__ __attribute__((aligned(16))) f16 lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
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

    if constexpr (PF > 0) {
        if (wave == WMW * 2) {              // ---- the prefetch wave ----
            constexpr int NPL = (BM + BN + 63) / 64;
            const lb_half* rowp[NPL];
#pragma unroll
            for (int j = 0; j < NPL; ++j) {
                const int r = lane + 64 * j;
                const lb_half* ptr = nullptr;
                if (r < BM) {
                    const int m = m0 + r;
                    if (!CONV && m < p.M) ptr = p.A + (long)m * p.lda;
                } else if (r < BM + BN) {
                    const int tr = r - BM;
                    int n;
                    bool ok;
                    if (GEGLU) {
                        const int sub = tr >> 4;
                        const int nn = n0 + (sub >> 1) * 16 + (tr & 15);
                        n = (sub & 1) * (p.N / 2) + nn;
                        ok = nn < p.N / 2;
                    } else {
                        n = n0 + tr;
                        ok = n < p.N;
                    }
                    if (ok) ptr = p.W + (long)n * p.ldw;
                }
                rowp[j] = ptr;
            }
            unsigned sink = 0;              // the loads' common destination: kept live (and in one register) to the end
            // tiles S-1 .. PF-1 are requested by the compute waves during their first iterations: warm them right away
            // (tiles 0 .. S-2 are in the compute waves' prologue burst already)
            for (int k0 = (kt_begin + S - 1) * BK; k0 < (kt_begin + PF) * BK && k0 < k_end; k0 += BK) {
#pragma unroll
                for (int j = 0; j < NPL; ++j)
                    if (rowp[j]) asm volatile("global_load_ubyte %0, %1, off" : "+v"(sink) : "v"(rowp[j] + k0) : "memory");
            }
            int k = (kt_begin + PF) * BK;
            for (int t = 0; t < nkt; ++t) {
                if (k < k_end) {
#pragma unroll
                    for (int j = 0; j < NPL; ++j)
                        if (rowp[j]) asm volatile("global_load_ubyte %0, %1, off" : "+v"(sink) : "v"(rowp[j] + k) : "memory");
                }
                k += BK;
                __builtin_amdgcn_s_barrier();   // (one barrier per K-tile, like the compute waves)
            }
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(sink)::"memory");
            return;
        }
    }
    // thread (r = tid>>3 (+32 i), s = tid&7) owns physical chunk s of tile row r and fetches logical
    // chunk s ^ (r & 7); (r & 7) == (lane >> 3) for every i because row groups start at multiples of 8
    const int row0 = tid >> 3;
    const int cl = (tid & 7) ^ ((tid >> 3) & 7);     // logical chunk fetched by this lane

    long a_off[AI];
    int a_iy[AI], a_ix[AI];
    bool a_ok[AI];
    int ci = 0, ky = 0, kx = 0;
    int kcur = kt_begin * BK + cl * 8;
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

    // request number idx (0..NL-1: AI rows of A, then WI rows of W) of the current K-tile into the ring
    // stage at `base` (branch-free: masked chunks read the zero page)
    auto issue_one = [&](int idx, f16* base, bool k_ok) {
        const lb_half* src;
        f16* dst;
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
        kcur += BK;
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
    auto issue_tile = [&](int st) {
        f16* base = lds + st * STAGE;
        const bool k_ok = kcur < k_end;
#pragma unroll
        for (int idx = 0; idx < NL; ++idx) issue_one(idx, base, k_ok);
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

    if constexpr (KD == 2) {
        // ---- double-step form: prologue tiles 0 .. S-3, then per episode tiles (t, t+1) ----
#pragma unroll
        for (int s = 0; s < S - 2; ++s) issue_tile(s);
        int st = 0;
        for (int t = 0; t < nkt; t += 2) {
            wait_vm_barrier<(S - 4) * NL>();  // tiles t, t+1 landed everywhere; the stages of tiles t-2, t-1 are free
            int r0 = st - 2, r1 = st - 1;
            if (r0 < 0) { r0 += S; r1 += S; }
            issue_tile(r0);                   // tiles t+S-2, t+S-1 (zero tiles past the end)
            issue_tile(r1);
            f16x8 a0[TM], w0[TN], a1[TM], w1[TN], a2[TM], w2[TN], a3[TM], w3[TN];
            read_frags(st, 0, a0, w0);
            read_frags(st, 1, a1, w1);
            read_frags(st + 1, 0, a2, w2);
            read_frags(st + 1, 1, a3, w3);
            if (LNA == 1) {
                ln_accumulate<TM>(a0, ln_sum, ln_sq); ln_accumulate<TM>(a1, ln_sum, ln_sq);
                ln_accumulate<TM>(a2, ln_sum, ln_sq); ln_accumulate<TM>(a3, ln_sum, ln_sq);
            }
            mma_rows(a0, w0, 0, TM);
            mma_rows(a1, w1, 0, TM);
            mma_rows(a2, w2, 0, TM);
            mma_rows(a3, w3, 0, TM);
            st = st + 2 == S ? 0 : st + 2;
        }
    } else {
    // ---- prologue: tiles 0 .. S-2 in flight ----
#pragma unroll
    for (int s = 0; s < S - 1; ++s) issue_tile(s);

    // One K-tile per iteration.  The NL requests of tile t+S-1 are NOT issued in one burst after the
    // barrier (all waves of a block would then sit in the VMEM issue queue together and start their
    // MFMAs together): they are spread over the four MFMA groups of tile t, so a wave blocked on a
    // full memory pipeline has MFMAs of its own still executing.  sched_barrier pins the order.
    // (tools/gemm_ablate.py: 5-13 % on the plain / GEGLU contractions, first > 1 PFLOP/s at 8192^3.)
    constexpr int HM = TM > 1 ? TM / 2 : 1;
    int st = 0;                               // ring stage of tile t
    for (int t = 0; t < nkt; ++t) {
        wait_vm_barrier<(S - 2) * NL>();      // tile t landed everywhere; stage (t-1)%S free everywhere
        int refill = st - 1;
        if (refill < 0) refill += S;
        if (LB_ABLATE == 2) {
            issue_tile(refill);
        } else if (LB_BURST || (CONV && WMW == 2)) {   // 4-wave conv tiles measured 3-6 % faster with the burst
            issue_tile(refill);               // tile t+S-1 (masked to zeros past the end)
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
                if (idx * 4 / NL == 0) issue_one(idx, base, k_ok);
            __builtin_amdgcn_sched_barrier(0);
            mma_rows(a0, w0, 0, HM);
            read_frags(st, 1, a1, w1);
            if (LNA == 1) ln_accumulate<TM>(a0, ln_sum, ln_sq);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int idx = 0; idx < NL; ++idx)
                if (idx * 4 / NL == 1) issue_one(idx, base, k_ok);
            __builtin_amdgcn_sched_barrier(0);
            mma_rows(a0, w0, HM, TM);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int idx = 0; idx < NL; ++idx)
                if (idx * 4 / NL == 2) issue_one(idx, base, k_ok);
            __builtin_amdgcn_sched_barrier(0);
            mma_rows(a1, w1, 0, HM);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int idx = 0; idx < NL; ++idx)
                if (idx * 4 / NL == 3) issue_one(idx, base, k_ok);
            advance_k();
            if (LNA == 1) ln_accumulate<TM>(a1, ln_sum, ln_sq);
            __builtin_amdgcn_sched_barrier(0);
            mma_rows(a1, w1, HM, TM);
        }
        st = st + 1 == S ? 0 : st + 1;
    }
    }   // KD
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the masked tail requests before exit/epilogue

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
            float su, sq;
            if (LNA == 1) {
                su = ln_sum[i];
                sq = ln_sq[i];
            } else {        // the producer's 32-column slots of this row, four interleaved subsets (one per g), fixed order
                const int m = m0 + wave_m * WROWS + i * 16 + l16;
                const float2* st = reinterpret_cast<const float2*>(p.row_stats) + (m < p.M ? m : p.M - 1);
                su = sq = 0.f;
                for (int sl = g; sl < p.ln_nslots; sl += 4) {
                    const float2 v = st[(long)sl * p.M];
                    su += v.x;
                    sq += v.y;
                }
            }
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
    if (STATS) {
        lb_gemm_tile_epilogue_stats<TM, TN>(p, acc, m0 + wave_m * WROWS + l16, n0 + wave_n * (BN / 2) + 4 * g,
                                            n0 + wave_n * (BN / 64) * 16 + 4 * g);
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

// EXTRA: the row-statistics kernels (LNA = 2 consumers, STATS producers) exist for this (tile, stages) pair - only the
// default stage count of every tile, to keep the number of instantiations down; lb_gemm_f16 picks that stage count
// whenever a launch asks for them.
static int g_glds_prefetch = 0;            // K-tiles ahead the prefetch wave runs (0 = no prefetch wave); lb_gemm_set_prefetch
void lb_gemm_glds_set_prefetch(int tiles_ahead) { g_glds_prefetch = tiles_ahead; }
#define LB_PF_AHEAD 5

template <int BM, int BN, int S, int WMW = 2, bool EXTRA = false>
static int launch_glds_variant(const LbGemmParams& p, dim3 grid, hipStream_t stream) {
    const size_t smem = (size_t)S * glds_stage_rows<BM, BN, WMW>() * BK * sizeof(f16);
    const bool geglu = (p.flags & LB_GEMM_GEGLU) != 0;
    const dim3 block(WMW * 128);
    const bool lna = (p.flags & LB_GEMM_LN_A) != 0;
    if constexpr (WMW >= 3) {               // prefetch-wave form: plain / GEGLU contractions on the 6- and 8-wave tiles
        if (g_glds_prefetch > 0 && !p.conv && !lna && !(p.flags & LB_GEMM_ROW_STATS) && p.splitk <= 1) {
            static bool allowed = false;
            if (!allowed) {
                hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f16_glds_kernel<BM, BN, false, false, S, WMW, 0, false, LB_PF_AHEAD>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
                hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f16_glds_kernel<BM, BN, false, true, S, WMW, 0, false, LB_PF_AHEAD>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
                allowed = true;
            }
            const dim3 blockp(WMW * 128 + 64);
            if (geglu) hipLaunchKernelGGL((gemm_f16_glds_kernel<BM, BN, false, true, S, WMW, 0, false, LB_PF_AHEAD>), grid, blockp, smem, stream, p);
            else hipLaunchKernelGGL((gemm_f16_glds_kernel<BM, BN, false, false, S, WMW, 0, false, LB_PF_AHEAD>), grid, blockp, smem, stream, p);
            return 0;
        }
    }
    const bool from_stats = lna && p.row_stats != nullptr;
    const bool stats = (p.flags & LB_GEMM_ROW_STATS) != 0;
    if constexpr (EXTRA) {
        if (stats) {
            hipLaunchKernelGGL((gemm_f16_glds_kernel<BM, BN, false, false, S, WMW, 0, true>), grid, block, smem, stream, p);
            return 0;
        }
        if (from_stats) {
            if (geglu) hipLaunchKernelGGL((gemm_f16_glds_kernel<BM, BN, false, true, S, WMW, 2>), grid, block, smem, stream, p);
            else hipLaunchKernelGGL((gemm_f16_glds_kernel<BM, BN, false, false, S, WMW, 2>), grid, block, smem, stream, p);
            return 0;
        }
    } else if (stats || from_stats) {
        LB_REQUIRE(false, "lb_gemm_f16: row-statistics kernels exist for the default stage count of a tile only");
    }
    if (p.conv) hipLaunchKernelGGL((gemm_f16_glds_kernel<BM, BN, true, false, S, WMW>), grid, block, smem, stream, p);
    else if (geglu && lna) hipLaunchKernelGGL((gemm_f16_glds_kernel<BM, BN, false, true, S, WMW, 1>), grid, block, smem, stream, p);
    else if (geglu) hipLaunchKernelGGL((gemm_f16_glds_kernel<BM, BN, false, true, S, WMW>), grid, block, smem, stream, p);
    else if (lna) hipLaunchKernelGGL((gemm_f16_glds_kernel<BM, BN, false, false, S, WMW, 1>), grid, block, smem, stream, p);
    else hipLaunchKernelGGL((gemm_f16_glds_kernel<BM, BN, false, false, S, WMW>), grid, block, smem, stream, p);
    return 0;
}

// double-step variants (KD = 2): 64x64 with 8 / 4 stages, 128x64 with 6 / 4 stages
template <int BM, int BN, int S>
static int launch_glds_double(const LbGemmParams& p, dim3 grid, hipStream_t stream) {
    constexpr size_t smem = (size_t)S * glds_stage_rows<BM, BN, 2>() * BK * sizeof(f16);
    const bool geglu = (p.flags & LB_GEMM_GEGLU) != 0, lna = (p.flags & LB_GEMM_LN_A) != 0;
    static bool allowed = false;
    if (!allowed) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f16_glds_kernel<BM, BN, true, false, S, 2, 0, false, 0, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f16_glds_kernel<BM, BN, false, true, S, 2, 0, false, 0, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f16_glds_kernel<BM, BN, false, false, S, 2, 0, false, 0, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f16_glds_kernel<BM, BN, false, true, S, 2, 1, false, 0, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f16_glds_kernel<BM, BN, false, false, S, 2, 1, false, 0, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        allowed = true;
    }
    const dim3 block(256);
    if (p.conv) hipLaunchKernelGGL((gemm_f16_glds_kernel<BM, BN, true, false, S, 2, 0, false, 0, 2>), grid, block, smem, stream, p);
    else if (geglu && lna) hipLaunchKernelGGL((gemm_f16_glds_kernel<BM, BN, false, true, S, 2, 1, false, 0, 2>), grid, block, smem, stream, p);
    else if (geglu) hipLaunchKernelGGL((gemm_f16_glds_kernel<BM, BN, false, true, S, 2, 0, false, 0, 2>), grid, block, smem, stream, p);
    else if (lna) hipLaunchKernelGGL((gemm_f16_glds_kernel<BM, BN, false, false, S, 2, 1, false, 0, 2>), grid, block, smem, stream, p);
    else hipLaunchKernelGGL((gemm_f16_glds_kernel<BM, BN, false, false, S, 2, 0, false, 0, 2>), grid, block, smem, stream, p);
    return 0;
}

// stages >= 16 selects the double-step form: 16 + S (S = 4, 6, 8)
int lb_gemm_launch_glds_double(const LbGemmParams& p, int tile, int stages, dim3 grid, hipStream_t stream) {
    if (tile == 3) {
        if (stages >= 8) return launch_glds_double<64, 64, 8>(p, grid, stream);
        if (stages >= 6) return launch_glds_double<64, 64, 6>(p, grid, stream);
        return launch_glds_double<64, 64, 4>(p, grid, stream);
    }
    if (stages >= 6) return launch_glds_double<128, 64, 6>(p, grid, stream);
    return launch_glds_double<128, 64, 4>(p, grid, stream);
}

// dynamic LDS above 64 KiB needs an opt-in per kernel; done once, outside of any stream capture
template <int BM, int BN, int S, int WMW = 2, bool EXTRA = false>
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
    if constexpr (EXTRA) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f16_glds_kernel<BM, BN, false, true, S, WMW, 2>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f16_glds_kernel<BM, BN, false, false, S, WMW, 2>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f16_glds_kernel<BM, BN, false, false, S, WMW, 0, true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    }
}

void lb_gemm_glds_init() {
    static bool done = false;
    if (done) return;
    done = true;
    allow_lds<128, 128, 2, 2, true>(); allow_lds<128, 128, 3>(); allow_lds<128, 128, 4>();
    allow_lds<128, 64, 2, 2, true>(); allow_lds<128, 64, 3>(); allow_lds<128, 64, 4>();
    allow_lds<64, 64, 2>(); allow_lds<64, 64, 3, 2, true>(); allow_lds<64, 64, 4>();
    allow_lds<256, 128, 2, 4>(); allow_lds<256, 128, 3, 4, true>();
    allow_lds<256, 256, 2, 4, true>();
    allow_lds<192, 128, 3, 3, true>();
}

// tile: 1 = 128x128, 2 = 128x64, 3 = 64x64, 4 = 256x128 (8 waves), 5 = 256x256 (8 waves, 64x128 per wave),
// 7 = 192x128 (6 waves, 3-stage ring); stages: 2..4 (0 = default for the tile); 16 + S = the double-step form of the
// 64x64 / 128x64 tiles with an S-stage ring (S = 4, 6, 8).  (Deeper single-step rings, S = 6 / 8, were measured and bought
// nothing: profiles/r03_small_m_sweep.txt - the small tiles are bound by the per-barrier episode, not by memory latency.)
int lb_gemm_launch_glds(const LbGemmParams& p, int tile, int stages, dim3 grid, hipStream_t stream) {
    if (stages >= 16 && (tile == 2 || tile == 3) && !(p.flags & LB_GEMM_ROW_STATS) && !((p.flags & LB_GEMM_LN_A) && p.row_stats != nullptr))
        return lb_gemm_launch_glds_double(p, tile, stages - 16, grid, stream);
    if (stages >= 16) stages = 0;
    if (tile == 7) return launch_glds_variant<192, 128, 3, 3, true>(p, grid, stream);            // 3 x 42 KiB
    if (tile == 5) return launch_glds_variant<256, 256, 2, 4, true>(p, grid, stream);     // 128 KiB: two stages only
    if (tile == 4) {
        if (stages == 3) return launch_glds_variant<256, 128, 3, 4, true>(p, grid, stream);
        return launch_glds_variant<256, 128, 2, 4>(p, grid, stream);
    }
    if (tile == 1) {
        if (stages == 2) return launch_glds_variant<128, 128, 2, 2, true>(p, grid, stream);
        if (stages == 4) return launch_glds_variant<128, 128, 4>(p, grid, stream);
        return launch_glds_variant<128, 128, 3>(p, grid, stream);
    }
    if (tile == 2) {
        if (stages == 2) return launch_glds_variant<128, 64, 2, 2, true>(p, grid, stream);
        if (stages == 4) return launch_glds_variant<128, 64, 4>(p, grid, stream);
        return launch_glds_variant<128, 64, 3>(p, grid, stream);
    }
    if (stages == 2) return launch_glds_variant<64, 64, 2>(p, grid, stream);
    if (stages == 3) return launch_glds_variant<64, 64, 3, 2, true>(p, grid, stream);
    return launch_glds_variant<64, 64, 4>(p, grid, stream);
}
