// Ping-pong MFMA GEMM for gfx950: 256 x 256 x 64 block tile, 8 waves in two groups that alternate between a "load"
// segment (LDS fragment reads + LDS-DMA requests) and a "compute" segment (16 MFMAs).  Same contract and epilogues as
// gemm.hip / gemm_glds.hip (plain and GEGLU GEMMs; lb_gemm_f16 routes chip-filling grids here, tile code 9), bit-identical
// results: every accumulator sees the same K order.
//
// Why a third main loop.  The lock-step 8-wave tiles of gemm_glds.hip stage whole 64 KiB K-tiles through a two-stage ring:
// at most one K-tile is in flight and it is drained at every barrier; they spend ~1/3 of their wave cycles parked at
// `vmcnt` + `s_barrier` (profiles/r02_gemm_pmc.json).  Here
//   * a K-tile is staged as FOUR half-tiles of 16 KiB, cut so that every half-tile is read in exactly ONE of the K-tile's
//     four phases (by all waves) and is dead afterwards:
//         A0 = rows {wr*128 + 0..63}, A1 = rows {wr*128 + 64..127}  (wr = 0, 1: the two wave rows)
//         B0 = cols {wc*64 + 0..31},  B1 = cols {wc*64 + 32..63}    (wc = 0..3: the four wave columns)
//     phase 0 multiplies A0 x B0 (reads A0, B0), phase 1 A0 x B1 (reads B1), phase 2 A1 x B1 (reads A1), phase 3 A1 x B0
//     (reads nothing: B0 stayed in registers) - 24 ds_read_b128 per wave per K-tile, the minimum for a 128 x 64 wave tile;
//   * half-tile h = 4 t + J (J = 0 A0, 1 B0, 2 B1, 3 A1 of K-tile t) lives in slot h mod 8 of a 128 KiB ring; load segment p
//     requests half-tile p + 6 (a slot is requested again two phases after its only read), and the only waits are counted:
//     four half-tiles = 64 KiB (late group: three) stay in flight per CU across every barrier;
//   * ONE barrier per phase.  Between two barriers the early group (wr = 0) runs [compute phase k, load phase k + 1] while
//     the late group (wr = 1; waves w and w + 4 share a SIMD) runs [load phase k, compute phase k]: matrix beside memory, then
//     memory beside matrix; the hand-over in the middle is not a rendezvous - the late group's MFMAs follow the early
//     group's through the SIMD's matrix pipe.
//   * tile order inside an XCD's run: GM block rows down, then the next block column, so that the ~32 tiles an XCD works on
//     at a time form a patch (GM + 32 / GM operand panels through its L2 per K-tile) instead of a 1 x 32 strip (33).
//
// Hazards (MI355X: an LDS-DMA is ordered for a ds_read only by the issuing wave's vmcnt followed by a barrier the reader
// has passed; a slot may be re-requested only after every wave's reads of it have RETURNED):
//   RAW: half-tile h is consumed in phase q = h - j + {0, 0, 1, 2}[j].  The early group reads it in interval q - 1 (second
//        half), the late group in interval q: every wave must have waited for its own part before the barrier that ends
//        interval q - 2 - the early group in load segment q - 1 (vmcnt(8): it has requested through half-tile q + 5), the
//        late group in load segment q - 2 (vmcnt(6): through q + 4).
//   WAR: load segment p requests into the slot of half-tile p - 2, last read in a phase <= p - 2, i.e. in an EARLIER interval
//        for both groups; reads issued at the end of an interval are waited for by their own wave before it computes.
//
// Where it stands (profiles/r04_gemm_bench_call*.txt, r05_gemm_bench_call4.txt): 1.3-1.4 PFLOP/s on 8192^3 against 1.46-1.51 for the
// vendor library's stream-K kernels (hand-scheduled Tensile assembly, 256x240 / 224x224 macro tiles); on the programs' shapes at or
// ahead of the library except where the grid is one partial round.  Variants with a deeper ring (10 slots / 8 ahead), a shallower
// one (4 ahead), two barriers per phase, the requests inside the compute segments, and a one-wave-per-SIMD 4 x (128 x 128) form were
// built, verified and measured - none beat this one (git history).  (The round-4 header derived a ceiling for this tile from the
// operand-feed probe tools/probes/feed_rate.cpp; that probe streams private panels per XCD and is HBM-bound by construction - the
// vendor kernels exceed its "strip" row - so the claim is withdrawn.)
// LayerNorm folding (LB_GEMM_LN_A) is NOT offered here: two forms were built and verified (row statistics by v_dot2 on the A
// fragments inside the compute segments; and split over the four wave columns, inside the load segments) and cost the
// loop +40 % / +19 % on the GEGLU projection and +28 % / +17 % on q|k|v - more than the 8.7 us LayerNorm launch they
// replace (profiles/r04_ln_pp_ab.txt): v_dot2 does not hide beside the MFMAs of either wave of a SIMD.
//
// Replaces (third party, reached from /root/reference/latentblending/diffusers_holder.py:336): the torch.nn.Linear
// layers of the SDXL UNet's transformer blocks at batch >= 8 (fused q|k|v, GEGLU, 640-wide feed-forward output) and the
// per-branch context K|V projection.
#include "lb_common.h"
#include "lb_gemm.h"

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

#define PP_BK 64
#define PP_SLOT_H 8192           // halves per half-tile slot: 128 rows x 64 halves = 16 KiB
#define PP_NS 8                  // ring slots
#define PP_D 6                   // half-tiles requested ahead of the consumer
#define PP_LDS_BYTES (PP_NS * PP_SLOT_H * 2)

template <int V> struct PPInt { static constexpr int value = V; };
typedef PPInt<0> J0; typedef PPInt<1> J1; typedef PPInt<2> J2; typedef PPInt<3> J3;

template <int N> __device__ __forceinline__ void pp_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void pp_barrier() { asm volatile("s_barrier" ::: "memory"); }

template <bool GEGLU>
__global__ void __launch_bounds__(512) gemm_f16_pp_kernel(const LbGemmParams p) {
    constexpr int BM = 256, BN = 256;
    extern __shared__ __attribute__((aligned(16))) f16 lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int g = lane >> 4, l16 = lane & 15;

    // ---- block -> tile: XCD-aware bijective remap, then the grouped order (p.reserved_ = GM block rows per group) ----
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
    int block_m, block_n;
    if (p.reserved_ > 0) {
        const int gm = p.reserved_;
        const int per_group = gm * n_blocks;
        const int group = bid / per_group, first_m = group * gm;
        const int rows = m_blocks - first_m < gm ? m_blocks - first_m : gm;
        const int in_group = bid - group * per_group;
        block_m = first_m + in_group % rows;
        block_n = in_group / rows;
    } else {                         // strip order of the other kernels (the operand with more bytes is the shared one)
        const bool w_dominant = n_eff > p.M;
        block_n = w_dominant ? bid / m_blocks : bid % n_blocks;
        block_m = w_dominant ? bid % m_blocks : bid / n_blocks;
    }
    const int m0 = block_m * BM;
    const int n0 = block_n * BN_OUT;

    const int k_tiles_total = p.K / PP_BK;                       // (launcher: K % 64 == 0)
    const int tiles_per_split = (k_tiles_total + p.splitk - 1) / p.splitk;
    const int kt_begin = blockIdx.z * tiles_per_split;
    int kt_end = kt_begin + tiles_per_split;
    if (kt_end > k_tiles_total) kt_end = k_tiles_total;
    const int nkt = kt_end - kt_begin;

    // ---- staging: a half-tile = 128 LDS rows of 128 B = two wave instructions per wave (rows n*64 + wave*8 + lr) ----
    // lane (lr = lane >> 3, s = lane & 7) owns physical chunk s of its row and fetches logical chunk s ^ (row & 7).
    // Per-lane state is ONE 32-bit byte offset per staged row; the K position is a scalar added to the uniform base.  No
    // masking: rows beyond M / N re-read the last valid row (their accumulator rows / columns are never stored), requests
    // beyond the K range re-read the last K-tile (staged into slots nobody reads again).
    const int lr = lane >> 3;
    const int cl = (lane & 7) ^ (lr & 7);
    unsigned a_off[4];               // [qm * 2 + n]: LDS row n*64 + wave*8 + lr of half-tile A_qm = block row n*128 + qm*64 + wave*8 + lr
    unsigned w_off[4];               // [qn * 2 + n]
#pragma unroll
    for (int qm = 0; qm < 2; ++qm)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            int m = m0 + n * 128 + qm * 64 + wave * 8 + lr;
            m = m < p.M ? m : p.M - 1;
            a_off[qm * 2 + n] = (unsigned)(((long)m * p.lda + cl * 8) * 2);
        }
#pragma unroll
    for (int qn = 0; qn < 2; ++qn)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            // LDS row r = n*64 + wave*8 + lr of half-tile B_qn belongs to wave column r >> 5, fragment (r >> 4) & 1, row r & 15
            const int wcc = 2 * n + (wave >> 2);
            long wrow;
            if (GEGLU) {             // fragment 0 = h, 1 = gate of output columns n0 + (wcc*2 + qn)*16 + 0..15
                int oc = n0 + (wcc * 2 + qn) * 16 + (wave & 1) * 8 + lr;
                oc = oc < n_eff ? oc : n_eff - 1;
                wrow = (long)((wave >> 1) & 1) * n_eff + oc;
            } else {
                const int col = n0 + wcc * 64 + qn * 32 + (wave & 3) * 8 + lr;
                wrow = col < p.N ? col : p.N - 1;
            }
            w_off[qn * 2 + n] = (unsigned)((wrow * p.ldw + cl * 8) * 2);
        }
    const char* const a_base = reinterpret_cast<const char*>(p.A) + (long)kt_begin * (PP_BK * 2);
    const char* const w_base = reinterpret_cast<const char*>(p.W) + (long)kt_begin * (PP_BK * 2);

    // ring slot (in halves) of half-tile J of K-tile `tile`
    auto slot_of = [&](int tile, int J) { return ((4 * tile + J) % PP_NS) * PP_SLOT_H; };
    // request half-tile J (0 = A0, 1 = B0, 2 = B1, 3 = A1) of K-tile `tile` (of this block's range) into its ring slot
    auto stage = [&](auto jc, int tile) {
        constexpr int J = decltype(jc)::value;
        constexpr bool IS_A = (J == 0 || J == 3);
        constexpr int Q = (J == 0 || J == 1) ? 0 : 1;                 // qm for A, qn for B
        const int kt = tile < nkt ? tile : nkt - 1;
        const long koff = (long)kt * (PP_BK * 2);                     // scalar
        f16* base = lds + slot_of(tile, J) + (wave * 8) * PP_BK;
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const char* src = (IS_A ? a_base : w_base) + koff + (size_t)(IS_A ? a_off[Q * 2 + n] : w_off[Q * 2 + n]);
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(base + n * 64 * PP_BK), 16, 0, 0);
        }
    };
    // load segment of phase (t, I) requests half-tile 4 t + I + D
    auto stage_ahead = [&](auto ic, int t) {
        constexpr int I = decltype(ic)::value, JJ = (I + PP_D) & 3, DT = (I + PP_D) >> 2;
        stage(PPInt<JJ>{}, t + DT);
    };

    // ---- fragment reads: lane (g, l16) reads row (16 i + l16) at 16-B chunk (4 ks + g) ^ (row & 7) ----
    const f16* a_rd[2];
    const f16* b_rd[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int ch = ((ks * 4 + g) ^ (l16 & 7)) << 3;
        a_rd[ks] = lds + (wr * 64 + l16) * PP_BK + ch;
        b_rd[ks] = lds + (wc * 32 + l16) * PP_BK + ch;
    }
    f16x8 af[4][2];                  // [i][ks]   rows qm*64 + 16 i of the wave's 128
    f16x8 b0[2][2], b1[2][2];        // [jj][ks]  columns qn*32 + 16 jj of the wave's 64
    auto read_a = [&](int soff) {    // soff = slot_of(t, J)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                af[i][ks] = *reinterpret_cast<const f16x8*>(a_rd[ks] + soff + i * 16 * PP_BK);
    };
    auto read_b = [&](int soff, f16x8 (&bf)[2][2]) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
                bf[jj][ks] = *reinterpret_cast<const f16x8*>(b_rd[ks] + soff + jj * 16 * PP_BK);
    };

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // the 16 MFMAs of one phase (s_setprio 1 around them: the partner wave's load segment never delays an MFMA issue)
    auto compute = [&](auto qmc, auto qnc, const f16x8 (&bf)[2][2]) {
        constexpr int QM = decltype(qmc)::value, QN = decltype(qnc)::value;
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
                    acc[QM * 4 + i][QN * 2 + jj] =
                        __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[jj][ks], af[i][ks], acc[QM * 4 + i][QN * 2 + jj], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- prologue: half-tiles 0 .. 5 (K-tile 0 whole, A0 / B0 of K-tile 1) ----
    stage(J0{}, 0); stage(J1{}, 0); stage(J2{}, 0); stage(J3{}, 0); stage(J0{}, 1); stage(J1{}, 1);
    pp_wait_vm<2 * (PP_D - 2)>();            // A0 / B0 of K-tile 0 have landed (this wave's part)
    pp_barrier();

    // load segments of the four phases of K-tile t (VM = the vmcnt this group waits for, -1 = none)
    auto load0 = [&](int t, auto vm) {       // reads A0, B0
        read_b(slot_of(t, 1), b0);
        read_a(slot_of(t, 0));
        stage_ahead(J0{}, t);
        if constexpr (decltype(vm)::value >= 0) pp_wait_vm<decltype(vm)::value>();
    };
    auto load1 = [&](int t, auto vm) {       // reads B1
        read_b(slot_of(t, 2), b1);
        stage_ahead(J1{}, t);
        if constexpr (decltype(vm)::value >= 0) pp_wait_vm<decltype(vm)::value>();
    };
    auto load2 = [&](int t, auto vm) {       // reads A1
        read_a(slot_of(t, 3));
        stage_ahead(J2{}, t);
        if constexpr (decltype(vm)::value >= 0) pp_wait_vm<decltype(vm)::value>();
    };
    auto load3 = [&](int t, auto vm) {       // reads nothing (B0 stayed in registers)
        stage_ahead(J3{}, t);
        if constexpr (decltype(vm)::value >= 0) pp_wait_vm<decltype(vm)::value>();
    };
    typedef PPInt<-1> NOWAIT;
    typedef PPInt<2 * (PP_D - 2)> VM_EARLY;  // 4 half-tiles stay in flight at the early group's waits
    typedef PPInt<2 * (PP_D - 3)> VM_LATE;   // 3 at the late group's (it waits one phase earlier)

    if (wr == 0) {
        // early group: [compute phase k | load phase k + 1] per interval
        load0(0, VM_EARLY{});
        pp_barrier();
        __builtin_amdgcn_sched_barrier(0);
        for (int t = 0; t < nkt; ++t) {
            compute(J0{}, J0{}, b0);
            load1(t, VM_EARLY{});            // ... A1 of this K-tile landed
            pp_barrier();
            compute(J0{}, J1{}, b1);
            load2(t, NOWAIT{});
            pp_barrier();
            compute(J1{}, J1{}, b1);
            load3(t, VM_EARLY{});            // ... A0 / B0 of the next K-tile landed
            pp_barrier();
            compute(J1{}, J0{}, b0);
            load0(t + 1, VM_EARLY{});        // ... B1 of the next K-tile landed (past the last K-tile: reads of stale slots into dead registers)
            pp_barrier();
        }
    } else {
        // late group: [load phase k | compute phase k] per interval; its waits cover what the EARLY group reads in the next interval
        pp_wait_vm<2 * (PP_D - 3)>();        // B1 of K-tile 0 (the early group reads it in interval 0)
        pp_barrier();
        __builtin_amdgcn_sched_barrier(0);
        for (int t = 0; t < nkt; ++t) {
            load0(t, VM_LATE{});             // ... A1 of this K-tile landed
            compute(J0{}, J0{}, b0);
            pp_barrier();
            load1(t, NOWAIT{});
            compute(J0{}, J1{}, b1);
            pp_barrier();
            load2(t, VM_LATE{});             // ... A0 / B0 of the next K-tile landed
            compute(J1{}, J1{}, b1);
            pp_barrier();
            load3(t, VM_LATE{});             // ... B1 of the next K-tile landed
            compute(J1{}, J0{}, b0);
            pp_barrier();
        }
    }
    pp_wait_vm<0>();                         // the tail requests (re-reads of the last K-tile) drained before the epilogue / exit

    // ---- epilogue (shared with the other GEMM kernels) ----
    if (p.splitk > 1) {
        float* slab = p.partial + (long)blockIdx.z * p.M * p.N;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int m = m0 + wr * 128 + i * 16 + l16;
            if (m >= p.M) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = n0 + wc * 64 + j * 16 + 4 * g;
                if (n < p.N) *reinterpret_cast<f32x4*>(slab + (long)m * p.N + n) = acc[i][j];
            }
        }
        return;
    }
    if constexpr (!GEGLU) {     // the one-round-trip form wherever it applies (lb_gemm.h); else the general epilogue below
        const int row_lo = m0 + wr * 128, row_hi = row_lo + 128;
        int batch = -1;
        if (p.rowvec) {
            const int last = (row_hi < p.M ? row_hi : p.M) - 1;
            const int b0 = row_lo / p.rows_per_batch;
            batch = (last >= row_lo && last / p.rows_per_batch == b0) ? b0 : -1;
        }
        const int r0 = row_lo + l16;
        if (lb_gemm_tile_epilogue_lean<8, 4, false, false>(p, acc, [r0](int i) { return r0 + i * 16; }, row_lo, row_hi,
                                                           [l16](int i) { return l16 + i * 16; }, (long)row_lo, batch, n0 + wc * 64))
            return;
    }
    lb_gemm_tile_epilogue<8, 4, GEGLU>(p, acc, m0 + wr * 128 + l16, n0 + wc * 64 + 4 * g, n0 + wc * 32 + 4 * g);
}

// tile order inside an XCD's run: GM block rows down, then the next block column; 0 = the strip order of the other kernels.
// 8 measured best of {0, 2, 4, 8} on the B = 17 shapes and the large squares (profiles/r04_gemm_bench_call6.txt)
static int g_pp_group = 8;
extern "C" void lb_gemm_pp_set_group(int gm) { g_pp_group = gm; }

int lb_gemm_pp_eligible(const LbGemmParams& p) {
    return !p.conv && p.K % PP_BK == 0 && !(p.flags & (LB_GEMM_LN_A | LB_GEMM_CH_STATS)) && p.lda % 8 == 0 && p.ldw % 8 == 0 &&
           ((reinterpret_cast<uintptr_t>(p.A) | reinterpret_cast<uintptr_t>(p.W)) & 15) == 0 &&        // 16-byte LDS-DMA requests
           (long)p.M * p.lda * 2 < (1l << 32) && (long)p.N * p.ldw * 2 < (1l << 32);       // 32-bit row offsets
}

template <bool GEGLU>
static void pp_launch(const LbGemmParams& p, dim3 grid, hipStream_t stream) {
    static unsigned long long seen = 0;
    LB_ONCE_PER_DEVICE(seen)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f16_pp_kernel<GEGLU>), hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS_BYTES);
    hipLaunchKernelGGL((gemm_f16_pp_kernel<GEGLU>), grid, dim3(512), PP_LDS_BYTES, stream, p);
}

int lb_gemm_launch_pp(const LbGemmParams& pin, dim3 grid, hipStream_t stream) {
    LbGemmParams p = pin;
    p.reserved_ = g_pp_group;
    if (p.flags & LB_GEMM_GEGLU) pp_launch<true>(p, grid, stream);
    else pp_launch<false>(p, grid, stream);
    return 0;
}
