// Ping-pong MFMA GEMM for gfx950: 256 x 256 x 64 block tile, 8 waves in two groups that alternate between a
// "load" segment (LDS fragment reads + LDS-DMA requests) and a "compute" segment (16 MFMAs) - while one wave of a SIMD
// multiplies, its partner on the same SIMD feeds itself.  Same contract and epilogues as gemm.hip / gemm_glds.hip (plain
// and GEGLU GEMMs; lb_gemm_f16 routes here, tile code 9), bit-identical results: every accumulator sees the same K order.
//
// Why a third main loop.  The lock-step 8-wave tiles of gemm_glds.hip spend ~1/3 of their wave cycles parked at
// `vmcnt` + `s_barrier` and keep the matrix pipe ~45-50 % busy (profiles/r02_gemm_pmc.json): all waves read LDS together,
// then all multiply together, and a two-stage ring of whole 64 KiB K-tiles has at most one K-tile in flight and must be
// drained at every barrier.  Here
//   * a K-tile is staged as FOUR half-tiles of 16 KiB, cut so that every half-tile is read in exactly ONE phase of the
//     K-tile's four phases (by all waves) and is dead afterwards:
//         A0 = rows {wr*128 + 0..63}, A1 = rows {wr*128 + 64..127}  (wr = 0, 1: the two wave rows)
//         B0 = cols {wc*64 + 0..31},  B1 = cols {wc*64 + 32..63}    (wc = 0..3: the four wave columns)
//     phase 0 multiplies A0 x B0 (reads A0, B0), phase 1 A0 x B1 (reads B1), phase 2 A1 x B1 (reads A1), phase 3 A1 x B0
//     (reads nothing: B0 stayed in registers) - 24 ds_read_b128 per wave per K-tile, the minimum for a 128 x 64 wave tile;
//   * the eight 16 KiB slots of the 128 KiB ring are refilled SIX half-tiles ahead of the consumer (a slot is requested
//     again two phases after its only read), one half-tile per phase, and the only waits are `s_waitcnt vmcnt(8)`: four
//     half-tiles = 64 KiB stay in flight per CU across every barrier (gemm_glds.hip's 256 x 256 tile: <= 64 KiB issued
//     and fully drained once per K-tile);
//   * the two wave groups (wr = 0 / 1: waves w and w + 4 share a SIMD) run one barrier apart, so a SIMD's matrix pipe
//     sees compute segments back to back while LDS reads, DMA issue and waits hide under the partner's MFMAs.
//
// Hazards (MI355X: an LDS-DMA is ordered for a ds_read only by the issuing wave's vmcnt followed by a barrier the reader
// has passed; a slot may be re-requested only after every wave's reads of it have RETURNED):
//   RAW: half-tile h (consumption order A0, B0, B1, A1 per K-tile, consumed in phases 0, 0, 1, 2) is requested in phase
//        h - 6; every wave waits for its own part in the load segment of the phase BEFORE the consuming one, i.e. one
//        full barrier interval before the earliest reader (the other group) starts reading.
//   WAR: the slot of half-tile h held h - 8, read >= 2 phases before the request; the late group's reads of it returned
//        (its MFMAs consumed them) one barrier interval before the early group issues the request.
//
// Replaces (third party, reached from /root/reference/latentblending/diffusers_holder.py:336): the torch.nn.Linear
// layers of the SDXL UNet's transformer blocks at batch >= 8 (q/k/v, attention out, GEGLU, feed-forward out).
#include "lb_common.h"
#include "lb_gemm.h"

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

#define PP_BK 64
#define PP_SLOT_H 8192           // halves per half-tile slot: 128 rows x 64 halves = 16 KiB
#define PP_LDS_BYTES (8 * PP_SLOT_H * 2)

template <int V> struct PPInt { static constexpr int value = V; };
typedef PPInt<0> J0; typedef PPInt<1> J1; typedef PPInt<2> J2; typedef PPInt<3> J3;

template <int N> __device__ __forceinline__ void pp_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void pp_barrier() { asm volatile("s_barrier" ::: "memory"); }

// PRIO: s_setprio 1 around the MFMA clusters (the partner wave's load segment then never delays an MFMA issue)
template <bool GEGLU, bool PRIO>
__global__ void __launch_bounds__(512) gemm_f16_pp_kernel(const LbGemmParams p) {
    constexpr int BM = 256, BN = 256;
    extern __shared__ __attribute__((aligned(16))) f16 lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
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

    const int k_tiles_total = p.K / PP_BK;                       // (launcher: K % 64 == 0)
    const int tiles_per_split = (k_tiles_total + p.splitk - 1) / p.splitk;
    const int kt_begin = blockIdx.z * tiles_per_split;
    int kt_end = kt_begin + tiles_per_split;
    if (kt_end > k_tiles_total) kt_end = k_tiles_total;
    const int nkt = kt_end - kt_begin;

    // ---- staging: a half-tile = 128 LDS rows of 128 B = two wave instructions per wave (rows n*64 + wave*8 + lr) ----
    // lane (lr = lane >> 3, s = lane & 7) owns physical chunk s of its row and fetches logical chunk s ^ (row & 7)
    const int lr = lane >> 3;
    const int cl = (lane & 7) ^ (lr & 7);
    const lb_half* zero = reinterpret_cast<const lb_half*>(p.zero_page);
    const lb_half* a_src[4];         // [qm * 2 + n]: LDS row n*64 + wave*8 + lr of half-tile A_qm = block row n*128 + qm*64 + wave*8 + lr
    const lb_half* w_src[4];         // [qn * 2 + n]
    int a_step[4], w_step[4];        // halves per K-tile (0 for masked rows: they keep reading the zero page)
#pragma unroll
    for (int qm = 0; qm < 2; ++qm)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int m = m0 + n * 128 + qm * 64 + wave * 8 + lr;
            const bool ok = m < p.M;
            a_src[qm * 2 + n] = ok ? p.A + (long)m * p.lda + (long)kt_begin * PP_BK + cl * 8 : zero;
            a_step[qm * 2 + n] = ok ? PP_BK : 0;
        }
#pragma unroll
    for (int qn = 0; qn < 2; ++qn)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            // LDS row r = n*64 + wave*8 + lr of half-tile B_qn belongs to wave column r >> 5, fragment (r >> 4) & 1, row r & 15
            const int wcc = 2 * n + (wave >> 2);
            long wrow;
            bool ok;
            if (GEGLU) {             // fragment 0 = h, 1 = gate of output columns n0 + (wcc*2 + qn)*16 + 0..15
                const int oc = n0 + (wcc * 2 + qn) * 16 + (wave & 1) * 8 + lr;
                ok = oc < n_eff;
                wrow = (long)((wave >> 1) & 1) * n_eff + oc;
            } else {
                const int col = n0 + wcc * 64 + qn * 32 + (wave & 3) * 8 + lr;
                ok = col < p.N;
                wrow = col;
            }
            w_src[qn * 2 + n] = ok ? p.W + wrow * p.ldw + (long)kt_begin * PP_BK + cl * 8 : zero;
            w_step[qn * 2 + n] = ok ? PP_BK : 0;
        }

    // request half-tile J (0 = A0, 1 = B0, 2 = B1, 3 = A1) of the next K-tile of that kind into ring slot 4*d + J
    auto stage = [&](auto jc, int d, bool live) {
        constexpr int J = decltype(jc)::value;
        constexpr bool IS_A = (J == 0 || J == 3);
        constexpr int Q = (J == 0 || J == 1) ? 0 : 1;                 // qm for A, qn for B
        f16* base = lds + (4 * d + J) * PP_SLOT_H + (wave * 8) * PP_BK;
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const lb_half*& ptr = IS_A ? a_src[Q * 2 + n] : w_src[Q * 2 + n];
            const int step = IS_A ? a_step[Q * 2 + n] : w_step[Q * 2 + n];
            const lb_half* src = live ? ptr : zero;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(base + n * 64 * PP_BK), 16, 0, 0);
            ptr += step;
        }
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
    auto read_a = [&](auto jc, int doff) {
        constexpr int J = decltype(jc)::value;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                af[i][ks] = *reinterpret_cast<const f16x8*>(a_rd[ks] + doff + J * PP_SLOT_H + i * 16 * PP_BK);
    };
    auto read_b = [&](auto jc, int doff, f16x8 (&bf)[2][2]) {
        constexpr int J = decltype(jc)::value;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
                bf[jj][ks] = *reinterpret_cast<const f16x8*>(b_rd[ks] + doff + J * PP_SLOT_H + jj * 16 * PP_BK);
    };

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto mma = [&](auto qmc, auto qnc, const f16x8 (&bf)[2][2]) {
        constexpr int QM = decltype(qmc)::value, QN = decltype(qnc)::value;
        if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
                    acc[QM * 4 + i][QN * 2 + jj] =
                        __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[jj][ks], af[i][ks], acc[QM * 4 + i][QN * 2 + jj], 0, 0, 0);
        if (PRIO) __builtin_amdgcn_s_setprio(0);
    };

    // ---- prologue: half-tiles 0..5 (K-tile 0 whole, A0 / B0 of K-tile 1) ----
    stage(J0{}, 0, true);
    stage(J1{}, 0, true);
    stage(J2{}, 0, true);
    stage(J3{}, 0, true);
    stage(J0{}, 1, 1 < nkt);
    stage(J1{}, 1, 1 < nkt);
    pp_wait_vm<8>();                         // A0 / B0 of K-tile 0 have landed (this wave's part)
    pp_barrier();
    if (wr == 1) pp_barrier();               // the second group runs one barrier interval behind the first
    __builtin_amdgcn_sched_barrier(0);

    for (int t = 0; t < nkt; ++t) {
        const int d = t & 1;
        const int doff = d * 4 * PP_SLOT_H;
        const bool live1 = t + 1 < nkt, live2 = t + 2 < nkt;
        // phase 0: A0 x B0
        read_b(J1{}, doff, b0);
        read_a(J0{}, doff);
        stage(J2{}, d ^ 1, live1);
        pp_wait_vm<8>();                     // B1 of this K-tile landed
        pp_barrier();
        __builtin_amdgcn_sched_barrier(0);
        mma(J0{}, J0{}, b0);
        __builtin_amdgcn_sched_barrier(0);
        pp_barrier();
        // phase 1: A0 x B1
        read_b(J2{}, doff, b1);
        stage(J3{}, d ^ 1, live1);
        pp_wait_vm<8>();                     // A1 of this K-tile landed
        pp_barrier();
        __builtin_amdgcn_sched_barrier(0);
        mma(J0{}, J1{}, b1);
        __builtin_amdgcn_sched_barrier(0);
        pp_barrier();
        // phase 2: A1 x B1
        read_a(J3{}, doff);
        stage(J0{}, d, live2);
        pp_barrier();
        __builtin_amdgcn_sched_barrier(0);
        mma(J1{}, J1{}, b1);
        __builtin_amdgcn_sched_barrier(0);
        pp_barrier();
        // phase 3: A1 x B0 (no reads)
        stage(J1{}, d, live2);
        pp_wait_vm<8>();                     // A0 / B0 of the next K-tile landed
        pp_barrier();
        __builtin_amdgcn_sched_barrier(0);
        mma(J1{}, J0{}, b0);
        __builtin_amdgcn_sched_barrier(0);
        pp_barrier();
    }
    if (wr == 0) pp_barrier();               // pairs with the late group's last barrier
    pp_wait_vm<0>();                         // masked tail requests (zero page) drained before the epilogue / exit

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
    lb_gemm_tile_epilogue<8, 4, GEGLU>(p, acc, m0 + wr * 128 + l16, n0 + wc * 64 + 4 * g, n0 + wc * 32 + 4 * g);
}

static int g_pp_prio = 1;
extern "C" void lb_gemm_pp_set_prio(int on) { g_pp_prio = on; }

int lb_gemm_pp_eligible(const LbGemmParams& p) {
    return !p.conv && p.zero_page != nullptr && p.K % PP_BK == 0 && !(p.flags & (LB_GEMM_LN_A | LB_GEMM_CH_STATS)) &&
           p.lda % 8 == 0 && p.ldw % 8 == 0;
}

int lb_gemm_launch_pp(const LbGemmParams& p, dim3 grid, hipStream_t stream) {
    static unsigned long long seen = 0;
    if (lb_first_call_on_device(seen)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f16_pp_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f16_pp_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f16_pp_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f16_pp_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS_BYTES);
    }
    const bool geglu = (p.flags & LB_GEMM_GEGLU) != 0;
    const dim3 block(512);
    if (geglu && g_pp_prio) hipLaunchKernelGGL((gemm_f16_pp_kernel<true, true>), grid, block, PP_LDS_BYTES, stream, p);
    else if (geglu) hipLaunchKernelGGL((gemm_f16_pp_kernel<true, false>), grid, block, PP_LDS_BYTES, stream, p);
    else if (g_pp_prio) hipLaunchKernelGGL((gemm_f16_pp_kernel<false, true>), grid, block, PP_LDS_BYTES, stream, p);
    else hipLaunchKernelGGL((gemm_f16_pp_kernel<false, false>), grid, block, PP_LDS_BYTES, stream, p);
    return 0;
}
