// 3x3 / stride 1 / pad 1 convolution - and the sub-pixel 2x2 form of the upsampler convs - from an LDS-resident HALO
// tile (gfx950).  On the default route of
// lb_gemm_f16 since round 2 (lb_gemm_set_halo; measured 1.27-1.79x the implicit GEMM, up to 1.18 PFLOP/s).
//
// Why: the implicit-GEMM conv (gemm_glds.hip) stages every input pixel once per tap - 9 times - and the
// ablation (profiles/r01_gemm_ablation.txt) shows the kernel family bound by exactly that global->LDS
// stream.  Staged operand bytes per MFLOP: 256x128 tile 11.4 KB, 256x256 tile 7.6 KB.  Here a block owns
// a 2-D tile of TH x TW = 256 output pixels; per 64-channel chunk of Cin it stages the (TH+2) x (TW+2)
// halo ONCE (43.5 KB) and all 9 taps read it through shifted LDS rows; only the weight tiles (16 KB per
// tap and chunk) stream per step:  5.05 KB / MFLOP.  The loader knows every halo pixel's (y, x), so
// out-of-image pixels are fetched from the zero page and the fragment reads need no masks.
//
// Block: 512 threads = 8 waves (4 along M x 2 along N), wave tile 64 pixels x 64 channels, MFMA
// 16x16x32 f16 with the same operand / LDS conventions as gemm_glds.hip ([rows][64] fp16, 16-B chunks
// XOR-swizzled by (row & 7), swizzle applied to the global source address of the direct-to-LDS loads).
// Local pixel m = wave_m*64 + 16 i + l16  ->  (py, px) = (m / TW, m % TW): the 16 lanes of an MFMA row
// group are 16 consecutive pixels of one image row (conflict-free LDS rows, contiguous NHWC stores).
//
// LDS (150 KiB): two halo buffers of HRP = 8*ceil((TH+2)(TW+2)/8) rows (chunk c in buffer c & 1) and a
// 4-slot ring of weight tiles (BN rows).  One STEP = one (chunk c, tap) pair: 32 MFMAs per wave.
//   * W(t) is requested 3 steps ahead (slot t & 3 was consumed by step t-4... t-1 before the barrier),
//   * the halo of chunk c+1 is requested during taps 0..5 of chunk c (<= 1 load per thread and step),
//   * per step every thread issues [halo load if any] then 2 weight loads, interleaved with the MFMAs.
// Wait before step t (then ONE s_barrier): everything except the loads of steps t-1 and t-2 must have
// landed => s_waitcnt vmcnt(cnt(tap-1) + cnt(tap-2)), cnt(tap) = 2 + [tap <= 4]; the 6th halo load exists
// only in waves 0..(HRG-41) (tap 5) and is simply counted as absent: a wave that has it waits for one
// more (older) load, never for fewer.  Past-the-end requests (last chunk's "next halo", the last three
// steps' weight tiles) are still issued, masked to the zero page, so the constants hold to the end.
//
// Round 2: PERSISTENT blocks (one per CU walks items bid, bid + G, ...) whose request streams run across tile
// boundaries (see the kernel).  profiles/r02_halo_study.txt splits a tile's time: MFMA + LDS alone run at 640-680 ns per
// step (427 ns at the MFMA peak); activation reads add ~2 us per 64-channel chunk and the epilogue 4-6 us per tile.
// Round 3: a third follow-up of the same kind - extra vmcnt slack (+8 / +16) in the first three steps of a tile, so that
// they do not wait for the previous tile's epilogue stores (vmcnt retires in order): bit-identical, -0.2 % on the VAE decode
// batch, nothing on the UNet (profiles/r03_halo_store_slack_ab.txt) - the tile-boundary cost is not the store drain either.
// Two follow-ups were built, verified bit-for-bit and measured, and are NOT in this file (git history has them):
//   * separate loader roles (4 halo waves + 4 weight waves, so that an HBM-latency halo request never sits in front of the
//     L2-latency weight requests in a wave's in-order vmcnt): +3-8 % on the VAE shapes in isolation, -5 % on the UNet's
//     deep-K shapes, no gain inside the programs (profiles/r02_halo_variants.txt) - the per-chunk cost is not the wait
//     order but the CU's request queue holding HBM-latency lines in front of the weight stream;
//   * a full drain before each epilogue so that the next tile's first three steps run without a vmcnt wait while the
//     stores retire: -1..-6 % on the VAE shapes in isolation, +3 % on the deep-K shapes, nothing inside the programs.
#include <type_traits>
#include "lb_common.h"
#include "lb_gemm.h"

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

#ifndef LB_HALO_LEAN_ADDR      // round 6: leaner address arithmetic in the step loop (see the kernel); 0 = the forms of rounds 2-5 (A/B builds)
#define LB_HALO_LEAN_ADDR 1
#endif

template <int N> __device__ __forceinline__ void halo_wait_barrier() {
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}

// KS = 3: the 3x3 / stride 1 / pad 1 convolution described above (9 taps per 64-channel chunk).
// KS = 2: the SUB-PIXEL form of "nearest-2x upsample, then 3x3 conv" (LbGemmParams.scatter == 2): four 2x2 convolutions
//   on the low-res grid, one per output parity (py, px), with pre-summed weights stacked [4][N][4 Cin].  ALL FOUR
//   parities run in this one launch (the parity is part of the block index: 4x the blocks of one parity launch, one
//   launch instead of four); a block's halo is (TH+1) x (TW+1) pixels starting at (y0 - (1-py), x0 - (1-px)), its
//   4 taps per chunk read it through shifted rows, and its 256 x BN results are scattered to pixels (2y+py, 2x+px).
//   The next chunk's halo (5 rounds of wave instructions) is requested during taps 0 and 1 (3 + 2 rounds), because
//   the wait before a step only guarantees everything older than the two previous steps.
template <int KS, int J> struct HaloSched {             // tap during which round J of the NEXT chunk's halo is requested
    static constexpr int tap = KS == 3 ? J : (J < 3 ? 0 : 1);
};
template <int KS, int NR, int TAP> constexpr int halo_full_rounds_in_tap() {   // rounds EVERY wave issues in step TAP
    int n = 0;
    for (int j = 0; j < NR - 1; ++j) n += ((KS == 3 ? j : (j < 3 ? 0 : 1)) == TAP) ? 1 : 0;   // (the last round is partial: counted as absent)
    return n;
}

// Round 4: a PING-PONG form of the step loop (two wave groups one barrier apart, a step = load segment | barrier | 32 MFMAs |
// barrier, as in gemm_pp.hip) was built, verified bit-identical on all shapes and measured 2-13 % SLOWER than this lock-step
// form on every conv shape of the programs (profiles/r04_halo_pp_ab.txt; git history has the code): the step body below
// already spreads its requests and its second fragment read between the MFMA groups, and the two waves of a SIMD drift apart
// on their own.  What a tile pays beyond its MFMA + LDS time is the epilogue and the once-per-chunk activation read.
template <int BN, int TW, int KS = 3>
__global__ void __launch_bounds__(512) conv3x3_halo_kernel(const LbGemmParams p) {
    constexpr int TH = 256 / TW;
    constexpr int NTAP = KS * KS;
    constexpr int HWP = TW + KS - 1;                    // halo width in pixels
    constexpr int HR = (TH + KS - 1) * HWP;             // halo pixels (LDS rows in use)
    constexpr int HRG = (HR + 7) / 8;                   // 8-row groups = wave instructions per halo
    constexpr int HRP = HRG * 8;                        // rows per halo buffer
    constexpr int NR = (HRG + 7) / 8;                   // rounds of 8 wave instructions (the last one partial)
    constexpr int EXTRA = HRG - 8 * (NR - 1);           // groups of the last, partial round (1..8)
    constexpr int WI = BN * 8 / 512;                    // weight loads per thread and step (2)
    constexpr int TM = 4, TN = BN / 32;                 // 16x16 tiles per wave (64 pixels x BN/2 channels)
    static_assert(NR == (KS == 3 ? 6 : 5), "halo rounds: 6 for the 3x3 form, 5 for the 2x2 form");
    static_assert(EXTRA >= 1 && EXTRA <= 8, "the last halo round is a partial one");
    static_assert(WI == 2, "the vmcnt schedule assumes two weight loads per thread and step");
    extern __shared__ __attribute__((aligned(16))) f16 lds[];
    f16* const halo0 = lds;
    f16* const wring = lds + 2 * HRP * 64;              // 4 slots of BN rows

    const int tid = threadIdx.x, lane = tid & 63;
    // (wave-uniform by construction: through readfirstlane, so that everything derived from it - the LDS destination of every direct-to-LDS
    //  request (M0), the `wave < EXTRA` tests - is scalar arithmetic instead of a VALU chain + v_readfirstlane per request)
    const int wave = LB_HALO_LEAN_ADDR ? __builtin_amdgcn_readfirstlane(tid >> 6) : tid >> 6;
    const int wave_m = wave >> 1, wave_n = wave & 1;
    const int g = lane >> 4, l16 = lane & 15;

    // ---- work items: (pixel tile [, parity], channel block), channel block fastest; XCD-contiguous block ids ------
    // PERSISTENT form: the grid is G <= #CUs blocks (one per CU: the LDS footprint allows no more) and block `bid`
    // walks items bid, bid + G, bid + 2G, ...  G is a multiple of the channel blocks (x4 parities for KS == 2), so a
    // block keeps ONE weight slab (block_n, parity) for its whole life and only the pixel tile changes.  The request
    // streams do not stop at a tile boundary: during the last chunk of tile k the "next halo" is chunk 0 of tile k+1
    // and W(step + 3) wraps to its first taps, so the epilogue of tile k (global stores, no LDS) runs with the next
    // tile's operands already in flight - a block of the one-item-per-block form paid launch + first-halo latency +
    // store drain (measured ~14 us, against 11 us of MFMA work for an 18-step Cin = 128 tile) once per tile.
    // With gridDim.x == number of items the same code is the one-item-per-block form.
    const int n_blocks = (p.N + BN - 1) / BN;
    const int tiles_x = p.Win / TW, tiles_y = p.Hin / TH;
    const int n_items = (p.M / (p.Hin * p.Win)) * tiles_x * tiles_y * (KS == 2 ? 4 : 1) * n_blocks;
    const int G = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = G / 8, r = G % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int block_n = bid % n_blocks;
    const int par_x = KS == 2 ? (bid / n_blocks) & 1 : 0;          // output parity of a sub-pixel block
    const int par_y = KS == 2 ? ((bid / n_blocks) >> 1) & 1 : 0;
    const int n0 = block_n * BN;
    const int org_y = KS == 3 ? 1 : 1 - par_y, org_x = KS == 3 ? 1 : 1 - par_x;     // halo row 0 / col 0 = pixel (y0 - org_y, x0 - org_x)
    const lb_half* Wp = p.W + (KS == 2 ? (long)(par_y * 2 + par_x) * p.N * p.ldw : 0);
    const int nchunks = p.Cin / 64;
    const lb_half* zero = reinterpret_cast<const lb_half*>(p.zero_page);

    struct TileAt { int b, y0, x0; };
    auto tile_of = [&](int item) {                      // item -> image and tile origin (block-uniform: scalar registers)
        int tile = item / n_blocks;
        if (KS == 2) tile >>= 2;
        TileAt t;
        t.x0 = (tile % tiles_x) * TW;
        tile /= tiles_x;
        t.y0 = (tile % tiles_y) * TH;
        t.b = tile / tiles_y;
        return t;
    };

    // ---- loader state -------------------------------------------------------------------------------
    // halo: wave instruction j (0..5) covers row group gidx(j); lane (r8 = lane>>3, slot = lane&7) fetches
    // logical chunk slot ^ r8 of halo row gidx*8 + r8
    const int r8 = lane >> 3;
    const int cl = (lane & 7) ^ r8;
    long h_off[NR];                                     // element offset of the pixel (chunk 0) in the tile being REQUESTED, -1 = zero page
    auto set_halo_offsets = [&](const TileAt& t, bool exists) {
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const int gidx = j * 8 + wave;
            const int row = gidx * 8 + r8;
            const int hy = row / HWP, hx = row - hy * HWP;
            const int y = t.y0 + hy - org_y, x = t.x0 + hx - org_x;
            const bool ok = exists && (j < NR - 1 || wave < EXTRA) && row < HR && y >= 0 && y < p.Hin && x >= 0 && x < p.Win;
            h_off[j] = ok ? ((long)(t.b * p.Hin + y) * p.Win + x) * p.ldx + cl * 8 : -1;
        }
    };
    // weights: thread stages rows (tid>>3) + 64 i of the BN x 64 tile.  Round 6 (LB_HALO_LEAN_ADDR): a 32-bit BYTE offset per lane on a
    // wave-uniform base (tap / chunk position: scalar), and no mask - rows past N re-read row N - 1 (their output columns are never
    // stored, and every column's accumulator is its own), requests past the end of the block's work re-read the last tap / chunk (into
    // ring slots nobody reads again) - instead of a 64-bit address + two selects against the zero page per request.
    long w_off[WI];
    unsigned w_off32[WI];
#pragma unroll
    for (int i = 0; i < WI; ++i) {
        const int n = n0 + (tid >> 3) + i * 64;
        w_off[i] = n < p.N ? (long)n * p.ldw + cl * 8 : -1;
        w_off32[i] = (unsigned)(((long)(n < p.N ? n : p.N - 1) * p.ldw + cl * 8) * 2);
    }

#ifdef LB_STUDY_BUILD
    const int study = p.reserved_;                      // timing studies only (lb_conv_halo_set_study): 1 no epilogue, 2 halo from the zero page, 4 weights from the zero page
#else
    constexpr int study = 0;                            // (the study switches exist only in -DLB_STUDY_BUILD libraries)
#endif
    auto issue_halo = [&](int j, int src_chunk, int buf) {   // one wave instruction of a halo (of the tile h_off describes)
        const lb_half* src = h_off[j] >= 0 && !(study & 2) ? p.A + h_off[j] + (long)src_chunk * 64 : zero;
        const int gidx = j * 8 + wave;
        f16* dst = halo0 + buf * (HRP * 64) + gidx * 8 * 64;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 0, 0);
    };
    auto issue_weight = [&](int i, int chunk, bool exists, int tap, int slot) {
        f16* dst = wring + slot * (BN * 64) + (wave * 8 + i * 64) * 64;
        if constexpr (LB_HALO_LEAN_ADDR) {
            if (!(study & 4)) {
                // (exists == false: chunk / tap are those of a request past the end: any staged bytes will do - the address stays inside W
                //  because the caller wraps them to chunk 0 / a tap < NTAP)
                const char* base = reinterpret_cast<const char*>(Wp) + ((long)tap * p.Cin + (long)chunk * 64) * 2;        // wave-uniform
                __builtin_amdgcn_global_load_lds((gptr_t)(base + (size_t)w_off32[i]), (lptr_t)dst, 16, 0, 0);
                return;
            }
        }
        const bool live = w_off[i] >= 0 && exists && !(study & 4);
        const lb_half* src = live ? Wp + w_off[i] + (long)tap * p.Cin + (long)chunk * 64 : zero;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 0, 0);
    };

    // ---- consumer state -----------------------------------------------------------------------------
    int hbase[TM];                                      // halo row of tap (0, 0) of the lane's pixel i
    int mloc[TM];                                       // the pixel's offset from the tile origin, in image pixels
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = wave_m * 64 + i * 16 + l16;
        const int py = m / TW, px = m - py * TW;
        hbase[i] = py * HWP + px;
        mloc[i] = py * p.Win + px;
    }
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto read_frags = [&](const f16* hb, const f16* wb, int shift, int s, f16x8 (&af)[TM], f16x8 (&wf)[TN]) {
        const int chunk = s * 4 + g;
        if constexpr (LB_HALO_LEAN_ADDR && TW == 32) {
            // TW = 32: the lane's pixels i and i + 1 (i even) are 16 pixels apart in ONE image row: halo rows r and r + 16, same
            // swizzle key (r & 7) - one address per pair, the second fragment 16 rows = 2 KiB further on (an instruction offset)
            static_assert(TM == 4, "pairs (0, 1) and (2, 3)");
#pragma unroll
            for (int i = 0; i < TM; i += 2) {
                const int r = hbase[i] + shift;
                const f16* a = hb + r * 64 + ((chunk ^ (r & 7)) << 3);
                af[i] = *reinterpret_cast<const f16x8*>(a);
                af[i + 1] = *reinterpret_cast<const f16x8*>(a + 16 * 64);
            }
        } else {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int r = hbase[i] + shift;
                af[i] = *reinterpret_cast<const f16x8*>(hb + r * 64 + ((chunk ^ (r & 7)) << 3));
            }
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int r = j * 16 + l16;
            wf[j] = *reinterpret_cast<const f16x8*>(wb + (wave_n * (BN / 2) + r) * 64 + ((chunk ^ (r & 7)) << 3));
        }
    };
    auto mma_rows = [&](const f16x8 (&af)[TM], const f16x8 (&wf)[TN], int i_lo, int i_hi) {
#pragma unroll
        for (int i = i_lo; i < i_hi; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[j], af[i], acc[i][j], 0, 0, 0);
    };

    // ---- prologue: halo of chunk 0 of the first tile, weight tiles of steps 0..2 ------------------------
    int item = bid;
    TileAt cur = tile_of(item);
    set_halo_offsets(cur, true);
#pragma unroll
    for (int j = 0; j < NR; ++j)
        if (j < NR - 1 || wave < EXTRA) issue_halo(j, 0, 0);
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int i = 0; i < WI; ++i) issue_weight(i, 0, true, t, t);

    // one step; TAP is a compile-time constant so that every count below is an immediate.  c = chunk within the tile,
    // cc = chunks since the block started (halo buffer / weight slot parity run on across tiles), more = another tile follows
    auto step = [&](auto tap_c, int c, int cc, bool more) {
        constexpr int TAP = decltype(tap_c)::value;
        constexpr int P1 = (TAP + NTAP - 1) % NTAP, P2 = (TAP + NTAP - 2) % NTAP;          // taps of the two previous steps
        constexpr int CNT = (2 + halo_full_rounds_in_tap<KS, NR, P1>()) + (2 + halo_full_rounds_in_tap<KS, NR, P2>());
        halo_wait_barrier<CNT>();
        const int slot = KS == 3 ? ((cc + TAP) & 3) : TAP;   // (NTAP cc + TAP) mod 4
        const f16* hb = halo0 + (cc & 1) * (HRP * 64);
        const f16* wb = wring + slot * (BN * 64);
        constexpr int KY = TAP / KS, KX = TAP % KS;
        // the tap's row shift goes through an opaque register: the swizzled fragment addresses of all taps are
        // loop-invariant, and hoisting them out of the chunk loop costs more registers than the file has
        int shift = KY * HWP + KX;
        asm volatile("" : "+v"(shift));
        // requests of this step: rounds of the NEXT halo scheduled here (chunk c+1, or chunk 0 of the next tile: h_off
        // was re-pointed at the start of the tile's last chunk), then W(step + 3) (wrapping into the next tile likewise)
        const bool last = c + 1 == nchunks;
        const int hc = last ? 0 : c + 1, hbuf = (cc + 1) & 1;
        constexpr int TAP3 = (TAP + 3) % NTAP;
        const bool wrap = (TAP + 3 >= NTAP) && last;
        const int c3 = wrap ? 0 : c + (TAP + 3) / NTAP;
        const bool w_exists = !wrap || more;
        const int slot3 = (slot + 3) & 3;
        f16x8 a0[TM], w0[TN], a1[TM], w1[TN];
        read_frags(hb, wb, shift, 0, a0, w0);
        if constexpr (HaloSched<KS, 0>::tap == TAP) issue_halo(0, hc, hbuf);
        if constexpr (NR > 2 && HaloSched<KS, 1>::tap == TAP) issue_halo(1, hc, hbuf);
        if constexpr (NR > 3 && HaloSched<KS, 2>::tap == TAP) issue_halo(2, hc, hbuf);
        if constexpr (NR > 4 && HaloSched<KS, 3>::tap == TAP) { if (NR - 1 > 3 || wave < EXTRA) issue_halo(3, hc, hbuf); }
        if constexpr (NR > 4 && HaloSched<KS, 4>::tap == TAP) { if (NR - 1 > 4 || wave < EXTRA) issue_halo(4, hc, hbuf); }
        if constexpr (NR > 5 && HaloSched<KS, 5>::tap == TAP) { if (wave < EXTRA) issue_halo(5, hc, hbuf); }
        __builtin_amdgcn_sched_barrier(0);
        mma_rows(a0, w0, 0, TM / 2);
        read_frags(hb, wb, shift, 1, a1, w1);
        __builtin_amdgcn_sched_barrier(0);
        issue_weight(0, c3, w_exists, TAP3, slot3);
        __builtin_amdgcn_sched_barrier(0);
        mma_rows(a0, w0, TM / 2, TM);
        __builtin_amdgcn_sched_barrier(0);
        issue_weight(1, c3, w_exists, TAP3, slot3);
        __builtin_amdgcn_sched_barrier(0);
        mma_rows(a1, w1, 0, TM);
    };

    LbGemmParams q = p;                                 // (KS == 2: this block's parity drives the scatter of the shared epilogue)
    if constexpr (KS == 2) {
        q.scatter = 1;
        q.sc_py = par_y;
        q.sc_px = par_x;
    }

    int cc = 0;
    for (;;) {
        const int next_item = item + G;
        const bool more = next_item < n_items;
        const TileAt nxt = tile_of(more ? next_item : item);
        for (int c = 0; c < nchunks; ++c, ++cc) {
            if (c + 1 == nchunks) set_halo_offsets(nxt, more);      // this tile's halos are all requested: aim at the next tile
            step(std::integral_constant<int, 0>{}, c, cc, more);
            step(std::integral_constant<int, 1>{}, c, cc, more);
            step(std::integral_constant<int, 2>{}, c, cc, more);
            step(std::integral_constant<int, 3>{}, c, cc, more);
            if constexpr (KS == 3) {
                step(std::integral_constant<int, 4>{}, c, cc, more);
                step(std::integral_constant<int, 5>{}, c, cc, more);
                step(std::integral_constant<int, 6>{}, c, cc, more);
                step(std::integral_constant<int, 7>{}, c, cc, more);
                step(std::integral_constant<int, 8>{}, c, cc, more);
            }
        }
        // epilogue of this tile (registers -> global memory; the next tile's first operands are in flight meanwhile)
        const int mbase = (cur.b * p.Hin + cur.y0) * p.Win + cur.x0;    // (low-res pixel index; the epilogue scatters it for KS == 2)
        // (the lane's first column goes through an opaque register: bias vectors, column masks and output offsets are
        // tile-invariant, and the compiler would otherwise hoist them out of the tile loop and hold ~30 registers
        // across the MFMA loop - which it then spills around this epilogue)
        int col0 = n0 + wave_n * (BN / 2) + 4 * g;
        asm volatile("" : "+v"(col0));
        if (!(study & 1)) {
            // geometry of the one-round-trip epilogue (lb_gemm.h): the tile's rows all exist and lie inside ONE image; output
            // rows relative to the tile's first output pixel (for KS == 2 on the 2x grid, at this block's parity) are
            // recomputed from the lane's local pixel index - compile-time shifts, nothing held across the MFMA loop
            int tid_o = tid;                            // (opaque: nothing derived from it is hoisted out of the tile loop)
            asm volatile("" : "+v"(tid_o));
            auto out_rel = [&](int i) {
                if constexpr (KS == 3) return mloc[i];
                const int ml = (tid_o >> 7) * 64 + i * 16 + (tid_o & 15), ly = ml / TW, lx = ml - ly * TW;
                return 4 * ly * p.Win + 2 * lx;
            };
            const long out_lo = KS == 3 ? (long)mbase
                                        : ((long)(cur.b * 2 * p.Hin + 2 * cur.y0 + par_y)) * (2 * p.Win) + 2 * cur.x0 + par_x;
            const int row_hi = mbase + (TH - 1) * p.Win + TW;
            const int colw = col0 - 4 * ((tid_o & 63) >> 4);
            if (p.flags & LB_GEMM_CH_STATS) {   // (wave-uniform) GroupNorm statistics of the stored tile: row block (item / n_blocks) * 4 + wave_m
                const long stat_rows = (long)(n_items / n_blocks) * 4;          // row blocks of the whole launch
                float2* chst = reinterpret_cast<float2*>(p.ch_stats) + ((long)(item / n_blocks) * 4 + wave_m);
                if (!lb_gemm_tile_epilogue_lean<TM, TN, true, true>(q, acc, [&](int i) { return mbase + mloc[i]; }, mbase, row_hi, out_rel,
                                                                    out_lo, cur.b, colw, chst, stat_rows))
                    lb_gemm_tile_epilogue_rows_ln<TM, TN, false, false, true>(q, acc, [&](int i) { return mbase + mloc[i]; }, col0, 0,
                                                                              (const LbLnRows<TM>*)nullptr, chst, stat_rows);
            } else {
                if (!lb_gemm_tile_epilogue_lean<TM, TN, false, true>(q, acc, [&](int i) { return mbase + mloc[i]; }, mbase, row_hi, out_rel,
                                                                     out_lo, cur.b, colw))
                    lb_gemm_tile_epilogue_rows<TM, TN, false>(q, acc, [&](int i) { return mbase + mloc[i]; }, col0, 0);
            }
        }
        else if (acc[0][0][0] == 12345.678f) *(float*)p.C = acc[1][1][1] + acc[2][2][2] + acc[3][3][3];   // (keeps the MFMAs alive)
        if (!more) break;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        item = next_item;
        cur = nxt;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the masked tail requests still target this block's LDS: drain before exit
}

// 1 (default): persistent blocks with the request streams running across tile boundaries; 0: one item per block
static int g_halo_persistent = 1;
extern "C" void lb_conv_halo_set_persistent(int on) { g_halo_persistent = on; }
#ifdef LB_STUDY_BUILD
// Timing studies (tools/halo_study.py, study builds only): bit 0 skip the epilogue, bit 1 halo loads from the zero page,
// bit 2 weight loads from the zero page.  Results are then WRONG by construction; 0 (default) = the real kernel.
static int g_halo_study = 0;
extern "C" void lb_conv_halo_set_study(int bits) { g_halo_study = bits; }
#endif
static int halo_num_cus() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        hipDeviceProp_t prop;
        n = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                ? prop.multiProcessorCount : 256;
    }
    return n;
}

// work items of a launch and the grid that walks them: persistent = the largest multiple of the weight-slab period
// (channel blocks, x4 parities for the 2x2 form) that fits the CUs, so that a block keeps one (channel block, parity)
static void halo_grid(const LbGemmParams& p, int tw, int ks, long& items, long& grid) {
    const int th = 256 / tw;
    const long tiles = (long)(p.M / (p.Hin * p.Win)) * (p.Hin / th) * (p.Win / tw) * (ks == 2 ? 4 : 1);
    const int n_blocks = (p.N + 127) / 128;
    items = tiles * n_blocks;
    const int period = n_blocks * (ks == 2 ? 4 : 1);
    grid = items;
    if (g_halo_persistent && period <= halo_num_cus() && items > halo_num_cus())
        grid = (long)(halo_num_cus() / period) * period;
}

// LB_GEMM_CH_STATS: row blocks of the whole launch = (items / channel blocks) * 4 wave rows - the host twin of `stat_rows` in
// the kernel's epilogue (0 = not a halo launch)
int lb_upconv_halo_eligible(const LbGemmParams& p);
int lb_conv3x3_halo_eligible(const LbGemmParams& p);
long lb_conv_halo_stat_rows_total(const LbGemmParams& p) {
    int tw = 0, ks = 0;
    if ((tw = lb_upconv_halo_eligible(p)) != 0) ks = 2;
    else if ((tw = lb_conv3x3_halo_eligible(p)) != 0) ks = 3;
    if (!ks) return 0;
    long items, grid;
    halo_grid(p, tw, ks, items, grid);
    return items / ((p.N + 127) / 128) * 4;
}

template <int BN, int TW, int KS = 3>
static int launch_halo(const LbGemmParams& p, hipStream_t stream) {
    constexpr int TH = 256 / TW;
    constexpr int HRP = (((TH + KS - 1) * (TW + KS - 1) + 7) / 8) * 8;
    constexpr int SMEM = (2 * HRP * 64 + 4 * BN * 64) * (int)sizeof(f16);
    static unsigned long long seen = 0;
    LB_ONCE_PER_DEVICE(seen)                  // (first call on a device happens at record time, outside any capture)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_halo_kernel<BN, TW, KS>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    static_assert(BN == 128, "halo_grid assumes 128-channel blocks");
    long nblk, grid;
    halo_grid(p, TW, KS, nblk, grid);
    LB_REQUIRE(nblk < (1l << 30), "halo conv: too many tiles for one launch");
    LbGemmParams pk = p;
#ifdef LB_STUDY_BUILD
    pk.reserved_ = g_halo_study;
#endif
    hipLaunchKernelGGL((conv3x3_halo_kernel<BN, TW, KS>), dim3((unsigned)grid), dim3(512), SMEM, stream, pk);
    return lb_check_launch(KS == 2 ? "lb_upconv2x_halo_f16" : "lb_conv3x3_halo_f16");
}

// Sub-pixel upsampler (scatter == 2): 0 = not eligible, else the tile width
int lb_upconv_halo_eligible(const LbGemmParams& p) {
    if (!p.conv || p.scatter != 2 || p.KH != 2 || p.KW != 2 || p.stride != 1 || p.ups) return 0;
    if (p.Hout != p.Hin || p.Wout != p.Win || p.Cin % 64 != 0 || p.K != 4 * p.Cin) return 0;
    if (p.zero_page == nullptr || (p.flags & (LB_GEMM_GEGLU | LB_GEMM_TRANS_OUT)) || p.N % 4 != 0 || p.residual != nullptr) return 0;
    if (p.M % (p.Hin * p.Win) != 0) return 0;
    if ((long)p.N * p.ldw * 2 >= (1l << 32)) return 0;         // (32-bit byte offsets into one parity's weight slab: LB_HALO_LEAN_ADDR)
    if (p.Win % 32 == 0 && p.Hin % 8 == 0) return 32;
    if (p.Win % 16 == 0 && p.Hin % 16 == 0) return 16;
    return 0;
}

int lb_upconv_halo_launch(LbGemmParams p, hipStream_t stream) {
    LB_REQUIRE(!(p.flags & LB_GEMM_CH_STATS) || (p.ch_stats != nullptr && p.ch_stats_rows == lb_conv_halo_stat_rows_total(p)),
               "halo upconv: LB_GEMM_CH_STATS needs ch_stats with ch_stats_rows = B * lb_gemm_ch_stat_rows()");
    if (p.alpha == 0.f) p.alpha = 1.f;
    p.splitk = 1;
    return lb_upconv_halo_eligible(p) == 32 ? launch_halo<128, 32, 2>(p, stream) : launch_halo<128, 16, 2>(p, stream);
}

// nearest-2x upsample + 3x3 conv as ONE launch: W = [4][N][ldw] stacked sub-pixel kernels (parity py*2+px), C = [B][2H][2W][ldc]
extern int g_lb_wide_store, g_lb_lean_epilogue;
extern "C" int lb_upconv2x_halo_f16(const LbGemmParams* pp, void* stream) {
    LbGemmParams p = *pp;
    p.reserved2_ = (g_lb_wide_store & 1) | (g_lb_lean_epilogue ? 2 : 0);
    LB_REQUIRE(p.M > 0 && p.N > 0 && p.K > 0, "lb_upconv2x_halo_f16: empty problem");
    LB_REQUIRE(lb_upconv_halo_eligible(p) != 0,
               "lb_upconv2x_halo_f16: needs scatter = 2, KH = KW = 2, stride 1, Cin % 64 == 0, W % 16 == 0, zero page, no residual");
    LB_REQUIRE(p.ldw % 8 == 0 && p.ldx % 8 == 0 && p.ldc % 4 == 0, "lb_upconv2x_halo_f16: ldw / ldx multiples of 8, ldc multiple of 4");
    LB_DISPATCH("lb_upconv2x_halo_f16", lb_upconv_halo_launch(p, s));
}

// 0 = not eligible, else the tile width the halo kernel would use
int lb_conv3x3_halo_eligible(const LbGemmParams& p) {
    if (!p.conv || p.KH != 3 || p.KW != 3 || p.stride != 1 || p.pad != 1 || p.ups || p.scatter) return 0;
    if (p.Hout != p.Hin || p.Wout != p.Win || p.Cin % 64 != 0 || p.K != 9 * p.Cin) return 0;
    if (p.zero_page == nullptr || (p.flags & LB_GEMM_GEGLU) || p.N % 4 != 0) return 0;
    if (p.M % (p.Hin * p.Win) != 0) return 0;
    if ((long)p.N * p.ldw * 2 >= (1l << 32)) return 0;         // (32-bit byte offsets into the weight matrix: LB_HALO_LEAN_ADDR)
    if (p.Win % 32 == 0 && p.Hin % 8 == 0) return 32;
    if (p.Win % 16 == 0 && p.Hin % 16 == 0) return 16;
    return 0;
}

long lb_conv3x3_halo_blocks(const LbGemmParams& p) {
    const long tiles = (long)p.M / 256;                 // 256 output pixels per block
    return tiles * ((p.N + 127) / 128);
}

int lb_conv3x3_halo_launch(LbGemmParams p, hipStream_t stream) {
    LB_REQUIRE(!(p.flags & LB_GEMM_CH_STATS) || (p.ch_stats != nullptr && !(p.flags & LB_GEMM_TRANS_OUT)),
               "halo conv: LB_GEMM_CH_STATS needs ch_stats and a row-major output");
    LB_REQUIRE(!(p.flags & LB_GEMM_CH_STATS) || p.ch_stats_rows == lb_conv_halo_stat_rows_total(p),
               "halo conv: ch_stats_rows must be B * lb_gemm_ch_stat_rows() (the row blocks this launch writes per channel)");
    if (p.alpha == 0.f) p.alpha = 1.f;
    p.splitk = 1;
    return lb_conv3x3_halo_eligible(p) == 32 ? launch_halo<128, 32>(p, stream) : launch_halo<128, 16>(p, stream);
}

extern "C" int lb_conv3x3_halo_f16(const LbGemmParams* pp, void* stream) {
    LbGemmParams p = *pp;
    p.reserved2_ = (g_lb_wide_store & 1) | (g_lb_lean_epilogue ? 2 : 0);
    LB_REQUIRE(p.M > 0 && p.N > 0 && p.K > 0, "lb_conv3x3_halo_f16: empty problem");
    LB_REQUIRE(lb_conv3x3_halo_eligible(p) != 0,
               "lb_conv3x3_halo_f16: needs a 3x3 / stride 1 / pad 1 conv, Cin % 64 == 0, W % 16 == 0, zero page");
    LB_REQUIRE(p.ldw % 8 == 0 && p.ldx % 8 == 0 && (p.ldc % 4 == 0 || (p.flags & LB_GEMM_TRANS_OUT)),
               "lb_conv3x3_halo_f16: ldw / ldx multiples of 8, ldc multiple of 4");
    LB_DISPATCH("lb_conv3x3_halo_f16", lb_conv3x3_halo_launch(p, s));
}

// Host arithmetic of a halo launch (no device work): kind = 0 (not a halo launch), 3 (3x3 form) or 2 (2x2 sub-pixel form),
// tile width, number of (tile [, parity], channel block) work items and the grid the launcher would use.
extern "C" void lb_conv_halo_plan(const LbGemmParams* pp, int* kind, int* tile_w, long* items, long* grid) {
    LbGemmParams p = *pp;
    int k = 0, tw = 0;
    if ((tw = lb_upconv_halo_eligible(p)) != 0) k = 2;
    else if ((tw = lb_conv3x3_halo_eligible(p)) != 0) k = 3;
    long it = 0, gr = 0;
    if (k) halo_grid(p, tw, k, it, gr);
    if (kind) *kind = k;
    if (tile_w) *tile_w = tw;
    if (items) *items = it;
    if (grid) *grid = gr;
}
