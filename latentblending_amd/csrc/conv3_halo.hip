// 3x3 / stride 1 / pad 1 convolution - and the sub-pixel 2x2 form of the upsampler convs - from an LDS-resident HALO
// tile (gfx950).  On the default route of lb_gemm_f16 since round 2 (lb_gemm_set_halo; 1.3-1.8x the implicit GEMM).
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
// LDS (150 KiB): two halo buffers of HRP = 8*ceil((TH+2)(TW+2)/8) rows (running chunk cc in buffer cc & 1) and a
// 4-slot ring of weight tiles (BN rows).  One STEP = one (chunk c, tap) pair: 32 MFMAs per wave, ONE s_barrier.
//
// Loader roles (round 2, after profiles/r02_halo_study.txt: with every wave issuing both streams the activation reads
// cost 175-230 ns of a 900-1200 ns step and the epilogue 6 us per tile, neither overlapped - vmcnt retires in order, so
// an HBM-latency halo request sat in front of the L2-latency weight requests every wave had to wait for 3 steps later,
// and the first wait after an epilogue drained its stores):
//   * waves 4..7 request the WEIGHT tiles: W(t) three steps ahead, 4 wave instructions per wave and step, and wait
//     with vmcnt(8) (= their requests of steps t-1, t-2) before step t;
//   * waves 0..3 request the HALO of the next chunk (11 rounds of 4 wave instructions during taps 0..5; 10 rounds during
//     taps 0..1 for the 2x2 form) and wait - vmcnt(0) - only before tap 0 of the chunk that reads it: an activation
//     request has up to 9 steps (~6 us) to land instead of 3.
//   Each wave waits for its own requests, then the step's barrier publishes them to the others.
// Persistent blocks (one per CU), request streams running across tile boundaries: see the kernel.  At a tile boundary
// every wave drains its requests BEFORE the epilogue (they are 1-3 steps old), so the first three steps of the next
// tile need no vmcnt wait at all (their operands have landed) and the epilogue's stores drain behind those steps.
// Past-the-end weight requests (the last three steps of a block) are still issued, masked to the zero page, so the
// weight waves' constant holds to the end; halo requests of a tile that does not exist are simply not issued.
#include <type_traits>
#include "lb_common.h"
#include "lb_gemm.h"

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int N> __device__ __forceinline__ void halo_wait_barrier() {
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}
__device__ __forceinline__ void halo_barrier_only() { asm volatile("s_barrier" ::: "memory"); }

// KS = 3: the 3x3 / stride 1 / pad 1 convolution described above (9 taps per 64-channel chunk).
// KS = 2: the SUB-PIXEL form of "nearest-2x upsample, then 3x3 conv" (LbGemmParams.scatter == 2): four 2x2 convolutions
//   on the low-res grid, one per output parity (py, px), with pre-summed weights stacked [4][N][4 Cin].  ALL FOUR
//   parities run in this one launch (the parity is part of the block index: 4x the blocks of one parity launch, one
//   launch instead of four); a block's halo is (TH+1) x (TW+1) pixels starting at (y0 - (1-py), x0 - (1-px)), its
//   4 taps per chunk read it through shifted rows, and its 256 x BN results are scattered to pixels (2y+py, 2x+px).
constexpr unsigned HALO_NONE = 0xffffffffu;             // "no such pixel / row": request nothing (halo) or the zero page (weights)

template <int BN, int TW, int KS = 3>
__global__ void __launch_bounds__(512) conv3x3_halo_kernel(const LbGemmParams p) {
    constexpr int TH = 256 / TW;
    constexpr int NTAP = KS * KS;
    constexpr int HWP = TW + KS - 1;                    // halo width in pixels
    constexpr int HR = (TH + KS - 1) * HWP;             // halo pixels (LDS rows in use)
    constexpr int HRG = (HR + 7) / 8;                   // 8-row groups = wave instructions per halo
    constexpr int HRP = HRG * 8;                        // rows per halo buffer
    constexpr int NRH = (HRG + 3) / 4;                  // rounds of 4 wave instructions (one per halo wave; the last one partial)
    constexpr int RPT = KS == 3 ? 2 : 5;                // halo rounds requested per tap (taps 0..5 / taps 0..1)
    constexpr int WI = BN * 8 / 64 / 4;                 // weight wave instructions per weight wave and step (4)
    constexpr int TM = 4, TN = BN / 32;                 // 16x16 tiles per wave (64 pixels x BN/2 channels)
    static_assert(NRH <= RPT * (KS == 3 ? 6 : 2), "halo rounds do not fit their taps");
    static_assert(WI == 4, "the weight waves' vmcnt(8) assumes four requests per wave and step");
    extern __shared__ __attribute__((aligned(16))) f16 lds[];
    f16* const halo0 = lds;
    f16* const wring = lds + 2 * HRP * 64;              // 4 slots of BN rows

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave >> 1, wave_n = wave & 1;
    const int g = lane >> 4, l16 = lane & 15;
    const bool halo_wave = wave < 4;
    const int lw = wave & 3;                            // index among the waves of the same role

    // ---- work items: (pixel tile [, parity], channel block), channel block fastest; XCD-contiguous block ids ------
    // PERSISTENT form: the grid is G <= #CUs blocks (one per CU: the LDS footprint allows no more) and block `bid`
    // walks items bid, bid + G, bid + 2G, ...  G is a multiple of the channel blocks (x4 parities for KS == 2), so a
    // block keeps ONE weight slab (block_n, parity) for its whole life and only the pixel tile changes.  The request
    // streams do not stop at a tile boundary: during the last chunk of tile k the "next halo" is chunk 0 of tile k+1
    // and W(step + 3) wraps to its first taps, so the epilogue of tile k (global stores, no LDS) runs with the next
    // tile's operands already resident.  With gridDim.x == number of items the same code is the one-item-per-block form.
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
    const int study = p.reserved_;                      // timing studies only (lb_conv_halo_set_study): 1 no epilogue, 2 halo from the zero page, 4 weights from the zero page

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

    // ---- loader state (offsets in 16-byte units: 32 bits reach 64 GiB) ------------------------------------------------
    // a wave instruction moves 8 LDS rows x 128 B: lane (r8 = lane>>3, slot = lane&7) fetches logical chunk slot ^ r8 of row r8
    const int r8 = lane >> 3;
    const int cl = (lane & 7) ^ r8;
    // halo wave: round j moves row group 4 j + lw.  The pixel address is recomputed at every request (a dozen VALU ops
    // in the shadow of 32 MFMAs) instead of held in 11 registers per lane across the MFMA loop.
    const int hrow0 = lw * 8 + r8;
    unsigned w_off[WI];                                 // weight wave: rows (lw + 4 i) * 8 + r8 of the BN x 64 tile
#pragma unroll
    for (int i = 0; i < WI; ++i) {
        const int n = n0 + (lw + 4 * i) * 8 + r8;
        w_off[i] = n < p.N ? (unsigned)(((long)n * p.ldw + cl * 8) >> 3) : HALO_NONE;
    }

    auto issue_halo = [&](int j, const TileAt& t, int src_chunk, int buf, bool tile_exists) {   // one wave instruction of a halo of tile t
        const int gidx = j * 4 + lw;
        if (gidx >= HRG || !tile_exists) return;        // (wave-uniform: no such row group in the partial last round / no such tile)
        const int row = j * 32 + hrow0;
        const int hy = row / HWP, hx = row - hy * HWP;
        const int y = t.y0 + hy - org_y, x = t.x0 + hx - org_x;
        // a row past the halo inside the last group, or a pixel outside the image: zero fill
        const bool ok = y >= 0 && y < p.Hin && x >= 0 && x < p.Win && row < HR && !(study & 2);
        const lb_half* src = ok ? p.A + ((long)(t.b * p.Hin + y) * p.Win + x) * p.ldx + cl * 8 + src_chunk * 64 : zero;
        f16* dst = halo0 + buf * (HRP * 64) + gidx * 8 * 64;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 0, 0);
    };
    auto issue_weight = [&](int i, int chunk, bool exists, int tap, int slot) {
        const bool live = w_off[i] != HALO_NONE && exists && !(study & 4);
        const lb_half* src = live ? Wp + ((long)w_off[i] << 3) + (long)tap * p.Cin + (long)chunk * 64 : zero;
        f16* dst = wring + slot * (BN * 64) + (lw + 4 * i) * 8 * 64;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 0, 0);
    };

    // ---- consumer state -----------------------------------------------------------------------------
    int hbase[TM];                                      // halo row of tap (0, 0) of the lane's pixel i
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = wave_m * 64 + i * 16 + l16;
        const int py = m / TW, px = m - py * TW;
        hbase[i] = py * HWP + px;
    }
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto read_frags = [&](const f16* hb, const f16* wb, int shift, int s, f16x8 (&af)[TM], f16x8 (&wf)[TN]) {
        const int chunk = s * 4 + g;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int r = hbase[i] + shift;
            af[i] = *reinterpret_cast<const f16x8*>(hb + r * 64 + ((chunk ^ (r & 7)) << 3));
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

    // ---- prologue: halo of chunk 0 of the first tile (halo waves), weight tiles of steps 0..2 (weight waves) ----------
    int item = bid;
    TileAt cur = tile_of(item);
    if (halo_wave) {
#pragma unroll
        for (int j = 0; j < NRH; ++j) issue_halo(j, cur, 0, 0, true);
    } else {
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int i = 0; i < WI; ++i) issue_weight(i, 0, true, t, t);
    }

    // One step.  TAP and the loader ROLE (1 = halo wave) are compile-time constants: each role runs its own copy of
    // the step code and all of them meet at the same barriers.  c = chunk within the tile, cc = chunks since the block
    // started (halo buffer / weight slot parity run on across tiles), more = another tile follows, landed = the
    // operands of this step were drained before the previous tile's epilogue (first three steps of a later tile).
    auto step = [&](auto tap_c, auto role_c, int c, int cc, bool more, bool landed, const TileAt& hreq) {
        constexpr int TAP = decltype(tap_c)::value;
        constexpr bool HALO = decltype(role_c)::value != 0;
        if (landed && TAP < 3) {
            halo_barrier_only();
        } else if constexpr (HALO) {
            if constexpr (TAP == 0) halo_wait_barrier<0>(); else halo_barrier_only();
        } else {
            halo_wait_barrier<2 * WI>();
        }
        const int slot = KS == 3 ? ((cc + TAP) & 3) : TAP;   // (NTAP cc + TAP) mod 4
        const f16* hb = halo0 + (cc & 1) * (HRP * 64);
        const f16* wb = wring + slot * (BN * 64);
        constexpr int KY = TAP / KS, KX = TAP % KS;
        // the tap's row shift goes through an opaque register: the swizzled fragment addresses of all taps are
        // loop-invariant, and hoisting them out of the chunk loop costs more registers than the file has
        int shift = KY * HWP + KX;
        asm volatile("" : "+v"(shift));
        // requests of this step: rounds of the NEXT halo scheduled here (chunk c+1 of this tile, or chunk 0 of the next
        // tile during the last chunk: hreq), or W(step + 3) (wrapping into the next tile likewise)
        const bool last = c + 1 == nchunks;
        const int hc = last ? 0 : c + 1, hbuf = (cc + 1) & 1;
        const bool h_exists = !last || more;
        constexpr int TAP3 = (TAP + 3) % NTAP;
        const bool wrap = (TAP + 3 >= NTAP) && last;
        const int c3 = wrap ? 0 : c + (TAP + 3) / NTAP;
        const bool w_exists = !wrap || more;
        const int slot3 = (slot + 3) & 3;
        f16x8 a0[TM], w0[TN], a1[TM], w1[TN];
        read_frags(hb, wb, shift, 0, a0, w0);
        if constexpr (HALO) {
#pragma unroll
            for (int r = 0; r < RPT; ++r)
                if (TAP * RPT + r < NRH) issue_halo(TAP * RPT + r, hreq, hc, hbuf, h_exists);
        }
        __builtin_amdgcn_sched_barrier(0);
        mma_rows(a0, w0, 0, TM / 2);
        read_frags(hb, wb, shift, 1, a1, w1);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!HALO) { issue_weight(0, c3, w_exists, TAP3, slot3); issue_weight(1, c3, w_exists, TAP3, slot3); }
        __builtin_amdgcn_sched_barrier(0);
        mma_rows(a0, w0, TM / 2, TM);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!HALO) { issue_weight(2, c3, w_exists, TAP3, slot3); issue_weight(3, c3, w_exists, TAP3, slot3); }
        __builtin_amdgcn_sched_barrier(0);
        mma_rows(a1, w1, 0, TM);
    };

    LbGemmParams q2;                                    // (KS == 2: this block's parity drives the scatter of the shared epilogue)
    if constexpr (KS == 2) {
        q2 = p;
        q2.scatter = 1;
        q2.sc_py = par_y;
        q2.sc_px = par_x;
    }
    const LbGemmParams& q = KS == 2 ? q2 : p;

    // the tile loop of one role
    auto run = [&](auto role_c) {
        constexpr bool HALO = decltype(role_c)::value != 0;
        int cc = 0;
        bool first = true;
        for (;;) {
            const int next_item = item + G;
            const bool more = next_item < n_items;
            const TileAt nxt = tile_of(more ? next_item : item);
            for (int c = 0; c < nchunks; ++c, ++cc) {
                const TileAt hreq = c + 1 == nchunks ? nxt : cur;       // this tile's halos are all requested in its last chunk: aim at the next tile
                const bool landed = c == 0 && !first;
                step(std::integral_constant<int, 0>{}, role_c, c, cc, more, landed, hreq);
                step(std::integral_constant<int, 1>{}, role_c, c, cc, more, landed, hreq);
                step(std::integral_constant<int, 2>{}, role_c, c, cc, more, landed, hreq);
                step(std::integral_constant<int, 3>{}, role_c, c, cc, more, landed, hreq);
                if constexpr (KS == 3) {
                    step(std::integral_constant<int, 4>{}, role_c, c, cc, more, landed, hreq);
                    step(std::integral_constant<int, 5>{}, role_c, c, cc, more, landed, hreq);
                    step(std::integral_constant<int, 6>{}, role_c, c, cc, more, landed, hreq);
                    step(std::integral_constant<int, 7>{}, role_c, c, cc, more, landed, hreq);
                    step(std::integral_constant<int, 8>{}, role_c, c, cc, more, landed, hreq);
                }
            }
            // every request so far belongs to the next tile's first steps (or is a masked tail request): let it land now,
            // so that those steps can run without a vmcnt wait while this epilogue's stores drain behind them
            __builtin_amdgcn_s_waitcnt(0x0F70);             // vmcnt(0) (as the builtin: the compiler's own wait bookkeeping sees it)
            const int mbase = (cur.b * p.Hin + cur.y0) * p.Win + cur.x0;    // (low-res pixel index; the epilogue scatters it for KS == 2)
            // (the lane's first column goes through an opaque register: bias vectors, column masks and output offsets are
            // tile-invariant, and the compiler would otherwise hoist them out of the tile loop and hold ~30 registers
            // across the MFMA loop - which it then spills around this epilogue)
            int col0 = n0 + wave_n * (BN / 2) + 4 * g;
            asm volatile("" : "+v"(col0));
            auto row_of = [&](int i) {                       // image pixel of the lane's i-th row: from the halo row of its tap (0, 0)
                const int py = hbase[i] / HWP, px = hbase[i] - py * HWP;
                return mbase + py * p.Win + px;
            };
            if (!(study & 1)) lb_gemm_tile_epilogue_rows<TM, TN, false>(q, acc, row_of, col0, 0);
            else if (acc[0][0][0] == 12345.678f) *(float*)p.C = acc[1][1][1] + acc[2][2][2] + acc[3][3][3];   // (keeps the MFMAs alive)
            if (!more) break;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            item = next_item;
            cur = nxt;
            first = false;
        }
    };
    if (halo_wave) run(std::integral_constant<int, 1>{});
    else run(std::integral_constant<int, 0>{});
}

// 1 (default): persistent blocks with the request streams running across tile boundaries; 0: one item per block
static int g_halo_persistent = 1;
static int g_halo_study = 0;
extern "C" void lb_conv_halo_set_persistent(int on) { g_halo_persistent = on; }
// Timing studies (tools/halo_study.py): bit 0 skip the epilogue, bit 1 halo loads from the zero page, bit 2 weight
// loads from the zero page.  Results are then WRONG by construction; 0 (default) = the real kernel.
extern "C" void lb_conv_halo_set_study(int bits) { g_halo_study = bits; }
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

template <int BN, int TW, int KS = 3>
static int launch_halo(const LbGemmParams& p, hipStream_t stream) {
    constexpr int TH = 256 / TW;
    constexpr int HRP = (((TH + KS - 1) * (TW + KS - 1) + 7) / 8) * 8;
    constexpr int SMEM = (2 * HRP * 64 + 4 * BN * 64) * (int)sizeof(f16);
    static bool allowed = false;
    if (!allowed) {                                     // (first call happens at record time, outside any capture)
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_halo_kernel<BN, TW, KS>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
        allowed = true;
    }
    const long tiles = (long)(p.M / (p.Hin * p.Win)) * (p.Hin / TH) * (p.Win / TW) * (KS == 2 ? 4 : 1);
    const int n_blocks = (p.N + BN - 1) / BN;
    const long nblk = tiles * n_blocks;
    LB_REQUIRE(nblk < (1l << 30), "halo conv: too many tiles for one launch");
    // persistent grid: the largest multiple of the weight-slab period (channel blocks, x4 parities) that fits the CUs
    const int period = n_blocks * (KS == 2 ? 4 : 1);
    long grid = nblk;
    if (g_halo_persistent && period <= halo_num_cus() && nblk > halo_num_cus())
        grid = (long)(halo_num_cus() / period) * period;
    LbGemmParams pk = p;
    pk.reserved_ = g_halo_study;
    hipLaunchKernelGGL((conv3x3_halo_kernel<BN, TW, KS>), dim3((unsigned)grid), dim3(512), SMEM, stream, pk);
    return lb_check_launch(KS == 2 ? "lb_upconv2x_halo_f16" : "lb_conv3x3_halo_f16");
}

// Sub-pixel upsampler (scatter == 2): 0 = not eligible, else the tile width
int lb_upconv_halo_eligible(const LbGemmParams& p) {
    if (!p.conv || p.scatter != 2 || p.KH != 2 || p.KW != 2 || p.stride != 1 || p.ups) return 0;
    if (p.Hout != p.Hin || p.Wout != p.Win || p.Cin % 64 != 0 || p.K != 4 * p.Cin) return 0;
    if (p.zero_page == nullptr || (p.flags & (LB_GEMM_GEGLU | LB_GEMM_TRANS_OUT)) || p.N % 4 != 0 || p.residual != nullptr) return 0;
    if (p.M % (p.Hin * p.Win) != 0) return 0;
    if (p.Win % 32 == 0 && p.Hin % 8 == 0) return 32;
    if (p.Win % 16 == 0 && p.Hin % 16 == 0) return 16;
    return 0;
}

int lb_upconv_halo_launch(LbGemmParams p, hipStream_t stream) {
    if (p.alpha == 0.f) p.alpha = 1.f;
    p.splitk = 1;
    return lb_upconv_halo_eligible(p) == 32 ? launch_halo<128, 32, 2>(p, stream) : launch_halo<128, 16, 2>(p, stream);
}

// nearest-2x upsample + 3x3 conv as ONE launch: W = [4][N][ldw] stacked sub-pixel kernels (parity py*2+px), C = [B][2H][2W][ldc]
extern "C" int lb_upconv2x_halo_f16(const LbGemmParams* pp, void* stream) {
    LbGemmParams p = *pp;
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
    if (p.Win % 32 == 0 && p.Hin % 8 == 0) return 32;
    if (p.Win % 16 == 0 && p.Hin % 16 == 0) return 16;
    return 0;
}

long lb_conv3x3_halo_blocks(const LbGemmParams& p) {
    const long tiles = (long)p.M / 256;                 // 256 output pixels per block
    return tiles * ((p.N + 127) / 128);
}

int lb_conv3x3_halo_launch(LbGemmParams p, hipStream_t stream) {
    if (p.alpha == 0.f) p.alpha = 1.f;
    p.splitk = 1;
    return lb_conv3x3_halo_eligible(p) == 32 ? launch_halo<128, 32>(p, stream) : launch_halo<128, 16>(p, stream);
}

extern "C" int lb_conv3x3_halo_f16(const LbGemmParams* pp, void* stream) {
    LbGemmParams p = *pp;
    LB_REQUIRE(p.M > 0 && p.N > 0 && p.K > 0, "lb_conv3x3_halo_f16: empty problem");
    LB_REQUIRE(lb_conv3x3_halo_eligible(p) != 0,
               "lb_conv3x3_halo_f16: needs a 3x3 / stride 1 / pad 1 conv, Cin % 64 == 0, W % 16 == 0, zero page");
    LB_REQUIRE(p.ldw % 8 == 0 && p.ldx % 8 == 0 && (p.ldc % 4 == 0 || (p.flags & LB_GEMM_TRANS_OUT)),
               "lb_conv3x3_halo_f16: ldw / ldx multiples of 8, ldc multiple of 4");
    LB_DISPATCH("lb_conv3x3_halo_f16", lb_conv3x3_halo_launch(p, s));
}
