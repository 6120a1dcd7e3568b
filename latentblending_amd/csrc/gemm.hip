// MFMA GEMM / implicit-GEMM convolution for gfx950 (fp16 operands, fp32 accumulate).
//
//   C[M,N] = epilogue( A[M,K] . W[N,K]^T )
//
// One kernel family serves every dense contraction of the SDXL UNet / VAE decoder / LPIPS net:
//   * Linear layers (A = tokens x channels, NHWC activations are already [B*H*W, C]),
//   * 3x3 / kxk convolutions as implicit GEMM: the A-loader gathers NHWC pixels on the fly
//     (K index = (ky*KW + kx)*Cin + c), with stride, zero padding and an optional fused
//     nearest-2x upsample of the input (UNet/VAE upsamplers),
//   * GEGLU (the block computes matching h / gate column groups and stores h*gelu(gate)),
//   * epilogues: alpha, bias[n], per-sample row vector (time-embedding add), residual add
//     (fp16 or fp32), fp16 / fp32 / transposed stores, split-K partial slabs.
//
// Tiling (wave64, 4 waves as 2x2): BMxBNx64 block tile, v_mfma_f32_16x16x32_f16, operands
// swapped (a = W fragment, b = A fragment) so that each lane owns 4 CONSECUTIVE output columns
// of one row -> 8-byte stores and vector bias/residual loads.  LDS tiles are [rows][64] halves
// (128-B rows) with the 16-B chunk index XOR-swizzled by (row & 7) so ds_read_b128 fragment
// reads spread over the 64 banks.
//
// Memory pipeline: a D-deep REGISTER ring of K-tiles.  Tile t+D is requested from HBM/L2 while
// tile t is multiplied out of LDS and tile t+1 is moved from registers to the other LDS buffer
// (one barrier per K-tile).  With the small tiles the UNet needs at batch 1-8 a K-tile's MFMAs
// take ~200-500 cycles but a global load ~1-2k cycles, so one tile in flight (D=1) leaves the
// loop latency-bound; D = 3-4 keeps enough bytes in flight per CU.  All loads are branch-free:
// out-of-range chunks read a valid dummy address and are zeroed when they are written to LDS.
//
// Replaces (third party, reached from /root/reference/latentblending/diffusers_holder.py:336
// and :135): every torch.nn.Linear / Conv2d inside diffusers' UNet2DConditionModel and
// AutoencoderKL decoder, and the AlexNet convolutions of lpips (blending_engine.py:756).
#include "lb_common.h"
#include "lb_gemm.h"

#define BK 64

template <int N> struct Int { static constexpr int value = N; };

template <int BM, int BN, bool CONV, bool GEGLU, int D>
__global__ void __launch_bounds__(256) gemm_f16_kernel(const LbGemmParams p) {
    constexpr int AI = BM * 8 / 256;        // 16-B chunks of A per thread per K-tile
    constexpr int WI = BN * 8 / 256;
    constexpr int TM = BM / 32;             // 16-row fragments per wave along M
    constexpr int TN = BN / 32;
    __shared__ __attribute__((aligned(16))) f16 lds[2 * (BM + BN) * BK];
    f16* const As = lds;
    f16* const Ws = lds + 2 * BM * BK;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wave_m = wave >> 1, wave_n = wave & 1;
    const int g = lane >> 4, l16 = lane & 15;

    const int n_blocks = GEGLU ? (p.N / 2 + BN / 2 - 1) / (BN / 2) : (p.N + BN - 1) / BN;
    // XCD-aware remap: consecutive block ids land on different XCDs (id % 8); give each XCD a
    // contiguous run of tiles so that neighbours sharing A rows / W panels hit the same L2.
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x;
        const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    // Tile order inside an XCD's contiguous run: the operand with MORE bytes must be the one that is
    // shared.  Weight-dominated problems (N > M: every UNet GEMM at batch 1-8) walk the M blocks of
    // a weight panel first, so each XCD streams only its own few weight panels from HBM (with the
    // other order every XCD's L2 pulls the whole weight matrix: up to 8x redundant HBM reads);
    // activation-dominated problems (VAE convs, M >> N) keep the A panel resident instead.
    const int m_blocks = (p.M + BM - 1) / BM;
    const bool w_dominant = (GEGLU ? p.N / 2 : p.N) > p.M;
    const int block_n = w_dominant ? bid / m_blocks : bid % n_blocks;
    const int block_m = w_dominant ? bid % m_blocks : bid / n_blocks;
    const int m0 = block_m * BM;
    const int n0 = GEGLU ? block_n * (BN / 2) : block_n * BN;

    // split-K range of this block
    const int k_tiles_total = (p.K + BK - 1) / BK;
    const int tiles_per_split = (k_tiles_total + p.splitk - 1) / p.splitk;
    const int kt_begin = blockIdx.z * tiles_per_split;
    int kt_end = kt_begin + tiles_per_split;
    if (kt_end > k_tiles_total) kt_end = k_tiles_total;
    const int nkt = kt_end - kt_begin;
    const int k_end = kt_end * BK < p.K ? kt_end * BK : p.K;   // loads at k >= k_end are masked to zero

    const int slot = tid & 7;               // logical 16-B chunk (8 halves) within the K-tile
    const int row0 = tid >> 3;              // first tile row this thread stages (then +32)

    // ---- per-thread A source description --------------------------------------------------
    long a_off[AI];                         // plain: row offset; conv: batch base offset
    int a_iy[AI], a_ix[AI];
    bool a_ok[AI];
    int ci = 0, ky = 0, kx = 0;             // conv: decomposition of this thread's k index
    int kcur = kt_begin * BK + slot * 8;
#pragma unroll
    for (int i = 0; i < AI; ++i) {
        const int m = m0 + row0 + i * 32;
        a_ok[i] = m < p.M;
        if (CONV) {
            const int hw = p.Hout * p.Wout;
            const int b = m / hw, rem = m - b * hw;
            const int oy = rem / p.Wout, ox = rem - oy * p.Wout;
            a_off[i] = a_ok[i] ? (long)b * p.Hin * p.Win * p.ldx : 0;
            a_iy[i] = oy * p.stride - (p.scatter ? 1 - p.sc_py : p.pad);
            a_ix[i] = ox * p.stride - (p.scatter ? 1 - p.sc_px : p.pad);
        } else {
            a_off[i] = a_ok[i] ? (long)m * p.lda : 0;
            a_iy[i] = a_ix[i] = 0;
        }
    }
    if (CONV) {
        const int tap = kcur / p.Cin;
        ci = kcur - tap * p.Cin;
        ky = tap / p.KW;
        kx = tap - ky * p.KW;
    }
    // ---- per-thread W source rows ---------------------------------------------------------
    long w_off[WI];
    bool w_ok[WI];
#pragma unroll
    for (int i = 0; i < WI; ++i) {
        const int tr = row0 + i * 32;
        int n;
        if (GEGLU) {
            const int sub = tr >> 4;
            n = (sub & 1) * (p.N / 2) + n0 + (sub >> 1) * 16 + (tr & 15);
            w_ok[i] = (n0 + (sub >> 1) * 16 + (tr & 15)) < p.N / 2;
        } else {
            n = n0 + tr;
            w_ok[i] = n < p.N;
        }
        w_off[i] = w_ok[i] ? (long)n * p.ldw : 0;
    }

    f16x8 a_reg[D][AI], w_reg[D][WI];
    unsigned valid[D];                      // bit i: A chunk i real, bit 8+i: W chunk i real
    const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    const int hin_eff = p.Hin << p.ups, win_eff = p.Win << p.ups;

    // Branch-free tile request into ring slot S (addresses of masked chunks fall back to offset 0
    // of the same operand, which is always mapped); advances this thread's k state by one K-tile.
    auto load_tile = [&](auto sc) {
        constexpr int S = decltype(sc)::value;
        const bool k_ok = kcur < k_end;
        const int kk = k_ok ? kcur : 0;
        unsigned v = 0;
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            if (CONV) {
                const int iy = a_iy[i] + ky, ix = a_ix[i] + kx;
                const bool ok = a_ok[i] && k_ok && iy >= 0 && iy < hin_eff && ix >= 0 && ix < win_eff;
                const long off = ok ? a_off[i] + ((long)(iy >> p.ups) * p.Win + (ix >> p.ups)) * p.ldx + ci : 0;
                a_reg[S][i] = *reinterpret_cast<const f16x8*>(p.A + off);
                v |= ok ? (1u << i) : 0u;
            } else {
                const bool ok = a_ok[i] && k_ok;
                a_reg[S][i] = *reinterpret_cast<const f16x8*>(p.A + a_off[i] + (a_ok[i] ? kk : 0));
                v |= ok ? (1u << i) : 0u;
            }
        }
#pragma unroll
        for (int i = 0; i < WI; ++i) {
            const bool ok = w_ok[i] && k_ok;
            w_reg[S][i] = *reinterpret_cast<const f16x8*>(p.W + w_off[i] + (w_ok[i] ? kk : 0));
            v |= ok ? (1u << (8 + i)) : 0u;
        }
        valid[S] = v;
        kcur += BK;
        if (CONV) {
            ci += BK;
            if (p.Cin >= BK) {              // at most one tap boundary per K-tile: branch-free wrap
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

    auto store_tile = [&](auto sc, int buf) {
        constexpr int S = decltype(sc)::value;
        const unsigned v = valid[S];
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            const int r = row0 + i * 32;
            *reinterpret_cast<f16x8*>(As + buf * BM * BK + r * BK + ((slot ^ (r & 7)) << 3)) =
                (v >> i) & 1u ? a_reg[S][i] : zero8;
        }
#pragma unroll
        for (int i = 0; i < WI; ++i) {
            const int r = row0 + i * 32;
            *reinterpret_cast<f16x8*>(Ws + buf * BN * BK + r * BK + ((slot ^ (r & 7)) << 3)) =
                (v >> (8 + i)) & 1u ? w_reg[S][i] : zero8;
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto compute = [&](int buf) {
        const f16* Ab = As + buf * BM * BK + (wave_m * (BM / 2)) * BK;
        const f16* Wb = Ws + buf * BN * BK + (wave_n * (BN / 2)) * BK;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            f16x8 af[TM], wf[TN];
            const int chunk = s * 4 + g;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int r = i * 16 + l16;
                af[i] = *reinterpret_cast<const f16x8*>(Ab + r * BK + ((chunk ^ (r & 7)) << 3));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int r = j * 16 + l16;
                wf[j] = *reinterpret_cast<const f16x8*>(Wb + r * BK + ((chunk ^ (r & 7)) << 3));
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[j], af[i], acc[i][j], 0, 0, 0);
        }
    };

    // One K-tile, NO branches: tile t sits in LDS buf (t&1) and ring slot S is free (its tile is
    // the one in LDS).  Requests past the end of this block's K range are masked (zero tiles), so
    // the steady-state loop is straight-line code and the compiler can keep counted vmcnt waits
    // (a conditional around the loads makes it fall back to vmcnt(0) = one tile in flight).
    auto stage = [&](auto sc, int t) {
        constexpr int S = decltype(sc)::value;
        load_tile(Int<S>{});                                        // request tile t+D
        compute(t & 1);
        store_tile(Int<(S + 1) % D>{}, (t + 1) & 1);                // tile t+1: registers -> LDS
        __syncthreads();
    };

    // ---- prologue: tiles 0..D-1 in flight, tile 0 into LDS ---------------------------------
    load_tile(Int<0>{});
    if (D > 1) load_tile(Int<1 % D>{});
    if (D > 2) load_tile(Int<2 % D>{});
    if (D > 3) load_tile(Int<3 % D>{});
    store_tile(Int<0>{}, 0);
    __syncthreads();

    int t = 0;
    for (; t + D <= nkt; t += D) {                                  // steady state: straight-line body
        stage(Int<0>{}, t);
        if (D > 1) stage(Int<1 % D>{}, t + 1);
        if (D > 2) stage(Int<2 % D>{}, t + 2);
        if (D > 3) stage(Int<3 % D>{}, t + 3);
    }
    {                                                               // remainder (< D tiles), once
        const int r = nkt - t;
        if (r > 0) stage(Int<0>{}, t);
        if (D > 2 && r > 1) stage(Int<1 % D>{}, t + 1);
        if (D > 3 && r > 2) stage(Int<2 % D>{}, t + 2);
    }

    // ---- epilogue -------------------------------------------------------------------------
    // lane owns C[m][n .. n+3] with m = frag row (lane&15), n = frag col 4*g + r
    if (p.splitk > 1) {
        float* slab = p.partial + (long)blockIdx.z * p.M * p.N;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + wave_m * (BM / 2) + i * 16 + l16;
            if (m >= p.M) continue;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wave_n * (BN / 2) + j * 16 + 4 * g;
                if (n < p.N) *reinterpret_cast<f32x4*>(slab + (long)m * p.N + n) = acc[i][j];
            }
        }
        return;
    }

    lb_gemm_tile_epilogue<TM, TN, GEGLU>(p, acc, m0 + wave_m * (BM / 2) + l16, n0 + wave_n * (BN / 2) + 4 * g,
                                         n0 + wave_n * (BN / 64) * 16 + 4 * g);
}

// Sum split-K slabs and run the same epilogue.
__global__ void gemm_splitk_reduce_kernel(const LbGemmParams p) {
    const long quads = (long)p.M * (p.N / 4);
    const long stride = (long)gridDim.x * blockDim.x;
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < quads; q += stride) {
        const int m = (int)(q / (p.N / 4));
        const int n = (int)(q - (long)m * (p.N / 4)) * 4;
        f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int z = 0; z < p.splitk; ++z)
            s += *reinterpret_cast<const f32x4*>(p.partial + ((long)z * p.M + m) * p.N + n);
        lb_gemm_store4(p, m, n, p.rowvec ? m / p.rows_per_batch : 0, s);
    }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
template <int BM, int BN, int D>
static void launch_variant(const LbGemmParams& p, dim3 grid, hipStream_t stream) {
    const bool geglu = (p.flags & LB_GEMM_GEGLU) != 0;
    if (p.conv) {
        hipLaunchKernelGGL((gemm_f16_kernel<BM, BN, true, false, D>), grid, dim3(256), 0, stream, p);
    } else if (geglu) {
        hipLaunchKernelGGL((gemm_f16_kernel<BM, BN, false, true, D>), grid, dim3(256), 0, stream, p);
    } else {
        hipLaunchKernelGGL((gemm_f16_kernel<BM, BN, false, false, D>), grid, dim3(256), 0, stream, p);
    }
}

static int g_pp_auto = 1;          // 0: the automatic policy never picks the ping-pong kernel (A/B studies: lb_gemm_set_pp_auto)
extern "C" void lb_gemm_set_pp_auto(int on) { g_pp_auto = on; }
static int g_force_tile = 0;      // 0 auto, else 1=128x128 2=128x64 3=64x64, direct-to-LDS only: 4=256x128 5=256x256 7=192x128
static int g_force_splitk = 0;    // 0 auto
// 192x128 tile: 1 (default, round 6) = 8 waves x (48 x 64) - tile code 10; 0 = the 6-wave form of rounds 2-5 (3 x 2 waves of 64 x 64 - tile
// code 7: two of the four SIMDs carry two waves, i.e. 64 MFMAs per K-tile against 32 on the other two; with 8 waves every SIMD issues 48).
// Bit-identical (same K order per accumulator).  profiles/r06_gemm_bench_call2.txt: M 4352 x N 1280, K 1280 / 2560 / 5120 =
// 23.4 -> 21.9, 36.2 -> 34.0, 69.5 -> 64.4 us.
static int g_t192_waves8 = 1;
extern "C" void lb_gemm_set_t192_waves8(int on) { g_t192_waves8 = on; }
static int g_kgroups = 1;         // 1 (default, round 6) = small unsplit 64x64 grids run two K-groups per block (tile code 11); 0 = never
extern "C" void lb_gemm_set_kgroups(int on) { g_kgroups = on; }
static int g_depth = 0;           // 0 = per-tile default ring depth, 1..4 = forced (A/B testing)
extern "C" void lb_gemm_set_tuning(int tile, int splitk) { g_force_tile = tile; g_force_splitk = splitk; }
extern "C" void lb_gemm_set_depth(int depth) { g_depth = depth; }
// A/B studies of the tile policy (tools/ab_policy.py): bit 0 no 256x128, bit 1 no 256x256, bit 2 no
// 256x256 for convolutions, bit 3 no 256x256 for plain/GEGLU, bit 4 no 256x128 for convolutions
#ifdef LB_STUDY_BUILD
static int g_policy_off = 0;
extern "C" void lb_gemm_set_policy(int disable_mask) { g_policy_off = disable_mask; }
#else
static constexpr int g_policy_off = 0;     // (the A/B switches of the tile policy exist only in -DLB_STUDY_BUILD libraries)
#endif
// 3x3 / stride 1 / pad 1 convs from an LDS-resident halo tile (conv3_halo.hip): 1.27-1.79x the implicit GEMM on
// every conv of the benchmark's B=17 programs (profiles/r02_halo_bench.txt).  mode 0 = never, 1 = whenever the
// halo grid has at least LB_HALO_MIN_BLOCKS blocks (default), 2 = whenever eligible (tests / A-B studies).
int lb_conv3x3_halo_eligible(const LbGemmParams& p);
long lb_conv3x3_halo_blocks(const LbGemmParams& p);
int lb_conv3x3_halo_launch(LbGemmParams p, hipStream_t stream);
int lb_conv3x3_narrow_eligible(const LbGemmParams& p);
int lb_conv3x3_narrow_launch(LbGemmParams p, hipStream_t stream);
int lb_upconv_halo_eligible(const LbGemmParams& p);
int lb_upconv_halo_launch(LbGemmParams p, hipStream_t stream);
#define LB_HALO_MIN_BLOCKS 96
static int g_halo = 1;
extern "C" void lb_gemm_set_halo(int mode) { g_halo = mode; }
static bool use_halo(const LbGemmParams& p) {
    if (g_halo == 0 || g_force_tile || !lb_conv3x3_halo_eligible(p)) return false;
    return g_halo == 2 || lb_conv3x3_halo_blocks(p) >= LB_HALO_MIN_BLOCKS;
}

// ping-pong 256x256 main loop (gemm_pp.hip): tile code 9
int lb_gemm_pp_eligible(const LbGemmParams& p);
int lb_gemm_launch_pp(const LbGemmParams& p, dim3 grid, hipStream_t stream);
// direct-to-LDS variant (gemm_glds.hip)
int lb_gemm_launch_glds(const LbGemmParams& p, int tile, int stages, dim3 grid, hipStream_t stream);
void lb_gemm_glds_init();
static int g_variant = 1, g_stages = 0;   // default: direct-to-LDS staging (wins the MI355X sweep by 5-20 %)
extern "C" void lb_gemm_set_variant(int variant, int stages) {   // variant < 0: back to the default
    g_variant = variant < 0 ? 1 : variant;
    g_stages = variant < 0 ? 0 : stages;
    if (variant == 1) lb_gemm_glds_init();
}

// 1 (default) = fp16 row-major epilogues store 16 B per lane (column-group pairs exchanged through v_permlane16_swap); 0 = 8-B stores
int g_lb_wide_store = 1;     // measured: VAE decode B=17 45.9 -> 44.8 ms, UNet B=17 38.11 -> 37.69 ms, bit-identical (profiles/r03_wide_store_ab.txt)
extern "C" void lb_gemm_set_wide_store(int on) { g_lb_wide_store = on; }
// 1 (default) = tiles whose epilogue is "alpha, bias, row vector or fp16 residual, fp16 row-major out" take the one-round-trip form
// (lb_gemm.h: lb_gemm_tile_epilogue_lean); 0 = the per-row form everywhere.  Bit-identical; read when a launch is issued / recorded.
int g_lb_lean_epilogue = 1;
extern "C" void lb_gemm_set_lean_epilogue(int on) { g_lb_lean_epilogue = on; }

extern "C" long lb_gemm_workspace_bytes(int M, int N) {
    // enough for the largest split the heuristic can pick (<= 16 slabs)
    return (long)16 * M * N * (long)sizeof(float);
}

template <int BM, int BN>
static void launch_depth(const LbGemmParams& p, int depth, dim3 grid, hipStream_t stream) {
    switch (depth) {
        case 1: launch_variant<BM, BN, 1>(p, grid, stream); break;
        case 2: launch_variant<BM, BN, 2>(p, grid, stream); break;
        case 3: launch_variant<BM, BN, 3>(p, grid, stream); break;
        default: launch_variant<BM, BN, 4>(p, grid, stream); break;
    }
}

// default ring depth per tile (measured on MI355X, tools/sweep_gemm.py); 0 in g_depth = use these
static int default_depth(const LbGemmParams& p, int tile) {
    if (tile == 1) return p.conv ? 1 : 3;    // 128x128: conv gather + 3 slots exceeds the register budget
    if (tile == 2) return 3;
    return 4;
}

static int gemm_launch_impl(LbGemmParams p, int tile, int depth, int variant, int stages, dim3 grid,
                            hipStream_t stream) {
    if (tile == 9) {
        lb_gemm_launch_pp(p, grid, stream);
    } else if (variant == 1 && p.zero_page != nullptr) {
        lb_gemm_launch_glds(p, tile, stages, grid, stream);
    } else {
        if (depth <= 0) depth = default_depth(p, tile);
        if (tile == 1) launch_depth<128, 128>(p, depth, grid, stream);
        else if (tile == 2) launch_depth<128, 64>(p, depth, grid, stream);
        else launch_depth<64, 64>(p, depth, grid, stream);
    }
    int rc = lb_check_launch("lb_gemm_f16");
    if (rc) return rc;
    if (p.splitk > 1) {
        const long quads = (long)p.M * (p.N / 4);
        long gsz = (quads + 255) / 256;
        if (gsz > 2048) gsz = 2048;
        hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3((unsigned)gsz), dim3(256), 0, stream, p);
        rc = lb_check_launch("lb_gemm_f16(split-K reduce)");
    }
    return rc;
}

// Tile / split-K policy of one launch (pure host arithmetic; also exported as lb_gemm_plan so that the
// policy can be pinned by CPU tests and inspected by tools without launching anything).
static void gemm_plan(const LbGemmParams& p, int& tile_out, int& splitk_out, long& nblk_out) {
    const bool geglu = (p.flags & LB_GEMM_GEGLU) != 0;
    const int n_eff = geglu ? p.N / 2 : p.N;

    // tile choice: the largest tile that still gives >= ~1 block per CU; small grids fall back to
    // 64x64 and are widened with split-K so that weight streaming is spread over the chip.
    auto blocks = [&](int bm, int bn) {
        const int bn_eff = geglu ? bn / 2 : bn;
        return (long)((p.M + bm - 1) / bm) * ((n_eff + bn_eff - 1) / bn_eff);
    };
    int tile = g_force_tile;
    if (tile == 9 && !lb_gemm_pp_eligible(p)) tile = 0;       // (forced for A/B studies: ineligible problems keep the automatic choice)
    if (!tile) {
        // MI355X sweeps (tools/sweep_gemm.py; profiles/r01_gemm_variant_sweep.txt, r01_gemm_tile4_sweep.txt).
        // Direct-to-LDS family: the 8-wave 256x128 tile with a 3-stage ring (144 KiB, two K-tiles = 96 KiB
        // in flight per CU, 25 % less operand traffic per FLOP) wins whenever it fills ~2/3 of the chip,
        // N does not pad badly to 128 columns and K is long enough to amortise its prologue.  Otherwise
        // (Convolutions only with >= 4 rounds of such tiles: the UNet's convs measured better on the 4-wave
        // tiles.)  >= 3 rounds of 128x128 tiles -> 128x128, about one round -> 128x64, small grids -> 64x64.
        const long b128 = blocks(128, 128);
        tile = b128 >= 700 ? 1 : (b128 >= 224 ? (g_variant == 1 ? 2 : 1) : 3);
        if (n_eff <= 64) tile = 3;                                              // (VAE conv_out: 3 real columns)
        if (g_variant == 1 && p.zero_page != nullptr) {
            const long b256 = blocks(256, 128);
            const int unit = geglu ? 64 : 128;
            const int n_pad = (n_eff + unit - 1) / unit * unit;
            const bool n_fits = (long)n_pad * 10 <= (long)n_eff * 11;           // <= 10 % padded columns
            const bool allow4 = !(g_policy_off & 1) && !(p.conv && (g_policy_off & 16));
            const bool allow5 = !(g_policy_off & 2) && !(p.conv ? (g_policy_off & 4) : (g_policy_off & 8));
            if (allow4 && n_fits && (b256 >= 1024 || (!p.conv && b256 >= 160 && p.K >= 1024))) tile = 4;
            // 256x256 (64x128 per wave, two 64 KiB stages): half the operand bytes per FLOP of 128x128.
            // Measured cost per tile = 1.74 x a 256x128 tile; taken when that saves rounds over the chip.
            const long b512 = blocks(256, 256);
            const long rounds4 = (b256 + 255) / 256, rounds5 = (b512 + 255) / 256;
            if (allow5 && (b512 >= 1024 || (b512 >= 160 && p.K >= 1024)) && rounds5 * 174 < rounds4 * 100) tile = 5;
            // 192x128 (6 waves): 3/4 of a 256x128 tile's work per block, meant for wave quantisation (M = 4352 x N = 1280
            // is 170 blocks of 256x128 = 2/3 of the chip, but 230 blocks of 192x128).  MEASURED (profiles/
            // r02_tile7_bench.txt): a 192x128 block costs ~0.95 of a 256x128 block, not 0.78 - the K loop is bound per CU
            // (1.1-1.3 us per K-tile whether the chip is full or 2/3 empty), so the smaller tile only wins 3-6 % on two
            // shapes and loses on the rest.  OFF by default; lb_gemm_set_policy bit 5 turns the rule on for A/B studies.
            if ((g_policy_off & 32) && !geglu && !p.conv && (tile == 4 || tile == 5) && n_fits) {
                const long b192 = blocks(192, 128);
                const long rounds7 = (b192 + 255) / 256;
                const long cur = tile == 5 ? rounds5 * 174 : rounds4 * 100;
                if (b192 >= 160 && rounds7 * 78 < cur) tile = g_t192_waves8 ? 10 : 7;
            }
            // ... except where it did win in that measurement (24.9 vs 26.5 us): a GEMM whose 256x128 grid is a single partial
            // round (M = 4352, N = 1280: 170 blocks, 230 of 192x128).  Round 5 (profiles/r05_gemm_tile7_vs_auto.txt, with the
            // one-round-trip epilogue): it wins there at every K - K 1280: 24.2 vs 26.2 us, K 2560: 37.2 vs 39.0, K 5120 (the
            // feed-forward output projection, 60 per forward): 71.9 vs 73.7 - so the K <= 1536 condition of rounds 2-4 is gone.
            if (!(g_policy_off & 64) && !geglu && !p.conv && tile == 4 && n_fits && b256 < 224) {
                const long b192 = blocks(192, 128);
                if (b192 <= 256 && b192 * 10 >= b256 * 13) tile = g_t192_waves8 ? 10 : 7;
            }
        }
    }
    // GEGLU projections of the small-batch programs (M 512 x N 10240, M 2048 x N 5120: 256x128 / 4-wave tiles until round 6): one round (or a short
    // second one) of 8-wave 192x128 tiles - 24.9 -> 21.7 us and 28.2 -> 25.1 us, rocBLAS 22.3 / 25.1 (profiles/r06_gemm_bench_call14.txt).
    if (!g_force_tile && g_t192_waves8 && geglu && !p.conv && (tile == 1 || tile == 2 || tile == 4) && p.M <= 2048 && g_variant == 1 &&
        p.zero_page != nullptr && blocks(192, 128) <= 448 && p.K >= 512)
        tile = 10;
    if (tile >= 4 && tile != 9 && (g_variant != 1 || p.zero_page == nullptr)) tile = 1;
    // Ping-pong 256x256 loop (gemm_pp.hip): 1.15-1.27x the lock-step 8-wave tiles wherever 256-wide tiles fill the chip and
    // the K loop is long enough to pay for its deeper prologue - on the B = 17 programs the GEGLU projections (M 4352 /
    // 17408), the fused q|k|v projection, the 640-wide feed-forward output and the per-branch context projection
    // (profiles/r04_gemm_bench_call6.txt).  Shapes with ~1/3 of the chip in 256x256 tiles (N = 1280 at M = 4352), and K = 640
    // unless the grid runs for several rounds, stay on the smaller lock-step tiles.
    if (!g_force_tile && g_pp_auto && lb_gemm_pp_eligible(p)) {
        const int bn_out = geglu ? 128 : 256;
        const long n_pad = (long)((n_eff + bn_out - 1) / bn_out) * bn_out;
        const long b9 = blocks(256, 256);
        if (b9 >= 192 && (p.K >= 1024 || b9 >= 1024) && n_pad * 4 <= (long)n_eff * 5) tile = 9;
    }   // 6- / 8-wave tiles: direct-to-LDS family only
    // long-K problems with 1-2.5 small tiles per CU (B=2 convs of the UNet: M=2048, N=640, K=5760): no split
    // is possible at 64x64 (> 256 blocks), so they crawl through ~90 K-tiles per block.  Take 128x64 tiles
    // (half the blocks) and let the split-K rule below spread K instead.
    if (!g_force_tile && tile == 3 && !geglu && p.partial != nullptr && (p.K + BK - 1) / BK >= 64) {
        const long b64 = blocks(64, 64);
        if (b64 > 256 && b64 <= 640 && blocks(128, 64) <= 256) tile = 2;
    }
    const int bm = (tile == 7 || tile == 10) ? 192 : (tile == 11 ? 64 : (tile >= 4 ? 256 : (tile == 3 ? 64 : 128)));
    const int bn = (tile == 5 || tile == 9) ? 256 : ((tile == 1 || tile == 4 || tile == 7 || tile == 10) ? 128 : 64);
    const long nblk = blocks(bm, bn);
    int splitk = 1;
    // (LN_A: a block must see whole rows of A)
    if (!geglu && p.partial != nullptr && !(p.flags & LB_GEMM_LN_A)) {
        const int k_tiles = (p.K + BK - 1) / BK;
        if (g_force_splitk) splitk = g_force_splitk;
        else if (nblk <= 256) {
            // long-K, few-tile problems (M = 256..512 rows against K up to 23040): aim at ~2.5 blocks
            // per CU but keep >= 16 K-tiles per slice so the slab round trip stays negligible
            splitk = (int)((640 + nblk - 1) / nblk);
            const int max_by_k = k_tiles / 16 > 0 ? k_tiles / 16 : 1;
            if (splitk > max_by_k) splitk = max_by_k;
            if (splitk > 16) splitk = 16;
        }
        if (splitk > k_tiles) splitk = k_tiles;
        if (splitk < 1) splitk = 1;
        if (tile == 9 && !g_force_splitk) splitk = 1;         // (the policy takes the ping-pong kernel only for chip-filling grids)
        // 6- / 8-wave tiles are only chosen for grids that already cover >= 62 % of the chip: their fp32 slabs (M x N x 4 B per slice,
        // written and read back) outweigh the operands.  Measured (profiles/r05_gemm_bench_call2.txt): M 4352, N 1280, K 2560 / 5120 on
        // the 256x128 tile = 39.1 / 75.1 us unsplit against 57.6 / 96.5 us with the 2 / 4 slices the "<= 256 blocks" rule above gave
        // them (120 such launches per transition).  The rule is meant for the 4-wave tiles of the M = 256..1024 programs.
        if (tile >= 4 && tile != 11 && !g_force_splitk) splitk = 1;
    }
    // 64x64 grids that leave the chip a single wave per SIMD (<= 200 blocks, unsplit): two K-groups per block (tile code 11, gemm_glds.hip).
    // MI355X, cold weights (profiles/r06_gemm_bench_call5.txt): M 512 x N 1280 x K 1280 (192 per B = 2 forward) 9.4 -> 7.9 us; grids with two
    // or more blocks per CU lose (N 3840: 15.0 -> 18.1, GEGLU N 10240: 24.9 -> 42.9), as does a grid already split over K (K 5120).
    if (!g_force_tile && g_kgroups && tile == 3 && splitk == 1 && !p.conv && g_variant == 1 && p.zero_page != nullptr && nblk <= 200 &&
        (p.K + BK - 1) / BK >= 8)
        tile = 11;
    tile_out = tile;
    splitk_out = splitk;
    nblk_out = nblk;
}

extern "C" int lb_gemm_plan(const LbGemmParams* pp, int* tile, int* splitk, long* blocks) {
    LB_REQUIRE(pp != nullptr && pp->M > 0 && pp->N > 0 && pp->K > 0, "lb_gemm_plan: empty problem");
    int t = 0, sk = 1;
    long nb = 0;
    if (g_halo != 0 && !g_force_tile && lb_conv3x3_narrow_eligible(*pp)) { t = 8; nb = (long)pp->M / 256; }   // tile code 8 = narrow-N conv kernel
    else if (use_halo(*pp)) { t = 6; nb = lb_conv3x3_halo_blocks(*pp); }     // tile code 6 = halo-tile conv kernel
    else gemm_plan(*pp, t, sk, nb);
    if (tile) *tile = t;
    if (splitk) *splitk = sk;
    if (blocks) *blocks = nb;
    return 0;
}

// Rows per sample of the LB_GEMM_CH_STATS buffer for the launch lb_gemm_f16 would make (same routing as below); 0 = no statistics.
long lb_conv_halo_stat_rows_total(const LbGemmParams& p);
extern "C" int lb_gemm_ch_stat_rows(const LbGemmParams* pp) {
    if (pp == nullptr || !pp->conv || pp->M <= 0 || pp->Hout <= 0 || pp->Wout <= 0) return 0;
    const LbGemmParams& p = *pp;
    bool halo;
    if (p.scatter == 2) halo = lb_upconv_halo_eligible(p) != 0;
    else halo = !(g_halo != 0 && !g_force_tile && lb_conv3x3_narrow_eligible(p)) && use_halo(p);
    if (!halo) return 0;
    const long samples = (long)p.M / ((long)p.Hout * p.Wout);
    return samples > 0 ? (int)(lb_conv_halo_stat_rows_total(p) / samples) : 0;
}

extern "C" int lb_gemm_f16(const LbGemmParams* pp, void* stream) {
    LbGemmParams p = *pp;
    const bool geglu = (p.flags & LB_GEMM_GEGLU) != 0;
    LB_REQUIRE(p.M > 0 && p.N > 0 && p.K > 0, "lb_gemm_f16: empty problem");
    LB_REQUIRE(p.K % 8 == 0 && p.ldw % 8 == 0, "lb_gemm_f16: K and ldw must be multiples of 8");
    LB_REQUIRE(p.N % 4 == 0 && (geglu ? p.N % 8 == 0 : true), "lb_gemm_f16: N must be a multiple of 4");
    LB_REQUIRE(p.ldc % 4 == 0 || (p.flags & LB_GEMM_TRANS_OUT), "lb_gemm_f16: ldc must be a multiple of 4");
    if (p.conv) {
        LB_REQUIRE(p.Cin % 8 == 0 && p.ldx % 8 == 0, "lb_gemm_f16: conv Cin/ldx must be multiples of 8");
        LB_REQUIRE(p.K == p.KH * p.KW * p.Cin, "lb_gemm_f16: conv K != KH*KW*Cin");
        LB_REQUIRE(p.M % (p.Hout * p.Wout) == 0, "lb_gemm_f16: conv M must be B*Hout*Wout");
        if (p.scatter)
            LB_REQUIRE(p.KH == 2 && p.KW == 2 && p.stride == 1 && p.ups == 0 && p.Hout == p.Hin && p.Wout == p.Win &&
                           !(p.flags & (LB_GEMM_TRANS_OUT | LB_GEMM_GEGLU)) && p.residual == nullptr,
                       "lb_gemm_f16: sub-pixel conv needs KH=KW=2, stride 1, Hout=Hin, no residual");
    } else {
        LB_REQUIRE(p.lda % 8 == 0, "lb_gemm_f16: lda must be a multiple of 8");
    }
    if (p.flags & LB_GEMM_LN_A) {
        LB_REQUIRE(!p.conv && p.ln_colsum != nullptr && p.zero_page != nullptr && g_variant == 1,
                   "lb_gemm_f16: LB_GEMM_LN_A needs a plain / GEGLU GEMM of the direct-to-LDS family with ln_colsum");
        LB_REQUIRE(p.lda >= p.K && !(p.flags & LB_GEMM_TRANS_OUT), "lb_gemm_f16: LB_GEMM_LN_A normalises whole rows of A");
    }
    if (p.alpha == 0.f) p.alpha = 1.f;
    p.reserved2_ = (g_lb_wide_store & 1) | (g_lb_lean_epilogue ? 2 : 0);
    if (p.scatter == 2) {       // all four sub-pixel parities in one launch: only the halo kernel implements it
        LB_REQUIRE(lb_upconv_halo_eligible(p) != 0, "lb_gemm_f16: scatter = 2 needs Cin % 64 == 0, W % 16 == 0, stacked [4][N][K] weights");
        LB_DISPATCH("lb_upconv2x_halo_f16", lb_upconv_halo_launch(p, s));
    }
    if (g_halo != 0 && !g_force_tile && lb_conv3x3_narrow_eligible(p))      // N <= 16: conv_out of the VAE / UNet (conv3_narrow.hip)
        LB_DISPATCH("lb_conv3x3_narrow_f16", lb_conv3x3_narrow_launch(p, s));
    if (use_halo(p)) LB_DISPATCH("lb_conv3x3_halo_f16", lb_conv3x3_halo_launch(p, s));
    LB_REQUIRE(!(p.flags & LB_GEMM_CH_STATS), "lb_gemm_f16: LB_GEMM_CH_STATS is implemented by the halo-tile conv kernels only "
                                              "(check lb_gemm_plan == 6 / lb_conv_halo_plan before asking for it)");
    int tile = 0, splitk = 1;
    long nblk = 0;
    gemm_plan(p, tile, splitk, nblk);
    p.splitk = splitk;
    const int depth = g_depth, variant = g_variant;
    int stages = g_stages;
    if (variant == 1) {
        lb_gemm_glds_init();                       // (wrapper runs at record time, never inside a capture)
        if (stages == 0)
            stages = (tile == 3 || tile == 4 || tile == 7 || tile == 10 || tile == 11) ? 3 : 2;   // 256x128: 3 x 48 KiB; 128x128 / 128x64: 2 stages; 64x64: 3 x 16 KiB
    }
    const dim3 grid((unsigned)nblk, 1, (unsigned)splitk);
    LB_DISPATCH("lb_gemm_f16", gemm_launch_impl(p, tile, depth, variant, stages, grid, s));
}
