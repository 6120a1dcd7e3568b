// 3x3 / stride 1 / pad 1 convolution with a NARROW output (N <= 16 channels: the VAE's conv_out 128 -> 3 and the UNet's
// conv_out 320 -> 4) for gfx950.
//
// Why a kernel of its own: the 128-channel block of conv3_halo.hip computes 128 output columns whatever N is - the VAE's
// last conv ran 32x padded at 39 TFLOP/s (1.05 ms per B = 17 decode batch, profiles/r02_program_op_breakdown.txt) although
// it is a pure streaming problem: read 1.14 GB of activations once, write 71 MB.  Here one 16-column MFMA tile is the whole
// output width, the weights (N x 9 Cin, a few KB) live in REGISTERS, and the only LDS traffic is the halo tile.
//
// Block = 256 threads = 4 waves, 16 x 16 output pixels; wave w owns pixel rows 4w .. 4w+3 (four 16-pixel MFMA row
// groups).  Per 64-channel chunk of Cin: the 18 x 18 halo (324 pixels x 128 B = 40.5 KiB) is staged with
// global_load_lds_dwordx4 in the [rows][64] fp16 image with the 16-B chunk XOR-swizzle of the other kernels (swizzle on the
// source address, out-of-image pixels from the zero page), then 9 taps x 2 K-halves x 4 row groups = 72 MFMA 16x16x32 per
// wave read it through shifted rows.  One halo buffer per block, two blocks per CU (234 VGPRs): one block's staging overlaps the
// other's MFMAs.  Operands are swapped (a = weights, b = pixels) like everywhere else, so lane (g, l16) holds output
// columns 4g .. 4g+3 of pixel l16 of each row group: only the lanes with 4g < N store.
//
// Replaces: the last Conv2d of diffusers' AutoencoderKL decoder / UNet2DConditionModel (conv_out), reached from
// /root/reference/latentblending/diffusers_holder.py:135 and :336.
#include "lb_common.h"
#include "lb_gemm.h"

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

#define NARROW_T 16                                   // tile side in pixels
#define NARROW_HW (NARROW_T + 2)                      // halo side
#define NARROW_HR (NARROW_HW * NARROW_HW)             // 324 halo pixels
#define NARROW_HRG ((NARROW_HR + 7) / 8)              // 41 groups of 8 rows
#define NARROW_LDS (NARROW_HRG * 8 * 64)              // halves

__global__ void __launch_bounds__(256, 2) conv3x3_narrow_kernel(const LbGemmParams p) {
    __shared__ __attribute__((aligned(16))) f16 halo[NARROW_LDS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, l16 = lane & 15;
    const int tiles_x = p.Win / NARROW_T, tiles_y = p.Hin / NARROW_T;
    int tile = blockIdx.x;
    const int x0 = (tile % tiles_x) * NARROW_T;
    tile /= tiles_x;
    const int y0 = (tile % tiles_y) * NARROW_T;
    const int b = tile / tiles_y;
    const lb_half* zero = reinterpret_cast<const lb_half*>(p.zero_page);
    const int nchunks = p.Cin / 64;

    // loader: wave instruction j covers row group j * 4 + wave; lane (r8 = lane >> 3, slot = lane & 7) fetches logical
    // chunk slot ^ r8 of halo row 8 * group + r8
    constexpr int NR = (NARROW_HRG + 3) / 4;          // 11 rounds
    const int r8 = lane >> 3, cl = (lane & 7) ^ r8;
    int h_off[NR];                                    // offset of the lane's 16-B piece in units of 8 halves (-1: zero page)
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const int gidx = j * 4 + wave;
        const int row = gidx * 8 + r8;
        const int hy = row / NARROW_HW, hx = row - hy * NARROW_HW;
        const int y = y0 + hy - 1, x = x0 + hx - 1;
        const bool ok = gidx < NARROW_HRG && row < NARROW_HR && y >= 0 && y < p.Hin && x >= 0 && x < p.Win;
        h_off[j] = ok ? (int)((((long)(b * p.Hin + y) * p.Win + x) * p.ldx >> 3) + cl) : -1;
    }
    // weights of output column l16 (zero rows beyond N): the lane's 8 k-values of (tap, K-half)
    const bool n_ok = l16 < p.N;
    const lb_half* wrow = p.W + (long)(n_ok ? l16 : 0) * p.ldw + g * 8;

    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int hbase[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) hbase[i] = (wave * 4 + i) * NARROW_HW + l16;     // halo row of tap (0, 0) of the lane's pixel

    const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int c = 0; c < nchunks; ++c) {
        if (c) __syncthreads();                        // everybody finished reading the previous chunk
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const int gidx = j * 4 + wave;
            if (gidx < NARROW_HRG) {
                const lb_half* src = h_off[j] >= 0 ? p.A + (long)h_off[j] * 8 + (long)c * 64 : zero;
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(halo + gidx * 8 * 64), 16, 0, 0);
            }
        }
        f16x8 wf[9][2];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int s = 0; s < 2; ++s)
                wf[tap][s] = n_ok ? *reinterpret_cast<const f16x8*>(wrow + (long)tap * p.Cin + c * 64 + s * 32) : zero8;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this thread's halo pieces (LDS-DMA) and weight fragments have landed
        __syncthreads();
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int shift = (tap / 3) * NARROW_HW + tap % 3;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int chunk = s * 4 + g;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = hbase[i] + shift;
                    const f16x8 af = *reinterpret_cast<const f16x8*>(halo + r * 64 + ((chunk ^ (r & 7)) << 3));
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[tap][s], af, acc[i], 0, 0, 0);
                }
            }
        }
    }
    // epilogue: lane (g, l16) holds columns 4g .. 4g+3 of pixel (y0 + 4 wave + i, x0 + l16)
    const int n = 4 * g;
    if (n < p.N) {
        f32x4 bias = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) bias = *reinterpret_cast<const f32x4*>(p.bias + n);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long m = ((long)(b * p.Hin + y0 + wave * 4 + i)) * p.Win + x0 + l16;
            float o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = acc[i][r] * p.alpha + bias[r];
            if (p.flags & LB_GEMM_OUT_F32)
                *reinterpret_cast<f32x4*>((float*)p.C + m * p.ldc + n) = (f32x4){o[0], o[1], o[2], o[3]};
            else
                *reinterpret_cast<f16x4*>((f16*)p.C + m * p.ldc + n) = (f16x4){(f16)o[0], (f16)o[1], (f16)o[2], (f16)o[3]};
        }
    }
}

int lb_conv3x3_narrow_eligible(const LbGemmParams& p) {
    if (!p.conv || p.KH != 3 || p.KW != 3 || p.stride != 1 || p.pad != 1 || p.ups || p.scatter) return 0;
    if (p.Hout != p.Hin || p.Wout != p.Win || p.Cin % 64 != 0 || p.K != 9 * p.Cin) return 0;
    if (p.N > 16 || p.N % 4 != 0 || p.zero_page == nullptr) return 0;
    if (p.flags & ~LB_GEMM_OUT_F32) return 0;           // plain bias epilogue only
    if (p.residual != nullptr || p.rowvec != nullptr) return 0;
    if (p.Win % NARROW_T != 0 || p.Hin % NARROW_T != 0 || p.M % (p.Hin * p.Win) != 0) return 0;
    if ((long)p.M * p.ldx >= (1l << 34)) return 0;      // (32-bit piece offsets)
    return 1;
}

int lb_conv3x3_narrow_launch(LbGemmParams p, hipStream_t stream) {
    if (p.alpha == 0.f) p.alpha = 1.f;
    const long tiles = (long)(p.M / (p.Hin * p.Win)) * (p.Hin / NARROW_T) * (p.Win / NARROW_T);
    LB_REQUIRE(tiles < (1l << 31), "conv3x3 narrow: too many tiles for one launch");
    hipLaunchKernelGGL(conv3x3_narrow_kernel, dim3((unsigned)tiles), dim3(256), 0, stream, p);
    return lb_check_launch("lb_conv3x3_narrow_f16");
}

extern "C" int lb_conv3x3_narrow_f16(const LbGemmParams* pp, void* stream) {
    LbGemmParams p = *pp;
    LB_REQUIRE(p.M > 0 && p.N > 0 && p.K > 0, "lb_conv3x3_narrow_f16: empty problem");
    LB_REQUIRE(lb_conv3x3_narrow_eligible(p) != 0,
               "lb_conv3x3_narrow_f16: needs a 3x3 / stride 1 / pad 1 conv, N <= 16, Cin % 64 == 0, H and W multiples of 16, bias-only epilogue");
    LB_REQUIRE(p.ldw % 8 == 0 && p.ldx % 8 == 0 && p.ldc % 4 == 0, "lb_conv3x3_narrow_f16: ldw / ldx multiples of 8, ldc multiple of 4");
    LB_DISPATCH("lb_conv3x3_narrow_f16", lb_conv3x3_narrow_launch(p, s));
}
