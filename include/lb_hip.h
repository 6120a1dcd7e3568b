/*
 * lb_hip.h — C-ABI of liblbhip.so, the gfx950 (MI355X) kernel library behind
 * latentblending_amd.
 *
 * The reference (lunarring/latentblending) has NO FFI boundary of its own: it is pure Python and
 * reaches the device only through third-party wheels.  This header is the
 * boundary we define for its hot path; every entry point names the reference call site whose
 * device work it replaces (paths relative to /root/reference).  A maintainer of the reference
 * would bind these with ctypes exactly as latentblending_amd/hip/lib.py does.
 *
 * Conventions
 *   - plain pointers and sizes only; device pointers unless a parameter says "host";
 *   - every launcher takes the hipStream_t as `void* stream`, never allocates, never
 *     synchronises (hipGraph-capturable) and returns hipError_t as int (0 = success);
 *     lb_last_error_string() describes the last failure on the calling thread;
 *   - fp16 tensors are IEEE binary16 (`lb_half` = uint16_t storage); activations are NHWC
 *     ([B, H, W, C] == [B*H*W tokens, C]); weights are [N][K] row-major with K = KH*KW*Cin for
 *     convolutions ((ky, kx, cin) order).
 */
#ifndef LB_HIP_H
#define LB_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef uint16_t lb_half;

/* ---- library ------------------------------------------------------------------------- */
int lb_version(void);
const char* lb_last_error_string(void);
int lb_device_info(int device, int* cu_count, long* lds_per_cu, char* arch, int arch_len);

/* ---- latent mixing (latentblending/utils.py:29-71 interpolate_spherical; called from
 *      blending_engine.py:449 parental mix and diffusers_holder.py:324 crossfeed) -------- */
/* host arrays of device pointers; fracts is a HOST array of doubles; out dtype = f16 for the
 * f16 variant, f32 for f32 and f64 inputs (utils.py:47-50,66-69). */
int lb_slerp_pairs_f16(const void* const* p0, const void* const* p1, void* const* out,
                       const double* fracts, int npairs, long n, void* stream);
int lb_slerp_pairs_f32(const void* const* p0, const void* const* p1, void* const* out,
                       const double* fracts, int npairs, long n, void* stream);
int lb_slerp_pairs_f64(const void* const* p0, const void* const* p1, void* const* out,
                       const double* fracts, int npairs, long n, void* stream);
/* contiguous batch [npairs][n] with device-side fracts: the HBM-roofline form (6 B/element). */
int lb_slerp_batched_f16(const void* p0, const void* p1, void* out, const double* fracts_dev,
                         long npairs, long n, void* stream);
/* the same with ELEMENT strides between consecutive pairs of each input (0 = all pairs share that tensor: the
 * parental mix of one pair of anchor latents at npairs fractions, blending_engine.py:443-450); out is
 * [npairs][n] contiguous; n % 8 == 0 (register-staged single pass up to n = 32768, two passes beyond). */
int lb_slerp_strided_f16(const void* p0, long stride0, const void* p1, long stride1, void* out,
                         const double* fracts_dev, long npairs, long n, void* stream);


/* latentblending/utils.py:97 interpolate_linear on tensors (blending_engine.py:650) */
int lb_lerp_f16(const void* p0, const void* p1, void* out, long n, double fract, void* stream);
int lb_lerp_f32(const void* p0, const void* p1, void* out, long n, double fract, void* stream);

/* ---- scheduler (diffusers_holder.py:330 scale_model_input, :347-349 CFG combine,
 *      :356 scheduler.step — diffusers Euler / Euler-ancestral) ------------------------- */
/* params_dev: float[batch][8] = {sigma_from, sigma_next, sigma_up, guidance, dt, -, -, -} */
int lb_scale_model_input_f16(const void* x, void* out, const float* params_dev, long per_sample,
                             int batch, int dup_for_cfg, void* stream);
int lb_euler_step_f16(const void* x, const void* eps, const void* noise, void* out,
                      const float* params_dev, long per_sample, int batch, int cfg, int ancestral,
                      void* stream);
/* diffusers DDIMScheduler.step, eta = 0, epsilon prediction (diffusers_holder.py:356 with a DDIM scheduler on the pipe; the
 * reference's SDXL pipes carry Euler schedulers, :42).  params_dev: float[batch][8] = {0, sqrt(abar_t), sqrt(abar_prev),
 * guidance, sqrt(1 - abar_t), sqrt(1 - abar_prev), 1 / sqrt(abar_t), -} - slot 0 = 0 makes lb_scale_model_input_f16 the identity, as
 * DDIM's scale_model_input is; slot 6 is the fp32 reciprocal the kernel MULTIPLIES by where diffusers divides a device tensor by a
 * 0-dim host tensor (the tensor library's true-division kernel multiplies by 1 / b for a host scalar b).  fp16 tensor arithmetic
 * rounded op by op like diffusers' (no fp32 upcast in DDIM).  eps as above. */
int lb_ddim_step_f16(const void* x, const void* eps, void* out, const float* params_dev, long per_sample, int batch,
                     int cfg, void* stream);

/* ---- dense contractions (every nn.Linear / nn.Conv2d of the UNet at
 *      diffusers_holder.py:336, of the VAE decoder at :135 and of LPIPS-Alex at
 *      blending_engine.py:756) ----------------------------------------------------------- */
enum {
    LB_GEMM_OUT_F32 = 1,     /* C is float32 (VAE residual stream) */
    LB_GEMM_RES_F32 = 2,     /* residual is float32 */
    LB_GEMM_GEGLU = 4,       /* W = [h rows | gate rows], C[M, N/2] = h * gelu(gate) */
    LB_GEMM_TRANS_OUT = 8,   /* store C^T: C[n*ldc + m] */
    LB_GEMM_SILU = 16,       /* SiLU after bias/residual */
    LB_GEMM_RELU = 32,       /* ReLU after bias/residual */
    LB_GEMM_QUICK_GELU = 128,/* x * sigmoid(1.702 x) after bias (CLIP-L MLP) */
    LB_GEMM_GELU = 256,      /* erf GELU after bias (OpenCLIP bigG MLP) */
    LB_GEMM_CH_STATS = 512,  /* halo-tile convs only (lb_conv3x3_halo_f16 / lb_upconv2x_halo_f16, also when lb_gemm_f16 routes
                              * there): besides C, write per (64-pixel row block, output channel) the (sum, sum of squares) of
                              * the STORED values to ch_stats[channel][row block] (float2, channel-major) - the GroupNorm statistics of this conv's
                              * output without a second pass over it (lb_groupnorm_from_stats).  Row block = (tile [, parity])
                              * * 4 + wave row; rows per sample: lb_gemm_ch_stat_rows) */
    LB_GEMM_LN_A = 64        /* A is consumed through a LayerNorm over its K columns (K = the normalised width):
                                C = LN(A) . Wt computed as rstd_m * (A . W'^T - mean_m * colsum) + bias with
                                W' = W * gamma (folded by the caller), ln_colsum[n] = sum_k W'[n][k],
                                bias[n] = b[n] + sum_k W[n][k] * beta[k]; the row statistics are accumulated from
                                the A fragments inside the K loop (no LayerNorm launch, no normalised copy of A).
                                Plain / GEGLU GEMMs of the direct-to-LDS family only, never split along K. */
};

typedef struct LbGemmParams {
    const lb_half* A;        /* [M][lda] activations, or NHWC image for conv */
    const lb_half* W;        /* [N][ldw] */
    void* C;                 /* [M][ldc] f16 (default) / f32 */
    const float* bias;       /* [N] or NULL */
    const void* residual;    /* [M][ldr] f16 / f32 or NULL */
    const lb_half* rowvec;   /* [M / rows_per_batch][ld_rowvec]: per-sample vector added to every
                                row of that sample (time-embedding projection) or NULL */
    float* partial;          /* split-K workspace (lb_gemm_workspace_bytes) or NULL = no split */
    int M, N, K;
    int lda, ldw, ldc, ldr, ld_rowvec, rows_per_batch;
    float alpha;             /* 0 is read as 1 */
    int flags;
    /* implicit-GEMM convolution: A is [B][Hin][Win][ldx] NHWC, M = B*Hout*Wout */
    int conv;
    int Hin, Win, Cin, Hout, Wout, KH, KW, stride, pad;
    int ups;                 /* 1: nearest-2x upsample of the input fused into the gather */
    int ldx;                 /* pixel stride in elements (>= Cin) */
    int splitk;              /* set by the launcher */
    const void* zero_page;   /* >= 16 zero bytes, 16-B aligned (source of masked chunks for the
                                direct-to-LDS variant); NULL = register-ring variant only */
    /* sub-pixel form of "nearest-2x upsample + 3x3 conv": four 2x2 convs on the LOW-res grid, one per
     * output parity (sc_py, sc_px), with pre-summed weights (4/9 of the FLOPs).  scatter = 1:
     * KH = KW = 2, Hout = Hin, Wout = Win, pad is implied (1 - parity) and row m = (b, y, x) is
     * stored at pixel (2y + sc_py, 2x + sc_px) of the [B][2H][2W][ldc] output. */
    int scatter, sc_py, sc_px, reserved_;   /* scatter = 2: ALL four parities in one launch, W = [4][N][ldw] stacked
                                             * (parity py*2+px), halo-tile kernel only (lb_upconv2x_halo_f16) */
    const float* ln_colsum;  /* LB_GEMM_LN_A: [N] fp32 column sums of W' */
    float ln_eps;            /* LB_GEMM_LN_A: LayerNorm epsilon */
    int reserved2_;
    float* ch_stats;         /* LB_GEMM_CH_STATS: output [N][row blocks] float2 (sum, sum of squares), else unused */
    int ch_stats_rows;       /* LB_GEMM_CH_STATS: row blocks per channel the caller's buffer holds = B * lb_gemm_ch_stat_rows();
                              * the launcher refuses any other value (the kernel's layout and the buffer cannot drift apart) */
} LbGemmParams;

int lb_gemm_f16(const LbGemmParams* params, void* stream);
long lb_gemm_workspace_bytes(int M, int N);
void lb_gemm_set_tuning(int tile, int splitk);   /* testing: force tile 1..5, 7 (192x128) or 9 (ping-pong 256x256, gemm_pp.hip) and split-K */
void lb_gemm_set_pp_auto(int on);                 /* A/B studies: 0 = the automatic tile policy never picks the ping-pong kernel (default 1) */
void lb_gemm_pp_set_group(int gm);                /* tuning: tile order of the ping-pong kernel inside an XCD's run: gm block rows down, then the next block column (0 = strip order) */
void lb_gemm_set_depth(int depth);               /* testing: 1 = one K-tile in flight, 0 = default ring */
int lb_gemm_plan(const LbGemmParams* p, int* tile, int* splitk, long* blocks); /* the tile (1..5) / split-K / grid lb_gemm_f16 would use; launches nothing */
/* 3x3 / stride 1 / pad 1 conv from an LDS-resident halo tile; same parameter block as lb_gemm_f16 (conv = 1,
 * KH = KW = 3, Cin % 64 == 0, Win % 16 == 0, zero_page set).  lb_gemm_f16 routes eligible convs here by itself
 * (lb_gemm_plan reports tile code 6); lb_gemm_set_halo: 0 = never, 1 = when the halo grid fills the chip
 * (default), 2 = whenever eligible. */
int lb_conv3x3_halo_f16(const LbGemmParams* params, void* stream);
/* 3x3 / stride 1 / pad 1 conv with N <= 16 output channels (conv_out of the VAE decoder / UNet): weights in registers,
 * one 16-column MFMA tile = the whole output width.  lb_gemm_f16 routes eligible convs here (lb_gemm_plan: tile code 8). */
int lb_conv3x3_narrow_f16(const LbGemmParams* params, void* stream);
/* Tuning: 1 (default) = persistent blocks (one per CU) whose operand request streams run across tile boundaries;
 * 0 = one (tile, channel block) item per block. */
void lb_conv_halo_set_persistent(int on);
/* Host arithmetic of a halo launch (no device work): kind 0 = not eligible, 3 = 3x3 form, 2 = 2x2 sub-pixel form; the
 * tile width (32 / 16), the number of (tile [, parity], channel block) work items and the grid that walks them. */
void lb_conv_halo_plan(const LbGemmParams* params, int* kind, int* tile_w, long* items, long* grid);
/* LB_GEMM_CH_STATS: row blocks PER SAMPLE of the statistics buffer ([N][B * rows] float2) the launch lb_gemm_f16 would make
 * for these parameters writes - from the same routing and the same tile constants as the launch itself; 0 = this problem does
 * not run on a halo-tile kernel (no statistics: the consumer runs the two-pass GroupNorm).  Launches nothing. */
int lb_gemm_ch_stat_rows(const LbGemmParams* params);
/* "nearest-2x upsample, then 3x3 conv" (UNet / VAE upsamplers) as ONE launch of the halo-tile kernel in its 2x2 sub-pixel
 * form: conv = 1, scatter = 2, KH = KW = 2, W = [4][N][4*Cin] stacked pre-summed kernels, C = [B][2H][2W][ldc]. */
int lb_upconv2x_halo_f16(const LbGemmParams* params, void* stream);
void lb_gemm_set_halo(int mode);
void lb_gemm_set_wide_store(int on);              /* tuning: 1 = 16-byte epilogue stores for fp16 row-major outputs (same results) */
void lb_gemm_set_t192_waves8(int on);             /* tuning: 1 (default) = the 192x128 tile runs 8 waves of 48x64 (tile code 10), 0 = 6 waves of 64x64 (tile code 7); same results */
void lb_gemm_set_kgroups(int on);                 /* tuning: 1 (default) = 64x64 grids of <= 200 unsplit blocks run two K-groups of 4 waves per block (tile code 11: partial sums added through LDS), 0 = never */
void lb_gemm_set_lean_epilogue(int on);           /* tuning: 1 (default) = one-round-trip tile epilogue where it applies, 0 = the per-row form everywhere (same results) */
void lb_gemm_set_variant(int variant, int stages); /* 0 = register ring, 1 = direct-to-LDS (stages 2..4, 0 = default), <0 = library default */

/* ---- normalisation (torch.nn.GroupNorm / LayerNorm inside the UNet / VAE modules reached
 *      from diffusers_holder.py:336 and :135) ------------------------------------------- */
/* x: [B][HW][ldx] fp16 (or fp32 when x_is_f32); y: [B][HW][ldy] fp16 = GN(x)*gamma+beta, then
 * SiLU when `silu`.  workspace: lb_groupnorm_workspace_bytes(B, groups). */
/* Experiment knob: input bytes per statistics+apply pair (sample groups sized for the Infinity Cache); default 0 = one
 * pair over the whole batch, which measured faster at every group size (profiles/r02_groupnorm_l3.txt). */
void lb_groupnorm_set_l3_chunk(long bytes);
/* GroupNorm whose statistics were left by the producing conv (LB_GEMM_CH_STATS): ch_stats = [C][B * stat_rows_per_sample]
 * float2; a small fold launch (float64, fixed order) + the same apply pass as lb_groupnorm_nhwc - x is read ONCE. */
int lb_groupnorm_from_stats(const void* x, void* y, const float* gamma, const float* beta, const float* ch_stats,
                            void* workspace, int B, int HW, int C, int ldx, int ldy, int groups, float eps, int silu,
                            int x_is_f32, int stat_rows_per_sample, void* stream);
long lb_groupnorm_workspace_bytes(int B, int groups);
int lb_groupnorm_nhwc(const void* x, void* y, const float* gamma, const float* beta, void* workspace,
                      int B, int HW, int C, int ldx, int ldy, int groups, float eps, int silu,
                      int x_is_f32, void* stream);
int lb_groupnorm_plan(int HW, int C, int groups, int x_is_f32);   /* 1 = lb_groupnorm_nhwc runs this shape as one launch (x read once), 0 = statistics + apply launches; host arithmetic only */
void lb_groupnorm_set_fused(int on);    /* testing: 1 (default) = lb_groupnorm_nhwc runs as ONE launch (slab in registers) wherever a (sample, lcm(8, C / groups)-channel slab) fits a block: <= 24 pixels per thread; 0 = always statistics + apply launches */
int lb_layernorm_f16(const void* x, void* y, const float* gamma, const float* beta, int M, int C,
                     int ldx, int ldy, float eps, void* stream);
void lb_layernorm_set_form(int form);   /* testing: 1 (default) = all loads of a row in flight + permlane / DPP reductions (round 6), 0 = the round-1 kernel; bit-identical results */

/* ---- attention (AttnProcessor2_0 / scaled_dot_product_attention inside the UNet call at
 *      diffusers_holder.py:336; VAE mid-block attention at :135) ------------------------ */
typedef struct LbAttnParams {
    const lb_half* Q;    /* element (b, q, h, d) at Q[(b*Sq + q)*ldq + h*D + d], D = the head dim of the entry point (64 / 512) */
    const lb_half* K;    /* element (b, k, h, d) at K[(b*Skv + k)*ldk + h*D + d] */
    const lb_half* V;    /* element (b, k, h, d) at V[(b*Skv + k)*ldv + h*D + d]  (token-major, like K) */
    lb_half* O;          /* like Q with ldo */
    int B, H, Sq, Skv, Skv_valid;   /* keys >= Skv_valid are masked (context padding 77 -> 80) */
    int ldq, ldk, ldv, ldo;         /* Q, K, V may be column slices of one fused [tokens][3C] projection */
    float scale;         /* 1/sqrt(D) */
    int causal;          /* 1: key k is visible to query q only when k <= q (CLIP text towers); needs Sq == Skv */
    const void* zero_page;          /* >= 16 zero bytes, 16-B aligned: source of the direct-to-LDS loads of rows >= Skv */
    int reserved_;                  /* set by the launcher */
} LbAttnParams;
int lb_attn_fwd_d64(const LbAttnParams* params, void* stream);
/* head dim 512, no causal form: the VAE decoder's mid-block attention (AutoencoderKL.decode, diffusers_holder.py:135) as ONE launch -
 * no S x S score matrix exists (the three-launch form scores GEMM -> lb_softmax_rows_f16 -> PV GEMM wrote 32 MB per sample at 512^2,
 * 512 MB at 1024^2) */
int lb_attn_fwd_d512(const LbAttnParams* params, void* stream);
void lb_attn_set_tuning(int force);   /* testing: 0 = by shape; bits 0-1 = query groups per wave (1 / 2), bit 4 = always stream 64-key tiles,
                                       * bit 5 = 5-stage ring for the streaming form (A/B knob),
                                       * bit 6 = the former two-stage form of the one-tile kernel, bit 7 = 8-byte output stores */
int lb_softmax_rows_f16(void* x, int M, int N, int ld, float scale, void* stream);

/* ---- small kernels ----------------------------------------------------------------- */
/* diffusers Timesteps(dim, flip_sin_to_cos=True, freq_shift=0): [cos | sin] */
int lb_sinusoid_f16(const float* vals_dev, int rows, int per_row, int val_stride, int dim, void* out,
                    int ld_out, int col_off, void* stream);
/* torch.cat along channels: dst[r][dst_off + c] = src[r][c] */
int lb_copy_cols_f16(const void* src, void* dst, long rows, int cols, int ld_src, int ld_dst,
                     int dst_off, void* stream);
int lb_cast_f16_to_f32(const void* x, void* y, long n, void* stream);
int lb_cast_f32_to_f16(const void* x, void* y, long n, float mul, void* stream);
int lb_nchw_to_nhwc_f16(const void* x, void* y, int B, int C, int HW, int ld, float mul, void* stream);
int lb_nhwc_to_nchw_f16(const void* x, void* y, int B, int C, int HW, int ld, void* stream);
/* VaeImageProcessor.postprocess (diffusers_holder.py:141): uint8 HWC frames on device */
int lb_postprocess_u8(const void* x, void* out_u8, long pixels, int ld, int x_is_f32, void* stream);
/* lpips.LPIPS(net='alex') pieces (blending_engine.py:744-758) */
int lb_lpips_prep_u8(const void* img_u8, void* out_f16, long pixels, void* stream);
int lb_maxpool3s2_nhwc_f16(const void* x, void* y, int N, int H, int W, int C, void* stream);
/* feats_a / feats_b: HOST arrays of npairs (<= 16) device pointers to [HW][C] fp16 features;
 * acc[pair] += tap distance; workspace: 16*128 floats */
int lb_lpips_tap(const void* const* feats_a, const void* const* feats_b, const float* lin, float* acc,
                 float* workspace, int npairs, int HW, int C, void* stream);
int lb_fill_f32(void* x, long n, float v, void* stream);
/* CLIP text towers behind pipe.encode_prompt (diffusers_holder.py:79-96): CLIPTextEmbeddings and the EOS-row gather */
int lb_embed_tokens_f16(const int* ids_dev, const void* tok_emb, const void* pos_emb, void* out, int rows, int seq, int C,
                        int vocab, void* stream);
int lb_gather_rows_f16(const void* src, const int* rows_idx_dev, void* out, int n, int C, int ld_src, void* stream);
/* Movie in-betweening (utils.py:166-176 add_frames_linear_interp -> :97 interpolate_linear on the uint8 key frames):
 * frames = [n_key][frame_bytes] uint8 on the device, out[k] = uint8((1 - w[k]) * frames[left[k]] + w[k] * frames[left[k] + 1])
 * in float64 as numpy >= 2 evaluates it, truncating cast.  frame_bytes % 16 == 0, n_out <= 65535. */
int lb_frames_lerp_u8(const void* frames, const int* left_dev, const double* w_dev, void* out, long n_out,
                      long frame_bytes, void* stream);
int lb_copy_d2d(void* dst, const void* src, long bytes, void* stream);

/* ---- launch programs (the MI355X-native stand-in for the reference's optional stable-fast
 *      compile, blending_engine.py:88-96): record the launchers above once, replay them from
 *      C++ or as one hipGraph ------------------------------------------------------------ */
void* lb_program_create(void);
void lb_program_destroy(void* prog);
int lb_program_begin_record(void* prog);   /* launchers called on this thread are recorded, not run */
int lb_program_end_record(void* prog);
int lb_program_num_ops(void* prog);
const char* lb_program_op_name(void* prog, int i);
int lb_program_run(void* prog, void* stream);                        /* eager replay */
int lb_program_run_range(void* prog, int begin, int end, void* stream);
int lb_program_instantiate(void* prog);                              /* capture into a hipGraph */
int lb_program_launch(void* prog, void* stream);                     /* graph launch (or eager) */
/* eager replay with hipEvents between ops; ms_out[num_ops]; synchronises (measurement only) */
int lb_program_time_ops(void* prog, void* stream, float* ms_out);

/* ---- STUDY BUILDS ONLY (hipcc -DLB_STUDY_BUILD: `python -m latentblending_amd.csrc.build --study` -> liblbhip_study.so,
 *      loaded with LB_HIP_LIBRARY=...).  These switches change what the kernels COMPUTE or how the tile policy decides; the
 *      product library neither exports them nor contains the code behind them. ------------------------------------------ */
#ifdef LB_STUDY_BUILD
void lb_slerp_set_study(int variant);        /* tools/slerp_study.py: 1 = lerp weights, 2 = fp32 sum (NOT the reference's arithmetic) */
void lb_conv_halo_set_study(int bits);       /* tools/halo_study.py: bit 0 skip the epilogue, bit 1 / 2 halo / weight requests from the zero page */
void lb_gemm_set_policy(int disable_mask);   /* tools/ab_policy.py: bit0 no 256x128, bit1 no 256x256, bit2/3 no 256x256 for conv/plain, bit4 no 256x128 for conv, bit5 general 192x128 rule, bit6 no narrow 192x128 rule */
#endif

#ifdef __cplusplus
}
#endif
#endif /* LB_HIP_H */
