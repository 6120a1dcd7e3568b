// Shared GEMM parameter block (public struct from include/lb_hip.h) + the common epilogue.
#pragma once
#include "lb_common.h"
#include "../../include/lb_hip.h"

typedef _Float16 lb_h2x __attribute__((ext_vector_type(2)));

// v = 4 consecutive output columns n..n+3 of row m (fp32 accumulators).
__device__ __forceinline__ void lb_gemm_store4(const LbGemmParams& p, int m, int n, int bidx, f32x4 v) {
    float o[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = v[r] * p.alpha;
    if (p.bias) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(p.bias + n);
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] += b[r];
    }
    if (p.rowvec) {
        const f16x4 t = *reinterpret_cast<const f16x4*>(
            reinterpret_cast<const f16*>(p.rowvec) + (long)bidx * p.ld_rowvec + n);
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] += (float)t[r];
    }
    if (p.residual) {
        if (p.flags & LB_GEMM_RES_F32) {
            const f32x4 q = *reinterpret_cast<const f32x4*>((const float*)p.residual + (long)m * p.ldr + n);
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] += q[r];
        } else {
            const f16x4 q = *reinterpret_cast<const f16x4*>((const f16*)p.residual + (long)m * p.ldr + n);
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] += (float)q[r];
        }
    }
    if (p.flags & LB_GEMM_SILU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = lb_silu(o[r]);
    }
    if (p.flags & LB_GEMM_RELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = fmaxf(o[r], 0.f);
    }
    if (p.flags & (LB_GEMM_QUICK_GELU | LB_GEMM_GELU)) {
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (p.flags & LB_GEMM_GELU) ? lb_gelu_erf(o[r]) : lb_quick_gelu(o[r]);
    }
    long crow = m;                      // output row; sub-pixel convs scatter to the 2x-upsampled grid
    if (p.scatter) {
        const int hw = p.Hout * p.Wout;
        const int b = m / hw, rem = m - b * hw;
        const int y = rem / p.Wout, x = rem - y * p.Wout;
        crow = ((long)b * 2 * p.Hout + 2 * y + p.sc_py) * (2 * p.Wout) + 2 * x + p.sc_px;
    }
    if (p.flags & LB_GEMM_TRANS_OUT) {
        f16* c = (f16*)p.C;
#pragma unroll
        for (int r = 0; r < 4; ++r) c[(long)(n + r) * p.ldc + m] = (f16)o[r];
    } else if (p.flags & LB_GEMM_OUT_F32) {
        *reinterpret_cast<f32x4*>((float*)p.C + crow * p.ldc + n) = (f32x4){o[0], o[1], o[2], o[3]};
    } else {
        *reinterpret_cast<f16x4*>((f16*)p.C + crow * p.ldc + n) =
            (f16x4){(f16)o[0], (f16)o[1], (f16)o[2], (f16)o[3]};
    }
}

// ------------------------------------------------------------------------------------------------
// Tile epilogue shared by both GEMM kernels.  Lane (g = lane>>4, l16 = lane&15) of a wave owns rows
// row0 + 16 i (i < TM; row0 already contains l16) and the 4-column groups col0 + 16 j (j < TN; col0
// already contains 4 g).  Every global load of a row (bias, residual, time-embedding row vector) is
// issued branch-free, with clamped addresses, BEFORE the first dependent store: with divergent `continue`s around the loads the compiler must keep each 4-output group's
// load -> wait -> store chain separate, i.e. one memory round trip per group (measured in situ: +30 % on
// the residual / bias GEMMs of the UNet against the same launches without epilogue operands).
// GEGLU: accumulator column pairs (2 jp, 2 jp + 1) are (h, gate) of output column gcol0 + 16 jp.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void lb_gemm_write4(const LbGemmParams& p, long crow, int m, int n, const float (&o)[4]) {
    if (p.flags & LB_GEMM_TRANS_OUT) {
        f16* c = (f16*)p.C;
#pragma unroll
        for (int r = 0; r < 4; ++r) c[(long)(n + r) * p.ldc + m] = (f16)o[r];
    } else if (p.flags & LB_GEMM_OUT_F32) {
        *reinterpret_cast<f32x4*>((float*)p.C + crow * p.ldc + n) = (f32x4){o[0], o[1], o[2], o[3]};
    } else {
        *reinterpret_cast<f16x4*>((f16*)p.C + crow * p.ldc + n) =
            (f16x4){(f16)o[0], (f16)o[1], (f16)o[2], (f16)o[3]};
    }
}

// Row statistics of a fused LayerNorm (LB_GEMM_LN_A): mean / rstd of the lane's TM output rows.
template <int TM> struct LbLnRows { float mean[TM], rstd[TM]; };

// RowFn: i -> global output row of the lane's i-th 16-row group (row0 + 16 i for the GEMM kernels; the pixel
// index of a 2-D spatial tile for the halo conv kernel).
// 32 values per lane -> their sums over the 16 lanes of a DPP row (lanes that share lane >> 4), two per lane, by a
// halving butterfly: in each of four steps a lane keeps one half of its values, sends the other half to its partner and
// adds what the partner sent - 16 + 8 + 4 + 2 exchanges instead of 32 x 4 for "reduce every value everywhere" (the
// ds_bpermute form of __shfl_xor made this epilogue cost 5-12 % of a halo conv, the plain DPP form still ~5 %:
// profiles/r03_gn_stats_fusion.txt).  Partners: lane ^ 1 and lane ^ 2 by quad permutes, then the neighbouring quad by
// row_ror:4 (quads of opposite parity keep opposite halves, so the rotation delivers exactly the half its receiver
// keeps; after it quads {0, 3} / {2, 1} are folded) and lane ^ 8 by row_ror:8.  With l = lane & 15 the lane ends up
// holding values k = (l & 1) << 4 | (l >> 1 & 1) << 3 | (l >> 2 & 1) << 2 | (l >> 3) << 1 | {0, 1}.
template <int CTRL> __device__ __forceinline__ float lb_dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ void lb_row16_reduce32(const float (&v)[32], float (&out)[2]) {
    const int l = threadIdx.x & 15;
    const bool b0 = l & 1, b1 = l & 2, b2 = l & 4, b3 = l & 8;
    float w16[16], w8[8], w4[4];
#pragma unroll
    for (int k = 0; k < 16; ++k) w16[k] = (b0 ? v[k + 16] : v[k]) + lb_dpp_mov<0xB1>(b0 ? v[k] : v[k + 16]);      // quad_perm [1, 0, 3, 2]
#pragma unroll
    for (int k = 0; k < 8; ++k) w8[k] = (b1 ? w16[k + 8] : w16[k]) + lb_dpp_mov<0x4E>(b1 ? w16[k] : w16[k + 8]);   // quad_perm [2, 3, 0, 1]
#pragma unroll
    for (int k = 0; k < 4; ++k) w4[k] = (b2 ? w8[k + 4] : w8[k]) + lb_dpp_mov<0x124>(b2 ? w8[k] : w8[k + 4]);      // row_ror:4
#pragma unroll
    for (int k = 0; k < 2; ++k) out[k] = (b3 ? w4[k + 2] : w4[k]) + lb_dpp_mov<0x128>(b3 ? w4[k] : w4[k + 2]);     // row_ror:8
}

// CHST (LB_GEMM_CH_STATS, halo conv kernels): per output column, (sum, sum of squares) over the wave's 16 TM rows of the
// values this epilogue STORES (after the fp16 rounding when the output is fp16) go to chst[n * chst_ld] (float2: the statistics
// buffer is CHANNEL-major, [N][row blocks], so that the fold kernel reads a group's channels as contiguous runs): each lane sums its
// TM rows, the 16 lanes that share a column quad (l16 = 0..15) fold them with a DPP butterfly (fixed order), two values per lane.
template <int TM, int TN, bool GEGLU, bool LNA, bool CHST = false, typename RowFn>
__device__ __forceinline__ void lb_gemm_tile_epilogue_rows_ln(const LbGemmParams& p, const f32x4 (&acc)[TM][TN],
                                                           RowFn row_of, int col0, int gcol0, const LbLnRows<TM>* ln,
                                                           float2* chst = nullptr, long chst_ld = 1) {
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    if (GEGLU) {
        constexpr int TP = TN / 2 > 0 ? TN / 2 : 1;
        const int half = p.N / 2;
        f32x4 bh[TP], bg[TP];
#pragma unroll
        for (int jp = 0; jp < TP; ++jp) {
            const int n = gcol0 + jp * 16;
            const int nc = n < half ? n : 0;
            bh[jp] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + nc) : zero4;
            bg[jp] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + half + nc) : zero4;
        }
        f32x4 chs[LNA ? TP : 1], cgs[LNA ? TP : 1];      // LN_A column sums of the lane's column groups: loaded once per tile
        if (LNA) {                                       // (round 5; inside the row loop every load also waited for the stores before it)
#pragma unroll
            for (int jp = 0; jp < TP; ++jp) {
                const int n = gcol0 + jp * 16;
                const int nc = n < half ? n : 0;
                chs[jp] = *reinterpret_cast<const f32x4*>(p.ln_colsum + nc);
                cgs[jp] = *reinterpret_cast<const f32x4*>(p.ln_colsum + half + nc);
            }
        }
        // WIDE stores (see below): output column groups jp and jp + 1 paired through v_permlane16_swap -> 16-byte stores
        const bool gwide = TP % 2 == 0 && (p.reserved2_ & 1) && (p.ldc & 7) == 0 &&
                           (gcol0 - 4 * ((threadIdx.x & 63) >> 4)) + 16 * TP <= half;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = row_of(i);
            if (!gwide && m >= p.M) continue;
            const bool m_ok = m < p.M;
            unsigned glo[2] = {0u, 0u};
#pragma unroll
            for (int jp = 0; jp < TP; ++jp) {
                const int n = gcol0 + jp * 16;
                if (!gwide && n >= half) continue;
                const f32x4 ch = LNA ? chs[LNA ? jp : 0] : zero4, cg = LNA ? cgs[LNA ? jp : 0] : zero4;
                f16x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float ah = acc[i][2 * jp][r], ag = acc[i][(2 * jp + 1) % TN][r];
                    if (LNA) {
                        ah = (ah - ln->mean[i] * ch[r]) * ln->rstd[i];
                        ag = (ag - ln->mean[i] * cg[r]) * ln->rstd[i];
                    }
                    const float h = ah * p.alpha + bh[jp][r];
                    const float gt = ag * p.alpha + bg[jp][r];
                    o[r] = (f16)(h * lb_gelu_erf(gt));
                }
                if (gwide) {
                    const unsigned u0 = __builtin_bit_cast(unsigned, (lb_h2x){o[0], o[1]}), u1 = __builtin_bit_cast(unsigned, (lb_h2x){o[2], o[3]});
                    if ((jp & 1) == 0) {
                        glo[0] = u0;
                        glo[1] = u1;
                    } else {
                        const auto r0 = __builtin_amdgcn_permlane16_swap(glo[0], u0, false, false);
                        const auto r1 = __builtin_amdgcn_permlane16_swap(glo[1], u1, false, false);
                        const int nst = (((threadIdx.x & 63) >> 4) & 1) ? n - 4 : n - 16;
                        if (m_ok) {
                            typedef unsigned lb_u4g __attribute__((ext_vector_type(4)));
                            *reinterpret_cast<lb_u4g*>((f16*)p.C + (long)m * p.ldc + nst) = (lb_u4g){r0[0], r1[0], r0[1], r1[1]};
                        }
                    }
                    continue;
                }
                *reinterpret_cast<f16x4*>((f16*)p.C + (long)m * p.ldc + n) = o;
            }
        }
        return;
    }
    float cs_s[CHST ? TN : 1][4], cs_q[CHST ? TN : 1][4];
    if (CHST) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) cs_s[j][r] = cs_q[j][r] = 0.f;
    }
    constexpr bool CS_HOIST = LNA && TN <= 4;   // (the 256-wide tiles have no 32 registers to spare: they keep the per-group loads)
    f32x4 cs_v[CS_HOIST ? TN : 1];      // LN_A column sums: once per tile (see the GEGLU branch)
    if (CS_HOIST) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = col0 + j * 16;
            cs_v[j] = *reinterpret_cast<const f32x4*>(p.ln_colsum + (n < p.N ? n : 0));
        }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = row_of(i);
        const bool m_ok = m < p.M;
        const int mc = m_ok ? m : p.M - 1;
        f32x4 add[TN];                  // bias + row vector + residual of this row's TN column groups
#pragma unroll
        for (int j = 0; j < TN; ++j) {  // (re-read per row: L1 hits in the same request batch, no registers held)
            const int n = col0 + j * 16;
            add[j] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + (n < p.N ? n : 0)) : zero4;
        }
        if (p.rowvec) {
            const f16* rv = reinterpret_cast<const f16*>(p.rowvec) + (long)(mc / p.rows_per_batch) * p.ld_rowvec;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = col0 + j * 16;
                const f16x4 t = *reinterpret_cast<const f16x4*>(rv + (n < p.N ? n : 0));
#pragma unroll
                for (int r = 0; r < 4; ++r) add[j][r] += (float)t[r];
            }
        }
        if (p.residual) {
            if (p.flags & LB_GEMM_RES_F32) {
                const float* rr = (const float*)p.residual + (long)mc * p.ldr;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int n = col0 + j * 16;
                    add[j] += *reinterpret_cast<const f32x4*>(rr + (n < p.N ? n : 0));
                }
            } else {
                const f16* rr = (const f16*)p.residual + (long)mc * p.ldr;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int n = col0 + j * 16;
                    const f16x4 q = *reinterpret_cast<const f16x4*>(rr + (n < p.N ? n : 0));
#pragma unroll
                    for (int r = 0; r < 4; ++r) add[j][r] += (float)q[r];
                }
            }
        }
        long crow = m;                  // output row; sub-pixel convs scatter to the 2x-upsampled grid
        if (p.scatter) {
            const int hw = p.Hout * p.Wout;
            const int b = mc / hw, rem = mc - b * hw;
            const int y = rem / p.Wout, x = rem - y * p.Wout;
            crow = ((long)b * 2 * p.Hout + 2 * y + p.sc_py) * (2 * p.Wout) + 2 * x + p.sc_px;
        }
        // WIDE stores (p.reserved2_ & 1, fp16 row-major outputs whose wave column range is entirely inside N): the lane's
        // quads of column groups j and j + 1 are paired with the neighbouring 16-lane row through v_permlane16_swap, so
        // that every lane owns 8 CONSECUTIVE halves - 16-byte stores, half as many store instructions per tile.
        const bool wide = TN % 2 == 0 && (p.reserved2_ & 1) && !(p.flags & (LB_GEMM_TRANS_OUT | LB_GEMM_OUT_F32)) &&
                          (p.ldc & 7) == 0 && (col0 - 4 * ((threadIdx.x & 63) >> 4)) + 16 * TN <= p.N;
        unsigned wide_lo[2] = {0u, 0u};
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = col0 + j * 16;
            // (a use of the loaded bias / residual on EVERY path: on the masked path below they would otherwise stay
            // "pending" in the compiler's wait bookkeeping, and a kernel that calls this epilogue inside a loop - the
            // persistent halo conv - then gets a compiler-inserted vmcnt(0) at the head of its MFMA loop)
            asm volatile("" ::"v"(add[j]));
            if (!wide && (!m_ok || n >= p.N)) continue;
            float o[4];
            f32x4 cs = zero4;
            if (CS_HOIST) cs = cs_v[CS_HOIST ? j : 0];
            else if (LNA) cs = *reinterpret_cast<const f32x4*>(p.ln_colsum + n);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float a = acc[i][j][r];
                if (LNA) a = (a - ln->mean[i] * cs[r]) * ln->rstd[i];
                o[r] = a * p.alpha + add[j][r];
            }
            if (p.flags & LB_GEMM_SILU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = lb_silu(o[r]);
            }
            if (p.flags & LB_GEMM_RELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = fmaxf(o[r], 0.f);
            }
            if (p.flags & (LB_GEMM_QUICK_GELU | LB_GEMM_GELU)) {
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = (p.flags & LB_GEMM_GELU) ? lb_gelu_erf(o[r]) : lb_quick_gelu(o[r]);
            }
            if (CHST) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float h = (p.flags & LB_GEMM_OUT_F32) ? o[r] : (float)(f16)o[r];   // the value a GroupNorm pass would read back
                    cs_s[j][r] += h;
                    cs_q[j][r] += h * h;
                }
            }
            if (wide) {
                const f16x4 h4 = {(f16)o[0], (f16)o[1], (f16)o[2], (f16)o[3]};
                const unsigned u0 = __builtin_bit_cast(unsigned, (lb_h2x){h4[0], h4[1]}), u1 = __builtin_bit_cast(unsigned, (lb_h2x){h4[2], h4[3]});
                if ((j & 1) == 0) {
                    wide_lo[0] = u0;
                    wide_lo[1] = u1;
                } else {
                    // swap odd rows (g odd) of the (j-1) quad with even rows of the j quad: even g ends with columns
                    // 16 (j-1) + 4 g .. + 7, odd g with columns 16 j + 4 (g-1) .. + 7
                    const auto r0 = __builtin_amdgcn_permlane16_swap(wide_lo[0], u0, false, false);
                    const auto r1 = __builtin_amdgcn_permlane16_swap(wide_lo[1], u1, false, false);
                    const int g_ = (threadIdx.x & 63) >> 4;
                    const int nst = (g_ & 1) ? n - 4 : n - 16;
                    if (m_ok) {
                        typedef unsigned lb_u4 __attribute__((ext_vector_type(4)));
                        *reinterpret_cast<lb_u4*>((f16*)p.C + crow * p.ldc + nst) = (lb_u4){r0[0], r1[0], r0[1], r1[1]};
                    }
                }
                continue;
            }
            lb_gemm_write4(p, crow, m, n, o);
        }
    }
    if constexpr (CHST) {
        static_assert(TN == 4, "the channel-statistics epilogue folds 2 x 16 column values per lane");
        float v[32], red[2];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[j * 4 + r] = cs_s[j][r];
                v[16 + j * 4 + r] = cs_q[j][r];
            }
        lb_row16_reduce32(v, red);
        const int l = threadIdx.x & 15;
        const int which = l & 1, j = ((l >> 1) & 1) * 2 + ((l >> 2) & 1), r0 = (l >> 3) * 2;     // the two values this lane holds
        const int n = col0 + j * 16 + r0;
        if (n < p.N) {          // (N % 4 == 0: columns n and n + 1 are valid together)
            float* f = reinterpret_cast<float*>(chst);        // chst = &stats[0][row block]; channel n lives chst_ld float2 further on
            f[2 * (long)n * chst_ld + which] = red[0];
            f[2 * (long)(n + 1) * chst_ld + which] = red[1];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// LEAN epilogue (round 5): the same arithmetic as above for the launches that carry the programs' time - fp16 row-major
// output through the 16-byte stores, alpha, bias, and EITHER a time-embedding row vector OR an fp16 residual (optionally
// the channel statistics of the halo convs) - with ONE memory round trip per tile and 32-bit addressing.
//
// Why.  vmcnt retires in order and counts stores as well as loads on gfx950, so a load issued after a store also waits
// for that store's acknowledgement.  The general epilogue above loads a row's operands, waits, stores the row and goes on
// to the next row: TM dependent (load + store-acknowledge) round trips per tile - 4-6 us per 256-pixel tile of the halo
// conv (profiles/r02_halo_study.txt: a third of an 18-step Cin = 128 tile) and most of the ~13 us a 28 us projection GEMM
// spends outside its K loop (M 4352, N 1280: K 1280 / 2560 / 5120 = 28.1 / 43.1 / 76.0 us, profiles/r04_gemm_bench_call2.txt).
// Here every global load of the TILE is issued back to back before the first store (bias and row vector once per column
// group - the row vector of a tile that lies inside one sample is the same for all of its rows -, the residual for all
// TM x TN groups), and the stores then stream without a wait.  Operands are addressed as wave-uniform base pointer (SGPRs)
// + 32-bit lane offset: one VGPR and no 64-bit multiplies per access, which is what lets the 254-register halo conv hold
// the preloaded residual at all.  Same expressions, same order of the additions as the general form: bit-identical.
// In-place residuals (residual == C) stay correct: a wave reads all of its tile before it writes any of it.
//
// Returns false (nothing done; the caller runs the general epilogue) for every other flag / shape combination.
//   row_of(i)  logical row m of the lane's i-th 16-row group (row mask, residual row); row_of(i) >= row_lo
//   row_lo     wave-uniform: smallest logical row of the wave tile; row_hi: one past its largest (row masks only if row_hi > M)
//   out_rel(i) output row of row_of(i) minus out_lo (>= 0)      out_lo  wave-uniform origin of the output rows
//   batch      wave-uniform sample index of the tile for the row vector, -1 = its rows may span samples
//   colw       wave-uniform first column of the wave tile (the lane's quad of group j is colw + 16 j + 4 g)
//   SCATTERED  the caller's out_rel already implements p.scatter (halo conv); otherwise p.scatter != 0 is declined
// ------------------------------------------------------------------------------------------------
// A wave-uniform GLOBAL pointer in scalar registers (readfirstlane of both halves; address space 1, so that accesses
// through it are global_* instructions with the base in SGPRs and a 32-bit lane offset, not flat_* on a 64-bit VGPR pair).
typedef __attribute__((address_space(1))) char lb_gchar;
__device__ __forceinline__ lb_gchar* lb_uniform_ptr(const void* ptr) {
    const unsigned long long v = (unsigned long long)ptr;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (lb_gchar*)(((unsigned long long)hi << 32) | lo);
}
template <typename T> __device__ __forceinline__ T lb_gload(const lb_gchar* base, unsigned lane_off, int imm) {
    return *reinterpret_cast<const __attribute__((address_space(1))) T*>(base + (size_t)lane_off + imm);
}
template <typename T> __device__ __forceinline__ void lb_gstore(lb_gchar* base, unsigned lane_off, int imm, T v) {
    *reinterpret_cast<__attribute__((address_space(1))) T*>(base + (size_t)lane_off + imm) = v;
}

template <int TM, int TN, bool CHST, bool MASKED, typename RowFn, typename OutFn>
__device__ __forceinline__ void lb_gemm_tile_epilogue_lean_body(const LbGemmParams& p, const f32x4 (&acc)[TM][TN], RowFn row_of,
                                                                int row_lo, OutFn out_rel, long out_lo, int batch, int colw,
                                                                float2* chst, long chst_ld) {
    typedef unsigned lb_u4w __attribute__((ext_vector_type(4)));
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    // (the lane id goes through an opaque register: everything derived from it is tile-invariant, and inside the persistent
    // halo conv the compiler would otherwise hoist those values out of the tile loop, hold them across the MFMA loop and spill)
    int lane_ = threadIdx.x & 63;
    asm volatile("" : "+v"(lane_));
    const int g = lane_ >> 4;
    const unsigned cq = 4u * g;                                             // the lane's quad inside a 16-column group
    const unsigned sc = ((g & 1) ? 16u : 0u) + ((g & 2) ? 8u : 0u);         // its 8 stored columns of a group pair start here
    const bool has_res = p.residual != nullptr;
    lb_gchar* const cb = lb_uniform_ptr((f16*)p.C + out_lo * p.ldc + colw);
    const lb_gchar* const rb = lb_uniform_ptr((const f16*)p.residual + (has_res ? (long)row_lo * p.ldr + colw : 0));
    // ---- phase A: every global load of the tile ----
    unsigned ro[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        int m = row_of(i);
        if (MASKED) m = m < p.M ? m : p.M - 1;                              // (row_lo < M: checked by the caller)
        ro[i] = ((unsigned)(m - row_lo) * (unsigned)p.ldr + cq) * 2u;
    }
    f32x4 addv[TN];                                                         // bias (+ row vector) of the lane's column groups
#pragma unroll
    for (int j = 0; j < TN; ++j) addv[j] = zero4;
    if (p.bias) {
        const lb_gchar* const bb = lb_uniform_ptr(p.bias + colw);
#pragma unroll
        for (int j = 0; j < TN; ++j) addv[j] = lb_gload<f32x4>(bb, cq * 4u, j * 64);
    }
    f16x4 rv[TN];
    if (p.rowvec) {
        const lb_gchar* const vb = lb_uniform_ptr(reinterpret_cast<const f16*>(p.rowvec) + (long)batch * p.ld_rowvec + colw);
#pragma unroll
        for (int j = 0; j < TN; ++j) rv[j] = lb_gload<f16x4>(vb, cq * 2u, j * 32);
    }
    f16x4 pre[TM][TN];
    if (has_res) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) pre[i][j] = lb_gload<f16x4>(rb, ro[i], j * 32);
    }
    __builtin_amdgcn_sched_barrier(0);                                      // loads above; arithmetic and stores below
    if (p.rowvec) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) addv[j][r] += (float)rv[j][r];
    }
    // a use of every loaded value on every path, here: the one wait of the epilogue (a value still "pending" at the end
    // of a masked path would make the compiler drain vmcnt at the head of the persistent halo conv's MFMA loop)
#pragma unroll
    for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(addv[j]));
    if (has_res) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(pre[i][j]));
    }
    // ---- phase B: arithmetic and stores ----
    float cs_s[CHST ? TN : 1][4], cs_q[CHST ? TN : 1][4];
    if (CHST) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) cs_s[j][r] = cs_q[j][r] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const bool ok = !MASKED || row_of(i) < p.M;
        const unsigned co = ((unsigned)out_rel(i) * (unsigned)p.ldc + sc) * 2u;
        unsigned lo0 = 0u, lo1 = 0u;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            f32x4 add = addv[j];
            if (has_res) {
#pragma unroll
                for (int r = 0; r < 4; ++r) add[r] += (float)pre[i][j][r];
            }
            float o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float a = acc[i][j][r];
                o[r] = a * p.alpha + add[r];
            }
            // (the fp32 result goes through an opaque register: hipcc would otherwise fuse "fma, then round to fp16" into
            // v_fma_mix{lo,hi}_f16, which rounds ONCE - measured: not the bits of the general form's v_pk_fma_f32 + v_cvt)
            asm volatile("" : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]));
            const f16x4 h4 = {(f16)o[0], (f16)o[1], (f16)o[2], (f16)o[3]};
            if (CHST) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float h = (float)h4[r];                           // the value a GroupNorm pass would read back
                    cs_s[j][r] += h;
                    cs_q[j][r] += h * h;
                }
            }
            const unsigned u0 = __builtin_bit_cast(unsigned, (lb_h2x){h4[0], h4[1]}), u1 = __builtin_bit_cast(unsigned, (lb_h2x){h4[2], h4[3]});
            if ((j & 1) == 0) {
                lo0 = u0;
                lo1 = u1;
            } else {            // (the exchange of the general epilogue's WIDE stores)
                const auto r0 = __builtin_amdgcn_permlane16_swap(lo0, u0, false, false);
                const auto r1 = __builtin_amdgcn_permlane16_swap(lo1, u1, false, false);
                if (ok) lb_gstore<lb_u4w>(cb, co, (j >> 1) * 64, (lb_u4w){r0[0], r1[0], r0[1], r1[1]});
            }
        }
    }
    if constexpr (CHST) {
        static_assert(TN == 4, "the channel-statistics epilogue folds 2 x 16 column values per lane");
        float v[32], red[2];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[j * 4 + r] = cs_s[j][r];
                v[16 + j * 4 + r] = cs_q[j][r];
            }
        lb_row16_reduce32(v, red);
        const int l = lane_ & 15;
        const int which = l & 1, j = ((l >> 1) & 1) * 2 + ((l >> 2) & 1), r0 = (l >> 3) * 2;     // the two values this lane holds
        const int n = colw + 4 * g + j * 16 + r0;
        float* f = reinterpret_cast<float*>(chst);            // chst = &stats[0][row block]; channel n lives chst_ld float2 further on
        f[2 * (long)n * chst_ld + which] = red[0];
        f[2 * (long)(n + 1) * chst_ld + which] = red[1];
    }
}

template <int TM, int TN, bool CHST, bool SCATTERED, typename RowFn, typename OutFn>
__device__ __forceinline__ bool lb_gemm_tile_epilogue_lean(const LbGemmParams& p, const f32x4 (&acc)[TM][TN], RowFn row_of, int row_lo_,
                                                           int row_hi_, OutFn out_rel, long out_lo_, int batch_, int colw_,
                                                           float2* chst = nullptr, long chst_ld = 1) {
    static_assert(TN % 2 == 0, "the lean epilogue stores column-group pairs");
    // (wave-uniform by contract; through readfirstlane so that the compiler keeps them - and the branches below - scalar)
    const int row_lo = __builtin_amdgcn_readfirstlane(row_lo_), row_hi = __builtin_amdgcn_readfirstlane(row_hi_);
    const int batch = __builtin_amdgcn_readfirstlane(batch_), colw = __builtin_amdgcn_readfirstlane(colw_);
    const long out_lo = (long)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned long long)out_lo_ >> 32)) << 32) |
                               (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned long long)out_lo_));
    constexpr int UNSUPPORTED = LB_GEMM_TRANS_OUT | LB_GEMM_OUT_F32 | LB_GEMM_SILU | LB_GEMM_RELU | LB_GEMM_QUICK_GELU | LB_GEMM_GELU |
                                LB_GEMM_RES_F32 | LB_GEMM_GEGLU | LB_GEMM_LN_A;
    const bool eligible = !(p.flags & UNSUPPORTED) && (p.reserved2_ & 3) == 3 && (p.ldc & 7) == 0 && (p.ldr & 3) == 0 && colw + 16 * TN <= p.N &&
                          (SCATTERED || p.scatter == 0) && !(p.rowvec != nullptr && (p.residual != nullptr || batch < 0)) &&
                          (CHST || !(p.flags & LB_GEMM_CH_STATS));
    if (!eligible) return false;
    if (row_lo >= p.M) return true;                             // (a wave tile entirely below the last row)
    if (row_hi <= p.M)
        lb_gemm_tile_epilogue_lean_body<TM, TN, CHST, false>(p, acc, row_of, row_lo, out_rel, out_lo, batch, colw, chst, chst_ld);
    else
        lb_gemm_tile_epilogue_lean_body<TM, TN, CHST, true>(p, acc, row_of, row_lo, out_rel, out_lo, batch, colw, chst, chst_ld);
    return true;
}

template <int TM, int TN, bool GEGLU, typename RowFn>
__device__ __forceinline__ void lb_gemm_tile_epilogue_rows(const LbGemmParams& p, const f32x4 (&acc)[TM][TN],
                                                           RowFn row_of, int col0, int gcol0) {
    lb_gemm_tile_epilogue_rows_ln<TM, TN, GEGLU, false>(p, acc, row_of, col0, gcol0, (const LbLnRows<TM>*)nullptr);
}

template <int TM, int TN, bool GEGLU>
__device__ __forceinline__ void lb_gemm_tile_epilogue(const LbGemmParams& p, const f32x4 (&acc)[TM][TN],
                                                      int row0, int col0, int gcol0) {
    lb_gemm_tile_epilogue_rows<TM, TN, GEGLU>(p, acc, [row0](int i) { return row0 + i * 16; }, col0, gcol0);
}
