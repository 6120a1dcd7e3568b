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
                f32x4 ch = zero4, cg = zero4;       // (LN_A column sums: re-read per row, L1 hits, no registers held)
                if (LNA) {
                    ch = *reinterpret_cast<const f32x4*>(p.ln_colsum + n);
                    cg = *reinterpret_cast<const f32x4*>(p.ln_colsum + half + n);
                }
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
            if (LNA) cs = *reinterpret_cast<const f32x4*>(p.ln_colsum + n);
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
