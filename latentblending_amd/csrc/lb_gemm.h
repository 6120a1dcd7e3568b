// Shared GEMM parameter block (public struct from include/lb_hip.h) + the common epilogue.
#pragma once
#include "lb_common.h"
#include "../../include/lb_hip.h"

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
