// Common device/host helpers for the gfx950 kernels of liblbhip.so.
// Written for CDNA4 only: wave = 64 lanes, MFMA 16x16x32 f16, 160 KiB LDS per CU.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#define LB_WAVE 64

typedef _Float16 f16;
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Last launch error is kept per process (C-ABI: lb_last_error_string()).
extern "C" const char* lb_last_error_string(void);
void lb_set_error(const char* what, hipError_t e);

// ---- launch recording (program.hip) --------------------------------------------------------
#ifdef __cplusplus
#include <functional>
#include <mutex>
bool lb_recording();
void lb_record(const char* name, std::function<int(hipStream_t)> fn);
// Body of every extern "C" launcher: record a closure while a program is recording, otherwise
// launch on the caller's stream.  CALL is an expression using `s` (hipStream_t); everything it
// names is captured BY VALUE, so pointer-to-host arguments must be copied into locals first.
#define LB_DISPATCH(NAME, CALL)                                                   \
    do {                                                                          \
        if (lb_recording()) {                                                     \
            lb_record(NAME, [=](hipStream_t s) -> int { return CALL; });          \
            return 0;                                                             \
        }                                                                         \
        hipStream_t s = (hipStream_t)stream;                                      \
        return CALL;                                                              \
    } while (0)
// Statement form: STMT is a kernel-launch statement using `s`; the launch status is returned.
#define LB_DISPATCH_STMT(NAME, STMT) \
    LB_DISPATCH(NAME, ([&]() -> int { STMT; return lb_check_launch(NAME); })())
#endif

static inline int lb_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) lb_set_error(what, e);
    return (int)e;
}

// "Once per DEVICE" guard for per-kernel attributes (hipFuncSetAttribute applies to the current device only: a process
// that drives several GPUs must set it on each).  `seen` is the call site's own static bit mask (<= 64 devices).
//     LB_ONCE_PER_DEVICE(seen) hipFuncSetAttribute(...);
// The guard object holds one process-wide mutex for as long as the guarded statement runs: a second host thread that arrives while
// the first is still inside hipFuncSetAttribute WAITS for it instead of launching with the attribute unset (round 5's atomic
// test-and-set let the loser through at once: a launch asking for more than 64 KiB of dynamic LDS could then fail on that thread).
struct LbFirstCallOnDevice {
    std::unique_lock<std::mutex> lock;
    bool first;
    explicit LbFirstCallOnDevice(unsigned long long& seen) : lock(mutex()) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        const unsigned long long bit = 1ull << (dev & 63);
        first = (seen & bit) == 0;
        seen |= bit;
    }
    static std::mutex& mutex() {
        static std::mutex m;
        return m;
    }
};
#define LB_ONCE_PER_DEVICE(seen) if (LbFirstCallOnDevice lb_once_guard_{seen}; lb_once_guard_.first)

#define LB_REQUIRE(cond, what)                                   \
    do {                                                         \
        if (!(cond)) {                                           \
            lb_set_error(what, hipErrorInvalidValue);            \
            return (int)hipErrorInvalidValue;                    \
        }                                                        \
    } while (0)

__device__ __forceinline__ float lb_wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, LB_WAVE);
    return v;
}

// The same xor butterfly (32, 16, 8, 4, 2, 1: every lane ends with the same bits as lb_wave_sum - float addition commutes, so
// "v[i] + v[partner]" does not depend on who fetched whom) without the LDS crossbar: permlane32 / permlane16 swaps for the two
// widest stages, DPP row rotations / quad permutes for the rest (a rotation by 8, then by 4, reaches the xor partner's VALUE
// because lanes i and i ^ 8 already agree after the stage before).  12 ds_bpermute round trips -> 6 VALU-class exchanges.
__device__ __forceinline__ float lb_wave_sum_dpp(float v) {
    {
        const unsigned u = __float_as_uint(v);
        const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
        v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    {
        const unsigned u = __float_as_uint(v);
        const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
        v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x128, 0xf, 0xf, false));   // row_ror:8
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x124, 0xf, 0xf, false));   // row_ror:4
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x4e, 0xf, 0xf, false));    // quad_perm [2,3,0,1]
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0xb1, 0xf, 0xf, false));    // quad_perm [1,0,3,2]
    return v;
}

__device__ __forceinline__ double lb_wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, LB_WAVE);
    return v;
}

__device__ __forceinline__ float lb_wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, LB_WAVE));
    return v;
}

// f64 -> f16 with a single rounding: round-to-odd f64 -> f32, then RN f32 -> f16 (13 spare bits).  The round-to-odd
// value is built from the ROUND-TO-NEAREST conversion (one v_cvt_f32_f64) and the exact residual x - f: if the
// residual points towards zero, f overshot and the truncated value is one ulp below in magnitude; any non-zero
// residual sets the sticky LSB.  (__double2float_rz has no single-instruction form here: hipcc emulated it with ~7
// float64 operations per element, which made the batched slerp VALU-bound at 2.8 TB/s.)
__device__ __forceinline__ f16 lb_f64_to_f16(double x) {
    const float f = (float)x;
    const double r = x - (double)f;                                  // exact (Sterbenz / representable difference)
    unsigned u = __float_as_uint(f);
    const bool finite_nz = (u & 0x7fffffffu) - 1u < 0x7f7fffffu;      // f is neither 0, inf nor NaN
    if (finite_nz && r != 0.0) {
        if ((r < 0.0) != (f < 0.0f)) u -= 1u;                        // residual opposes f: truncate the magnitude
        u |= 1u;
    }
    return (f16)__uint_as_float(u);
}

// SiLU / erf-GELU for epilogues and bandwidth-bound passes: raw v_exp_f32 / v_rcp_f32 (1 ulp), no IEEE division
// sequence, no libm call.  x -> -inf: exp2 -> inf, rcp -> 0, x * 0 = -0;  x -> +inf: exp2 -> 0, result x.
__device__ __forceinline__ float lb_silu(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44269504088896340736f));
}

// erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, far below the fp16 output rounding): branch-free,
// 1 rcp + 1 exp2 + 8 FMA-class ops instead of the ~40-instruction libm erff with its two branches.
__device__ __forceinline__ float lb_erf(float x) {
    const float ax = __builtin_fabsf(x);
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, ax, 1.0f));
    float p = __builtin_fmaf(1.061405429f, t, -1.453152027f);
    p = __builtin_fmaf(p, t, 1.421413741f);
    p = __builtin_fmaf(p, t, -0.284496736f);
    p = __builtin_fmaf(p, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(ax * ax * -1.44269504088896340736f);
    const float r = __builtin_fmaf(-p * t, e, 1.0f);
    return __builtin_copysignf(r, x);
}

// CLIP's "quick_gelu": x * sigmoid(1.702 x)
__device__ __forceinline__ float lb_quick_gelu(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * (-1.702f * 1.44269504088896340736f)));
}

__device__ __forceinline__ float lb_gelu_erf(float x) {
    return 0.5f * x * (1.0f + lb_erf(x * 0.70710678118654752440f));
}
