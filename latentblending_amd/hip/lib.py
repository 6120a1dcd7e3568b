"""ctypes binding of ``liblbhip.so`` (contract: ``include/lb_hip.h``).

The library is built in-tree by ``latentblending_amd/csrc/build.py`` (``__graft_entry__.build()``)
and loaded AFTER ``import torch`` so that both share one HIP runtime.  There is no fallback: a
missing library raises ``ImportError``; a failing launch raises ``RuntimeError`` with the
library's own error string.
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (must be imported first: shares libamdhip64 with the extension)

_HERE = os.path.dirname(os.path.abspath(__file__))
# LB_HIP_LIBRARY: an alternate build of the SAME library (kernel ablation studies, tools/gemm_ablate.py)
LIB_PATH = os.environ.get("LB_HIP_LIBRARY") or os.path.join(_HERE, "liblbhip.so")

c_void_pp = C.POINTER(C.c_void_p)


class LbGemmParams(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("W", C.c_void_p), ("C", C.c_void_p), ("bias", C.c_void_p),
        ("residual", C.c_void_p), ("rowvec", C.c_void_p), ("partial", C.c_void_p),
        ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
        ("lda", C.c_int), ("ldw", C.c_int), ("ldc", C.c_int), ("ldr", C.c_int),
        ("ld_rowvec", C.c_int), ("rows_per_batch", C.c_int),
        ("alpha", C.c_float), ("flags", C.c_int),
        ("conv", C.c_int), ("Hin", C.c_int), ("Win", C.c_int), ("Cin", C.c_int),
        ("Hout", C.c_int), ("Wout", C.c_int), ("KH", C.c_int), ("KW", C.c_int),
        ("stride", C.c_int), ("pad", C.c_int), ("ups", C.c_int), ("ldx", C.c_int),
        ("splitk", C.c_int), ("zero_page", C.c_void_p),
        ("scatter", C.c_int), ("sc_py", C.c_int), ("sc_px", C.c_int), ("reserved_", C.c_int),
        ("ln_colsum", C.c_void_p), ("ln_eps", C.c_float), ("reserved2_", C.c_int), ("ch_stats", C.c_void_p), ("ch_stats_rows", C.c_int),
    ]


class LbAttnParams(C.Structure):
    _fields_ = [
        ("Q", C.c_void_p), ("K", C.c_void_p), ("V", C.c_void_p), ("O", C.c_void_p),
        ("B", C.c_int), ("H", C.c_int), ("Sq", C.c_int), ("Skv", C.c_int), ("Skv_valid", C.c_int),
        ("ldq", C.c_int), ("ldk", C.c_int), ("ldv", C.c_int), ("ldo", C.c_int),
        ("scale", C.c_float), ("causal", C.c_int), ("zero_page", C.c_void_p), ("reserved_", C.c_int),
    ]


GEMM_OUT_F32, GEMM_RES_F32, GEMM_GEGLU, GEMM_TRANS_OUT, GEMM_SILU, GEMM_RELU, GEMM_LN_A = 1, 2, 4, 8, 16, 32, 64
GEMM_QUICK_GELU, GEMM_GELU, GEMM_CH_STATS = 128, 256, 512

_vp, _i, _l, _f, _d = C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_double

# name -> (restype, argtypes); launchers (restype int) are wrapped with error checking
SIGNATURES = {
    "lb_version": (_i, []),
    "lb_last_error_string": (C.c_char_p, []),
    "lb_device_info": (_i, [_i, C.POINTER(_i), C.POINTER(_l), C.c_char_p, _i]),
    "lb_slerp_pairs_f16": (_i, [c_void_pp, c_void_pp, c_void_pp, C.POINTER(_d), _i, _l, _vp]),
    "lb_slerp_pairs_f32": (_i, [c_void_pp, c_void_pp, c_void_pp, C.POINTER(_d), _i, _l, _vp]),
    "lb_slerp_pairs_f64": (_i, [c_void_pp, c_void_pp, c_void_pp, C.POINTER(_d), _i, _l, _vp]),
    "lb_slerp_batched_f16": (_i, [_vp, _vp, _vp, _vp, _l, _l, _vp]),
    "lb_slerp_strided_f16": (_i, [_vp, _l, _vp, _l, _vp, _vp, _l, _l, _vp]),
    "lb_lerp_f16": (_i, [_vp, _vp, _vp, _l, _d, _vp]),
    "lb_lerp_f32": (_i, [_vp, _vp, _vp, _l, _d, _vp]),
    "lb_scale_model_input_f16": (_i, [_vp, _vp, _vp, _l, _i, _i, _vp]),
    "lb_euler_step_f16": (_i, [_vp, _vp, _vp, _vp, _vp, _l, _i, _i, _i, _vp]),
    "lb_ddim_step_f16": (_i, [_vp, _vp, _vp, _vp, _l, _i, _i, _vp]),
    "lb_gemm_f16": (_i, [C.POINTER(LbGemmParams), _vp]),
    "lb_gemm_workspace_bytes": (_l, [_i, _i]),
    "lb_gemm_set_tuning": (None, [_i, _i]),
    "lb_gemm_set_depth": (None, [_i]),
    "lb_gemm_set_variant": (None, [_i, _i]),
    "lb_gemm_set_wide_store": (None, [_i]),
    "lb_gemm_set_lean_epilogue": (None, [_i]),
    "lb_gemm_set_t192_waves8": (None, [_i]),
    "lb_gemm_set_kgroups": (None, [_i]),
    "lb_gemm_pp_set_group": (None, [_i]),
    "lb_gemm_set_pp_auto": (None, [_i]),
    "lb_gemm_set_halo": (None, [_i]),
    "lb_conv3x3_halo_f16": (_i, [C.POINTER(LbGemmParams), _vp]),
    "lb_conv3x3_narrow_f16": (_i, [C.POINTER(LbGemmParams), _vp]),
    "lb_upconv2x_halo_f16": (_i, [C.POINTER(LbGemmParams), _vp]),
    "lb_conv_halo_set_persistent": (None, [_i]),
    "lb_gemm_ch_stat_rows": (_i, [C.POINTER(LbGemmParams)]),
    "lb_conv_halo_plan": (None, [C.POINTER(LbGemmParams), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_long), C.POINTER(C.c_long)]),
    "lb_gemm_plan": (_i, [C.POINTER(LbGemmParams), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_long)]),
    "lb_groupnorm_workspace_bytes": (_l, [_i, _i]),
    "lb_groupnorm_set_l3_chunk": (None, [_l]),
    "lb_groupnorm_nhwc": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _i, _i, _vp]),
    "lb_groupnorm_from_stats": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _i, _i, _i, _vp]),
    "lb_layernorm_f16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp]),
    "lb_attn_fwd_d64": (_i, [C.POINTER(LbAttnParams), _vp]),
    "lb_attn_fwd_d512": (_i, [C.POINTER(LbAttnParams), _vp]),
    "lb_attn_set_tuning": (None, [_i]),
    "lb_layernorm_set_form": (None, [_i]),
    "lb_groupnorm_set_fused": (None, [_i]),
    "lb_groupnorm_plan": (_i, [_i, _i, _i, _i]),
    "lb_softmax_rows_f16": (_i, [_vp, _i, _i, _i, _f, _vp]),
    "lb_sinusoid_f16": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _i, _vp]),
    "lb_copy_cols_f16": (_i, [_vp, _vp, _l, _i, _i, _i, _i, _vp]),
    "lb_cast_f16_to_f32": (_i, [_vp, _vp, _l, _vp]),
    "lb_cast_f32_to_f16": (_i, [_vp, _vp, _l, _f, _vp]),
    "lb_nchw_to_nhwc_f16": (_i, [_vp, _vp, _i, _i, _i, _i, _f, _vp]),
    "lb_nhwc_to_nchw_f16": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "lb_postprocess_u8": (_i, [_vp, _vp, _l, _i, _i, _vp]),
    "lb_lpips_prep_u8": (_i, [_vp, _vp, _l, _vp]),
    "lb_maxpool3s2_nhwc_f16": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "lb_lpips_tap": (_i, [c_void_pp, c_void_pp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "lb_fill_f32": (_i, [_vp, _l, _f, _vp]),
    "lb_embed_tokens_f16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "lb_frames_lerp_u8": (_i, [_vp, _vp, _vp, _vp, _l, _l, _vp]),
    "lb_gather_rows_f16": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "lb_copy_d2d": (_i, [_vp, _vp, _l, _vp]),
    "lb_program_create": (_vp, []),
    "lb_program_destroy": (None, [_vp]),
    "lb_program_begin_record": (_i, [_vp]),
    "lb_program_end_record": (_i, [_vp]),
    "lb_program_num_ops": (_i, [_vp]),
    "lb_program_op_name": (C.c_char_p, [_vp, _i]),
    "lb_program_run": (_i, [_vp, _vp]),
    "lb_program_run_range": (_i, [_vp, _i, _i, _vp]),
    "lb_program_instantiate": (_i, [_vp]),
    "lb_program_launch": (_i, [_vp, _vp]),
    "lb_program_time_ops": (_i, [_vp, _vp, C.POINTER(C.c_float)]),
}

# exported only by study builds (-DLB_STUDY_BUILD, liblbhip_study.so): switches that change the arithmetic / the tile policy
STUDY_SIGNATURES = {
    "lb_slerp_set_study": (None, [_i]),
    "lb_gemm_set_policy": (None, [_i]),
    "lb_conv_halo_set_study": (None, [_i]),
}

_NO_CHECK = {"lb_version", "lb_last_error_string", "lb_gemm_workspace_bytes",
             "lb_groupnorm_workspace_bytes", "lb_groupnorm_set_l3_chunk", "lb_conv_halo_set_persistent", "lb_conv_halo_plan", "lb_gemm_ch_stat_rows", "lb_conv_halo_set_study", "lb_gemm_set_tuning", "lb_gemm_set_depth", "lb_gemm_set_variant", "lb_gemm_set_wide_store", "lb_gemm_set_lean_epilogue", "lb_gemm_set_t192_waves8", "lb_gemm_set_kgroups", "lb_gemm_pp_set_group", "lb_gemm_set_pp_auto", "lb_gemm_set_policy", "lb_gemm_set_halo", "lb_attn_set_tuning", "lb_layernorm_set_form", "lb_groupnorm_set_fused", "lb_groupnorm_plan", "lb_slerp_set_study", "lb_program_create",
             "lb_program_destroy", "lb_program_num_ops", "lb_program_op_name"}


def _load():
    if not os.path.isfile(LIB_PATH):
        raise ImportError(
            f"latentblending_amd: {LIB_PATH} is missing. Build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` (hipcc, gfx950). "
            "There is no CPU fallback for the kernel path.")
    return C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)


_lib = _load()


class _Api:
    pass


def _wrap(name, fn):
    def checked(*args):
        rc = fn(*args)
        if rc != 0:
            msg = _lib.lb_last_error_string()
            raise RuntimeError(f"{name} failed ({rc}): {msg.decode() if msg else '?'}")
        return rc
    checked.__name__ = name
    return checked


api = _Api()
for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(_lib, _name)          # AttributeError here == header/library mismatch
    _fn.restype = _res
    _fn.argtypes = _args
    setattr(api, _name, _fn if _name in _NO_CHECK else _wrap(_name, _fn))
for _name, (_res, _args) in STUDY_SIGNATURES.items():
    _fn = getattr(_lib, _name, None)
    if _fn is not None:
        _fn.restype = _res
        _fn.argtypes = _args
        setattr(api, _name, _fn)

if os.environ.get("LB_GEMM_WIDE_STORE"):        # tuning experiments (tools/): 16-byte epilogue stores
    api.lb_gemm_set_wide_store(int(os.environ["LB_GEMM_WIDE_STORE"]))
if os.environ.get("LB_GEMM_LEAN_EPILOGUE"):     # A/B studies (tools/): 0 = the per-row tile epilogue everywhere
    api.lb_gemm_set_lean_epilogue(int(os.environ["LB_GEMM_LEAN_EPILOGUE"]))
if os.environ.get("LB_GEMM_PP_AUTO"):           # A/B studies (tools/): 0 = the tile policy never picks the ping-pong GEMM
    api.lb_gemm_set_pp_auto(int(os.environ["LB_GEMM_PP_AUTO"]))
