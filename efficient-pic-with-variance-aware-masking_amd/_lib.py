"""ctypes binding of libvampic.so (C ABI declared in ``include/vampic.h``).

There is no CPU fallback: if the shared library is missing, or an op is called
without a usable gfx950 device, this module raises — loudly.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VAMPIC_LIB", os.path.join(_HERE, "libvampic.so"))      # override: A/B kernel experiments

VAM_MAX_SEG = 4
VAM_MAX_GROUP = 8

# enum vam_act
ACT_NONE, ACT_GELU, ACT_LEAKY, ACT_HALF_TANH, ACT_SIGMOID, ACT_CLAMP01, ACT_RSQRT, ACT_SQRT, ACT_DOUBLE = range(9)
# enum vam_conv_flags
CONV_SQUARE_IN, CONV_PS2, CONV_OUT_NCHW, CONV_IN_BF3, CONV_OUT_BF3 = 1, 2, 4, 8, 16
CONV_W_BF16, CONV_IN_BF16, CONV_OUT_BF16, CONV_AUX_BF16 = 32, 64, 128, 256      # bf16-storage mode (BASELINE configs[2])
CONV_MUL_GELU_GRAD = 512         # training: the mul operand is a GELU's pre-activation z, the result is multiplied by gelu'(z)
# enum vam_pack_mode
PACK_CONV, PACK_DECONV5S2, PACK_PS2, PACK_GDN, PACK_CONV_DGRAD, PACK_GDN_T = range(6)
# enum vam_ew_op
(EW_GELU_FWD, EW_GELU_BWD, EW_GATE_BWD, EW_GDN_APPLY, EW_GDN_BWD_PREP, EW_GDN_BWD_FIN, EW_CLAMP_BWD, EW_AXPY, EW_GATE_FWD,
 EW_REPARAM_BWD, EW_HTANH_FWD, EW_HTANH_BWD, EW_MASK_SPLIT) = range(13)
# enum vam_family
FAM_CONV, FAM_ATTN, FAM_MASK, FAM_TAIL, FAM_MISC = range(5)
FAMILY_NAMES = ("conv_igemm", "win_attn", "variance_mask", "gauss_tail", "misc")
# launch classes of the event profiler (vam_prof_set_class): which part of the path a conv launch belongs to
PROF_CLASSES = ("other", "g_a", "g_s", "hyperprior", "stack_heads", "slice_chain", "lrp_prog", "rem")


class VamSeg(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("C", C.c_int32), ("ld", C.c_int32)]


VAM_MAX_WGRAD_GROUP = 16
VAM_MAX_EW_GROUP = 8
VAM_MAX_TAIL_GROUP = 8


class VamWgrad(C.Structure):
    _fields_ = [("x", C.c_void_p), ("dy", C.c_void_p), ("dw", C.c_void_p), ("db", C.c_void_p),
                ("ld_x", C.c_int), ("ld_dy", C.c_int), ("B", C.c_int), ("H", C.c_int), ("W", C.c_int),
                ("kh", C.c_int), ("kw", C.c_int), ("C", C.c_int), ("N", C.c_int),
                ("cin_total", C.c_int), ("c_off", C.c_int), ("stride", C.c_int), ("Hx", C.c_int), ("Wx", C.c_int),
                ("splits", C.c_int), ("workspace", C.c_void_p), ("slot_share", C.c_float), ("flags", C.c_int32)]


WGRAD_X_P3 = 1


class VamAux(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("ld", C.c_int32), ("pad_", C.c_int32)]


class VamEw(C.Structure):
    _fields_ = [("inp", VamAux * 4), ("out", VamAux * 3), ("n_pix", C.c_long), ("C", C.c_int32), ("flag", C.c_int32),
                ("coef", C.c_float), ("pad_", C.c_int32)]


class VamStackTail(C.Structure):
    _fields_ = [("x", C.c_void_p), ("w4", C.c_void_p), ("b4", C.c_void_p), ("w5", C.c_void_p), ("b5", C.c_void_p), ("out", C.c_void_p),
                ("post", VamAux), ("post2", VamAux), ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("x_groups", C.c_int32),
                ("ld_out", C.c_int32), ("act", C.c_int32)]


class VamConv(C.Structure):
    _fields_ = [
        ("seg", VamSeg * VAM_MAX_SEG),
        ("n_seg", C.c_int32),
        ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
        ("kh", C.c_int32), ("kw", C.c_int32),
        ("stride", C.c_int32),
        ("pad_y", C.c_int32), ("pad_x", C.c_int32),
        ("Ho", C.c_int32), ("Wo", C.c_int32),
        ("N", C.c_int32),
        ("wpack", C.c_void_p),
        ("bias", C.c_void_p),
        ("out", C.c_void_p),
        ("ldo", C.c_int32),
        ("Hf", C.c_int32), ("Wf", C.c_int32),
        ("osy", C.c_int32), ("osx", C.c_int32), ("ooy", C.c_int32), ("oox", C.c_int32),
        ("Cq", C.c_int32),
        ("act", C.c_int32),
        ("flags", C.c_int32),
        ("pre", VamAux), ("mul", VamAux), ("post", VamAux), ("post2", VamAux),
        ("in_amax", C.c_void_p * VAM_MAX_SEG), ("out_amax", C.c_void_p),
        ("preact", VamAux),
    ]


class VamPackJob(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("bias", C.c_int32), ("mode", C.c_int32), ("phase", C.c_int32),
                ("kh", C.c_int32), ("kw", C.c_int32), ("cin", C.c_int32), ("n", C.c_int32), ("pad_", C.c_int32)]


class VamResunit(C.Structure):
    _fields_ = [("x", C.c_void_p), ("out", C.c_void_p), ("ldx", C.c_int32), ("ldo", C.c_int32),
                ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("C", C.c_int32),
                ("w1", C.c_void_p), ("b1", C.c_void_p), ("w2", C.c_void_p), ("b2", C.c_void_p),
                ("w3", C.c_void_p), ("b3", C.c_void_p), ("flags", C.c_int32), ("pad_", C.c_int32)]


RESUNIT_BF16 = 1


_SIGNATURES = {
    # name: (restype, argtypes)
    "vam_last_error": (C.c_char_p, []),
    "vam_version": (C.c_int, []),
    "vam_device_info": (C.c_int, [C.c_char_p, C.POINTER(C.c_int)]),
    "vam_conv_struct_size": (C.c_size_t, []),
    "vam_conv_wpack_floats": (C.c_size_t, [C.c_int] * 4),
    "vam_pack_conv_weights": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "vam_conv_wpack_bf16_bytes": (C.c_size_t, [C.c_int] * 4),
    "vam_pack_conv_weights_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "vam_pack_bias": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "vam_conv_force_tile": (C.c_int, [C.c_int] * 3),
    "vam_conv_force_epilogue": (C.c_int, [C.c_int]),
    "vam_conv_last_tile": (C.c_int, [C.c_void_p] * 3),
    "vam_conv_set_mode": (C.c_int, [C.c_int]),
    "vam_conv_get_mode": (C.c_int, []),
    "vam_conv_group": (C.c_int, [C.POINTER(VamConv), C.c_int, C.c_void_p]),
    "vam_resunit_struct_size": (C.c_size_t, []),
    "vam_resunit_supported": (C.c_int, [C.c_int] * 3),
    "vam_resunit_group": (C.c_int, [C.POINTER(VamResunit), C.c_int, C.c_void_p]),
    "vam_resunit_set_dma": (C.c_int, [C.c_int]),
    "vam_resunit_set_debug": (C.c_int, [C.c_void_p]),
    "vam_s2d_input": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "vam_nchw_to_nhwc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "vam_nhwc_to_nchw": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "vam_win_attention": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p] + [C.c_int] * 7 + [C.c_void_p]),
    "vam_attn_set_mfma": (C.c_int, [C.c_int]),
    "vam_attn_mfma": (C.c_int, []),
    "vam_variance_mask": (C.c_int, [C.c_void_p, C.c_int, C.c_long, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double,
                                    C.c_void_p, C.c_int, C.c_long, C.c_long, C.c_void_p, C.c_void_p]),
    "vam_gauss_tail": (C.c_int, [C.c_void_p, C.c_int] * 5 + [C.c_void_p, C.c_int] * 3 + [C.c_void_p, C.c_int, C.c_long, C.c_int, C.c_void_p]),
    "vam_build_indexes": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_long, C.c_int, C.c_void_p]),
    "vam_eb_forward": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_long, C.c_void_p]),
    "vam_eb_aux_loss": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "vam_dequantize": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_long, C.c_int, C.c_void_p]),
    "vam_add": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_long, C.c_int, C.c_void_p]),
    "vam_memset_zero": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p]),
    "vam_pack_group": (C.c_int, [C.POINTER(VamPackJob), C.c_int, C.c_void_p]),
    "vam_absmax": (C.c_int, [C.POINTER(VamSeg), C.c_int, C.c_long, C.c_void_p, C.c_void_p]),
    "vam_sqdiff_sum": (C.c_int, [C.c_void_p, C.c_void_p, C.c_long, C.c_void_p, C.c_void_p]),
    "vam_eb_forward_noise": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_long, C.c_void_p, C.c_int, C.c_void_p]),
    "vam_ssim_level": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "vam_avgpool2": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "vam_conv_wgrad": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int] + [C.c_int] * 7 + [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "vam_conv_wgrad_group": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "vam_conv_wgrad_plan": (C.c_int, [C.c_void_p, C.c_void_p]),
    "vam_conv_wgrad_lds_grid": (C.c_int, [C.c_int, C.c_int]),
    "vam_colsum_workspace": (C.c_size_t, [C.c_long, C.c_int]),
    "vam_colsum": (C.c_int, [C.c_void_p, C.c_int, C.c_long, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "vam_leaky_bwd": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_long, C.c_int, C.c_void_p]),
    "vam_mul": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_long, C.c_int, C.c_void_p]),
    "vam_gauss_train": (C.c_int, [C.c_void_p, C.c_int] * 10 + [C.c_long, C.c_int, C.c_void_p]),
    "vam_train_elementwise": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p]),
    "vam_train_axpy_group": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "vam_stack_tail_group": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "vam_win_attention_bwd_workspace": (C.c_size_t, [C.c_int] * 5),
    "vam_win_attention_bwd": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
                              + [C.c_int] * 7 + [C.c_void_p]),
    "vam_eb_train_bwd": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                   C.c_void_p, C.c_long, C.c_void_p]),
    "vam_ps2_unshuffle": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "vam_upsample2_zero": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "vam_pmf_to_quantized_cdf": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "vam_rans_encode": (C.c_long, [C.c_void_p, C.c_void_p, C.c_long, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_long]),
    "vam_rans_decode": (C.c_int, [C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "vam_graph_begin": (C.c_int, [C.c_void_p]),
    "vam_graph_end": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "vam_graph_launch": (C.c_int, [C.c_void_p, C.c_void_p]),
    "vam_graph_destroy": (C.c_int, [C.c_void_p]),
    "vam_prof_enable": (C.c_int, [C.c_int]),
    "vam_prof_reset": (C.c_int, []),
    "vam_prof_read": (C.c_int, [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_long), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "vam_prof_set_class": (C.c_int, [C.c_int]),
    "vam_prof_read_class": (C.c_int, [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_long), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)


class VamError(RuntimeError):
    pass


_lib = None


def load():
    """Load libvampic.so and bind every symbol of include/vampic.h."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VamError(
            f"libvampic.so not found at {LIB_PATH}: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.vam_conv_struct_size() != C.sizeof(VamConv):
        raise VamError(f"ABI mismatch: sizeof(vam_conv) is {lib.vam_conv_struct_size()} in libvampic.so, {C.sizeof(VamConv)} in the binding")
    if lib.vam_resunit_struct_size() != C.sizeof(VamResunit):
        raise VamError(f"ABI mismatch: sizeof(vam_resunit) is {lib.vam_resunit_struct_size()} in libvampic.so, {C.sizeof(VamResunit)} in the binding")
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().vam_last_error().decode(errors="replace")
        raise VamError(f"{what or 'libvampic'} failed ({rc}): {msg}")


def require_gpu():
    lib = load()
    name = C.create_string_buffer(128)
    cus = C.c_int(0)
    rc = lib.vam_device_info(name, C.byref(cus))
    if rc != 0:
        raise VamError("no HIP device: the variance-aware-masking hot path runs only on gfx950 (MI355X); "
                       + lib.vam_last_error().decode(errors="replace"))
    return name.value.decode(), cus.value
