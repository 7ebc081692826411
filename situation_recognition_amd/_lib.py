"""ctypes binding of libsrhip.so (the C ABI declared in include/srhip.h).

There is no fallback: if the shared library is missing or a call returns an error code
this module raises.  PyTorch is used only to own device memory and streams.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SR_LIB_PATH") or os.path.join(_HERE, "libsrhip.so")   # SR_LIB_PATH: diagnostic builds only

ABI_VERSION = 5                  # include/srhip.h: SR_ABI_VERSION
SR_F32, SR_BF16 = 0, 1
ROUTE_WS, ROUTE_C3D, ROUTE_STEM, ROUTE_C3D128, ROUTE_C3D256 = 16, 18, 19, 20, 21     # sr_conv_route codes
ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_TANH, ACT_SIGMOID_MUL, ACT_TANH_BLEND = range(6)
_ERR = {-1: "SR_ERR_ARG (bad shape/alignment/null pointer)", -2: "SR_ERR_DTYPE", -3: "SR_ERR_LAUNCH",
        -4: "SR_ERR_UNSUPPORTED"}


class SrError(RuntimeError):
    pass


class KPair(C.Structure):
    _fields_ = [("A", C.c_void_p), ("W", C.c_void_p), ("lda", C.c_int64), ("ldw", C.c_int64),
                ("K", C.c_int32), ("_pad", C.c_int32)]


class GemmArgs(C.Structure):
    _fields_ = [("kp", KPair * 3), ("npairs", C.c_int32), ("M", C.c_int32), ("N", C.c_int32), ("act", C.c_int32),
                ("C", C.c_void_p), ("ldc", C.c_int64), ("C2", C.c_void_p),
                ("bias", C.c_void_p), ("bias2", C.c_void_p), ("bias_scale", C.c_float), ("out_f32", C.c_int32),
                ("res", C.c_void_p), ("ldres", C.c_int64), ("aux1", C.c_void_p), ("aux2", C.c_void_p),
                ("stats", C.c_void_p)]


class ConvArgs(C.Structure):
    _fields_ = [("x", C.c_void_p), ("w", C.c_void_p),
                ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Cin", C.c_int32), ("Cout", C.c_int32),
                ("KH", C.c_int32), ("KW", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32), ("stem", C.c_int32),
                ("y", C.c_void_p), ("bias", C.c_void_p), ("res", C.c_void_p), ("act", C.c_int32), ("no_store", C.c_int32),
                ("stats", C.c_void_p), ("escale", C.c_void_p), ("in_scale", C.c_void_p), ("in_shift", C.c_void_p)]


class PairArgs(C.Structure):
    _fields_ = [("x", C.c_void_p), ("wpack", C.c_void_p), ("res", C.c_void_p), ("z", C.c_void_p), ("y", C.c_void_p),
                ("escale", C.c_void_p), ("eshift", C.c_void_p), ("in_scale", C.c_void_p), ("in_shift", C.c_void_p),
                ("stats", C.c_void_p), ("ybias", C.c_void_p), ("M", C.c_int64), ("Cmid", C.c_int32), ("Cexp", C.c_int32), ("Cred", C.c_int32),
                ("yrelu", C.c_int32)]


_P, _I, _L, _F, _U64 = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_uint64
SIGNATURES = {
    "sr_abi_version": [],
    "sr_set_cu_share": [_I],
    "sr_gemm": [C.POINTER(GemmArgs), _I, _P],
    "sr_gemm_stats_tiles": [_I, _I],
    "sr_gemm_tile_cfg": [_I, _I, _I, _I],
    "sr_debug_stamps": [_P, _I],
    "sr_conv2d": [C.POINTER(ConvArgs), _I, _P],
    "sr_conv_stats_rows": [C.POINTER(ConvArgs), _I],
    "sr_conv_route": [C.POINTER(ConvArgs), _I],
    "sr_conv_in_affine_supported": [C.POINTER(ConvArgs), _I],
    "sr_conv_pair_supported": [_L, _I, _I, _I, _I],
    "sr_conv_pair_pack_bytes": [_I, _I, _I],
    "sr_conv_pair_pack": [_P, _P, _P, _I, _I, _I, _I, _P],
    "sr_conv_pair_stats_rows": [_L, _I, _I, _I],
    "sr_conv_pair": [C.POINTER(PairArgs), _I, _P],
    "sr_stem_bn_relu_maxpool": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "sr_stem_prep": [_P, _P, _I, _I, _I, _I, _P],
    "sr_image_prep_u8": [_P, _P, _I, _I, _I, _I, _I, _P, _P, C.POINTER(C.c_float), C.POINTER(C.c_float), _I, _P],
    "sr_bn_finalize": [_P, _I, _I, _L, _P, _P, _P, _P, _F, _F, _P, _P, _P, _I, _P, _P, _F, _P],
    "sr_gram_plan": [_L, _I, C.POINTER(C.c_int64), C.POINTER(C.c_int64)],
    "sr_gram": [_P, _L, _I, _L, _I, _P, _L, _P],
    "sr_bn_apply_gram": [_P, _L, _I, _L, _I, _P, _P, _P, _L, _P],
    "sr_bn_gram": [_P, _L, _I, _L, _I, _P, _P, _P, _L, _P],
    "sr_bn_finalize_gram": [_P, _L, _I, _P, _L, _I, _I, _L, _P, _P, _P, _P, _F, _F, _P, _P, _P, _L, _P, _P, _F, _P],
    "sr_quantize_fp8": [_P, _P, _P, _P, _L, _I, _I, _F, _I, _P],
    "sr_conv3x3_fp8_stats_rows": [_I, _I],
    "sr_conv3x3_fp8": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "sr_bn_apply": [_P, _P, _P, _P, _P, _L, _I, _I, _I, _P],
    "sr_maxpool3x3s2": [_P, _P, _I, _I, _I, _I, _P, _P, _I, _P],
    "sr_avgpool": [_P, _P, _I, _I, _I, _I, _P],
    "sr_node_init_fwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P],
    "sr_node_init_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P],
    "sr_ggnn_aggregate": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P],
    "sr_gru_bwd1": [_P, _P, _P, _P, _P, _P, _P, _L, _I, _P],
    "sr_gru_bwd2": [_P, _P, _P, _P, _P, _L, _I, _P],
    "sr_transpose": [_P, _L, _P, _L, _L, _L, _I, _I, _P, _F, _P],
    "sr_gemm_tn_slices": [_L, _I, _I],
    "sr_gemm_tn": [_P, _L, _P, _L, _L, _I, _I, _I, _P, _I, _P, _L, _P],
    "sr_colsum": [_P, _L, _L, _L, _I, _P, _F, _P],
    "sr_cast_pad": [_P, _L, _P, _L, _L, _L, _L, _I, _I, _P],
    "sr_cast": [_P, _P, _L, _I, _I, _P],
    "sr_dropout_half": [_P, _P, _P, _L, _U64, _I, _P],
    "sr_comm_unique_id": [_P],
    "sr_comm_init": [_P, _I, _I, C.POINTER(C.c_void_p)],
    "sr_comm_world": [_P],
    "sr_allreduce_sum": [_P, _P, _L, _I, _P],
    "sr_comm_destroy": [_P],
}

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SrError("libsrhip.so not found at %s -- build it with `python situation_recognition_amd/csrc/build.py` "
                          "(there is no CPU fallback)" % LIB_PATH)
        l = C.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(l, name)          # AttributeError if the symbol is missing
            fn.argtypes = argtypes
            fn.restype = C.c_int
        if l.sr_abi_version() != ABI_VERSION:    # a stale .so would read a pointer where this binding passes a stream
            raise SrError("libsrhip.so at %s has ABI version %d, this binding expects %d: rebuild it "
                          "(python situation_recognition_amd/csrc/build.py --force)" % (LIB_PATH, l.sr_abi_version(), ABI_VERSION))
        _lib = l
    return _lib


def check(rc, what):
    if rc != 0:
        raise SrError("%s failed: %s" % (what, _ERR.get(rc, rc)))


def dtype_code(t):
    if t == torch.float32:
        return SR_F32
    if t == torch.bfloat16:
        return SR_BF16
    raise SrError("unsupported dtype %s (fp32 and bf16 only)" % t)


def ptr(t):
    return None if t is None else t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise SrError("situation_recognition_amd runs on an MI355X only: got a %s tensor (no CPU fallback)" % t.device)
        if t is not None and not t.is_contiguous():
            raise SrError("non-contiguous tensor passed to a HIP kernel")
