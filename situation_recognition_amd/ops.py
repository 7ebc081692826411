"""Tensor-level wrappers over the C ABI (include/srhip.h).  Each function checks shapes on the host,
allocates outputs with torch (device memory only) and enqueues the HIP kernel on the current stream.
Nothing here computes on the CPU or with torch operators."""
import ctypes as C
import os
import threading

import torch

from . import _lib as L
from ._lib import (ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_SIGMOID_MUL, ACT_TANH, ACT_TANH_BLEND,  # noqa: F401
                   check, dtype_code, lib, ptr, require_gpu, stream)


# When set to a list, every conv2d / gemm / aggregate launch is bracketed by HIP events on the launch
# stream and (kernel tag, start, end, algorithmic flops, algorithmic bytes) is appended (bench.py roofline leg).
PROFILE = None


def _timed(tag, flops, nbytes, launch):
    if PROFILE is None:
        return launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(torch.cuda.current_stream())
    rc = launch()
    e1.record(torch.cuda.current_stream())
    PROFILE.append((tag, e0, e1, flops, nbytes))
    return rc


def _capturing():
    """Inside a hipGraph capture the per-stream scratch caches are bypassed: every capture runs on the same capture stream, so a
    cached buffer would be shared by graphs that are later replayed concurrently on different streams (the two backbones)."""
    return torch.cuda.is_current_stream_capturing()


def _f32(t, name):
    if t is not None and t.dtype != torch.float32:
        raise L.SrError("%s must be fp32" % name)
    return t


def gemm(pairs, N=None, bias=None, bias_scale=1.0, bias2=None, act=ACT_NONE, res=None, aux1=None, aux2=None,
         out=None, out2=None, out_f32=False, stats=None):
    """out[M,N] = epilogue(sum_p A_p @ W_p.T + bias_scale*bias + bias2 (+res)).  A_p: [M,K_p], W_p: [>=N, K_p]
    (nn.Linear layout).  Rows may be strided (last dim contiguous)."""
    A0, W0 = pairs[0]
    M = A0.shape[0]
    N = W0.shape[0] if N is None else N
    dt = A0.dtype
    odt = torch.float32 if (out_f32 or dt == torch.float32) else dt
    a = L.GemmArgs()
    for i, (A, W) in enumerate(pairs):
        if A.dtype != dt or W.dtype != dt or A.shape[0] != M or W.shape[0] < N or A.shape[1] != W.shape[1]:
            raise L.SrError("gemm operand %d: shape/dtype mismatch %s %s" % (i, tuple(A.shape), tuple(W.shape)))
        if A.stride(1) != 1 or W.stride(1) != 1 or not A.is_cuda or not W.is_cuda:
            raise L.SrError("gemm operands must be CUDA tensors with contiguous rows")
        a.kp[i] = L.KPair(A.data_ptr(), W.data_ptr(), A.stride(0), W.stride(0), A.shape[1], 0)
    if out is None:
        ldc = (N + 7) // 8 * 8                   # rows stay 16-byte aligned for the vector epilogue
        buf = torch.empty((M, ldc), device=A0.device, dtype=odt)
        out = buf[:, :N] if ldc != N else buf
    if out.dtype != odt or out.shape[0] != M or out.shape[1] != N or out.stride(1) != 1:
        raise L.SrError("gemm: bad output tensor")
    two = act in (ACT_SIGMOID_MUL, ACT_TANH_BLEND)
    if two and out2 is None:
        out2 = torch.empty_like(out)
    for t in (res, aux1, aux2, out2):
        if t is not None and (t.dtype != odt or t.shape != out.shape or t.stride(1) != 1):
            raise L.SrError("gemm: res/aux/out2 must match the output's shape and dtype")
    for t in (aux1, aux2, out2):
        if t is not None and t.stride(0) != out.stride(0):
            raise L.SrError("gemm: aux/out2 row stride must equal the output's")
    a.npairs, a.M, a.N, a.act = len(pairs), M, N, act
    a.C, a.ldc, a.C2 = out.data_ptr(), out.stride(0), ptr(out2)
    a.bias, a.bias2, a.bias_scale = ptr(_f32(bias, "bias")), ptr(_f32(bias2, "bias2")), float(bias_scale)
    a.out_f32 = int(odt == torch.float32 and dt != torch.float32)
    a.res, a.ldres = ptr(res), (res.stride(0) if res is not None else 0)
    a.aux1, a.aux2, a.stats = ptr(aux1), ptr(aux2), ptr(_f32(stats, "stats"))
    ktot = sum(A.shape[1] for A, _ in pairs)
    tag = "gemm_gate" if act in (ACT_SIGMOID, ACT_SIGMOID_MUL, ACT_TANH_BLEND, ACT_TANH) else "gemm"
    check(_timed(tag, 2.0 * M * N * ktot, 0, lambda: lib().sr_gemm(C.byref(a), dtype_code(dt), stream())), "sr_gemm")
    return (out, out2) if two else out


_CU_SHARE = threading.local()


def set_cu_share(share):
    """Persistent grids of the launches that follow FROM THIS THREAD are sized for 1/share of the chip (sr_set_cu_share); returns the
    previous value."""
    rc = lib().sr_set_cu_share(int(share))
    check(min(rc, 0), "sr_set_cu_share")
    _CU_SHARE.value = int(share)
    return rc


def _default_cu_share():
    try:
        v = int(os.environ.get("SR_CU_SHARE", "1"))          # (the library reads the same variable for its process default)
    except ValueError:
        v = 1
    return v if 1 <= v <= 8 else 1


def cu_share():
    """The share this thread's launches are sized for -- the EFFECTIVE value: a thread that never called set_cu_share and one that set
    it back to the default report the same number.  Part of the key of every captured hipGraph, whose grids are baked in at capture
    time (with None for "never set" the first overlapped forward changed the key of an identical configuration and every graph was
    captured a second time, the old one kept alive with its static buffers)."""
    return getattr(_CU_SHARE, "value", None) or _default_cu_share()


def stats_tiles(M, N):
    return lib().sr_gemm_stats_tiles(int(M), int(N))


def _conv_geometry(a, x, Cout, KH, stride, pad, res, relu):
    a.B, a.H, a.W, a.Cin, a.Cout = x.shape[0], x.shape[1], x.shape[2], x.shape[3], Cout
    a.KH, a.KW, a.stride, a.pad, a.stem = KH, KH, stride, pad, 0
    a.res, a.act = ptr(res), (ACT_RELU if relu else ACT_NONE)


def conv_in_affine_supported(x, Cout, KH, stride, pad, res, relu, want_stats=False):
    """Will conv2d(x, ..., in_affine=...) be served (the kernels that normalise their input on load: expansion 1x1, layer1's 3x3)?"""
    a = L.ConvArgs()
    _conv_geometry(a, x, Cout, KH, stride, pad, res, relu)
    a.stats = x.data_ptr() if want_stats else None                 # (only tested against NULL)
    return lib().sr_conv_in_affine_supported(C.byref(a), dtype_code(x.dtype)) == 1


def conv_route(B, H, W_, Cin, Cout, KH, stride, pad, dtype=torch.bfloat16, res=False, relu=False, bias=False, escale=False,
               want_stats=False, stats_only=False, in_affine=False, stem=False):
    """The kernel sr_conv2d runs this launch on (sr_conv_route: 0/1/2/4 = generic implicit GEMM on that tile shape, L.ROUTE_* = the
    specialised kernels).  Geometry only: no tensor is needed, nothing is launched."""
    a = L.ConvArgs()
    dummy = 0x1000                                                   # non-null, 16-byte aligned; never dereferenced by the dry run
    a.x, a.w, a.y = dummy, dummy, dummy
    a.B, a.H, a.W, a.Cin, a.Cout = B, H, W_, Cin, Cout
    a.KH, a.KW, a.stride, a.pad, a.stem = KH, KH, stride, pad, int(stem)
    a.res, a.act = (dummy if res else None), (ACT_RELU if relu else ACT_NONE)
    a.bias, a.escale = (dummy if bias else None), (dummy if escale else None)
    a.stats, a.no_store = (dummy if (want_stats or stats_only) else None), int(stats_only)
    if in_affine:
        a.in_scale = a.in_shift = dummy
    rc = lib().sr_conv_route(C.byref(a), dtype_code(dtype))
    check(min(rc, 0), "sr_conv_route")
    return rc


def conv2d(x, w, Cout, KH, stride, pad, bias=None, res=None, relu=False, want_stats=False, stem_hw=None, escale=None,
           stats_only=False, out=None, in_affine=None):
    """x: NHWC [B,H,W,Cin] (or, with stem_hw=(H,W), the padded NHWC4 image from stem_prep);
    w: packed [Cout, KH*KW*Cin] (stem: [Cout, 256]).  Returns y [B,Ho,Wo,Cout] (and stats partials).
    in_affine = (scale, shift) over the INPUT channels: the convolution runs on relu(x*scale + shift) (see conv_in_affine_supported;
    an unsupported launch raises -- the affine is never dropped)."""
    require_gpu(x, w, bias, res)
    a = L.ConvArgs()
    B = x.shape[0]
    if stem_hw is not None:
        H, W_, Cin = stem_hw[0], stem_hw[1], 3
        Ho, Wo = (H + 6 - 7) // 2 + 1, (W_ + 6 - 7) // 2 + 1
    else:
        H, W_, Cin = x.shape[1], x.shape[2], x.shape[3]
        Ho, Wo = (H + 2 * pad - KH) // stride + 1, (W_ + 2 * pad - KH) // stride + 1
    if stats_only:
        want_stats, y = True, None
    else:
        y = out if out is not None else torch.empty((B, Ho, Wo, Cout), device=x.device, dtype=x.dtype)
    a.x, a.w, a.B, a.H, a.W, a.Cin, a.Cout = x.data_ptr(), w.data_ptr(), B, H, W_, Cin, Cout
    a.KH, a.KW, a.stride, a.pad, a.stem = KH, KH, stride, pad, int(stem_hw is not None)
    a.res, a.escale = ptr(res), ptr(_f32(escale, "escale"))
    stats = None
    if want_stats:
        rows = lib().sr_conv_stats_rows(C.byref(a), dtype_code(x.dtype))
        if rows <= 0:
            raise L.SrError("conv2d: no statistics layout for this launch")
        stats = torch.empty((rows, 2, Cout), device=x.device, dtype=torch.float32)
    a.y = x.data_ptr() if stats_only else y.data_ptr()          # never written when stats_only
    a.bias, a.res, a.act, a.stats = ptr(_f32(bias, "bias")), ptr(res), (ACT_RELU if relu else ACT_NONE), ptr(stats)
    a.escale, a.no_store = ptr(_f32(escale, "escale")), int(stats_only)
    if in_affine is not None:
        require_gpu(*in_affine)
        a.in_scale, a.in_shift = _f32(in_affine[0], "in_scale").data_ptr(), _f32(in_affine[1], "in_shift").data_ptr()
    if res is not None and (tuple(res.shape) != (B, Ho, Wo, Cout) or res.dtype != x.dtype):
        raise L.SrError("conv2d: residual shape/dtype mismatch")
    flops = 2.0 * B * Ho * Wo * Cout * KH * KH * Cin
    es = x.element_size()
    # algorithmic bytes: input and weights read once, output written once, residual read once
    nbytes = es * (x.numel() + w.numel()) + (0 if stats_only else es * B * Ho * Wo * Cout) + (es * res.numel() if res is not None else 0)
    # a statistics-only launch re-does work the storing launch also does: its TIME counts, its FLOPs are not algorithmic
    tag = "conv7x7" if stem_hw is not None else "conv%dx%d" % (KH, KH)
    check(_timed(tag, 0.0 if stats_only else flops, float(nbytes), lambda: lib().sr_conv2d(C.byref(a), dtype_code(x.dtype), stream())), "sr_conv2d")
    if stats_only:
        return stats
    return (y, stats) if want_stats else y


def conv_pair_supported(M, Cmid, Cexp, Cred=None, dtype=torch.bfloat16):
    """Does sr_conv_pair (expansion conv fused with the next block's reduce conv, Cred output channels: Cmid inside a layer, 2 Cmid
    across a layer boundary) serve this shape?"""
    Cred = Cmid if Cred is None else Cred
    return dtype == torch.bfloat16 and lib().sr_conv_pair_supported(int(M), int(Cmid), int(Cexp), int(Cred), L.SR_BF16) == 1


def conv_pair_pack(w_exp, w_red):
    """Packed weight stream of sr_conv_pair: w_exp [Cexp, Cmid] (the expansion conv), w_red [Cred, Cexp] (the next block's reduce conv)."""
    require_gpu(w_exp, w_red)
    Cexp, Cmid = w_exp.shape
    Cred = w_red.shape[0]
    if w_red.shape[1] != Cexp or w_exp.dtype != torch.bfloat16 or w_red.dtype != torch.bfloat16:
        raise L.SrError("conv_pair_pack: expected bf16 w_exp [Cexp, Cmid] and w_red [Cred, Cexp]")
    nbytes = lib().sr_conv_pair_pack_bytes(Cmid, Cexp, Cred)
    check(min(nbytes, 0), "sr_conv_pair_pack_bytes")
    out = torch.empty(nbytes // 2, device=w_exp.device, dtype=torch.bfloat16)
    check(lib().sr_conv_pair_pack(w_exp.data_ptr(), w_red.data_ptr(), out.data_ptr(), Cmid, Cexp, Cred, L.SR_BF16, stream()), "sr_conv_pair_pack")
    return out


def conv_pair(x, wpack, res, escale, eshift, in_affine=None, ybias=None, yrelu=True):
    """z = relu((f(x) @ w_exp.T) * escale + eshift + res), y = z @ w_red.T (raw) and y's BatchNorm partials in one launch (sr_conv_pair);
    f = relu(x*in_scale + in_shift) with in_affine, identity without.  x: NHWC [B,H,W,Cmid], res: [B,H,W,Cexp]; the reduce conv's width
    follows from the weight stream's size.  Returns (z, y [B,H,W,Cred], stats).
    Eval mode (`ybias` [Cred]: the reduce conv's folded BatchNorm bias): y = [relu](z @ w_red.T + ybias), stats is None."""
    require_gpu(x, wpack, res, escale, eshift)
    B, H, W_, Cmid = x.shape
    Cexp = res.shape[3]
    M = B * H * W_
    if tuple(res.shape) != (B, H, W_, Cexp) or x.dtype != torch.bfloat16 or res.dtype != torch.bfloat16 or wpack.dtype != torch.bfloat16:
        raise L.SrError("conv_pair: shape/dtype mismatch")
    Cred = wpack.numel() // Cexp - Cmid
    if wpack.numel() != (Cmid + Cred) * Cexp or Cred <= 0 or escale.numel() != Cexp or eshift.numel() != Cexp:
        raise L.SrError("conv_pair: weight stream / scale vectors do not match (Cmid, Cexp)")
    rows = lib().sr_conv_pair_stats_rows(M, Cmid, Cexp, Cred)
    check(min(rows, 0), "sr_conv_pair_stats_rows")
    z = torch.empty_like(res)
    y = torch.empty((B, H, W_, Cred), device=x.device, dtype=x.dtype)
    stats = None if ybias is not None else torch.empty((rows, 2, Cred), device=x.device, dtype=torch.float32)
    if ybias is not None and (in_affine is not None or ybias.numel() != Cred):
        raise L.SrError("conv_pair: eval mode takes no in_affine and a bias of Cred elements")
    a = L.PairArgs()
    a.x, a.wpack, a.res, a.z, a.y = x.data_ptr(), wpack.data_ptr(), res.data_ptr(), z.data_ptr(), y.data_ptr()
    a.escale, a.eshift = _f32(escale, "escale").data_ptr(), _f32(eshift, "eshift").data_ptr()
    if in_affine is not None:
        require_gpu(*in_affine)
        if in_affine[0].numel() != Cmid or in_affine[1].numel() != Cmid:
            raise L.SrError("conv_pair: in_affine must have Cmid elements")
        a.in_scale, a.in_shift = _f32(in_affine[0], "in_scale").data_ptr(), _f32(in_affine[1], "in_shift").data_ptr()
    a.stats, a.M, a.Cmid, a.Cexp, a.Cred = ptr(stats), M, Cmid, Cexp, Cred
    if ybias is not None:
        require_gpu(ybias)
        a.ybias, a.yrelu = _f32(ybias, "ybias").data_ptr(), int(bool(yrelu))
    flops = 2.0 * M * Cexp * (Cmid + Cred)
    nbytes = 2.0 * (M * Cmid + 2 * M * Cexp + M * Cred + (Cmid + Cred) * Cexp)     # x, res read; z, y written; both weight matrices once
    check(_timed("conv1x1_pair", flops, nbytes, lambda: lib().sr_conv_pair(C.byref(a), L.SR_BF16, stream())), "sr_conv_pair")
    return z, y, stats


def stem_bn_relu_maxpool(xp, w, scale, shift, hw):
    """Fused stem: conv7x7/2 -> scale/shift -> ReLU -> maxpool3x3/2 of the padded NHWC4 image `xp` (see sr_stem_bn_relu_maxpool)."""
    require_gpu(xp, w, scale, shift)
    B, (H, W_) = xp.shape[0], hw
    Ho, Wo = (H - 1) // 2 + 1, (W_ - 1) // 2 + 1
    y = torch.empty((B, (Ho - 1) // 2 + 1, (Wo - 1) // 2 + 1, 64), device=xp.device, dtype=xp.dtype)
    flops = 2.0 * B * Ho * Wo * 64 * 147
    check(_timed("conv7x7", flops, float(xp.numel() * 2 + y.numel() * 2),
                 lambda: lib().sr_stem_bn_relu_maxpool(xp.data_ptr(), w.data_ptr(), _f32(scale, "scale").data_ptr(), _f32(shift, "shift").data_ptr(),
                                                       y.data_ptr(), B, H, W_, dtype_code(xp.dtype), stream())), "sr_stem_bn_relu_maxpool")
    return y


def stem_prep(img, dtype):
    require_gpu(img)
    _f32(img, "img")
    B, c, H, W_ = img.shape
    if c != 3:
        raise L.SrError("stem_prep expects [B,3,H,W]")
    out = torch.empty((B, (H + 7) & ~1, (W_ + 7) & ~1, 4), device=img.device, dtype=dtype)
    check(lib().sr_stem_prep(img.data_ptr(), out.data_ptr(), B, H, W_, dtype_code(dtype), stream()), "sr_stem_prep")
    return out


_BN_SCRATCH = {}


_MEAN3 = (C.c_float * 3)(0.485, 0.456, 0.406)
_STD3 = (C.c_float * 3)(0.229, 0.224, 0.225)


def image_prep_u8(img_u8, dtype, out_hw=None, crop_yx=None, flip=None):
    """uint8 NHWC [B,H0,W0,3] -> padded NHWC4 stem input of an HxW crop, ImageNet-normalised (see sr_image_prep_u8)."""
    require_gpu(img_u8, crop_yx, flip)
    if img_u8.dtype != torch.uint8 or img_u8.dim() != 4 or img_u8.shape[3] != 3:
        raise L.SrError("image_prep_u8 expects uint8 [B,H,W,3]")
    B, H0, W0, _ = img_u8.shape
    H, W_ = out_hw if out_hw is not None else (H0, W0)
    if crop_yx is not None and (crop_yx.dtype != torch.int32 or tuple(crop_yx.shape) != (B, 2)):
        raise L.SrError("crop_yx must be int32 [B,2]")
    if flip is not None and (flip.dtype != torch.uint8 or tuple(flip.shape) != (B,)):
        raise L.SrError("flip must be uint8 [B]")
    out = torch.empty((B, (H + 7) & ~1, (W_ + 7) & ~1, 4), device=img_u8.device, dtype=dtype)
    check(lib().sr_image_prep_u8(img_u8.data_ptr(), out.data_ptr(), B, H0, W0, H, W_, ptr(crop_yx), ptr(flip), _MEAN3, _STD3,
                                 dtype_code(dtype), stream()), "sr_image_prep_u8")
    return out


def bn_finalize(stats, count, gamma, beta, running_mean, running_var, momentum, eps, twin=None):
    """`twin` = (running_mean2, running_var2, momentum2): a second BatchNorm's buffers updated from the same batch statistics."""
    require_gpu(stats, gamma, beta, running_mean, running_var)
    rm2, rv2, m2 = twin if twin is not None else (None, None, 0.0)
    Cc = stats.shape[2]
    scale = torch.empty(Cc, device=stats.device, dtype=torch.float32)
    shift = torch.empty_like(scale)
    key = (stats.device, torch.cuda.current_stream().cuda_stream)
    scratch = None if _capturing() else _BN_SCRATCH.get(key)
    if scratch is None or scratch.shape[2] < Cc:
        scratch = torch.empty((1024, 2, max(Cc, 2048)), device=stats.device, dtype=torch.float64)
        if not _capturing():
            _BN_SCRATCH[key] = scratch
    check(lib().sr_bn_finalize(stats.data_ptr(), stats.shape[0], Cc, int(count), gamma.data_ptr(), beta.data_ptr(),
                               ptr(running_mean), ptr(running_var), float(momentum), float(eps), scale.data_ptr(),
                               shift.data_ptr(), scratch.data_ptr(), 1024, ptr(rm2), ptr(rv2), float(m2), stream()), "sr_bn_finalize")
    return scale, shift


_GRAM_SCRATCH = {}
_TN_SCRATCH = {}


def gemm_tn(A, B, out, accumulate=True):
    """out[N1,N2] (+)= A^T @ B for A [M,N1], B [M,N2] (bf16, rows may be strided), out fp32 contiguous; N1, N2 multiples
    of 256.  dW = dY^T X without transposed copies."""
    require_gpu(out)
    M, N1 = A.shape
    N2 = B.shape[1]
    if B.shape[0] != M or tuple(out.shape) != (N1, N2) or out.dtype != torch.float32 or A.stride(1) != 1 or B.stride(1) != 1:
        raise L.SrError("gemm_tn: shape / layout mismatch")
    ns = lib().sr_gemm_tn_slices(M, N1, N2)
    if ns < 0:
        raise L.SrError("gemm_tn: N1, N2 must be multiples of 256")
    key = (out.device, torch.cuda.current_stream().cuda_stream)
    scratch = _TN_SCRATCH.get(key)
    if scratch is None or scratch.numel() < ns * N1 * N2:
        scratch = _TN_SCRATCH[key] = torch.empty(ns * N1 * N2, device=out.device, dtype=torch.float32)
    check(_timed("gemm_tn", 2.0 * M * N1 * N2, 0, lambda: lib().sr_gemm_tn(
        A.data_ptr(), A.stride(0), B.data_ptr(), B.stride(0), M, N1, N2, dtype_code(A.dtype), out.data_ptr(), int(accumulate),
        scratch.data_ptr(), scratch.numel(), stream())), "sr_gemm_tn")
    return out



def gram_plan(M, Cc):
    """(number of partials, floats per partial) `gram` writes for an [M, Cc] input."""
    import ctypes
    n, f = ctypes.c_int64(), ctypes.c_int64()
    check(lib().sr_gram_plan(int(M), int(Cc), ctypes.byref(n), ctypes.byref(f)), "sr_gram_plan")
    return n.value, f.value


def gram_valid_mask(Cc, device="cpu"):
    """[C, C] bool: the entries of a partial's Gram part that sr_gram / sr_bn_apply_gram WRITE.  For C <= 256 only the 16 x 16 blocks
    on or below the block diagonal are computed (G is symmetric: sr_bn_finalize_gram mirrors them); C = 512 writes all of it."""
    if Cc > 256:
        return torch.ones(Cc, Cc, dtype=torch.bool, device=device)
    blk = torch.arange(Cc, device=device) // 16
    return blk.view(-1, 1) >= blk.view(1, -1)


def gram(x2d):
    """Per-slice partial Gram matrices + column sums of x2d [M, C] (bf16, C in 64/128/256/512); see gram_valid_mask."""
    require_gpu(x2d)
    M, Cc = x2d.shape
    n, f = gram_plan(M, Cc)
    part = torch.empty((n, f), device=x2d.device, dtype=torch.float32)
    check(_timed("gram", 0.0, 2.0 * M * Cc, lambda: lib().sr_gram(x2d.data_ptr(), M, Cc, x2d.stride(0), dtype_code(x2d.dtype),
                                                                 part.data_ptr(), n, stream())), "sr_gram")
    return part


def bn_apply_gram(x2d, scale, shift):
    """x2d [M, C] <- relu(x2d*scale + shift) in place, and the Gram partials of the result (one pass)."""
    require_gpu(x2d, scale, shift)
    M, Cc = x2d.shape
    n, f = gram_plan(M, Cc)
    part = torch.empty((n, f), device=x2d.device, dtype=torch.float32)
    check(_timed("gram", 0.0, 4.0 * M * Cc, lambda: lib().sr_bn_apply_gram(x2d.data_ptr(), M, Cc, x2d.stride(0), dtype_code(x2d.dtype),
                                                                          scale.data_ptr(), shift.data_ptr(), part.data_ptr(), n,
                                                                          stream())), "sr_bn_apply_gram")
    return part


def bn_gram(x2d, scale, shift):
    """Gram partials of relu(x2d*scale + shift); x2d [M, C] itself is NOT modified (its consumer applies the affine on load)."""
    require_gpu(x2d, scale, shift)
    M, Cc = x2d.shape
    n, f = gram_plan(M, Cc)
    part = torch.empty((n, f), device=x2d.device, dtype=torch.float32)
    check(_timed("gram", 0.0, 2.0 * M * Cc, lambda: lib().sr_bn_gram(x2d.data_ptr(), M, Cc, x2d.stride(0), dtype_code(x2d.dtype),
                                                                    scale.data_ptr(), shift.data_ptr(), part.data_ptr(), n, stream())),
          "sr_bn_gram")
    return part


def bn_finalize_gram(part, w, count, gamma, beta, running_mean, running_var, momentum, eps, twin=None):
    """Train-mode BN scale/shift (+ EMA) of the 1x1 conv with packed weights w [N, C] whose input has the partial Grams `part`."""
    require_gpu(part, w, gamma, beta, running_mean, running_var)
    rm2, rv2, m2 = twin if twin is not None else (None, None, 0.0)
    N, Cc = w.shape
    E = Cc * Cc + Cc
    scale = torch.empty(N, device=part.device, dtype=torch.float32)
    shift = torch.empty_like(scale)
    key = (part.device, torch.cuda.current_stream().cuda_stream)
    scratch = None if _capturing() else _GRAM_SCRATCH.get(key)
    if scratch is None or scratch.numel() < 66 * E:
        scratch = torch.empty(66 * (512 * 512 + 512), device=part.device, dtype=torch.float64)
        if not _capturing():
            _GRAM_SCRATCH[key] = scratch
    check(lib().sr_bn_finalize_gram(part.data_ptr(), part.shape[0], Cc, w.data_ptr(), w.stride(0), N, dtype_code(w.dtype), int(count),
                                    gamma.data_ptr(), beta.data_ptr(), ptr(running_mean), ptr(running_var), float(momentum),
                                    float(eps), scale.data_ptr(), shift.data_ptr(), scratch.data_ptr(), scratch.numel(), ptr(rm2), ptr(rv2),
                                    float(m2), stream()),
          "sr_bn_finalize_gram")
    return scale, shift


def bn_apply(x, scale, shift, res=None, relu=True, out=None):
    require_gpu(x, scale, shift, res)
    out = torch.empty_like(x) if out is None else out
    Cc = x.shape[-1]
    nbytes = (2 + (res is not None)) * x.numel() * x.element_size()
    check(_timed("bn_apply", 0, nbytes, lambda: lib().sr_bn_apply(x.data_ptr(), scale.data_ptr(), shift.data_ptr(), ptr(res), out.data_ptr(),
                                                                  x.numel() // Cc, Cc, int(relu), dtype_code(x.dtype), stream())),
          "sr_bn_apply")
    return out


def quantize_fp8(x, act_scale, scale=None, shift=None, relu=False):
    """e4m3(f(x) * act_scale) with f = [relu](x*scale + shift) -- the BatchNorm-apply that feeds an fp8 convolution -- or identity.
    Returns a uint8 tensor of x's shape holding OCP e4m3 bytes."""
    require_gpu(x, scale, shift)
    Cc = x.shape[-1]
    out = torch.empty(x.shape, device=x.device, dtype=torch.uint8)
    check(_timed("bn_apply", 0, x.numel() * (x.element_size() + 1),
                 lambda: lib().sr_quantize_fp8(x.data_ptr(), ptr(_f32(scale, "scale")), ptr(_f32(shift, "shift")), out.data_ptr(), x.numel() // Cc,
                                               Cc, int(relu), float(act_scale), dtype_code(x.dtype), stream())), "sr_quantize_fp8")
    return out


def conv3x3_fp8(xq, wq, dq, Cout, stride=1, want_stats=False):
    """3x3 / pad 1 convolution of e4m3 activations xq [B,H,W,Cin] (uint8 bytes) with e4m3 weights wq [Cout, 9*Cin]; returns the
    dequantised result in bf16 (and BatchNorm partial statistics)."""
    require_gpu(xq, wq, dq)
    if xq.dtype != torch.uint8 or wq.dtype != torch.uint8 or dq.dtype != torch.float32:
        raise L.SrError("conv3x3_fp8: xq / wq must be uint8 (e4m3 bytes) and dq fp32")
    B, H, W_, Cin = xq.shape
    Ho, Wo = (H - 1) // stride + 1, (W_ - 1) // stride + 1
    if tuple(wq.shape) != (Cout, 9 * Cin) or dq.shape[0] != Cout:
        raise L.SrError("conv3x3_fp8: weight / scale shapes do not match")
    y = torch.empty((B, Ho, Wo, Cout), device=xq.device, dtype=torch.bfloat16)
    stats = None
    if want_stats:
        rows = lib().sr_conv3x3_fp8_stats_rows(B * Ho * Wo, Cout)
        if rows < 0:
            raise L.SrError("conv3x3_fp8: unsupported shape")
        stats = torch.empty((rows, 2, Cout), device=xq.device, dtype=torch.float32)
    flops = 2.0 * B * Ho * Wo * Cout * 9 * Cin
    check(_timed("conv3x3_fp8", flops, float(xq.numel() + wq.numel() + 2 * y.numel()),
                 lambda: lib().sr_conv3x3_fp8(xq.data_ptr(), wq.data_ptr(), dq.data_ptr(), y.data_ptr(), ptr(stats), B, H, W_, Cin, Cout, stride,
                                              stream())), "sr_conv3x3_fp8")
    return (y, stats) if want_stats else y


def maxpool3x3s2(x, scale=None, shift=None):
    require_gpu(x, scale, shift)
    B, H, W_, Cc = x.shape
    y = torch.empty((B, (H - 1) // 2 + 1, (W_ - 1) // 2 + 1, Cc), device=x.device, dtype=x.dtype)
    check(_timed("maxpool", 0, (x.numel() + y.numel()) * x.element_size(),
                 lambda: lib().sr_maxpool3x3s2(x.data_ptr(), y.data_ptr(), B, H, W_, Cc, ptr(scale), ptr(shift), dtype_code(x.dtype), stream())),
          "sr_maxpool3x3s2")
    return y


def avgpool(x):
    require_gpu(x)
    B, H, W_, Cc = x.shape
    y = torch.empty((B, Cc), device=x.device, dtype=x.dtype)
    check(lib().sr_avgpool(x.data_ptr(), y.data_ptr(), B, H * W_, Cc, dtype_code(x.dtype), stream()), "sr_avgpool")
    return y


def _check_offs(offs, B):
    if offs is not None and (offs.dtype != torch.int32 or tuple(offs.shape) != (B + 1,) or not offs.is_cuda or not offs.is_contiguous()):
        raise L.SrError("packed rows: offs must be a contiguous int32 [B+1] CUDA tensor")


def node_init_fwd(feat, role_emb, verb_emb, verbs, role_table, offs=None, rows=None):
    """`offs` int32 [B+1] + `rows` = offs[B] (known to the host): the packed form -- only real roles' rows plus ONE shared row for all
    padded slots (row `rows`, zero), see sr_node_init_fwd."""
    require_gpu(feat, role_emb, verb_emb, verbs, role_table)
    B, D = feat.shape
    R = role_table.shape[1]
    if verbs.dtype != torch.int64 or role_table.dtype != torch.int32:
        raise L.SrError("verbs must be int64 and role_table int32")
    _check_offs(offs, B)
    node = torch.empty((B * R if offs is None else rows + 1, D), device=feat.device, dtype=feat.dtype)
    if offs is not None:
        node[rows:].zero_()
    check(lib().sr_node_init_fwd(feat.data_ptr(), _f32(role_emb, "role_emb").data_ptr(), _f32(verb_emb, "verb_emb").data_ptr(),
                                 verbs.data_ptr(), role_table.data_ptr(), node.data_ptr(), B, R, D, dtype_code(feat.dtype),
                                 ptr(offs), stream()), "sr_node_init_fwd")
    return node


def _role_inverted_index(role_table, NR):
    """CSR inverted index of the [V,R] role table: for every role id the slots v*R+r that hold it (static per encoder).
    Cached ON the tensor object (and its version counter), not in a table keyed by address: a freed table's address is reused
    by the next allocation of that size, and a stale index silently mis-routes the role gradients."""
    hit = getattr(role_table, "_sr_inv_index", None)
    if hit is None or hit[0] != (role_table._version, NR, role_table.data_ptr()):
        flat = role_table.reshape(-1).long()
        slots = torch.argsort(flat, stable=True)
        ptrs = torch.zeros(NR + 2, dtype=torch.int64, device=role_table.device)
        ptrs[1:] = torch.cumsum(torch.bincount(flat, minlength=NR + 1), 0)
        hit = ((role_table._version, NR, role_table.data_ptr()), ptrs[: NR + 1].to(torch.int32).contiguous(), slots.to(torch.int32).contiguous())
        role_table._sr_inv_index = hit
    return hit[1], hit[2]


def node_init_bwd(dnode, feat, role_emb, verb_emb, verbs, role_table, d_role_emb, d_verb_emb, offs=None):
    """Writes d_role_emb [NR+1,D] and d_verb_emb [V,D] in full (deterministic: see sr_node_init_bwd)."""
    require_gpu(dnode, feat, role_emb, verb_emb, verbs, role_table, d_role_emb, d_verb_emb)
    B, D = feat.shape
    V, R = role_table.shape
    NR = role_emb.shape[0] - 1
    _check_offs(offs, B)
    if tuple(d_role_emb.shape) != (NR + 1, D) or tuple(d_verb_emb.shape) != (V, D) or verb_emb.shape[0] != V:
        raise L.SrError("node_init_bwd: gradient / table shapes do not match")
    sorted_verbs, order = torch.sort(verbs, stable=True)            # (no host synchronisation: bincount would need one)
    seg = torch.searchsorted(sorted_verbs, torch.arange(V + 1, device=verbs.device, dtype=verbs.dtype)).to(torch.int32)
    order = order.to(torch.int32)
    inv_ptr, inv_slot = _role_inverted_index(role_table, NR)
    scratch = torch.empty(((V + 2 * ((B + 31) // 32)) * R, D), device=feat.device, dtype=torch.float32)   # srhip.h: S | PH | PT
    check(lib().sr_node_init_bwd(dnode.data_ptr(), feat.data_ptr(), role_emb.data_ptr(), verb_emb.data_ptr(), order.data_ptr(),
                                 seg.data_ptr(), role_table.data_ptr(), inv_ptr.data_ptr(), inv_slot.data_ptr(),
                                 scratch.data_ptr(), _f32(d_role_emb, "d_role_emb").data_ptr(),
                                 _f32(d_verb_emb, "d_verb_emb").data_ptr(), B, R, D, V, NR, dtype_code(feat.dtype), ptr(offs), stream()),
          "sr_node_init_bwd")


def aggregate(h, adj_table, verbs, R, transpose=False, add=None, out=None, offs=None):
    """`offs` int32 [B+1]: packed role rows (h has offs[B] + 1 rows: the real roles' rows and the shared padded-slot row)."""
    require_gpu(h, adj_table, verbs, add)
    M, D = h.shape
    B = M // R if offs is None else verbs.shape[0]
    _check_offs(offs, B)
    out = torch.empty_like(h) if out is None else out
    nbytes = (2 + (add is not None)) * M * D * h.element_size() + 4 * B * R * R
    check(_timed("aggregate", 0, nbytes,
                 lambda: lib().sr_ggnn_aggregate(h.data_ptr(), _f32(adj_table, "adj_table").data_ptr(), verbs.data_ptr(), ptr(add),
                                                 out.data_ptr(), B, R, D, int(transpose), dtype_code(h.dtype), ptr(offs), stream())),
          "sr_ggnn_aggregate")
    return out


def gru_bwd1(dh, z, c, h, dc=None, dz=None):
    """`dc`, `dz`: optional contiguous destinations (slices of the stacked [T, M, D] gradients the weight-gradient GEMMs read)."""
    require_gpu(dh, z, c, h, dc, dz)
    dc = torch.empty_like(dh) if dc is None else dc
    dz = torch.empty_like(dh) if dz is None else dz
    dacc = torch.empty_like(dh)
    for t in (dc, dz):
        if t.shape != dh.shape or t.dtype != dh.dtype or not t.is_contiguous():
            raise L.SrError("gru_bwd1: bad destination")
    check(_timed("gru_bwd", 0, 7 * dh.numel() * dh.element_size(),       # reads dh, z, c, h; writes dc, dz, dacc
                 lambda: lib().sr_gru_bwd1(dh.data_ptr(), z.data_ptr(), c.data_ptr(), h.data_ptr(), dc.data_ptr(), dz.data_ptr(),
                                           dacc.data_ptr(), dh.numel(), dtype_code(dh.dtype), stream())), "sr_gru_bwd1")
    return dc, dz, dacc


def gru_bwd2(drh, r, h, dh_acc, dr=None):
    require_gpu(drh, r, h, dh_acc, dr)
    dr = torch.empty_like(drh) if dr is None else dr
    if dr.shape != drh.shape or dr.dtype != drh.dtype or not dr.is_contiguous():
        raise L.SrError("gru_bwd2: bad destination")
    check(_timed("gru_bwd", 0, 6 * drh.numel() * drh.element_size(),     # reads drh, r, h, dacc; writes dr, dacc
                 lambda: lib().sr_gru_bwd2(drh.data_ptr(), r.data_ptr(), h.data_ptr(), dr.data_ptr(), dh_acc.data_ptr(), drh.numel(),
                                           dtype_code(drh.dtype), stream())), "sr_gru_bwd2")
    return dr


def transpose(x, out_dtype=None, colsum=None, colsum_scale=1.0, pad_to=1, out=None):
    """[R,C] (rows may be strided) -> [C, Rpad] with Rpad = R rounded up to `pad_to`, zero filled.
    `out`: optional preallocated contiguous [C, Rpad] destination (e.g. a slice of a stacked operand)."""
    require_gpu(colsum)
    R, Cc = x.shape
    if x.stride(1) != 1 or not x.is_cuda:
        raise L.SrError("transpose: rows must be contiguous CUDA memory")
    Rp = (R + pad_to - 1) // pad_to * pad_to
    if out is None:
        out = torch.empty((Cc, Rp), device=x.device, dtype=out_dtype or x.dtype)
    elif tuple(out.shape) != (Cc, Rp) or not out.is_contiguous():
        raise L.SrError("transpose: bad `out`")
    check(lib().sr_transpose(x.data_ptr(), x.stride(0), out.data_ptr(), R, Cc, Rp, dtype_code(x.dtype), dtype_code(out.dtype),
                             ptr(_f32(colsum, "colsum")), float(colsum_scale), stream()), "sr_transpose")
    return out


def colsum(x, acc, scale=1.0):
    require_gpu(acc)
    R, Cc = x.shape
    if x.stride(1) != 1 or not x.is_cuda:
        raise L.SrError("colsum: rows must be contiguous CUDA memory")
    check(lib().sr_colsum(x.data_ptr(), x.stride(0), R, Cc, dtype_code(x.dtype), _f32(acc, "acc").data_ptr(), float(scale),
                          stream()), "sr_colsum")
    return acc


def cast_pad(x, dtype, pad_to=1):
    """[R,C] (strided rows allowed) -> contiguous [R, Cpad] in `dtype`, zero filled."""
    R, Cc = x.shape
    if x.stride(1) != 1 or not x.is_cuda:
        raise L.SrError("cast_pad: rows must be contiguous CUDA memory")
    Cp = (Cc + pad_to - 1) // pad_to * pad_to
    out = torch.empty((R, Cp), device=x.device, dtype=dtype)
    check(lib().sr_cast_pad(x.data_ptr(), x.stride(0), out.data_ptr(), Cp, R, Cc, Cp, dtype_code(x.dtype), dtype_code(dtype),
                            stream()), "sr_cast_pad")
    return out


def cast(x, dtype):
    require_gpu(x)
    out = torch.empty(x.shape, device=x.device, dtype=dtype)
    check(lib().sr_cast(x.data_ptr(), out.data_ptr(), x.numel(), dtype_code(x.dtype), dtype_code(dtype), stream()), "sr_cast")
    return out


def dropout_half(x, seed, want_mask=False):
    require_gpu(x)
    y = torch.empty_like(x)
    mask = torch.empty(x.shape, device=x.device, dtype=torch.uint8) if want_mask else None
    check(lib().sr_dropout_half(x.data_ptr(), y.data_ptr(), ptr(mask), x.numel(), int(seed) & (2 ** 64 - 1),
                                dtype_code(x.dtype), stream()), "sr_dropout_half")
    return (y, mask) if want_mask else y
