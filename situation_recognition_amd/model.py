"""The reference's model surface (reference model.py: `resnet`, `GGSNN`, `FCGGNN`) executed by the
HIP kernels of libsrhip.so.  Same class / method / parameter names and state-dict keys as the
reference, so `sr.py`-style drivers and reference checkpoints work unchanged; the arithmetic runs
only on an MI355X (no CPU fallback: CPU tensors raise).

What PyTorch does here: owns parameters and device memory, chains the hand-written backward
functions through autograd, computes the cross-entropy losses and runs the optimizer
(north_star: "PyTorch-ROCm for autograd/optimizer plumbing").  Everything between the input image
and the logits -- and its backward -- is a libsrhip kernel.

Knobs the reference hard-codes are constructor arguments: GGNN steps (model.py:60 fixes 4), backbone
depth (model.py:16 fixes ResNet-152), storage dtype (the reference uses fp16 autocast on CUDA, fp32 on CPU).
"""
import itertools
from math import sqrt

import os

import torch
import torch.nn as nn

from . import ops
from ._lib import SrError

_ARCH = {18: ("basic", (2, 2, 2, 2)), 34: ("basic", (3, 4, 6, 3)), 50: ("bottleneck", (3, 4, 6, 3)),
         101: ("bottleneck", (3, 4, 23, 3)), 152: ("bottleneck", (3, 8, 36, 3))}


def _pad_channels(c):
    """Channel counts the implicit-GEMM kernel accepts: a power of two >= 64.  Every real ResNet
    width already is one; narrower test nets are zero-padded (padded channels stay exactly 0)."""
    p = 64
    while p < c:
        p *= 2
    return p


# ----------------------------------------------------------------------------- backbone
class _Block(nn.Module):
    """Parameter container for one residual block (torchvision names); never called."""

    def __init__(self, kind, cin, planes, stride):
        super().__init__()
        self.kind, self.stride = kind, stride
        if kind == "bottleneck":
            cout = planes * 4
            self.conv1 = nn.Conv2d(cin, planes, 1, bias=False)
            self.bn1 = nn.BatchNorm2d(planes)
            self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
            self.bn2 = nn.BatchNorm2d(planes)
            self.conv3 = nn.Conv2d(planes, cout, 1, bias=False)
            self.bn3 = nn.BatchNorm2d(cout)
        else:
            cout = planes
            self.conv1 = nn.Conv2d(cin, planes, 3, stride=stride, padding=1, bias=False)
            self.bn1 = nn.BatchNorm2d(planes)
            self.conv2 = nn.Conv2d(planes, planes, 3, padding=1, bias=False)
            self.bn2 = nn.BatchNorm2d(planes)
        self.cout = cout
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride=stride, bias=False), nn.BatchNorm2d(cout))


class _ResNetParams(nn.Module):
    """Parameters/buffers of a torchvision-style ResNet (v1.5: stride on the 3x3), with torchvision's
    state-dict key names.  The reference obtains this object from `tv.models.resnet152(pretrained=True)`
    (model.py:16); ImageNet weights are a network fetch, so here the net starts from torchvision's own
    initialisation (He-normal fan_out convs, BN gamma=1 beta=0) and real weights arrive through
    `load_state_dict` of a reference checkpoint."""

    def __init__(self, depth, width, blocks):
        super().__init__()
        kind, default_blocks = _ARCH[depth]
        blocks = tuple(blocks) if blocks is not None else default_blocks
        self.conv1 = nn.Conv2d(3, width, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(width)
        cin = width
        for s in range(4):
            stage = []
            for i in range(blocks[s]):
                blk = _Block(kind, cin, width << s, 2 if (i == 0 and s > 0) else 1)
                stage.append(blk)
                cin = blk.cout
            setattr(self, "layer%d" % (s + 1), nn.Sequential(*stage))
        self.fc = nn.Linear(cin, 1000)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def forward(self, x):
        raise SrError("_ResNetParams is a parameter container; call the owning `resnet` module")


class _ConvBN:
    """One conv + BatchNorm unit: packs the weights for the implicit-GEMM kernel
    ([Cout][KH][KW][Cin], channel-padded, storage dtype) and, for eval mode, folds the BN affine of the
    running statistics into them.  Packs are cached and rebuilt when a parameter changes."""

    def __init__(self, conv, bn, stem=False):
        self.conv, self.bn, self.stem = conv, bn, stem
        self.k, self.stride, self.pad = conv.kernel_size[0], conv.stride[0], conv.padding[0]
        self.cin, self.cout = conv.in_channels, conv.out_channels
        self.cin_p = 3 if stem else _pad_channels(self.cin)
        self.cout_p = _pad_channels(self.cout)
        self._cache = {}

    def _sig(self, with_stats):
        t = [self.conv.weight, self.bn.weight, self.bn.bias] + ([self.bn.running_mean, self.bn.running_var] if with_stats else [])
        return tuple((x.data_ptr(), x._version) for x in t)

    def _pack(self, w, dtype):
        co, ci = w.shape[0], w.shape[1]
        if self.stem:                                   # [co][r][q*4+c], 8 rows x 8 pixels x 4 channels
            out = w.new_zeros(self.cout_p, 8, 8, 4)
            out[:co, :7, :7, :3] = w.permute(0, 2, 3, 1)
            return out.reshape(self.cout_p, 256).to(dtype).contiguous()
        out = w.new_zeros(self.cout_p, self.k, self.k, self.cin_p)
        out[:co, :, :, :ci] = w.permute(0, 2, 3, 1)
        return out.reshape(self.cout_p, -1).to(dtype).contiguous()

    def _padded(self, t, fill):
        if t.shape[0] == self.cout_p:
            return t
        out = t.new_full((self.cout_p,), fill)
        out[: t.shape[0]] = t
        return out

    def raw(self, dtype):
        key = ("raw", dtype)
        sig = self._sig(False)
        hit = self._cache.get(key)
        if hit is None or hit[0] != sig:
            with torch.no_grad():
                hit = (sig, self._pack(self.conv.weight.detach().float(), dtype),
                       self._padded(self.bn.weight.detach().float(), 1.0).contiguous(),
                       self._padded(self.bn.bias.detach().float(), 0.0).contiguous())
            self._cache[key] = hit
        return hit[1], hit[2], hit[3]

    def folded(self, dtype, stats_epoch):
        key = ("fold", dtype)
        sig = (self._sig(True), stats_epoch)
        hit = self._cache.get(key)
        if hit is None or hit[0] != sig:
            bn = self.bn
            with torch.no_grad():
                scale = bn.weight.float() * torch.rsqrt(bn.running_var.float() + bn.eps)
                shift = bn.bias.float() - bn.running_mean.float() * scale
                w = self.conv.weight.detach().float() * scale.view(-1, 1, 1, 1)
                hit = (sig, self._pack(w, dtype), self._padded(shift, 0.0).contiguous())
            self._cache[key] = hit
        return hit[1], hit[2]

    def pair_pack(self, nxt, dtype):
        """Weight stream of the fused launch `sr_conv_pair`: this unit (a bottleneck's expansion conv) followed by `nxt` (the next block's
        reduce conv), both in MFMA-fragment order (ops.conv_pair_pack); rebuilt when either weight changes."""
        key = ("pair", dtype)
        sig = (self._sig(False), nxt._sig(False))
        hit = self._cache.get(key)
        if hit is None or hit[0] != sig:
            with torch.no_grad():
                hit = (sig, ops.conv_pair_pack(self.raw(dtype)[0].view(self.cout_p, self.cin_p), nxt.raw(dtype)[0].view(nxt.cout_p, nxt.cin_p)))
            self._cache[key] = hit
        return hit[1]

    def pair_pack_folded(self, nxt, dtype, stats_epoch):
        """The same stream for eval mode: both convolutions with their BatchNorm folded in (`folded`); returns (stream, bias of this
        unit, bias of `nxt`).  Rebuilt when a weight or the running statistics change."""
        key = ("pair_fold", dtype)
        sig = (self._sig(True), nxt._sig(True), stats_epoch)
        hit = self._cache.get(key)
        if hit is None or hit[0] != sig:
            with torch.no_grad():
                w3, b3 = self.folded(dtype, stats_epoch)
                w1, b1 = nxt.folded(dtype, stats_epoch)
                hit = (sig, ops.conv_pair_pack(w3.view(self.cout_p, self.cin_p), w1.view(nxt.cout_p, nxt.cin_p)), b3, b1,
                       torch.ones_like(b3))
            self._cache[key] = hit
        return hit[1], hit[2], hit[3], hit[4]

    def fp8_eligible(self):
        """3x3 convolutions the e4m3 kernel serves (csrc/fp8.hip): 128 / 256 / 512 input channels, output channels a multiple of 128."""
        return self.k == 3 and self.pad == 1 and self.cin_p in (128, 256, 512) and self.cout_p % 128 == 0 and not self.stem

    def fp8_pack(self, act_scale):
        """(e4m3 weights [Cout, 9*Cin] as uint8, dq [Cout] fp32): w * s_w[co] rounded to e4m3 with s_w = 448 / max|w[co]|, and the
        factor 1 / (act_scale * s_w[co]) that brings the fp32 accumulator back to the convolution's scale."""
        key = ("fp8", float(act_scale))
        sig = self._sig(False)
        hit = self._cache.get(key)
        if hit is None or hit[0] != sig:
            with torch.no_grad():
                w = self.conv.weight.detach().float()
                s_w = 448.0 / w.abs().amax(dim=(1, 2, 3)).clamp_min(1e-30)
                wq = (w * s_w.view(-1, 1, 1, 1)).clamp_(-448.0, 448.0).to(torch.float8_e4m3fn).view(torch.uint8)
                out = wq.new_zeros(self.cout_p, self.k, self.k, self.cin_p)
                out[: w.shape[0], :, :, : w.shape[1]] = wq.permute(0, 2, 3, 1)
                dq = self._padded(1.0 / (float(act_scale) * s_w), 0.0).contiguous()
                hit = (sig, out.reshape(self.cout_p, -1).contiguous(), dq)
            self._cache[key] = hit
        return hit[1], hit[2]

    def eval_affine(self, stats_epoch):
        """(scale, shift) of the eval-mode BatchNorm (running statistics), for paths that cannot fold it into the weights."""
        key = ("affine",)
        sig = (self._sig(True), stats_epoch)
        hit = self._cache.get(key)
        if hit is None or hit[0] != sig:
            bn = self.bn
            with torch.no_grad():
                scale = bn.weight.float() * torch.rsqrt(bn.running_var.float() + bn.eps)
                shift = bn.bias.float() - bn.running_mean.float() * scale
                hit = (sig, self._padded(scale, 1.0).contiguous(), self._padded(shift, 0.0).contiguous())
            self._cache[key] = hit
        return hit[1], hit[2]

    def running(self):
        """running_mean / running_var tensors the finalize kernel updates in place (padded copies when
        the unit is channel-padded; `writeback` copies them home)."""
        bn = self.bn
        if self.cout_p == self.cout:
            return bn.running_mean, bn.running_var, False
        return self._padded(bn.running_mean, 0.0).contiguous(), self._padded(bn.running_var, 1.0).contiguous(), True

    def writeback(self, rm, rv):
        with torch.no_grad():
            self.bn.running_mean.copy_(rm[: self.cout])
            self.bn.running_var.copy_(rv[: self.cout])


class resnet(nn.Module):
    """reference model.py:8-35 -- frozen ResNet feature extractor ([B,3,H,W] -> [B,2048] pooled features).

    HIP execution: NCHW fp32 image -> padded NHWC4 (`sr_stem_prep`) -> 7x7/2 stem and every
    1x1 / 3x3 conv as implicit GEMM on MFMA (`sr_conv2d`), NHWC activations in `dtype`.
      * train mode (the reference trains with `model.train()`, sr.py:16, so the frozen BatchNorms use
        batch statistics and update their running statistics): conv writes the raw output and per-tile
        channel sums from the fp32 accumulators, `sr_bn_finalize` reduces them to scale/shift (+ EMA),
        `sr_bn_apply` normalises (+residual, ReLU) in place;
      * eval mode: BN is folded into the packed weights; bias, residual add and ReLU run in the conv epilogue.
    """

    def __init__(self, out_layers, depth=152, width=64, blocks=None, dtype=torch.bfloat16, fp8=False, fp8_act_scale=16.0):
        super().__init__()
        if fp8 and dtype != torch.bfloat16:
            raise SrError("the fp8 path of the 3x3 convolutions sits inside a bf16 backbone (dtype=torch.bfloat16)")
        # fp8: the bottlenecks' 3x3 convolutions (the matrix-bound family) take e4m3 activations and weights on the 2x-rate
        # scaled MFMA (csrc/fp8.hip; BASELINE config 5).  Activations are quantised as e4m3(a * fp8_act_scale) by the BatchNorm
        # apply in front of the conv (post-BN-ReLU values are O(1): 16 keeps them in e4m3's normal range up to 28).
        self.fp8, self.fp8_act_scale = bool(fp8), float(fp8_act_scale)
        self.model = _ResNetParams(depth, width, blocks)
        for p in self.model.parameters():               # model.py:17-18
            p.requires_grad = False
        self.out_features = self.model.fc.in_features
        # model.py:21-31 builds a fresh fc and then discards it for Identity; only the Identity survives
        self.model.fc = nn.Identity()
        self.dtype = dtype
        self.depth = depth
        self._units = None
        self.two_pass = True           # train-mode BN of output-heavy 1x1 convs in two conv launches (see _unit)
        self.gram_stats = True         # ... with launch 1 replaced by the input's Gram matrix for the expansion convs (bf16)
        self.fuse_stem_pool = True     # stem + BN + ReLU + maxpool as one kernel (bf16, 64-channel stem)
        self.lazy_bn2 = os.environ.get("SR_NO_LAZY_BN2") != "1"   # bn2 + ReLU applied by conv3 on load (train mode, bf16)
        self.fuse_pairs = os.environ.get("SR_NO_PAIR") != "1"     # expansion conv + the next block's reduce conv in one launch (train, bf16)
        self.use_graphs = False        # eval-mode passes replayed from a captured hipGraph (opt-in: FCGGNN.enable_graphs())
        self.graph_train = False       # ... train-mode passes too (small per-GPU batches: ~900 launches of 10-200 us, 3 us apart)
        self._capturing = False
        self._graphs = {}
        self.graph_replays = 0         # passes served by a graph launch (tests assert that the replay branch really ran)
        self._stats_epoch = 0          # bumped whenever a train-mode pass changed running statistics
        self._pending_tracked = 0      # num_batches_tracked increments not yet written to the buffers
        self._gram_stash = None        # (data_ptr, Gram partials) a fused BN-apply left for the expansion conv that follows
        self._lazy_in = None           # (data_ptr, (scale, shift)): a RAW tensor whose BatchNorm + ReLU its consumer applies on load
        self.register_state_dict_pre_hook(lambda m, prefix, keep_vars: m._flush_counters())
        self.register_load_state_dict_post_hook(lambda m, incompatible: m._after_load())

    # -- bookkeeping
    def _flush_counters(self):
        if self._pending_tracked:
            with torch.no_grad():
                for m in self.model.modules():
                    if isinstance(m, nn.BatchNorm2d):
                        m.num_batches_tracked += self._pending_tracked
            self._pending_tracked = 0

    def _after_load(self):
        """load_state_dict replaced weights and counters: captured graphs hold folded packs of the OLD weights, and
        increments recorded before the load must not be added on top of the loaded num_batches_tracked."""
        self._graphs.clear()
        self._pending_tracked = 0
        self._stats_epoch += 1

    def _apply(self, fn, *a, **kw):                     # .to() / .cuda() / .float(): parameters move, captured graphs do not
        self._graphs.clear()
        return super()._apply(fn, *a, **kw)

    def _weights_signature(self, buffers=True):
        """Changes whenever any backbone parameter (or, with `buffers`, BatchNorm buffer) is modified in place or replaced (sum
        of tensor versions + storage addresses): the key of the captured graphs.  Eval-mode graphs bake in weight packs folded
        with the running statistics, so the buffers are part of their key; train-mode graphs read only the parameters and UPDATE
        the buffers themselves (a channel-padded net even through `copy_`, which bumps their versions every pass), so theirs is
        keyed on the parameters alone."""
        v = 0
        for t in itertools.chain(self.model.parameters(), self.model.buffers() if buffers else ()):
            v += t._version + (t.data_ptr() & 0xffffffff)
        return v

    def _plan(self):
        if self._units is None:
            m = self.model
            stem = _ConvBN(m.conv1, m.bn1, stem=True)
            blocks = []
            for s in range(4):
                for blk in getattr(m, "layer%d" % (s + 1)):
                    convs = [_ConvBN(blk.conv1, blk.bn1), _ConvBN(blk.conv2, blk.bn2)]
                    if blk.kind == "bottleneck":
                        convs.append(_ConvBN(blk.conv3, blk.bn3))
                    ds = _ConvBN(blk.downsample[0], blk.downsample[1]) if blk.downsample is not None else None
                    blocks.append((convs, ds))
            self._units = (stem, blocks)
        return self._units

    # -- execution
    def _gram_route(self, u, n_pixels):
        """Does expansion conv `u` take its batch statistics from the Gram matrix of its input?"""
        return (self.two_pass and self.gram_stats and self.dtype == torch.bfloat16 and u.k == 1 and u.stride == 1 and u.cout_p > 128
                and u.cin_p in (64, 128, 256, 512) and u.cout_p >= 4 * u.cin_p
                and n_pixels >= 256 * u.cin_p)    # (below ~256*C pixels the fixed cost of the fp64 finalize loses to launch 1)

    def _fused_stem(self, u, stem_hw, pool_after):
        return (self.fuse_stem_pool and stem_hw is not None and pool_after and self.dtype == torch.bfloat16 and u.cout_p == 64
                and os.environ.get("SR_NO_STEM_DIRECT") != "1")

    def _lazy_ok(self, y, then):
        """Can the expansion unit `then` consume the RAW tensor y and normalise it on load?  (It is always called with the block's
        identity as residual and ReLU -- the form the kernel serves; geometry is asked of the library.)"""
        return (then.k == 1 and then.stride == 1 and y.dtype == torch.bfloat16
                and ops.conv_in_affine_supported(y, then.cout_p, 1, 1, 0, res=y, relu=True))

    def _pair_ok(self, u, nxt, n_pixels):
        """Is expansion unit `u` followed by reduce unit `nxt` (the next block's conv1, fed this block's output and nothing else) served
        by the fused launch `sr_conv_pair`?"""
        return (self.fuse_pairs and nxt is not None and self.dtype == torch.bfloat16 and u.k == 1 and u.stride == 1 and nxt.k == 1
                and nxt.stride == 1 and nxt.pad == 0 and nxt.cin_p == u.cout_p and n_pixels >= 128 * 256
                and ops.conv_pair_supported(n_pixels, u.cin_p, u.cout_p, nxt.cout_p, self.dtype))

    def _unit(self, x, u, train, momentum, relu, res=None, stem_hw=None, pool_after=False, then=None, twin=None, quant_out=False,
              pre=None, fuse_next=None):
        """`then`: the unit that consumes this one's output next (lets BN-apply and the consumer's Gram pass share one sweep).
        `twin` = (unit of a weight-identical backbone, its momentum): its running statistics are updated from the same batch.
        `fuse_next`: the NEXT block's conv1 unit when this (expansion) unit may run fused with it (`sr_conv_pair`): the return value is
        then (block output, (raw output of that conv1, its statistics partials)).  `pre` = such a pair, handed to that conv1 unit: its
        convolution has already run."""
        dt = self.dtype
        f8_in = x.dtype == torch.uint8 and stem_hw is None         # e4m3 activations from the preceding unit (quant_out)
        if not train:
            if f8_in:                                              # raw fp8 convolution, then the eval-mode affine + ReLU
                wq, dq = u.fp8_pack(self.fp8_act_scale)
                y = ops.conv3x3_fp8(x, wq, dq, u.cout_p, stride=u.stride)
                sc, sh = u.eval_affine(self._stats_epoch)
                return ops.bn_apply(y, sc, sh, res=res, relu=relu, out=y)
            if pre is not None:                                    # this convolution (+ folded BN + ReLU) ran inside the previous block's fused launch
                return ops.quantize_fp8(pre[0], self.fp8_act_scale) if quant_out else pre[0]
            if fuse_next is not None:
                if res is not None and relu and x.dtype == dt and self._pair_ok(u, fuse_next, x.shape[0] * x.shape[1] * x.shape[2]):
                    # expansion conv + the next block's reduce conv (both with their BatchNorm folded in) in one pass over the block output
                    wp, b3, b1, one = u.pair_pack_folded(fuse_next, dt, self._stats_epoch)
                    z, y1, _ = ops.conv_pair(x, wp, res, one, b3, ybias=b1, yrelu=True)
                    return z, (y1, None)
                w, b = u.folded(dt, self._stats_epoch)
                return ops.conv2d(x, w, u.cout_p, u.k, u.stride, u.pad, bias=b, res=res, relu=relu), None
            w, b = u.folded(dt, self._stats_epoch)
            if self._fused_stem(u, stem_hw, pool_after):           # conv1 + folded BN + ReLU + maxpool in one launch
                return ops.stem_bn_relu_maxpool(x, w, torch.ones_like(b), b, stem_hw)
            y = ops.conv2d(x, w, u.cout_p, u.k, u.stride, u.pad, bias=b, res=res, relu=relu, stem_hw=stem_hw)
            if quant_out:
                return ops.quantize_fp8(y, self.fp8_act_scale)
            return ops.maxpool3x3s2(y) if pool_after else y
        w, gamma, beta = u.raw(dt)
        rm, rv, padded = u.running()
        tw = None
        if twin is not None:
            trm, trv, _ = twin[0].running()
            tw = (trm, trv, twin[1])

        def done():                                     # channel-padded test nets: copy the updated statistics home
            if padded:
                u.writeback(rm, rv)
                if twin is not None:
                    twin[0].writeback(tw[0], tw[1])
        if self._fused_stem(u, stem_hw, pool_after):
            # the stem in two launches: statistics only (nothing written), then conv + BatchNorm + ReLU + max-pool in one kernel --
            # the raw conv1 output (9.9 GB at batch 6144) is never written nor read back
            st = ops.conv2d(x, w, u.cout_p, u.k, u.stride, u.pad, stats_only=True, stem_hw=stem_hw)
            Ho, Wo = (stem_hw[0] - 1) // 2 + 1, (stem_hw[1] - 1) // 2 + 1
            scale, shift = ops.bn_finalize(st, x.shape[0] * Ho * Wo, gamma, beta, rm, rv, momentum, u.bn.eps, twin=tw)
            done()
            return ops.stem_bn_relu_maxpool(x, w, scale, shift, stem_hw)
        if self.two_pass and u.k == 1 and u.cout_p >= 2 * u.cin_p and u.cout_p > 128 and not pool_after:
            # Output-heavy 1x1 conv (bottleneck expansion / downsample): launch it twice instead of conv -> raw tensor ->
            # elementwise pass.  Launch 1 only produces the batch statistics (nothing is written); launch 2 recomputes the
            # cheap GEMM and applies scale/shift (+identity, ReLU) in its epilogue.  HBM traffic per output element drops
            # from 5 accesses (write raw, read raw, read identity, write) to 2 (read identity, write).
            Ho, Wo = (x.shape[1] - 1) // u.stride + 1, (x.shape[2] - 1) // u.stride + 1
            in_affine = None
            if self._gram_route(u, x.shape[0] * Ho * Wo):
                # Expansion conv (N = 4C): its batch statistics follow from the C x C Gram matrix of the input
                # (sum y^2 = w G w^T), a quarter of the conv's MFMA work and one read of x -- no launch 1 at all.
                stash, self._gram_stash = self._gram_stash, None
                hit = stash is not None and stash[0] == x.data_ptr()
                part = stash[1] if hit else ops.gram(x.view(-1, u.cin_p))
                if hit and stash[2] is not None:          # x is still the RAW output of the preceding conv: this conv normalises it on load
                    in_affine = stash[2]
                scale, shift = ops.bn_finalize_gram(part, w.view(u.cout_p, u.cin_p), x.shape[0] * Ho * Wo, gamma, beta, rm, rv,
                                                    momentum, u.bn.eps, twin=tw)
            else:
                st = ops.conv2d(x, w, u.cout_p, u.k, u.stride, u.pad, stats_only=True)
                scale, shift = ops.bn_finalize(st, x.shape[0] * Ho * Wo, gamma, beta, rm, rv, momentum, u.bn.eps, twin=tw)
            done()
            if fuse_next is not None:
                if res is not None and relu and self._pair_ok(u, fuse_next, x.shape[0] * Ho * Wo):
                    # expansion conv + the next block's reduce conv in ONE pass over the block output: it is written once and not read
                    # back by the reduce conv (reference chain conv3 -> bn3 -> add -> relu -> next.conv1, model.py:35)
                    z, y1, st1 = ops.conv_pair(x, u.pair_pack(fuse_next, dt), res, scale, shift, in_affine=in_affine)
                    return z, (y1, st1)
                return ops.conv2d(x, w, u.cout_p, u.k, u.stride, u.pad, bias=shift, escale=scale, res=res, relu=relu, in_affine=in_affine), None
            return ops.conv2d(x, w, u.cout_p, u.k, u.stride, u.pad, bias=shift, escale=scale, res=res, relu=relu, in_affine=in_affine)
        lazy_in, self._lazy_in = self._lazy_in, None
        if pre is not None:                             # this convolution ran inside the previous block's fused launch
            y, st = pre
        elif f8_in:
            wq, dq = u.fp8_pack(self.fp8_act_scale)
            y, st = ops.conv3x3_fp8(x, wq, dq, u.cout_p, stride=u.stride, want_stats=True)
        else:
            # (x may still be the RAW output of the unit in front: this conv then applies that unit's BatchNorm + ReLU on load)
            aff = lazy_in[1] if lazy_in is not None and lazy_in[0] == x.data_ptr() else None
            y, st = ops.conv2d(x, w, u.cout_p, u.k, u.stride, u.pad, want_stats=True, stem_hw=stem_hw, in_affine=aff)
        scale, shift = ops.bn_finalize(st, y.numel() // u.cout_p, gamma, beta, rm, rv, momentum, u.bn.eps, twin=tw)
        done()
        if quant_out:                                   # BatchNorm + ReLU + e4m3 in one sweep: the fp8 conv's input, half the bytes of bf16
            return ops.quantize_fp8(y, self.fp8_act_scale, scale, shift, relu=True)
        if pool_after:                                  # BN + ReLU applied inside the pooling window
            return ops.maxpool3x3s2(y, scale, shift)
        if (then is not None and relu and res is None and u.cout_p <= 256 and then.cin_p == u.cout_p
                and self._gram_route(then, y.numel() // u.cout_p)):
            # the consumer is an expansion conv on the Gram route: normalise in place AND accumulate its Gram partials in one
            # sweep over y (`sr_bn_apply_gram`) instead of bn_apply now and a second read of the same tensor by sr_gram
            if self.lazy_bn2 and self._lazy_ok(y, then):
                # ... and the normalised tensor need not exist at all: the Gram sweep normalises in LDS only and the expansion conv
                # applies the same scale / shift + ReLU to the raw tensor on load (one write + two reads of y instead of two + two)
                self._gram_stash = (y.data_ptr(), ops.bn_gram(y.view(-1, u.cout_p), scale, shift), (scale, shift))
                return y
            self._gram_stash = (y.data_ptr(), ops.bn_apply_gram(y.view(-1, u.cout_p), scale, shift), None)
            return y
        if (self.lazy_bn2 and then is not None and relu and res is None and then.k == 3 and y.dtype == torch.bfloat16 and not self.fp8
                and ops.conv_in_affine_supported(y, then.cout_p, 3, then.stride, then.pad, res=None, relu=False, want_stats=True)):
            # the consumer is the 3x3 that stages its whole input patch once (layer1): it normalises the raw tensor in LDS
            self._lazy_in = (y.data_ptr(), (scale, shift))
            return y
        return ops.bn_apply(y, scale, shift, res=res, relu=relu, out=y)

    def _next_reduce(self, bi, train):
        """The conv1 unit of block bi + 1 when it may run fused with block bi's expansion conv (train and eval mode): both blocks bottlenecks.
        (Across a layer boundary the next block's downsample branch reads block bi's output as well: the output is written either
        way, what the fused launch saves is conv1's read of it.)  Shapes are checked by `_pair_ok`."""
        blocks = self._plan()[1]
        if not self.fuse_pairs or bi + 1 >= len(blocks):
            return None
        convs, ds = blocks[bi + 1]
        return convs[0] if len(convs) == 3 and len(blocks[bi][0]) == 3 else None

    def _run_block(self, a, bi, train, momentum, tblock=None, T=lambda tu: None, pre=None, fuse_next=False):
        """One residual block on the NHWC activation `a` (torchvision Bottleneck / BasicBlock: conv-BN-ReLU chain, optional
        1x1 downsample of the identity, add, ReLU).  `tblock`: the same block of a weight-identical twin backbone.
        `pre`: (raw conv1 output, statistics) when the previous block's fused launch already ran this block's conv1.
        `fuse_next`: try to run the NEXT block's conv1 inside this block's expansion launch; the return value is then (out, pre for
        the next block or None)."""
        convs, ds = self._plan()[1][bi]
        tconvs, tds = tblock if tblock is not None else ([None] * len(convs), None)
        idn = a if ds is None else self._unit(a, ds, train, momentum, relu=False, twin=T(tds))
        y = a
        for i, u in enumerate(convs[:-1]):
            q8 = self.fp8 and len(convs) == 3 and i == 0 and convs[1].fp8_eligible()
            y = self._unit(y, u, train, momentum, relu=True, then=convs[i + 1], twin=T(tconvs[i]), quant_out=q8, pre=pre if i == 0 else None)
        nxt = self._next_reduce(bi, train) if fuse_next else None
        out = self._unit(y, convs[-1], train, momentum, relu=True, res=idn, twin=T(tconvs[-1]), fuse_next=nxt)
        if not fuse_next:
            return out
        return out if isinstance(out, tuple) else (out, None)   # (a unit that does not take the two-launch route ignores fuse_next)

    def num_blocks(self):
        return len(self._plan()[1])

    def stem_forward(self, x):
        """conv1 -> bn1 -> relu -> maxpool alone (the teacher-forced per-block parity tests feed every block the oracle's input):
        image batch -> NHWC activation [B, H/4, W/4, 64]."""
        stem = self._plan()[0]
        m = self.model.bn1.momentum if self.model.bn1.momentum is not None else 0.1
        with torch.no_grad():
            xp, H, W = self.prepare_input(x)
            a = self._unit(xp, stem, self.training, m, relu=True, stem_hw=(H, W), pool_after=True)
        if self.training:
            self._stats_epoch += 1
            self.model.bn1.num_batches_tracked += 1
        return a

    def block_forward(self, a, bi, fuse_next=False, pre=None):
        """Residual block `bi` (0 .. num_blocks()-1, torchvision order layer1.0 ... layer4.2) alone, through exactly the launches
        the full pass makes for it: NHWC activation in the backbone's dtype -> NHWC activation.  Train mode updates that block's
        running statistics once.
        `fuse_next`: as the full train-mode pass does, run the NEXT block's conv1 inside this block's expansion launch where
        `sr_conv_pair` serves the pair; returns (output, pre) with pre = (raw conv1 output of block bi + 1, its statistics partials)
        or None.  `pre`: such a pair from block bi - 1: this block's conv1 has already run."""
        if not a.is_cuda or a.dtype != self.dtype or a.dim() != 4:
            raise SrError("block_forward expects an NHWC activation in the backbone's dtype on the GPU")
        m = self.model.bn1.momentum if self.model.bn1.momentum is not None else 0.1
        self._gram_stash = self._lazy_in = None
        with torch.no_grad():
            out = self._run_block(a.contiguous(), bi, self.training, m, pre=pre, fuse_next=fuse_next)
        if self.training:
            self._stats_epoch += 1
            convs, ds = self._plan()[1][bi]
            for u in convs + ([ds] if ds is not None else []):
                u.bn.num_batches_tracked += 1
        return out

    def forward(self, x, bn_updates=1, twin=None, twin_updates=0, prepped=None):
        """`bn_updates`=2 gives the running-statistics state of two consecutive train-mode passes over the
        same batch in one pass (FCGGNN.forward runs convnet_nouns twice on the same images, model.py:176-178).
        `twin`: a second `resnet` with IDENTICAL weights (train mode only): its BatchNorm buffers receive `twin_updates`
        passes' worth of the same batch statistics, so this one pass stands for the twin's passes too."""
        if self.use_graphs and x.is_cuda and twin is None and ops.PROFILE is None:
            if not self.training:
                return self._graph_forward(x)
            if self.graph_train:
                return self._graph_forward_train(x, bn_updates)
        return self._forward_impl(x, bn_updates, twin, twin_updates, prepped)

    def weights_equal(self, other):
        """Are all convolution / BatchNorm affine parameters of the two backbones identical?  (The reference loads the same
        `pretrained=True` weights into both, model.py:16,100-101, and freezes them, model.py:17-18.)  Cached on the
        parameters' versions and addresses: the comparison itself runs once."""
        mine, theirs = list(self.model.parameters()), list(other.model.parameters())
        key = (id(other), tuple((t._version, t.data_ptr()) for t in mine), tuple((t._version, t.data_ptr()) for t in theirs))
        hit = getattr(self, "_equal_cache", None)
        if hit is None or hit[0] != key:
            same = (self.depth == other.depth and self.dtype == other.dtype and len(mine) == len(theirs)
                    and all(a.shape == b.shape and a.device == b.device for a, b in zip(mine, theirs)))
            if same:
                with torch.no_grad():
                    same = not bool(torch.stack([(a != b).any() for a, b in zip(mine, theirs)]).any().item())
            hit = self._equal_cache = (key, same)
        return hit[1]

    # -- eval-mode pass replayed from a hipGraph (single-image inference is launch-bound: ~320 kernel launches of a few
    #    microseconds each; one graph launch replaces them)
    def _graph_forward(self, x):
        key = (tuple(x.shape), x.dtype, self.dtype, self._weights_signature(), self._stats_epoch, ops.cu_share())
        hit = self._graphs.get(key)
        if hit is None:
            self._forward_impl(x, 1)                        # eager warm-up: builds the folded packs, sets kernel attributes
            torch.cuda.synchronize()
            static_in = (x if x.dtype == torch.uint8 else x.detach().float()).contiguous().clone()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                static_out = self._forward_impl(static_in, 1)
            if len(self._graphs) >= 4:
                self._graphs.pop(next(iter(self._graphs)))
            hit = self._graphs[key] = (g, static_in, static_out)
        g, static_in, static_out = hit
        static_in.copy_(x)
        g.replay()
        self.graph_replays += 1
        return static_out.clone()

    def _graph_forward_train(self, x, bn_updates):
        """Train-mode pass (batch statistics, running-statistics update) replayed from a hipGraph.  The backbone is frozen and
        runs under no_grad, every launch of the pass is device-side work on persistent parameters / buffers and graph-private
        temporaries, so the captured graph IS the pass; what stays in Python is the bookkeeping of num_batches_tracked.  The first
        call with a given shape runs eagerly (that is the call's result) and only then captures -- capturing executes nothing, so
        the running statistics are not updated twice."""
        key = ("train", tuple(x.shape), x.dtype, self.dtype, self._weights_signature(buffers=False),
               tuple(t.data_ptr() for t in self.model.buffers()), bn_updates, ops.cu_share())
        hit = self._graphs.get(key)
        if hit is None:
            out = self._forward_impl(x, bn_updates)
            torch.cuda.synchronize()
            static_in = (x if x.dtype == torch.uint8 else x.detach().float()).contiguous().clone()
            g = torch.cuda.CUDAGraph()
            self._capturing = True
            try:
                with torch.cuda.graph(g):
                    static_out = self._forward_impl(static_in, bn_updates)
            finally:
                self._capturing = False
            if len(self._graphs) >= 4:
                self._graphs.pop(next(iter(self._graphs)))
            self._graphs[key] = (g, static_in, static_out)
            return out
        g, static_in, static_out = hit
        static_in.copy_(x)
        g.replay()
        self.graph_replays += 1
        self._stats_epoch += 1
        self._pending_tracked += bn_updates
        return static_out.clone()

    def prepare_input(self, x):
        """The stem's input layout (zero-padded NHWC4 in the backbone's dtype) of an image batch: fp32 [B,3,H,W] (reference layout) or
        decoded uint8 [B,H,W,3] (ToTensor + Normalize fused in).  Returns (tensor, H, W); `forward(..., prepped=)` takes it, so two
        backbones fed the same images share one layout pass."""
        if not x.is_cuda:
            raise SrError("situation_recognition_amd.resnet runs on an MI355X only (got a CPU tensor; no CPU fallback)")
        u8 = x.dtype == torch.uint8
        if x.dim() != 4 or (x.shape[3] if u8 else x.shape[1]) != 3:
            raise SrError("expected an image batch: fp32 [B,3,H,W] (reference layout) or decoded uint8 [B,H,W,3]")
        with torch.no_grad():
            if u8:      # decoded images: ToTensor + Normalize fused into the stem's layout kernel (no fp32 batch)
                return ops.image_prep_u8(x.contiguous(), self.dtype), x.shape[1], x.shape[2]
            return ops.stem_prep(x.float().contiguous(), self.dtype), x.shape[2], x.shape[3]

    def _forward_impl(self, x, bn_updates=1, twin=None, twin_updates=0, prepped=None):
        if not x.is_cuda:
            raise SrError("situation_recognition_amd.resnet runs on an MI355X only (got a CPU tensor; no CPU fallback)")
        u8 = x.dtype == torch.uint8
        if x.dim() != 4 or (x.shape[3] if u8 else x.shape[1]) != 3:
            raise SrError("expected an image batch: fp32 [B,3,H,W] (reference layout) or decoded uint8 [B,H,W,3]")
        stem, blocks = self._plan()
        train = self.training
        bn0 = self.model.bn1
        m = bn0.momentum if bn0.momentum is not None else 0.1
        momentum = 1.0 - (1.0 - m) ** bn_updates
        if twin is not None and not (train and twin.training and twin_updates > 0):
            raise SrError("a twin backbone is only meaningful in train mode (eval folds each backbone's own running statistics)")
        tstem, tblocks, tm = None, None, 0.0
        if twin is not None:
            tstem, tblocks = twin._plan()
            tm = 1.0 - (1.0 - m) ** twin_updates
        T = lambda tu: None if twin is None else (tu, tm)
        with torch.no_grad():
            xp, H, W = prepped if prepped is not None else self.prepare_input(x)
            a = self._unit(xp, stem, train, momentum, relu=True, stem_hw=(H, W), pool_after=True, twin=T(tstem))
            pre = None
            for bi in range(len(blocks)):
                a, pre = self._run_block(a, bi, train, momentum, tblocks[bi] if twin is not None else None, T, pre=pre, fuse_next=True)
            feat = ops.avgpool(a)
        if train and not self._capturing:
            self._stats_epoch += 1
            self._pending_tracked += bn_updates
            if twin is not None:
                twin._stats_epoch += 1
                twin._pending_tracked += twin_updates
        return feat[:, : self.out_features] if feat.shape[1] != self.out_features else feat


# ----------------------------------------------------------------------------- GGNN
class _Shadow:
    """Storage-dtype (and transposed) copies of the fp32 master parameters for the MFMA kernels,
    refreshed when the parameter changes (optimizer step, load_state_dict, .to())."""

    def __init__(self):
        self._c = {}

    def get(self, p, dtype, transposed=False, pad_to=1):
        key = (id(p), dtype, transposed, pad_to)
        sig = (p.data_ptr(), p._version)
        hit = self._c.get(key)
        cur = torch.cuda.current_stream(p.device) if p.is_cuda else None
        if hit is None or hit[0] != sig:
            with torch.no_grad():
                d = p.detach()
                if transposed:
                    t = ops.transpose(d, out_dtype=dtype, pad_to=pad_to)
                else:
                    t = d if dtype == torch.float32 else ops.cast(d, dtype)
            ev = None
            if cur is not None and t is not d:            # a copy made on `cur`: another stream must wait for it before reading
                ev = torch.cuda.Event()
                ev.record(cur)
            hit = (sig, t, ev, cur)
            self._c[key] = hit
        elif hit[2] is not None and cur != hit[3]:
            cur.wait_event(hit[2])
            hit[1].record_stream(cur)
        return hit[1]


def _kpad(dtype):
    return 32 if dtype == torch.float32 else 64


class _GGNNFunction(torch.autograd.Function):
    """T steps of model.py:60-84 with a hand-written backward.

    forward per step (4 GEMM launches + 1 aggregate, no standalone elementwise kernel):
        agg = A.h                                   sr_ggnn_aggregate        (verb path: agg = h)
        n   = agg W_p^T + (R|1) b_p                 sr_gemm
        z   = sigmoid(n W_z^T + h U_z^T + b)        sr_gemm, 2 operand pairs, sigmoid epilogue
        r   = sigmoid(n W_r^T + h U_r^T + b), r*h   sr_gemm, SIGMOID_MUL epilogue
        c   = tanh(n W_h^T + (r*h) U_h^T + b), h' = (1-z) h + z c     sr_gemm, TANH_BLEND epilogue
    """

    @staticmethod
    def forward(ctx, h0, adj, idx, R, verb, steps, shadow, offs, *params):
        (Wp, bp, Wz, bz, Uz, buz, Wr, br, Ur, bur, Wh, bh, Uh, buh) = params
        dt = h0.dtype
        g = lambda p: shadow.get(p, dt)
        M, D = h0.shape
        # Everything the backward needs is kept STACKED over the steps ([T, M, D], step t = slice t; the GEMMs write their outputs
        # straight into the slices): a weight gradient is then ONE TN GEMM over all T * M rows -- dW_z = [dz_0; ..; dz_T-1]^T [n_0; ..;
        # n_T-1] -- instead of T launches of M rows each plus T partial reductions (round 4: at the 8-GPU share the verb path's M is
        # 768 rows, six K-steps per slice: 70 such launches per step cost 3 ms for almost no arithmetic).
        if not any(ctx.needs_input_grad):
            # evaluation / no_grad: nothing is kept for a backward -- two ping-pong state buffers and one set of per-step temporaries
            # instead of 6-7 tensors of [T, M, D] (several GB at T = 8, 36 864 rows), and the result is a tensor of its own
            hbuf = [h0, torch.empty_like(h0), torch.empty_like(h0)]
            agg_b = None if verb else torch.empty_like(h0)
            n_b, z_b, r_b, rh_b, c_b = (torch.empty_like(h0) for _ in range(5))
            h = h0
            for t in range(steps):
                agg = h if verb else ops.aggregate(h, adj, idx, R, offs=offs, out=agg_b)
                n = ops.gemm([(agg, g(Wp))], bias=bp, bias_scale=1.0 if verb else float(R), out=n_b)
                z = ops.gemm([(n, g(Wz)), (h, g(Uz))], bias=bz, bias2=buz, act=ops.ACT_SIGMOID, out=z_b)
                _, rh = ops.gemm([(n, g(Wr)), (h, g(Ur))], bias=br, bias2=bur, act=ops.ACT_SIGMOID_MUL, aux1=h, out=r_b, out2=rh_b)
                nxt = hbuf[1 + (t & 1)]
                ops.gemm([(n, g(Wh)), (rh, g(Uh))], bias=bh, bias2=buh, act=ops.ACT_TANH_BLEND, aux1=h, aux2=z, out=nxt, out2=c_b)
                h = nxt
            return h if steps > 0 else h0.clone()
        H = torch.empty((steps + 1, M, D), device=h0.device, dtype=dt)           # h_0 .. h_T
        H[0].copy_(h0)
        AGG = None if verb else torch.empty((steps, M, D), device=h0.device, dtype=dt)
        N_, Z, R_, RH, C_ = (torch.empty((steps, M, D), device=h0.device, dtype=dt) for _ in range(5))
        for t in range(steps):
            h = H[t]
            agg = h if verb else ops.aggregate(h, adj, idx, R, offs=offs, out=AGG[t])
            n = ops.gemm([(agg, g(Wp))], bias=bp, bias_scale=1.0 if verb else float(R), out=N_[t])
            z = ops.gemm([(n, g(Wz)), (h, g(Uz))], bias=bz, bias2=buz, act=ops.ACT_SIGMOID, out=Z[t])
            _, rh = ops.gemm([(n, g(Wr)), (h, g(Ur))], bias=br, bias2=bur, act=ops.ACT_SIGMOID_MUL, aux1=h, out=R_[t], out2=RH[t])
            ops.gemm([(n, g(Wh)), (rh, g(Uh))], bias=bh, bias2=buh, act=ops.ACT_TANH_BLEND, aux1=h, aux2=z, out=H[t + 1], out2=C_[t])
        ctx.meta = (R, verb, steps, shadow, adj, idx, offs)
        saved = (H, N_, Z, R_, RH, C_) + (() if verb else (AGG,))
        ctx.save_for_backward(*saved, *params)
        # (a VIEW of the saved stack: it shares the stack's version counter, so the result must not be modified in place -- autograd
        #  would refuse the backward; every caller in this package only reads it)
        return H[steps]

    @staticmethod
    def backward(ctx, dh):
        R, verb, steps, shadow, adj, idx, offs = ctx.meta
        tensors = ctx.saved_tensors
        ns = 6 if verb else 7
        saved, params = tensors[:ns], tensors[ns:]
        H, N_, Z, R_, RH, C_ = saved[:6]
        AGG = None if verb else saved[6]
        (Wp, bp, Wz, bz, Uz, buz, Wr, br, Ur, bur, Wh, bh, Uh, buh) = params
        dt = H.dtype
        kp = _kpad(dt)
        gT = lambda p: shadow.get(p, dt, transposed=True)
        D = Wp.shape[0]
        dev = dh.device
        M = H.shape[1]
        # bf16, D a multiple of 256: dW = dY^T X straight from the row-major operands (`sr_gemm_tn`: transposed LDS reads, rows
        # cut into slices so that each 2048 x 2048 gradient fills the chip) -- no transposed copies -- and ONE launch per weight matrix
        # over the T stacked steps.  fp32 storage / narrow test models: NT GEMMs over transposed operands, step by step
        # (the four z/r matrices out of ONE GEMM per step, [dz^T; dr^T] x [h^T; n^T]; the two candidate matrices out of dc^T x [n^T; (r*h)^T]).
        tn = dt == torch.bfloat16 and D % 256 == 0
        if tn:
            DZ, DR, DC, DN = (torch.empty((steps, M, D), device=dev, dtype=dt) for _ in range(4))
        else:
            gZR = torch.zeros(2 * D, 2 * D, device=dev, dtype=torch.float32)      # rows: z | r ; cols: U (h) | W (n)
            gC = torch.zeros(D, 2 * D, device=dev, dtype=torch.float32)           # cols: W_h (n) | U_h (r*h)
            gP = torch.zeros(D, D, device=dev, dtype=torch.float32)
        gb = {k: torch.zeros(D, device=dev, dtype=torch.float32) for k in ("p", "z", "r", "h")}
        dh = dh.contiguous()
        if dh.dtype != dt:
            dh = ops.cast(dh, dt)
        Mp = (M + kp - 1) // kp * kp
        for t in reversed(range(steps)):
            h, n, z, r, rh, c = H[t], N_[t], Z[t], R_[t], RH[t], C_[t]
            agg = h if verb else AGG[t]
            if tn:
                dc, dz, dacc = ops.gru_bwd1(dh, z, c, h, dc=DC[t], dz=DZ[t])
            else:
                dc, dz, dacc = ops.gru_bwd1(dh, z, c, h)
            drh = ops.gemm([(dc, gT(Uh))])
            dr = ops.gru_bwd2(drh, r, h, dacc, dr=DR[t] if tn else None)     # dacc += drh * r
            dn = ops.gemm([(dc, gT(Wh)), (dz, gT(Wz)), (dr, gT(Wr))], out=DN[t] if tn else None)
            dacc = ops.gemm([(dz, gT(Uz)), (dr, gT(Ur))], res=dacc, out=dacc)
            if verb:
                dh = ops.gemm([(dn, gT(Wp))], res=dacc)
            else:
                dagg = ops.gemm([(dn, gT(Wp))])
                dh = ops.aggregate(dagg, adj, idx, R, transpose=True, add=dacc, offs=offs)
            if tn:
                continue
            # transposed operands (dW = dY^T X as NT GEMMs); bias gradients (column sums) are reduced inside the transpose kernel
            YT = torch.empty((2, D, Mp), device=dev, dtype=dt)         # dz^T, dr^T
            XT = torch.empty((3, D, Mp), device=dev, dtype=dt)         # h^T, n^T, (r*h)^T
            ops.transpose(dz, colsum=gb["z"], pad_to=kp, out=YT[0])
            ops.transpose(dr, colsum=gb["r"], pad_to=kp, out=YT[1])
            dcT = ops.transpose(dc, colsum=gb["h"], pad_to=kp)
            dnT = ops.transpose(dn, colsum=gb["p"], colsum_scale=1.0 if verb else float(R), pad_to=kp)
            ops.transpose(h, pad_to=kp, out=XT[0])
            ops.transpose(n, pad_to=kp, out=XT[1])
            ops.transpose(rh, pad_to=kp, out=XT[2])
            aggT = XT[0] if verb else ops.transpose(agg, pad_to=kp)
            ops.gemm([(YT.view(2 * D, Mp), XT[0:2].view(2 * D, Mp))], res=gZR, out=gZR, out_f32=True)
            ops.gemm([(dcT, XT[1:3].view(2 * D, Mp))], res=gC, out=gC, out_f32=True)
            ops.gemm([(dnT, aggT)], res=gP, out=gP, out_f32=True)
        if tn:
            TM = steps * M
            flat = lambda x: x.view(TM, D)
            Hs, Ns, RHs = H[:steps].reshape(TM, D), flat(N_), flat(RH)      # (H[:steps] is a contiguous prefix: a view)
            dz, dr, dc, dn = flat(DZ), flat(DR), flat(DC), flat(DN)
            ops.colsum(dz, gb["z"]); ops.colsum(dr, gb["r"]); ops.colsum(dc, gb["h"])
            ops.colsum(dn, gb["p"], scale=1.0 if verb else float(R))
            gW = {k: torch.empty(D, D, device=dev, dtype=torch.float32) for k in ("Wp", "Wz", "Uz", "Wr", "Ur", "Wh", "Uh")}
            ops.gemm_tn(dz, Ns, gW["Wz"], accumulate=False); ops.gemm_tn(dz, Hs, gW["Uz"], accumulate=False)
            ops.gemm_tn(dr, Ns, gW["Wr"], accumulate=False); ops.gemm_tn(dr, Hs, gW["Ur"], accumulate=False)
            ops.gemm_tn(dc, Ns, gW["Wh"], accumulate=False); ops.gemm_tn(dc, RHs, gW["Uh"], accumulate=False)
            ops.gemm_tn(dn, Hs if verb else flat(AGG), gW["Wp"], accumulate=False)
            grads = (gW["Wp"], gb["p"], gW["Wz"], gb["z"], gW["Uz"], gb["z"].clone(), gW["Wr"], gb["r"], gW["Ur"], gb["r"].clone(),
                     gW["Wh"], gb["h"], gW["Uh"], gb["h"].clone())
            return (dh, None, None, None, None, None, None, None) + grads
        blk = lambda g_, i, j: g_[i * D:(i + 1) * D, j * D:(j + 1) * D].contiguous()
        grads = (gP, gb["p"], blk(gZR, 0, 1), gb["z"], blk(gZR, 0, 0), gb["z"].clone(), blk(gZR, 1, 1), gb["r"], blk(gZR, 1, 0),
                 gb["r"].clone(), blk(gC, 0, 0), gb["h"], blk(gC, 0, 1), gb["h"].clone())
        return (dh, None, None, None, None, None, None, None) + grads


class GGSNN(nn.Module):
    """reference model.py:38-86.  `forward(hidden_state, mask=None, verb=False)` keeps the reference
    signature (mask: [B,R,R] adjacency per image); `steps` generalises the hard-coded 4 (model.py:60)."""

    _ORDER = ("W_p", "W_z", "U_z", "W_r", "U_r", "W_h", "U_h")

    def __init__(self, layersize, steps=4):
        super().__init__()
        for n in self._ORDER:                                      # model.py:47-56
            setattr(self, n, nn.Linear(layersize, layersize))
        self.steps = steps
        self._shadow = _Shadow()

    def _params(self):
        return tuple(itertools.chain.from_iterable((getattr(self, n).weight, getattr(self, n).bias) for n in self._ORDER))

    def run(self, hidden_state, adj_table, verbs, R, verb, offs=None):
        """`offs` (int32 [B+1]): `hidden_state` holds PACKED role rows -- see FCGGNN._pack_plan / sr_node_init_fwd."""
        if not hidden_state.is_cuda:
            raise SrError("GGSNN runs on an MI355X only (got a CPU tensor; no CPU fallback)")
        return _GGNNFunction.apply(hidden_state, adj_table, verbs, R, verb, self.steps, self._shadow, offs, *self._params())

    def forward(self, hidden_state, mask=None, verb=False):
        if verb:
            return self.run(hidden_state, None, None, 1, True)
        B, R = mask.shape[0], mask.shape[1]
        idx = torch.arange(B, device=mask.device, dtype=torch.int64)
        return self.run(hidden_state, mask.float().contiguous(), idx, R, False)


# ----------------------------------------------------------------------------- head pieces
class _NodeInitFunction(torch.autograd.Function):
    """model.py:117-144 (encoder lookup + two embedding gathers + product + ReLU) in one kernel; the backward
    scatters into the two embedding gradients.  The image features get no gradient (frozen backbone)."""

    @staticmethod
    def forward(ctx, feat, role_w, verb_w, verbs, role_table, offs=None, rows=None):
        ctx.save_for_backward(feat, role_w, verb_w, verbs, role_table)
        ctx.offs = offs
        return ops.node_init_fwd(feat.contiguous(), role_w, verb_w, verbs, role_table, offs=offs, rows=rows)

    @staticmethod
    def backward(ctx, dnode):
        feat, role_w, verb_w, verbs, role_table = ctx.saved_tensors
        d_role, d_verb = torch.empty_like(role_w), torch.empty_like(verb_w)     # written in full, fixed summation order
        dnode = dnode.contiguous()
        if dnode.dtype != feat.dtype:
            dnode = ops.cast(dnode, feat.dtype)
        ops.node_init_bwd(dnode, feat.contiguous(), role_w, verb_w, verbs, role_table, d_role, d_verb, offs=ctx.offs)
        return None, d_role, d_verb, None, None, None, None


class _ExpandRowsFunction(torch.autograd.Function):
    """Packed classifier output [rows + 1, L] -> the reference's full [B*R, L]: every real role takes its own row, every padded slot
    the shared last row.  Backward: a real role's gradient goes to its row; the shared row receives the SUM of all padded slots'
    gradients (zero whenever the loss ignores padded targets, as the reference's `ignore_index` does -- model.py:196-199 -- but exact
    in any case; fixed-order reduction, no atomics)."""

    @staticmethod
    def forward(ctx, packed, packed_of_full, valid, pad_rows):
        ctx.save_for_backward(valid, pad_rows)
        ctx.rows = packed.shape[0]
        return packed.index_select(0, packed_of_full)

    @staticmethod
    def backward(ctx, d_full):
        valid, pad_rows = ctx.saved_tensors
        d = torch.empty((ctx.rows, d_full.shape[1]), device=d_full.device, dtype=d_full.dtype)
        torch.index_select(d_full, 0, valid, out=d[:-1])
        if pad_rows.numel():
            d[-1] = d_full.index_select(0, pad_rows).sum(0)
        else:
            d[-1].zero_()
        return d, None, None, None


class _ClassifierFunction(torch.autograd.Function):
    """Dropout(0.5) + Linear (model.py:105-111): counter-hash dropout kernel, MFMA GEMM with fp32 logits."""

    @staticmethod
    def forward(ctx, x, W, b, shadow, drop_seed):
        dt = x.dtype
        xd = ops.dropout_half(x.contiguous(), drop_seed) if drop_seed is not None else x.contiguous()
        N = W.shape[0]
        ld = (N + 63) // 64 * 64
        buf = torch.empty((x.shape[0], ld), device=x.device, dtype=torch.float32)
        logits = ops.gemm([(xd, shadow.get(W, dt))], bias=b, out=buf[:, :N], out_f32=True)
        ctx.meta = (shadow, drop_seed, dt)
        ctx.save_for_backward(xd, W)
        return logits

    @staticmethod
    def backward(ctx, dy):
        shadow, drop_seed, dt = ctx.meta
        xd, W = ctx.saved_tensors
        N, kp = W.shape[0], _kpad(dt)
        if dy.stride(1) != 1:
            dy = dy.contiguous()
        db = torch.zeros(N, device=dy.device, dtype=torch.float32)
        ops.colsum(dy, db)
        tn = dt == torch.bfloat16 and W.shape[1] % 256 == 0
        pad = 256 if tn else 64
        dyp = ops.cast_pad(dy, dt, pad_to=pad)                        # [M, Npad] storage dtype, zero padded
        dx = ops.gemm([(dyp, shadow.get(W, dt, transposed=True, pad_to=pad))])
        if drop_seed is not None:
            dx = ops.dropout_half(dx, drop_seed)
        if tn:                                                        # dW = dy^T x from the row-major operands
            dWp = torch.empty((dyp.shape[1], W.shape[1]), device=dy.device, dtype=torch.float32)
            ops.gemm_tn(dyp, xd, dWp, accumulate=False)
            return dx, dWp[:N], db, None, None
        dyT = ops.transpose(dyp, pad_to=kp)                           # [Npad, Mpad]
        xT = ops.transpose(xd, pad_to=kp)                             # [D, Mpad]
        dW = ops.gemm([(dyT[:N], xT)], out_f32=True)
        return dx, dW, db, None, None


class FCGGNN(nn.Module):
    """reference model.py:89-201: verb path and role-graph noun path over two frozen backbones.

    Differences that do not change results: (1) `forward` runs convnet_nouns once and reuses the features for
    both noun branches (the reference recomputes identical features, model.py:176-178) while giving the BatchNorm
    running statistics the state two passes would leave; (2) relu() on the pooled verb features (model.py:160) is
    skipped because a global average of post-ReLU activations is already non-negative.
    """

    def __init__(self, encoder, D_hidden_state, steps=4, backbone=152, dtype=torch.bfloat16, width=64, blocks=None, fp8=False):
        super().__init__()
        self.encoder = encoder
        self.dtype = dtype
        nr, nv, nl = encoder.get_num_roles(), encoder.get_num_verbs(), encoder.get_num_labels()
        self.role_emb = nn.Embedding(nr + 1, D_hidden_state, padding_idx=nr)          # model.py:95-97
        self.verb_emb = nn.Embedding(nv, D_hidden_state)                                # model.py:98
        self.convnet_verbs = resnet(nv, backbone, width, blocks, dtype, fp8=fp8)        # model.py:100
        self.convnet_nouns = resnet(nl, backbone, width, blocks, dtype, fp8=fp8)        # model.py:101
        if self.convnet_verbs.out_features != D_hidden_state:
            raise SrError("D_hidden_state (%d) must equal the backbone feature width (%d)"
                          % (D_hidden_state, self.convnet_verbs.out_features))
        self.ggsnn = GGSNN(layersize=D_hidden_state, steps=steps)                       # model.py:103
        self.verb_classifier = nn.Sequential(nn.Dropout(0.5), nn.Linear(D_hidden_state, nv))     # model.py:105-107
        self.nouns_classifier = nn.Sequential(nn.Dropout(0.5), nn.Linear(D_hidden_state, nl))    # model.py:109-111
        self._shadow = _Shadow()
        self._drop_counter = 0
        self.drop_seed_base = 0x5eed
        # noun backbone on a second stream (see forward): True / False / None = on (worth 8 % at per-GPU batch 768, 4 % at 1536
        # and 3072, 3 % at 6144; profiling legs switch it off so that a kernel's duration does not depend on its neighbour's)
        env = os.environ.get("SR_OVERLAP")
        self.overlap_backbones = None if env is None else env not in ("0", "")
        self._side_streams = {}
        self._noun_feat_cache = None
        # packed role rows (see _nouns_from_features): None = automatic (noun paths of at least pack_roles_min_rows rows: one host
        # synchronisation per forward is cheaper than the padded slots' GEMM rows from there on), True / False force it
        env = os.environ.get("SR_PACK_ROLES")
        self.pack_roles = None if env in (None, "", "auto") else env not in ("0",)
        self.pack_roles_min_rows = int(os.environ.get("SR_PACK_ROLES_MIN_ROWS", "6144"))
        self._pad_ok, self._pad_check_epoch = None, 0
        self.register_load_state_dict_post_hook(lambda m, incompatible: setattr(m, "_pad_check_epoch", m._pad_check_epoch + 1))
        self.overlap_gt_branch = os.environ.get("SR_OVERLAP_GT", "1") not in ("0", "")     # see forward()
        self.backbone_cu_share = int(os.environ.get("SR_BACKBONE_CU_SHARE", "2"))   # see forward(): 2 = each overlapped pass on half the CUs
        # one train-mode pass for both backbones while their (frozen) weights are identical -- see forward()
        self.share_identical_backbones = os.environ.get("SR_SHARE_BACKBONES", "1") not in ("0", "")

    def enable_graphs(self, on=True, train=False):
        """Replay the eval-mode backbone passes from captured hipGraphs (latency path for single-image inference); with
        `train=True` the train-mode passes as well (small per-GPU batches, where the ~900 launches of a pass are a few
        microseconds apart: 3 ms of a 39 ms pass at batch 768).  Pass sharing (`share_identical_backbones`) stays eager."""
        for net in (self.convnet_verbs, self.convnet_nouns):
            net.use_graphs = on
            net.graph_train = bool(on and train)
        return self

    # -- helpers
    def _drop_seed(self, p):
        if not self.training or p == 0.0:
            return None
        if p != 0.5:
            raise SrError("only Dropout(0.5) (the reference's value) or p=0 is implemented")
        self._drop_counter += 1
        return (self.drop_seed_base * 0x9E3779B1 + self._drop_counter * 0x85EBCA77) & (2 ** 63 - 1)

    _DRAW = object()         # "draw the dropout seed now" (the reference's call order: verb, predicted-verb nouns, gt-verb nouns)

    def _classify(self, seq, x, seed=_DRAW):
        lin = seq[1]
        if seed is FCGGNN._DRAW:
            seed = self._drop_seed(seq[0].p)
        return _ClassifierFunction.apply(x, lin.weight, lin.bias, self._shadow, seed)

    def _pack_plan(self, verbs, B, R):
        """Packing of a batch's role rows (see sr_node_init_fwd in include/srhip.h): (offs int32 [B+1], rows, full row index of every
        packed row, packed row of every full row, full row indices of the padded slots).  ONE host synchronisation (the number of
        real roles sizes every tensor of the noun path); the noun backbone keeps the GPU busy on its own stream meanwhile."""
        dev = verbs.device
        counts = self.encoder.device_tables(dev)[2][verbs]
        offs = torch.zeros(B + 1, device=dev, dtype=torch.int32)
        offs[1:] = torch.cumsum(counts, 0)
        real = (torch.arange(R, device=dev)[None, :] < counts[:, None]).reshape(-1)
        valid = real.nonzero().squeeze(1)                                               # (the synchronisation)
        rows = valid.numel()
        packed_of_full = torch.full((B * R,), rows, device=dev, dtype=torch.int64)
        packed_of_full[valid] = torch.arange(rows, device=dev)
        pad_rows = (~real).nonzero().squeeze(1)
        return offs, rows, valid, packed_of_full, pad_rows

    def _pad_row_is_zero(self):
        """The packed form relies on role_emb's padding row being exactly zero (nn.Embedding(padding_idx) creates it so and never
        updates it; a state dict could carry anything).  Checked once per parameter storage / load."""
        w = self.role_emb.weight
        key = (w.data_ptr(), self._pad_check_epoch)
        if self._pad_ok is None or self._pad_ok[0] != key:
            self._pad_ok = (key, not bool((w.detach()[self.role_emb.padding_idx] != 0).any().item()))
        return self._pad_ok[1]

    def _use_packed(self, B, R):
        if self.pack_roles is not None:
            on = bool(self.pack_roles)
        else:
            on = B * R >= self.pack_roles_min_rows
        return on and R > 1 and self._pad_row_is_zero()

    def _nouns_from_features(self, feat, verbs, batch_size, seed=_DRAW):
        dev = feat.device
        role_table, adj_table, _ = self.encoder.device_tables(dev)
        verbs = verbs.to(device=dev, dtype=torch.int64).contiguous()
        R = self.encoder.get_max_role_count()
        if self._use_packed(batch_size, R):
            # Only the REAL roles' rows go through node init, the T GGNN steps and the classifier; all padded slots share one row
            # (identical results: see sr_node_init_fwd).  imSitu verbs have 3.5 of 6 roles on average: 0.6 of the noun path's rows.
            offs, rows, valid, packed_of_full, pad_rows = self._pack_plan(verbs, batch_size, R)
            node = _NodeInitFunction.apply(feat, self.role_emb.weight, self.verb_emb.weight, verbs, role_table, offs, rows)
            out = self.ggsnn.run(node, adj_table, verbs, R, False, offs=offs)
            packed = self._classify(self.nouns_classifier, out, seed)
            logits = _ExpandRowsFunction.apply(packed, packed_of_full, valid, pad_rows)
            return logits.reshape(batch_size, R, -1)
        node = _NodeInitFunction.apply(feat, self.role_emb.weight, self.verb_emb.weight, verbs, role_table)
        out = self.ggsnn.run(node, adj_table, verbs, R, False)                          # model.py:151
        logits = self._classify(self.nouns_classifier, out, seed)                       # model.py:152
        return logits.reshape(batch_size, R, -1)                                        # model.py:155

    # -- reference surface
    def predict_nouns(self, img, gt_verb, batch_size):                                  # model.py:115-155
        feat = self.convnet_nouns(img)
        return self._nouns_from_features(feat, gt_verb, batch_size)

    def _verb_from_features(self, feat, batch_size, seed=_DRAW):
        out = self.ggsnn.run(feat.reshape(batch_size, -1), None, None, 1, True)
        return self._classify(self.verb_classifier, out, seed)

    def predict_verb(self, img, batch_size):                                            # model.py:158-168
        return self._verb_from_features(self.convnet_verbs(img), batch_size)

    def forward(self, img, gt_verb):                                                    # model.py:172-180
        batch_size = img.size(0)
        overlap = self.overlap_backbones if self.overlap_backbones is not None else True
        if (self.share_identical_backbones and self.training and self.convnet_verbs.training and self.convnet_nouns.training
                and img.is_cuda and self.convnet_verbs.weights_equal(self.convnet_nouns)):
            # Both backbones still hold the SAME frozen weights (what the reference's two `pretrained=True` loads give) and
            # train-mode BatchNorm ignores the running statistics: their features are identical, so ONE pass serves the verb
            # path and both noun branches.  The noun backbone's BatchNorm buffers still receive their two updates.
            feat = self.convnet_verbs(img, bn_updates=1, twin=self.convnet_nouns, twin_updates=2)
            pred_verb = self._verb_from_features(feat, batch_size)
        elif overlap and img.is_cuda:
            # The two backbones are independent: the noun backbone runs on a second HIP stream beside the verb path.  Every
            # conv launch is a persistent grid of one workgroup per CU, so the other stream's workgroups move in as a
            # kernel's last round of tiles drains, and its elementwise kernels fill the gaps between launches.
            main = torch.cuda.current_stream()
            side = self._side_streams.get(img.device)
            if side is None:
                side = self._side_streams[img.device] = torch.cuda.Stream(device=img.device)
            # (a CPU `gt_verb`, which the reference's predict_nouns accepts and moves itself, model.py:118-119: the copy is made here,
            #  on the main stream, before the streams fork -- a copy issued on the side stream would not be ordered against main-stream users)
            if not gt_verb.is_cuda:
                gt_verb = gt_verb.to(device=img.device, dtype=torch.int64)
            # Both backbones take the same images: the layout kernel in front of the stem (fp32 NCHW or decoded uint8 -> padded bf16
            # NHWC4) runs once, on the main stream, and its output feeds both passes.
            share = self.convnet_verbs.dtype == self.convnet_nouns.dtype and not (self.convnet_verbs.use_graphs or self.convnet_nouns.use_graphs)
            prepped = self.convnet_verbs.prepare_input(img) if share else None
            side.wait_stream(main)
            # Each pass sizes its persistent grids for HALF of the compute units (`sr_set_cu_share`): the two streams' launches then
            # co-reside on disjoint CUs for their whole duration, instead of two full-chip grids of which the second only moves in as
            # the first one's last round of tiles drains -- 2 x 588 row tiles of a 768-image layer3 launch are 4.6 of 5 rounds on
            # 128 CUs each, not 2 x (2.3 of 3) rounds on 256.
            prev = ops.set_cu_share(self.backbone_cu_share)
            gt_early, pinned = None, None
            try:
                with torch.cuda.stream(side):
                    feat = self.convnet_nouns(img, bn_updates=2, prepped=prepped)
                    feat_ready = side.record_event()
                img.record_stream(side)
                if prepped is not None:
                    prepped[0].record_stream(side)
                feat_v = self.convnet_verbs(img, prepped=prepped)
            finally:
                ops.set_cu_share(prev)
            R = self.encoder.get_max_role_count()
            if self.overlap_gt_branch and not self._use_packed(batch_size, R):
                # Small per-GPU batches: a GGNN GEMM of 768 x 6 rows is 144 tiles on 256 CUs.  The ground-truth-verb noun branch needs
                # only the noun features and gt_verb, so it is queued on the side stream right behind the noun backbone and runs BESIDE
                # the verb path and the predicted-verb branch on the main stream.  (With packed role rows -- large batches, whose
                # launches fill the chip alone -- the branch starts with a host synchronisation and stays on the main stream.)
                for lin in (getattr(self.ggsnn, n) for n in self.ggsnn._ORDER):
                    self.ggsnn._shadow.get(lin.weight, self.dtype)               # storage-dtype copies exist before the streams fork
                self._shadow.get(self.nouns_classifier[1].weight, self.dtype)
                self._shadow.get(self.verb_classifier[1].weight, self.dtype)
                if torch.is_grad_enabled():
                    # the parameters' gradient-accumulation nodes are created by their first use in a graph and belong to the stream
                    # current at that moment: make that the main stream (where their gradients will arrive), not the side stream
                    pinned = [p.view_as(p) for p in self.parameters() if p.requires_grad]
                side.wait_stream(main)
                # (dropout seeds in the reference's call order -- verb, predicted-verb nouns, gt-verb nouns -- whatever the launch order)
                seed_v, seed_p = self._drop_seed(self.verb_classifier[0].p), self._drop_seed(self.nouns_classifier[0].p)
                seed_g = self._drop_seed(self.nouns_classifier[0].p)
                with torch.cuda.stream(side):
                    gt_early = self._nouns_from_features(feat, gt_verb, batch_size, seed_g)
                gt_verb.record_stream(side)
            else:
                seed_v = seed_p = FCGGNN._DRAW
            pred_verb = self._verb_from_features(feat_v, batch_size, seed_v)
            main.wait_event(feat_ready)
            feat.record_stream(main)
            pred_nouns = self._nouns_from_features(feat, torch.argmax(pred_verb, 1), batch_size, seed_p)
            if gt_early is not None:
                main.wait_stream(side)
                gt_early.record_stream(main)
                pinned = None
                return pred_verb, pred_nouns, gt_early
            main.wait_stream(side)
            return pred_verb, pred_nouns, self._nouns_from_features(feat, gt_verb, batch_size)
        else:
            pred_verb = self.predict_verb(img, batch_size)
            feat = self.convnet_nouns(img, bn_updates=2)
        pred_nouns = self._nouns_from_features(feat, torch.argmax(pred_verb, 1), batch_size)
        gt_pred_nouns = self._nouns_from_features(feat, gt_verb, batch_size)
        return pred_verb, pred_nouns, gt_pred_nouns

    def verb_loss(self, pred_verb, gt_verb, denom=None):                                # model.py:183-187
        """`denom` (data parallel, see parallel.global_batch_loss): divide the SUM of this rank's terms by the global batch size
        instead of taking the rank-local mean."""
        if denom is None:
            return nn.functional.cross_entropy(pred_verb.float(), gt_verb)
        return nn.functional.cross_entropy(pred_verb.float(), gt_verb, reduction="sum") / denom

    def nouns_loss(self, pred_nouns, gt_nouns, denoms=None):                            # model.py:190-201
        """`denoms` [3] (data parallel): the global numbers of non-ignored targets per annotator; each term is then this rank's
        SUM over its targets divided by the global count (a rank without valid targets contributes exactly 0)."""
        L = self.encoder.get_num_labels()
        # The reference calls cross_entropy three times on the transposed [B, L, R] view (model.py:196-199): three strided
        # log-softmax passes over the same 295 MB of logits (and three backward passes).  Same value from ONE row-wise
        # log-softmax of the contiguous [B, R, L] logits and one gather of the three annotators' targets:
        #     sum_i  sum_{valid (b, r)} -logp[b, r, t_i[b, r]] / count_i          (count_i = 0 -> nan, as cross_entropy gives)
        logp = nn.functional.log_softmax(pred_nouns.float(), dim=-1)
        t = gt_nouns.permute(0, 2, 1)                                      # [B, R, 3]
        valid = t != L
        picked = logp.gather(-1, t.clamp(max=L - 1))                       # [B, R, 3] (ignored slots: any in-range index, weight 0)
        per = -(picked * valid).sum((0, 1))                                # [3]
        return (per / (valid.sum((0, 1)) if denoms is None else denoms)).sum()
