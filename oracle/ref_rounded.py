"""The oracle's eval-mode forward with bf16 ROUNDING AT THE HIP PATH'S STORAGE POINTS (fp32 arithmetic everywhere else).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Why it exists.  The pure-fp32 oracle (oracle/ref_model.py, pinned to the reference by the golden vectors) and a bf16-storage
implementation of a randomly initialised 152-layer net differ END TO END by what rounding every stored activation to 8 significant
bits does to such a net (0.2 relative L2 on the pooled features): no end-to-end tolerance against the fp32 oracle can tell that
from a wrong layer.  This module restates the SAME graph (reference model.py:33-35 backbone call, 59-86 GGSNN, 115-168 heads) in
fp32 and rounds to bf16 exactly where the HIP path writes a bf16 tensor, so that what remains between the two is fp32 summation
order plus the values that land on the other side of a rounding boundary because of it.  With `rnd=identity` every function here
must reproduce the pinned fp32 oracle (tests/test_oracle_rounded.py, CPU).

What it can and cannot buy (measured, round 4: DESIGN.md section 2).  Boundary flips do NOT stay rare: a flipped value changes every
downstream accumulator a little, which flips more values, and after a few dozen layers two such evaluations make essentially
independent rounding decisions.  Two runs of THIS module that differ only in the convolutions' summation precision
(`conv_precision(f64=True)`) end 0.14 apart (relative L2, pooled features) on the randomly initialised ResNet-152 of the config-3
test, against 0.19 for either of them from the pure fp32 oracle -- that net amplifies any perturbation to O(1) by layer4, whatever
is compared with whatever.  On a net whose residual branches are damped (bn3 gamma x 0.2: the regime of a trained ResNet) the
figures are 0.014 / 0.023, and one BatchNorm bias off by 0.5 in one of 155 layers moves the features by 0.087.  So the composed pass
is gated on the damped net, with a tolerance derived in the test from the fp32-vs-fp64 floor measured on the spot.

Storage points of the HIP eval path (situation_recognition_amd/model.py `resnet._unit(train=False)`, `_GGNNFunction.forward`,
`_ClassifierFunction.forward`; csrc/*.hip epilogues):
  image            fp32 NCHW -> bf16 NHWC4 (`sr_stem_prep`)
  conv weights     w * gamma / sqrt(running_var + eps) in fp32, then bf16 (`_ConvBN.folded`); the shift stays an fp32 bias
  every conv       bf16 x bf16 products, fp32 accumulation, + bias (+ bf16 identity), ReLU, ONE rounding to bf16 at the store
                   (stem: before the max-pool, whose maximum of bf16 values is exact).  Exception: a residual convolution on the
                   GENERIC implicit-GEMM kernel (csrc/gemm.hip EPIX 6: every shape the weight-stationary kernel of csrc/expand.hip
                   does not serve -- in ResNet-152 layer4's 512 -> 2048 expansion) stages acc + bias as bf16 in LDS, and adds the
                   identity + ReLU to that staged value: TWO roundings (`hip_staged_residual`)
  pooled features  fp32 mean of bf16 values -> bf16
  GGNN             weights bf16, biases fp32; agg, n, z, r*h, h' stored bf16; the epilogues use the UNROUNDED sigmoid / tanh of
                   their own accumulator (r in r*h, c in the blend) and the STORED bf16 z, h (csrc/gemm.hip EPI 4 / 5)
  node init        relu((feat * role_emb) * verb_emb), fp32 embeddings, bf16 store (csrc/ggnn.hip node_init_fwd_kernel)
  classifiers      bf16 activations x bf16 weights, fp32 accumulation + fp32 bias, fp32 logits
"""
import torch
import torch.nn.functional as F


def bf16(t):
    return t.to(torch.bfloat16).float()


def identity(t):
    return t


def hip_staged_residual(conv):
    """Does the HIP path run this residual convolution on the generic kernel (bf16-staged accumulator, see the module docstring)?
    Mirrors srx_conv1x1_expand's shape test (csrc/expand.hip) for launches of at least 32 768 output pixels."""
    ws = (conv.kernel_size == (1, 1) and conv.stride == (1, 1) and conv.out_channels % 256 == 0 and conv.out_channels <= 1024
          and conv.in_channels in (64, 128, 256))
    return not ws


def _fold(conv, bn, rnd):
    scale = bn.weight.float() * torch.rsqrt(bn.running_var.float() + bn.eps)
    shift = bn.bias.float() - bn.running_mean.float() * scale
    return rnd(conv.weight.float() * scale.view(-1, 1, 1, 1)), shift


_CONV_F64 = False


class conv_precision:
    """`with conv_precision(f64=True):` -- the convolutions of this module accumulate in fp64 (inputs, folded weights and rounding
    points unchanged).  Two evaluations that differ ONLY in this are two equally correct bf16-storage implementations with
    different summation arithmetic: their distance is the floor under any comparison of such an implementation with another one
    (tests/test_full_configs_gpu.py derives its gate from it)."""

    def __init__(self, f64):
        self.f64 = f64

    def __enter__(self):
        global _CONV_F64
        self.prev, _CONV_F64 = _CONV_F64, self.f64

    def __exit__(self, *a):
        global _CONV_F64
        _CONV_F64 = self.prev


def _conv(x, conv, bn, rnd):
    w, b = _fold(conv, bn, rnd)
    if _CONV_F64:
        return F.conv2d(x.double(), w.double(), b.double(), stride=conv.stride, padding=conv.padding).float()
    return F.conv2d(x, w, b, stride=conv.stride, padding=conv.padding)


def resnet_eval_features(net, img, rnd=bf16, taps=None, staged=hip_staged_residual):
    """Eval-mode pooled features of an oracle.ref_resnet.RefResNet (reference model.py:35 -> torchvision forward).
    `taps`: optional list that receives (name, tensor) of the stem output and every block output (for localising a failure)."""
    with torch.no_grad():
        x = rnd(img.float())
        x = rnd(F.relu(_conv(x, net.conv1, net.bn1, rnd)))
        x = F.max_pool2d(x, 3, stride=2, padding=1)
        if taps is not None:
            taps.append(("stem", x))
        for s in range(4):
            for bi, blk in enumerate(getattr(net, "layer%d" % (s + 1))):
                idn = x if blk.downsample is None else rnd(_conv(x, blk.downsample[0], blk.downsample[1], rnd))
                y = rnd(F.relu(_conv(x, blk.conv1, blk.bn1, rnd)))
                if hasattr(blk, "conv3"):                                   # Bottleneck
                    y = rnd(F.relu(_conv(y, blk.conv2, blk.bn2, rnd)))
                    last = blk.conv3
                    y = _conv(y, blk.conv3, blk.bn3, rnd)
                else:                                                       # BasicBlock
                    last = blk.conv2
                    y = _conv(y, blk.conv2, blk.bn2, rnd)
                if staged(last):
                    y = rnd(y)
                x = rnd(F.relu(y + idn))
                if taps is not None:
                    taps.append(("layer%d.%d" % (s + 1, bi), x))
        return rnd(x.mean((2, 3)))


def ggsnn(g, h, mask=None, verb=False, rnd=bf16):
    """oracle.ref_model.RefGGSNN.forward (reference model.py:59-86) in the algebraic neighbour form the HIP path computes
    (W_p(A h) + R b_p; its equality with the reference's mask-expand form is tested in tests/test_oracle_golden.py)."""
    with torch.no_grad():
        W = {n: rnd(getattr(g, n).weight.float()) for n in g.NAMES}
        b = {n: getattr(g, n).bias.float() for n in g.NAMES}
        h = rnd(h)
        for _ in range(g.steps):
            if verb:
                agg, scale = h, 1.0
            else:
                B, R = mask.shape[0], mask.shape[1]
                agg, scale = rnd(torch.bmm(mask.float(), h.reshape(B, R, -1)).reshape(B * R, -1)), float(R)
            n = rnd(agg @ W["W_p"].t() + scale * b["W_p"])
            z = rnd(torch.sigmoid(n @ W["W_z"].t() + h @ W["U_z"].t() + b["W_z"] + b["U_z"]))
            r_full = torch.sigmoid(n @ W["W_r"].t() + h @ W["U_r"].t() + b["W_r"] + b["U_r"])
            rh = rnd(r_full * h)
            c_full = torch.tanh(n @ W["W_h"].t() + rh @ W["U_h"].t() + b["W_h"] + b["U_h"])
            h = rnd((1.0 - z) * h + z * c_full)
        return h


def fcggnn_eval(model, img, gt_verb, rnd=bf16):
    """Eval-mode (pred_verb, gt_pred_nouns, verb features, noun features) of an oracle.ref_model.RefFCGGNN (reference model.py:158-168
    and 115-155 with the ground-truth verb); dropout is the identity in eval mode.  The predicted-verb noun branch is left out: it
    is the same function fed argmax(pred_verb), and near-ties of an untrained head's argmax are not a property of either side."""
    with torch.no_grad():
        B = img.shape[0]
        fv = resnet_eval_features(model.convnet_verbs.model, img, rnd)
        fn = resnet_eval_features(model.convnet_nouns.model, img, rnd)
        out = ggsnn(model.ggsnn, fv.reshape(B, -1), None, True, rnd)       # (relu of a mean of post-ReLU values: the identity)
        lin = model.verb_classifier[1]
        pred_verb = out @ rnd(lin.weight.float()).t() + lin.bias.float()
        R = model.encoder.get_max_role_count()
        role_idx = model.encoder.get_role_ids_batch(gt_verb)
        ve = model.verb_emb.weight.float()[gt_verb]
        ro = model.role_emb.weight.float()[role_idx]
        node = rnd(F.relu((fn[:, None, :] * ro) * ve[:, None, :])).reshape(B * R, -1)
        mask = model.encoder.get_adj_matrix_noself(gt_verb)
        out = ggsnn(model.ggsnn, node, mask, False, rnd)
        lin = model.nouns_classifier[1]
        logits = out @ rnd(lin.weight.float()).t() + lin.bias.float()
        return pred_verb, logits.reshape(B, R, -1), fv, fn
