"""Per-shape time table of one train-mode backbone pass: every wrapper in ops is bracketed by HIP events and keyed by
its shape/flags; prints count, total ms, average us, algorithmic TFLOP/s and GB/s, and the time both rooflines allow."""
import collections
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from situation_recognition_amd import ops
from situation_recognition_amd.model import resnet

B = int(sys.argv[1]) if len(sys.argv) > 1 else 6144
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 152
REC = []


def wrap(name, keyfn):
    orig = getattr(ops, name)

    def f(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = orig(*a, **k)
        e1.record()
        REC.append((keyfn(*a, **k), e0, e1))
        return r
    setattr(ops, name, f)


def conv_key(x, w, Cout, KH, stride, pad, bias=None, res=None, relu=False, want_stats=False, stem_hw=None, escale=None,
             stats_only=False, out=None, in_affine=None):
    Bn = x.shape[0]
    if stem_hw is not None:
        H, W, Cin = stem_hw[0], stem_hw[1], 3
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        kk = 49
    else:
        H, W, Cin = x.shape[1:]
        Ho, Wo = (H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KH) // stride + 1
        kk = KH * KH
    M = Bn * Ho * Wo
    fl = 2.0 * M * Cout * kk * Cin
    by = 2.0 * (x.numel() + w.numel()) + (0 if stats_only else 2.0 * M * Cout) + (2.0 * M * Cout if res is not None else 0)
    flags = (("S" if want_stats else "") + ("O" if stats_only else "") + ("E" if escale is not None else "") + ("R" if res is not None else "")
             + ("A" if in_affine is not None else ""))
    return ("conv%dx%d/%d %4d->%4d @%3d %s" % (KH, KH, stride, Cin, Cout, Ho, flags), fl, by)


def apply_key(x, scale, shift, res=None, relu=True, out=None):
    return ("bn_apply C=%4d @%3d %s" % (x.shape[-1], x.shape[1], "R" if res is not None else ""), 0.0,
            2.0 * x.numel() * (3 if res is not None else 2))


wrap("conv2d", conv_key)
wrap("bn_apply", apply_key)
wrap("conv_pair", lambda x, wp, res, esc, esh, in_affine=None: (
    "pair 1x1 %4d->%4d->%4d @%3d %s" % (x.shape[3], res.shape[3], wp.numel() // res.shape[3] - x.shape[3], x.shape[1], "ERAS" if in_affine is not None else "ERS"),
    2.0 * x.shape[0] * x.shape[1] * x.shape[2] * wp.numel(), 2.0 * (x.numel() + 2 * res.numel() + x.numel() // x.shape[3] * (wp.numel() // res.shape[3] - x.shape[3]))))
wrap("gram", lambda x: ("gram C=%4d M=%d" % (x.shape[1], x.shape[0]), 2.0 * x.shape[0] * x.shape[1] ** 2, 2.0 * x.numel()))
wrap("bn_apply_gram", lambda x, sc, sh: ("bn_apply+gram C=%4d M=%d" % (x.shape[1], x.shape[0]), 2.0 * x.shape[0] * x.shape[1] ** 2, 4.0 * x.numel()))
wrap("bn_gram", lambda x, sc, sh: ("bn+gram (no write) C=%4d M=%d" % (x.shape[1], x.shape[0]), 2.0 * x.shape[0] * x.shape[1] ** 2, 2.0 * x.numel()))
wrap("bn_finalize_gram", lambda part, w, *a, **k: ("bn_finalize_gram C=%4d N=%4d" % (w.shape[1], w.shape[0]), 0.0, 4.0 * part.numel()))
wrap("bn_finalize", lambda st, *a, **k: ("bn_finalize C=%4d tiles=%d" % (st.shape[2], st.shape[0]), 0.0, 4.0 * st.numel()))
wrap("maxpool3x3s2", lambda x, *a: ("maxpool", 0.0, 2.0 * x.numel() * 1.25))
wrap("avgpool", lambda x: ("avgpool", 0.0, 2.0 * x.numel()))

net = resnet(None, depth=depth).cuda().train()
img = torch.randn(B, 3, 224, 224, device="cuda")
net(img)
torch.cuda.synchronize()
REC.clear()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
net(img)
e1.record()
torch.cuda.synchronize()
agg = collections.OrderedDict()
for (k, fl, by), a, b in REC:
    t = agg.setdefault(k, [0, 0.0, fl, by])
    t[0] += 1
    t[1] += a.elapsed_time(b)
tot = sum(v[1] for v in agg.values())
print("pass %.1f ms, bracketed %.1f ms" % (e0.elapsed_time(e1), tot))
print("%-42s %4s %9s %9s %8s %8s %9s %9s" % ("op", "n", "total ms", "avg us", "TFLOP/s", "GB/s", "mfma us", "hbm us"))
for k, (n, ms, fl, by) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    us = ms / n * 1e3
    print("%-42s %4d %9.2f %9.1f %8.1f %8.0f %9.1f %9.1f" % (k, n, ms, us, fl / us / 1e6, by / us / 1e3, fl / 2.5e15 * 1e6, by / 8e12 * 1e6))
