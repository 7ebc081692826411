"""Times the stride-1 3x3 of layer3 (256 -> 256 @14) or layer2 (128 -> 128 @28) on the channel-slice direct kernel (c3ds.hip), batch 6144
by default: train-mode form, the same with BatchNorm + ReLU of its input applied on load, and -- for comparison -- what the two forms
replace: SR_NO_C3_256=1 / SR_NO_C3_128S=1 select the generic kernel, which needs the `bn_apply` sweep in front.
usage: python tools/slices_time.py [l3|l2] [batch]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from situation_recognition_amd import ops
layer = sys.argv[1] if len(sys.argv) > 1 else "l3"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 6144
H, C = {"l3": (14, 256), "l2": (28, 128)}[layer]
dev, dt = "cuda", torch.bfloat16
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn(B, H, H, C, device=dev, generator=g).to(dt)
w = (torch.randn(C, 9 * C, device=dev, generator=g) * (9 * C) ** -0.5).to(dt)
sc, sh = 0.5 + torch.rand(C, device=dev), 0.1 * torch.randn(C, device=dev)


def timed(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


fl = 2.0 * B * H * H * C * 9 * C
t_plain = timed(lambda: ops.conv2d(x, w, C, 3, 1, 1, want_stats=True))
out = "3x3 %d->%d @%d batch %d route %s: raw+stats %.1f us (%.0f TFLOP/s)" % (C, C, H, B, ops.conv_route(B, H, H, C, C, 3, 1, 1, want_stats=True), t_plain, fl / t_plain / 1e6)
if ops.conv_in_affine_supported(x, C, 3, 1, 1, res=None, relu=False, want_stats=True):
    t_in = timed(lambda: ops.conv2d(x, w, C, 3, 1, 1, want_stats=True, in_affine=(sc, sh)))
    out += " | BatchNorm on load %.1f us (%.0f TFLOP/s)" % (t_in, fl / t_in / 1e6)
else:
    xb = x.clone()
    t_bn = timed(lambda: ops.bn_apply(xb, sc, sh, relu=True, out=xb))
    out += " | bn_apply sweep in front %.1f us -> %.1f us together" % (t_bn, t_bn + t_plain)
print(out, flush=True)
