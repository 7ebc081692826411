"""Debug aid for the two-waves-per-SIMD pair kernel: where do Z / Y differ from the four-wave kernel (same inputs)?"""
import os, sys, subprocess
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from situation_recognition_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
H, C, CR, CX = 14, 256, 256, 1024
dev, dt = "cuda", torch.bfloat16
g = torch.Generator(device=dev).manual_seed(3)
x = torch.randn(B, H, H, C, device=dev, generator=g).to(dt)
res = torch.relu(torch.randn(B, H, H, CX, device=dev, generator=g)).to(dt)
w3 = (torch.randn(CX, C, device=dev, generator=g) * C ** -0.5).to(dt)
w1 = (torch.randn(CR, CX, device=dev, generator=g) * CX ** -0.5).to(dt)
insc, insh = 0.5 + torch.rand(C, device=dev), 0.1 * torch.randn(C, device=dev)
esc, esh = 0.2 + 0.3 * torch.rand(CX, device=dev), 0.1 * torch.randn(CX, device=dev)
M = B * H * H
wp = ops.conv_pair_pack(w3, w1)
z0 = ops.conv2d(x, w3, CX, 1, 1, 0, bias=esh, escale=esc, res=res, relu=True, in_affine=(insc, insh))
y0, st0 = ops.conv2d(z0, w1, CR, 1, 1, 0, want_stats=True)
z1, y1, st1 = ops.conv_pair(x, wp, res, esc, esh, in_affine=(insc, insh))
torch.cuda.synchronize()
TM = 128
Mp = (M // TM) * TM
for name, a, b, W in (("z", z0, z1, CX), ("y", y0, y1, CR)):
    d = (a.view(M, W).view(torch.int16) != b.view(M, W).view(torch.int16))[:Mp].view(-1, TM, W)
    print(name, "differing:", int(d.sum()), "of", d.numel())
    per_row = d.sum((0, 2)); per_col = d.sum((0, 1)); per_tile = d.sum((1, 2))
    print("  rows-in-tile with diffs:", [(i, int(v)) for i, v in enumerate(per_row.tolist()) if v][:40])
    nz = [i for i, v in enumerate(per_col.tolist()) if v]
    print("  columns with diffs: %d, first %s" % (len(nz), nz[:48]))
    tz = [i for i, v in enumerate(per_tile.tolist()) if v]
    print("  tiles with diffs: %d of %d, first %s" % (len(tz), d.shape[0], tz[:40]))
