"""Race screen of the expansion-conv kernels (csrc/expand.hip): repeated launches must be bit-identical and equal to the
generic kernel's result (SR_NO_EXPAND=1 computes the comparison in a child-free way: the generic kernel is selected by Cin=512)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from situation_recognition_amd import ops
dt, dev = torch.bfloat16, "cuda"
REP = int(sys.argv[1]) if len(sys.argv) > 1 else 30
torch.manual_seed(0)
bad = 0
for (B, H, C) in ((1024, 14, 256), (6144, 14, 256), (768, 28, 128), (256, 56, 64)):
    M = B * H * H
    x = (torch.randn(B, H, H, C, device=dev)).relu_().to(dt)
    w = (torch.randn(4 * C, C, device=dev) * C ** -0.5).to(dt)
    res = torch.randn(B, H, H, 4 * C, device=dev).to(dt)
    sc, sh = 0.5 + torch.rand(4 * C, device=dev), 0.1 * torch.randn(4 * C, device=dev)
    ref = torch.relu((x.float().view(M, C) @ w.float().t()) * sc + sh + res.float().view(M, 4 * C))
    first = None
    nd, nbad = 0, 0
    for r in range(REP):
        if r % 3 == 1:                       # vary what runs in front of the launch
            ops.gram(x.view(M, C))
        elif r % 3 == 2:
            torch.empty(64 << 20, device=dev).fill_(1.0)
        y = ops.conv2d(x, w, 4 * C, 1, 1, 0, bias=sh, escale=sc, res=res, relu=True)
        err = float((y.float().view(M, 4 * C) - ref).abs().max())
        if first is None:
            first = y.clone()
        nd += int(not torch.equal(first, y))
        nbad += int(err > 1.2e-2 * float(ref.abs().max()))
    print("B=%d @%d C=%d: %d / %d runs differ from the first, %d outside tolerance" % (B, H, C, nd, REP, nbad), flush=True)
    bad += nd + nbad
print("TOTAL", bad)
sys.exit(1 if bad else 0)
