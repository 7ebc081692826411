import torch, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from situation_recognition_amd import ops
B = 6144
x = torch.randn(B, 28, 28, 128, device='cuda').relu_().to(torch.bfloat16)
w = (torch.randn(128, 9 * 128, device='cuda') * 0.03).to(torch.bfloat16)
sc, sh = 0.5 + torch.rand(128, device='cuda'), 0.1 * torch.randn(128, device='cuda')
for _ in range(4):
    ops.conv2d(x, w, 128, 3, 1, 1, want_stats=True, in_affine=(sc, sh))
torch.cuda.synchronize()
