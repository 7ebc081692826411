import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
from situation_recognition_amd import ops
from bench_layers import timeit, line
B = 6144
img = torch.randn(B, 3, 224, 224, device='cuda')
w = (torch.randn(64, 256, device='cuda') * 0.1).to(torch.bfloat16)
sc, sh = 0.5 + torch.rand(64, device='cuda'), 0.1 * torch.randn(64, device='cuda')
xp = ops.stem_prep(img, torch.bfloat16)
M = B * 112 * 112
line("stem stats-only launch", timeit(lambda: ops.conv2d(xp, w, 64, 7, 2, 3, stats_only=True, stem_hw=(224, 224))), 2.0 * M * 64 * 147, xp.numel() * 2)
line("stem + BN + ReLU + maxpool fused", timeit(lambda: ops.stem_bn_relu_maxpool(xp, w, sc, sh, (224, 224))), 2.0 * M * 64 * 147, xp.numel() * 2 + M // 4 * 128)
