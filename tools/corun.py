"""Can an MFMA-bound and an HBM-bound kernel, each sized for HALF the compute units (sr_set_cu_share), run side by side faster than one
after the other on the whole chip?  layer3 shapes at batch 6144: the 3x3 256 -> 256 (MFMA) beside the expansion 1x1 256 -> 1024 (HBM)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from situation_recognition_amd import ops
B, H, C = 6144, 14, 256
dt, dev = torch.bfloat16, "cuda"
x4 = torch.randn(B, H, H, 4 * C, device=dev).relu_().to(dt)
x1 = torch.randn(B, H, H, C, device=dev).relu_().to(dt)
x1b = torch.randn(B, H, H, C, device=dev).relu_().to(dt)
w1 = (torch.randn(C, 4 * C, device=dev) * (4 * C) ** -0.5).to(dt)
w2 = (torch.randn(C, 9 * C, device=dev) * (9 * C) ** -0.5).to(dt)
w3 = (torch.randn(4 * C, C, device=dev) * C ** -0.5).to(dt)
sc4, sh4 = 0.5 + torch.rand(4 * C, device=dev), 0.1 * torch.randn(4 * C, device=dev)
c3 = lambda: ops.conv2d(x1, w2, C, 3, 1, 1, want_stats=True)
ex = lambda: ops.conv2d(x1b, w3, 4 * C, 1, 1, 0, bias=sh4, escale=sc4, res=x4, relu=True)
rd = lambda: ops.conv2d(x4, w1, C, 1, 1, 0, want_stats=True)
N = 10
def wall(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
for f in (c3, ex, rd): f()
seq = wall(lambda: [(c3(), ex()) for _ in range(N)])
print("sequential, whole chip: %d x (3x3 + expand)        %.2f ms" % (N, seq))
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def co(fa, fb, share, na=N, nb=N):
    prev = ops.set_cu_share(share)
    for st, f, n in ((s1, fa, na), (s2, fb, nb)):
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            for _ in range(n): f()
    ops.set_cu_share(prev)
    torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2)
for share in (1, 2):
    co(c3, ex, share)
    print("two streams, share %d: 3x3 || expand                 %.2f ms" % (share, wall(lambda: co(c3, ex, share))))
    print("two streams, share %d: 3x3 || 3x3                    %.2f ms (sequential 2N x 3x3: %.2f)" % (share, wall(lambda: co(c3, c3, share)), wall(lambda: [c3() for _ in range(2 * N)])))
    print("two streams, share %d: expand || expand              %.2f ms (sequential 2N x expand: %.2f)" % (share, wall(lambda: co(ex, ex, share)), wall(lambda: [ex() for _ in range(2 * N)])))
    print("two streams, share %d: 3x3 || (expand, reduce)       %.2f ms (sequential: %.2f)" % (share, wall(lambda: co(c3, lambda: (ex(), rd()), share)), wall(lambda: [(c3(), ex(), rd()) for _ in range(N)])))
