"""Race screen for the pipelined kernels: every launch is repeated and must reproduce its first result BIT FOR BIT (none of
these kernels uses atomics; a read that beats its DMA, or a DMA that beats a read, shows up as run-to-run differences)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from situation_recognition_amd import ops
dt, dev = torch.bfloat16, "cuda"
REP = int(sys.argv[1]) if len(sys.argv) > 1 else 40
torch.manual_seed(0)


def t(*s, scale=1.0):
    return (torch.randn(*s, device=dev) * scale).to(dt)


def flat(r):
    r = r if isinstance(r, (tuple, list)) else (r,)
    return [x.clone() for x in r if torch.is_tensor(x)]


def gm(part, C):
    """the entries of Gram partials the kernels WRITE (block-lower triangle for C <= 256: the rest of the buffer is never touched)"""
    valid = torch.cat([ops.gram_valid_mask(C).reshape(-1), torch.ones(C, dtype=torch.bool)]).to(part.device)
    return part[:, valid]


def screen(name, fn):
    ref = flat(fn())
    bad = 0
    for _ in range(REP):
        out = flat(fn())
        bad += int(any(not torch.equal(a, b) for a, b in zip(ref, out)))
    torch.cuda.synchronize()
    print("%-44s %3d / %d runs differ" % (name, bad, REP), flush=True)
    return bad


bad = 0
for B in (96, 768):
    x256, x1024 = t(B, 14, 14, 256), t(B, 14, 14, 1024)
    w33, w3, w1 = t(256, 9 * 256, scale=.03), t(1024, 256, scale=.06), t(256, 1024, scale=.03)
    sc, sh = torch.rand(1024, device=dev), torch.rand(1024, device=dev)
    bad += screen("B=%d 3x3 256->256 @14 + stats (direct kernel, c3ds.hip)" % B, lambda: ops.conv2d(x256, w33, 256, 3, 1, 1, want_stats=True))
    x512, w512 = t(4 * B, 7, 7, 512), t(512, 9 * 512, scale=.02)
    bad += screen("B=%d 3x3 512->512 @7 + stats (generic 256x256 ping-pong kernel)" % (4 * B), lambda: ops.conv2d(x512, w512, 512, 3, 1, 1, want_stats=True))
    bad += screen("B=%d 1x1 256->1024 scale/shift/res/relu" % B, lambda: ops.conv2d(x256, w3, 1024, 1, 1, 0, bias=sh, escale=sc, res=x1024, relu=True))
    bad += screen("B=%d 1x1 1024->256 + stats" % B, lambda: ops.conv2d(x1024, w1, 256, 1, 1, 0, want_stats=True))
    wp3 = ops.conv_pair_pack(w3, w1)
    bad += screen("B=%d fused pair 256->1024->256 @14 (pair.hip)" % B, lambda: ops.conv_pair(x256, wp3, x1024, sc, sh, in_affine=(sc[:256].contiguous(), sh[:256].contiguous())))
    bad += screen("B=%d gram 256" % B, lambda: gm(ops.gram(x256.view(-1, 256)), 256))
    s2, h2 = torch.rand(256, device=dev) + .5, torch.randn(256, device=dev) * .1
    def fused():
        y = x256.view(-1, 256).clone()
        return y, gm(ops.bn_apply_gram(y, s2, h2), 256)
    bad += screen("B=%d bn_apply+gram 256" % B, fused)
    bad += screen("B=%d bn_gram 256 (no write-back)" % B, lambda: gm(ops.bn_gram(x256.view(-1, 256), s2, h2), 256))
    x64 = t(B, 56, 56, 64); w64 = t(64, 9 * 64, scale=.05)
    bad += screen("B=%d 3x3 64->64 (direct kernel, c3d.hip)" % B, lambda: ops.conv2d(x64, w64, 64, 3, 1, 1, want_stats=True))
    bad += screen("B=%d 3x3 64->64 direct, BN on load" % B, lambda: ops.conv2d(x64, w64, 64, 3, 1, 1, want_stats=True, in_affine=(s2[:64].contiguous(), h2[:64].contiguous())))
    x128 = t(B, 28, 28, 128); w128 = t(128, 9 * 128, scale=.04)
    bad += screen("B=%d 3x3 128->128 @28 (direct kernel, c3ds.hip)" % B, lambda: ops.conv2d(x128, w128, 128, 3, 1, 1, want_stats=True))
    bad += screen("B=%d 3x3 128->128 @28 direct, BN on load" % B, lambda: ops.conv2d(x128, w128, 128, 3, 1, 1, want_stats=True, in_affine=(s2[:128].contiguous(), h2[:128].contiguous())))
    bad += screen("B=%d 3x3 256->256 @14 direct, BN on load (c3ds.hip)" % B, lambda: ops.conv2d(x256, w33, 256, 3, 1, 1, want_stats=True, in_affine=(s2, h2)))
    x512r, wp2 = t(B, 28, 28, 512), ops.conv_pair_pack(t(512, 128, scale=.08), t(128, 512, scale=.04))
    bad += screen("B=%d fused pair 128->512->128 @28 (pair.hip)" % B, lambda: ops.conv_pair(x128, wp2, x512r, sc[:512].contiguous(), sh[:512].contiguous(), in_affine=(s2[:128].contiguous(), h2[:128].contiguous())))
    x256r, wp1 = t(B, 56, 56, 256), ops.conv_pair_pack(t(256, 64, scale=.1), t(64, 256, scale=.06))
    bad += screen("B=%d fused pair 64->256->64 @56 (pair.hip)" % B, lambda: ops.conv_pair(x64, wp1, x256r, sc[:256].contiguous(), sh[:256].contiguous(), in_affine=(s2[:64].contiguous(), h2[:64].contiguous())))
    del x512r, x256r
    x128b = t(B, 24, 24, 128)
    bad += screen("B=%d 3x3 128->128 @24 (256x128 tiles)" % B, lambda: ops.conv2d(x128b, w128, 128, 3, 1, 1, want_stats=True))
    bad += screen("B=%d gram 128 / bn_gram 128" % B, lambda: (gm(ops.gram(x128.view(-1, 128)), 128), gm(ops.bn_gram(x128.view(-1, 128), s2[:128].contiguous(), h2[:128].contiguous()), 128)))
    bad += screen("B=%d gram 64 / bn_gram 64" % B, lambda: (gm(ops.gram(x64.view(-1, 64)), 64), gm(ops.bn_gram(x64.view(-1, 64), s2[:64].contiguous(), h2[:64].contiguous()), 64)))
M = 36864
n, h, W = t(M, 2048), t(M, 2048), t(2048, 2048, scale=.02)
U = t(2048, 2048, scale=.02); b1, b2 = torch.randn(2048, device=dev), torch.randn(2048, device=dev)
bad += screen("GRU gate GEMM 2 pairs sigmoid", lambda: ops.gemm([(n, W), (h, U)], bias=b1, bias2=b2, act=ops.ACT_SIGMOID))
bad += screen("GRU gate GEMM sigmoid*h", lambda: ops.gemm([(n, W), (h, U)], bias=b1, bias2=b2, act=ops.ACT_SIGMOID_MUL, aux1=h))
ns, hs, zs = n[:805], h[:805], torch.rand(805, 2048, device=dev).to(dt)      # 256 x 128 tiles (launches covering at most half the CUs)
bad += screen("GRU gate GEMM sigmoid*h, 805 rows (256x128 tiles)", lambda: ops.gemm([(ns, W), (hs, U)], bias=b1, bias2=b2, act=ops.ACT_SIGMOID_MUL, aux1=hs))
bad += screen("GRU gate GEMM tanh & blend, 805 rows (256x128 tiles)", lambda: ops.gemm([(ns, W), (hs, U)], bias=b1, bias2=b2, act=ops.ACT_TANH_BLEND, aux1=hs, aux2=zs))
out = torch.zeros(2048, 2048, device=dev)
bad += screen("TN GEMM 2048x2048 K=36864", lambda: ops.gemm_tn(n, h, out, accumulate=False))
print("TOTAL differing runs:", bad)
sys.exit(1 if bad else 0)
