"""Times the GGNN's GEMM launches at the row counts of one GPU's share: the fused gate epilogues (z, r & r*h, candidate & blend: two
operand pairs, D = 2048) and the linear ones (W_p, backward data gradients).  SR_GEMM_GATE_NARROW=0 keeps the gate launches on 256x256
tiles whatever the tile count (the A/B of the 256x128 instantiations).    usage: python tools/gate_gemm_time.py [rows ...]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from situation_recognition_amd import ops
D, dev, dt = 2048, "cuda", torch.bfloat16
rows = [int(a) for a in sys.argv[1:]] or [768, 4608, 9216, 36864]


def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


g = torch.Generator(device=dev).manual_seed(3)
W = [(torch.randn(D, D, device=dev, generator=g) * D ** -0.5).to(dt) for _ in range(2)]
b = torch.zeros(D, device=dev)
for M in rows:
    n, h = [torch.randn(M, D, device=dev, generator=g).to(dt) for _ in range(2)]
    z = torch.rand(M, D, device=dev, generator=g).to(dt)
    out, out2 = torch.empty_like(n), torch.empty_like(n)
    res = []
    res.append(("sigmoid", timed(lambda: ops.gemm([(n, W[0]), (h, W[1])], bias=b, bias2=b, act=ops.ACT_SIGMOID, out=out)), 2))
    res.append(("sigmoid & r*h", timed(lambda: ops.gemm([(n, W[0]), (h, W[1])], bias=b, bias2=b, act=ops.ACT_SIGMOID_MUL, aux1=h, out=out, out2=out2)), 2))
    res.append(("tanh & blend", timed(lambda: ops.gemm([(n, W[0]), (h, W[1])], bias=b, bias2=b, act=ops.ACT_TANH_BLEND, aux1=z, aux2=h, out=out, out2=out2)), 2))
    res.append(("linear", timed(lambda: ops.gemm([(n, W[0])], bias=b, out=out)), 1))
    print("rows %6d (gate tiles cfg %d, linear cfg %d): " % (M, ops.lib().sr_gemm_tile_cfg(M, D, 0, 1), ops.lib().sr_gemm_tile_cfg(M, D, 1, 1))
          + "  ".join("%s %.1f us (%.0f TF/s)" % (k, t, 2.0 * M * D * D * p / t / 1e6) for k, t, p in res), flush=True)
