"""In-kernel time stamps (SR_STAMPS build, SR_GEMM_DEBUG=4) of the three layer3 convs of a bottleneck block."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from situation_recognition_amd import ops, _lib
dt, dev = torch.bfloat16, "cuda"
def report(name, nsteps_tile):
    buf = (ctypes.c_ulonglong * (256 * 8 * 8))()
    _lib.check(_lib.lib().sr_debug_stamps(buf, 256 * 8 * 8), "stamps")
    a = np.frombuffer(buf, dtype=np.uint64).reshape(256, 8, 8).astype(np.float64)
    steps = a[:, :, 5].mean(); seg = a[:, :, :5].mean(axis=(0, 1))
    ex = a[:, :, [2, 6, 7]].mean(axis=(0, 1)) / steps * nsteps_tile
    print("%-26s steps/wg %5.0f | per step: wait %4.0f barrier %4.0f mfma %5.0f | per TILE (%d steps): epilogue %6.0f = prep %5.0f + stats %5.0f + store %5.0f + rest %5.0f" %
          (name, steps, seg[0]/steps, seg[1]/steps, seg[3]/steps, nsteps_tile, seg[4]/steps*nsteps_tile, ex[0], ex[1], ex[2], seg[4]/steps*nsteps_tile - ex.sum()), flush=True)
B, H = int(sys.argv[1]) if len(sys.argv) > 1 else 6144, 14
def t(n, *s): return (torch.randn(*s, device=dev) * n).to(dt)
x256, x1024 = t(1, B, H, H, 256), t(1, B, H, H, 1024)
w3, w1, w33 = t(.05, 1024, 256), t(.05, 256, 1024), t(.05, 256, 9 * 256)
sc, sh = torch.rand(1024, device=dev), torch.rand(1024, device=dev)
ALL = (("1x1 256->1024 fused ER", lambda: ops.conv2d(x256, w3, 1024, 1, 1, 0, bias=sh, escale=sc, res=x1024, relu=True), 8),
                     ("3x3 256->256 S", lambda: ops.conv2d(x256, w33, 256, 3, 1, 1, want_stats=True), 72),
                     ("1x1 1024->256 S", lambda: ops.conv2d(x1024, w1, 256, 1, 1, 0, want_stats=True), 32))

sel = os.environ.get("STAMP_ONLY", "")
for name, fn, ns in ALL:
    if sel and sel not in name:
        continue
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    print("%8.1f us  " % (e0.elapsed_time(e1) * 1e3), end="")
    report(name, ns)
