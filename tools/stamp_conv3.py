import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from situation_recognition_amd import ops, _lib
dt, dev = torch.bfloat16, "cuda"
def report(name):
    buf = (ctypes.c_ulonglong * (256 * 8 * 8))()
    _lib.check(_lib.lib().sr_debug_stamps(buf, 256 * 8 * 8), "stamps")
    a = np.frombuffer(buf, dtype=np.uint64).reshape(256, 8, 8).astype(np.float64)
    steps = a[:, :, 5].mean(); seg = a[:, :, :5].mean(axis=(0, 1))
    ex = a[:, :, [2, 6, 7]].mean(axis=(0, 1)) / steps * 8
    print("%-28s steps/wg %5.0f | per step: wait %4.0f barrier %4.0f mfma %5.0f | per TILE: epilogue %6.0f = prep %5.0f + stats %5.0f + store %5.0f + rest %5.0f" %
          (name, steps, seg[0]/steps, seg[1]/steps, seg[3]/steps, seg[4]/steps*8, ex[0], ex[1], ex[2], seg[4]/steps*8 - ex.sum()), flush=True)
B, H, Cin, Cout = 6144, 14, 256, 1024
x = torch.randn(B, H, H, Cin, device=dev).to(dt); w = (torch.randn(Cout, Cin, device=dev) * 0.05).to(dt)
idn = torch.randn(B, H, H, Cout, device=dev).to(dt); sc, sh = torch.rand(Cout, device=dev), torch.rand(Cout, device=dev)
for name, fn in (("stats only", lambda: ops.conv2d(x, w, Cout, 1, 1, 0, stats_only=True)),
                 ("conv+stats", lambda: ops.conv2d(x, w, Cout, 1, 1, 0, want_stats=True)),
                 ("fused scale/shift/res/relu", lambda: ops.conv2d(x, w, Cout, 1, 1, 0, bias=sh, escale=sc, res=idn, relu=True))):
    fn(); torch.cuda.synchronize(); fn(); report(name)
