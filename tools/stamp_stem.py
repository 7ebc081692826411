"""In-kernel stamps (SR_STAMPS build, SR_GEMM_DEBUG=4) of the narrow-tile convs: stem 7x7 and layer1."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from situation_recognition_amd import ops, _lib
dt, dev = torch.bfloat16, "cuda"
def report(name, nsteps_tile):
    buf = (ctypes.c_ulonglong * (256 * 8 * 8))()
    _lib.check(_lib.lib().sr_debug_stamps(buf, 256 * 8 * 8), "stamps")
    a = np.frombuffer(buf, dtype=np.uint64).reshape(256, 8, 8).astype(np.float64)
    a = a[:, :4]                                  # 4 waves per workgroup
    steps = a[:, :, 5].mean(); seg = a[:, :, :5].mean(axis=(0, 1))
    ex = a[:, :, [2, 6, 7]].mean(axis=(0, 1)) / steps * nsteps_tile
    print("%-22s steps/wg %6.0f | per step: wait %4.0f barrier %4.0f mfma %5.0f | per TILE (%d steps): epilogue %6.0f = prep %5.0f + stats %5.0f + store %5.0f + rest %5.0f" %
          (name, steps, seg[0]/steps, seg[1]/steps, seg[3]/steps, nsteps_tile, seg[4]/steps*nsteps_tile, ex[0], ex[1], ex[2], seg[4]/steps*nsteps_tile - ex.sum()), flush=True)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
img = torch.randn(B, 3, 224, 224, device=dev)
xp = ops.stem_prep(img, dt)
ws = (torch.randn(64, 256, device=dev) * .05).to(dt)
x64 = torch.randn(B, 56, 56, 64, device=dev).to(dt)
w33 = (torch.randn(64, 9 * 64, device=dev) * .05).to(dt); w11 = (torch.randn(64, 64, device=dev) * .05).to(dt)
for name, fn, ns in (("stem 7x7/2 3->64", lambda: ops.conv2d(xp, ws, 64, 7, 2, 3, want_stats=True, stem_hw=(224, 224)), 8),
                     ("3x3 64->64 @56", lambda: ops.conv2d(x64, w33, 64, 3, 1, 1, want_stats=True), 18),
                     ("1x1 64->64 @56", lambda: ops.conv2d(x64, w11, 64, 1, 1, 0, want_stats=True), 2)):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    print("%8.1f us  " % (e0.elapsed_time(e1) * 1e3), end=""); report(name, ns)
