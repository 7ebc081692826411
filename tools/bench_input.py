"""Host->device input rate of the uint8 shard pipeline (imsitu_shards.ShardLoader) on synthetic shards: memory-mapped gather
on a background thread, pinned uint8 copy over PCIe, crop + flip gather on the GPU.  No model."""
import json, os, sys, tempfile, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from situation_recognition_amd import imsitu_shards as sh
from situation_recognition_amd.imsitu_encoder import imsitu_encoder

N, B = int(sys.argv[1]) if len(sys.argv) > 1 else 12288, int(sys.argv[2]) if len(sys.argv) > 2 else 6144
ann1 = json.load(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "overfitting.json")))
one = next(iter(ann1.values()))
names = ["img_%06d.jpg" % i for i in range(N)]
ann = {n: one for n in names}
enc = imsitu_encoder(ann1, quiet=True)
d = tempfile.mkdtemp(prefix="shards_", dir=os.environ.get("TMPDIR", "/tmp"))
rng = np.random.default_rng(0)
per = 2048
sizes = []
for s0 in range(0, N, per):
    n = min(per, N - s0)
    np.save(os.path.join(d, "shard_%05d.npy" % len(sizes)), rng.integers(0, 256, (n, sh.CANVAS, sh.CANVAS, 3), dtype=np.uint8))
    sizes.append(n)
rects = np.tile(np.array([[32, 0, 224, 288, 32, 32]], dtype=np.int32), (N, 1))
np.save(os.path.join(d, "rects.npy"), rects)
json.dump({"names": names, "sizes": sizes, "canvas": sh.CANVAS, "crop": sh.CROP}, open(os.path.join(d, "index.json"), "w"))
dl = sh.ShardLoader(d, ann, enc, B, "cuda", train=True)
for ep in range(2):
    t0 = time.perf_counter(); n = 0
    for _, img, verb, labels in dl:
        n += img.shape[0]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("epoch %d: %d images in %.2f s = %.0f images/s (%.2f GB/s of uint8 canvases over PCIe)" %
          (ep, n, dt, n / dt, n * sh.CANVAS * sh.CANVAS * 3 / dt / 1e9), flush=True)
