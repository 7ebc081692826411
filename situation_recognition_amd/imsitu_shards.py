"""Pre-decoded uint8 image shards: the input pipeline that keeps eight MI355X fed.

The reference decodes a JPEG, resizes and augments it on a DataLoader worker for every sample of every epoch
(utils/imsitu_loader.py:13-20, utils/imsitu_encoder.py:21-36); at ~10 000 images/s per GPU that is the bottleneck long before
the model is.  Here the decode + Resize(224) is done ONCE (`write_shards`): every image is stored as uint8 on an S x S canvas
(S = 288) with the rectangle it occupies; at train time a batch is a gather from memory-mapped shards, one host->device copy of
uint8 pixels (a quarter of the fp32 bytes), and the random crop + flip on the device (`gpu_augment`); ToTensor + Normalize are
fused into the stem's layout kernel (`sr_image_prep_u8`, the model's uint8 input path).

Same augmentation as the reference: Resize(shorter side -> 224), RandomCrop(224) over the resized image, horizontal flip with
p = 0.5 (train); CenterCrop(224) (dev).  The only deviation: a longer side above S pixels (aspect ratio > 1.29) is clipped
to its central S pixels, so the random crop of a very elongated image never reaches its outer margins.
"""
import json
import os

import numpy as np
import torch

CANVAS = 288
CROP = 224


def _resized_rgb(path):
    from PIL import Image
    from .imsitu_encoder import _resize_shorter
    with Image.open(path) as im:
        return np.asarray(_resize_shorter(im.convert('RGB'), CROP), dtype=np.uint8)


def write_shards(img_dir, names, out_dir, per_shard=2048, quiet=True):
    """Decode + Resize(224) every image once; writes out_dir/shard_%05d.npy ([n, S, S, 3] uint8), rects.npy ([N, 6] int32:
    y0, x0, h, w of the image on its canvas and cy, cx = canvas position of the reference's CenterCrop(224) origin) and
    index.json (names in order, shard sizes)."""
    os.makedirs(out_dir, exist_ok=True)
    names = list(names)
    rects = np.zeros((len(names), 6), dtype=np.int32)
    sizes = []
    for s0 in range(0, len(names), per_shard):
        chunk = names[s0:s0 + per_shard]
        buf = np.zeros((len(chunk), CANVAS, CANVAS, 3), dtype=np.uint8)
        for i, n in enumerate(chunk):
            a = _resized_rgb(os.path.join(img_dir, n))
            h, w = a.shape[:2]
            ch, cw = min(h, CANVAS), min(w, CANVAS)                  # clip the longer side to its central S pixels
            ty, tx = (h - ch) // 2, (w - cw) // 2
            a = a[ty:ty + ch, tx:tx + cw]
            y0, x0 = (CANVAS - ch) // 2, (CANVAS - cw) // 2
            buf[i, y0:y0 + ch, x0:x0 + cw] = a
            # CenterCrop origin of the UNclipped resized image (imsitu_encoder.dev_transform), in canvas coordinates
            rects[s0 + i] = (y0, x0, ch, cw, y0 + int(round((h - CROP) / 2.0)) - ty, x0 + int(round((w - CROP) / 2.0)) - tx)
        np.save(os.path.join(out_dir, "shard_%05d.npy" % len(sizes)), buf)
        sizes.append(len(chunk))
        if not quiet:
            print("shard %d: %d images" % (len(sizes) - 1, len(chunk)), flush=True)
    np.save(os.path.join(out_dir, "rects.npy"), rects)
    with open(os.path.join(out_dir, "index.json"), "w") as f:
        json.dump({"names": names, "sizes": sizes, "canvas": CANVAS, "crop": CROP}, f)
    return len(sizes)


def gpu_augment(canvas_u8, rects, train, generator=None):
    """canvas_u8 [B,S,S,3] uint8 and rects [B,6] (y0,x0,h,w,cy,cx) on the device -> uint8 [B,224,224,3]: RandomCrop(224) inside
    each image's rectangle + horizontal flip with p=0.5 (train) or the reference's CenterCrop(224) (dev), as ONE gather."""
    B, dev = canvas_u8.shape[0], canvas_u8.device
    y0, x0, h, w, cy, cx = (rects[:, i].long() for i in range(6))
    if train:
        u = torch.rand((B, 3), device=dev, generator=generator)
        oy = (u[:, 0] * (h - CROP + 1).float()).long().clamp_(max=CANVAS)          # uniform over 0 .. h-224
        ox = (u[:, 1] * (w - CROP + 1).float()).long().clamp_(max=CANVAS)
        oy, ox = torch.minimum(oy, h - CROP), torch.minimum(ox, w - CROP)
        flip = u[:, 2] < 0.5
    else:
        oy, ox = cy - y0, cx - x0
        flip = torch.zeros(B, dtype=torch.bool, device=dev)
    ar = torch.arange(CROP, device=dev)
    ys = (y0 + oy)[:, None] + ar[None, :]                                          # [B,224]
    xs = (x0 + ox)[:, None] + torch.where(flip[:, None], CROP - 1 - ar[None, :], ar[None, :])
    b = torch.arange(B, device=dev)[:, None, None]
    return canvas_u8[b, ys[:, :, None], xs[:, None, :]]                            # [B,224,224,3]


class ShardLoader:
    """Iterable with the item layout of the reference's DataLoader over imsitu_loader: (names, img, verb, labels) per batch, with
    img = augmented uint8 NHWC [B,224,224,3] already on `device`.  One rank's share of every (shuffled) epoch under
    torch.distributed-style sharding: samples rank, rank+world, ... of the epoch's permutation.

    train=True: every rank gets the SAME number of samples (the permutation is padded by wrapping, as
    torch's DistributedSampler does), hence the same number of batches -- each training batch ends in a gradient all-reduce,
    and a rank with one batch more than its peers would wait in it forever.  train=False (evaluation: no collective inside
    the loop): no padding, every sample is seen exactly once across the ranks (sr.eval reduces sums and counts afterwards)."""

    def __init__(self, shard_dir, annotations, encoder, batch_size, device, train, rank=0, world=1, seed=0):
        with open(os.path.join(shard_dir, "index.json")) as f:
            idx = json.load(f)
        if idx["canvas"] != CANVAS or idx["crop"] != CROP:
            raise ValueError("shards were written with a different canvas / crop size")
        self.names = idx["names"]
        self.rects = torch.from_numpy(np.load(os.path.join(shard_dir, "rects.npy")))
        self.shards = [np.load(os.path.join(shard_dir, "shard_%05d.npy" % i), mmap_mode="r") for i in range(len(idx["sizes"]))]
        starts = np.cumsum([0] + idx["sizes"])
        self._where = [(s, i) for s, n in enumerate(idx["sizes"]) for i in range(n)]
        assert len(self._where) == len(self.names) == int(starts[-1])
        enc = [encoder.encode(annotations[n]) for n in self.names]                  # (verb id, labels [3,R]) once
        self.verbs = torch.tensor([v for v, _ in enc], dtype=torch.int64)
        self.labels = torch.stack([l for _, l in enc])
        self.batch_size, self.device, self.train = batch_size, torch.device(device), train
        self.rank, self.world, self.seed, self.epoch = rank, world, seed, 0
        self._gen = None

    def _per_rank(self):
        n = len(self.names)
        if self.train:
            return (n + self.world - 1) // self.world                          # padded: equal on every rank
        return (n - self.rank + self.world - 1) // self.world

    def __len__(self):
        return (self._per_rank() + self.batch_size - 1) // self.batch_size

    def set_epoch(self, epoch):
        """Same contract as DistributedSampler.set_epoch: the permutation of the next iteration."""
        self.epoch = int(epoch)

    def _gather(self, ids):
        buf = np.empty((len(ids), CANVAS, CANVAS, 3), dtype=np.uint8)
        for k, j in enumerate(ids):
            s, i = self._where[j]
            buf[k] = self.shards[s][i]
        return torch.from_numpy(buf)

    def __iter__(self):
        g = torch.Generator().manual_seed(self.seed + self.epoch)
        order = torch.randperm(len(self.names), generator=g) if self.train else torch.arange(len(self.names))
        if self.train and len(order) % self.world:
            order = torch.cat([order, order[: self.world - len(order) % self.world]])      # pad by wrapping
        mine = order[self.rank::self.world].tolist()
        assert len(mine) == self._per_rank()
        self.epoch += 1
        if self._gen is None and self.device.type == "cuda":
            self._gen = torch.Generator(device=self.device).manual_seed(self.seed * 7919 + self.rank)
        batches = [mine[b0:b0 + self.batch_size] for b0 in range(0, len(mine), self.batch_size)]
        for ids, canvas in self._prefetched(batches):
            if self.device.type == "cuda":
                canvas = canvas.to(self.device, non_blocking=True)
            t = torch.tensor(ids)
            img = gpu_augment(canvas, self.rects[t].to(self.device), self.train, self._gen)
            yield [self.names[j] for j in ids], img, self.verbs[t], self.labels[t]

    def _prefetched(self, batches, depth=2):
        """The host side of a batch (gather from the memory-mapped shards into pinned memory: ~1.5 GB at B=6144) runs on a
        background thread, `depth` batches ahead of the training step that consumes them."""
        import queue
        import threading
        q = queue.Queue(maxsize=depth)
        pin = self.device.type == "cuda"

        def work():
            try:
                for ids in batches:
                    c = self._gather(ids)
                    q.put((ids, c.pin_memory() if pin else c))
                q.put(None)
            except BaseException as e:           # surface loader errors in the consumer
                q.put(e)

        th = threading.Thread(target=work, daemon=True)
        th.start()
        while True:
            item = q.get()
            if item is None:
                break
            if isinstance(item, BaseException):
                raise item
            yield item
        th.join()
