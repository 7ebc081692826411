"""Pre-decoded uint8 shards (situation_recognition_amd/imsitu_shards.py): the dev path reproduces the reference's
Resize(224) + CenterCrop(224) pixels exactly, the train path is a RandomCrop(224) inside the resized image + flip, the loader
yields the reference's batch layout and shards an epoch over ranks without overlap.  CPU only (the gather is plain torch)."""
import json
import os

import numpy as np
import pytest
import torch

from situation_recognition_amd import imsitu_shards as sh
from situation_recognition_amd.imsitu_encoder import _resize_shorter, imsitu_encoder

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def shard_dir(tmp_path_factory):
    from PIL import Image
    d = tmp_path_factory.mktemp("imgs")
    ann = json.load(open(os.path.join(ROOT, "tests", "golden", "overfitting.json")))
    g = np.random.default_rng(0)
    sizes = [(300, 260), (224, 224), (500, 333), (333, 500), (640, 301)]          # (w, h): near-square, exact, 3:2, portrait, panoramic
    names = list(ann)[: len(sizes)]
    for n, (w, h) in zip(names, sizes):
        Image.fromarray(g.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(os.path.join(d, n), format="PNG")
    out = tmp_path_factory.mktemp("shards")
    assert sh.write_shards(str(d), names, str(out), per_shard=2) == 3
    return str(d), str(out), names, {n: ann[n] for n in names}


def test_dev_crop_equals_reference_transform(shard_dir):
    from PIL import Image
    img_dir, out, names, ann = shard_dir
    enc = imsitu_encoder(ann, quiet=True)
    dl = sh.ShardLoader(out, ann, enc, batch_size=3, device="cpu", train=False)
    assert len(dl) == 2
    seen = []
    for nm, img, verb, labels in dl:
        assert img.dtype == torch.uint8 and img.shape[1:] == (224, 224, 3)
        assert verb.shape == (len(nm),) and labels.shape == (len(nm), 3, enc.get_max_role_count())
        for k, n in enumerate(nm):
            im = _resize_shorter(Image.open(os.path.join(img_dir, n)).convert("RGB"), 224)
            w, h = im.size
            x0, y0 = int(round((w - 224) / 2.0)), int(round((h - 224) / 2.0))
            ref = np.asarray(im.crop((x0, y0, x0 + 224, y0 + 224)), dtype=np.uint8)
            assert np.array_equal(img[k].numpy(), ref), n
            v, l = enc.encode(ann[n])
            assert int(verb[k]) == v and torch.equal(labels[k], l)
        seen += nm
    assert seen == names


def test_train_crops_stay_inside_the_image_and_flip(shard_dir):
    img_dir, out, names, ann = shard_dir
    enc = imsitu_encoder(ann, quiet=True)
    dl = sh.ShardLoader(out, ann, enc, batch_size=7, device="cpu", train=True, seed=3)
    canvas = dl._gather(list(range(len(names))))
    rects = dl.rects
    g = torch.Generator().manual_seed(1)
    a = sh.gpu_augment(canvas, rects, True, g)
    assert a.shape == (len(names), 224, 224, 3)
    for k in range(len(names)):                      # every crop is a 224x224 window of the image's rectangle, possibly mirrored
        y0, x0, h, w = (int(v) for v in rects[k, :4])
        region = canvas[k, y0:y0 + h, x0:x0 + w]
        win = region.unfold(0, 224, 1).unfold(1, 224, 1).permute(0, 1, 3, 4, 2)       # [h-223, w-223, 224, 224, 3]
        hit = (win == a[k]).flatten(2).all(-1).any() or (win == a[k].flip(1)).flatten(2).all(-1).any()
        assert bool(hit), k
    flips = 0
    for s in range(20):                              # the flip really happens about half the time
        g = torch.Generator().manual_seed(100 + s)
        b = sh.gpu_augment(canvas[1:2], rects[1:2], True, g)                          # 224x224 image: the crop is the image
        flips += int(torch.equal(b[0], canvas[1, rects[1, 0]:rects[1, 0] + 224, rects[1, 1]:rects[1, 1] + 224].flip(1)))
    assert 3 <= flips <= 17


def test_ranks_partition_the_epoch(shard_dir):
    """Training shards: every rank gets the SAME number of samples and batches (the epoch's permutation is padded by
    wrapping, as DistributedSampler does) -- a rank with one batch more than its peers would block in the gradient
    all-reduce forever -- and together the ranks cover every sample.  Evaluation shards: unpadded, every sample once."""
    _, out, names, ann = shard_dir
    enc = imsitu_encoder(ann, quiet=True)
    N = len(names)
    for world, bs in ((2, 2), (3, 1), (3, 2), (4, 3)):
        got, lens = [], []
        for r in range(world):
            dl = sh.ShardLoader(out, ann, enc, batch_size=bs, device="cpu", train=True, rank=r, world=world, seed=5)
            batches = [list(nm) for nm, *_ in dl]
            assert len(batches) == len(dl)
            lens.append([len(b) for b in batches])
            got.append([n for b in batches for n in b])
        assert all(l == lens[0] for l in lens), (world, bs, lens)          # same batch count AND sizes on every rank
        per = -(-N // world)
        assert all(len(g) == per for g in got)
        assert set(sum(got, [])) == set(names)
        assert len(sum(got, [])) - len(set(sum(got, []))) == per * world - N   # only the wrap-around padding repeats
        ev = []
        for r in range(world):
            dl = sh.ShardLoader(out, ann, enc, batch_size=bs, device="cpu", train=False, rank=r, world=world)
            ev += [n for nm, *_ in dl for n in nm]
        assert sorted(ev) == sorted(names)
    # set_epoch selects the permutation (same contract as DistributedSampler.set_epoch)
    dl = sh.ShardLoader(out, ann, enc, batch_size=2, device="cpu", train=True, rank=0, world=1, seed=5)
    dl.set_epoch(3); a = [n for nm, *_ in dl for n in nm]
    dl.set_epoch(3); b = [n for nm, *_ in dl for n in nm]
    dl.set_epoch(4); c = [n for nm, *_ in dl for n in nm]
    assert a == b and sorted(a) == sorted(c) == sorted(names)


def test_driver_writes_shards_without_a_gpu(shard_dir, tmp_path):
    """`sr.py --make_shards`: one shard directory per annotation file, readable by ShardLoader."""
    from situation_recognition_amd import sr
    img_dir, _, names, ann = shard_dir
    ds = tmp_path / "imSitu"
    ds.mkdir()
    json.dump(ann, open(ds / "train.json", "w"))
    json.dump({n: ann[n] for n in names[:2]}, open(ds / "dev.json", "w"))
    out = tmp_path / "shards"
    sr.main(["--make_shards", "--shards", str(out), "--imgset_dir", img_dir, "--dataset_folder", str(ds)])
    enc = imsitu_encoder(ann, quiet=True)
    assert len(sh.ShardLoader(str(out / "train"), ann, enc, 8, "cpu", train=True).names) == len(names)
    assert len(sh.ShardLoader(str(out / "dev"), ann, enc, 8, "cpu", train=False).names) == 2
    assert not (out / "test").exists()
