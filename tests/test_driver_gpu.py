"""The sr.py-compatible driver end to end on the reference's own 5-image overfitting fixture (random pixels stand in
for the JPEGs, which the reference does not ship): two epochs of training, checkpoint, resume, evaluation."""
import json
import os

import numpy as np
import pytest
import torch

from golden_util import overfitting_json

pytestmark = pytest.mark.gpu


def test_train_checkpoint_resume_eval(tmp_path, capsys):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from PIL import Image
    from situation_recognition_amd import sr
    ann = overfitting_json()
    ds, imgs, ck = tmp_path / "imSitu", tmp_path / "img", tmp_path / "ck"
    for d in (ds, imgs, ck):
        d.mkdir()
    rng = np.random.default_rng(0)
    for name in ann:
        Image.fromarray((rng.random((240, 300, 3)) * 255).astype(np.uint8)).save(imgs / name)
    for f in ("train.json", "dev.json", "test.json"):
        json.dump(ann, open(ds / f, "w"))
    common = ["--dataset_folder", str(ds), "--imgset_dir", str(imgs), "--saving_folder", str(ck), "--batch_size", "5",
              "--num_workers", "0", "--backbone", "18", "--dtype", "fp32", "--lr", "0.01"]
    sr.main(common + ["--epochs", "3"])
    out = capsys.readouterr().out
    assert "Model training started!" in out and "Epoch-2, lr: 0.0100" in out and "training losses = [v:" in out and "5-verb:" in out
    state = torch.load(ck / "sr", map_location="cpu", weights_only=True)
    assert state["epoch"] == 3 and len(state["verb_losses"]) == 3
    # 5 random-pixel images, batch statistics over 5 samples, Dropout(0.5) and random crops: the loss is noisy; it must
    # stay finite and in the range of ln(#classes) per term (convergence is covered by the oracle-parity tests)
    assert all(np.isfinite(v) and 0 < v < 10 for v in state["verb_losses"])
    assert all(np.isfinite(v) and 0 < v < 40 for v in state["nouns_losses"])
    assert "convnet_nouns.model.layer4.1.bn2.running_var" in state["model_state_dict"]
    assert os.path.isfile(ck / "encoder.json")
    sr.main(common + ["--epochs", "4", "--resume_model", "sr"])            # resumes at epoch 3
    out = capsys.readouterr().out
    assert "Resume training from: sr" in out and "Epoch-3" in out and "Epoch-2," not in out
    sr.main(common + ["--evaluate_dev", "--resume_model", "sr"])
    assert "val losses = [v:" in capsys.readouterr().out


def test_train_and_eval_from_uint8_shards(tmp_path, capsys):
    """The same driver fed from pre-decoded uint8 shards: --make_shards once (host only), then training and evaluation read
    the shards, crop / flip on the GPU and enter the model through its uint8 NHWC input path."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from PIL import Image
    from situation_recognition_amd import sr
    ann = overfitting_json()
    ds, imgs, ck, shards = tmp_path / "imSitu", tmp_path / "img", tmp_path / "ck", tmp_path / "shards"
    for d in (ds, imgs, ck):
        d.mkdir()
    rng = np.random.default_rng(1)
    for name in ann:
        Image.fromarray((rng.random((250, 320, 3)) * 255).astype(np.uint8)).save(imgs / name)
    for f in ("train.json", "dev.json"):
        json.dump(ann, open(ds / f, "w"))
    common = ["--dataset_folder", str(ds), "--imgset_dir", str(imgs), "--saving_folder", str(ck), "--batch_size", "5",
              "--num_workers", "0", "--backbone", "18", "--dtype", "bf16", "--lr", "0.01", "--shards", str(shards)]
    sr.main(common + ["--make_shards"])
    assert os.path.isfile(shards / "train" / "index.json") and os.path.isfile(shards / "dev" / "shard_00000.npy")
    capsys.readouterr()
    os.rename(imgs, tmp_path / "img_gone")                                   # nothing may touch the image files any more
    sr.main(common + ["--epochs", "2"])
    out = capsys.readouterr().out
    assert "Epoch-1, lr: 0.0100" in out and "training losses = [v:" in out and "val losses = [v:" in out
    state = torch.load(ck / "sr", map_location="cpu", weights_only=True)
    assert state["epoch"] == 2 and all(np.isfinite(v) and 0 < v < 10 for v in state["verb_losses"])
    # dev-set evaluation from shards equals evaluation from the image files (same pixels: CenterCrop of the same resize)
    os.rename(tmp_path / "img_gone", imgs)
    sr.main(common + ["--evaluate_dev", "--resume_model", "sr"])
    a = capsys.readouterr().out
    sr.main([c for c in common if c not in ("--shards", str(shards))] + ["--evaluate_dev", "--resume_model", "sr"])
    b = capsys.readouterr().out
    line = lambda s: [l for l in s.splitlines() if l.startswith("val losses")][0]
    assert line(a) == line(b)
