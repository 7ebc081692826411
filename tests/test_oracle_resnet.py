"""The backbone arithmetic is third-party (torchvision, absent here): the oracle's
restatement is cross-checked against an independent implementation of the same
published graph, transformers.ResNetModel built FROM CONFIG (no download)."""
import pytest
import torch

from oracle.ref_resnet import RefResNet, perturb_batchnorm_

transformers = pytest.importorskip("transformers")


def _hf(depths, hidden, emb, layer_type):
    cfg = transformers.ResNetConfig(num_channels=3, embedding_size=emb, hidden_sizes=hidden, depths=depths,
                                    layer_type=layer_type, hidden_act="relu", downsample_in_bottleneck=False)
    return transformers.ResNetModel(cfg).eval()


def _copy(ours, hf, kind):
    """Map torchvision-style names onto the transformers module tree."""
    def cp(conv_bn_dst, conv, bn):
        conv_bn_dst.convolution.load_state_dict(conv.state_dict())
        conv_bn_dst.normalization.load_state_dict(bn.state_dict())
    cp(hf.embedder.embedder, ours.conv1, ours.bn1)
    for s in range(4):
        for i, blk in enumerate(getattr(ours, "layer%d" % (s + 1))):
            dst = hf.encoder.stages[s].layers[i]
            n = 3 if kind == "bottleneck" else 2
            for j in range(n):
                cp(dst.layer[j], getattr(blk, "conv%d" % (j + 1)), getattr(blk, "bn%d" % (j + 1)))
            if blk.downsample is not None:
                cp(dst.shortcut, blk.downsample[0], blk.downsample[1])


@pytest.mark.parametrize("kind,depth,blocks", [("bottleneck", 50, (1, 2, 1, 1)), ("basic", 18, (1, 1, 2, 1))])
def test_restatement_matches_independent_implementation(kind, depth, blocks):
    torch.manual_seed(0)
    w = 8
    ours = perturb_batchnorm_(RefResNet(depth, width=w, blocks=blocks), 1).eval()
    exp = 4 if kind == "bottleneck" else 1
    hf = _hf(list(blocks), [w * exp << s for s in range(4)], w, kind)
    _copy(ours, hf, kind)
    ours.fc = torch.nn.Identity()
    x = torch.randn(2, 3, 64, 64)
    with torch.no_grad():
        a = ours(x)
        b = hf(x).pooler_output.flatten(1)
    assert a.shape == b.shape
    assert (a - b).abs().max() < 1e-5 * max(1.0, float(b.abs().max()))


def test_resnet152_parameter_count():
    n = sum(p.numel() for p in RefResNet(152).parameters())
    assert n - (2048 * 1000 + 1000) == 58_143_808
