"""oracle/ref_rounded.py restates the oracle's eval-mode forward with a rounding hook at the HIP path's storage points.  With the hook
switched off it must BE the pinned fp32 oracle (same graph, BatchNorm folded into the weights): this is what lets the GPU test hold
the whole composed bf16 pass against it (tests/test_full_configs_gpu.py).  With bf16 rounding on, a small net stays close to the
fp32 oracle and the double-rounding predicate follows csrc/expand.hip's shape test."""
import torch

from oracle import ref_rounded
from oracle.ref_encoder import SyntheticEncoder
from oracle.ref_model import RefBackbone, RefFCGGNN
from oracle.ref_resnet import RefResNet, calibrate_batchnorm_, perturb_batchnorm_


def _model(depth, width, blocks, D, seed=0):
    torch.manual_seed(seed)
    enc = SyntheticEncoder(V=9, NR=7, L=13, R=4, seed=3)
    m = RefFCGGNN(enc, D, steps=3, backbone_factory=lambda: RefBackbone(depth, width, blocks))
    g = torch.Generator().manual_seed(seed + 1)
    img = torch.randn(6, 3, 64, 64, generator=g).clamp_(-2.2, 2.7)
    for i, net in enumerate((m.convnet_verbs, m.convnet_nouns)):
        perturb_batchnorm_(net, i + 1)
        calibrate_batchnorm_(net, img)
    return m.eval(), enc, img, torch.randint(0, 9, (6,), generator=g)


def test_unrounded_restatement_is_the_fp32_oracle():
    for depth, width, blocks, D in ((50, 16, (2, 1, 1, 1), 512), (18, 16, (1, 1, 1, 1), 128)):
        m, enc, img, verb = _model(depth, width, blocks, D)
        with torch.no_grad():
            want_v, _, want_g = m(img, verb)
            want_f = m.convnet_verbs(img)
        pv, pg, fv, fn = ref_rounded.fcggnn_eval(m, img, verb, rnd=ref_rounded.identity)
        assert (fv - want_f).abs().max() < 2e-5 * float(want_f.abs().max())
        assert (pv - want_v).abs().max() < 5e-5 * max(1.0, float(want_v.abs().max()))
        assert (pg - want_g).abs().max() < 5e-5 * max(1.0, float(want_g.abs().max()))


def test_bf16_rounding_stays_near_the_oracle_on_a_small_net_and_taps_every_block():
    m, enc, img, verb = _model(50, 16, (2, 1, 1, 1), 512)
    taps = []
    f16 = ref_rounded.resnet_eval_features(m.convnet_verbs.model, img, taps=taps)
    assert [n for n, _ in taps] == ["stem", "layer1.0", "layer1.1", "layer2.0", "layer3.0", "layer4.0"]
    assert torch.equal(f16, ref_rounded.bf16(f16))                       # stored values ARE bf16 values
    with torch.no_grad():
        f32 = m.convnet_verbs(img)
    rel = float((f16 - f32).norm() / f32.norm())
    assert 1e-4 < rel < 0.05, rel                                        # rounding happened, and only rounding


def test_double_rounding_predicate_follows_the_weight_stationary_kernels_shape_test():
    net = RefResNet(152)
    assert not ref_rounded.hip_staged_residual(net.layer1[0].conv3)      # 64 -> 256
    assert not ref_rounded.hip_staged_residual(net.layer2[1].conv3)      # 128 -> 512
    assert not ref_rounded.hip_staged_residual(net.layer3[7].conv3)      # 256 -> 1024
    assert ref_rounded.hip_staged_residual(net.layer4[1].conv3)          # 512 -> 2048: generic kernel, bf16-staged accumulator
    assert ref_rounded.hip_staged_residual(RefResNet(18).layer1[0].conv2)
