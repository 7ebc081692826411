"""FP8 (OCP e4m3) path of the matrix-bound 3x3 convolutions (BASELINE config 5).

Oracle: the fp32 restatement with the SAME quantised operands.  Products of two e4m3 numbers are exact in fp32 and both
sides accumulate in fp32, so the only differences are summation order (~1e-6 relative) and the bf16 rounding of the stored
output: tolerance 1.2e-2 of the output range on the stored tensor (as for every bf16 output in tests/test_kernels_gpu.py), 1e-4
relative on the fp32 BatchNorm statistics.  Quantisation itself (the e4m3 bytes the device writes) is compared bit for bit
with torch's float8_e4m3fn conversion.  The reference has no fp8 code (its reduced-precision route is autocast,
model.py:33,58,114,157,171): what an fp8 backbone SHOULD compute is defined here by that fake-quantised fp32 convolution.
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
F8 = torch.float8_e4m3fn


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from situation_recognition_amd import ops as o
    o.lib()
    return o


def q8(x, scale):
    """e4m3 fake quantisation as the device does it: clamp to +-448, round to nearest even."""
    return (x.float() * scale).clamp(-448.0, 448.0).to(F8)


def test_quantize_bytes_equal_torch_float8(ops):
    g = torch.Generator(device="cuda").manual_seed(0)
    x = (torch.randn(1000, 256, device="cuda", generator=g) * 3).to(torch.bfloat16)
    x[0, :8] = torch.tensor([0.0, 1e-4, -1e-4, 27.9, 28.1, 1000.0, -1000.0, 0.0155], device="cuda").to(torch.bfloat16)
    out = ops.quantize_fp8(x, 16.0)
    assert torch.equal(out, q8(x, 16.0).view(torch.uint8))
    sc, sh = 0.5 + torch.rand(256, device="cuda"), 0.3 * torch.randn(256, device="cuda")
    out2 = ops.quantize_fp8(x, 16.0, sc, sh, relu=True)
    want = q8(F.relu(x.float() * sc + sh), 16.0).view(torch.uint8)
    # (x*sc + sh is an FMA on the device and a multiply + add in torch: a value on a rounding boundary may land one code apart)
    diff = (out2.view(torch.int8).int() - want.view(torch.int8).int()).abs()
    assert int(diff.max()) <= 1 and float((diff > 0).float().mean()) < 1e-3
    xf = torch.randn(64, 64, device="cuda")
    assert torch.equal(ops.quantize_fp8(xf, 4.0), q8(xf, 4.0).view(torch.uint8))


@pytest.mark.parametrize("B,H,Cin,Cout,stride", [(64, 14, 256, 256, 1), (1024, 14, 256, 256, 1), (130, 28, 128, 128, 1),
                                                 (96, 28, 128, 128, 2), (77, 7, 512, 512, 1), (33, 15, 256, 384, 2)])
def test_conv3x3_fp8_vs_fp32_conv_of_the_same_quantised_operands(ops, B, H, Cin, Cout, stride):
    g = torch.Generator(device="cuda").manual_seed(B + H)
    s_a = 16.0
    act = F.relu(torch.randn(B, H, H, Cin, device="cuda", generator=g))              # post-BN-ReLU activations
    w = torch.randn(Cout, Cin, 3, 3, device="cuda", generator=g) * (9 * Cin) ** -0.5
    xq = q8(act, s_a)
    s_w = 448.0 / w.abs().amax(dim=(1, 2, 3))
    wq = q8(w * s_w.view(-1, 1, 1, 1), 1.0)
    dq = (1.0 / (s_a * s_w)).contiguous()
    wq_packed = wq.view(torch.uint8).permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous()
    y, stats = ops.conv3x3_fp8(xq.view(torch.uint8), wq_packed, dq, Cout, stride=stride, want_stats=True)
    ref = F.conv2d(xq.float().permute(0, 3, 1, 2), wq.float(), stride=stride, padding=1) * dq.view(1, -1, 1, 1)
    ref = ref.permute(0, 2, 3, 1)
    assert y.shape == ref.shape
    err = float((y.float() - ref).abs().max())
    assert err <= 1.2e-2 * float(ref.abs().max()), (err, float(ref.abs().max()))
    M = ref.numel() // Cout
    s1, s2 = stats[:, 0].double().sum(0), stats[:, 1].double().sum(0)
    r = ref.reshape(M, Cout).double()
    assert float((s1 - r.sum(0)).abs().max()) <= 1e-4 * float(r.abs().sum(0).max())
    assert float(((s2 - (r ** 2).sum(0)).abs() / (r ** 2).sum(0)).max()) <= 1e-4
    # the quantisation error itself, for the record: against the convolution of the UNquantised operands
    full = F.conv2d(act.permute(0, 3, 1, 2), w, stride=stride, padding=1).permute(0, 2, 3, 1)
    rel = float((ref - full).norm() / full.norm())
    assert rel < 0.06, rel                                  # e4m3 x e4m3: ~3-4 % relative L2 on a K = 9*Cin contraction


def test_conv3x3_fp8_is_bit_reproducible(ops):
    g = torch.Generator(device="cuda").manual_seed(3)
    B, H, Cc = 768, 14, 256
    xq = q8(F.relu(torch.randn(B, H, H, Cc, device="cuda", generator=g)), 16.0).view(torch.uint8)
    wq = q8(torch.randn(Cc, 9 * Cc, device="cuda", generator=g), 100.0).view(torch.uint8)
    dq = torch.full((Cc,), 1e-3, device="cuda")
    first = None
    for r in range(12):
        if r % 2:
            torch.empty(32 << 20, device="cuda").fill_(1.0)
        y, st = ops.conv3x3_fp8(xq, wq, dq, Cc, want_stats=True)
        if first is None:
            first = (y.clone(), st.clone())
        assert torch.equal(y, first[0]) and torch.equal(st, first[1])


def _fake_quant_conv2(net, s_a):
    """Oracle side of the fp8 backbone: every bottleneck 3x3 takes fake-quantised activations and weights (same scales as the
    HIP path: fixed activation scale, 448 / max|w| per output channel)."""
    import torch.nn as nn

    class FQ(nn.Module):
        def __init__(self, conv):
            super().__init__()
            self.conv = conv

        @property
        def weight(self):
            return self.conv.weight

        def forward(self, x):
            w = self.conv.weight
            s_w = 448.0 / w.abs().amax(dim=(1, 2, 3), keepdim=True)
            xq = (x * s_a).clamp(-448.0, 448.0).to(F8).float() / s_a
            wq = (w * s_w).clamp(-448.0, 448.0).to(F8).float() / s_w
            return F.conv2d(xq, wq, stride=self.conv.stride, padding=self.conv.padding)

    for m in list(net.modules()):
        if hasattr(m, "conv2") and hasattr(m, "conv3") and m.conv2.in_channels in (128, 256, 512) and m.conv2.out_channels % 128 == 0:
            m.conv2 = FQ(m.conv2)
    return net


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_fp8_backbone_vs_fake_quantised_oracle(ops, mode):
    """ResNet-50-shaped backbone, width 128 (so that layer1's 3x3 already has 128 channels), bf16 storage + fp8 3x3 against the
    fp32 oracle with fake-quantised 3x3 operands.  bf16 storage of every activation over 50 layers of a randomly initialised
    net bounds the agreement (tests/test_full_configs_gpu.py): relative L2 on the pooled features: measured 0.08, asserted at 0.12."""
    from oracle.ref_model import RefBackbone
    from oracle.ref_resnet import calibrate_batchnorm_, perturb_batchnorm_
    from situation_recognition_amd.model import resnet
    torch.manual_seed(0)
    ora = RefBackbone(depth=50, width=128, blocks=(1, 1, 2, 1))
    perturb_batchnorm_(ora, 4)
    img = torch.randn(24, 3, 96, 96).clamp_(-2.2, 2.7)
    calibrate_batchnorm_(ora, img)
    net = resnet(None, depth=50, width=128, blocks=(1, 1, 2, 1), dtype=torch.bfloat16, fp8=True)
    net.load_state_dict(ora.state_dict(), strict=True)
    net.cuda()
    plain = resnet(None, depth=50, width=128, blocks=(1, 1, 2, 1), dtype=torch.bfloat16)
    plain.load_state_dict(ora.state_dict(), strict=True)
    plain.cuda()
    _fake_quant_conv2(ora, net.fp8_act_scale)
    for m in (ora, net, plain):
        m.train(mode == "train")
    calls = []
    orig = ops.conv3x3_fp8
    ops.conv3x3_fp8 = lambda *a, **k: calls.append(1) or orig(*a, **k)
    try:
        with torch.no_grad():
            got = net(img.cuda()).float().cpu()
    finally:
        ops.conv3x3_fp8 = orig
    with torch.no_grad():
        want = ora(img)
        base = plain(img.cuda()).float().cpu()
    assert len(calls) == 4                                     # layers 1-3 (128 / 256 / 512 channels); layer4's 1024 stays bf16
    rel = float((got - want).norm() / want.norm())
    rel_plain = float((base - want).norm() / want.norm())
    print("fp8 backbone (%s) vs fake-quantised oracle: %.4f; bf16 backbone vs the same oracle: %.4f" % (mode, rel, rel_plain))
    assert rel <= 0.12, rel
    if mode == "train":                                        # running statistics moved as the oracle's did
        a, b = net.state_dict(), ora.state_dict()
        k = "model.layer3.1.bn2.running_var"
        assert float((a[k].cpu() - b[k.replace("bn2", "bn2")]).abs().max()) <= 5e-2 * float(b[k].abs().max())
