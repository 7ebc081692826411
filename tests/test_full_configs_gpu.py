"""BASELINE.json configs at their real sizes.

config 2  ResNet-50 + 6-role GGNN T=4, batch 256, fp32: eval-mode logits of the first 8 images against the CPU oracle
          (<= 1e-3, the north_star tolerance), and -- size-independent property -- every image's logits in the batch-256
          run equal its logits when run in a batch of 8 (eval-mode BatchNorm makes images independent).
config 3  ResNet-152 + 6-role GGNN T=5, bf16, imSitu-sized vocabulary, batch 6144 (the benchmark's per-GPU size): 1024-image slices
          equal the full run bit for bit, train-mode invariants (finite losses, running statistics move, gradients reach every
          trainable parameter); the end-to-end distance from the fp32 oracle is REPORTED (bf16 parity is gated per block in
          tests/test_blocks_teacher_forced_gpu.py).
config 5  ResNet-152 with e4m3 3x3 convolutions + 6-role GGNN T=8, batch 8192: launch count, slicing property, one training step.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _nets(backbone, steps, dtype, calib, seed=0):
    """Oracle + HIP model with the same weights, in the numerical regime of a TRAINED model: random BN affine parameters,
    running statistics calibrated on `calib` images (oracle.ref_resnet.calibrate_batchnorm_), embeddings and the last BN
    scaled so that node states and logits are O(1)-O(10) -- the range in which the reference's 1e-3 logit tolerance is meant."""
    from oracle.ref_encoder import SyntheticEncoder
    from oracle.ref_model import RefBackbone, RefFCGGNN
    from oracle.ref_resnet import calibrate_batchnorm_, perturb_batchnorm_
    from situation_recognition_amd.imsitu_encoder import imsitu_encoder
    from situation_recognition_amd.model import FCGGNN
    torch.manual_seed(seed)
    ora = RefFCGGNN(SyntheticEncoder(), 2048, steps=steps, backbone_factory=lambda: RefBackbone(backbone))
    perturb_batchnorm_(ora.convnet_verbs, 1)
    perturb_batchnorm_(ora.convnet_nouns, 2)
    torch.set_num_threads(max(torch.get_num_threads(), 16))
    calibrate_batchnorm_(ora.convnet_verbs, calib)
    calibrate_batchnorm_(ora.convnet_nouns, calib)
    with torch.no_grad():
        # pooled features of a calibrated net are O(1); keep node states = feature * role_emb * verb_emb in that range too
        ora.verb_emb.weight.mul_(0.5)
        ora.role_emb.weight.mul_(0.5)
    net = FCGGNN(imsitu_encoder.synthetic(), 2048, steps=steps, backbone=backbone, dtype=dtype)
    net.load_state_dict(ora.state_dict(), strict=True)
    return net.cuda(), ora


def test_config2_resnet50_fp32_batch256_vs_oracle():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    g = torch.Generator().manual_seed(5)
    img = torch.randn(256, 3, 224, 224, generator=g).clamp_(-2.2, 2.7)
    verb = torch.randint(0, 504, (256,), generator=g)
    net, ora = _nets(50, 4, torch.float32, img[200:216])
    net.eval(); ora.eval()
    with torch.no_grad():
        full = net(img.cuda(), verb.cuda())
        small = net(img[:8].cuda(), verb[:8].cuda())
        want = ora(img[:8], verb[:8])
    for f, s, w in zip(full, small, want):
        scale = max(1.0, float(w.abs().max()))
        err = float((s.cpu() - w).abs().max())
        print("config2 fp32 vs oracle: max abs err %.2e (logit range %.2f)" % (err, float(w.abs().max())))
        assert err < 1e-3, err                                              # north_star tolerance: ABSOLUTE 1e-3 on the logits
        assert float((f[:8] - s).abs().max()) < 2e-5 * scale                # batch independence in eval mode
    assert torch.equal(full[0].argmax(1)[:8].cpu(), want[0].argmax(1))


def _device_images(B, seed):
    gd = torch.Generator(device="cuda").manual_seed(seed)
    img = torch.empty((B, 3, 224, 224), device="cuda")
    for i in range(0, B, 512):
        img[i:i + 512] = torch.randn((min(512, B - i), 3, 224, 224), device="cuda", generator=gd).clamp_(-2.2, 2.7)
    return img


def _slices_equal_full(net, img, verb, full, starts, n):
    """Size-independent property of eval mode: BatchNorm uses running statistics, so images are independent, and a slice of
    `n` images that is dispatched to the same kernels as the full batch (n * 49 >= 32768 rows keeps even layer4 on the
    weight-stationary / 256-wide kernels; the K order of an output element does not depend on the tile it sits in) must
    reproduce the full run's logits BIT FOR BIT."""
    for lo in starts:
        with torch.no_grad():
            part = net(img[lo:lo + n].contiguous(), verb[lo:lo + n].contiguous())
        for k, (f, p) in enumerate(zip(full, part)):
            assert torch.equal(f[lo:lo + n], p), "output %d of images %d..%d differs between the full batch and the slice" % (k, lo, lo + n)


def test_config3_resnet152_bf16_batch6144_slicing_oracle_and_train_invariants():
    """BASELINE config 3 at its FULL size (per-GPU batch 6144: the largest activation is 9.9 GB, row counts up to 77 M --
    the >2^31-byte regime).  What is ASSERTED: finite logits; 1024-image slices of the batch reproduce the full run bit for bit
    (eval-mode images are independent); one train-mode step keeps the invariants of reference model.py:172-180 / sr.py:63-83.
    What is REPORTED, not gated: the distance of eight images' features / logits from the fp32 oracle.  A randomly initialised
    152-layer net amplifies the bf16 rounding of every activation (measured in round 2: 0.20 relative L2 on pooled features, 0.06
    absolute on logits whose range is 0.26), so an end-to-end tolerance would have to be fitted to the measurement and would
    let a wrong layer pass.  The bf16 parity claim rests on tests/test_blocks_teacher_forced_gpu.py (every block against the
    oracle on the oracle's own input, production-size launches, 1.2e-2) and tests/test_production_shapes_gpu.py (every kernel)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    g = torch.Generator().manual_seed(6)
    B = 6144
    img = _device_images(B, 6)
    verb = torch.randint(0, 504, (B,), generator=g).cuda()
    net, ora = _nets(152, 5, torch.bfloat16, img[100:116].cpu())
    net.eval(); ora.eval()
    with torch.no_grad():
        full = net(img, verb)
        want = ora(img[3040:3048].float().cpu(), verb[3040:3048].cpu())
    for f in full:
        assert torch.isfinite(f).all()
    _slices_equal_full(net, img, verb, full, (0, 3072, B - 1024), 1024)
    # packed role rows (automatic at this size: 21 6xx of the 36 864 role rows are real) against the full form: bit-identical
    assert net._use_packed(B, 6)
    net.pack_roles = False
    with torch.no_grad():
        unpacked = net(img, verb)
    net.pack_roles = None
    for f, u in zip(full, unpacked):
        assert torch.equal(f, u)
    del unpacked
    with torch.no_grad():
        fv = net.convnet_verbs(img[3040:3048].contiguous()).float().cpu()
        wv = ora.convnet_verbs(img[3040:3048].float().cpu())
    rel = float((fv - wv).norm() / wv.norm())
    print("REPORTED config3 bf16 vs fp32 oracle (8 images, end to end through 152 layers), pooled verb features: relative L2 error %.4f" % rel)
    assert rel < 1.0                                   # sanity only (an unrelated tensor gives ~1.4); see the docstring
    for f, w, name in ((full[0], want[0], "verb"), (full[2], want[2], "gt_nouns")):
        err = float((f[3040:3048].float().cpu() - w).abs().max())
        print("REPORTED config3 bf16 vs fp32 oracle, %s logits: max abs err %.4f (logit range %.3f)" % (name, err, float(w.abs().max())))
    del full
    # train mode: one step's invariants at the full batch
    net.train()
    nouns = torch.randint(0, 2001, (B, 3, 6), generator=g).cuda()
    rv0 = net.convnet_verbs.model.layer3[5].bn2.running_var.clone()
    pv, pn, pg = net(img, verb)
    for t in (pv, pn, pg):
        assert torch.isfinite(t).all()
    loss = net.verb_loss(pv, verb) + net.nouns_loss(pn, nouns)
    assert torch.isfinite(loss) and 5.0 < float(net.verb_loss(pv, verb)) < 9.0          # ~ ln(504) at init
    loss.backward()
    for k, p in net.named_parameters():
        if p.requires_grad:
            assert p.grad is not None and torch.isfinite(p.grad).all(), k
            assert float(p.grad.abs().max()) > 0, k
        else:
            assert p.grad is None
    assert not torch.equal(rv0, net.convnet_verbs.model.layer3[5].bn2.running_var)
    assert int(net.state_dict()["convnet_nouns.model.bn1.num_batches_tracked"]) == 2        # two passes' worth (model.py:176-178)
    assert int(net.state_dict()["convnet_verbs.model.bn1.num_batches_tracked"]) == 1


def test_config5_resnet152_fp8_T8_batch8192_at_its_workload():
    """BASELINE config 5 at its per-GPU workload: ResNet-152 with the e4m3 3x3 convolutions (`fp8=True`), T = 8, batch 8192.
    The reference has no fp8 code (parity unpinned: what the fp8 kernels must compute is defined and tested at kernel level in
    tests/test_fp8_gpu.py); here the FULL-SIZE run is held to size-independent properties: the expected number of fp8 launches
    (every eligible 3x3 of both backbones: 8 + 36 + 3 = 47 per pass), finite logits, 1024-image slices bit-identical to the full
    batch in eval mode, and one training step (reference sr.py:63-83) with finite loss, a gradient on every trainable parameter
    and the BatchNorm buffers of both backbones updated (model.py:176-178: the noun backbone twice)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from situation_recognition_amd import ops
    from situation_recognition_amd.imsitu_encoder import imsitu_encoder
    from situation_recognition_amd.model import FCGGNN
    torch.manual_seed(21)
    B, T = 8192, 8
    net = FCGGNN(imsitu_encoder.synthetic(), 2048, steps=T, backbone=152, dtype=torch.bfloat16, fp8=True).cuda()
    eligible = sum(1 for convs, _ in net.convnet_verbs._plan()[1] if convs[1].fp8_eligible())
    assert eligible == 8 + 36 + 3
    img = _device_images(B, 8)
    g = torch.Generator().manual_seed(8)
    verb = torch.randint(0, 504, (B,), generator=g).cuda()
    nouns = torch.randint(0, 2001, (B, 3, 6), generator=g).cuda()
    calls = []
    fp8_conv = ops.conv3x3_fp8

    def counted(xq, wq, dq, Cout, stride=1, want_stats=False):
        calls.append((tuple(xq.shape), Cout, stride))
        return fp8_conv(xq, wq, dq, Cout, stride=stride, want_stats=want_stats)

    ops.conv3x3_fp8 = counted
    try:
        net.eval()
        with torch.no_grad():
            full = net(img, verb)
        assert len(calls) == 2 * eligible, len(calls)                   # one pass per backbone, every eligible 3x3 in e4m3
        assert all(c[0][0] == B for c in calls)
        for f in full:
            assert torch.isfinite(f).all()
        _slices_equal_full(net, img, verb, full, (0, B - 1024), 1024)
        del full
        calls.clear()
        net.train()
        rv0 = net.convnet_nouns.model.layer3[20].bn2.running_var.clone()
        pv, pn, pg = net(img, verb)
        assert len(calls) == 2 * eligible
    finally:
        ops.conv3x3_fp8 = fp8_conv
    for t in (pv, pn, pg):
        assert torch.isfinite(t).all()
    assert tuple(pv.shape) == (B, 504) and tuple(pn.shape) == (B, 6, 2001) and tuple(pg.shape) == (B, 6, 2001)
    loss = net.verb_loss(pv, verb) + net.nouns_loss(pn, nouns)
    assert torch.isfinite(loss) and 5.0 < float(net.verb_loss(pv, verb)) < 9.0          # ~ ln(504) at init
    loss.backward()
    params = [p for p in net.parameters() if p.requires_grad]
    for k, p in net.named_parameters():
        if p.requires_grad:
            assert p.grad is not None and torch.isfinite(p.grad).all() and float(p.grad.abs().max()) > 0, k
    gn = torch.nn.utils.clip_grad_norm_(params, 1.0)
    assert torch.isfinite(gn)
    torch.optim.Adamax(params, lr=0.002).step()
    assert not torch.equal(rv0, net.convnet_nouns.model.layer3[20].bn2.running_var)
    sd = net.state_dict()
    assert int(sd["convnet_nouns.model.bn1.num_batches_tracked"]) == 2 and int(sd["convnet_verbs.model.bn1.num_batches_tracked"]) == 1


def test_gram_statistics_route_matches_statistics_only_launch_in_the_backbone():
    """Train-mode ResNet-50 pass (bf16, batch 352: the expansion convs of layers 1-3 are above the 256*C-pixel threshold
    and take the Gram route, layer4 the statistics-only launch).  At every expansion conv that takes the Gram route the
    scale/shift it produces are compared IN SITU with the ones a statistics-only launch of the conv gives on the same
    input (end-to-end features are not compared: 50 bf16 layers of batch-normalised random weights amplify one-ulp
    differences to several percent)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from situation_recognition_amd import ops
    from situation_recognition_amd.model import resnet
    torch.manual_seed(3)
    net = resnet(None, depth=50, dtype=torch.bfloat16).cuda().train()
    img = torch.randn(352, 3, 224, 224, device="cuda").clamp_(-2.2, 2.7)
    seen, last_x = [], {}
    gram0, fin0, fused0, lazy0 = ops.gram, ops.bn_finalize_gram, ops.bn_apply_gram, ops.bn_gram

    def gram(x2d):
        last_x["x"] = x2d
        return gram0(x2d)

    def fused(x2d, scale, shift):                # BN-apply of the 3x3's output + Gram partials in one sweep (in place)
        part = fused0(x2d, scale, shift)
        last_x["x"] = x2d
        return part

    def lazy(x2d, scale, shift):                 # ... or the Gram partials alone: the expansion conv normalises the raw tensor on load
        part = lazy0(x2d, scale, shift)
        last_x["x"] = ops.bn_apply(x2d, scale, shift, relu=True)        # (a copy: what the conv will see)
        return part

    def fin(part, w, count, gamma, beta, rm, rv, momentum, eps, twin=None):
        scale, shift = fin0(part, w, count, gamma, beta, rm, rv, momentum, eps, twin=twin)
        x2d = last_x["x"]
        st = ops.conv2d(x2d.view(1, x2d.shape[0], 1, x2d.shape[1]), w.reshape(w.shape[0], -1), w.shape[0], 1, 1, 0, stats_only=True)
        s2, h2 = ops.bn_finalize(st, count, gamma, beta, None, None, momentum, eps)
        seen.append((w.shape[1], float(((scale - s2).abs() / s2.abs().clamp_min(1e-3)).max()), float((shift - h2).abs().max())))
        return scale, shift

    ops.gram, ops.bn_finalize_gram, ops.bn_apply_gram, ops.bn_gram = gram, fin, fused, lazy
    try:
        f = net(img)
    finally:
        ops.gram, ops.bn_finalize_gram, ops.bn_apply_gram, ops.bn_gram = gram0, fin0, fused0, lazy0
    assert torch.isfinite(f).all()
    assert sorted(set(c for c, _, _ in seen)) == [64, 128, 256] and len(seen) == (3 + 1) + 4 + 6   # (+1: layer1's stride-1 downsample)
    for c, ds, dh in seen:
        assert ds < 5e-5 and dh < 5e-4, (c, ds, dh)
