"""BASELINE.json configs at their real sizes.

config 2  ResNet-50 + 6-role GGNN T=4, batch 256, fp32: eval-mode logits of the first 8 images against the CPU oracle
          (<= 1e-3, the north_star tolerance), and -- size-independent property -- every image's logits in the batch-256
          run equal its logits when run in a batch of 8 (eval-mode BatchNorm makes images independent).
config 3  ResNet-152 + 6-role GGNN T=5, bf16, imSitu-sized vocabulary, batch 6144 (the benchmark's per-GPU size): 1024-image slices
          equal the full run bit for bit, train-mode invariants (finite losses, running statistics move, gradients reach every
          trainable parameter); the composed eval pass (both backbones end to end + heads) is GATED against the rounding-matched
          oracle (oracle/ref_rounded.py); the distance from the pure fp32 oracle is reported next to it.
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


def _composed_pass_gate(net, ora, img, verb, lo8, s0, damp=0.2):
    """The composed eval pass (reference model.py:172-180: both 152-layer backbones end to end, GGNN, classifiers) GATED against the
    rounding-matched oracle (oracle/ref_rounded.py) on a net in the regime of a TRAINED ResNet: the same weights with every block's
    last BatchNorm gamma x `damp` and the running statistics re-calibrated -- residual branches that add a fraction to the trunk
    instead of doubling it, so that a perturbation is carried, not amplified to O(1) (see ref_rounded's docstring for the numbers).
    Tolerance, derived on the spot: FLOOR = the distance between two rounding-matched oracle runs that differ only in the
    convolutions' summation precision (fp32 vs fp64) = what two CORRECT bf16-storage implementations differ by; the HIP pass must be
    within 2 x FLOOR of the oracle (measured: 1.03-1.2 x; the wrong-layer probe lands at 6.4 x).  And the gate must be able to see a wrong layer: ONE BatchNorm bias of the oracle shifted by 0.5
    (layer3.17.bn1, one of 155) must land outside it.  HIP values come from batch-6144 / 1024-image runs (the benchmark's kernels)."""
    from oracle import ref_rounded
    from oracle.ref_resnet import calibrate_batchnorm_
    B = img.shape[0]
    with torch.no_grad():
        for bb in (ora.convnet_verbs, ora.convnet_nouns):
            for m in bb.modules():
                if hasattr(m, "bn3"):
                    m.bn3.weight.mul_(damp)
            calibrate_batchnorm_(bb, img[100:116].float().cpu())
    net.load_state_dict(ora.state_dict(), strict=True)
    net.eval(); ora.eval()
    img8, verb8 = img[lo8:lo8 + 8].float().cpu(), verb[lo8:lo8 + 8].cpu()
    with torch.no_grad():
        full = net(img, verb)
        fv = net.convnet_verbs(img[s0:s0 + 1024].contiguous())[lo8 - s0:lo8 - s0 + 8].float().cpu()
        fn = net.convnet_nouns(img[s0:s0 + 1024].contiguous())[lo8 - s0:lo8 - s0 + 8].float().cpu()
        pure = ora(img8, verb8)
        pure_fv = ora.convnet_verbs(img8)
    mv, mg, mfv, mfn = ref_rounded.fcggnn_eval(ora, img8, verb8)
    with ref_rounded.conv_precision(f64=True):
        dv, dg, dfv, dfn = ref_rounded.fcggnn_eval(ora, img8, verb8)
    rel = lambda a, b: float((a - b).norm() / b.norm())
    # features
    for got, m32, m64, name in ((fv, mfv, dfv, "verb"), (fn, mfn, dfn, "noun")):
        floor, err = rel(m32, m64), rel(got, m32)
        print("GATED config3 (damped net) pooled %s features vs rounding-matched oracle: relative L2 %.5f; floor (fp32 vs fp64 summation) %.5f; "
              "gate 2 x floor = %.5f" % (name, err, floor, 2 * floor))
        assert err < 2 * floor, (name, err, floor)
    print("REPORTED config3 (damped net) pooled verb features vs pure fp32 oracle: relative L2 %.5f" % rel(fv, pure_fv))
    # logits (verb branch, ground-truth-verb noun branch)
    for got, m32, m64, w, name in ((full[0], mv, dv, pure[0], "verb"), (full[2], mg, dg, pure[2], "gt_nouns")):
        got = got[lo8:lo8 + 8].float().cpu()
        floor, err, rng = float((m32 - m64).abs().max()), float((got - m32).abs().max()), float(w.abs().max())
        print("GATED config3 (damped net) %s logits vs rounding-matched oracle: max abs err %.5f; floor %.5f; gate %.5f; logit range %.3f; "
              "vs pure fp32 oracle %.5f" % (name, err, floor, 2 * floor, rng, float((got - w).abs().max())))
        assert err < 2 * floor, (name, err, floor)
    # the gate sees a wrong layer: one BatchNorm bias of 155 off by 0.5
    blk = ora.convnet_verbs.model.layer3[17]
    with torch.no_grad():
        blk.bn1.bias.add_(0.5)
        bad = ref_rounded.resnet_eval_features(ora.convnet_verbs.model, img8)
        blk.bn1.bias.sub_(0.5)
    moved = rel(bad, mfv)
    print("config3 (damped net): one BatchNorm bias (layer3.17.bn1) shifted by 0.5 moves the pooled features by %.5f" % moved)
    assert moved > 2 * rel(mfv, dfv), "the gate could not see a wrong layer"


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
    the >2^31-byte regime).  ASSERTED: finite logits; 1024-image slices of the batch reproduce the full run bit for bit
    (eval-mode images are independent); packed role rows equal the full form bit for bit; the COMPOSED eval pass -- both 152-layer
    backbones end to end, GGNN, classifiers -- of eight images against the rounding-matched oracle (oracle/ref_rounded.py: fp32
    arithmetic, bf16 rounding at the HIP path's storage points; FEAT_GATE / LOGIT_GATE above); one train-mode step keeps the
    invariants of reference model.py:172-180 / sr.py:63-83.  REPORTED next to the gate: the distance from the pure fp32 oracle
    (0.20 relative L2 on pooled features, 0.06 absolute on logits of range 0.26 -- what rounding every stored activation to 8 bits
    does to a randomly initialised 152-layer net; a tolerance on THAT number would have to be fitted to it and would let a wrong
    layer pass, which is why the gate is on the rounding-matched comparison).  Per-block and per-kernel bf16 parity against the fp32
    oracle: tests/test_blocks_teacher_forced_gpu.py, tests/test_production_shapes_gpu.py."""
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
    # ---- REPORTED: distances of the composed bf16 pass (both 152-layer backbones + heads, eight images) on THIS net -- torchvision's
    # initialisation with perturbed BatchNorm affines.  It amplifies ANY perturbation to O(1) by layer4 (two rounding-matched oracle
    # runs that differ only in summation precision end 0.14 apart: oracle/ref_rounded.py), so no tolerance on these numbers can tell a
    # wrong layer from rounding.  The gate is `_composed_pass_gate` below, on the same net with damped residual branches.
    from oracle import ref_rounded
    lo8, s0 = 3040, 2560
    img8, verb8 = img[lo8:lo8 + 8].float().cpu(), verb[lo8:lo8 + 8].cpu()
    with torch.no_grad():
        fv = net.convnet_verbs(img[s0:s0 + 1024].contiguous())[lo8 - s0:lo8 - s0 + 8].float().cpu()
        wv = ora.convnet_verbs(img8)
    rel32 = float((fv - wv).norm() / wv.norm())
    relm = float((fv - ref_rounded.resnet_eval_features(ora.convnet_verbs.model, img8)).norm() / wv.norm())
    print("REPORTED config3 bf16 vs fp32 oracle (8 images, end to end through 152 layers), pooled verb features: relative L2 error %.4f "
          "(vs the rounding-matched oracle: %.4f)" % (rel32, relm))
    assert rel32 < 1.0                                   # sanity only (an unrelated tensor gives ~1.4)
    for f, w, name in ((full[0], want[0], "verb"), (full[2], want[2], "gt_nouns")):
        err = float((f[lo8:lo8 + 8].float().cpu() - w).abs().max())
        print("REPORTED config3 bf16 vs fp32 oracle, %s logits: max abs err %.4f (logit range %.3f)" % (name, err, float(w.abs().max())))
    del full
    _composed_pass_gate(net, ora, img, verb, lo8, s0)
    net.train(); ora.train()
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
