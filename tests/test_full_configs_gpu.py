"""BASELINE.json configs at their real sizes.

config 2  ResNet-50 + 6-role GGNN T=4, batch 256, fp32: eval-mode logits of the first 8 images against the CPU oracle
          (<= 1e-3, the north_star tolerance), and -- size-independent property -- every image's logits in the batch-256
          run equal its logits when run in a batch of 8 (eval-mode BatchNorm makes images independent).
config 3  ResNet-152 + 6-role GGNN T=5, bf16, imSitu-sized vocabulary, batch 6144 (the benchmark's per-GPU size): same slicing
          property, eight images against the fp32 oracle, plus train-mode invariants (finite losses, running statistics move,
          gradients reach every trainable parameter).
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


def test_config3_resnet152_bf16_batch6144_slicing_oracle_and_train_invariants():
    """BASELINE config 3 at its FULL size (per-GPU batch 6144: the largest activation is 9.9 GB, row counts up to 77 M --
    the >2^31-byte regime): eval-mode logits of slices of the batch equal the same images run alone (eval BatchNorm
    makes images independent; same K order per output element whatever the batch), eight of them are compared with the fp32
    CPU oracle, and one train-mode step keeps the invariants of reference model.py:172-180 / sr.py:63-83."""
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
        parts = {lo: net(img[lo:lo + 32].contiguous(), verb[lo:lo + 32].contiguous()) for lo in (0, 3040, B - 32)}
        want = ora(img[3040:3048].float().cpu(), verb[3040:3048].cpu())
    for f in full:
        assert torch.isfinite(f).all()
    # A 32-image launch runs other kernels than the 6144-image one (narrow tiles; the expansion-conv kernel of csrc/expand.hip,
    # which rounds to bf16 once instead of twice, only serves launches of >= 32 768 rows), so the two differ by bf16 rounding
    # noise amplified through 152 layers -- the same bound as against the fp32 oracle below, not bit equality.
    # (pred_nouns is conditioned on argmax(pred_verb): only images whose predicted verb agrees between the two runs are compared)
    for lo, part in parts.items():
        same = (full[0][lo:lo + 32].float().argmax(1) == part[0].float().argmax(1))
        for k, (f, p) in enumerate(zip(full, part)):
            d = (f[lo:lo + 32].float() - p.float()).abs().flatten(1).max(1)[0]
            if k == 1:
                d = d[same]
            assert d.numel() == 0 or float(d.max()) <= 0.15, (lo, k, float(d.max()))
    # bf16 storage through 152 layers against the fp32 oracle: no 1e-3 claim here (that is config 2, fp32 storage).  Rounding
    # every activation and folded weight to 8 significant bits perturbs each layer by ~0.4 %, and a RANDOMLY INITIALISED
    # 152-layer net amplifies perturbations (it is not the contraction a trained net is): measured 0.20 relative L2 on the
    # pooled features and 0.06 absolute on the logits (|W_classifier| is small at init); asserted at 0.35 / 0.15.  The
    # arithmetic of every kernel on this path is pinned at 1.2e-2 by tests/test_production_shapes_gpu.py.
    # pred_nouns is not compared: it is conditioned on argmax(pred_verb), and this randomly initialised head's 504 verb logits
    # span 0.26 -- bf16 legitimately flips near-ties, which swaps the whole role table (the gt-verb branch has no such switch).
    with torch.no_grad():
        fv = net.convnet_verbs(img[3040:3048].contiguous()).float().cpu()
        wv = ora.convnet_verbs(img[3040:3048].float().cpu())
    rel = float((fv - wv).norm() / wv.norm())
    print("config3 bf16 vs fp32 oracle, pooled verb features: relative L2 error %.4f" % rel)
    assert rel <= 0.35, rel
    for f, w, name in ((full[0], want[0], "verb"), (full[2], want[2], "gt_nouns")):
        err = float((f[3040:3048].float().cpu() - w).abs().max())
        print("config3 bf16 vs fp32 oracle, %s logits: max abs err %.4f (logit range %.3f)" % (name, err, float(w.abs().max())))
        assert err <= 0.15, (name, err)
    del full, parts
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
