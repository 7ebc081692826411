"""End-to-end parity of the HIP model (situation_recognition_amd.model, through libsrhip.so) against
(a) golden vectors produced by the reference itself and (b) the CPU oracle on the same seeded inputs.
Tolerance for fp32 logits: 1e-3 max-abs (BASELINE north_star)."""
import numpy as np
import pytest
import torch

from golden_util import load, oracle_fcggnn, overfitting_json, sub

pytestmark = pytest.mark.gpu
TOL = 1e-3


@pytest.fixture(scope="module")
def sra():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import situation_recognition_amd.model as m
    from situation_recognition_amd import ops
    from situation_recognition_amd.imsitu_encoder import imsitu_encoder
    ops.lib()
    return m, imsitu_encoder


def hip_fcggnn(sra, g3, dtype=torch.float32, steps=4):
    m, Enc = sra
    enc = Enc(overfitting_json(), quiet=True)
    net = m.FCGGNN(enc, int(g3["D"]), steps=steps, backbone=int(g3["cfg_depth"]), dtype=dtype,
                   width=int(g3["cfg_width"]), blocks=tuple(int(b) for b in g3["cfg_blocks"]))
    net.load_state_dict(sub(g3, "state/"), strict=True)
    return net.cuda(), enc


@pytest.mark.parametrize("path", ["noun", "verb"])
def test_g2_ggsnn_forward_backward_fp32(sra, path):
    m, _ = sra
    g = load("g2_ggsnn.npz")
    gg = m.GGSNN(64, steps=4)
    gg.load_state_dict(sub(g, "w/"))
    gg.cuda()
    if path == "noun":
        h = torch.from_numpy(g["h0"]).cuda().requires_grad_(True)
        out = gg(h, mask=torch.from_numpy(g["mask"]).cuda(), verb=False)
        c, want, gp, gh = g["cn"], g["out_n"], "gn/", "gn/h0"
    else:
        h = torch.from_numpy(g["hv"]).cuda().requires_grad_(True)
        out = gg(h, mask=None, verb=True)
        c, want, gp, gh = g["cv"], g["out_v"], "gv/", "gv/hv"
    assert np.abs(out.detach().cpu().numpy() - want).max() < 1e-4
    (out * torch.from_numpy(c).cuda()).sum().backward()
    assert np.abs(h.grad.cpu().numpy() - g[gh]).max() < 1e-4 * max(1.0, np.abs(g[gh]).max())
    for k, p in gg.named_parameters():
        ref = g[gp + k]
        assert np.abs(p.grad.cpu().numpy() - ref).max() <= 2e-4 * max(1.0, np.abs(ref).max()), k


def test_g2_ggsnn_bf16_close(sra):
    """bf16 storage vs the fp32 oracle on the G2 inputs.  G2's weights are the default init x3 (to saturate the
    gates), which amplifies bf16 rounding ~50x over 4 steps, so this check uses them at their natural scale."""
    from oracle.ref_model import RefGGSNN
    m, _ = sra
    g = load("g2_ggsnn.npz")
    w = {k: v / 3.0 for k, v in sub(g, "w/").items()}
    gg, ora = m.GGSNN(64, steps=4), RefGGSNN(64, steps=4)
    gg.load_state_dict(w); ora.load_state_dict(w)
    gg.cuda()
    h0, mask = torch.from_numpy(g["h0"]), torch.from_numpy(g["mask"])
    with torch.no_grad():
        want = ora(h0.bfloat16().float(), mask=mask, verb=False)
        out = gg(h0.cuda().bfloat16(), mask=mask.cuda(), verb=False)
    assert out.dtype == torch.bfloat16
    assert float((out.float().cpu() - want).abs().max()) < 0.05 * max(1.0, float(want.abs().max()))


@pytest.mark.parametrize("tag", ["bottleneck", "basic"])
def test_g3_fcggnn_eval_logits_vs_reference_golden(sra, tag):
    g = load("g3_fcggnn_%s.npz" % tag)
    net, enc = hip_fcggnn(sra, g)
    net.eval()
    img, gv = torch.from_numpy(g["img"]).cuda(), torch.from_numpy(g["gt_verb"]).cuda()
    with torch.no_grad():
        fv = net.convnet_verbs(img)
        assert np.abs(fv.cpu().numpy() - g["feat_verbs"]).max() < TOL
        pv, pn, pg = net(img, gv)
        given = net.predict_nouns(img, torch.from_numpy(g["given_verbs"]).cuda(), 5)
    assert np.abs(pv.cpu().numpy() - g["pred_verb"]).max() < TOL
    assert np.array_equal(pv.argmax(1).cpu().numpy(), g["pred_verb"].argmax(1))
    assert np.abs(pn.cpu().numpy() - g["pred_nouns"]).max() < TOL
    assert np.abs(pg.cpu().numpy() - g["gt_pred_nouns"]).max() < TOL
    assert np.abs(given.cpu().numpy() - g["pred_nouns_given"]).max() < TOL


def test_g5_training_step_vs_reference_golden(sra):
    """One full step (sr.py:63-83): train-mode BatchNorm in both backbones, both GGNN paths,
    hand-written backward, clip_grad_norm_, Adamax -- against the state the reference reached."""
    g3, g = load("g3_fcggnn_basic.npz"), load("g5_train_step.npz")
    net, enc = hip_fcggnn(sra, g3)
    net.train()
    net.verb_classifier[0].p = 0.0
    net.nouns_classifier[0].p = 0.0
    opt = torch.optim.Adamax([p for p in net.parameters() if p.requires_grad], lr=0.002)
    img, verb, nouns = (torch.from_numpy(g[k]).cuda() for k in ("img", "gt_verb", "gt_nouns"))
    opt.zero_grad()
    pv, pn, pg = net(img, verb)
    vl, nl, gl = net.verb_loss(pv, verb), net.nouns_loss(pn, nouns), net.nouns_loss(pg, nouns)
    assert np.abs(pv.detach().cpu().numpy() - g["pred_verb"]).max() < TOL
    assert np.abs(pn.detach().cpu().numpy() - g["pred_nouns"]).max() < TOL
    assert np.abs(pg.detach().cpu().numpy() - g["gt_pred_nouns"]).max() < TOL
    (vl + nl).backward()
    for k, p in net.named_parameters():
        if p.requires_grad:
            ref = g["grad/" + k]
            err = np.abs(p.grad.cpu().numpy() - ref).max()
            assert err <= 1e-3 * max(1e-2, np.abs(ref).max()), (k, err, np.abs(ref).max())
    gn = torch.nn.utils.clip_grad_norm_(net.parameters(), 1)
    opt.step()
    assert abs(float(vl) - float(g["verb_loss"])) < 1e-3
    assert abs(float(nl) - float(g["nouns_loss"])) < 1e-3
    assert abs(float(gl) - float(g["gt_nouns_loss"])) < 1e-3
    assert abs(float(gn) - float(g["grad_norm"])) < 1e-3 * float(g["grad_norm"])
    after = sub(g, "after/")
    sd = net.state_dict()
    assert set(sd) == set(after)
    for k, v in sd.items():
        ref = after[k].float()
        err = float((v.float().cpu() - ref).abs().max())
        assert err <= 1e-3 * max(1.0, float(ref.abs().max())), (k, err)


@pytest.mark.parametrize("steps", [5, 8])
def test_other_step_counts_vs_oracle(sra, steps):
    """T != 4 is not expressible in the reference (model.py:60); checked against the oracle, which is
    pinned to the reference at T=4."""
    g = load("g3_fcggnn_bottleneck.npz")
    net, _ = hip_fcggnn(sra, g, steps=steps)
    ora, _, _ = oracle_fcggnn(g, steps=steps)
    net.eval(); ora.eval()
    img, gv = torch.from_numpy(g["img"]), torch.from_numpy(g["gt_verb"])
    with torch.no_grad():
        a = net(img.cuda(), gv.cuda())
        b = ora(img, gv)
    for x, y in zip(a, b):
        assert float((x.cpu() - y).abs().max()) < TOL


def test_train_mode_with_dropout_mask_vs_oracle(sra):
    """Train-mode forward/backward with Dropout(0.5) active: the HIP path's counter-hash masks are read back
    and injected into the oracle, then logits and gradients must agree."""
    from situation_recognition_amd import ops
    g3, g5 = load("g3_fcggnn_bottleneck.npz"), load("g5_train_step.npz")
    net, _ = hip_fcggnn(sra, g3)
    ora, _, _ = oracle_fcggnn(g3)
    net.train(); ora.train()
    img, verb, nouns = torch.from_numpy(g3["img"]), torch.from_numpy(g3["gt_verb"]), torch.from_numpy(g5["gt_nouns"])
    B, R, D = 5, 4, int(g3["D"])
    net._drop_counter = 0
    seeds = []
    for _ in range(3):
        net._drop_counter += 1
        seeds.append((net.drop_seed_base * 0x9E3779B1 + net._drop_counter * 0x85EBCA77) & (2 ** 63 - 1))
    net._drop_counter = 0
    masks = [ops.dropout_half(torch.ones(n, D, device="cuda"), s, want_mask=True)[1].float().cpu() * 2
             for n, s in zip((B, B * R, B * R), seeds)]

    class Inject(torch.nn.Module):
        def __init__(self, queue): super().__init__(); self.queue = queue
        def forward(self, x): return x * self.queue.pop(0)
    q_v, q_n = [masks[0]], [masks[1], masks[2]]
    ora.verb_classifier[0] = Inject(q_v)
    ora.nouns_classifier[0] = Inject(q_n)
    pv, pn, pg = net(img.cuda(), verb.cuda())
    ov, on, og = ora(img, verb)
    for x, y in ((pv, ov), (pn, on), (pg, og)):
        assert float((x.detach().cpu() - y.detach()).abs().max()) < TOL
    (net.verb_loss(pv, verb.cuda()) + net.nouns_loss(pn, nouns.cuda())).backward()
    (ora.verb_loss(ov, verb) + ora.nouns_loss(on, nouns)).backward()
    ref = dict(ora.named_parameters())
    for k, p in net.named_parameters():
        if p.requires_grad:
            r = ref[k].grad
            assert float((p.grad.cpu() - r).abs().max()) <= 1e-3 * max(1e-2, float(r.abs().max())), k


def test_overfits_the_reference_fixture(sra):
    """sr.py's overfitting recipe (README: `--train_file overfitting.json`): 40 Adamax steps on the 5 annotated images must drive
    the loss down -- exercises forward, hand-written backward, clipping and the optimizer together, in bf16 AND fp32."""
    m, Enc = sra
    enc = Enc(overfitting_json(), quiet=True)
    g = torch.Generator().manual_seed(0)
    img = torch.randn(5, 3, 64, 64, generator=g).clamp_(-2.2, 2.7).cuda()
    verb = torch.arange(5).cuda()
    nouns = torch.stack([enc.encode(a)[1] for a in overfitting_json().values()]).cuda()
    for dtype in (torch.float32, torch.bfloat16):
        torch.manual_seed(1)
        net = m.FCGGNN(enc, 128, steps=4, backbone=18, dtype=dtype, width=16, blocks=(1, 1, 1, 1)).cuda()
        net.train()
        net.verb_classifier[0].p = 0.0
        net.nouns_classifier[0].p = 0.0
        params = [p for p in net.parameters() if p.requires_grad]
        opt = torch.optim.Adamax(params, lr=0.01)
        losses = []
        for _ in range(40):
            opt.zero_grad()
            pv, pn, _ = net(img, verb)
            loss = net.verb_loss(pv, verb) + net.nouns_loss(pn, nouns)
            loss.backward()
            torch.nn.utils.clip_grad_norm_(params, 1)
            opt.step()
            losses.append(float(loss))
        assert all(l == l for l in losses)                       # no NaN
        assert losses[-1] < 0.35 * losses[0], (dtype, losses[0], losses[-1])
        assert int(pv.argmax(1).eq(verb).sum()) == 5             # the five verbs are memorised


def test_uint8_nhwc_input_equals_normalised_fp32_input(sra):
    """Decoded uint8 images in (fused ToTensor+Normalize+layout kernel) == the reference's fp32 NCHW tensors in."""
    g3 = load("g3_fcggnn_basic.npz")
    net, _ = hip_fcggnn(sra, g3)
    net.eval()
    gen = torch.Generator().manual_seed(3)
    u8 = torch.randint(0, 256, (5, 64, 64, 3), generator=gen, dtype=torch.uint8)
    mean, std = torch.tensor([0.485, 0.456, 0.406]), torch.tensor([0.229, 0.224, 0.225])
    f32 = ((u8.float() / 255.0 - mean) / std).permute(0, 3, 1, 2).contiguous()
    verb = torch.arange(5).cuda()
    with torch.no_grad():
        a = net(u8.cuda(), verb)
        b = net(f32.cuda(), verb)
    for x, y in zip(a, b):
        assert float((x - y).abs().max()) < 1e-4


def test_two_stream_backbones_give_identical_results(sra):
    """FCGGNN.forward with the noun backbone on a second stream (the default for per-GPU batches up to 4096) against the
    single-stream order: same kernels on the same data -> bit-identical outputs and running statistics, equal gradients."""
    import copy
    m, Enc = sra
    enc = Enc.synthetic(V=12, NR=9, L=40, R=4)
    torch.manual_seed(5)
    a = m.FCGGNN(enc, 512, steps=3, backbone=50, width=16, dtype=torch.bfloat16).cuda().train()
    b = copy.deepcopy(a)
    a.overlap_backbones, b.overlap_backbones = True, False
    a.drop_seed_base = b.drop_seed_base = 77
    img = torch.randn(6, 3, 96, 96, device="cuda")
    verb = torch.randint(0, 12, (6,), device="cuda")
    nouns = torch.randint(0, 40, (6, 3, 4), device="cuda")
    outs = []
    for net in (a, b):
        pv, pn, pg = net(img, verb)
        loss = net.verb_loss(pv, verb) + net.nouns_loss(pn, nouns)
        loss.backward()
        torch.cuda.synchronize()
        outs.append((pv, pn, pg))
    for x, y in zip(*outs):
        assert torch.equal(x, y)
    sa, sb = a.state_dict(), b.state_dict()
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k
    for (k, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
        if p.requires_grad:
            assert torch.allclose(p.grad, q.grad, rtol=1e-4, atol=1e-6 * float(q.grad.abs().max())), k
            if "emb" in k:       # sr_node_init_bwd sums in a fixed order (no atomics): bit-reproducible
                assert torch.equal(p.grad, q.grad), k


def test_identical_backbones_share_one_train_pass(sra):
    """The reference loads the SAME pretrained weights into both backbones and freezes them (model.py:16-18,100-101); train-mode
    BatchNorm ignores running statistics, so one pass serves both.  Shared pass vs separate passes: bit-identical logits,
    gradients and BatchNorm buffers (verbs: one update; nouns: two, model.py:176-178); and both against the oracle, whose
    three passes are executed literally."""
    import copy
    g3, g5 = load("g3_fcggnn_bottleneck.npz"), load("g5_train_step.npz")
    net, _ = hip_fcggnn(sra, g3)
    ora, _, _ = oracle_fcggnn(g3)
    net.convnet_nouns.load_state_dict(net.convnet_verbs.state_dict())
    ora.convnet_nouns.load_state_dict(ora.convnet_verbs.state_dict())
    with torch.no_grad():                                  # different running statistics must not matter (train mode) nor be mixed up
        for bn in (m_ for m_ in net.convnet_nouns.modules() if isinstance(m_, torch.nn.BatchNorm2d)):
            bn.running_mean.add_(0.25); bn.running_var.mul_(1.5)
        for bn in (m_ for m_ in ora.convnet_nouns.modules() if isinstance(m_, torch.nn.BatchNorm2d)):
            bn.running_mean.add_(0.25); bn.running_var.mul_(1.5)
    assert net.convnet_verbs.weights_equal(net.convnet_nouns)
    shared, separate = net, copy.deepcopy(net)
    separate.share_identical_backbones = False
    img, verb, nouns = torch.from_numpy(g3["img"]), torch.from_numpy(g3["gt_verb"]), torch.from_numpy(g5["gt_nouns"])
    outs = []
    for m_ in (shared, separate, ora):
        m_.train()
        m_.verb_classifier[0].p = 0.0
        m_.nouns_classifier[0].p = 0.0
        dev = "cpu" if m_ is ora else "cuda"
        pv, pn, pg = m_(img.to(dev), verb.to(dev))
        (m_.verb_loss(pv, verb.to(dev)) + m_.nouns_loss(pn, nouns.to(dev))).backward()
        outs.append((pv, pn, pg))
    calls = []
    orig = shared.convnet_nouns._forward_impl
    shared.convnet_nouns._forward_impl = lambda *a, **k: calls.append(1) or orig(*a, **k)
    shared(img.cuda(), verb.cuda())                        # (second step: the noun backbone itself never runs)
    assert not calls
    for x, y, z in zip(*outs):
        assert torch.equal(x, y)
        assert float((x.detach().cpu() - z.detach()).abs().max()) < TOL
    sa, sb, so = shared.state_dict(), separate.state_dict(), ora.state_dict()
    for (k, p), (_, q) in zip(shared.named_parameters(), separate.named_parameters()):
        if p.requires_grad:
            assert torch.equal(p.grad, q.grad), k
    with torch.no_grad():
        separate(img.cuda(), verb.cuda())
        ora(img, verb)
    sa, sb, so = shared.state_dict(), separate.state_dict(), ora.state_dict()
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k
        if "running" in k or "num_batches" in k:
            assert float((sa[k].float().cpu() - so[k].float()).abs().max()) <= 1e-3 * max(1.0, float(so[k].float().abs().max())), k
    assert int(sa["convnet_nouns.model.bn1.num_batches_tracked"]) == 4 and int(sa["convnet_verbs.model.bn1.num_batches_tracked"]) == 2
    # diverging weights switch the sharing off again
    with torch.no_grad():
        shared.convnet_nouns.model.layer1[0].conv1.weight.mul_(1.01)
    assert not shared.convnet_verbs.weights_equal(shared.convnet_nouns)


@pytest.mark.parametrize("width", [64, 16])
def test_train_mode_graph_replay_equals_eager(sra, width):
    """FCGGNN.enable_graphs(train=True): the frozen backbones' train-mode passes replayed from captured hipGraphs (the first step
    runs eagerly and captures, later steps REPLAY -- asserted by the replay counter: the graph key must not move when the pass
    updates the running statistics, neither in a real-width net, whose buffers the finalize kernel updates in place, nor in a
    channel-padded one, whose buffers are written with copy_) against the eager model over three steps with different images:
    logits and every BatchNorm buffer (running statistics: the verb backbone's one update per step, the noun backbone's two;
    num_batches_tracked) bit-identical, gradients equal."""
    import copy
    m, Enc = sra
    enc = Enc.synthetic(V=12, NR=9, L=40, R=4)
    torch.manual_seed(9)
    a = m.FCGGNN(enc, 32 * width, steps=2, backbone=50, width=width, dtype=torch.bfloat16).cuda().train()
    b = copy.deepcopy(a)
    a.enable_graphs(True, train=True)
    a.drop_seed_base = b.drop_seed_base = 31
    for step in range(3):
        img = torch.randn(6, 3, 96, 96, device="cuda")
        verb = torch.randint(0, 12, (6,), device="cuda")
        nouns = torch.randint(0, 40, (6, 3, 4), device="cuda")
        outs = []
        for net in (a, b):
            net.zero_grad()
            pv, pn, pg = net(img, verb)
            (net.verb_loss(pv, verb) + net.nouns_loss(pn, nouns)).backward()
            torch.cuda.synchronize()
            outs.append((pv, pn, pg))
        for x, y in zip(*outs):
            assert torch.equal(x, y), step
    for net in (a.convnet_verbs, a.convnet_nouns):
        assert sum(k[0] == "train" for k in net._graphs) == 1           # ONE capture: the key is stable across the passes
        assert net.graph_replays == 2                                   # step 0 ran eagerly and captured, steps 1 and 2 replayed
    sa, sb = a.state_dict(), b.state_dict()
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k
    assert int(sa["convnet_verbs.model.bn1.num_batches_tracked"]) == 3 and int(sa["convnet_nouns.model.bn1.num_batches_tracked"]) == 6
    for (k, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
        if p.requires_grad:
            assert torch.allclose(p.grad, q.grad, rtol=1e-4, atol=1e-6 * float(q.grad.abs().max())), k
