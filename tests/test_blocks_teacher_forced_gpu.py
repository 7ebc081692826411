"""Teacher-forced per-block parity of the bf16 ResNet-152 backbone at production-size launches.

The end-to-end comparison of a 152-layer bf16 net with the fp32 oracle cannot carry a tight tolerance: a randomly initialised
residual net amplifies the rounding of every activation (tests/test_full_configs_gpu.py reports those numbers, it does not
gate on them).  Here every residual block (all 50 bottlenecks of ResNet-152, plus the stem and the pooling head) is run ALONE,
through exactly the launches the full pass makes for it (`resnet.block_forward`), on a batch large enough that every convolution
is dispatched to the kernel and tile shape the benchmark batch (6144) uses -- asserted launch by launch with `sr_conv_route` --
and is fed the ORACLE's input for that block, so no error is carried from one block to the next:

  eval mode   the first 8 images of the batch are the fp32 oracle's block input (reference model.py:35 -> torchvision
              Bottleneck, restated in oracle/ref_resnet.py) rounded to bf16; eval-mode BatchNorm keeps images independent, so
              their block output must equal the oracle's block output on those 8 images: 1.2e-2 of the output range (bf16
              rounding of the three intermediate tensors), the tolerance of the single-kernel tests.
  train mode  the same batch through the train-mode launches (batch statistics: statistics kernels, Gram-matrix statistics,
              BatchNorm-on-load, running-statistics update) against an fp32 convolution / batch-norm chain of the oracle block's
              weights on the device, over the whole batch, and the updated running statistics against the fp32 batch statistics.
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
PROD_B = 6144                      # BASELINE config 3 (per-GPU batch of the headline benchmark)
TOL = 1.2e-2


@pytest.fixture(scope="module")
def world():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from oracle.ref_model import RefBackbone
    from oracle.ref_resnet import calibrate_batchnorm_, perturb_batchnorm_
    from situation_recognition_amd import ops
    from situation_recognition_amd.model import resnet
    ops.lib()
    torch.manual_seed(11)
    torch.set_num_threads(max(torch.get_num_threads(), 16))
    ora = RefBackbone(152)
    perturb_batchnorm_(ora, 3)
    g = torch.Generator().manual_seed(12)
    calibrate_batchnorm_(ora, torch.randn(16, 3, 224, 224, generator=g).clamp_(-2.2, 2.7))
    ora.eval()
    net = resnet(None, depth=152, dtype=BF)
    net.model.load_state_dict(ora.model.state_dict(), strict=True)
    net.cuda()
    # the oracle's own chain on 8 images, every block's input and output kept (fp32, NCHW)
    img = torch.randn(8, 3, 224, 224, generator=g).clamp_(-2.2, 2.7)
    m = ora.model
    acts = []
    with torch.no_grad():
        a = F.max_pool2d(F.relu(m.bn1(m.conv1(img))), 3, stride=2, padding=1)
        stem_out = a
        blocks = [b for s in range(4) for b in getattr(m, "layer%d" % (s + 1))]
        for blk in blocks:
            out = blk(a)
            acts.append((a, out))
            a = out
        feat = torch.flatten(F.adaptive_avg_pool2d(a, 1), 1)
    return dict(ops=ops, net=net, ora=ora, img=img, stem_out=stem_out, blocks=blocks, acts=acts, feat=feat)


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def batch_from(x8_nhwc, B, seed):
    """[B,H,W,C] bf16 whose first 8 images are x8 (rounded to bf16); the others are the same eight images rescaled per image and
    per channel (realistic post-ReLU activation statistics, every image different)."""
    g = torch.Generator(device="cuda").manual_seed(seed)
    x8 = x8_nhwc.cuda()
    reps = (B + 7) // 8
    out = x8.repeat(reps, 1, 1, 1)[:B].clone()
    s = 0.6 + 0.8 * torch.rand(B, 1, 1, x8.shape[3], device="cuda", generator=g)
    s[:8] = 1.0
    return (out * s).to(BF)


def stage_batch(bi, train=False):
    """layers 1-3 at batch 1024 (3.2 M / 0.8 M / 0.2 M output rows: >= 3 row tiles per workgroup everywhere); layer4 (49 pixels per
    image) at 4096, so that its launches have as many rows as layer3's.  (Train-mode layer1 at 512: its fp32 reference tensors
    are 3.2 GB each at 1024.)"""
    if bi >= 3 + 8 + 36:
        return 4096
    return 512 if (train and bi < 3) else 1024


def block_launches(blk, H):
    """(Cin, Cout, k, stride, H_in, kwargs) of the conv launches `resnet._unit` makes for a bottleneck in eval and train mode."""
    planes, cin, cout = blk.conv1.out_channels, blk.conv1.in_channels, blk.conv3.out_channels
    st = blk.conv2.stride[0]
    Ho = (H - 1) // st + 1
    ev = [(cin, planes, 1, 1, H, dict(bias=True, relu=True)),
          (planes, planes, 3, st, H, dict(bias=True, relu=True)),
          (planes, cout, 1, 1, Ho, dict(bias=True, relu=True, res=True))]
    tr = [(cin, planes, 1, 1, H, dict(want_stats=True)),
          (planes, planes, 3, st, H, dict(want_stats=True)),
          (planes, cout, 1, 1, Ho, dict(bias=True, escale=True, relu=True, res=True))]
    if blk.downsample is not None:
        ev.append((cin, cout, 1, st, H, dict(bias=True)))
        tr.append((cin, cout, 1, st, H, dict(bias=True, escale=True)))
    return ev, tr


def test_every_block_is_dispatched_as_at_the_benchmark_batch(world):
    """The premise of this file: at the test batches every convolution launch of every block takes the kernel (and tile shape) it
    takes at batch 6144."""
    ops = world["ops"]
    seen = set()
    for bi, blk in enumerate(world["blocks"]):
        H = world["acts"][bi][0].shape[2]
        for train, mode in enumerate(block_launches(blk, H)):
            B = stage_batch(bi, bool(train))
            for cin, cout, k, st, h, kw in mode:
                pad = 1 if k == 3 else 0
                got = ops.conv_route(B, h, h, cin, cout, k, st, pad, **kw)
                want = ops.conv_route(PROD_B, h, h, cin, cout, k, st, pad, **kw)
                assert got == want, (bi, cin, cout, k, st, kw, got, want)
                seen.add(got)
    from situation_recognition_amd import _lib
    assert {1, 2, 4, _lib.ROUTE_WS}.issubset(seen), seen           # narrow tiles, 256x256 tiles and the weight-stationary kernel


def test_stem_against_the_oracle(world):
    """conv1 7x7/2 -> bn1 -> relu -> maxpool 3x3/2 (one fused launch in eval mode) on 1024 images whose first 8 are the oracle's."""
    net, img = world["net"], world["img"]
    g = torch.Generator(device="cuda").manual_seed(5)
    batch = torch.randn(1024, 3, 224, 224, device="cuda", generator=g).clamp_(-2.2, 2.7)
    batch[:8] = img.cuda()
    net.eval()
    got = net.stem_forward(batch)[:8].float().cpu()
    want = nhwc(world["stem_out"])
    err = float((got - want).abs().max())
    assert err <= TOL * float(want.abs().max()), (err, float(want.abs().max()))


def test_all_50_bottlenecks_eval_mode_against_the_oracle(world):
    net, acts = world["net"], world["acts"]
    net.eval()
    worst = []
    for bi, (xin, want) in enumerate(acts):
        x = batch_from(nhwc(xin), stage_batch(bi), 100 + bi)
        y = net.block_forward(x, bi)
        got = y[:8].float().cpu()
        ref = nhwc(want)
        assert tuple(got.shape) == tuple(ref.shape), bi
        err, rng = float((got - ref).abs().max()), float(ref.abs().max())
        worst.append(err / rng)
        assert err <= TOL * rng, "block %d: max err %.4g, output range %.4g" % (bi, err, rng)
        # the rest of the batch must be finite and the first 8 must not depend on it (eval BatchNorm: images are independent)
        assert torch.isfinite(y.float()).all(), bi
        del x, y
    print("teacher-forced eval: worst block error %.4f of the output range (block %d)" % (max(worst), worst.index(max(worst))))
    # pooling head: global average of the last block's oracle output
    a = batch_from(nhwc(acts[-1][1]), 1024, 999)
    feat = world["ops"].avgpool(a)[:8].float().cpu()
    want = world["feat"]
    assert float((feat - want).abs().max()) <= TOL * float(want.abs().max())


def conv_fp32(x2d_nhwc, w, stride):
    """fp32 convolution as explicit shifted fp32 matmuls on the device (no library convolution: nothing to JIT on a fresh box).
    x [B,H,W,C] fp32, w [Cout,Cin,k,k] fp32 -> [B,Ho,Wo,Cout] fp32; k = 1 (pad 0) or 3 (pad 1)."""
    B, H, W_, Cc = x2d_nhwc.shape
    k = w.shape[2]
    Ho, Wo = (H - 1) // stride + 1, (W_ - 1) // stride + 1
    if k == 1:
        xs = x2d_nhwc[:, ::stride, ::stride, :].reshape(-1, Cc)
        return (xs @ w.view(w.shape[0], Cc).t()).view(B, Ho, Wo, -1)
    xp = F.pad(x2d_nhwc, (0, 0, 1, 1, 1, 1))
    out = torch.zeros(B * Ho * Wo, w.shape[0], device=x2d_nhwc.device)
    for r in range(3):
        for q in range(3):
            xs = xp[:, r:r + (Ho - 1) * stride + 1:stride, q:q + (Wo - 1) * stride + 1:stride, :].reshape(-1, Cc)
            out += xs @ w[:, :, r, q].t()
    return out.view(B, Ho, Wo, -1)


def bn_train_fp32(y, bn):
    """Train-mode BatchNorm of y [B,H,W,C] in fp32 (fp64 statistics); returns (normalised, batch mean, unbiased batch variance)."""
    Cc = y.shape[3]
    flat = y.reshape(-1, Cc)
    mean = flat.double().mean(0)
    var = (flat.double() - mean).pow(2).mean(0)
    n = flat.shape[0]
    scale = bn.weight.double().cuda() * torch.rsqrt(var + bn.eps)
    out = (y.double() - mean) * scale + bn.bias.double().cuda()
    return out.float(), mean.float(), (var * n / (n - 1)).float()


def test_all_50_bottlenecks_train_mode_against_fp32_chain(world):
    """Train-mode launches (the benchmark's): per block conv1 + statistics, [bn1+relu sweep or on-load], conv2 + statistics,
    bn2-relu fused with the Gram statistics of conv3 (or on load), conv3 with scale / shift + identity + ReLU in its epilogue,
    stride-2 downsample through a statistics-only launch -- against the fp32 chain over the WHOLE batch, and the running
    statistics every BatchNorm of the block ends with."""
    net, acts, blocks = world["net"], world["acts"], world["blocks"]
    keep = {k: v.clone() for k, v in net.model.state_dict().items()}
    worst = []
    try:
        net.train()
        for bi, (xin, _) in enumerate(acts):
            blk = blocks[bi]
            B = stage_batch(bi, train=True)
            x = batch_from(nhwc(xin), B, 300 + bi)
            y = net.block_forward(x, bi)
            xf = x.float()
            wt = lambda c: c.weight.detach().cuda().to(BF).float()          # the kernels see bf16 weights
            stats = {}
            t, stats["bn1"] = (lambda r: (r[0], r[1:]))(bn_train_fp32(conv_fp32(xf, wt(blk.conv1), 1), blk.bn1))
            t = F.relu(t).to(BF).float()                                    # the tensor conv2 multiplies is bf16
            t, stats["bn2"] = (lambda r: (r[0], r[1:]))(bn_train_fp32(conv_fp32(t, wt(blk.conv2), blk.conv2.stride[0]), blk.bn2))
            t = F.relu(t).to(BF).float()
            t, stats["bn3"] = (lambda r: (r[0], r[1:]))(bn_train_fp32(conv_fp32(t, wt(blk.conv3), 1), blk.bn3))
            idn = xf
            if blk.downsample is not None:
                idn, stats["downsample.1"] = (lambda r: (r[0], r[1:]))(
                    bn_train_fp32(conv_fp32(xf, wt(blk.downsample[0]), blk.downsample[0].stride[0]), blk.downsample[1]))
                idn = idn.to(BF).float()                                    # the identity tensor is stored in bf16
            ref = F.relu(t + idn)
            err, rng = float((y.float() - ref).abs().max()), float(ref.abs().max())
            worst.append(err / rng)
            # three chained bf16 tensors inside the block, each normalised by statistics of ~1e6 values: twice the single-kernel bound
            assert err <= 2 * TOL * rng, "block %d (train): max err %.4g, output range %.4g" % (bi, err, rng)
            hip_blk = [b for s in range(4) for b in getattr(net.model, "layer%d" % (s + 1))][bi]
            for bn_name, (mean, var) in stats.items():
                hb = hip_blk
                for part in bn_name.split("."):
                    hb = hb[int(part)] if part.isdigit() else getattr(hb, part)
                ob = blk
                for part in bn_name.split("."):
                    ob = ob[int(part)] if part.isdigit() else getattr(ob, part)
                want_m = 0.9 * ob.running_mean.cuda() + 0.1 * mean
                want_v = 0.9 * ob.running_var.cuda() + 0.1 * var
                em = float((hb.running_mean - want_m).abs().max() / want_m.abs().max().clamp_min(1e-3))
                evr = float(((hb.running_var - want_v).abs() / want_v.abs().clamp_min(1e-6)).max())
                assert em < 3e-3 and evr < 3e-3, (bi, bn_name, em, evr)
                assert int(hb.num_batches_tracked) == int(ob.num_batches_tracked) + 1
            del x, y, xf, t, idn, ref
        print("teacher-forced train: worst block error %.4f of the output range (block %d)" % (max(worst), worst.index(max(worst))))
    finally:
        net.model.load_state_dict(keep)
        net.eval()


def test_expansion_fused_with_the_next_reduce_conv(world):
    """The train-mode pass runs the expansion conv of every block of layers 1-3 FUSED with the next block's conv1 (`sr_conv_pair`), also
    across the layer boundaries 1 -> 2 and 2 -> 3, where that conv1 has twice the mid channels (3 + 8 + 35 pairs per pass; layer3's last
    block is not served: 512 reduce outputs).  Every such pair on the oracle's block input (teacher forced, the
    batches of the test above): the block output must be BIT-IDENTICAL to the unfused launches' (which that test holds to the fp32
    chain), the raw conv1 output of the next block bit-identical to the generic kernel fed that output, its statistics equal up to
    summation order, and the next block, continued from the fused launch's tensors, equal to the same block run from scratch."""
    net, acts, ops = world["net"], world["acts"], world["ops"]
    keep = {k: v.clone() for k, v in net.model.state_dict().items()}
    layer_of = [0] * 3 + [1] * 8 + [2] * 36 + [3] * 3
    fused = [0, 0, 0, 0]
    try:
        net.train()
        for bi in range(len(layer_of) - 1):
            x = batch_from(nhwc(acts[bi][0]), stage_batch(bi, train=True), 700 + bi)
            y, pre = net.block_forward(x, bi, fuse_next=True)
            if layer_of[bi] == 3 or layer_of[bi + 1] == 3:
                # layer4 (512 mid channels) and layer3's last block (512 reduce outputs) are not served: not fused
                assert pre is None, bi
                del x, y
                continue
            assert pre is not None, "block %d: the fused route was not taken" % bi
            fused[layer_of[bi]] += 1
            y_ref = net.block_forward(x, bi)                                    # unfused launches
            assert torch.equal(y.view(torch.int16), y_ref.view(torch.int16)), bi
            nxt = net._plan()[1][bi + 1][0][0]
            w1 = nxt.raw(BF)[0]
            y1_ref, st_ref = ops.conv2d(y_ref, w1, nxt.cout_p, 1, 1, 0, want_stats=True)
            assert torch.equal(pre[0].view(torch.int16), y1_ref.view(torch.int16)), bi
            s0, s1 = st_ref.double().sum(0), pre[1].double().sum(0)
            assert float((s0 - s1).abs().max() / s0.abs().max()) < 1e-6, bi
            if bi % 6 == 0 or layer_of[bi] < 2:                                 # the consumer side: block bi + 1 continued from `pre`
                z_ref = net.block_forward(y_ref, bi + 1)
                z = net.block_forward(y, bi + 1, pre=pre)
                err = float((z.float() - z_ref.float()).abs().max())
                assert err <= 2e-2 * float(z_ref.float().abs().max()), (bi, err)   # (scale / shift of bn1 from sums in another order)
                del z, z_ref
            del x, y, y_ref, pre, y1_ref
        assert fused == [3, 8, 35, 0], fused
    finally:
        net.model.load_state_dict(keep)
        net.eval()


def test_eval_mode_expansion_fused_with_the_next_reduce_conv(world):
    """Eval mode takes the fused launch too (folded BatchNorms; `sr_conv_pair` with a bias + ReLU epilogue on the reduce conv and no
    statistics): for every pair of layers 1-3 the block output, the next block's conv1 output and the next block continued from it are
    BIT-IDENTICAL to the unfused launches (which the eval test above holds to the oracle)."""
    net, acts, ops = world["net"], world["acts"], world["ops"]
    net.eval()
    layer_of = [0] * 3 + [1] * 8 + [2] * 36 + [3] * 3
    fused = 0
    for bi in range(len(layer_of) - 1):
        x = batch_from(nhwc(acts[bi][0]), stage_batch(bi), 900 + bi)
        y, pre = net.block_forward(x, bi, fuse_next=True)
        if layer_of[bi] == 3 or layer_of[bi + 1] == 3:
            assert pre is None, bi
            continue
        assert pre is not None and pre[1] is None, bi
        fused += 1
        y_ref = net.block_forward(x, bi)
        assert torch.equal(y.view(torch.int16), y_ref.view(torch.int16)), bi
        if bi % 5 == 0 or layer_of[bi] < 2:
            z_ref = net.block_forward(y_ref, bi + 1)
            z = net.block_forward(y, bi + 1, pre=pre)
            assert torch.equal(z.view(torch.int16), z_ref.view(torch.int16)), bi
            del z, z_ref
        del x, y, y_ref, pre
    assert fused == 46
