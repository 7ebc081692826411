"""Kernel-level parity (through the C ABI) against plain fp32 CPU torch ops.
fp32 storage: tight tolerances.  bf16 storage: inputs are rounded to bf16 first and the
fp32 reference is computed from the rounded values; tolerance covers bf16 output rounding."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DT = [torch.float32, torch.bfloat16]


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from situation_recognition_amd import ops as o
    o.lib()
    return o


def rnd(*shape, dtype=torch.float32, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype)


def tol(dtype, ref, k=1.0):
    m = float(ref.abs().max()) + 1e-6
    return (2e-5 if dtype == torch.float32 else 1.2e-2) * m * k


def close(got, ref, dtype, k=1.0):
    got = got.float().cpu()
    err = float((got - ref).abs().max())
    assert err <= tol(dtype, ref, k), "max err %g > tol %g" % (err, tol(dtype, ref, k))


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 200, 128), (1000, 72, 320), (64, 504, 512), (37, 40, 64)])
def test_gemm_plain(ops, dtype, M, N, K):
    A, W, b = rnd(M, K, dtype=dtype, seed=1), rnd(N, K, dtype=dtype, seed=2), rnd(N, seed=3)
    ref = A.float() @ W.float().t() + 2.0 * b
    out = ops.gemm([(A.cuda(), W.cuda())], bias=b.cuda(), bias_scale=2.0)
    assert out.shape == (M, N)
    close(out, ref, dtype)


@pytest.mark.parametrize("dtype", DT)
def test_gemm_pairs_res_acts_and_f32_out(ops, dtype):
    M, N = 200, 136
    A1, A2, A3 = rnd(M, 64, dtype=dtype, seed=1), rnd(M, 128, dtype=dtype, seed=2), rnd(M, 192, dtype=dtype, seed=3)
    W1, W2, W3 = rnd(N, 64, dtype=dtype, seed=4, scale=.1), rnd(N, 128, dtype=dtype, seed=5, scale=.1), rnd(N, 192, dtype=dtype, seed=6, scale=.1)
    b1, b2 = rnd(N, seed=7), rnd(N, seed=8)
    pre = A1.float() @ W1.float().t() + A2.float() @ W2.float().t() + A3.float() @ W3.float().t() + b1 + b2
    g = lambda t: t.cuda()
    pairs = [(g(A1), g(W1)), (g(A2), g(W2)), (g(A3), g(W3))]
    for act, fn in ((ops.ACT_NONE, lambda x: x), (ops.ACT_RELU, torch.relu), (ops.ACT_SIGMOID, torch.sigmoid), (ops.ACT_TANH, torch.tanh)):
        close(ops.gemm(pairs, bias=g(b1), bias2=g(b2), act=act), fn(pre), dtype)
    res = rnd(M, N, dtype=dtype, seed=9)
    close(ops.gemm(pairs, bias=g(b1), bias2=g(b2), act=ops.ACT_RELU, res=g(res)), torch.relu(pre + res.float()), dtype)
    out = ops.gemm(pairs, bias=g(b1), bias2=g(b2), out_f32=True)
    assert out.dtype == torch.float32
    close(out, pre, torch.float32, k=1.0 if dtype == torch.float32 else 1.0)
    # strided A (a column block of a wider matrix) and an accumulate-into-fp32 residual
    wide = rnd(M, 256, dtype=dtype, seed=10)
    acc = rnd(M, N, seed=11)
    out = ops.gemm([(g(wide)[:, 64:128], g(W1))], res=g(acc), out_f32=True)
    close(out, wide[:, 64:128].float() @ W1.float().t() + acc, torch.float32)


@pytest.mark.parametrize("dtype", DT)
def test_gemm_gru_epilogues(ops, dtype):
    M, N, K = 150, 128, 128
    A, W, b = rnd(M, K, dtype=dtype, seed=1), rnd(N, K, dtype=dtype, seed=2, scale=.2), rnd(N, seed=3)
    h, z = rnd(M, N, dtype=dtype, seed=4), torch.rand(M, N, generator=torch.Generator().manual_seed(5)).to(dtype)
    pre = A.float() @ W.float().t() + b
    r, rh = ops.gemm([(A.cuda(), W.cuda())], bias=b.cuda(), act=ops.ACT_SIGMOID_MUL, aux1=h.cuda())
    close(r, torch.sigmoid(pre), dtype)
    close(rh, torch.sigmoid(pre) * h.float(), dtype)
    hn, c = ops.gemm([(A.cuda(), W.cuda())], bias=b.cuda(), act=ops.ACT_TANH_BLEND, aux1=h.cuda(), aux2=z.cuda())
    close(c, torch.tanh(pre), dtype)
    close(hn, (1 - z.float()) * h.float() + z.float() * torch.tanh(pre), dtype)


def pack_w(w):  # [Cout,Cin,KH,KW] -> [Cout, KH*KW*Cin]
    return w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).contiguous()


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("B,H,W,Cin,Cout,k,s,p", [
    (3, 14, 14, 64, 128, 1, 1, 0), (2, 14, 14, 128, 64, 1, 2, 0), (2, 12, 10, 64, 64, 3, 1, 1),
    (3, 15, 15, 64, 128, 3, 2, 1), (2, 8, 8, 128, 256, 3, 1, 1), (1, 7, 7, 256, 64, 3, 1, 1)])
def test_conv2d(ops, dtype, B, H, W, Cin, Cout, k, s, p):
    x = rnd(B, Cin, H, W, dtype=dtype, seed=1)
    w = rnd(Cout, Cin, k, k, dtype=dtype, seed=2, scale=(Cin * k * k) ** -0.5)
    ref = F.conv2d(x.float(), w.float(), stride=s, padding=p)            # NCHW
    xh = x.permute(0, 2, 3, 1).contiguous().cuda()
    y, stats = ops.conv2d(xh, pack_w(w).cuda(), Cout, k, s, p, want_stats=True)
    close(y.permute(0, 3, 1, 2), ref, dtype)
    n = ref.numel() // Cout
    s1 = stats[:, 0].sum(0).cpu()
    s2 = stats[:, 1].sum(0).cpu()
    assert (s1 / n - ref.mean((0, 2, 3))).abs().max() < 1e-3 * (1 + ref.abs().max())
    assert (s2 / n - (ref ** 2).mean((0, 2, 3))).abs().max() < 2e-3 * (1 + (ref ** 2).max())
    # fused epilogue: bias + residual + relu
    bias, res = rnd(Cout, seed=3), rnd(*ref.shape, dtype=dtype, seed=4)
    y2 = ops.conv2d(xh, pack_w(w).cuda(), Cout, k, s, p, bias=bias.cuda(), res=res.permute(0, 2, 3, 1).contiguous().cuda(), relu=True)
    close(y2.permute(0, 3, 1, 2), torch.relu(ref + bias.view(1, -1, 1, 1) + res.float()), dtype)


def pack_stem(w):   # [Cout,3,7,7] -> [Cout, 8, 32]: [co][r][q*4+c]
    Cout = w.shape[0]
    out = torch.zeros(Cout, 8, 8, 4, dtype=w.dtype)
    out[:, :7, :7, :3] = w.permute(0, 2, 3, 1)
    return out.reshape(Cout, 256).contiguous()


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("B,H,W", [(2, 64, 64), (1, 37, 51), (2, 224, 224)])
def test_stem_conv(ops, dtype, B, H, W):
    img = rnd(B, 3, H, W, seed=1)
    w = rnd(64, 3, 7, 7, dtype=dtype, seed=2, scale=0.1)
    xp = ops.stem_prep(img.cuda(), dtype)
    assert xp.shape == (B, (H + 7) & ~1, (W + 7) & ~1, 4)
    ref_in = img.to(dtype).float()
    assert torch.equal(xp[:, 3:3 + H, 3:3 + W, :3].float().cpu(), ref_in.permute(0, 2, 3, 1))
    assert float(xp[:, :3].abs().max()) == 0 and float(xp[..., 3].abs().max()) == 0
    ref = F.conv2d(ref_in, w.float(), stride=2, padding=3)
    y = ops.conv2d(xp, pack_stem(w).cuda(), 64, 7, 2, 3, stem_hw=(H, W))
    close(y.permute(0, 3, 1, 2), ref, dtype)
    # train-mode form (raw output + BatchNorm partial sums; bf16: the direct-convolution kernel of csrc/stem.hip, whose edge
    # tiles at 37 x 51 are partly outside the image) and eval-mode form (folded bias + ReLU)
    y2, stats = ops.conv2d(xp, pack_stem(w).cuda(), 64, 7, 2, 3, stem_hw=(H, W), want_stats=True)
    assert torch.equal(y2, y)
    s1, s2 = stats[:, 0].double().sum(0).cpu(), stats[:, 1].double().sum(0).cpu()
    assert float((s1 - ref.double().sum((0, 2, 3))).abs().max()) <= 1e-4 * float(ref.abs().sum((0, 2, 3)).max())
    assert float(((s2 - (ref.double() ** 2).sum((0, 2, 3))).abs() / (ref.double() ** 2).sum((0, 2, 3))).max()) <= 1e-4
    bias = rnd(64, seed=5)
    y3 = ops.conv2d(xp, pack_stem(w).cuda(), 64, 7, 2, 3, stem_hw=(H, W), bias=bias.cuda(), relu=True)
    close(y3.permute(0, 3, 1, 2), torch.relu(ref + bias.view(1, -1, 1, 1)), dtype)


@pytest.mark.parametrize("B,H,W", [(3, 64, 64), (2, 37, 51), (2, 224, 224), (5, 100, 60)])
def test_fused_stem_bn_relu_maxpool(ops, B, H, W):
    """csrc/stem.hip `stem_pool_kernel`: conv1 -> scale/shift -> ReLU -> maxpool(3, 2, 1) in one launch (overlapping 16 x 16 conv
    tiles, 7 x 7 pooled pixels each; edge tiles partly outside the image) against the unfused fp32 chain, and the train-mode
    pair (statistics-only launch + finalize + fused launch) against F.batch_norm(training=True)."""
    dtype = torch.bfloat16
    img = rnd(B, 3, H, W, seed=1)
    w = rnd(64, 3, 7, 7, dtype=dtype, seed=2, scale=0.1)
    xp = ops.stem_prep(img.cuda(), dtype)
    conv = F.conv2d(img.to(dtype).float(), w.float(), stride=2, padding=3)
    sc, sh = 0.5 + torch.rand(64), rnd(64, seed=3, scale=0.3)
    ref = F.max_pool2d(F.relu(conv * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)), 3, 2, 1)
    y = ops.stem_bn_relu_maxpool(xp, pack_stem(w).cuda(), sc.cuda(), sh.cuda(), (H, W))
    assert y.shape == (B, ref.shape[2], ref.shape[3], 64)
    close(y.permute(0, 3, 1, 2), ref, dtype)
    gamma, beta = 0.5 + torch.rand(64), rnd(64, seed=4, scale=0.2)
    st = ops.conv2d(xp, pack_stem(w).cuda(), 64, 7, 2, 3, stem_hw=(H, W), stats_only=True)
    scale, shift = ops.bn_finalize(st, conv.numel() // 64, gamma.cuda(), beta.cuda(), None, None, 0.1, 1e-5)
    y2 = ops.stem_bn_relu_maxpool(xp, pack_stem(w).cuda(), scale, shift, (H, W))
    ref2 = F.max_pool2d(F.relu(F.batch_norm(conv, None, None, gamma, beta, training=True, eps=1e-5)), 3, 2, 1)
    close(y2.permute(0, 3, 1, 2), ref2, dtype, k=2.0)


@pytest.mark.parametrize("dtype", DT)
def test_batchnorm_train_two_phase(ops, dtype):
    B, H, W, Cin, Cout = 4, 9, 9, 64, 128
    x = rnd(B, Cin, H, W, dtype=dtype, seed=1)
    w = rnd(Cout, Cin, 3, 3, dtype=dtype, seed=2, scale=0.05)
    gamma, beta = 0.5 + torch.rand(Cout), rnd(Cout, seed=3, scale=0.2)
    rm, rv = rnd(Cout, seed=4, scale=0.1), 0.5 + torch.rand(Cout)
    conv = F.conv2d(x.float(), w.float(), padding=1) + 3.0        # +3: a mean well away from 0
    rm_ref, rv_ref = rm.clone(), rv.clone()
    ref = F.relu(F.batch_norm(conv, rm_ref, rv_ref, gamma, beta, training=True, momentum=0.1, eps=1e-5))
    y, stats = ops.conv2d(x.permute(0, 2, 3, 1).contiguous().cuda(), pack_w(w).cuda(), Cout, 3, 1, 1,
                          bias=torch.full((Cout,), 3.0).cuda(), want_stats=True)
    rm_d, rv_d = rm.cuda(), rv.cuda()
    scale, shift = ops.bn_finalize(stats, B * H * W, gamma.cuda(), beta.cuda(), rm_d, rv_d, 0.1, 1e-5)
    out = ops.bn_apply(y, scale, shift, relu=True)
    close(out.permute(0, 3, 1, 2), ref, dtype, k=2.0)
    assert (rm_d.cpu() - rm_ref).abs().max() < 2e-3
    assert (rv_d.cpu() - rv_ref).abs().max() < 2e-3
    res = rnd(B, H, W, Cout, dtype=dtype, seed=9)
    out2 = ops.bn_apply(y, scale, shift, res=res.cuda(), relu=True)
    ref2 = F.relu(F.batch_norm(conv, None, None, gamma, beta, training=True, eps=1e-5) + res.float().permute(0, 3, 1, 2))
    close(out2.permute(0, 3, 1, 2), ref2, dtype, k=2.0)


@pytest.mark.parametrize("dtype", DT)
def test_batchnorm_train_two_launch_conv(ops, dtype):
    """Output-heavy 1x1 conv: statistics-only launch, finalize, then the conv again with scale/shift (+identity, ReLU)
    in its epilogue -- no raw tensor, no elementwise pass."""
    B, H, W, Cin, Cout = 3, 10, 9, 64, 256
    x = rnd(B, Cin, H, W, dtype=dtype, seed=1)
    w = rnd(Cout, Cin, 1, 1, dtype=dtype, seed=2, scale=0.2)
    idn = rnd(B, H, W, Cout, dtype=dtype, seed=5)
    gamma, beta = 0.5 + torch.rand(Cout), rnd(Cout, seed=3, scale=0.2)
    rm, rv = rnd(Cout, seed=4, scale=0.1), 0.5 + torch.rand(Cout)
    conv = F.conv2d(x.float(), w.float())
    rm_ref, rv_ref = rm.clone(), rv.clone()
    ref = F.relu(F.batch_norm(conv, rm_ref, rv_ref, gamma, beta, training=True, momentum=0.1, eps=1e-5)
                 + idn.float().permute(0, 3, 1, 2))
    xh, wp = x.permute(0, 2, 3, 1).contiguous().cuda(), pack_w(w).cuda()
    st = ops.conv2d(xh, wp, Cout, 1, 1, 0, stats_only=True)
    rm_d, rv_d = rm.cuda(), rv.cuda()
    scale, shift = ops.bn_finalize(st, B * H * W, gamma.cuda(), beta.cuda(), rm_d, rv_d, 0.1, 1e-5)
    y = ops.conv2d(xh, wp, Cout, 1, 1, 0, bias=shift, escale=scale, res=idn.cuda(), relu=True)
    close(y.permute(0, 3, 1, 2), ref, dtype, k=2.0)
    assert (rm_d.cpu() - rm_ref).abs().max() < 2e-3 and (rv_d.cpu() - rv_ref).abs().max() < 2e-3


@pytest.mark.parametrize("M,C", [(980, 64), (70001, 64), (980, 128), (33333, 128), (37, 256), (980, 256), (50000, 256),
                                 (245, 512), (9000, 512)])
def test_gram_partials_sum_to_xtx(ops, M, C):
    """sr_gram: the partials add up to x^T x and colsum(x) (ragged row counts: the last stage of the last slice is
    padded from the zero page); bf16 products are exact in fp32, so only the fp32 accumulation order differs."""
    x = rnd(M, C, dtype=torch.bfloat16, seed=M + C)
    x[:, C - 3] = 0                                        # a padded (all-zero) channel
    part = ops.gram(x.cuda())
    n, f = ops.gram_plan(M, C)
    assert part.shape == (n, f) and f == C * C + C
    tot = part.double().sum(0).cpu()
    xd = x.double()
    G = xd.t() @ xd
    tol = 1e-5 * float(G.diagonal().max())
    valid = ops.gram_valid_mask(C)                        # (C <= 256: the blocks on or below the block diagonal; the rest is mirrored later)
    got = tot[: C * C].view(C, C)
    assert (got - G)[valid].abs().max() < tol
    assert (tot[C * C:] - xd.sum(0)).abs().max() < 1e-5 * float(xd.abs().sum(0).max())
    assert float(got[C - 3][valid[C - 3]].abs().max()) == 0.0


@pytest.mark.parametrize("B,H,W,Cin", [(3, 10, 9, 64), (2, 14, 14, 256), (5, 7, 7, 512), (6, 28, 28, 128)])
def test_batchnorm_train_gram_statistics(ops, B, H, W, Cin):
    """Expansion conv (N = 4C): scale/shift/EMA from the input's Gram matrix equal the ones the statistics-only conv
    launch gives, and the fused second launch reproduces F.batch_norm(train) + identity + ReLU."""
    dtype, Cout = torch.bfloat16, 4 * Cin
    x = F.relu(rnd(B, Cin, H, W, dtype=dtype, seed=1)) + 0.25         # post-ReLU input with a non-zero mean
    x = x.to(dtype)
    w = rnd(Cout, Cin, 1, 1, dtype=dtype, seed=2, scale=2.0 / Cin ** 0.5)
    idn = rnd(B, H, W, Cout, dtype=dtype, seed=5)
    gamma, beta = 0.5 + torch.rand(Cout), rnd(Cout, seed=3, scale=0.2)
    rm, rv = rnd(Cout, seed=4, scale=0.1), 0.5 + torch.rand(Cout)
    conv = F.conv2d(x.float(), w.float())
    rm_ref, rv_ref = rm.clone(), rv.clone()
    ref = F.relu(F.batch_norm(conv, rm_ref, rv_ref, gamma, beta, training=True, momentum=0.1, eps=1e-5)
                 + idn.float().permute(0, 3, 1, 2))
    xh, wp = x.permute(0, 2, 3, 1).contiguous().cuda(), pack_w(w).cuda()
    rm_d, rv_d = rm.cuda(), rv.cuda()
    part = ops.gram(xh.view(-1, Cin))
    scale, shift = ops.bn_finalize_gram(part, wp.view(Cout, Cin), B * H * W, gamma.cuda(), beta.cuda(), rm_d, rv_d, 0.1, 1e-5)
    st = ops.conv2d(xh, wp, Cout, 1, 1, 0, stats_only=True)
    scale2, shift2 = ops.bn_finalize(st, B * H * W, gamma.cuda(), beta.cuda(), None, None, 0.1, 1e-5)
    assert ((scale - scale2).abs() / scale2.abs().clamp_min(1e-3)).max() < 2e-5
    assert (shift - shift2).abs().max() < 2e-4
    y = ops.conv2d(xh, wp, Cout, 1, 1, 0, bias=shift, escale=scale, res=idn.cuda(), relu=True)
    close(y.permute(0, 3, 1, 2), ref, dtype, k=2.0)
    assert (rm_d.cpu() - rm_ref).abs().max() < 2e-3 and (rv_d.cpu() - rv_ref).abs().max() < 2e-3


@pytest.mark.parametrize("dtype", DT)
def test_pools(ops, dtype):
    x = rnd(3, 64, 13, 17, dtype=dtype, seed=1)
    xh = x.permute(0, 2, 3, 1).contiguous().cuda()
    close(ops.maxpool3x3s2(xh).permute(0, 3, 1, 2), F.max_pool2d(x.float(), 3, 2, 1), dtype)
    sc, sh = rnd(64, seed=2), rnd(64, seed=3)
    ref = F.max_pool2d(F.relu(x.float() * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)), 3, 2, 1)
    close(ops.maxpool3x3s2(xh, sc.cuda(), sh.cuda()).permute(0, 3, 1, 2), ref, dtype)
    close(ops.avgpool(xh), x.float().mean((2, 3)), dtype)


@pytest.mark.parametrize("layout,D", [("skewed", 256), ("aligned", 256), ("one_verb", 256), ("skewed", 64), ("aligned", 128)])
def test_node_init_backward_chunked_runs(ops, layout, D):
    """sr_node_init_bwd sums a verb's images in 32-image chunks of the verb-sorted batch (one wave per chunk and column strip,
    partials added in chunk order): runs inside one chunk, runs crossing chunk boundaries, runs that START on a boundary and span
    whole chunks, absent verbs, and a batch that is almost all one verb (argmax of an untrained verb head) -- against autograd.
    D = 64 / 128 (bf16): fewer column-strip lanes (8 / 16) than the 32 chunk positions a wave hands around with readlane -- lanes past
    the last strip must stay alive for that (they used to exit before loading their image index)."""
    R, V, NR = 6, 12, 17
    g = torch.Generator().manual_seed(3)
    if layout == "skewed":
        sizes = {0: 5, 2: 151, 3: 1, 5: 40, 6: 3, 9: 77, 11: 20}          # verbs 1, 4, 7, 8, 10 absent
    elif layout == "aligned":
        sizes = {1: 32, 2: 64, 4: 5, 5: 27, 7: 96, 8: 1}                    # runs that begin exactly on chunk boundaries
    else:
        sizes = {7: 290, 3: 1}
    verbs = torch.cat([torch.full((n,), v) for v, n in sizes.items()])
    verbs = verbs[torch.randperm(len(verbs), generator=g)]
    B = len(verbs)
    feat = rnd(B, D, dtype=torch.bfloat16, seed=1)
    role_emb = rnd(NR + 1, D, seed=2); role_emb[NR] = 0
    verb_emb = rnd(V, D, seed=3)
    counts = torch.randint(1, R + 1, (V,), generator=g)
    table = torch.full((V, R), NR, dtype=torch.int32)
    for v in range(V):
        k = int(counts[v])
        table[v, :k] = torch.randperm(NR, generator=g)[:k].int()
    re, ve = role_emb.clone().requires_grad_(True), verb_emb.clone().requires_grad_(True)
    ref = F.relu(feat.float()[:, None] * re[table.long()[verbs]] * ve[verbs][:, None]).reshape(B * R, D)
    dn = rnd(B * R, D, dtype=torch.bfloat16, seed=5)
    ref.backward(dn.float())
    re.grad[NR] = 0
    dre, dve = torch.full((NR + 1, D), 5.0).cuda(), torch.full((V, D), -2.0).cuda()      # written in full
    args = (dn.cuda(), feat.cuda(), role_emb.cuda(), verb_emb.cuda(), verbs.cuda(), table.cuda())
    ops.node_init_bwd(*args, dre, dve)
    close(dre, re.grad, torch.float32, k=40)
    close(dve, ve.grad, torch.float32, k=40)
    absent = [v for v in range(V) if v not in sizes]
    assert float(dve[absent].abs().max()) == 0.0 and float(dre[NR].abs().max()) == 0.0
    dre2, dve2 = torch.empty_like(dre), torch.empty_like(dve)
    ops.node_init_bwd(*args, dre2, dve2)
    assert torch.equal(dre, dre2) and torch.equal(dve, dve2)


@pytest.mark.parametrize("dtype", DT)
def test_node_init_and_aggregate(ops, dtype):
    B, R, D, V, NR = 9, 6, 128, 7, 11
    g = torch.Generator().manual_seed(0)
    feat = rnd(B, D, dtype=dtype, seed=1)
    role_emb = rnd(NR + 1, D, seed=2); role_emb[NR] = 0
    verb_emb = rnd(V, D, seed=3)
    verbs = torch.randint(0, V, (B,), generator=g)
    counts = torch.randint(1, R + 1, (V,), generator=g)
    table = torch.full((V, R), NR, dtype=torch.int32)
    adj = torch.zeros(V, R, R)
    for v in range(V):
        k = int(counts[v])
        table[v, :k] = torch.randperm(NR, generator=g)[:k].int()
        adj[v, :k, :k] = 1; adj[v].fill_diagonal_(0)
        for p_ in range(k, R): adj[v, p_, p_] = 1
    adj += 0.25 * torch.rand(V, R, R, generator=g)          # asymmetric: catches a transposed read
    re, ve = role_emb.clone().requires_grad_(True), verb_emb.clone().requires_grad_(True)
    ref = F.relu(feat.float()[:, None] * re[table.long()[verbs]] * ve[verbs][:, None]).reshape(B * R, D)
    node = ops.node_init_fwd(feat.cuda(), role_emb.cuda(), verb_emb.cuda(), verbs.cuda(), table.cuda())
    close(node, ref.detach(), dtype)
    dn = rnd(B * R, D, dtype=dtype, seed=5)
    ref.backward(dn.float())
    dre, dve = torch.zeros(NR + 1, D).cuda(), torch.zeros(V, D).cuda()
    ops.node_init_bwd(dn.cuda(), feat.cuda(), role_emb.cuda(), verb_emb.cuda(), verbs.cuda(), table.cuda(), dre, dve)
    re.grad[NR] = 0                                            # padding_idx semantics
    close(dre, re.grad, torch.float32, k=20)
    close(dve, ve.grad, torch.float32, k=20)
    assert float(dre[NR].abs().max()) == 0.0
    # fixed summation order: a second launch (into garbage-initialised buffers: both are written in full) is bit-identical
    dre2, dve2 = torch.full((NR + 1, D), 7.0).cuda(), torch.full((V, D), -3.0).cuda()
    ops.node_init_bwd(dn.cuda(), feat.cuda(), role_emb.cuda(), verb_emb.cuda(), verbs.cuda(), table.cuda(), dre2, dve2)
    assert torch.equal(dre, dre2) and torch.equal(dve, dve2)
    h = rnd(B * R, D, dtype=dtype, seed=6)
    A = adj[verbs]
    close(ops.aggregate(h.cuda(), adj.cuda(), verbs.cuda(), R), torch.bmm(A, h.float().view(B, R, D)).view(B * R, D), dtype)
    add = rnd(B * R, D, dtype=dtype, seed=7)
    close(ops.aggregate(h.cuda(), adj.cuda(), verbs.cuda(), R, transpose=True, add=add.cuda()),
          torch.bmm(A.transpose(1, 2), h.float().view(B, R, D)).view(B * R, D) + add.float(), dtype)


@pytest.mark.parametrize("dtype", DT)
def test_gru_backward_halves(ops, dtype):
    n = (77, 64)
    dh, c, h = rnd(*n, dtype=dtype, seed=1), torch.tanh(rnd(*n, seed=2)).to(dtype), rnd(*n, dtype=dtype, seed=3)
    z, r = torch.sigmoid(rnd(*n, seed=4)).to(dtype), torch.sigmoid(rnd(*n, seed=5)).to(dtype)
    drh = rnd(*n, dtype=dtype, seed=6)
    f = lambda t: t.float()
    dc, dz, dacc = ops.gru_bwd1(dh.cuda(), z.cuda(), c.cuda(), h.cuda())
    close(dc, f(dh) * f(z) * (1 - f(c) ** 2), dtype)
    close(dz, f(dh) * (f(c) - f(h)) * f(z) * (1 - f(z)), dtype)
    close(dacc, f(dh) * (1 - f(z)), dtype)
    acc0 = dacc.clone()
    dr = ops.gru_bwd2(drh.cuda(), r.cuda(), h.cuda(), dacc)
    close(dr, f(drh) * f(h) * f(r) * (1 - f(r)), dtype)
    close(dacc, acc0.float().cpu() + f(drh) * f(r), dtype)


@pytest.mark.parametrize("dtype", DT)
def test_transpose_colsum_cast_dropout(ops, dtype):
    x = rnd(333, 200, dtype=dtype, seed=1)
    cs = torch.ones(200).cuda()
    t = ops.transpose(x.cuda(), colsum=cs, colsum_scale=3.0)
    assert torch.equal(t.cpu(), x.t().contiguous())
    close(cs, 1 + 3.0 * x.float().sum(0), torch.float32, k=10)
    t2 = ops.transpose(rnd(130, 70, seed=2).cuda(), out_dtype=torch.bfloat16)
    assert torch.equal(t2.cpu(), rnd(130, 70, seed=2).t().contiguous().to(torch.bfloat16))
    cs2 = torch.zeros(200).cuda()
    ops.colsum(x.cuda(), cs2, 1.0)
    close(cs2, x.float().sum(0), torch.float32, k=10)
    assert torch.equal(ops.cast(x.cuda(), torch.float32).cpu(), x.float())
    xx = rnd(4096, 64, dtype=dtype, seed=3)
    y, m = ops.dropout_half(xx.cuda(), 1234, want_mask=True)
    y2 = ops.dropout_half(xx.cuda(), 1234)
    assert torch.equal(y, y2)
    assert torch.equal(y.cpu().float(), xx.float() * 2 * m.cpu().float())
    assert 0.48 < float(m.float().mean()) < 0.52
    assert not torch.equal(m, ops.dropout_half(xx.cuda(), 99, want_mask=True)[1])


@pytest.mark.parametrize("dtype", DT)
def test_image_prep_u8_matches_totensor_normalize_crop_flip(ops, dtype):
    """uint8 NHWC -> padded NHWC4: must equal crop + flip + ToTensor + Normalize (reference imsitu_encoder.py:21-36) followed
    by the fp32 stem_prep path."""
    g = torch.Generator().manual_seed(0)
    B, H0, W0, H, W = 3, 40, 52, 32, 36
    img = torch.randint(0, 256, (B, H0, W0, 3), generator=g, dtype=torch.uint8)
    crop = torch.tensor([[0, 0], [8, 16], [5, 3]], dtype=torch.int32)
    flip = torch.tensor([0, 1, 1], dtype=torch.uint8)
    mean, std = torch.tensor([0.485, 0.456, 0.406]), torch.tensor([0.229, 0.224, 0.225])
    ref = []
    for b in range(B):
        y0, x0 = int(crop[b, 0]), int(crop[b, 1])
        c = img[b, y0:y0 + H, x0:x0 + W].float() / 255.0
        if flip[b]:
            c = c.flip(1)
        ref.append(((c - mean) / std).permute(2, 0, 1))
    ref = torch.stack(ref)                                              # [B,3,H,W] fp32, what the DataLoader would deliver
    want = ops.stem_prep(ref.cuda(), dtype)
    got = ops.image_prep_u8(img.cuda(), dtype, out_hw=(H, W), crop_yx=crop.cuda(), flip=flip.cuda())
    assert got.shape == want.shape
    assert float((got.float() - want.float()).abs().max()) <= (1e-6 if dtype == torch.float32 else 2e-2)
    full = ops.image_prep_u8(img.cuda(), dtype)                         # no crop / flip
    want2 = ops.stem_prep(((img.float() / 255.0 - mean) / std).permute(0, 3, 1, 2).contiguous().cuda(), dtype)
    assert float((full.float() - want2.float()).abs().max()) <= (1e-6 if dtype == torch.float32 else 2e-2)


@pytest.mark.parametrize("M,N1,N2", [(1000, 256, 256), (4096 + 37, 512, 256), (36864, 256, 768)])
def test_gemm_tn_weight_gradient(ops, M, N1, N2):
    """sr_gemm_tn: out (+)= A^T B from row-major operands (ragged M: the last stage of a slice is padded from the zero
    page; several row slices; accumulation into an existing gradient)."""
    A = rnd(M, N1, dtype=torch.bfloat16, seed=1)
    B = rnd(M, N2, dtype=torch.bfloat16, seed=2)
    ref = A.double().t() @ B.double()
    out = torch.zeros(N1, N2, device="cuda")
    ops.gemm_tn(A.cuda(), B.cuda(), out, accumulate=False)
    tol_ = 2e-6 * M ** 0.5 * float(A.double().abs().max() * B.double().abs().max()) + 1e-5 * float(ref.abs().max())
    assert float((out.double().cpu() - ref).abs().max()) < tol_
    ops.gemm_tn(A.cuda(), B.cuda(), out, accumulate=True)
    assert float((out.double().cpu() - 2 * ref).abs().max()) < 2 * tol_
    # strided operands (column slices of wider matrices)
    wide = rnd(M, N1 + 256, dtype=torch.bfloat16, seed=3).cuda()
    out2 = torch.empty(N1, N2, device="cuda")
    ops.gemm_tn(wide[:, 256:], B.cuda(), out2, accumulate=False)
    ref2 = wide[:, 256:].double().cpu().t() @ B.double()
    assert float((out2.double().cpu() - ref2).abs().max()) < tol_


@pytest.mark.parametrize("M,C", [(980, 64), (70001, 64), (33333, 128), (980, 256), (50000, 256)])
def test_bn_apply_gram_fused(ops, M, C):
    """sr_bn_apply_gram = sr_bn_apply (ReLU, in place) followed by sr_gram, in one pass: the normalised tensor equals
    bn_apply's (both are one fp32 FMA + ReLU + round to bf16 per element) and the partials equal gram's on that tensor."""
    x = rnd(M, C, dtype=torch.bfloat16, seed=M + C)
    sc, sh = 0.5 + torch.rand(C), rnd(C, seed=3, scale=0.5)
    xa = x.cuda().clone()
    ops.bn_apply(xa, sc.cuda(), sh.cuda(), relu=True, out=xa)
    pa = ops.gram(xa)
    xb = x.cuda().clone()
    pb = ops.bn_apply_gram(xb, sc.cuda(), sh.cuda())
    ref = torch.relu(x.float() * sc + sh)
    assert float((xb.float().cpu() - ref).abs().max()) <= 1e-2 * float(ref.abs().max())
    assert float((xb.float() - xa.float()).abs().max()) <= 8e-3 * float(ref.abs().max())       # (at most one bf16 ulp apart)
    assert pb.shape == pa.shape
    ta, tb = pa.double().sum(0), pb.double().sum(0)
    xd = xb.double()
    G = (xd.t() @ xd).cpu()
    valid = ops.gram_valid_mask(C)
    assert float((tb[: C * C].view(C, C).cpu() - G)[valid].abs().max()) < 1e-5 * float(G.diagonal().max())
    assert float((tb[C * C:].cpu() - xd.sum(0).cpu()).abs().max()) < 1e-5 * float(xd.abs().sum(0).max())
    keep = torch.cat([valid.reshape(-1), torch.ones(C, dtype=torch.bool)]).to(ta.device)
    assert float((ta - tb)[keep].abs().max()) <= 2e-2 * float(ta[keep].abs().max())
