"""bf16 parity of the kernels that carry the benchmark, at shapes that select them.

Round 1's kernel tests all landed on the narrow tiles or the fallback kernels (`sr_gemm_tile_cfg` = 1 / 2 / 0); the
256x256 half-step ping-pong kernels -- `conv_igemm_v3_kernel<bf16,bf16,4,0|1>` (layer3's convolutions: the row-layout
residual, the staged bf16 store inside a ring slot, running BatchNorm sums over several tiles per workgroup) and
`gemm_nt_v3_kernel<bf16,bf16,4,EPI>` (GGNN gates) -- were only compared with themselves.  Here every case asserts
`sr_gemm_tile_cfg(...) == 4`, runs >= 3 tiles per workgroup (tile-to-tile ring hand-over) and is compared with a plain
fp32 reference computed from the bf16-rounded operands (torch fp32 matmul on the device over an explicit im2col; a CPU
`F.conv2d` / `F.batch_norm` cross-check of the first images guards the reference itself).
Arithmetic of reference model.py:35 (torchvision Bottleneck conv/BN/ReLU/residual) and model.py:80-84 (GRU gates).
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from situation_recognition_amd import ops as o
    o.lib()
    return o


def rnd(*shape, seed=0, scale=1.0, dtype=BF):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return (torch.randn(*shape, generator=g, device="cuda") * scale).to(dtype)


def tol(ref, k=1.0):                                  # bf16 output rounding (2^-9 relative) with margin, as tests/test_kernels_gpu.py
    return 1.2e-2 * (float(ref.abs().max()) + 1e-6) * k


def close(got, ref, k=1.0):
    err = float((got.float() - ref).abs().max())
    assert err <= tol(ref, k), "max err %g > tol %g" % (err, tol(ref, k))


def cfg(ops, M, N, linear=True):
    return ops.lib().sr_gemm_tile_cfg(int(M), int(N), int(linear), 1)


def conv_ref(x_nhwc, w, k, pad):
    """fp32 convolution of bf16-rounded operands as an explicit im2col + fp32 matmul on the device.
    x_nhwc [B,H,W,C] bf16, w [Cout,Cin,k,k] bf16 -> [B*H*W, Cout] fp32 (stride 1)."""
    B, H, W_, Cc = x_nhwc.shape
    xf = x_nhwc.float()
    if k == 1:
        return xf.view(-1, Cc) @ w.float().view(w.shape[0], Cc).t()
    xp = F.pad(xf, (0, 0, pad, pad, pad, pad))
    out = torch.zeros(B * H * W_, w.shape[0], device=xf.device)
    for r in range(k):
        for q in range(k):
            out += xp[:, r:r + H, q:q + W_, :].reshape(-1, Cc) @ w[:, :, r, q].float().t()
    return out


def pack_w(w):                                        # [Cout,Cin,KH,KW] -> [Cout, KH*KW*Cin]
    return w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).contiguous()


B14 = 1024                                            # 1024 x 14 x 14 = 200 704 output rows = 784 row tiles of 256


def test_conv3x3_generic_statistics_kernel(ops):
    """The 3x3 the GENERIC implicit-GEMM kernel still carries in the benchmark: layer4's 512 -> 512 @7 (2 launches per pass; layer1-3's
    stride-1 3x3s have direct kernels since rounds 2-4: c3d.hip, c3ds.hip) -- `<bf16,bf16,4,7>` (one-pass raw + statistics epilogue): raw bf16
    output + running BatchNorm partial sums over 3 tiles per workgroup; padding taps through out-of-range buffer loads.  Then the
    GENERAL statistics kernel `<bf16,bf16,4,1>` (statistics of conv + bias, masked edge path) on the same operands."""
    from situation_recognition_amd import _lib
    Cc, HH, BB = 512, 7, 4096
    M = BB * HH * HH
    assert cfg(ops, M, Cc) == 4
    assert ops.conv_route(BB, HH, HH, Cc, Cc, 3, 1, 1, want_stats=True) == 4 == ops.conv_route(6144, HH, HH, Cc, Cc, 3, 1, 1, want_stats=True)
    x = rnd(BB, HH, HH, Cc, seed=1)
    w = rnd(Cc, Cc, 3, 3, seed=2, scale=(Cc * 9) ** -0.5)
    y, stats = ops.conv2d(x, pack_w(w), Cc, 3, 1, 1, want_stats=True)
    ref = conv_ref(x, w, 3, 1)
    close(y.view(M, Cc), ref)
    cpu = F.conv2d(x[:2].float().cpu().permute(0, 3, 1, 2), w.float().cpu(), padding=1).permute(0, 2, 3, 1).reshape(-1, Cc)
    assert float((ref[: 2 * HH * HH].cpu() - cpu).abs().max()) < 1e-4 * float(cpu.abs().max())
    s1, s2 = stats[:, 0].double().sum(0), stats[:, 1].double().sum(0)
    r1, r2 = ref.double().sum(0), (ref.double() ** 2).sum(0)
    assert float((s1 - r1).abs().max()) < 1e-4 * float(ref.abs().sum(0).max())
    assert float(((s2 - r2).abs() / r2).max()) < 1e-4
    # and through the BatchNorm it feeds: finalize + apply against F.batch_norm(train) + ReLU of the fp32 reference
    gamma, beta = 0.5 + torch.rand(Cc, device="cuda"), 0.2 * torch.randn(Cc, device="cuda")
    rm, rv = torch.zeros(Cc, device="cuda"), torch.ones(Cc, device="cuda")
    scale, shift = ops.bn_finalize(stats, M, gamma, beta, rm, rv, 0.1, 1e-5)
    want = F.relu(F.batch_norm(ref.t().reshape(1, Cc, M), None, None, gamma, beta, training=True, eps=1e-5)).view(Cc, M).t()
    out = ops.bn_apply(y, scale, shift, relu=True)
    close(out.view(M, Cc), want, k=2.0)
    del out, want
    # a bias moves the launch to the general statistics kernel: output and sums are those of conv + bias; and a statistics-only
    # launch of either kernel returns exactly the sums of its storing launch
    bias = 0.3 * torch.randn(Cc, device="cuda")
    yb, sb = ops.conv2d(x, pack_w(w), Cc, 3, 1, 1, bias=bias, want_stats=True)
    refb = ref + bias
    close(yb.view(M, Cc), refb)
    b1, b2 = sb[:, 0].double().sum(0), sb[:, 1].double().sum(0)
    assert float((b1 - refb.double().sum(0)).abs().max()) < 1e-4 * float(refb.abs().sum(0).max())
    assert float(((b2 - (refb.double() ** 2).sum(0)).abs() / (refb.double() ** 2).sum(0)).max()) < 1e-4
    assert torch.equal(ops.conv2d(x, pack_w(w), Cc, 3, 1, 1, stats_only=True), stats)
    assert torch.equal(ops.conv2d(x, pack_w(w), Cc, 3, 1, 1, bias=bias, stats_only=True), sb)


def test_conv1x1_256_1024_scale_residual_relu(ops):
    """layer3's expansion conv (36 launches per pass, 18 % of the step) on the weight-stationary kernel `conv1x1_ws_kernel<8,8,...>`
    (csrc/expand.hip; asserted with sr_conv_route): per-channel multiplier, bias, residual prefetched one tile ahead, ReLU -- and the
    Gram-matrix statistics route in front of it.  (The generic kernel's residual epilogue `<bf16,bf16,4,0>` is covered by
    test_generic_residual_epilogue_512_2048_and_stride2_downsample below: layer4's expansion conv and the downsamples run on it.)"""
    from situation_recognition_amd import _lib
    Cin, Cout = 256, 1024
    M = B14 * 196
    assert ops.conv_route(B14, 14, 14, Cin, Cout, 1, 1, 0, bias=True, escale=True, res=True, relu=True) == _lib.ROUTE_WS
    assert ops.conv_route(6144, 14, 14, Cin, Cout, 1, 1, 0, bias=True, escale=True, res=True, relu=True) == _lib.ROUTE_WS
    x = F.relu(rnd(B14, 14, 14, Cin, seed=3)).to(BF)
    w = rnd(Cout, Cin, 1, 1, seed=4, scale=Cin ** -0.5)
    idn = rnd(B14, 14, 14, Cout, seed=5)
    esc, bias = 0.5 + torch.rand(Cout, device="cuda"), 0.3 * torch.randn(Cout, device="cuda")
    y = ops.conv2d(x, pack_w(w), Cout, 1, 1, 0, bias=bias, escale=esc, res=idn, relu=True)
    ref = F.relu(conv_ref(x, w, 1, 0) * esc + bias + idn.float().view(M, Cout))
    close(y.view(M, Cout), ref)
    # the statistics route in front of it (Gram matrix of the input) against F.batch_norm(train) of the fp32 conv
    gamma, beta = 0.5 + torch.rand(Cout, device="cuda"), 0.2 * torch.randn(Cout, device="cuda")
    rm, rv = torch.zeros(Cout, device="cuda"), torch.ones(Cout, device="cuda")
    part = ops.gram(x.view(M, Cin))
    scale, shift = ops.bn_finalize_gram(part, pack_w(w), M, gamma, beta, rm, rv, 0.1, 1e-5)
    y2 = ops.conv2d(x, pack_w(w), Cout, 1, 1, 0, bias=shift, escale=scale, res=idn, relu=True)
    conv = conv_ref(x, w, 1, 0)
    want = F.relu(F.batch_norm(conv.t().reshape(1, Cout, M), None, None, gamma, beta, training=True, eps=1e-5).view(Cout, M).t()
                  + idn.float().view(M, Cout))
    close(y2.view(M, Cout), want, k=2.0)
    mean, var = conv.double().mean(0), conv.double().var(0, unbiased=True)
    assert float((rm.double() - 0.1 * mean).abs().max()) < 2e-3 and float((rv.double() - (0.9 + 0.1 * var)).abs().max()) < 2e-3


def test_conv1x1_1024_256_statistics_kernel(ops):
    """layer3's reduce conv (35 launches per pass): K = 1024 -> 32 K-steps per tile, statistics kernel."""
    Cin, Cout = 1024, 256
    M = B14 * 196
    assert cfg(ops, M, Cout) == 4
    x = F.relu(rnd(B14, 14, 14, Cin, seed=6)).to(BF)
    w = rnd(Cout, Cin, 1, 1, seed=7, scale=Cin ** -0.5)
    y, stats = ops.conv2d(x, pack_w(w), Cout, 1, 1, 0, want_stats=True)
    ref = conv_ref(x, w, 1, 0)
    close(y.view(M, Cout), ref)
    s1, s2 = stats[:, 0].double().sum(0), stats[:, 1].double().sum(0)
    assert float((s1 - ref.double().sum(0)).abs().max()) < 1e-4 * float(ref.abs().sum(0).max())
    assert float(((s2 - (ref.double() ** 2).sum(0)).abs() / (ref.double() ** 2).sum(0)).max()) < 1e-4
    # ragged M (a last row tile that is partly out of range) on the same kernel
    Br = 1021
    y2 = ops.conv2d(x[:Br].contiguous(), pack_w(w), Cout, 1, 1, 0)
    close(y2.view(-1, Cout), ref[: Br * 196])


@pytest.mark.parametrize("M", [6144, 36864 + 100, 768 + 37])
def test_gemm_2048_all_epilogues_on_the_ping_pong_kernel(ops, M):
    """GGNN shapes (reference model.py:64,75,80-84): N = K = 2048, M = the verb path's 6144 rows (192 tiles) and the noun
    path's 36 864 (+100: a ragged last tile), 1-3 operand pairs, all five epilogues of `gemm_nt_v3_kernel<bf16,bf16,4,*>`.
    M = 805 (the verb path of an 8-GPU share, ragged): 32 tiles of 256 x 256 would occupy an eighth of the chip, so the gate epilogues run
    on the 256 x 128 instantiations `gemm_nt_v3_kernel<bf16,bf16,2,{2,3,4,5}>` (two operand pairs, two outputs) and the linear ones on
    256 x 64."""
    D = 2048
    if M > 4096:
        for lin in (True, False):
            assert cfg(ops, M, D, lin) == 4
    else:
        assert cfg(ops, M, D, False) == 2 and cfg(ops, M, D, True) in (1, 2)
    n, h, rh = rnd(M, D, seed=1), rnd(M, D, seed=2), rnd(M, D, seed=3)
    Ws = [rnd(D, D, seed=10 + i, scale=0.6 * D ** -0.5) for i in range(3)]
    b1, b2 = 0.3 * torch.randn(D, device="cuda"), 0.3 * torch.randn(D, device="cuda")
    f = lambda t: t.float()
    mm = lambda a, w: f(a) @ f(w).t()
    # linear, one pair, bias scaled by R (model.py:75: n = W_p(sum_j A_ij h_j) + R b_p)
    close(ops.gemm([(n, Ws[0])], bias=b1, bias_scale=6.0), mm(n, Ws[0]) + 6.0 * b1)
    # linear + residual + ReLU, three pairs (the backward's dn GEMM has three: model.py:80-83 transposed)
    res = rnd(M, D, seed=4)
    pre3 = mm(n, Ws[0]) + mm(h, Ws[1]) + mm(rh, Ws[2])
    close(ops.gemm([(n, Ws[0]), (h, Ws[1]), (rh, Ws[2])], bias=b1, bias2=b2, res=res, act=ops.ACT_RELU),
          F.relu(pre3 + b1 + b2 + f(res)))
    # z = sigmoid(W_z n + U_z h)                                                 model.py:80
    pre2 = mm(n, Ws[0]) + mm(h, Ws[1]) + b1 + b2
    close(ops.gemm([(n, Ws[0]), (h, Ws[1])], bias=b1, bias2=b2, act=ops.ACT_SIGMOID), torch.sigmoid(pre2))
    close(ops.gemm([(n, Ws[0]), (h, Ws[1])], bias=b1, bias2=b2, act=ops.ACT_TANH), torch.tanh(pre2))
    # r = sigmoid(W_r n + U_r h), r*h                                            model.py:81,83
    r, rmul = ops.gemm([(n, Ws[0]), (h, Ws[1])], bias=b1, bias2=b2, act=ops.ACT_SIGMOID_MUL, aux1=h)
    close(r, torch.sigmoid(pre2))
    close(rmul, torch.sigmoid(pre2) * f(h))
    # c = tanh(W_h n + U_h (r*h)), h' = (1-z) h + z c                            model.py:82-84
    z = torch.rand(M, D, device="cuda", generator=torch.Generator(device="cuda").manual_seed(9)).to(BF)
    pre_c = mm(n, Ws[0]) + mm(rh, Ws[2]) + b1 + b2
    hn, c = ops.gemm([(n, Ws[0]), (rh, Ws[2])], bias=b1, bias2=b2, act=ops.ACT_TANH_BLEND, aux1=h, aux2=z)
    close(c, torch.tanh(pre_c))
    close(hn, (1 - f(z)) * f(h) + f(z) * torch.tanh(pre_c))
    # fp32 output (classifier logits: model.py:152,168), N = 2001 -> ragged last column tile
    Wc = rnd(2001, D, seed=20, scale=D ** -0.5)
    out = ops.gemm([(n, Wc)], bias=b1[:2001].contiguous(), out_f32=True)
    assert out.dtype == torch.float32
    ref = mm(n, Wc) + b1[:2001]
    assert float((out - ref).abs().max()) < 2e-5 * float(ref.abs().max()) * 8


@pytest.mark.parametrize("Cin,Cout,rows", [(64, 256, 40000), (128, 512, 33000), (256, 1024, 32768 + 77), (256, 256, 50001), (128, 768, 36000)])
def test_output_heavy_1x1_options_and_ragged_rows(ops, Cin, Cout, rows):
    """Output-heavy 1x1 convolutions with every epilogue option and ragged row counts.  The first four shapes run on the
    weight-stationary kernel `conv1x1_ws_kernel` (csrc/expand.hip: every K variant 64 / 128 / 256, N = 256 on the 4-wave-column
    form, N = 512 / 1024 on the 8-wave-column form; row counts that are not a multiple of the 32 / 64-row tile -- the tail rows
    are dropped by the buffer descriptor's range check, their A rows read as zeros); N = 768 is a shape it does not serve and
    takes the generic implicit-GEMM kernel's `escale` / residual epilogue (asserted with sr_conv_route).  With and without
    multiplier / bias / residual / ReLU, against the fp32 matmul; a sentinel behind the last row must survive."""
    from situation_recognition_amd import _lib
    route = ops.conv_route(1, rows, 1, Cin, Cout, 1, 1, 0, bias=True, escale=True, res=True, relu=True)
    assert route == (_lib.ROUTE_WS if Cout in (256, 512, 1024) else 4)
    x = rnd(1, rows, 1, Cin, seed=rows)
    w = rnd(Cout, Cin, 1, 1, seed=Cin, scale=Cin ** -0.5)
    idn = rnd(1, rows, 1, Cout, seed=Cout)
    esc, bias = 0.5 + torch.rand(Cout, device="cuda"), 0.3 * torch.randn(Cout, device="cuda")
    base = x.float().view(rows, Cin) @ w.float().view(Cout, Cin).t()
    sentinel = torch.full((1, rows + 64, 1, Cout), 7.0, device="cuda", dtype=BF)
    for use_esc, use_bias, use_res, relu in ((1, 1, 1, 1), (0, 1, 1, 1), (1, 1, 0, 0), (0, 0, 0, 0), (0, 1, 0, 1), (1, 0, 1, 0)):
        ref = base * (esc if use_esc else 1.0) + (bias if use_bias else 0.0) + (idn.float().view(rows, Cout) if use_res else 0.0)
        ref = F.relu(ref) if relu else ref
        out = sentinel.clone()
        y = ops.conv2d(x, pack_w(w), Cout, 1, 1, 0, bias=bias if use_bias else None, escale=esc if use_esc else None,
                       res=idn if use_res else None, relu=bool(relu), out=out[:, :rows])
        close(y.view(rows, Cout), ref)
        assert float((out[:, rows:].float() - 7.0).abs().max()) == 0.0          # nothing written past the last row


@pytest.mark.parametrize("B,H", [(3, 56), (40, 56), (90, 56), (5, 24)])
def test_conv3x3_64_64_direct_kernel(ops, B, H):
    """layer1's 3x3 (3 launches per pass) on the direct-convolution kernel (csrc/c3d.hip: eight image rows per tile, weights in
    registers): train-mode form (raw output + BatchNorm partial sums; 1, 2 and 3 tiles per workgroup), statistics-only form and
    eval form (bias + ReLU), against F.conv2d in fp32 on the bf16-rounded operands; the generic implicit-GEMM kernel
    (SR_NO_C3_DIRECT=1 is read once per process, so it is reached here through a width the direct kernel does not serve)
    must agree with it to bf16 rounding."""
    Cc, W = 64, 56
    M = B * H * W
    x = F.relu(rnd(B, H, W, Cc, seed=B)).to(BF)
    w = rnd(Cc, Cc, 3, 3, seed=B + 1, scale=(Cc * 9) ** -0.5)
    y, stats = ops.conv2d(x, pack_w(w), Cc, 3, 1, 1, want_stats=True)
    assert stats.shape[0] == min(B * (H // 8), torch.cuda.get_device_properties(0).multi_processor_count) * 4
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float(), padding=1).permute(0, 2, 3, 1).reshape(M, Cc)
    close(y.view(M, Cc), ref)
    s1, s2 = stats[:, 0].double().sum(0), stats[:, 1].double().sum(0)
    r1, r2 = ref.double().sum(0), (ref.double() ** 2).sum(0)
    assert float((s1 - r1).abs().max()) < 1e-4 * float(ref.abs().sum(0).max())
    assert float(((s2 - r2).abs() / r2).max()) < 1e-4
    st2 = ops.conv2d(x, pack_w(w), Cc, 3, 1, 1, stats_only=True)
    assert torch.equal(st2, stats)
    bias = 0.3 * torch.randn(Cc, device="cuda")
    y3 = ops.conv2d(x, pack_w(w), Cc, 3, 1, 1, bias=bias, relu=True)
    close(y3.view(M, Cc), F.relu(ref + bias))
    for _ in range(3):                                 # bit-reproducible (fixed summation order, no atomics)
        yb, sb = ops.conv2d(x, pack_w(w), Cc, 3, 1, 1, want_stats=True)
        assert torch.equal(yb, y) and torch.equal(sb, stats)


@pytest.mark.parametrize("Cin,Cout,H,B", [(256, 1024, 14, 200), (128, 512, 28, 48), (64, 256, 56, 12), (256, 1024, 14, 170)])
def test_expansion_conv_normalises_its_input_on_load(ops, Cin, Cout, H, B):
    """bn2 -> relu -> conv3 of a bottleneck without the normalised tensor: `conv2d(..., in_affine=(scale2, shift2))` on the RAW conv2
    output against the same launch on the tensor `bn_apply` wrote (both round relu(x*scale+shift) to bf16 before the matrix
    cores: bit-identical), and against the fp32 reference; `bn_gram` (the Gram statistics of the normalised tensor, computed without
    writing it) against `bn_apply_gram`; an unsupported launch must raise instead of dropping the affine."""
    M = B * H * H
    y2 = rnd(B, H, H, Cin, seed=Cin + B)                              # raw conv2 output
    sc2, sh2 = 0.5 + torch.rand(Cin, device="cuda"), 0.3 * torch.randn(Cin, device="cuda")
    w = rnd(Cout, Cin, 1, 1, seed=7, scale=Cin ** -0.5)
    idn = rnd(B, H, H, Cout, seed=8)
    esc, bias = 0.5 + torch.rand(Cout, device="cuda"), 0.3 * torch.randn(Cout, device="cuda")
    assert ops.conv_in_affine_supported(y2, Cout, 1, 1, 0, res=idn, relu=True)
    lazy = ops.conv2d(y2, pack_w(w), Cout, 1, 1, 0, bias=bias, escale=esc, res=idn, relu=True, in_affine=(sc2, sh2))
    z2 = ops.bn_apply(y2, sc2, sh2, relu=True)
    eager = ops.conv2d(z2, pack_w(w), Cout, 1, 1, 0, bias=bias, escale=esc, res=idn, relu=True)
    assert torch.equal(lazy, eager)
    zf = F.relu(y2.float() * sc2 + sh2).to(BF)
    ref = F.relu(conv_ref(zf, w, 1, 0) * esc + bias + idn.float().view(M, Cout))
    close(lazy.view(M, Cout), ref)
    pa = ops.bn_gram(y2.view(M, Cin), sc2, sh2)
    keep = y2.clone()
    pb = ops.bn_apply_gram(keep.view(M, Cin), sc2, sh2)
    valid = torch.cat([ops.gram_valid_mask(Cin).reshape(-1), torch.ones(Cin, dtype=torch.bool)]).cuda()
    assert torch.equal(pa[:, valid], pb[:, valid]) and torch.equal(keep, z2)
    with pytest.raises(Exception):                                     # stride 2: not the expansion form -> loud failure
        ops.conv2d(y2, pack_w(w), Cout, 1, 2, 0, in_affine=(sc2, sh2))


@pytest.mark.parametrize("B,H", [(3, 56), (90, 56), (5, 24)])
def test_direct_3x3_normalises_its_input_on_load(ops, B, H):
    """bn1 -> relu -> conv2 of a layer1 bottleneck without the normalised tensor: the direct 3x3 kernel applies scale / shift + ReLU
    to its input patch in LDS (pad pixels and the rows above / below the image stay zero: the convolution pads the NORMALISED
    tensor).  Output and BatchNorm partial sums bit-identical to the same kernel on the tensor bn_apply wrote."""
    Cc, W = 64, 56
    y1 = rnd(B, H, W, Cc, seed=B + 3)
    sc, sh = 0.5 + torch.rand(Cc, device="cuda"), 0.3 * torch.randn(Cc, device="cuda") + 0.2     # (relu(shift) != 0 on the pads if transformed)
    w = rnd(Cc, Cc, 3, 3, seed=B + 4, scale=(Cc * 9) ** -0.5)
    assert ops.conv_in_affine_supported(y1, Cc, 3, 1, 1, res=None, relu=False, want_stats=True)
    lazy, st_l = ops.conv2d(y1, pack_w(w), Cc, 3, 1, 1, want_stats=True, in_affine=(sc, sh))
    z1 = ops.bn_apply(y1, sc, sh, relu=True)
    eager, st_e = ops.conv2d(z1, pack_w(w), Cc, 3, 1, 1, want_stats=True)
    ref = F.conv2d(F.relu(y1.float() * sc + sh).to(BF).float().permute(0, 3, 1, 2), w.float(), padding=1).permute(0, 2, 3, 1)
    # A mismatch must say WHERE (round 3's one red run printed a truncated repr: gpurun_out/r3_t6.log): the differing (image, row, column,
    # channel) coordinates, both values and the fp32 reference there -- channel-wise at one pixel = an accumulator / epilogue-strip
    # register, all channels of a pixel = a patch value (a pad or row -1 chunk that was transformed, or a chunk that was not).
    if not torch.equal(lazy, eager):
        bad = (lazy != eager).nonzero()
        lines = ["%d differing outputs; first: " % len(bad)]
        for b_, y_, x_, c_ in bad[:12].tolist():
            lines.append("  out[%d,%d,%d,%d]: on-load %.6f  bn_apply-fed %.6f  fp32 reference %.6f"
                         % (b_, y_, x_, c_, float(lazy[b_, y_, x_, c_]), float(eager[b_, y_, x_, c_]), float(ref[b_, y_, x_, c_])))
        pytest.fail("\n".join(lines))
    if not torch.equal(st_l, st_e):
        bad = (st_l != st_e).nonzero()
        pytest.fail("%d differing partial-statistics entries; first (row, which, channel): %s; values %s vs %s"
                    % (len(bad), bad[:8].tolist(), st_l[st_l != st_e][:8].tolist(), st_e[st_l != st_e][:8].tolist()))
    close(lazy.view(-1, Cc), ref.reshape(-1, Cc))
    # the positions the in-LDS normalisation can get wrong and a max-over-the-tensor tolerance would average away: image corners and
    # edges (patch row 0 / the row below the image / the pad columns must be ZERO after the transform, not relu(shift)), and the rows
    # where one tile ends and the next begins (8-row tiles: 7|8, 15|16, ...), each against the fp32 reference at the tensor's tolerance
    edge = torch.zeros(H, W, dtype=torch.bool, device="cuda")
    edge[0], edge[-1], edge[:, 0], edge[:, -1] = True, True, True, True
    edge[7::8], edge[8::8] = True, True
    close(lazy[:, edge], ref[:, edge], k=float(ref.abs().max() / ref[:, edge].abs().max()))
    corners = lazy[:, [0, 0, -1, -1], [0, -1, 0, -1]].float()
    wrong_pad = F.conv2d(F.pad(F.relu(y1.float() * sc + sh).to(BF).float().permute(0, 3, 1, 2), (1, 1, 1, 1), value=0.0)
                         + F.pad(torch.zeros_like(y1.float()).permute(0, 3, 1, 2), (1, 1, 1, 1), value=1.0) * F.relu(sh).view(1, -1, 1, 1),
                         w.float()).permute(0, 2, 3, 1)[:, [0, 0, -1, -1], [0, -1, 0, -1]]
    good = ref[:, [0, 0, -1, -1], [0, -1, 0, -1]]
    # (the test can tell the two apart: a patch whose pads were transformed differs from the reference at the corners by far more than the tolerance)
    assert float((wrong_pad - good).abs().max()) > 4 * tol(ref)
    assert float((corners - good).abs().max()) <= tol(ref)
    # repeated launches reproduce the first bit for bit (a read that beats its LDS-DMA or the in-place transform shows as run-to-run noise)
    for _ in range(6):
        again, st_a = ops.conv2d(y1, pack_w(w), Cc, 3, 1, 1, want_stats=True, in_affine=(sc, sh))
        assert torch.equal(again, lazy) and torch.equal(st_a, st_l)
    with pytest.raises(Exception):                                     # eval form (bias + ReLU): not served with an input affine
        ops.conv2d(y1, pack_w(w), Cc, 3, 1, 1, bias=sh, relu=True, in_affine=(sc, sh))


# ------------------------------------------------------------------------------------------------------------------------------
# Narrow-tile kernels (`conv_igemm_v3_kernel<bf16,bf16,2,*>` / `<...,1,*>`: 11 % of the benchmark's GPU time) and the generic
# kernel's residual epilogue at the shapes the benchmark runs, >= 3 tiles per workgroup (512 workgroup slots: two per CU).
def _stats_close(stats, ref):
    s1, s2 = stats[:, 0].double().sum(0), stats[:, 1].double().sum(0)
    r1, r2 = ref.double().sum(0), (ref.double() ** 2).sum(0)
    assert float((s1 - r1).abs().max()) < 1e-4 * float(ref.abs().sum(0).max())
    assert float(((s2 - r2).abs() / r2).max()) < 1e-4


@pytest.mark.parametrize("Cin,Cout,k,H,B,want_cfg", [(128, 128, 3, 24, 700, 2),       # a 128-channel 3x3 the direct kernel does not serve (24 x 24)
                                                     (512, 128, 1, 28, 512, 2),       # layer2's reduce conv (7 per pass)
                                                     (256, 64, 1, 56, 160, 1),        # layer1's reduce conv (2 per pass)
                                                     (64, 64, 1, 56, 160, 1)])        # layer1.0.conv1
def test_narrow_tile_kernels_at_multi_tile_production_shapes(ops, Cin, Cout, k, H, B, want_cfg):
    """256x128 (`WAVES_N = 2`) and 256x64 (`WAVES_N = 1`) tiles, four waves, two workgroups per CU, lock-step ring: train-mode form
    (raw bf16 output + running BatchNorm partial sums across the 3+ tiles a workgroup walks -- the tile-to-tile ring hand-over),
    statistics-only form and eval form (bias + ReLU), against the fp32 im2col matmul of the bf16-rounded operands."""
    M = B * H * H
    assert cfg(ops, M, Cout) == want_cfg == cfg(ops, 6144 * 28 * 28 if H == 24 else 6144 * H * H, Cout)
    pad = k // 2
    assert ops.conv_route(B, H, H, Cin, Cout, k, 1, pad, want_stats=True) == want_cfg
    slots = 2 * torch.cuda.get_device_properties(0).multi_processor_count
    assert (M + 255) // 256 * ((Cout + 64 * want_cfg - 1) // (64 * want_cfg)) >= 3 * slots       # >= 3 tiles per workgroup
    x = F.relu(rnd(B, H, H, Cin, seed=Cin + k)).to(BF)
    w = rnd(Cout, Cin, k, k, seed=Cout + k, scale=(Cin * k * k) ** -0.5)
    y, stats = ops.conv2d(x, pack_w(w), Cout, k, 1, pad, want_stats=True)
    ref = conv_ref(x, w, k, pad)
    close(y.view(M, Cout), ref)
    _stats_close(stats, ref)
    cpu = F.conv2d(x[:2].float().cpu().permute(0, 3, 1, 2), w.float().cpu(), padding=pad).permute(0, 2, 3, 1).reshape(-1, Cout)
    assert float((ref[: 2 * H * H].cpu() - cpu).abs().max()) < 1e-4 * float(cpu.abs().max())
    st2 = ops.conv2d(x, pack_w(w), Cout, k, 1, pad, stats_only=True) if Cout > 128 else None      # (no_store: Cout > 128 only)
    assert st2 is None or torch.equal(st2, stats)
    bias = 0.3 * torch.randn(Cout, device="cuda")
    y2 = ops.conv2d(x, pack_w(w), Cout, k, 1, pad, bias=bias, relu=True)
    close(y2.view(M, Cout), F.relu(ref + bias))
    # eval form of a block's last conv: bias + row-layout residual + ReLU (the residual epilogue is a kernel of its own, EPIX 6)
    idn = rnd(B, H, H, Cout, seed=Cout + 7).to(BF)
    y4 = ops.conv2d(x, pack_w(w), Cout, k, 1, pad, bias=bias, res=idn, relu=True)
    close(y4.view(M, Cout), F.relu(ref + bias + idn.float().view(M, Cout)))
    del y4, idn
    # through the BatchNorm it feeds
    gamma, beta = 0.5 + torch.rand(Cout, device="cuda"), 0.2 * torch.randn(Cout, device="cuda")
    scale, shift = ops.bn_finalize(stats, M, gamma, beta, None, None, 0.1, 1e-5)
    want = F.relu(F.batch_norm(ref.t().reshape(1, Cout, M), None, None, gamma, beta, training=True, eps=1e-5)).view(Cout, M).t()
    close(ops.bn_apply(y, scale, shift, relu=True).view(M, Cout), want, k=2.0)
    # ragged batch (a partly filled last row tile) on the same kernel
    Br = B - 3
    y3 = ops.conv2d(x[:Br].contiguous(), pack_w(w), Cout, k, 1, pad)
    close(y3.view(-1, Cout), ref[: Br * H * H])


def conv_ref_strided(x_nhwc, w, stride):
    """1x1 convolution with a stride: fp32 matmul over the subsampled pixels."""
    xs = x_nhwc[:, ::stride, ::stride, :].float()
    return xs.reshape(-1, xs.shape[3]) @ w.float().view(w.shape[0], -1).t()


def test_generic_residual_epilogue_512_2048_and_stride2_downsample(ops):
    """`conv_igemm_v3_kernel<bf16,bf16,4,0>` with `escale` + bias + row-layout residual + ReLU at the shapes the benchmark runs on
    it: layer4's expansion conv 512 -> 2048 @7 (N = 2048 is not a weight-stationary shape) at batch 2048 -- 392 row x 8 column
    tiles = 12 per workgroup -- and the stride-2 1x1 downsample 256 -> 512 (56 -> 28; statistics-only launch, then the storing
    launch with scale / shift and no residual), against the fp32 matmul of the bf16-rounded operands."""
    Cin, Cout, B, H = 512, 2048, 2048, 7
    M = B * H * H
    assert cfg(ops, M, Cout) == 4 == cfg(ops, 6144 * 49, Cout)
    assert ops.conv_route(B, H, H, Cin, Cout, 1, 1, 0, bias=True, escale=True, res=True, relu=True) == 4
    x = F.relu(rnd(B, H, H, Cin, seed=31)).to(BF)
    w = rnd(Cout, Cin, 1, 1, seed=32, scale=Cin ** -0.5)
    idn = rnd(B, H, H, Cout, seed=33)
    esc, bias = 0.5 + torch.rand(Cout, device="cuda"), 0.3 * torch.randn(Cout, device="cuda")
    y = ops.conv2d(x, pack_w(w), Cout, 1, 1, 0, bias=bias, escale=esc, res=idn, relu=True)
    conv = conv_ref(x, w, 1, 0)
    close(y.view(M, Cout), F.relu(conv * esc + bias + idn.float().view(M, Cout)))
    # train-mode route of that layer: statistics-only launch -> finalize -> storing launch (model.resnet._unit)
    st = ops.conv2d(x, pack_w(w), Cout, 1, 1, 0, stats_only=True)
    _stats_close(st, conv)
    gamma, beta = 0.5 + torch.rand(Cout, device="cuda"), 0.2 * torch.randn(Cout, device="cuda")
    scale, shift = ops.bn_finalize(st, M, gamma, beta, None, None, 0.1, 1e-5)
    y2 = ops.conv2d(x, pack_w(w), Cout, 1, 1, 0, bias=shift, escale=scale, res=idn, relu=True)
    want = F.relu(F.batch_norm(conv.t().reshape(1, Cout, M), None, None, gamma, beta, training=True, eps=1e-5).view(Cout, M).t()
                  + idn.float().view(M, Cout))
    close(y2.view(M, Cout), want, k=2.0)
    # eval form: bias + residual + ReLU, no multiplier
    y3 = ops.conv2d(x, pack_w(w), Cout, 1, 1, 0, bias=bias, res=idn, relu=True)
    close(y3.view(M, Cout), F.relu(conv + bias + idn.float().view(M, Cout)))
    del x, w, idn, y, y2, y3, conv, want
    # stride-2 downsample of layer2.0: 256 -> 512, 56 x 56 -> 28 x 28
    Cin, Cout, B, H = 256, 512, 512, 56
    M = B * 28 * 28
    assert ops.conv_route(B, H, H, Cin, Cout, 1, 2, 0, bias=True, escale=True) == 4 == ops.conv_route(6144, H, H, Cin, Cout, 1, 2, 0, bias=True, escale=True)
    x = F.relu(rnd(B, H, H, Cin, seed=41)).to(BF)
    w = rnd(Cout, Cin, 1, 1, seed=42, scale=Cin ** -0.5)
    conv = conv_ref_strided(x, w, 2)
    st = ops.conv2d(x, pack_w(w), Cout, 1, 2, 0, stats_only=True)
    _stats_close(st, conv)
    gamma, beta = 0.5 + torch.rand(Cout, device="cuda"), 0.2 * torch.randn(Cout, device="cuda")
    scale, shift = ops.bn_finalize(st, M, gamma, beta, None, None, 0.1, 1e-5)
    y = ops.conv2d(x, pack_w(w), Cout, 1, 2, 0, bias=shift, escale=scale)
    assert tuple(y.shape) == (B, 28, 28, Cout)
    want = F.batch_norm(conv.t().reshape(1, Cout, M), None, None, gamma, beta, training=True, eps=1e-5).view(Cout, M).t()
    close(y.view(M, Cout), want, k=2.0)
    ye = ops.conv2d(x, pack_w(w), Cout, 1, 2, 0, bias=beta)                  # eval form of the downsample: bias only
    close(ye.view(M, Cout), conv + beta)


@pytest.mark.parametrize("B", [1, 3, 37, 300, 1100])
def test_conv3x3_128_128_direct_kernel(ops, B):
    """layer2's 3x3 (7 launches per pass) on the direct-convolution kernel (csrc/c3ds.hip, the channel-slice kernel of layer3 instantiated
    for 128 channels @28: fourteen image rows per tile, the input staged in 32-channel slices, every wave streaming its own 32 weight rows
    through a private LDS ring): train-mode form (raw output +
    BatchNorm partial sums; batches from less than one tile per workgroup to 9 tiles per workgroup, so the rings wrap across tiles),
    statistics-only form and eval form (bias + ReLU) against F.conv2d in fp32 on the bf16-rounded operands; bit-reproducible."""
    from situation_recognition_amd import _lib
    Cc, H = 128, 28
    M = B * H * H
    assert ops.conv_route(B, H, H, Cc, Cc, 3, 1, 1, want_stats=True) == _lib.ROUTE_C3D128 == ops.conv_route(6144, H, H, Cc, Cc, 3, 1, 1, want_stats=True)
    assert ops.conv_route(B, H, H, Cc, Cc, 3, 1, 1, bias=True, relu=True) == _lib.ROUTE_C3D128
    x = F.relu(rnd(B, H, H, Cc, seed=B)).to(BF)
    w = rnd(Cc, Cc, 3, 3, seed=B + 1, scale=(Cc * 9) ** -0.5)
    y, stats = ops.conv2d(x, pack_w(w), Cc, 3, 1, 1, want_stats=True)
    assert stats.shape[0] == min(B * 2, torch.cuda.get_device_properties(0).multi_processor_count)      # (14-row tiles: two per image)
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float(), padding=1).permute(0, 2, 3, 1).reshape(M, Cc)
    close(y.view(M, Cc), ref)
    _stats_close(stats, ref)
    st2 = ops.conv2d(x, pack_w(w), Cc, 3, 1, 1, stats_only=True)
    assert torch.equal(st2, stats)
    bias = 0.3 * torch.randn(Cc, device="cuda")
    y3 = ops.conv2d(x, pack_w(w), Cc, 3, 1, 1, bias=bias, relu=True)
    close(y3.view(M, Cc), F.relu(ref + bias))
    y4 = ops.conv2d(x, pack_w(w), Cc, 3, 1, 1)
    assert torch.equal(y4, y)
    for _ in range(3):                                 # bit-reproducible (fixed summation order, no atomics)
        yb, sb = ops.conv2d(x, pack_w(w), Cc, 3, 1, 1, want_stats=True)
        assert torch.equal(yb, y) and torch.equal(sb, stats)


@pytest.mark.parametrize("B", [2, 37, 600])
def test_direct_128_3x3_normalises_its_input_on_load(ops, B):
    """bn1 -> relu -> conv2 of a layer2 bottleneck without the normalised tensor: the channel-slice kernel (c3ds.hip, 14-row tiles) applies
    scale / shift + ReLU to the NEXT 32-channel slice of the patch in LDS while the current slice multiplies (pad pixels and the rows
    above / below the image stay zero: the convolution pads the NORMALISED tensor).  Output and BatchNorm partial sums bit-identical to the same kernel on the
    tensor bn_apply wrote, and within bf16 rounding of the fp32 reference."""
    Cc, H = 128, 28
    y1 = rnd(B, H, H, Cc, seed=B + 3)
    sc, sh = 0.5 + torch.rand(Cc, device="cuda"), 0.3 * torch.randn(Cc, device="cuda") + 0.2     # (relu(shift) != 0 on the pads if transformed)
    w = rnd(Cc, Cc, 3, 3, seed=B + 4, scale=(Cc * 9) ** -0.5)
    assert ops.conv_in_affine_supported(y1, Cc, 3, 1, 1, res=None, relu=False, want_stats=True)
    lazy, st_l = ops.conv2d(y1, pack_w(w), Cc, 3, 1, 1, want_stats=True, in_affine=(sc, sh))
    z1 = ops.bn_apply(y1, sc, sh, relu=True)
    eager, st_e = ops.conv2d(z1, pack_w(w), Cc, 3, 1, 1, want_stats=True)
    ref = F.conv2d(F.relu(y1.float() * sc + sh).to(BF).float().permute(0, 3, 1, 2), w.float(), padding=1).permute(0, 2, 3, 1)
    # A mismatch must say WHERE (round 3's one red run printed a truncated repr: gpurun_out/r3_t6.log): the differing (image, row, column,
    # channel) coordinates, both values and the fp32 reference there -- channel-wise at one pixel = an accumulator / epilogue-strip
    # register, all channels of a pixel = a patch value (a pad or row -1 chunk that was transformed, or a chunk that was not).
    if not torch.equal(lazy, eager):
        bad = (lazy != eager).nonzero()
        lines = ["%d differing outputs; first: " % len(bad)]
        for b_, y_, x_, c_ in bad[:12].tolist():
            lines.append("  out[%d,%d,%d,%d]: on-load %.6f  bn_apply-fed %.6f  fp32 reference %.6f"
                         % (b_, y_, x_, c_, float(lazy[b_, y_, x_, c_]), float(eager[b_, y_, x_, c_]), float(ref[b_, y_, x_, c_])))
        pytest.fail("\n".join(lines))
    if not torch.equal(st_l, st_e):
        bad = (st_l != st_e).nonzero()
        pytest.fail("%d differing partial-statistics entries; first (row, which, channel): %s; values %s vs %s"
                    % (len(bad), bad[:8].tolist(), st_l[st_l != st_e][:8].tolist(), st_e[st_l != st_e][:8].tolist()))
    close(lazy.view(-1, Cc), ref.reshape(-1, Cc))
    # the positions the in-LDS normalisation can get wrong and a max-over-the-tensor tolerance would average away: image corners and
    # edges (patch row 0 / the row below the image / the pad columns must be ZERO after the transform, not relu(shift)), and the rows
    # where one tile ends and the next begins (the channel-slice kernel's tiles are 14 image rows: rows 13 | 14 of a 28-row image are the
    # one seam, the place where the patch row above / below a tile is a real image row and not a pad), each against the fp32 reference
    # at the tensor's tolerance
    edge = torch.zeros(H, H, dtype=torch.bool, device="cuda")
    edge[0], edge[-1], edge[:, 0], edge[:, -1] = True, True, True, True
    for seam in range(14, H, 14):
        edge[seam - 1], edge[seam] = True, True
    close(lazy[:, edge], ref[:, edge], k=float(ref.abs().max() / ref[:, edge].abs().max()))
    corners = lazy[:, [0, 0, -1, -1], [0, -1, 0, -1]].float()
    wrong_pad = F.conv2d(F.pad(F.relu(y1.float() * sc + sh).to(BF).float().permute(0, 3, 1, 2), (1, 1, 1, 1), value=0.0)
                         + F.pad(torch.zeros_like(y1.float()).permute(0, 3, 1, 2), (1, 1, 1, 1), value=1.0) * F.relu(sh).view(1, -1, 1, 1),
                         w.float()).permute(0, 2, 3, 1)[:, [0, 0, -1, -1], [0, -1, 0, -1]]
    good = ref[:, [0, 0, -1, -1], [0, -1, 0, -1]]
    # (the test can tell the two apart: a patch whose pads were transformed differs from the reference at the corners by far more than the tolerance)
    assert float((wrong_pad - good).abs().max()) > 4 * tol(ref)
    assert float((corners - good).abs().max()) <= tol(ref)
    # repeated launches reproduce the first bit for bit (a read that beats its LDS-DMA or the in-place transform shows as run-to-run noise)
    for _ in range(6):
        again, st_a = ops.conv2d(y1, pack_w(w), Cc, 3, 1, 1, want_stats=True, in_affine=(sc, sh))
        assert torch.equal(again, lazy) and torch.equal(st_a, st_l)
    with pytest.raises(Exception):                                     # eval form (bias + ReLU): not served with an input affine
        ops.conv2d(y1, pack_w(w), Cc, 3, 1, 1, bias=sh, relu=True, in_affine=(sc, sh))


def test_cu_share_changes_grids_not_results(ops):
    """`sr_set_cu_share(2)` (FCGGNN.forward sizes the two backbones' persistent grids for half the compute units each, so that both
    streams' launches co-reside): the convolution OUTPUT of every kernel family must be bit-identical to the full-grid launch (an
    output element's K order does not depend on the workgroup that computes it); the BatchNorm partial sums come in a different
    number of rows (one per workgroup) and must reduce to the same totals up to the fp32 rounding of the per-workgroup running sums."""
    from situation_recognition_amd import _lib
    cases = [(256, 256, 3, 14, 700), (1024, 256, 1, 14, 700), (128, 128, 3, 28, 300), (64, 64, 3, 56, 60), (512, 128, 1, 28, 300)]
    for Cin, Cout, k, H, B in cases:
        x = F.relu(rnd(B, H, H, Cin, seed=Cin + k)).to(BF)
        w = rnd(Cout, Cin * k * k, seed=Cout, scale=(Cin * k * k) ** -0.5)
        y1, s1 = ops.conv2d(x, w, Cout, k, 1, k // 2, want_stats=True)
        prev = ops.set_cu_share(2)
        try:
            y2, s2 = ops.conv2d(x, w, Cout, k, 1, k // 2, want_stats=True)
        finally:
            assert ops.set_cu_share(prev) == 2
        assert torch.equal(y1, y2), (Cin, Cout, k)
        assert s2.shape[0] < s1.shape[0], (Cin, Cout, k, s1.shape, s2.shape)       # fewer workgroups -> fewer partial rows
        for c in (0, 1):
            a, b = s1[:, c].double().sum(0), s2[:, c].double().sum(0)
            assert float(((a - b).abs() / a.abs().clamp_min(1e-3)).max()) < 2e-5, (Cin, Cout, k, c)
    # the weight-stationary expansion kernel and the Gram sweep
    x = F.relu(rnd(700, 14, 14, 256, seed=1)).to(BF)
    w = rnd(1024, 256, seed=2, scale=256 ** -0.5)
    idn = rnd(700, 14, 14, 1024, seed=3)
    esc, bias = 0.5 + torch.rand(1024, device="cuda"), 0.3 * torch.randn(1024, device="cuda")
    y1 = ops.conv2d(x, w, 1024, 1, 1, 0, bias=bias, escale=esc, res=idn, relu=True)
    g1 = ops.gram(x.view(-1, 256))
    prev = ops.set_cu_share(2)
    try:
        y2 = ops.conv2d(x, w, 1024, 1, 1, 0, bias=bias, escale=esc, res=idn, relu=True)
        g2 = ops.gram(x.view(-1, 256))
    finally:
        ops.set_cu_share(prev)
    assert torch.equal(y1, y2)
    valid = torch.cat([ops.gram_valid_mask(256).reshape(-1), torch.ones(256, dtype=torch.bool)]).cuda()
    a, b = g1[:, valid].double().sum(0), g2[:, valid].double().sum(0)
    assert g2.shape[0] < g1.shape[0] and float(((a - b).abs() / a.abs().clamp_min(1.0)).max()) < 2e-5


@pytest.mark.parametrize("B", [1, 3, 37, 300, 1100])
def test_conv3x3_256_256_direct_kernel(ops, B):
    """layer3's 3x3 (35 launches per pass, the largest item of the step) on the direct-convolution kernel (csrc/c3ds.hip: one image
    per tile, the input staged in 32-channel slices that serve all nine taps, every wave streaming its own 64 weight rows through a
    private LDS ring): train-mode form (raw output + BatchNorm partial sums; batches from a single tile to several tiles per workgroup,
    so the rings and the chunk buffers wrap across tiles), statistics-only form and eval form (bias + ReLU) against F.conv2d in fp32
    on the bf16-rounded operands; bit-reproducible.  (Reference call site model.py:35 -> torchvision Bottleneck.conv2 of layer3.)"""
    from situation_recognition_amd import _lib
    Cc, H = 256, 14
    M = B * H * H
    assert ops.conv_route(B, H, H, Cc, Cc, 3, 1, 1, want_stats=True) == _lib.ROUTE_C3D256 == ops.conv_route(6144, H, H, Cc, Cc, 3, 1, 1, want_stats=True)
    assert ops.conv_route(B, H, H, Cc, Cc, 3, 1, 1, bias=True, relu=True) == _lib.ROUTE_C3D256
    x = F.relu(rnd(B, H, H, Cc, seed=B)).to(BF)
    w = rnd(Cc, Cc, 3, 3, seed=B + 1, scale=(Cc * 9) ** -0.5)
    y, stats = ops.conv2d(x, pack_w(w), Cc, 3, 1, 1, want_stats=True)
    assert stats.shape[0] == min(B, torch.cuda.get_device_properties(0).multi_processor_count)
    ref4 = F.conv2d(x.float().permute(0, 3, 1, 2), w.float(), padding=1).permute(0, 2, 3, 1)
    ref = ref4.reshape(M, Cc)
    if not (y.float() - ref4).abs().max() <= tol(ref):
        bad = ((y.float() - ref4).abs() > tol(ref)).nonzero()
        pytest.fail("%d outputs off; first (image, row, column, channel): %s; channels hit: %s; pixels hit: %s" % (
            len(bad), bad[:6].tolist(), sorted(set(bad[:, 3].tolist()))[:16], sorted(set((bad[:, 1] * 14 + bad[:, 2]).tolist()))[:20]))
    close(y.view(M, Cc), ref)
    _stats_close(stats, ref)
    st2 = ops.conv2d(x, pack_w(w), Cc, 3, 1, 1, stats_only=True)
    assert torch.equal(st2, stats)
    bias = 0.3 * torch.randn(Cc, device="cuda")
    y3 = ops.conv2d(x, pack_w(w), Cc, 3, 1, 1, bias=bias, relu=True)
    close(y3.view(M, Cc), F.relu(ref + bias))
    y4 = ops.conv2d(x, pack_w(w), Cc, 3, 1, 1)
    assert torch.equal(y4, y)
    for _ in range(3):                                 # bit-reproducible (fixed summation order, no atomics)
        yb, sb = ops.conv2d(x, pack_w(w), Cc, 3, 1, 1, want_stats=True)
        assert torch.equal(yb, y) and torch.equal(sb, stats)


@pytest.mark.parametrize("B", [2, 37, 600])
def test_direct_256_3x3_normalises_its_input_on_load(ops, B):
    """bn1 -> relu -> conv2 of a layer3 bottleneck without the normalised tensor (round 4: the last `bn_apply` sweep in front of a 3x3):
    the 256-channel direct kernel applies scale / shift + ReLU to the NEXT channel slice's patch in LDS while the current slice
    multiplies (pad pixels stay zero: the convolution pads the NORMALISED tensor).  Output and BatchNorm partial sums bit-identical to
    the same kernel on the tensor bn_apply wrote, and within bf16 rounding of the fp32 reference; corners and edges on their own."""
    Cc, H = 256, 14
    y1 = rnd(B, H, H, Cc, seed=B + 3)
    sc, sh = 0.5 + torch.rand(Cc, device="cuda"), 0.3 * torch.randn(Cc, device="cuda") + 0.2     # (relu(shift) != 0 on the pads if transformed)
    w = rnd(Cc, Cc, 3, 3, seed=B + 4, scale=(Cc * 9) ** -0.5)
    assert ops.conv_in_affine_supported(y1, Cc, 3, 1, 1, res=None, relu=False, want_stats=True)
    lazy, st_l = ops.conv2d(y1, pack_w(w), Cc, 3, 1, 1, want_stats=True, in_affine=(sc, sh))
    z1 = ops.bn_apply(y1, sc, sh, relu=True)
    eager, st_e = ops.conv2d(z1, pack_w(w), Cc, 3, 1, 1, want_stats=True)
    ref = F.conv2d(F.relu(y1.float() * sc + sh).to(BF).float().permute(0, 3, 1, 2), w.float(), padding=1).permute(0, 2, 3, 1)
    if not torch.equal(lazy, eager):
        bad = (lazy != eager).nonzero()
        lines = ["%d differing outputs; first: " % len(bad)]
        for b_, y_, x_, c_ in bad[:12].tolist():
            lines.append("  out[%d,%d,%d,%d]: on-load %.6f  bn_apply-fed %.6f  fp32 reference %.6f"
                         % (b_, y_, x_, c_, float(lazy[b_, y_, x_, c_]), float(eager[b_, y_, x_, c_]), float(ref[b_, y_, x_, c_])))
        pytest.fail("\n".join(lines))
    if not torch.equal(st_l, st_e):
        bad = (st_l != st_e).nonzero()
        pytest.fail("%d differing partial-statistics entries; first (row, which, channel): %s" % (len(bad), bad[:8].tolist()))
    close(lazy.view(-1, Cc), ref.reshape(-1, Cc))
    edge = torch.zeros(H, H, dtype=torch.bool, device="cuda")
    edge[0], edge[-1], edge[:, 0], edge[:, -1] = True, True, True, True
    close(lazy[:, edge], ref[:, edge], k=float(ref.abs().max() / ref[:, edge].abs().max()))
    for _ in range(4):
        again, st_a = ops.conv2d(y1, pack_w(w), Cc, 3, 1, 1, want_stats=True, in_affine=(sc, sh))
        assert torch.equal(again, lazy) and torch.equal(st_a, st_l)
    with pytest.raises(Exception):                                     # eval form (bias + ReLU): not served with an input affine
        ops.conv2d(y1, pack_w(w), Cc, 3, 1, 1, bias=sh, relu=True, in_affine=(sc, sh))
