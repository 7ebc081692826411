"""The fused launch `sr_conv_pair` (csrc/pair.hip): a bottleneck's expansion conv + the next block's reduce conv in one pass over the
block output (reference chain conv3 -> bn3 -> add -> relu -> next.conv1, model.py:35 -> torchvision Bottleneck).

Parity: both outputs against an fp32 reference of the same arithmetic AND bit for bit against the two launches it replaces (the
weight-stationary expansion kernel and the generic reduce conv: same fp32 FMA / rounding points, same K order); the BatchNorm partial
sums against the unfused launch's (summation order differs) and against fp64 sums."""
import pytest
import torch

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
# (image side, mid channels, output channels of the reduce conv): expansion = 4 x mid; "a>b": layer a's last block + layer b's first conv1
SHAPES = {"layer3": (14, 256, 256), "layer2": (28, 128, 128), "layer1": (56, 64, 64), "layer2>3": (28, 128, 256), "layer1>2": (56, 64, 128)}
TILE_ROWS = {"layer3": 192, "layer2": 256, "layer1": 256, "layer2>3": 192, "layer1>2": 256}


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from situation_recognition_amd import ops
    ops.lib()
    return ops


def operands(B, seed, layer="layer3"):
    H, C, CR = SHAPES[layer]
    CX = 4 * C
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = torch.randn(B, H, H, C, device="cuda", generator=g).to(BF)                       # raw 3x3 output
    res = torch.relu(torch.randn(B, H, H, CX, device="cuda", generator=g)).to(BF)        # identity: a block output
    w3 = (torch.randn(CX, C, device="cuda", generator=g) * C ** -0.5).to(BF)
    w1 = (torch.randn(CR, CX, device="cuda", generator=g) * CX ** -0.5).to(BF)
    insc, insh = 0.5 + torch.rand(C, device="cuda", generator=g), 0.1 * torch.randn(C, device="cuda", generator=g)
    esc, esh = 0.2 + 0.3 * torch.rand(CX, device="cuda", generator=g), 0.1 * torch.randn(CX, device="cuda", generator=g)
    return x, res, w3, w1, (insc, insh), esc, esh


def fp32_reference(x, res, w3, w1, aff, esc, esh, rows):
    CX, C = w3.shape
    M = x.numel() // C
    xa = x.view(M, C)[rows].float()
    if aff is not None:
        xa = torch.relu(xa * aff[0] + aff[1]).to(BF).float()                             # the matrix cores see the normalised tensor in bf16
    z = torch.relu(xa @ w3.float().t() * esc + esh + res.view(M, CX)[rows].float())
    return z


@pytest.mark.parametrize("layer,B,in_affine", [("layer3", 1024, True), ("layer3", 1024, False), ("layer2", 301, True), ("layer1", 75, True),
                                               ("layer2>3", 253, True), ("layer1>2", 75, False)])
def test_pair_equals_the_two_launches_it_replaces(ops, layer, B, in_affine):
    """Layer3 at batch 1024: 1046 tiles of 192 rows (the last one ragged: 64 rows, two waves of it empty) on 256 workgroups = 4 tiles
    each; layer2 / layer1 (256-row tiles) at batches with as many rows and a ragged last tile."""
    H, C, CR = SHAPES[layer]
    CX = 4 * C
    x, res, w3, w1, aff, esc, esh = operands(B, 21, layer)
    M = B * H * H
    TM = TILE_ROWS[layer]
    assert M % TM != 0 and (M + TM - 1) // TM >= 3 * 256
    if not in_affine:
        x = torch.relu(x.float() * aff[0] + aff[1]).to(BF)
        aff = None
    from situation_recognition_amd import _lib
    assert ops.conv_route(B, H, H, C, CX, 1, 1, 0, res=True, relu=True, bias=True, escale=True, in_affine=in_affine) == _lib.ROUTE_WS
    assert ops.conv_route(B, H, H, CX, CR, 1, 1, 0, want_stats=True) == ops.conv_route(6144, H, H, CX, CR, 1, 1, 0, want_stats=True)
    z0 = ops.conv2d(x, w3, CX, 1, 1, 0, bias=esh, escale=esc, res=res, relu=True, in_affine=aff)
    y0, st0 = ops.conv2d(z0, w1, CR, 1, 1, 0, want_stats=True)
    wp = ops.conv_pair_pack(w3, w1)
    z1, y1, st1 = ops.conv_pair(x, wp, res, esc, esh, in_affine=aff)
    assert torch.equal(z0.view(torch.int16), z1.view(torch.int16))
    assert torch.equal(y0.view(torch.int16), y1.view(torch.int16))
    s0, s1 = st0.double().sum(0), st1.double().sum(0)
    assert float((s0[0] - s1[0]).abs().max() / s0[0].abs().max()) < 1e-6
    assert float((s0[1] - s1[1]).abs().max() / s0[1].abs().max()) < 1e-6
    # fp32 reference on the first / last rows (the ragged tile) and a random sample
    g = torch.Generator(device="cuda").manual_seed(5)
    rows = torch.cat([torch.arange(0, 512, device="cuda"), torch.arange(M - 512, M, device="cuda"), torch.randint(0, M, (4096,), device="cuda", generator=g)])
    zr = fp32_reference(x, res, w3, w1, aff, esc, esh, rows)
    assert float((zr - z1.view(M, CX)[rows].float()).abs().max()) <= 1.2e-2 * float(zr.abs().max())
    yr = z1.view(M, CX)[rows].float() @ w1.float().t()
    assert float((yr - y1.view(M, CR)[rows].float()).abs().max()) <= 1.2e-2 * float(yr.abs().max())
    # statistics = column sums / sums of squares of the fp32 product over ALL rows (rows past M contribute nothing)
    yf = (z1.view(M, CX).float() @ w1.float().t()).double()
    assert float((yf.sum(0) - s1[0]).abs().max() / s1[0].abs().max()) < 1e-5
    assert float(((yf * yf).sum(0) - s1[1]).abs().max() / s1[1].abs().max()) < 1e-5
    # bit-reproducible (every wait in the kernel is a counted vmcnt: a race would show as run-to-run differences)
    for _ in range(5):
        z2, y2, st2 = ops.conv_pair(x, wp, res, esc, esh, in_affine=aff)
        assert torch.equal(z1.view(torch.int16), z2.view(torch.int16)) and torch.equal(y1.view(torch.int16), y2.view(torch.int16))
        assert torch.equal(st1, st2)


@pytest.mark.parametrize("layer,B", [("layer3", 1), ("layer3", 3), ("layer3", 700), ("layer2", 1), ("layer2", 50), ("layer1", 1), ("layer1", 9),
                                     ("layer2>3", 1), ("layer2>3", 41), ("layer1>2", 2)])
def test_pair_small_and_ragged_row_counts(ops, layer, B):
    """One tile and a few rows, grids smaller than the chip; against the fp32 reference over all rows."""
    H, C, CR = SHAPES[layer]
    CX = 4 * C
    x, res, w3, w1, aff, esc, esh = operands(B, 30 + B, layer)
    M = B * H * H
    TM = TILE_ROWS[layer]
    wp = ops.conv_pair_pack(w3, w1)
    z1, y1, st1 = ops.conv_pair(x, wp, res, esc, esh, in_affine=aff)
    rows = torch.arange(0, M, device="cuda")
    zr = fp32_reference(x, res, w3, w1, aff, esc, esh, rows)
    assert float((zr - z1.view(M, CX).float()).abs().max()) <= 1.2e-2 * float(zr.abs().max())
    yf = z1.view(M, CX).float() @ w1.float().t()
    assert float((yf - y1.view(M, CR).float()).abs().max()) <= 1.2e-2 * float(yf.abs().max())
    s1 = st1.double().sum(0)
    assert st1.shape[0] == min(256, (M + TM - 1) // TM)
    assert float((yf.double().sum(0) - s1[0]).abs().max() / s1[0].abs().max()) < 1e-5
    assert float(((yf.double() ** 2).sum(0) - s1[1]).abs().max() / s1[1].abs().max()) < 1e-5


@pytest.mark.parametrize("layer,B", [("layer3", 1024), ("layer2", 301), ("layer1>2", 75)])
def test_pair_eval_mode_equals_the_two_folded_launches(ops, layer, B):
    """Eval mode (BatchNorm folded into weights and biases): z = relu(x W3^T + b3 + identity), y = relu(z W1^T + b1), no statistics --
    bit for bit the weight-stationary expansion launch followed by the generic reduce launch with its bias + ReLU epilogue."""
    H, C, CR = SHAPES[layer]
    CX = 4 * C
    x, res, w3, w1, aff, esc, esh = operands(B, 77, layer)
    x = torch.relu(x.float() * aff[0] + aff[1]).to(BF)              # (the 3x3's own eval epilogue has normalised its output)
    b1 = 0.2 * torch.randn(CR, device="cuda")
    z0 = ops.conv2d(x, w3, CX, 1, 1, 0, bias=esh, res=res, relu=True)
    y0 = ops.conv2d(z0, w1, CR, 1, 1, 0, bias=b1, relu=True)
    z1, y1, st = ops.conv_pair(x, ops.conv_pair_pack(w3, w1), res, torch.ones_like(esh), esh, ybias=b1, yrelu=True)
    assert st is None
    assert torch.equal(z0.view(torch.int16), z1.view(torch.int16))
    assert torch.equal(y0.view(torch.int16), y1.view(torch.int16))
    M = B * H * H
    rows = torch.cat([torch.arange(0, 256, device="cuda"), torch.arange(M - 256, M, device="cuda")])
    zr = torch.relu(x.view(M, C)[rows].float() @ w3.float().t() + esh + res.view(M, CX)[rows].float())
    assert float((zr - z1.view(M, CX)[rows].float()).abs().max()) <= 1.2e-2 * float(zr.abs().max())
    yr = torch.relu(z1.view(M, CX)[rows].float() @ w1.float().t() + b1)
    assert float((yr - y1.view(M, CR)[rows].float()).abs().max()) <= 1.2e-2 * float(yr.abs().max())
    from situation_recognition_amd import _lib
    with pytest.raises(_lib.SrError):                                # eval mode takes an already normalised x
        ops.conv_pair(x, ops.conv_pair_pack(w3, w1), res, esc, esh, in_affine=aff, ybias=b1)


def test_pair_rejects_what_it_does_not_serve(ops):
    from situation_recognition_amd import _lib
    assert ops.conv_pair_supported(200704, 256, 1024)
    assert ops.conv_pair_supported(200704, 128, 512) and ops.conv_pair_supported(200704, 64, 256)
    assert ops.conv_pair_supported(200704, 128, 512, 256) and ops.conv_pair_supported(200704, 64, 256, 128)
    assert not ops.conv_pair_supported(200704, 512, 2048) and not ops.conv_pair_supported(200704, 256, 512)
    assert not ops.conv_pair_supported(200704, 256, 1024, 512)         # layer3 -> layer4: 512 reduce outputs do not fit the accumulators
    assert not ops.conv_pair_supported(200704, 256, 1024, dtype=torch.float32)
    x, res, w3, w1, aff, esc, esh = operands(2, 3)
    wp = ops.conv_pair_pack(w3, w1)
    with pytest.raises(_lib.SrError):
        ops.conv_pair(x, wp[:-8], res, esc, esh)
    with pytest.raises(_lib.SrError):
        ops.conv_pair(x.cpu(), wp, res, esc, esh)
