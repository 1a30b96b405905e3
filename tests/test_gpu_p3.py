"""Parity tests of the conversion-free split-bf16 GEMM on pre-split "P3" operands (csrc/rih_gemm3.hip) against torch in
fp64 on the CPU: format round trip, implicit-GEMM convolutions (3x3 / 1x1, strided, ragged M and N, every tile), the fused
epilogue (bias, residual, ReLU), the flipped data-gradient operand with parity-class output rows, and the per-tile
BatchNorm statistics + their merge.  Tolerance = the fp32 bar (1e-4 relative + 1e-5 of the largest value): the operands
are split exactly (to 2^-24), so the result must be fp32-grade."""
import math
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from renderih_amd.testing import assert_close, experiments_built

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not experiments_built(), reason='csrc/rih_gemm3.hip is an experiment outside the default library; '
                                                                  'build with RIH_BUILD_EXPERIMENTS=1 to test it')]


def dev():
    assert torch.cuda.is_available(), 'GPU tests need a GPU'
    return torch.device('cuda:0')


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return torch.randn(*shape, generator=g) * scale


def p3_to_f32(p3, rows, C):
    """Decode a P3 byte tensor [rows][6*C] on the host: (hi, mid, lo) planes as fp32 [3][rows][C]."""
    u = p3.cpu().contiguous().view(torch.int16).reshape(rows, C // 8, 3, 8).to(torch.int32)
    f = (u << 16).view(torch.float32)
    return f.permute(2, 0, 1, 3).reshape(3, rows, C)


def test_p3_format_round_trip():
    from renderih_amd import ops
    rows, C = 37, 72
    x = rnd(rows, C, seed=3) * torch.logspace(-6, 6, C)[None, :]
    x[0, :8] = torch.tensor([0.0, -0.0, 1.0, -1.0, 1e-30, 3.0e38, 1 + 2 ** -23, -(1 - 2 ** -24)])
    p3 = ops.p3_from_f32(rows, C, x.to(dev()))
    planes = p3_to_f32(p3, rows, C).double()
    rec = planes.sum(0)
    err = (rec - x.double()).abs()
    assert bool((err <= x.double().abs() * 2.0 ** -24).all()), float((err / x.double().abs().clamp_min(1e-300)).max())
    # each plane is a bf16 value (low 16 bits zero by construction) and the planes shrink by >= 2^-8 per level
    assert bool((planes[1].abs() <= planes[0].abs() * 2.0 ** -8 + 1e-300).all())
    assert bool((planes[2].abs() <= planes[0].abs() * 2.0 ** -16 + 1e-300).all())


def conv_p3(x_nhwc, w, stride, pad, tile, bias=None, res=None, relu=False, stats=False, layout=0):
    """conv through rih_gemm_p3 on device tensors: x [N,H,W,Cin] fp32, w OIHW; layout 1 = slab-major operands."""
    from renderih_amd import ops
    N, H, W, Cin = x_nhwc.shape
    Cout, _, KH, KW = w.shape
    Ho, Wo = (H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1
    M, K = N * Ho * Wo, KH * KW * Cin
    xp = ops.p3_from_f32(N * H * W, Cin, x_nhwc.contiguous(), layout=layout)
    wp, Kp = ops.p3_weight(w.contiguous(), Cin, False, layout=layout)
    y = torch.full((N, Ho, Wo, Cout), float('nan'), device=x_nhwc.device)
    st = None
    if stats:
        bm = ops.P3_TILES[tile][0]
        assert M % bm == 0
        st = torch.full((M // bm, Cout, 2), float('nan'), device=x_nhwc.device)
    ops.gemm_p3(xp, wp, y, M, Cout, K, Cin, Kp, Cout, (H, W, Cin, Ho, Wo, KH, KW, stride, pad, pad), bias=bias, R=res,
                ldr=Cout, relu=relu, stats=st, tile=tile, layout=layout)
    return y, st


P3_CONV_CASES = [
    # N, H, W, Cin, Cout, k, stride, pad, tile
    (2, 8, 8, 64, 128, 3, 1, 1, 1),           # one 128x128 tile, 18 k-tiles, halo taps
    (1, 16, 16, 32, 128, 3, 1, 1, 0),         # one 256x128 tile
    (2, 8, 8, 64, 64, 1, 1, 0, 2),            # 1x1, 128x64 tile, two k-tiles
    (2, 9, 7, 32, 96, 3, 2, 1, 2),            # strided, odd sizes, ragged M (40 rows) and N
    (1, 12, 12, 64, 160, 1, 2, 0, 1),         # strided 1x1 (downsample), N tail
    (3, 10, 10, 32, 136, 3, 1, 1, 0),         # M = 300: two 256-row tiles, second ragged; N tail
]


@pytest.mark.parametrize('layout', [0, 1])
@pytest.mark.parametrize('case', P3_CONV_CASES)
def test_p3_conv_forward(case, layout):
    N, H, W, Cin, Cout, k, s, p, tile = case
    x = rnd(N, H, W, Cin, seed=1)
    w = rnd(Cout, Cin, k, k, seed=2, scale=1.0 / math.sqrt(Cin * k * k))
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), stride=s, padding=p).permute(0, 2, 3, 1)
    y, _ = conv_p3(x.to(dev()), w.to(dev()), s, p, tile, layout=layout)
    assert_close(y, ref.float(), what='p3 conv %s layout %d' % (case, layout))


def test_p3_epilogue_bias_residual_relu():
    N, H, W, Cin, Cout = 2, 8, 8, 64, 72
    x, w = rnd(N, H, W, Cin, seed=1), rnd(Cout, Cin, 1, 1, seed=2, scale=0.125)
    b, r = rnd(Cout, seed=3), rnd(N, H, W, Cout, seed=4)
    ref = F.relu(F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), b.double()).permute(0, 2, 3, 1) + r.double())
    d = dev()
    for layout in (0, 1):
        y, _ = conv_p3(x.to(d), w.to(d), 1, 0, 2, bias=b.to(d), res=r.to(d), relu=True, layout=layout)
        assert_close(y, ref.float(), what='p3 epilogue layout %d' % layout)


@pytest.mark.parametrize('tile,shape', [(0, (2, 16, 16, 64, 128)), (1, (2, 8, 16, 32, 200)), (2, (4, 8, 8, 32, 64))])
def test_p3_tile_statistics_and_merge(tile, shape):
    """Per-tile (mean, centred sum of squares) from the GEMM epilogue merged by rih_bn_stats_merge = batch statistics of
    the convolution output (nn.BatchNorm2d training forward: biased variance)."""
    from renderih_amd import ops, _lib
    N, H, W, Cin, Cout = shape
    x = rnd(N, H, W, Cin, seed=1) + 0.7          # non-zero mean: the centred accumulation matters
    w = rnd(Cout, Cin, 3, 3, seed=2, scale=1.0 / math.sqrt(Cin * 9))
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), padding=1).permute(0, 2, 3, 1).reshape(-1, Cout)
    d = dev()
    y, st = conv_p3(x.to(d), w.to(d), 1, 1, tile, stats=True)
    T = st.shape[0]
    mean, var = torch.empty(Cout, device=d), torch.empty(Cout, device=d)
    ops.check(_lib.load().rih_bn_stats_merge(st.data_ptr(), T, Cout, ops.P3_TILES[tile][0], mean.data_ptr(), var.data_ptr(),
                                             ops._stream()), 'rih_bn_stats_merge')
    assert_close(mean, ref.mean(0).float(), 1e-4, 1e-5, 'p3 stats mean')
    assert_close(var, ref.var(0, unbiased=False).float(), 1e-4, 1e-5, 'p3 stats var')
    # the same partials finalised like nn.BatchNorm2d's training forward (what ops.batchnorm(tile_stats=...) consumes)
    invstd = torch.empty(Cout, device=d)
    rm, rv = torch.zeros(Cout, device=d), torch.ones(Cout, device=d)
    ops.check(_lib.load().rih_bn_stats_from_tiles(st.data_ptr(), T, Cout, ops.P3_TILES[tile][0], 1e-5, 0.1, mean.data_ptr(),
                                                  invstd.data_ptr(), rm.data_ptr(), rv.data_ptr(), ops._stream()), 'from_tiles')
    assert_close(invstd, (1.0 / torch.sqrt(ref.var(0, unbiased=False) + 1e-5)).float(), 1e-4, 1e-5, 'invstd')
    assert_close(rm, (0.1 * ref.mean(0)).float(), 1e-4, 1e-5, 'running mean')
    assert_close(rv, (0.9 + 0.1 * ref.var(0, unbiased=True)).float(), 1e-4, 1e-5, 'running var')


@pytest.mark.parametrize('layout', [0, 1])
@pytest.mark.parametrize('case', [(2, 8, 8, 64, 32, 3, 1, 1), (2, 9, 7, 32, 64, 3, 2, 1), (2, 8, 8, 64, 32, 1, 2, 0)])
def test_p3_data_gradient(case, layout):
    """dx of a convolution = GEMM of the (P3) output gradient with the flipped P3 weight operand; for stride s as s*s
    parity classes written in place through the strided output rows (the scheme of ops.Conv2dFn.backward)."""
    from renderih_amd import ops
    N, H, W, Cin, Cout, k, s, p = case
    x = rnd(N, Cin, H, W, seed=1).double().requires_grad_(True)
    w = rnd(Cout, Cin, k, k, seed=2, scale=1.0 / math.sqrt(Cin * k * k))
    yr = F.conv2d(x, w.double(), stride=s, padding=p)
    gy = rnd(*yr.shape, seed=4)
    yr.backward(gy.double())
    d = dev()
    Ho, Wo = yr.shape[2], yr.shape[3]
    dy = gy.permute(0, 2, 3, 1).contiguous().to(d)
    dyp = ops.p3_from_f32(N * Ho * Wo, Cout, dy, layout=layout)
    wg = w.to(d)
    dx = torch.zeros(N, H, W, Cin, device=d)
    for oh in range(s):
        for ow in range(s):
            kh0, kw0 = (oh + p) % s, (ow + p) % s
            Th, Tw = len(range(kh0, k, s)), len(range(kw0, k, s))
            Hc, Wc = len(range(oh, H, s)), len(range(ow, W, s))
            if Th == 0 or Tw == 0 or Hc == 0 or Wc == 0:
                continue
            padh, padw = Th - 1 - (oh + p - kh0) // s, Tw - 1 - (ow + p - kw0) // s
            wd, Kp = ops.p3_weight(wg, Cin, True, (kh0, kw0, s, Th, Tw), layout=layout)
            ops.gemm_p3(dyp, wd, dx, N * Hc * Wc, Cin, Th * Tw * Cout, Cout, Kp, Cin,
                        (Ho, Wo, Cout, Hc, Wc, Th, Tw, 1, padh, padw), cstride=(s, oh, ow, H, W) if s > 1 else None, tile=2,
                        layout=layout)
    assert_close(dx.permute(0, 3, 1, 2), x.grad.float(), 1e-4, 1e-5, 'p3 dgrad %s' % (case,))
