"""Input preparation (core/loader.py:104-219).
  * oracle/input_oracle.py against fixtures produced by the reference's own `process_data` (tests/golden/make_input_golden.py);
  * the HIP kernels (csrc/rih_input.hip), compiled for the host, and the host logic of renderih_amd/input_pipeline.py against
    the oracle and the fixtures."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from oracle import input_oracle           # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'input_pipeline.npz')
LABEL_TOL = 5e-7        # float32 label arithmetic: multiply-add contraction differs by an ulp (measured 1.1e-7)


def cases():
    g = np.load(GOLD)
    for name in g['case_names']:
        name = str(name)
        hd = {s: {k: g['%s.in.%s.%s' % (name, s, k)] for k in ('verts3d', 'joints3d', 'verts2d', 'joints2d')}
              for s in ('left', 'right')}
        p = g[name + '.params']
        br = g[name + '.bright']
        want = [g['%s.out.%d' % (name, i)] for i in range(11)]
        yield name, g[name + '.img'], hd, name != 'eval', (p[0], p[1], p[2], p[3], bool(p[4])), (br[:3], br[3]), want


def check_outputs(name, got, want, label_tol=LABEL_TOL):
    for i, (a, b) in enumerate(zip(got, want)):
        a = np.asarray(a)
        assert a.shape == b.shape and a.dtype == np.float32, (name, i, a.shape, b.shape, a.dtype)
        if i < 2:       # integer pixel pipeline, then one correctly rounded fp32 division (and subtraction): bit-exact
            assert np.array_equal(a, b), (name, i, np.abs(a - b).max())
        else:
            scale = max(1.0, float(np.abs(b).max()))
            assert np.abs(a - b).max() <= label_tol * scale, (name, i, np.abs(a - b).max(), scale)


def test_oracle_matches_reference_fixtures():
    n = 0
    for name, img, hd, train, params, bright, want in cases():
        got = input_oracle.process_data(img, hd, train, params, bright)
        check_outputs(name, got, want, label_tol=0.0)         # same numpy: identical to the last bit
        n += 1
    assert n == 5


def test_warp_restatement_properties():
    """No OpenCV here: size-independent properties of the restated fixed-point warp."""
    rs = np.random.RandomState(0)
    img = rs.randint(0, 256, (40, 40, 3)).astype(np.uint8)
    ident = np.array([[1, 0, 0], [0, 1, 0]], np.float32)
    assert np.array_equal(input_oracle.warp_affine_u8(img, ident, (40, 40)), img)
    shift = np.array([[1, 0, 3], [0, 1, -2]], np.float32)          # integer translation: exact copy, zero border
    out = input_oracle.warp_affine_u8(img, shift, (40, 40))
    assert np.array_equal(out[0:38, 3:40], img[2:40, 0:37]) and not out[38:].any() and not out[:, :3].any()
    half = np.array([[1, 0, 0.5], [0, 1, 0]], np.float32)          # half-pixel shift: rounded mean of neighbours
    out = input_oracle.warp_affine_u8(img, half, (40, 40))
    want = (img[:, :-1].astype(np.int32) + img[:, 1:].astype(np.int32) + 1) >> 1
    assert np.array_equal(out[:, 1:], want.astype(np.uint8))
    rot = input_oracle.get_affine_mat(90.0, 1.0, 0, 0, 40, 40)     # the reference's pi = 3.14159: nearly a pure rotation
    out = input_oracle.warp_affine_u8(np.full((40, 40, 3), 200, np.uint8), rot[:2], (40, 40))
    assert (out[2:-2, 2:-2] == 200).all()


def test_warp_restatement_agrees_with_an_independent_bilinear_resampler():
    """The one unpinned piece (cv.warpAffine is not installable here) cross-checked against an implementation that shares no
    code with it: scipy.ndimage.affine_transform (order 1, 'grid-constant' = taps outside the image read 0, like
    BORDER_CONSTANT) on the reference's own augmentation matrices (`get_affine_mat`: rotation about the image centre, scale,
    translation).  OpenCV quantises source coordinates to 1/32 pixel and its weights to 15 bits; on an image whose gradient is
    <= 9 grey levels per pixel that is worth < 1 level, so the two must agree to 1 level wherever all four taps lie inside the
    source image (and to 255/32 levels where a tap crosses its edge).  This
    pins the matrix convention (forward matrix, inverted inside), the axis order, the centre convention (integer coordinates
    = pixel centres) and the border rule; the exact fixed-point rounding of OpenCV stays unpinned."""
    from scipy import ndimage
    S = 96
    yy, xx = np.mgrid[0:S, 0:S].astype(np.float64)
    img = np.stack([127 + 90 * np.sin(xx / 11.0 + c) * np.cos(yy / 13.0 - c) for c in (0.0, 1.0, 2.0)], -1)
    img = np.clip(np.rint(img), 0, 255).astype(np.uint8)
    assert np.abs(np.diff(img.astype(np.int32), axis=0)).max() <= 9 and np.abs(np.diff(img.astype(np.int32), axis=1)).max() <= 9
    rs = np.random.RandomState(3)
    worst = 0
    for _ in range(12):
        theta, scale = rs.uniform(-60, 60), rs.uniform(0.7, 1.4)
        u, v = rs.uniform(-12, 12), rs.uniform(-12, 12)
        A = input_oracle.get_affine_mat(theta, scale, u, v, S, S)[:2]
        got = input_oracle.warp_affine_u8(img, A, (S, S)).astype(np.int32)
        m = input_oracle.invert_affine(A)               # destination (x, y) -> source (m0 x + m1 y + m2, m3 x + m4 y + m5)
        mat = np.array([[m[4], m[3]], [m[1], m[0]]])    # scipy works in (row, col) = (y, x)
        ref = np.stack([ndimage.affine_transform(img[..., c].astype(np.float64), mat, offset=[m[5], m[2]], output_shape=(S, S),
                                                 order=1, mode='grid-constant', cval=0.0) for c in range(3)], -1)
        diff = np.abs(got - np.rint(ref)).max(-1)
        sx, sy = m[0] * xx + m[1] * yy + m[2], m[3] * xx + m[4] * yy + m[5]
        inside = (sx >= 0) & (sx <= S - 1) & (sy >= 0) & (sy <= S - 1)       # all four taps inside the source image
        worst = max(worst, int(diff[inside].max()))
        # where a tap crosses the image edge the signal jumps to the border value: 1/32 pixel is worth up to 255/32 levels there
        assert diff[inside].max() <= 1 and diff.max() <= 8 and diff.mean() < 0.25, (theta, scale, u, v, diff.max(), diff.mean())
    assert worst <= 1


def prepare_vs_fixtures(dev):
    """renderih_amd.input_pipeline.BatchPreparer against the reference's own outputs, one sample per call and the three
    64-pixel cases as one batch (per-sample matrices, flips and brightness in one launch)."""
    from renderih_amd.input_pipeline import BatchPreparer, pack_labels
    all_cases = list(cases())
    for name, img, hd, train, params, bright, want in all_cases:
        prep = BatchPreparer(train=train, fp16_nhwc8=True)
        p2, p3 = pack_labels([hd])
        out, h8 = prep(torch.from_numpy(img)[None].to(dev), p2.to(dev), p3.to(dev), params=[params], bright=[bright])
        check_outputs(name, [t[0].cpu().numpy() for t in out], want)
        ref16 = torch.from_numpy(want[1]).permute(1, 2, 0).to(torch.float16)
        assert torch.equal(h8[0, ..., :3].cpu(), ref16) and not bool(h8[0, ..., 3:].any())
    batch = [c for c in all_cases if c[1].shape[0] == 64 and c[3]]
    assert len(batch) >= 2
    prep = BatchPreparer(train=True)
    p2, p3 = pack_labels([c[2] for c in batch])
    out = prep(torch.from_numpy(np.stack([c[1] for c in batch])).to(dev), p2.to(dev), p3.to(dev),
               params=[c[4] for c in batch], bright=[c[5] for c in batch])
    for i, c in enumerate(batch):
        check_outputs(c[0] + '[batched]', [t[i].cpu().numpy() for t in out], c[6])


def test_host_logic_matches_reference_fixtures():
    from abi_emulator import emulated_abi
    with emulated_abi():
        prepare_vs_fixtures(torch.device('cpu'))


def test_kernels_match_reference_fixtures_on_cpu():
    from hipcpu.host_kernels import host_kernels_abi
    with host_kernels_abi():
        prepare_vs_fixtures(torch.device('cpu'))


def test_random_draws_follow_the_reference_order():
    """augm_params consumes random.random() exactly like core/loader.py:96-102."""
    import random
    from renderih_amd.input_pipeline import BatchPreparer
    prep = BatchPreparer(theta=(-90, 90), scale=(0.75, 1.25), uv=(-10, 10))
    random.seed(5)
    got = prep.augm_params()
    random.seed(5)
    r = [random.random() for _ in range(5)]
    assert got == (r[0] * 180 - 90, r[1] * 0.5 + 0.75, r[2] * 20 - 10, r[3] * 20 - 10, r[4] > 0.5)
