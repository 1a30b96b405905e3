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
