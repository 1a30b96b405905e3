"""On-disk format reader (renderih_amd/dataset.py; dataset/interhand.py:230-268) on a synthetic folder."""
import os
import pickle
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from renderih_amd.dataset import InterHandFolder, collate_for_gpu, rotmat_to_axis_angle      # noqa: E402


def test_rodrigues_matches_scipy():
    from scipy.spatial.transform import Rotation
    rs = np.random.RandomState(0)
    vecs = [rs.randn(3) * s for s in (1e-9, 1e-3, 0.5, 1.5, 3.0) for _ in range(4)] + [np.zeros(3), np.array([np.pi, 0, 0]),
                                                                                      np.array([0, 0, -np.pi]) * 0.999999]
    for v in vecs:
        R = Rotation.from_rotvec(v).as_matrix()
        got = rotmat_to_axis_angle(R)
        near_pi = abs(np.linalg.norm(v) - np.pi) < 1e-4          # like OpenCV: sin(theta) < 1e-5 snaps to the pi branch
        assert np.allclose(Rotation.from_rotvec(got).as_matrix(), R, atol=2e-5 if near_pi else 1e-7), (v, got)
        if np.linalg.norm(v) < 3.0:
            assert np.allclose(got, v, atol=1e-7), (v, got)


def test_folder_round_trip(tmp_path):
    from PIL import Image
    rs = np.random.RandomState(1)
    root = str(tmp_path)
    for d in ('img', 'ori_handdict', 'anno'):
        os.makedirs(os.path.join(root, 'test', d))
    dicts = []
    for i in range(3):
        yy, xx = np.mgrid[0:64, 0:64]
        rgb = np.stack([xx * 4, yy * 4, (xx + yy) * 2], -1).astype(np.uint8)           # smooth: JPEG keeps it within a few levels
        Image.fromarray(rgb).save(os.path.join(root, 'test', 'img', '%d.jpg' % i), quality=98)
        hd = {}
        for side in ('left', 'right'):
            q, _ = np.linalg.qr(rs.randn(3, 3))
            q *= np.sign(np.linalg.det(q))
            hd[side] = {'verts3d': rs.randn(778, 3).astype(np.float32), 'joints3d': rs.randn(21, 3).astype(np.float32),
                        'verts2d': rs.rand(778, 2).astype(np.float32) * 64, 'joints2d': rs.rand(21, 2).astype(np.float32) * 64,
                        'R': q.astype(np.float32), 'pose': rs.randn(45).astype(np.float32), 'shape': rs.randn(10).astype(np.float32)}
        np.save(os.path.join(root, 'test', 'ori_handdict', '%d.npy' % i), hd, allow_pickle=True)
        with open(os.path.join(root, 'test', 'anno', '%d.pkl' % i), 'wb') as f:
            pickle.dump({}, f)
        dicts.append((rgb, hd))
    ds = InterHandFolder(root, 'test')
    assert len(ds) == 3
    samples = [ds[i] for i in range(3)]
    for (img, hd), (rgb, src) in zip(samples, dicts):
        assert img.dtype == np.uint8 and img.shape == (64, 64, 3)
        assert np.abs(img[..., ::-1].astype(int) - rgb.astype(int)).max() <= 12          # BGR order, lossy codec
        assert hd['left']['pose'].shape == (48,) and hd['left']['pose'].dtype == np.float32
        assert np.array_equal(hd['right']['pose'][3:], src['right']['pose'])
    imgs, p2, p3 = collate_for_gpu(samples)
    assert imgs.shape == (3, 64, 64, 3) and p2.shape == (3, 1598, 2) and p3.shape == (3, 1598, 3)
    assert np.array_equal(p3[1, 778:799].numpy(), dicts[1][1]['left']['joints3d'])
    assert np.array_equal(p2[2, 799:1577].numpy(), dicts[2][1]['right']['verts2d'])
