"""Generates tests/golden/input_pipeline.npz by executing the REFERENCE's own per-sample input preparation
(`core/loader.py::handDataset.process_data`, /root/reference) on synthetic samples.  Run in the build container only.

Import stubs (the packages are absent here): cv2 (flip, cvtColor = exact index operations; warpAffine = the restatement in
oracle/input_oracle.py -- so the warp itself is not pinned by these fixtures, everything around it is), torchvision.transforms
(Normalize = sub mean, div std, as torchvision does), imgaug (unused: `self.seq` is None), and the dataset classes the
loader imports but process_data never touches.
    python tests/golden/make_input_golden.py
"""
import os
import random
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, '..', '..'))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from oracle import input_oracle            # noqa: E402
import ref_stubs                           # noqa: E402

REF = '/root/reference'


def install():
    ref_stubs.install()
    cv = sys.modules['cv2']
    cv.COLOR_BGR2RGB = 4
    cv.INTER_LINEAR = 1
    cv.BORDER_REPLICATE = 1
    cv.BORDER_CONSTANT = 0
    cv.flip = lambda img, code: np.ascontiguousarray(img[:, ::-1]) if code == 1 else (_ for _ in ()).throw(NotImplementedError())
    cv.cvtColor = lambda img, code: np.ascontiguousarray(img[..., ::-1])
    cv.warpAffine = lambda src, M, dsize, **kw: input_oracle.warp_affine_u8(src, M, dsize)
    tv = sys.modules['torchvision']
    tr = types.ModuleType('torchvision.transforms')

    class Normalize:
        def __init__(self, mean, std):
            self.mean, self.std = torch.tensor(mean).view(-1, 1, 1), torch.tensor(std).view(-1, 1, 1)

        def __call__(self, t):
            return (t - self.mean) / self.std
    tr.Normalize = Normalize
    tv.transforms = tr
    sys.modules['torchvision.transforms'] = tr
    ia = types.ModuleType('imgaug')
    iaa = types.ModuleType('imgaug.augmenters')
    ia.augmenters = iaa
    sys.modules['imgaug'] = ia
    sys.modules['imgaug.augmenters'] = iaa
    # dataset classes imported at module level by core/loader.py; process_data uses none of them
    for name in ('interhand', 'interhand_withother', 'interhand_orisyn', 'interhand_subset',
                 'interhand_fullsyn_realsubset', 'interhand_realsubset'):
        m = types.ModuleType('dataset.' + name)
        for cls in ('InterHand_dataset', 'InterHand_other', 'InterHand_orisyn', 'InterHand_subset', 'InterHand_mixsubset',
                    'InterHand_realsubset'):
            setattr(m, cls, object)
        sys.modules['dataset.' + name] = m
    sys.path.insert(0, REF)


def sample(seed, S=256):
    rs = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:S, 0:S]
    base = (np.stack([xx, yy, (xx + yy) // 2], -1) % 256).astype(np.uint8)
    img = ((base.astype(np.int32) + rs.randint(0, 64, (S, S, 3))) % 256).astype(np.uint8)
    hd = {}
    for side, off in (('left', -0.05), ('right', 0.05)):
        j3 = (rs.randn(21, 3) * 0.04 + np.array([off, 0, 0.6])).astype(np.float32)
        v3 = (rs.randn(778, 3) * 0.04 + np.array([off, 0, 0.6])).astype(np.float32)
        hd[side] = {'verts3d': v3, 'joints3d': j3,
                    'verts2d': (rs.rand(778, 2) * S).astype(np.float32), 'joints2d': (rs.rand(21, 2) * S).astype(np.float32)}
    return img, hd


CASES = [   # name, train, (theta, scale, u, v, flip), image side (small sides keep the fixture small)
    ('eval', False, None, 64),
    ('train_plain', True, (0.0, 1.0, 0.0, 0.0, False), 48),
    ('train_rot', True, (37.5, 1.1, 4.0, -7.5, False), 128),
    ('train_flip', True, (-80.0, 0.8, -10.0, 10.0, True), 64),
    ('train_big', True, (90.0, 1.25, 9.9, 3.3, True), 64),
]


def main():
    install()
    from core.loader import handDataset, BONE_LENGTH
    out = {'case_names': np.array([c[0] for c in CASES])}
    for ci, (name, train, params, S) in enumerate(CASES):
        img, hd = sample(100 + ci, S)
        ds = object.__new__(handDataset)
        ds.train, ds.seq, ds.noise, ds.bone_length, ds.flip = train, None, 0.0, BONE_LENGTH, True
        from torchvision.transforms import Normalize
        ds.normalize_img = Normalize(mean=[0.485, 0.456, 0.406], std=[0.229, 0.224, 0.225])
        ds.augm_params = (lambda p: (lambda: p))(params)
        np.random.seed(7 + ci)
        random.seed(70 + ci)
        a = np.random.uniform(1 - 0.3, 1 + 0.3, 3)              # the draws add_noise will make with these seeds
        b = 255.0 * 0.05 * (2 * random.random() - 1)
        np.random.seed(7 + ci)
        random.seed(70 + ci)
        res = ds.process_data(img.copy(), {s: {k: v.copy() for k, v in hd[s].items()} for s in hd})
        out[name + '.img'] = img
        for s in ('left', 'right'):
            for k, v in hd[s].items():
                out['%s.in.%s.%s' % (name, s, k)] = v
        out[name + '.params'] = np.array([0, 1, 0, 0, 0] if params is None else [float(x) for x in params], np.float64)
        out[name + '.bright'] = np.concatenate([a, [b]])
        for i, t in enumerate(res):
            out['%s.out.%d' % (name, i)] = t.numpy()
    path = os.path.join(HERE, 'input_pipeline.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path) // 1024, 'KiB')


if __name__ == '__main__':
    main()
