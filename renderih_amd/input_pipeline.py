"""Batch input preparation on the GPU -- the counterpart of the reference's per-sample `handDataset.process_data`
(core/loader.py:104-219) for crops that are already decoded and resident in HBM.

The reference prepares one sample at a time on a CPU worker: affine augmentation (`cv.warpAffine` + the same map on the 2-D
labels, the in-plane rotation on the 3-D labels), brightness, flip with left/right swap, BGR->RGB, /255, ImageNet
normalisation, root-relative + bone-length label normalisation.  `BatchPreparer` keeps that contract -- same random draws in
the same order (`augm_params`, `brightness_params`), same 11 outputs, now with a leading batch dimension -- and does the
per-pixel and per-point work in two kernel launches (csrc/rih_input.hip).  The 3x3 matrices are built on the host with the
reference's own float32 recipe (utils/manoutils.py:138-194, pi = 3.14159 included), nine numbers per sample.

    prep = BatchPreparer(theta=(-90, 90), scale=(0.75, 1.25), uv=(-10, 10), train=True)
    out = prep(img_u8, p2, p3)          # img_u8 [B,S,S,3] uint8 BGR, p2 [B,1598,2], p3 [B,1598,3] (see pack_labels)
    out.imgTensor, out.v3d_l, ...       # or tuple(out): the reference's order

STATUS: harness-verified against fixtures made by the reference's own process_data; not yet run on a GPU.
"""
import math
import random
from collections import namedtuple

import numpy as np
import torch

from . import ops
from .ops import check

BONE_LENGTH = 0.095         # dataset/dataset_utils.py:9
NV, NJ, ROOT = 778, 21, 9

Prepared = namedtuple('Prepared', 'ori_img imgTensor v2d_l j2d_l v2d_r j2d_r v3d_l j3d_l v3d_r j3d_r root_rel')


def _about(center, m):
    t = np.matmul((np.identity(3, dtype='float32') - m), center)
    m[0, 2] = t[0]
    m[1, 2] = t[1]
    return m


def rotation_mat3d(theta):
    """utils/manoutils.py:171-180."""
    t = theta * (3.14159 / 180)
    r = np.zeros((3, 3), dtype='float32')
    r[0, 0] = r[1, 1] = math.cos(t)
    r[0, 1] = -math.sin(t)
    r[1, 0] = math.sin(t)
    r[2, 2] = 1.0
    return r


def affine_mat(theta, scale, u, v, height, width):
    """utils/manoutils.py:182-194 (`imgUtils.get_affine_mat`), float32 like the reference."""
    center = np.array([width / 2, height / 2, 1], dtype='float32')
    rot = _about(center, rotation_mat3d(theta))
    sc = np.zeros((3, 3), dtype='float32')
    sc[0, 0] = sc[1, 1] = scale
    sc[2, 2] = 1.0
    sc = _about(center, sc)
    trans = np.identity(3, dtype='float32')
    trans[0, 2] = u
    trans[1, 2] = v
    return np.matmul(trans, np.matmul(sc, rot))


def invert_for_warp(M):
    """What cv.warpAffine does to a forward 2x3 matrix before walking the destination (double precision)."""
    m = np.asarray(M, np.float64).reshape(6).copy()
    D = m[0] * m[4] - m[1] * m[3]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = m[4] * D, m[0] * D
    m[0], m[4] = A11, A22
    m[1] *= -D
    m[3] *= -D
    b1 = -m[0] * m[2] - m[1] * m[5]
    b2 = -m[3] * m[2] - m[4] * m[5]
    m[2], m[5] = b1, b2
    return m


def pack_labels(hand_dicts):
    """List of the reference's hand dicts ({'left'|'right': {verts2d, joints2d, verts3d, joints3d}}) ->
    (p2 [B,1598,2], p3 [B,1598,3]) float32 CPU tensors, points ordered [verts_left, joints_left, verts_right, joints_right]."""
    p2 = np.stack([np.concatenate([d[s][k] for s in ('left', 'right') for k in ('verts2d', 'joints2d')]) for d in hand_dicts])
    p3 = np.stack([np.concatenate([d[s][k] for s in ('left', 'right') for k in ('verts3d', 'joints3d')]) for d in hand_dicts])
    return torch.from_numpy(p2.astype(np.float32)), torch.from_numpy(p3.astype(np.float32))


class BatchPreparer:
    def __init__(self, theta=(-90, 90), scale=(0.75, 1.25), uv=(-10, 10), flip=True, train=True, bone_length=BONE_LENGTH,
                 brightness=(0.3, 0.05), fp16_nhwc8=False):
        self.theta, self.scale, self.uv, self.flip, self.train = theta, scale, uv, flip, train
        self.bone_length, self.brightness, self.fp16_nhwc8 = bone_length, brightness, fp16_nhwc8

    def augm_params(self):
        """core/loader.py:96-102: five `random.random()` draws in the reference's order."""
        theta = random.random() * (self.theta[1] - self.theta[0]) + self.theta[0]
        scale = random.random() * (self.scale[1] - self.scale[0]) + self.scale[0]
        u = random.random() * (self.uv[1] - self.uv[0]) + self.uv[0]
        v = random.random() * (self.uv[1] - self.uv[0]) + self.uv[0]
        flip = random.random() > 0.5 if self.flip else False
        return theta, scale, u, v, flip

    def brightness_params(self):
        """utils/manoutils.py:253-260 (`add_noise`, noise = 0): per-channel gain, then one offset."""
        alpha, beta = self.brightness
        a = np.random.uniform(1 - alpha, 1 + alpha, 3)
        b = 255.0 * beta * (2 * random.random() - 1)
        return a, b

    def __call__(self, img_u8, p2, p3, params=None, bright=None):
        """img_u8 [B,S,S,3] uint8 BGR, p2 [B,NP,2], p3 [B,NP,3] fp32 (pack_labels), all on the GPU.  params / bright:
        explicit per-sample draws (lists of length B); drawn like the reference when omitted and `train`."""
        ops._chk(img_u8, dtype=torch.uint8)
        ops._chk(p2, p3)
        B, S, S2, Cc = img_u8.shape
        assert S == S2 and Cc == 3 and p2.shape[0] == B and p3.shape[0] == B
        NP = p2.shape[1]
        assert NP == 2 * (NV + NJ) and p2.shape == (B, NP, 2) and p3.shape == (B, NP, 3)
        dev = img_u8.device
        img_u8, p2, p3 = img_u8.contiguous(), p2.contiguous(), p3.contiguous()
        minv = brt = flp = A = R = None
        if self.train:
            if params is None:
                drawn = []
                for _ in range(B):                  # per sample: geometry draws, then brightness draws (reference order)
                    g = self.augm_params()
                    drawn.append((g, self.brightness_params()))
                params = [d[0] for d in drawn]
                bright = [d[1] for d in drawn]
            mats = [affine_mat(t, s, u, v, S, S) for (t, s, u, v, _f) in params]
            A = torch.from_numpy(np.stack([m[0:2, :].reshape(6) for m in mats]).astype(np.float32)).to(dev)
            R = torch.from_numpy(np.stack([rotation_mat3d(p[0]).reshape(9) for p in params]).astype(np.float32)).to(dev)
            minv = torch.from_numpy(np.stack([invert_for_warp(m[0:2, :]) for m in mats])).to(dev)
            flp = torch.tensor([1 if p[4] else 0 for p in params], dtype=torch.uint8).to(dev)
            if bright is not None:
                brt = torch.from_numpy(np.stack([np.concatenate([np.asarray(a, np.float64), [float(b)]])
                                                 for (a, b) in bright])).to(dev)
        ori = torch.empty((B, 3, S, S), device=dev, dtype=torch.float32)
        norm = torch.empty((B, 3, S, S), device=dev, dtype=torch.float32)
        h8 = torch.empty((B, S, S, 8), device=dev, dtype=torch.float16) if self.fp16_nhwc8 else None
        p = ops._p
        check(ops._L().rih_prepare_images(img_u8.data_ptr(), B, S, p(minv), p(brt), p(flp), 0, ori.data_ptr(), norm.data_ptr(),
                                          p(h8), ops._stream()), 'rih_prepare_images')
        o2, o3 = torch.empty_like(p2), torch.empty_like(p3)
        root_rel = torch.empty((B, 3), device=dev, dtype=torch.float32)
        check(ops._L().rih_prepare_labels(p2.data_ptr(), p3.data_ptr(), B, NV, NJ, p(A), p(R), p(flp),
                                          float(self.bone_length or 0.0), ROOT, float(S), o2.data_ptr(), o3.data_ptr(),
                                          root_rel.data_ptr(), ops._stream()), 'rih_prepare_labels')
        NH = NV + NJ
        out = Prepared(ori, norm, o2[:, :NV], o2[:, NV:NH], o2[:, NH:NH + NV], o2[:, NH + NV:],
                       o3[:, :NV], o3[:, NV:NH], o3[:, NH:NH + NV], o3[:, NH + NV:], root_rel)
        return (out, h8) if self.fp16_nhwc8 else out
