"""On-disk sample format of the reference (dataset/interhand.py:230-268 `InterHand_dataset`):

    {data_path}/{split}/img/{i}.jpg              256 x 256 crop (read as BGR uint8, like cv.imread)
    {data_path}/{split}/ori_handdict/{i}.npy     pickled dict {'left' | 'right': {verts3d [778,3], joints3d [21,3],
                                                 verts2d [778,2], joints2d [21,2], R [3,3], pose [45], shape [10], ...}}
    {data_path}/{split}/anno/*.pkl               one per sample; only counted (dataset length)

`InterHandFolder` returns the reference's `(img, hand_dict)` per index -- including the 48-vector `pose` the reference builds by
prepending the axis-angle of the root rotation `R` (`cv.Rodrigues`) -- and `collate_for_gpu` turns a list of samples into the
three tensors `renderih_amd.input_pipeline.BatchPreparer` consumes, so a torch DataLoader only decodes JPEGs and stacks
arrays; augmentation and normalisation run on the GPU.  Host-side glue (numpy / PIL), no kernels.
JPEG decoding uses PIL (libjpeg), the reference OpenCV's bundled decoder: the same standard, not guaranteed bit-identical
pixels.
"""
import os
from glob import glob

import numpy as np
import torch

from .input_pipeline import pack_labels


def rotmat_to_axis_angle(R):
    """cv.Rodrigues(R)[0] for a rotation matrix: the rotation vector (log map), numpy float64 -> float64 [3]."""
    R = np.asarray(R, np.float64)
    u, _, vt = np.linalg.svd(R)                      # OpenCV orthonormalises its input the same way
    R = u @ vt
    r = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    s = np.linalg.norm(r) / 2
    c = np.clip((np.trace(R) - 1) / 2, -1.0, 1.0)
    theta = np.arccos(c)
    if s < 1e-5:
        if c > 0:
            return np.zeros(3)
        # theta = pi: the axis comes from the diagonal, signs from the off-diagonal terms
        ax = np.sqrt(np.maximum((np.diag(R) + 1) / 2, 0))
        if R[0, 1] < 0:
            ax[1] = -ax[1]
        if R[0, 2] < 0:
            ax[2] = -ax[2]
        if abs(ax[0]) < abs(ax[1]) and abs(ax[0]) < abs(ax[2]) and (R[1, 2] > 0) != (ax[1] * ax[2] > 0):
            ax[2] = -ax[2]
        return ax * (theta / max(np.linalg.norm(ax), 1e-30))
    return r * (theta / (2 * s))


def imread_bgr(path):
    from PIL import Image
    with Image.open(path) as im:
        rgb = np.asarray(im.convert('RGB'))
    return np.ascontiguousarray(rgb[..., ::-1])


class InterHandFolder:
    def __init__(self, data_path, split):
        assert split in ['train', 'test', 'val']
        self.data_path, self.split = data_path, split
        self.size = len(glob(os.path.join(data_path, split, 'anno', '*.pkl')))

    def __len__(self):
        return self.size

    def __getitem__(self, idx):
        img = imread_bgr(os.path.join(self.data_path, self.split, 'img', '{}.jpg'.format(idx)))
        hand_dict = np.load(os.path.join(self.data_path, self.split, 'ori_handdict', '{}.npy'.format(idx)),
                            allow_pickle=True)[()]
        for side in ('left', 'right'):
            root_pose = rotmat_to_axis_angle(hand_dict[side]['R']).reshape(-1)
            hand_dict[side]['pose'] = np.concatenate([root_pose, hand_dict[side]['pose']], axis=0).reshape(48).astype(np.float32)
        return img, hand_dict


def collate_for_gpu(samples):
    """[(img BGR uint8 [S,S,3], hand_dict)] -> (img_u8 [B,S,S,3] uint8, p2 [B,1598,2], p3 [B,1598,3]) CPU tensors (pin and copy
    them to the GPU, then call BatchPreparer)."""
    imgs = torch.from_numpy(np.stack([s[0] for s in samples]))
    p2, p3 = pack_labels([s[1] for s in samples])
    return imgs, p2, p3
