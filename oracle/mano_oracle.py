"""ORACLE (test infrastructure, not product): CPU fp32 restatement of the reference MANO layer.

Follows /root/reference/models/manolayer.py (`ManoLayer.forward` :250-322, `rodrigues_batch` :32-48,
`pca2axis` :163-166) as plain PyTorch-CPU ops over a dict of constant tensors, so autograd gives the
gradient oracle as well.  Pinned by `tests/golden/mano_*.npz`, produced by the real reference
`ManoLayer` on a synthetic MANO-shaped pickle (`tests/golden/make_golden.py`).
Only tests/, smoke() and bench.py's cpu_baseline may import this.
"""
import numpy as np
import torch

NEW_ORDER = [0, 13, 14, 15, 16, 1, 2, 3, 17, 4, 5, 6, 18, 10, 11, 12, 19, 7, 8, 9, 20]   # manolayer.py:110-115
TIPS = [745, 317, 444, 556, 673]                                                          # manolayer.py:296


def constants_from_dict(d):
    """The buffers `ManoLayer.__init__` registers (manolayer.py:117-151), as float32 tensors."""
    c = {
        'hands_components': torch.from_numpy(np.asarray(d['hands_components'], np.float32)),
        'J_regressor': torch.from_numpy(np.asarray(d['J_regressor'].todense(), np.float32)),
        'weights': torch.from_numpy(np.asarray(d['weights'], np.float32)),
        'posedirs': torch.from_numpy(np.asarray(d['posedirs'], np.float32)),
        'v_template': torch.from_numpy(np.asarray(d['v_template'], np.float32)),
        'shapedirs': torch.from_numpy(np.asarray(d['shapedirs'], np.float32)),
        'hands_mean': torch.from_numpy(np.asarray(d['hands_mean'], np.float32)),
    }
    c['parent'] = [-1] + [int(d['kintree_table'][0, i]) for i in range(1, 16)]
    return c


def rodrigues(axis):
    """manolayer.py:32-48.  angle = ||axis|| + 1e-8 (epsilon added after the norm, note N7)."""
    bs = axis.shape[0]
    eye = torch.eye(3, dtype=axis.dtype).repeat(bs, 1, 1)
    angle = torch.norm(axis, p=2, dim=1, keepdim=True) + 1e-8
    a = axis / angle
    s = torch.sin(angle).unsqueeze(2)
    c = torch.cos(angle).unsqueeze(2)
    z = torch.zeros(bs, dtype=axis.dtype)
    K = torch.stack([torch.stack([z, -a[:, 2], a[:, 1]], 1),
                     torch.stack([a[:, 2], z, -a[:, 0]], 1),
                     torch.stack([-a[:, 1], a[:, 0], z], 1)], 1)
    return eye + s * K + (1 - c) * K.bmm(K)


def _se3(R, t):
    bs = R.shape[0]
    pad = torch.zeros((bs, 1, 4), dtype=R.dtype)
    pad[:, 0, 3] = 1.0
    return torch.cat([torch.cat([R, t], 2), pad], 1)


def mano_forward(c, root_rotation, pose, shape, trans=None, scale=None,
                 center_idx=9, use_pca=True, new_skel=False):
    """manolayer.py:250-322."""
    bs = root_rotation.shape[0]
    if use_pca:
        axis = pose.mm(c['hands_components'][:pose.shape[1]]) + c['hands_mean']
        rot = rodrigues(axis.view(-1, 3)).view(-1, 15, 3, 3)
    else:
        rot = pose
    v_shaped = c['v_template'] + torch.matmul(c['shapedirs'], shape.permute(1, 0)).permute(2, 0, 1)
    j = torch.matmul(c['J_regressor'], v_shaped)
    eye = torch.eye(3, dtype=rot.dtype)
    pose_shape = (rot - eye).reshape(bs, -1)
    v_t = v_shaped + torch.matmul(c['posedirs'], pose_shape.permute(1, 0)).permute(2, 0, 1)

    se3 = []
    R = root_rotation
    se3.append(_se3(R, (eye - R).bmm(j[:, 0].unsqueeze(2))))
    for i in range(1, 16):
        R = rot[:, i - 1]
        loc = _se3(R, (eye - R).bmm(j[:, i].unsqueeze(2)))
        se3.append(torch.matmul(se3[c['parent'][i]], loc))
    se3 = torch.stack(se3, 1)
    jl = [j[:, 0]]
    for i in range(1, 16):
        T = se3[:, c['parent'][i]]
        jl.append(T[:, :3, :3].bmm(j[:, i].unsqueeze(2))[:, :, 0] + T[:, :3, 3])
    sv = torch.matmul(c['weights'], se3.view(bs, 16, 16)).view(bs, -1, 4, 4)
    v = (sv[:, :, :3, :3].matmul(v_t.unsqueeze(3)) + sv[:, :, :3, 3:4])[..., 0]
    jo = torch.stack(jl + [v[:, t] for t in TIPS], 1)[:, NEW_ORDER]
    if center_idx is not None:
        ctr = jo[:, center_idx:center_idx + 1]
        v = v - ctr
        jo = jo - ctr
    if scale is not None:
        v = v * scale.view(-1, 1, 1)
        jo = jo * scale.view(-1, 1, 1)
    if trans is not None:
        v = v + trans.unsqueeze(1)
        jo = jo + trans.unsqueeze(1)
    if new_skel:
        jo = jo.clone()
        jo[:, 5] = (v[:, 63] + v[:, 144]) / 2
        jo[:, 9] = (v[:, 271] + v[:, 220]) / 2
        jo[:, 13] = (v[:, 148] + v[:, 290]) / 2
        jo[:, 17] = (v[:, 770] + v[:, 83]) / 2
    return v, jo
