"""ORACLE (test infrastructure, not product): torch restatement of the MANO parameter head of the reference's
`load_new_model` network (common/myhand/decoder_lijun_mano.py) -- rotation conversions, ParamRegressor, MANO-in-the-
forward with bone-length normalisation.  Differentiable through autograd (the gradient oracle).  Pinned by
tests/golden/pose_head.npz / net_newlijun_*.npz, produced from the real reference functions (make_golden.py newmodel).
Only tests/ may import this file."""
import torch
import torch.nn.functional as F


def rot6d_to_rotmat(x):
    """decoder_lijun_mano.py:118-125 (ParamRegressor.rot6d_to_rotmat)."""
    x = x.view(-1, 3, 2)
    a1, a2 = x[:, :, 0], x[:, :, 1]
    b1 = F.normalize(a1)
    b2 = F.normalize(a2 - torch.einsum('bi,bi->b', b1, a2).unsqueeze(-1) * b1)
    b3 = torch.cross(b1, b2, dim=1)
    return torch.stack((b1, b2, b3), dim=-1)


def rotation_matrix_to_quaternion(R, eps=1e-6):
    """common/myhand/utils/comm.py:280-323 on [N,3,3] (the reference appends an unused 4th column)."""
    m = R.transpose(1, 2)
    d2 = m[:, 2, 2] < eps
    d0_d1 = m[:, 0, 0] > m[:, 1, 1]
    d0_nd1 = m[:, 0, 0] < -m[:, 1, 1]
    t0 = 1 + m[:, 0, 0] - m[:, 1, 1] - m[:, 2, 2]
    q0 = torch.stack([m[:, 1, 2] - m[:, 2, 1], t0, m[:, 0, 1] + m[:, 1, 0], m[:, 2, 0] + m[:, 0, 2]], -1)
    t1 = 1 - m[:, 0, 0] + m[:, 1, 1] - m[:, 2, 2]
    q1 = torch.stack([m[:, 2, 0] - m[:, 0, 2], m[:, 0, 1] + m[:, 1, 0], t1, m[:, 1, 2] + m[:, 2, 1]], -1)
    t2 = 1 - m[:, 0, 0] - m[:, 1, 1] + m[:, 2, 2]
    q2 = torch.stack([m[:, 0, 1] - m[:, 1, 0], m[:, 2, 0] + m[:, 0, 2], m[:, 1, 2] + m[:, 2, 1], t2], -1)
    t3 = 1 + m[:, 0, 0] + m[:, 1, 1] + m[:, 2, 2]
    q3 = torch.stack([t3, m[:, 1, 2] - m[:, 2, 1], m[:, 2, 0] - m[:, 0, 2], m[:, 0, 1] - m[:, 1, 0]], -1)
    c0, c1, c2, c3 = ((d2 & d0_d1), (d2 & ~d0_d1), (~d2 & d0_nd1), (~d2 & ~d0_nd1))
    c0, c1, c2, c3 = (c.view(-1, 1).type_as(q0) for c in (c0, c1, c2, c3))
    q = q0 * c0 + q1 * c1 + q2 * c2 + q3 * c3
    q = q / torch.sqrt(t0.view(-1, 1) * c0 + t1.view(-1, 1) * c1 + t2.view(-1, 1) * c2 + t3.view(-1, 1) * c3)
    return q * 0.5


def quaternion_to_angle_axis(q):
    """comm.py:227-247."""
    q1, q2, q3 = q[..., 1], q[..., 2], q[..., 3]
    s2 = q1 * q1 + q2 * q2 + q3 * q3
    s = torch.sqrt(s2)
    c = q[..., 0]
    two_theta = 2.0 * torch.where(c < 0.0, torch.atan2(-s, -c), torch.atan2(s, c))
    k = torch.where(s2 > 0.0, two_theta / s, 2.0 * torch.ones_like(s))
    return torch.stack([q1 * k, q2 * k, q3 * k], -1)


def rotation_matrix_to_angle_axis(R):
    """comm.py:176-200."""
    aa = quaternion_to_angle_axis(rotation_matrix_to_quaternion(R))
    return torch.where(torch.isnan(aa), torch.zeros_like(aa), aa)


def rodrigues_batch(axis):
    """common/utils/manolayer.py:32-48."""
    bs = axis.shape[0]
    I = torch.eye(3, dtype=axis.dtype).repeat(bs, 1, 1)
    angle = torch.norm(axis, p=2, dim=1, keepdim=True) + 1e-8
    e = axis / angle
    sin, cos = torch.sin(angle).unsqueeze(2), torch.cos(angle).unsqueeze(2)
    z = torch.zeros_like(e[:, 0])
    L = torch.stack([torch.stack([z, -e[:, 2], e[:, 1]], -1), torch.stack([e[:, 2], z, -e[:, 0]], -1),
                     torch.stack([-e[:, 1], e[:, 0], z], -1)], 1)
    return I + sin * L + (1 - cos) * L.bmm(L)
