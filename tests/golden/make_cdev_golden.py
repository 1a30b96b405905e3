"""tests/golden/cdev.npz: inputs and outputs of the REFERENCE's `utils/eval_metrics.py::compute_cdev` (/root/reference) on
synthetic touching / non-touching hand pairs.  pytorch3d is absent here: `pytorch3d.ops.knn_points` is stubbed with the
exhaustive K = 1 nearest neighbour on squared distances (its definition); everything else is the reference's own code.
    python tests/golden/make_cdev_golden.py"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def knn_points(p1, p2, l1, l2, K=1, return_nn=True):
    d2 = ((p1[:, :, None, :] - p2[:, None, :, :]) ** 2).sum(-1)
    dist, idx = d2.min(dim=2)
    return dist[:, :, None], idx[:, :, None], None


def main():
    p3d = types.ModuleType('pytorch3d')
    ops = types.ModuleType('pytorch3d.ops')
    ops.knn_points = knn_points
    p3d.ops = ops
    sys.modules['pytorch3d'] = p3d
    sys.modules['pytorch3d.ops'] = ops
    sys.path.insert(0, '/root/reference')
    from utils.eval_metrics import compute_cdev
    g = torch.Generator().manual_seed(21)
    B, V = 4, 778
    gt_r = torch.randn(B, V, 3, generator=g) * 0.03
    gt_l = torch.randn(B, V, 3, generator=g) * 0.03 + torch.tensor([[0.02, 0, 0], [0.05, 0, 0], [0.5, 0, 0], [0.0, 0.01, 0]])[:, None]
    pr = gt_r + 0.004 * torch.randn(B, V, 3, generator=g)
    pl = gt_l + 0.004 * torch.randn(B, V, 3, generator=g)
    out = compute_cdev(pl.clone(), pr.clone(), gt_l.clone(), gt_r.clone())
    np.savez_compressed(os.path.join(HERE, 'cdev.npz'), pred_left=pl.numpy(), pred_right=pr.numpy(), gt_left=gt_l.numpy(),
                        gt_right=gt_r.numpy(), cdev=out.numpy())
    print('cdev', out)


if __name__ == '__main__':
    main()
