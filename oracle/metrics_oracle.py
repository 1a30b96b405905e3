"""ORACLE (test infrastructure, not product): CPU restatement of the reference's evaluation metrics for one hand.

Follows common/utils/intag_eval.py:92-143 (`batch_compute_similarity_transform_torch`, SVD Procrustes with the
reflection fix) and :217-283 (`eval_hand2`: root alignment at joint 0, rescaling by the 0-9 bone length, per-joint /
per-vertex L2 errors) -- the same steps as apps/eval_interhand.py:334-415, which regresses the joints from the vertices
with `Jr` first.  Pinned by tests/golden/metrics.npz, produced by calling those reference functions
(tests/golden/make_golden.py metrics).  Only tests/ may import this file."""
import torch


def similarity_transform(S1, S2):
    """intag_eval.py:92-143 on [B,N,3] inputs: S1 mapped onto S2 by the optimal (scale, rotation, translation)."""
    S1, S2 = S1.permute(0, 2, 1), S2.permute(0, 2, 1)
    mu1, mu2 = S1.mean(-1, keepdim=True), S2.mean(-1, keepdim=True)
    X1, X2 = S1 - mu1, S2 - mu2
    var1 = (X1 ** 2).sum(1).sum(1)
    K = X1.bmm(X2.permute(0, 2, 1))
    U, s, Vh = torch.linalg.svd(K)
    V = Vh.transpose(1, 2)
    Z = torch.eye(3, dtype=S1.dtype).unsqueeze(0).repeat(U.shape[0], 1, 1)
    Z[:, -1, -1] *= torch.sign(torch.det(U.bmm(V.permute(0, 2, 1))))
    R = V.bmm(Z.bmm(U.permute(0, 2, 1)))
    scale = torch.stack([torch.trace(x) for x in R.bmm(K)]) / var1
    t = mu2 - scale.view(-1, 1, 1) * R.bmm(mu1)
    return (scale.view(-1, 1, 1) * R.bmm(S1) + t).permute(0, 2, 1)


def hand_metrics(v_pred, v_gt, Jreg=None, j_pred=None, j_gt=None, root_idx=0, bone=(9, 0)):
    """eval_hand2 (intag_eval.py:217-283) for one hand + the PA errors of apps/eval_interhand.py:388-407."""
    if j_pred is None:
        j_pred = torch.matmul(Jreg, v_pred)                   # Jr.__call__, intag_eval.py:30-31
    if j_gt is None:
        j_gt = torch.matmul(Jreg, v_gt)
    root_g, root_p = j_gt[:, root_idx:root_idx + 1], j_pred[:, root_idx:root_idx + 1]
    len_g = torch.linalg.norm(j_gt[:, bone[0]] - j_gt[:, bone[1]], dim=-1)
    len_p = torch.linalg.norm(j_pred[:, bone[0]] - j_pred[:, bone[1]], dim=-1)
    sc = (len_g / len_p).view(-1, 1, 1)
    jg, vg = j_gt - root_g, v_gt - root_g
    jp, vp = j_pred - root_p, v_pred - root_p
    out = {'j_pred': j_pred,
           'j_err_ori': torch.linalg.norm(jp - jg, dim=-1), 'v_err_ori': torch.linalg.norm(vp - vg, dim=-1),
           'j_err': torch.linalg.norm(jp * sc - jg, dim=-1), 'v_err': torch.linalg.norm(vp * sc - vg, dim=-1)}
    out['pa_mpjpe'] = torch.sqrt(((similarity_transform(jp, jg) - jg) ** 2).sum(-1)).mean(-1)
    out['pa_mpvpe'] = torch.sqrt(((similarity_transform(vp, vg) - vg) ** 2).sum(-1)).mean(-1)
    return out


def compute_cdev(pred_v3d_o, pred_v3d_r, gt_left, gt_right, contact_dist=3e-3):
    """utils/eval_metrics.py:30-50 (`compute_idx` + `compute_cdev`) with pytorch3d's `knn_points(K=1)` -- a third-party
    dependency that is absent here -- restated as the exhaustive nearest neighbour (squared distances, first minimum):
    everything else follows the reference line by line.  tests/test_metrics.py additionally runs the reference's own
    `compute_cdev` with a stub `pytorch3d.ops.knn_points` of that definition."""
    d2 = ((gt_right[:, :, None, :] - gt_left[:, None, :, :]) ** 2).sum(-1)
    dist_ro, idx_ro = d2.min(dim=2)
    dist_ro = dist_ro.sqrt()
    vo = torch.gather(pred_v3d_o, 1, idx_ro[:, :, None].repeat(1, 1, 3))
    disp = vo - pred_v3d_r
    disp[dist_ro > contact_dist] = float('nan')
    cd = (disp ** 2).sum(dim=2).sqrt()
    nan = torch.isnan(cd)
    cd = cd.clone()
    cd[nan] = 0
    return cd.sum(1) / (~nan).float().sum(1)
