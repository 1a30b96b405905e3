"""GPU-resident evaluation metrics of the predicted hand meshes: the per-hand steps of
apps/eval_interhand.py:334-415 / common/utils/intag_eval.py:217-283 (`eval_hand2`) on one HIP kernel per hand
(`rih_hand_metrics`): joint regression, root alignment, bone-length rescaling, per-joint / per-vertex errors and the
Procrustes-aligned errors that the reference computes with torch.svd on the host."""
import torch

from . import ops
from ._lib import check


def joint_regressor_21(J_regressor):
    """`Jr.process_J_regressor` (intag_eval.py:14-28): the 16 MANO joints + 5 finger-tip vertices, re-ordered to the
    21-joint convention.  J_regressor: [16, 778] tensor."""
    J = J_regressor.clone().detach()
    tips = torch.zeros_like(J[:5])
    for i, v in enumerate((745, 317, 444, 556, 673)):
        tips[i, v] = 1.0
    J = torch.cat([J, tips], 0)
    order = [0, 13, 14, 15, 16, 1, 2, 3, 17, 4, 5, 6, 18, 10, 11, 12, 19, 7, 8, 9, 20]
    return J[order].contiguous()


def hand_metrics(verts_pred, verts_gt, Jreg=None, joints_pred=None, joints_gt=None, root_idx=0, bone=(9, 0)):
    """All tensors on the GPU, fp32: verts [B,V,3], Jreg [NJ,V] (needed when a joint set is not given), joints
    [B,NJ,3].  Returns a dict of GPU tensors: j_pred [B,NJ,3], j_err_ori / j_err [B,NJ], v_err_ori / v_err [B,V],
    pa_mpjpe / pa_mpvpe [B]."""
    ops._chk(verts_pred, verts_gt, Jreg, joints_pred, joints_gt)
    vp, vg = ops._c(verts_pred), ops._c(verts_gt)
    B, V, _ = vp.shape
    if Jreg is not None:
        Jreg = ops._c(Jreg)
        NJ = Jreg.shape[0]
    else:
        NJ = joints_pred.shape[1]
    jp = ops._c(joints_pred) if joints_pred is not None else None
    jg = ops._c(joints_gt) if joints_gt is not None else None
    dev = vp.device
    out = {'j_pred': torch.empty((B, NJ, 3), device=dev), 'j_err_ori': torch.empty((B, NJ), device=dev),
           'v_err_ori': torch.empty((B, V), device=dev), 'j_err': torch.empty((B, NJ), device=dev),
           'v_err': torch.empty((B, V), device=dev)}
    pa = torch.empty((B, 2), device=dev)
    check(ops._L().rih_hand_metrics(vp.data_ptr(), vg.data_ptr(), ops._p(jp), ops._p(jg), ops._p(Jreg), B, V, NJ, root_idx,
                                    bone[0], bone[1], out['j_err_ori'].data_ptr(), out['v_err_ori'].data_ptr(),
                                    out['j_err'].data_ptr(), out['v_err'].data_ptr(), pa.data_ptr(),
                                    out['j_pred'].data_ptr(), ops._stream()), 'rih_hand_metrics')
    out['pa_mpjpe'], out['pa_mpvpe'] = pa[:, 0], pa[:, 1]
    return out


def eval_hand2(verts_left_gt, verts_right_gt, joints_left_gt, joints_right_gt, verts_left_pred, verts_right_pred,
               joints_left_pred, joints_right_pred, joints_loss, verts_loss, pajoints_loss, paverts_loss):
    """Drop-in for common/utils/intag_eval.py:217-283: appends the per-joint / per-vertex error arrays of this batch to
    the caller's lists.  As in the reference, the `pa*` lists of this function receive the *rescaled* errors (its true
    Procrustes lines are commented out there, :268-281); use `hand_metrics` for PA-MPJPE / PA-MPVPE."""
    for side, vg, jg, vp, jp in (('left', verts_left_gt, joints_left_gt, verts_left_pred, joints_left_pred),
                                 ('right', verts_right_gt, joints_right_gt, verts_right_pred, joints_right_pred)):
        m = hand_metrics(vp, vg, joints_pred=jp, joints_gt=jg, root_idx=0, bone=(9, 0))
        je, ve = m['j_err'].detach().cpu().numpy(), m['v_err'].detach().cpu().numpy()
        joints_loss[side].append(je)
        verts_loss[side].append(ve)
        pajoints_loss[side].append(je)
        paverts_loss[side].append(ve)


def compute_cdev(pred_left, pred_right, gt_left, gt_right, contact=3e-3):
    """utils/eval_metrics.py:36-50 (`compute_cdev(pred_v3d_o, pred_v3d_r, gt_left, gt_right)`): [B] contact deviation in
    metres, NaN for samples whose ground-truth hands do not touch; one launch, no pytorch3d."""
    ops._chk(pred_left, pred_right, gt_left, gt_right)
    pl, pr, gl, gr = (ops._c(t) for t in (pred_left, pred_right, gt_left, gt_right))
    B, V, _ = gl.shape
    out = torch.empty(B, device=gl.device, dtype=torch.float32)
    check(ops._L().rih_cdev(pl.data_ptr(), pr.data_ptr(), gl.data_ptr(), gr.data_ptr(), B, V, float(contact), out.data_ptr(),
                            ops._stream()), 'rih_cdev')
    return out
