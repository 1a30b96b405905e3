"""Evaluation loop of apps/eval_interhand.py:298-470 on the GPU-resident metrics kernel: per batch the network forward, the
21-joint regression, root alignment at joint 0, bone (1, 0) rescaling, per-joint / per-vertex errors, Procrustes-aligned
errors (`renderih_amd.metrics.hand_metrics`, one launch per hand instead of ~60 torch ops and a host SVD) and the relative
root position error; the summary keys follow the reference's printed lines (millimetres).  MRRPE comes twice: `mrrpe` is the
Euclidean norm of the root-offset error per sample; `mrrpe_ref_script` reproduces the reference script's printed number
(mean absolute error per coordinate, up to sqrt(3) smaller -- its `.sum(axis=1)` runs over a size-1 joint axis).  The contact deviation `cdev`
(utils/eval_metrics.py:36-50) is one more launch per batch (`metrics.compute_cdev`, no pytorch3d).  The IoU-binned breakdown
of the script needs its side file (`iou_0_27w.npy`) and stays with the caller: `per_sample` carries the arrays it indexes.
"""
import numpy as np
import torch

from .metrics import compute_cdev, hand_metrics


@torch.no_grad()
def evaluate(network, batches, jreg_left, jreg_right, device=None):
    """network: HandNET_GCN (eval mode).  batches: iterable of (imgTensor, joints_left_gt, verts_left_gt, joints_right_gt,
    verts_right_gt) like the reference's handDataset (the joint tensors are ignored: the script regresses them from the
    vertices, eval_interhand.py:306-307).  jreg_*: [21,778] from `metrics.joint_regressor_21`.  Returns (summary, per_sample)."""
    acc = {k: {'left': [], 'right': []} for k in ('j_err_ori', 'v_err_ori', 'j_err', 'v_err', 'pa_mpjpe', 'pa_mpvpe')}
    pred_trans, gt_trans, cdev = [], [], []
    for data in batches:
        img, _, v_l, _, v_r = [t if device is None else t.to(device) for t in data[:5]]
        result = network(img)[0]
        roots = {}
        for side, vg, jr in (('left', v_l, jreg_left), ('right', v_r, jreg_right)):
            m = hand_metrics(result['verts3d'][side], vg.float(), Jreg=jr, root_idx=0, bone=(1, 0))
            for k in acc:
                acc[k][side].append(m[k].cpu().numpy())
            jg = torch.einsum('jv,bvc->bjc', jr, vg.float())
            roots[side] = (m['j_pred'][:, 0], jg[:, 0])
        cdev.append(compute_cdev(result['verts3d']['left'], result['verts3d']['right'], v_l.float(), v_r.float()).cpu().numpy())
        pred_trans.append((roots['left'][0] - roots['right'][0]).cpu().numpy())        # eval_interhand.py:416-417
        gt_trans.append((roots['left'][1] - roots['right'][1]).cpu().numpy())           # :322
    per = {k: {s: np.concatenate(v[s], 0) for s in v} for k, v in acc.items()}
    dtrans = np.concatenate(pred_trans, 0) - np.concatenate(gt_trans, 0)               # [N,3]
    mrrpe = np.sqrt((dtrans ** 2).sum(axis=1))                                         # Euclidean norm per sample
    per['mrrpe'] = mrrpe
    # what the reference script prints: it keeps the translations as [N,1,3] and sums over the size-1 joint axis
    # (eval_interhand.py:432-434), i.e. sqrt(d^2) = |d| per coordinate, then takes the mean over [N,3]
    per['mrrpe_ref_script'] = np.abs(dtrans)
    per['cdev'] = np.concatenate(cdev, 0)                                              # NaN: hands not in contact (:481-490)
    mm = lambda k: {s: float(per[k][s].mean() * 1000) for s in ('left', 'right')}
    summary = {}
    for name, key in (('ori joint mpjpe', 'j_err_ori'), ('ori vert mean error', 'v_err_ori'), ('joint mean error', 'j_err'),
                      ('vert mean error', 'v_err'), ('pa joint mean error', 'pa_mpjpe'), ('pa vert mean error', 'pa_mpvpe')):
        d = mm(key)
        d['all'] = (d['left'] + d['right']) / 2
        summary[name] = d
    summary['mrrpe'] = float(mrrpe.mean())                         # Euclidean MRRPE (the metric's definition)
    summary['mrrpe_ref_script'] = float(np.abs(dtrans).mean())     # the number apps/eval_interhand.py prints
    touching = ~np.isnan(per['cdev'])
    summary['cdev'] = float(per['cdev'][touching].mean()) if touching.any() else float('nan')
    return summary, per
