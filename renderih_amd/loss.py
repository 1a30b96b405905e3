"""Training loss of the graph model as plain torch ops on the GPU (host-side mirror of core/Loss.py).

SURVEY.md 8a row a15 / 8f-2: the loss sits right after the hot path; its FLOPs are negligible, so round 1 keeps it
as PyTorch elementwise ops (fusing it into HIP kernels is the "next" row).  Semantics follow
core/Loss.py:20-164 (GraphLoss) and :201-277 (calc_loss_GCN): SmoothL1 on 3-D vertices and regressed joints,
MSE on normalised 2-D vertices, face-normal and edge-length terms, the same at the coarse (252-vertex) level, aux
(hms/mask/dense) loss disabled, edge term gated by epoch >= NORM_EPOCH, right hand shifted by root_rel.
"""
import numpy as np
import torch
import torch.nn.functional as F

NEW_ORDER = [0, 13, 14, 15, 16, 1, 2, 3, 17, 4, 5, 6, 18, 10, 11, 12, 19, 7, 8, 9, 20]
TIPS = [745, 317, 444, 556, 673]

DEFAULT_WEIGHTS = {'LABEL_3D': 100.0, 'LABEL_2D': 50.0, 'NORMAL': 10.0, 'EDGE': 2000.0, 'NORM_EPOCH': 50,
                   'UPSAMPLE': 1.0}


def joint_regressor_21(J_regressor):
    """core/Loss.py:38-53: 16 MANO joints + 5 one-hot finger tips, reordered to the 21-joint convention."""
    tips = torch.zeros((5, J_regressor.shape[1]), dtype=J_regressor.dtype, device=J_regressor.device)
    for i, v in enumerate(TIPS):
        tips[i, v] = 1.0
    return torch.cat([J_regressor, tips], 0)[NEW_ORDER].contiguous()


class GraphLoss:
    def __init__(self, J_regressor, faces, level=4, device='cuda', upsample_weight=None):
        self.device = device
        self.level = level + 1
        self.J_regressor = joint_regressor_21(J_regressor.clone().detach().float()).to(device)
        self.faces = torch.from_numpy(np.asarray(faces).astype(np.int64)).to(device)
        self.upsample_weight = None if upsample_weight is None else upsample_weight.to(device)

    @staticmethod
    def _edges(v, faces):
        t = v[:, faces]                                             # B x F x 3 x 3
        return torch.stack([t[:, :, 0] - t[:, :, 1], t[:, :, 1] - t[:, :, 2], t[:, :, 2] - t[:, :, 0]], dim=2)

    def norm_loss(self, pred, gt):
        eg, ep = self._edges(gt, self.faces), self._edges(pred, self.faces)
        n = F.normalize(torch.cross(eg[:, :, 0], eg[:, :, 1], dim=-1), dim=-1).unsqueeze(2)
        d = torch.sum(F.normalize(ep, dim=-1) * n, dim=-1)
        return F.smooth_l1_loss(d, torch.zeros_like(d))

    def edge_loss(self, pred, gt):
        lg = torch.linalg.norm(self._edges(gt, self.faces), dim=-1)
        lp = torch.linalg.norm(self._edges(pred, self.faces), dim=-1)
        return F.smooth_l1_loss(lp, lg)

    def calc_mano_loss(self, v3d_pred, v2d_pred, v3d_gt, v2d_gt, img_size):
        return {'vert2d_loss': F.mse_loss(v2d_pred / img_size * 2 - 1, v2d_gt / img_size * 2 - 1),
                'vert3d_loss': F.smooth_l1_loss(v3d_pred, v3d_gt),
                'joint_loss': F.smooth_l1_loss(torch.matmul(self.J_regressor, v3d_pred),
                                               torch.matmul(self.J_regressor, v3d_gt)),
                'norm_loss': self.norm_loss(v3d_pred, v3d_gt),
                'edge_loss': self.edge_loss(v3d_pred, v3d_gt)}

    @staticmethod
    def _down(x, p=2):
        B, V, D = x.shape
        return x[:, :V // p * p].reshape(B, V // p, p, D).mean(2)   # AvgPool1d(p) along V (floor, like torch)

    def calc_loss(self, converter, v3d_gt, v2d_gt, v3d_pred, v2d_pred, v3dList, v2dList, img_size):
        mano = self.calc_mano_loss(v3d_pred, v2d_pred, v3d_gt, v2d_gt, img_size)
        g3, g2 = converter.vert_to_GCN(v3d_gt), converter.vert_to_GCN(v2d_gt)
        l3, l2 = [], []
        for _ in range(self.level):
            l3.append(g3)
            l2.append(g2)
            g3, g2 = self._down(g3), self._down(g2)
        coarse = {'v3d_loss': [], 'v2d_loss': []}
        for p3, p2 in zip(v3dList, v2dList):
            j = [t.shape[1] for t in l3].index(p3.shape[1])
            coarse['v3d_loss'].append(F.smooth_l1_loss(p3, l3[j]))
            coarse['v2d_loss'].append(F.mse_loss(p2 / img_size * 2 - 1, l2[j] / img_size * 2 - 1))
        return mano, coarse


def calc_loss_GCN(weights, epoch, loss_left, loss_right, converter_left, converter_right, result, paramsDict,
                  handDictList, otherInfo, v2d_l, v2d_r, v3d_l, v3d_r, root_rel, img_size=256, upsample_weight=None):
    """core/Loss.py:201-277 (aux loss disabled there, :211)."""
    w = dict(DEFAULT_WEIGHTS)
    w.update(weights or {})
    v3d_r = v3d_r + root_rel.unsqueeze(1)
    out = {}
    for side, gl, conv, v3, v2 in (('left', loss_left, converter_left, v3d_l, v2d_l),
                                   ('right', loss_right, converter_right, v3d_r, v2d_r)):
        out[side] = gl.calc_loss(conv, v3, v2, result['verts3d'][side], result['verts2d'][side],
                                 [h['verts3d'][side] for h in handDictList],
                                 [h['verts2d'][side] for h in handDictList], img_size)
    mano = {k: (out['left'][0][k] + out['right'][0][k]) / 2 for k in out['left'][0]}
    alpha = 0 if epoch < w['NORM_EPOCH'] else 1
    total = w['LABEL_3D'] * mano['vert3d_loss'] + w['LABEL_2D'] * mano['vert2d_loss'] + \
        w['LABEL_3D'] * mano['joint_loss'] + w['NORMAL'] * mano['norm_loss'] + alpha * w['EDGE'] * mano['edge_loss']
    for i in range(len(out['left'][1]['v3d_loss'])):
        total = total + w['LABEL_3D'] * (out['left'][1]['v3d_loss'][i] + out['right'][1]['v3d_loss'][i]) / 2 \
            + w['LABEL_2D'] * (out['left'][1]['v2d_loss'][i] + out['right'][1]['v2d_loss'][i]) / 2
    if upsample_weight is not None and loss_left.upsample_weight is not None:
        total = total + w['UPSAMPLE'] * F.smooth_l1_loss(upsample_weight - loss_left.upsample_weight,
                                                         torch.zeros_like(upsample_weight))
    return total, mano


# ------------------------------------------------------------------------------------------------ fused HIP loss
class _MeshLossFn(torch.autograd.Function):
    """Total loss of calc_loss_GCN for both hands in three launches (rih_mesh_loss x 2 + rih_mesh_loss_final); the
    kernel already produced the gradients, backward only scales them by the incoming gradient."""

    @staticmethod
    def forward(ctx, fused, v3l, v2l, c3l, c2l, v3r, v2r, c3r, c2r, gt3l, gt2l, gt3r, gt2r, root_rel):
        import ctypes as C
        from . import _lib
        from .ops import _stream, check
        lib = _lib.load()
        B = v3l.shape[0]
        wa, ca = fused.device_weights(B, v3l.device)     # device-resident: a captured graph follows set_epoch()
        preds = [t.contiguous() for t in (v3l, v2l, c3l, c2l, v3r, v2r, c3r, c2r)]
        grads = [torch.empty_like(t) for t in preds]
        parts = torch.empty((2, B, 8), device=v3l.device, dtype=torch.float32)
        out = torch.empty((8,), device=v3l.device, dtype=torch.float32)
        for h, (gt3, gt2, shift) in enumerate(((gt3l, gt2l, None), (gt3r, gt2r, root_rel))):
            p, g = preds[4 * h:4 * h + 4], grads[4 * h:4 * h + 4]
            topo = fused.topo('left' if h == 0 else 'right', v3l.device)
            check(lib.rih_mesh_loss(C.byref(topo), p[0].data_ptr(), p[1].data_ptr(), p[2].data_ptr(), p[3].data_ptr(),
                                    gt3.contiguous().data_ptr(), gt2.contiguous().data_ptr(),
                                    0 if shift is None else shift.contiguous().data_ptr(), wa.data_ptr(),
                                    float(fused.img_size),
                                    g[0].data_ptr(), g[1].data_ptr(), g[2].data_ptr(), g[3].data_ptr(),
                                    parts[h].data_ptr(), B, _stream()), 'rih_mesh_loss')
        check(lib.rih_mesh_loss_final(parts[0].data_ptr(), parts[1].data_ptr(), B, wa.data_ptr(), ca.data_ptr(),
                                      out.data_ptr(), _stream()),
              'rih_mesh_loss_final')
        ctx.save_for_backward(*grads)
        ctx.mark_non_differentiable(out)
        return out[0], out

    @staticmethod
    def backward(ctx, g_total, _g_terms):
        # out of place: the saved gradients may be read again (retain_graph); ONE multi-tensor launch instead of a clone per tensor
        grads = torch._foreach_mul(list(ctx.saved_tensors), g_total)
        return (None,) + tuple(grads) + (None,) * 5


class FusedMeshLoss:
    """GPU drop-in for `calc_loss_GCN` (same arguments and total) on the fused HIP kernel.  Holds the constant
    topology of both hands on the device: faces, vertex->face adjacency, 21-joint regressor, graph permutation."""

    def __init__(self, loss_left, loss_right, converter_left, converter_right, weights=None, img_size=256):
        self.w = dict(DEFAULT_WEIGHTS)
        self.w.update(weights or {})
        self.img_size = img_size
        self.epoch = 0
        self._host = {}
        for side, gl, conv in (('left', loss_left, converter_left), ('right', loss_right, converter_right)):
            faces = gl.faces.detach().cpu().numpy().astype(np.int32)
            V = gl.J_regressor.shape[1]
            order = np.argsort(faces.reshape(-1), kind='stable')            # entries (face*3 + corner) grouped by vertex
            counts = np.bincount(faces.reshape(-1), minlength=V)
            vptr = np.zeros(V + 1, np.int32)
            vptr[1:] = np.cumsum(counts)
            perm = np.asarray(conv.graph_perm, dtype=np.int32)
            self._host[side] = dict(faces=faces, vptr=vptr, vlist=order.astype(np.int32),
                                    J=gl.J_regressor.detach().cpu().float().contiguous(), perm=perm, V=V, F=faces.shape[0],
                                    NJ=gl.J_regressor.shape[0])
        self._dev = {}
        self.Vc = None

    def weights(self, B):
        h = self._host['left']
        V, F, NJ, Vc = h['V'], h['F'], h['NJ'], self.Vc
        cnt = [B * V * 2, B * V * 3, B * NJ * 3, B * F * 3, B * F * 3, B * Vc * 3, B * Vc * 2]
        alpha = 0.0 if self.epoch < self.w['NORM_EPOCH'] else 1.0
        lw = [self.w['LABEL_2D'], self.w['LABEL_3D'], self.w['LABEL_3D'], self.w['NORMAL'], alpha * self.w['EDGE'],
              self.w['LABEL_3D'], self.w['LABEL_2D']]
        return [0.5 * a / c for a, c in zip(lw, cnt)], [float(c) for c in cnt]

    def device_weights(self, B, device):
        """(weights[7], counts[7]) as views of one device tensor that the kernels read at run time.  The tensor is
        rewritten IN PLACE (outside any stream capture) only when the values change -- batch size, or the epoch gate of
        the edge term (core/Loss.py:211-220, NORM_EPOCH) -- so a hipGraph captured at epoch 0 picks the edge term up
        when the trainer calls `set_epoch()` (or this loss eagerly) at a later epoch."""
        w, cnt = self.weights(B)
        vals = tuple(w) + tuple(cnt)
        # one slot per (device, batch size): evaluating this loss on another batch size between two replays of a captured
        # step must not rewrite the weights / counts that the graph reads
        key = ('wdev', device, B)
        if key not in self._dev:
            self._dev[key] = [torch.zeros(14, device=device, dtype=torch.float32), None]
        slot = self._dev[key]
        if slot[1] != vals:
            if torch.cuda.is_available() and device.type == 'cuda' and torch.cuda.is_current_stream_capturing():
                raise RuntimeError('FusedMeshLoss: the term weights changed during stream capture (epoch gate or batch '
                                   'size); call set_epoch() / run one eager step before capturing')
            slot[0].copy_(torch.tensor(vals, dtype=torch.float32))
            slot[1] = vals
        return slot[0][:7], slot[0][7:]

    def set_epoch(self, epoch, B=None, device=None):
        """Move the epoch gate (edge term on from NORM_EPOCH).  With a replayed hipGraph pass the captured batch size and
        device so that the device-resident weights are refreshed in place."""
        self.epoch = epoch
        if self.Vc is None:
            return
        if B is not None and device is not None:
            self.device_weights(B, torch.device(device))
        else:                                       # refresh every slot that exists (all captured batch sizes / devices)
            for key in [k for k in self._dev if k[0] == 'wdev']:
                self.device_weights(key[2], key[1])

    def topo(self, side, device):
        from ._lib import MeshTopo
        key = (side, device)
        if key not in self._dev:
            h = self._host[side]
            t = {k: torch.as_tensor(h[k], device=device) for k in ('faces', 'vptr', 'vlist', 'perm')}
            t['J'] = h['J'].to(device)
            pool = h['perm'].shape[0] // self.Vc
            assert pool * self.Vc == h['perm'].shape[0] and pool & (pool - 1) == 0
            self._dev[key] = (t, MeshTopo(t['faces'].data_ptr(), t['vptr'].data_ptr(), t['vlist'].data_ptr(),
                                          t['J'].data_ptr(), t['perm'].data_ptr(), h['V'], h['F'], h['NJ'], self.Vc, pool))
        return self._dev[key][1]

    def __call__(self, epoch, result, handDictList, v2d_l, v2d_r, v3d_l, v3d_r, root_rel):
        """Returns (total, terms) with terms = [total, vert2d, vert3d, joint, norm, edge, coarse3d, coarse2d].
        epoch=None keeps the gate where `set_epoch()` put it (what a `loss_fn` handed to TrainStep should pass: a literal 0
        there would switch the edge term off again on every eager step)."""
        if epoch is not None:
            self.epoch = epoch
        assert len(handDictList) == 1, 'one coarse level (the reference decoder emits exactly one)'
        hd = handDictList[0]
        Vc = hd['verts3d']['left'].shape[1]
        if self.Vc not in (None, Vc):
            self._dev = {k: v for k, v in self._dev.items() if k[0] == 'wdev'}
            for v in self._dev.values():
                v[1] = None
        self.Vc = Vc
        total, terms = _MeshLossFn.apply(self, result['verts3d']['left'], result['verts2d']['left'], hd['verts3d']['left'],
                                         hd['verts2d']['left'], result['verts3d']['right'], result['verts2d']['right'],
                                         hd['verts3d']['right'], hd['verts2d']['right'], v3d_l, v2d_l, v3d_r, v2d_r, root_rel)
        return total, terms


def calc_loss_GCN_fused(fused, epoch, result, paramsDict, handDictList, otherInfo, v2d_l, v2d_r, v3d_l, v3d_r, root_rel,
                        upsample_weight=None, upsample_target=None):
    """`calc_loss_GCN` on the fused kernel: same total; the mano dict carries the reference's five terms.
    `upsample_weight` / `upsample_target`: the trainable up-sampling matrix and its initial value -- the UPSAMPLE term of
    core/Loss.py:222-224, which the reference adds when the up-sampling layer is not frozen (a tiny torch expression on
    one 778x252 matrix; with the default frozen layer both are None and the term is absent, as in the reference)."""
    if (upsample_weight is None) != (upsample_target is None):
        raise ValueError('calc_loss_GCN_fused: pass upsample_weight and upsample_target together (or neither)')
    total, terms = fused(epoch, result, handDictList, v2d_l, v2d_r, v3d_l, v3d_r, root_rel)
    if upsample_weight is not None:
        total = total + fused.w['UPSAMPLE'] * F.smooth_l1_loss(upsample_weight - upsample_target,
                                                                torch.zeros_like(upsample_weight))
    mano = {'vert2d_loss': terms[1], 'vert3d_loss': terms[2], 'joint_loss': terms[3], 'norm_loss': terms[4],
            'edge_loss': terms[5]}
    return total, mano
