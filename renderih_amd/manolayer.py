"""MANO layer on the fused HIP kernels (drop-in for the reference's models/manolayer.py).

Same constructor (`ManoLayer(manoPath, center_idx=9, use_pca=True, new_skel=False)`), buffers, `forward`
signature/returns and helper methods as the reference (manolayer.py:100-322).  `forward` runs
rih_mano_fwd / rih_mano_bwd (csrc/rih_mano.hip); the helpers used only by dataset builders
(`pca2axis`, `Rmat2axis`, `get_local_frame`, ...) are small host-side torch routines.

Buffers are read at call time (callers mutate `shapedirs` in place after construction,
dataset/interhand.py:22-25 `fix_shape`), so nothing is cached or pre-folded.
"""
import ctypes as C
import pickle
import numpy as np
import torch
from torch.nn import Module

import os

from . import _lib, ops

# RIH_MANO_BWD_SPLIT=0: the one-kernel backward of rounds 1-2 (A/B timing; tools/mano_bench.py)
BWD_SPLIT = os.environ.get('RIH_MANO_BWD_SPLIT', '1') == '1'
from ._lib import ManoModel, check


def rodrigues_batch(axis):
    """axis [bs,3] -> rotation matrices [bs,3,3]; manolayer.py:32-48 semantics (angle = |axis| + 1e-8)."""
    angle = torch.norm(axis, p=2, dim=1, keepdim=True) + 1e-8
    a = axis / angle
    s = torch.sin(angle).unsqueeze(2)
    c = torch.cos(angle).unsqueeze(2)
    z = torch.zeros_like(a[:, 0])
    K = torch.stack([torch.stack([z, -a[:, 2], a[:, 1]], 1),
                     torch.stack([a[:, 2], z, -a[:, 0]], 1),
                     torch.stack([-a[:, 1], a[:, 0], z], 1)], 1)
    eye = torch.eye(3, dtype=axis.dtype, device=axis.device).unsqueeze(0)
    return eye + s * K + (1 - c) * K.bmm(K)


def vec2mat(vec):
    """6-D rotation representation -> matrix (manolayer.py:20-29)."""
    x = vec[:, 0:3]
    y = vec[:, 3:6]
    x = x / (torch.norm(x, p=2, dim=1, keepdim=True) + 1e-8)
    y = y - torch.sum(x * y, dim=1, keepdim=True) * x
    y = y / (torch.norm(y, p=2, dim=1, keepdim=True) + 1e-8)
    z = torch.cross(x, y, dim=1)
    return torch.stack([x, y, z], dim=2)


def _unit(v):
    return v / torch.norm(v, dim=-1, keepdim=True)


def _frame_transfer(old_z, new_z):
    """Rotation taking old_z to new_z about their common normal (manolayer.py:51-60)."""
    x = _unit(torch.cross(old_z, new_z, dim=-1))
    old = torch.stack((x, torch.cross(old_z, x, dim=-1), old_z), dim=2)
    new = torch.stack((x, torch.cross(new_z, x, dim=-1), new_z), dim=2)
    return new.matmul(old.transpose(1, 2))


def build_mano_frame(skelBatch):
    """Per-joint local frames from a 21-joint skeleton (manolayer.py:63-97): columns = [splay, bend, twist]."""
    son = [2, 3, 17, 5, 6, 18, 8, 9, 20, 11, 12, 19, 14, 15, 16]
    parent = [-1, 0, 1, 2, 0, 4, 5, 0, 7, 8, 0, 10, 11, 0, 13, 14]
    palm = [13, 1, 4, 10, 7]
    order = [0, 5, 6, 7, 9, 10, 11, 17, 18, 19, 13, 14, 15, 1, 2, 3, 4, 8, 12, 16, 20]
    bs = skelBatch.shape[0]
    skel = skelBatch[:, order]
    z = _unit(skel[:, son] - skel[:, 1:16])
    z = torch.cat((torch.zeros_like(z[:, :1]), z), dim=1)
    x = torch.zeros_like(z)
    x[:, :, 1] = 1.0
    y = torch.zeros_like(z)
    pv = skel[:, palm] - skel[:, 0:1]
    n = _unit(torch.cross(pv[:, :-1], pv[:, 1:], dim=-1))
    px = torch.zeros((bs, 5, 3), dtype=n.dtype, device=n.device)
    px[:, :-1] += n
    px[:, 1:] += n
    x[:, palm] = _unit(px)
    y[:, palm] = _unit(torch.cross(z[:, palm], x[:, palm], dim=-1))
    x[:, palm] = torch.cross(y[:, palm], z[:, palm], dim=-1)
    frame = torch.stack((x, y, z), dim=3)
    for i in range(1, 16):
        if i in palm:
            continue
        frame[:, i] = _frame_transfer(z[:, parent[i]], z[:, i]).matmul(frame[:, parent[i]])
    return frame[:, 1:]


# 0: the fused single-launch forward (hand-chunk major from 4096 hands on); 2 / 3: force its hand-major / tile-major form;
# 1: round 1's two-kernel forward (A/B timing, tools/mano_bench.py --variant 1)
VARIANT = 0


class _ManoFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, layer, root, pose, shape, trans, scale):
        ops._chk(root, pose, shape, trans, scale)        # fp32 GPU tensors only: there is no CPU fallback
        lib = _lib.load()
        B = root.shape[0]
        root_c, pose_c, shape_c = root.contiguous(), pose.contiguous(), shape.contiguous()
        trans_c = trans.contiguous() if trans is not None else None
        scale_c = scale.contiguous() if scale is not None else None
        ncomp = pose_c.shape[1] if layer.use_pca else 0
        dev = root.device
        v = torch.empty((B, 778, 3), device=dev, dtype=torch.float32)
        j = torch.empty((B, 21, 3), device=dev, dtype=torch.float32)
        # the backward's workspace (29.8 KB per hand) is written only when a gradient can be asked for; inference moves
        # nothing but the inputs, v and j
        need_ws = any(ctx.needs_input_grad)
        variant = VARIANT
        ws = (torch.empty((int(lib.rih_mano_ws_floats(B)),), device=dev, dtype=torch.float32)
              if (need_ws or variant == 1) else None)
        mm = layer._model_struct()
        packed = layer._packed_basis(mm)
        stream = ops._stream()
        cidx = -1 if layer.center_idx is None else int(layer.center_idx)
        check(lib.rih_mano_fwd(C.byref(mm), packed.data_ptr(), root_c.data_ptr(), pose_c.data_ptr(), ncomp,
                               shape_c.data_ptr(), 0 if trans_c is None else trans_c.data_ptr(),
                               0 if scale_c is None else scale_c.data_ptr(), cidx, 1 if layer.new_skel else 0,
                               v.data_ptr(), j.data_ptr(), 0 if ws is None else ws.data_ptr(), B, variant, stream),
              'rih_mano_fwd')
        ctx.layer = layer
        ctx.save_for_backward(root_c, pose_c, shape_c, trans_c, scale_c, ws)
        ctx.cfg = (ncomp, cidx, 1 if layer.new_skel else 0, tuple(pose.shape), tuple(root.shape))
        return v, j

    @staticmethod
    def backward(ctx, dv, dj):
        root, pose, shape, trans, scale, ws = ctx.saved_tensors
        ncomp, cidx, new_skel, pose_shape, root_shape = ctx.cfg
        lib = _lib.load()
        B = root.shape[0]
        dev = root.device
        dv = dv.contiguous() if dv is not None else torch.zeros((B, 778, 3), device=dev)
        dj = dj.contiguous() if dj is not None else torch.zeros((B, 21, 3), device=dev)
        d_root = torch.empty((B, 9), device=dev, dtype=torch.float32)
        d_pose = torch.empty((B, ncomp if ncomp > 0 else 135), device=dev, dtype=torch.float32)
        d_shape = torch.empty((B, 10), device=dev, dtype=torch.float32)
        d_trans = torch.empty((B, 3), device=dev, dtype=torch.float32) if trans is not None else None
        d_scale = torch.empty((B,), device=dev, dtype=torch.float32) if scale is not None else None
        mm = ctx.layer._model_struct()
        split = BWD_SPLIT
        packed = ctx.layer._packed_basis(mm) if split else None
        wsb = torch.empty((int(lib.rih_mano_bwd_ws_floats(B)),), device=dev, dtype=torch.float32) if split else None
        check(lib.rih_mano_bwd(C.byref(mm), 0 if packed is None else packed.data_ptr(), root.data_ptr(), pose.data_ptr(), ncomp,
                               shape.data_ptr(),
                               0 if trans is None else trans.data_ptr(), 0 if scale is None else scale.data_ptr(),
                               cidx, new_skel, dv.data_ptr(), dj.data_ptr(), ws.data_ptr(), d_root.data_ptr(),
                               d_pose.data_ptr(), d_shape.data_ptr(),
                               0 if d_trans is None else d_trans.data_ptr(),
                               0 if d_scale is None else d_scale.data_ptr(), 0 if wsb is None else wsb.data_ptr(), B,
                               ops._stream()), 'rih_mano_bwd')
        return None, d_root.view(root_shape), d_pose.view(pose_shape), d_shape, d_trans, d_scale


class ManoLayer(Module):
    def __init__(self, manoPath, center_idx=9, use_pca=True, new_skel=False):
        super().__init__()
        self.center_idx = center_idx
        self.use_pca = use_pca
        self.new_skel = new_skel
        if isinstance(manoPath, dict):
            manoData = manoPath
        else:
            with open(manoPath, 'rb') as f:
                manoData = pickle.load(f, encoding='latin1')
        self.new_order = [0, 13, 14, 15, 16, 1, 2, 3, 17, 4, 5, 6, 18, 10, 11, 12, 19, 7, 8, 9, 20]

        def f32(a):
            return torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float32)))

        self.register_buffer('hands_components', f32(manoData['hands_components']))
        self.register_buffer('hands_components_inv', torch.inverse(self.hands_components))
        Jr = manoData['J_regressor']
        Jr = np.asarray(Jr.todense()) if hasattr(Jr, 'todense') else np.asarray(Jr)
        self.register_buffer('J_regressor', f32(Jr), persistent=False)
        self.register_buffer('J_zero', f32(manoData['J']), persistent=False)
        self.register_buffer('weights', f32(manoData['weights']), persistent=False)
        self.register_buffer('posedirs', f32(manoData['posedirs']), persistent=False)
        self.register_buffer('v_template', f32(manoData['v_template']), persistent=False)
        sd = manoData['shapedirs']
        sd = sd if isinstance(sd, np.ndarray) else np.asarray(sd.r)     # chumpy object in the original MANO pickle
        self.register_buffer('shapedirs', f32(sd), persistent=False)
        self.register_buffer('hands_mean', f32(manoData['hands_mean']), persistent=False)
        self.faces = manoData['f']
        self.parent = [-1] + [int(manoData['kintree_table'][0, i]) for i in range(1, 16)]
        self.is_train = True

    def get_faces(self):
        return self.faces

    def train(self, mode=True):          # the reference overrides train/eval to a bare flag (manolayer.py:157-161)
        self.is_train = mode

    def eval(self):
        self.train(False)

    # ---- small host-side helpers (dataset builders / fitting code call these; not the hot path)
    def pca2axis(self, pca):
        return pca.mm(self.hands_components[:pca.shape[1]]) + self.hands_mean

    def axis2Rmat(self, axis):
        return rodrigues_batch(axis.view(-1, 3)).view(-1, 15, 3, 3)

    def pca2Rmat(self, pca):
        return self.axis2Rmat(self.pca2axis(pca))

    def axis2pca(self, axis):
        return (axis - self.hands_mean).mm(self.hands_components_inv)

    def Rmat2pca(self, R):
        return self.axis2pca(self.Rmat2axis(R))

    def Rmat2axis(self, R):
        """Rotation matrices -> axis-angle [bs,45] (manolayer.py:177-207, incl. its 3.14159 branch constants)."""
        R = R.view(-1, 3, 3)
        skew = (R - R.transpose(1, 2)) / 2
        L = torch.stack((skew[:, 2, 1], skew[:, 0, 2], skew[:, 1, 0]), dim=1)
        sin = torch.norm(L, dim=1)
        L = L / (sin.unsqueeze(-1) + 1e-8)
        eye = torch.eye(3, dtype=R.dtype, device=R.device)
        sym = (R + R.transpose(1, 2)) / 2 - eye
        outer = L.unsqueeze(-1) * L.unsqueeze(1) - eye
        t1 = sym.diagonal(dim1=1, dim2=2).sum(1)
        t2 = outer.diagonal(dim1=1, dim2=2).sum(1)
        cos = 1 - t1 / (t2 + 1e-8)
        sin = torch.clamp(sin, min=-1 + 1e-7, max=1 - 1e-7)
        theta = torch.asin(sin)
        theta2 = theta.clone()
        m1 = (cos < 0) & (sin > 0)
        m2 = (cos < 0) & (sin < 0)
        theta2[m1] = 3.14159 - theta[m1]
        theta2[m2] = -3.14159 - theta[m2]
        return (theta2.unsqueeze(-1) * L).view(-1, 45)

    def get_local_frame(self, shape):
        """Local joint frames at zero pose (manolayer.py:217-227; note the tip indices differ from forward's)."""
        with torch.no_grad():
            blend = torch.matmul(self.shapedirs, shape.permute(1, 0)).permute(2, 0, 1)
            v_shaped = self.v_template + blend
            j = torch.matmul(self.J_regressor, v_shaped)
            j21 = torch.cat((j, v_shaped[:, [744, 320, 444, 555, 672]]), dim=1)[:, self.new_order]
            return build_mano_frame(j21)

    @staticmethod
    def buildSE3_batch(R, t):
        bs = R.shape[0]
        bottom = torch.zeros((bs, 1, 4), dtype=R.dtype, device=R.device)
        bottom[:, 0, 3] = 1.0
        return torch.cat([torch.cat([R, t], 2), bottom], 1)

    @staticmethod
    def SE3_apply(SE3, v):
        return (SE3[:, :3, :3].bmm(v.unsqueeze(2)) + SE3[:, :3, 3:4])[:, :, 0]

    def _packed_basis(self, mm):
        """The packed blend basis of the fused kernel (rih_mano_pack), rebuilt whenever one of the buffers it is made of was
        replaced or modified in place (callers do that: dataset/interhand.py:22-25 `fix_shape` flips shapedirs) -- keyed on
        the tensors' data pointers and version counters, so the steady state is one launch per forward."""
        srcs = (self.shapedirs, self.posedirs, self.v_template, self.J_regressor)
        key = tuple((t.data_ptr(), t._version) for t in srcs)
        cache = getattr(self, '_pack_cache', None)
        if cache is None or cache[0] != key:
            lib = _lib.load()
            buf = torch.empty((int(lib.rih_mano_pack_floats()),), device=self.posedirs.device, dtype=torch.float32)
            check(lib.rih_mano_pack(C.byref(mm), buf.data_ptr(), ops._stream()), 'rih_mano_pack')
            cache = (key, buf)
            object.__setattr__(self, '_pack_cache', cache)
        return cache[1]

    def _model_struct(self):
        mm = ManoModel()
        mm.comps = self.hands_components.data_ptr()
        mm.hands_mean = self.hands_mean.data_ptr()
        mm.shapedirs = self.shapedirs.data_ptr()
        mm.posedirs = self.posedirs.data_ptr()
        mm.v_template = self.v_template.data_ptr()
        mm.J_reg = self.J_regressor.data_ptr()
        mm.weights = self.weights.data_ptr()
        for i in range(16):
            mm.parent[i] = self.parent[i]
        for t in (self.hands_components, self.hands_mean, self.shapedirs, self.posedirs, self.v_template,
                  self.J_regressor, self.weights):
            ops._chk(t)
            if not t.is_contiguous():
                raise RuntimeError('ManoLayer buffers must be contiguous GPU tensors (call .cuda() on the layer)')
        return mm

    def forward(self, root_rotation, pose, shape, trans=None, scale=None):
        """root_rotation [bs,3,3]; pose [bs,ncomps] (PCA) or [bs,15,3,3]; shape [bs,10]; trans [bs,3]|None;
        scale [bs]|None  ->  v [bs,778,3], j [bs,21,3]   (manolayer.py:250-322)."""
        return _ManoFn.apply(self, root_rotation, pose, shape, trans, scale)
