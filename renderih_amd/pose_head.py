"""Autograd wrappers of the MANO parameter head kernels (csrc/rih_pose.hip): Hardswish / scaled tanh, rot6d -> rotation
matrix + axis-angle, Rodrigues, root-centred bone-length-normalised mesh.  Reference: common/myhand/decoder_lijun_mano.py
:112-160, 247-300."""
import torch

from . import ops
from ._lib import check


class HardswishFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ops._chk(x)
        x = ops._c(x)
        y = torch.empty_like(x)
        check(ops._L().rih_hardswish_fwd(x.data_ptr(), y.data_ptr(), x.numel(), ops._stream()), 'rih_hardswish_fwd')
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, = ctx.saved_tensors
        dy = ops._c(dy)
        dx = torch.empty_like(x)
        check(ops._L().rih_hardswish_bwd(dy.data_ptr(), x.data_ptr(), dx.data_ptr(), x.numel(), ops._stream()),
              'rih_hardswish_bwd')
        return dx


class TanhScaleFn(torch.autograd.Function):
    """scale * tanh(x)"""

    @staticmethod
    def forward(ctx, x, scale):
        ops._chk(x)
        x = ops._c(x)
        y = torch.empty_like(x)
        check(ops._L().rih_tanh_scale_fwd(x.data_ptr(), y.data_ptr(), x.numel(), scale, ops._stream()), 'rih_tanh_scale_fwd')
        ctx.save_for_backward(y)
        ctx.scale = scale
        return y

    @staticmethod
    def backward(ctx, dy):
        y, = ctx.saved_tensors
        dy = ops._c(dy)
        dx = torch.empty_like(y)
        check(ops._L().rih_tanh_scale_bwd(dy.data_ptr(), y.data_ptr(), dx.data_ptr(), y.numel(), ctx.scale, ops._stream()),
              'rih_tanh_scale_bwd')
        return dx, None


class Rot6dFn(torch.autograd.Function):
    """x [n,6] -> (R [n,3,3], axis-angle [n,3])"""

    @staticmethod
    def forward(ctx, x):
        ops._chk(x)
        x = ops._c(x)
        n = x.shape[0]
        R = torch.empty((n, 3, 3), device=x.device, dtype=torch.float32)
        aa = torch.empty((n, 3), device=x.device, dtype=torch.float32)
        check(ops._L().rih_rot6d_fwd(x.data_ptr(), R.data_ptr(), aa.data_ptr(), n, ops._stream()), 'rih_rot6d_fwd')
        ctx.save_for_backward(x)
        return R, aa

    @staticmethod
    def backward(ctx, dR, daa):
        x, = ctx.saved_tensors
        # autograd materialises an all-zero gradient for an unused output; both are tiny, pass them through
        dR, daa = ops._c(dR), ops._c(daa)
        dx = torch.empty_like(x)
        check(ops._L().rih_rot6d_bwd(x.data_ptr(), dR.data_ptr(), daa.data_ptr(), dx.data_ptr(), x.shape[0], ops._stream()),
              'rih_rot6d_bwd')
        return dx


class RodriguesFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a):
        ops._chk(a)
        a = ops._c(a)
        n = a.shape[0]
        R = torch.empty((n, 3, 3), device=a.device, dtype=torch.float32)
        check(ops._L().rih_rodrigues_fwd(a.data_ptr(), R.data_ptr(), n, ops._stream()), 'rih_rodrigues_fwd')
        ctx.save_for_backward(a)
        return R

    @staticmethod
    def backward(ctx, dR):
        a, = ctx.saved_tensors
        dR = ops._c(dR)
        da = torch.empty_like(a)
        check(ops._L().rih_rodrigues_bwd(a.data_ptr(), dR.data_ptr(), da.data_ptr(), a.shape[0], ops._stream()),
              'rih_rodrigues_bwd')
        return da


class CenterScaleFn(torch.autograd.Function):
    """(v - j[root]) * s with s = target / |j[a] - j[b]|  ->  (mesh [B,V,3], s [B])"""

    @staticmethod
    def forward(ctx, v, j, root, ja, jb, target):
        ops._chk(v, j)
        v, j = ops._c(v), ops._c(j)
        B, V, _ = v.shape
        NJ = j.shape[1]
        out = torch.empty_like(v)
        s = torch.empty((B,), device=v.device, dtype=torch.float32)
        check(ops._L().rih_center_scale_fwd(v.data_ptr(), j.data_ptr(), B, V, NJ, root, ja, jb, target, out.data_ptr(),
                                            s.data_ptr(), ops._stream()), 'rih_center_scale_fwd')
        ctx.save_for_backward(v, j)
        ctx.cfg = (root, ja, jb, target)
        return out, s

    @staticmethod
    def backward(ctx, dout, ds):
        v, j = ctx.saved_tensors
        root, ja, jb, target = ctx.cfg
        dout, ds = ops._c(dout), ops._c(ds)
        B, V, _ = v.shape
        dv, dj = torch.empty_like(v), torch.empty_like(j)
        check(ops._L().rih_center_scale_bwd(v.data_ptr(), j.data_ptr(), dout.data_ptr(), ds.data_ptr(), B, V, j.shape[1],
                                            root, ja, jb, target, dv.data_ptr(), dj.data_ptr(), ops._stream()),
              'rih_center_scale_bwd')
        return dv, dj, None, None, None, None


def hardswish(x):
    return HardswishFn.apply(x)


def tanh_scale(x, scale):
    return TanhScaleFn.apply(x, float(scale))


def rot6d_to_rotmat_aa(x):
    return Rot6dFn.apply(x)


def rodrigues(a):
    return RodriguesFn.apply(a)


def center_scale(v, j, root=0, bone=(9, 0), target=0.095):
    return CenterScaleFn.apply(v, j, root, bone[0], bone[1], float(target))
