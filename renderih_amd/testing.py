"""Seeded, construction-order-independent weights and tensor signatures shared by the golden
generator (applied to the *reference* modules) and by the tests (applied to this package's modules).

39 M parameters cannot be committed as a fixture, so both sides fill a reference-keyed `state_dict`
from `crc32(key) ^ seed`; identical keys/shapes => identical weights.
"""
import zlib
import numpy as np
import torch


def deterministic_state(sd, seed=0):
    """Return a new state dict with the same keys/shapes/dtypes as `sd`, filled deterministically."""
    out = {}
    for k in sorted(sd.keys()):
        t = sd[k]
        rs = np.random.RandomState((zlib.crc32(k.encode()) ^ (seed * 7919)) & 0x7FFFFFFF)
        shape = tuple(t.shape)
        leaf = k.rsplit('.', 1)[-1]
        if leaf == 'num_batches_tracked':
            v = np.zeros(shape, np.int64)
        elif leaf == 'running_mean':
            v = rs.randn(*shape) * 0.1
        elif leaf == 'running_var':
            v = rs.rand(*shape) + 0.5
        elif leaf == 'dense_coor':
            v = rs.rand(*shape)
        elif k.endswith('unsample_layer.weight'):
            v = rs.rand(*shape) * (2.0 / shape[1])
        elif len(shape) >= 2:
            fan_in = int(np.prod(shape[1:]))
            if 'position_embeddings' in k:
                v = rs.randn(*shape) * 0.5
            else:
                v = rs.randn(*shape) * np.sqrt(2.0 / fan_in)
        elif leaf == 'weight':          # BN / LN scale
            v = rs.rand(*shape) + 0.5
        else:                           # biases
            v = rs.randn(*shape) * 0.1
        out[k] = torch.from_numpy(np.asarray(v)).to(t.dtype).reshape(shape)
    return out


def signature(t, nsamp=256):
    """Compact fingerprint of a tensor: [mean, mean|x|, l2, max|x|] + a strided sample."""
    a = t.detach().to(torch.float64).flatten().cpu().numpy()
    n = a.size
    stride = max(1, n // nsamp)
    samp = a[::stride][:nsamp].astype(np.float32)
    stats = np.array([a.mean(), np.abs(a).mean(), np.sqrt((a * a).sum()), np.abs(a).max(), n], np.float64)
    return stats, samp


def seeded_image(batch, seed=0):
    """Synthetic ImageNet-normalised crop stand-in: N(0,1), NCHW fp32."""
    g = torch.Generator().manual_seed(seed)
    return torch.randn(batch, 3, 256, 256, generator=g)


def flatten_outputs(outputs):
    """4-tuple of nested dicts -> flat {name: tensor} (stable names used in fixtures)."""
    result, params, hd, other = outputs
    flat = {}
    for side in ('left', 'right'):
        flat['result.verts3d.' + side] = result['verts3d'][side]
        flat['result.verts2d.' + side] = result['verts2d'][side]
        flat['params.scale.' + side] = params['scale'][side]
        flat['params.trans2d.' + side] = params['trans2d'][side]
        flat['hand0.verts3d.' + side] = hd[0]['verts3d'][side]
        flat['hand0.verts2d.' + side] = hd[0]['verts2d'][side]
        ml = other['verts3d_MANO_list'][side]
        if isinstance(ml, dict):                        # MANO model of the second family: per-hand prediction dict
            for k2 in ('verts3d', 'joints3d', 'mano_pose', 'mano_shape'):
                flat['other.mano.%s.%s' % (side, k2)] = ml[k2]
            flat['result.v3d_' + side] = result['v3d_' + side]
            flat['params.scalelength_' + side] = params['scalelength_' + side]
        elif ml:                                        # empty list in the graph model of the second family
            flat['other.verts3d_MANO.' + side] = ml[0]
            flat['other.verts2d_MANO.' + side] = other['verts2d_MANO_list'][side][0]
    for k in ('hms', 'mask', 'dense', 'length', 'root_rel'):
        if k in other:
            flat['other.' + k] = other[k]
    return flat


def rel_err(a, b):
    """max |a-b| / max|b| (tensor-level relative error)."""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def assert_close(a, b, rtol=1e-4, atol_frac=1e-5, what=''):
    """|a-b| <= rtol*|b| + atol_frac*max|b|  (fp32 parity bar used throughout tests/)."""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    tol = rtol * b.abs() + atol_frac * b.abs().max()
    bad = (a - b).abs() > tol
    if bad.any() or not torch.isfinite(a).all():
        i = int(((a - b).abs() - tol).argmax())
        raise AssertionError('%s: %d/%d elements off; worst idx %d got %.8g want %.8g (max|b|=%.4g, relerr=%.3g)'
                             % (what, int(bad.sum()), bad.numel(), i, a.flatten()[i], b.flatten()[i],
                                float(b.abs().max()), rel_err(a, b)))


def assert_fp32_equivalent(got, ref32, ref64, k=4.0, floor=2e-5, what=''):
    """Parity bar for ill-conditioned configurations (train-mode BatchNorm over a few hundred samples amplifies
    fp32 round-off to ~1e-4): the HIP result must be as close to the fp64 result as the reference's own fp32 CPU
    result is, up to a factor `k` (summation-order freedom) plus a small floor -- all relative to max|ref64|."""
    got, ref32, ref64 = (t.detach().double().cpu() for t in (got, ref32, ref64))
    assert got.shape == ref64.shape, (what, got.shape, ref64.shape)
    assert torch.isfinite(got).all(), what
    scale = float(ref64.abs().max().clamp_min(1e-30))
    e_ref = float((ref32 - ref64).abs().max()) / scale
    e_got = float((got - ref64).abs().max()) / scale
    if e_got > k * e_ref + floor:
        raise AssertionError('%s: err vs fp64 %.3g exceeds %.1f x fp32-reference err %.3g + %.1g'
                             % (what, e_got, k, e_ref, floor))
    return e_got, e_ref


def is_null_gradient(name):
    """Parameters whose gradient is exactly zero in exact arithmetic, so that any computed value is round-off noise
    and cannot be compared between implementations: key biases of softmax attention (shift invariance of softmax) and
    convolution biases that feed straight into a training-mode BatchNorm (HRNet heads / mid head, models/encoder.py:
    215-217, 300-305, 312-320)."""
    if name.endswith('w_ks.bias'):
        return True
    return name in ('encoder.hms_decoder.0.bias', 'encoder.dp_decoder.0.bias', 'mid_model.final_layer.0.bias') or \
        (name.startswith('mid_model.downsamp_modules.') and name.endswith('.0.bias'))
