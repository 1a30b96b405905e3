"""TEST INFRASTRUCTURE -- CPU restatement (numpy, float32, vectorised over voxels) of the reference's SDF voxeliser,
`pose_data_optimize/sdf/sdf/csrc/sdf_cuda_kernel.cu:242-308` (`sdf_cuda_kernel`) with its helpers `point_segment_distance`
(:69-88), `intersect_triangle` (:91-140), `point_triangle_distance` (:158-236).  Only tests/ may import this file.

PINNED: the reference implementation exists only as a CUDA kernel, but its device code is plain C++, so oracle/Makefile
compiles the reference's OWN source file for the host (oracle/_ref/libsdf_ref.so: `sdf_cuda_kernel<float>` from where it lies
under /root/reference, run on the CPU by the fiber shim of tests/hipcpu).  tests/golden/sdf_ref.npz holds its outputs
(tests/golden/make_sdf_golden.py); this restatement reproduces them with the same inside / outside decision for every voxel
and |diff| <= 1.5e-7 (tests/test_sdf.py::test_oracle_matches_reference_kernel_golden; the reference evaluates the voxel centre
in double and rounds once, numpy rounds twice).  One reference quirk is NOT restated: its host wrapper launches voxels / 512
blocks rounded DOWN (sdf_cuda_kernel.cu:313), so for B * G^3 not a multiple of 512 the trailing voxels keep the caller's zeros;
this oracle and the HIP kernel evaluate every voxel (identical for the reference's own G = 32).
"""
import numpy as np

F32 = np.float32


def _dot(a, b):
    return a[..., 0] * b[..., 0] + a[..., 1] * b[..., 1] + a[..., 2] * b[..., 2]


def _dist(a, b):
    d = a - b
    return np.sqrt(d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1] + d[..., 2] * d[..., 2], dtype=F32)


def _point_segment(x0, x1, x2):
    dx = x2 - x1
    m2 = _dot(dx, dx)
    s = (_dot(x2, dx) - _dot(x0, dx)) / m2
    s = np.clip(s, F32(0), F32(1))[..., None]
    r = s * x1 + (F32(1) - s) * x2
    return _dist(x0, r), r


def _point_triangle(x0, x1, x2, x3):
    """x0 [N,3] voxel centres, x1..x3 [3] one triangle -> closest points [N,3]."""
    x13, x23, x03 = x1 - x3, x2 - x3, x0 - x3
    m13, m23, d = _dot(x13, x13), _dot(x23, x23), _dot(x13, x23)
    invdet = F32(1) / np.maximum(m13 * m23 - d * d, F32(1e-30))
    a, b = _dot(x13, x03), _dot(x23, x03)
    w23 = invdet * (m23 * a - d * b)
    w31 = invdet * (m13 * b - d * a)
    w12 = F32(1) - w23 - w31
    inside = (w23 >= 0) & (w31 >= 0) & (w12 >= 0)
    r_in = w23[:, None] * x1 + w31[:, None] * x2 + w12[:, None] * x3
    d12, r12 = _point_segment(x0, x1, x2)
    d13, r13 = _point_segment(x0, x1, x3)
    d23, r23 = _point_segment(x0, x2, x3)
    c1 = w23 > 0                                   # rules out edge 2-3: min(1-2, 1-3)
    c2 = ~c1 & (w31 > 0)                           # rules out edge 1-3: min(1-2, 2-3)
    rA = np.where(c1[:, None] | c2[:, None], r12, r13)
    dA = np.where(c1 | c2, d12, d13)
    rB = np.where(c1[:, None], r13, r23)
    dB = np.where(c1, d13, d23)
    r_out = np.where((dA < dB)[:, None], rA, rB)
    return np.where(inside[:, None], r_in, r_out).astype(F32)


def _ray_triangle(orig, dirs, v0, v1, v2):
    e1, e2 = v1 - v0, v2 - v0
    pv = np.stack([dirs[:, 1] * e2[2] - dirs[:, 2] * e2[1], dirs[:, 2] * e2[0] - dirs[:, 0] * e2[2],
                   dirs[:, 0] * e2[1] - dirs[:, 1] * e2[0]], -1).astype(F32)
    det = (e1[0] * pv[:, 0] + e1[1] * pv[:, 1] + e1[2] * pv[:, 2]).astype(F32)
    ok = ~((det > F32(-0.000001)) & (det < F32(0.000001)))
    with np.errstate(divide='ignore', invalid='ignore', over='ignore'):
        inv = (F32(1) / det).astype(F32)
        tv = orig - v0
        u = (_dot(tv, pv) * inv).astype(F32)
        ok &= ~((u < 0) | (u > 1))
        qv = np.stack([tv[:, 1] * e1[2] - tv[:, 2] * e1[1], tv[:, 2] * e1[0] - tv[:, 0] * e1[2],
                       tv[:, 0] * e1[1] - tv[:, 1] * e1[0]], -1).astype(F32)
        v = (_dot(dirs, qv) * inv).astype(F32)
        ok &= ~((v < 0) | (u + v > 1))
        t = ((e2[0] * qv[:, 0] + e2[1] * qv[:, 1] + e2[2] * qv[:, 2]) * inv).astype(F32)
    return ok & (t >= 0)


def sdf(faces, vertices, grid_size=32):
    """faces [F,3] int, vertices [B,V,3] float32 in [-1,1]^3 -> phi [B,G,G,G] float32, indexed [b][k (z)][j (y)][i (x)]."""
    faces = np.asarray(faces, np.int64)
    vertices = np.asarray(vertices, F32)
    G = grid_size
    dx = F32(2.0) / F32(G - 1)
    ax = (F32(-1) + (np.arange(G, dtype=F32) + F32(0.5)) * dx).astype(F32)
    kk, jj, ii = np.meshgrid(np.arange(G), np.arange(G), np.arange(G), indexing='ij')
    c = np.stack([ax[ii.ravel()], ax[jj.ravel()], ax[kk.ravel()]], -1).astype(F32)
    dirs = (F32(-1) - c).astype(F32)
    out = np.zeros((vertices.shape[0], G, G, G), F32)
    for b in range(vertices.shape[0]):
        best = np.full(c.shape[0], F32(1000))
        hits = np.zeros(c.shape[0], np.int64)
        for f in faces:
            v1, v2, v3 = vertices[b, f[0]], vertices[b, f[1]], vertices[b, f[2]]
            r = _point_triangle(c, v1, v2, v3)
            best = np.minimum(best, _dist(c, r))
            hits += _ray_triangle(c, dirs, v1, v2, v3)
        out[b] = np.where(hits % 2 == 1, best, F32(0)).reshape(G, G, G)
    return out
