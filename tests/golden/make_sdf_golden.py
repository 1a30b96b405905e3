#!/usr/bin/env python
"""Golden vectors of the SDF voxeliser from the REFERENCE's own kernel: oracle/_ref/libsdf_ref.so is the reference's
`sdf_cuda_kernel.cu` compiled for the host from where it lies under /root/reference (oracle/Makefile) and executed on the CPU.
Writes tests/golden/sdf_ref.npz (inputs + reference phi) -- small fixtures that travel to the GPU box, where /root/reference
does not exist.      python tests/golden/make_sdf_golden.py"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, '..', '..'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def reference_lib():
    path = os.path.join(ROOT, 'oracle', '_ref', 'libsdf_ref.so')
    if not os.path.exists(path) and os.path.isdir('/root/reference'):
        subprocess.check_call(['make', '-C', os.path.join(ROOT, 'oracle')])
    if not os.path.exists(path):
        return None
    lib = C.CDLL(path)
    lib.sdf_ref_f32.restype = C.c_int
    lib.sdf_ref_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
    return lib


def reference_sdf(lib, faces, vertices, G, fill=0.0):
    """phi [B,G,G,G] from the reference kernel; `fill` is what the caller put into phi beforehand (the reference launches
    voxels / 512 blocks rounded down, so trailing voxels keep it -- sdf.py:24 of the reference passes zeros)."""
    faces = np.ascontiguousarray(faces, np.int32)
    vertices = np.ascontiguousarray(vertices, np.float32)
    B, V = vertices.shape[0], vertices.shape[1]
    phi = np.full((B, G, G, G), fill, np.float32)
    rc = lib.sdf_ref_f32(phi.ctypes.data, faces.ctypes.data, vertices.ctypes.data, B, faces.shape[0], V, G)
    assert rc == 0
    return phi


def cases():
    from test_sdf import icosphere
    out = {}
    v1, f = icosphere(0.6, 1, (0.1, 0.0, -0.1))
    v2, _ = icosphere(0.35, 1, (-0.3, 0.2, 0.3))
    out['spheres_g16'] = (f, np.stack([v1, v2]), 16)
    out['spheres_g12'] = (f, np.stack([v2, v1]), 12)          # 2 x 1728 voxels: 6 blocks of 512 + an unwritten tail
    rs = np.random.RandomState(3)
    v3, f3 = icosphere(0.5, 2, (0.0, 0.05, 0.0))
    v3 = (v3 + rs.randn(*v3.shape).astype(np.float32) * 0.02).astype(np.float32)       # a bumpy closed surface
    out['bumpy_g20'] = (f3, v3[None], 20)
    return out


if __name__ == '__main__':
    lib = reference_lib()
    assert lib is not None, 'needs /root/reference (or a prebuilt oracle/_ref/libsdf_ref.so)'
    store = {}
    for name, (f, v, G) in cases().items():
        store[name + '/faces'] = f.astype(np.int32)
        store[name + '/vertices'] = v.astype(np.float32)
        store[name + '/grid'] = np.int32(G)
        store[name + '/phi'] = reference_sdf(lib, f, v, G)
    np.savez_compressed(os.path.join(HERE, 'sdf_ref.npz'), **store)
    print('wrote', os.path.join(HERE, 'sdf_ref.npz'), {k: v.shape for k, v in store.items() if k.endswith('phi')})
