"""Build librenderih_amd.so (hand-written HIP kernels + C ABI) for gfx950, in-tree.

`hipcc --offload-arch=gfx950` cross-compiles without a GPU.  The .so is git-ignored but travels to the
GPU box with the repo snapshot.  Rebuilds only when a source is newer than the library.
"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'librenderih_amd.so')
SOURCES = ['rih_gemm.hip', 'rih_elem.hip', 'rih_mano.hip', 'rih_loss.hip', 'rih_metrics.hip', 'rih_pose.hip', 'rih_attn.hip', 'rih_half.hip', 'rih_input.hip', 'rih_sdf.hip']
HEADERS = ['rih_procrustes.h', 'rih_pose_math.h']


def hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return 'hipcc'


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.join(HERE, '..', 'include', 'renderih_amd.h')]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    cmd = [hipcc(), '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared',
           '-Wno-unused-result',
           # hipcc's SLP pass packs neighbouring f32 adds into v_pk_add_f32, which issues at a fraction of the scalar
           # rate next to MFMAs (MI355X_MICROARCH.md, cycle constants): keep the split arithmetic scalar
           '-fno-slp-vectorize', '-o', LIB] + srcs
    if verbose:
        print('[renderih_amd] building:', ' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    build(force=True)
