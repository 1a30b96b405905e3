"""TEST INFRASTRUCTURE: build the kernel sources of renderih_amd/csrc for the HOST (clang++ -x c++ with the fiber-based
HIP shim in this directory, gfx950 builtins emulated) and route renderih_amd.ops through that library, so that the CPU
test-suite executes the real kernels -- LDS tiling, barriers, wavefront shuffles, buffer-load range checks, MFMA operand
layouts -- on small problems.  Nothing here is used by the product; see hip/hip_runtime.h for the execution model."""
import contextlib
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, '..', '..'))
CSRC = os.path.join(ROOT, 'renderih_amd', 'csrc')
# HIPCPU_ASAN=1: build with AddressSanitizer into a separate library (run python with LD_PRELOAD=<libclang_rt.asan-x86_64.so>
# and ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0): out-of-bounds reads / writes of the kernels on the
# caller's buffers, which plain host execution would silently tolerate, abort the test
ASAN = os.environ.get('HIPCPU_ASAN', '0') == '1'
OUT = os.path.join(HERE, '_build', 'librenderih_cpu%s.so' % ('_asan' if ASAN else '',))


def clangxx():
    for c in (os.environ.get('HIPCPU_CXX'), '/opt/rocm/lib/llvm/bin/clang++', 'clang++'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError('no clang++ for the host build of the kernels')


def build(force=False):
    from renderih_amd import _build
    srcs = [os.path.join(CSRC, s) for s in _build.SOURCES]
    deps = srcs + [os.path.join(CSRC, h) for h in _build.HEADERS] + \
        [os.path.join(HERE, 'hip', 'hip_runtime.h'), os.path.join(HERE, 'hipcpu_gfx950.h'),
         os.path.join(ROOT, 'include', 'renderih_amd.h')]
    if not force and os.path.exists(OUT) and all(os.path.getmtime(d) <= os.path.getmtime(OUT) for d in deps):
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    objs = []
    procs = []
    for s in srcs:                                   # one translation unit per process: the GEMM file dominates
        o = os.path.join(os.path.dirname(OUT), os.path.basename(s) + ('.asan.o' if ASAN else '.o'))
        objs.append(o)
        procs.append(subprocess.Popen([clangxx(), '-x', 'c++', '-std=c++20', '-O1', '-fPIC', '-c', '-I', HERE,
                                       '-DRIH_CONST_AS='] +
                                      (['-fsanitize=address', '-fno-omit-frame-pointer', '-g'] if ASAN else []) + [
                                       '-I', os.path.join(ROOT, 'include'), '-Wno-unused-value', '-Wno-pass-failed', '-o', o, s]))
    for p in procs:
        if p.wait() != 0:
            raise RuntimeError('host build of the kernels failed')
    subprocess.check_call([clangxx(), '-shared', '-o', OUT] + (['-fsanitize=address', '-shared-libasan'] if ASAN else []) + objs)
    return OUT


def load():
    from renderih_amd import _lib
    lib = C.CDLL(build())
    sigs = dict(_lib.SIGNATURES)
    for name, (res, args) in sigs.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


@contextlib.contextmanager
def host_kernels_abi():
    """renderih_amd.ops on CPU tensors through the host-compiled kernels (the counterpart of abi_emulator.emulated_abi,
    which restates every entry point in numpy instead)."""
    from renderih_amd import _lib, ops
    saved = (_lib._lib, ops._chk, ops._stream)
    _lib._lib = load()
    ops._chk = lambda *a, **k: None
    ops._stream = lambda: 0
    try:
        yield
    finally:
        _lib._lib, ops._chk, ops._stream = saved
