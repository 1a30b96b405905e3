"""The REAL kernel sources of renderih_amd/csrc, compiled for the host and executed on the CPU (tests/hipcpu: threads of
a block as cooperative fibers; barriers, 64-lane shuffles, buffer-load range checks and both MFMA shapes emulated), run
through the same parity tests as on the GPU, on small problems.  This is not a CPU fallback of the product -- only this
file and tests/hipcpu know about it -- but it lets the GPU-less suite check the kernels' arithmetic, indexing, LDS
staging and synchronisation, not just the host logic around them (tests/abi_emulator.py restates every entry point in
numpy instead)."""
import os
import sys
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'hipcpu'))

import test_gpu_loss as TL          # noqa: E402
import test_gpu_mano as TMANO       # noqa: E402
import test_gpu_ops as G            # noqa: E402
import test_metrics as TMET         # noqa: E402
import test_pose_head as TP         # noqa: E402

CPU = torch.device('cpu')


@pytest.fixture(autouse=True)
def _host_kernels(monkeypatch):
    from host_kernels import host_kernels_abi
    for mod in (G, TL, TMANO):
        monkeypatch.setattr(mod, 'dev', lambda: CPU)
    with host_kernels_abi():
        yield


def test_library_built_for_the_host_exports_the_c_abi():
    from host_kernels import load
    from renderih_amd import _lib
    lib = load()
    for name in _lib.SIGNATURES:
        assert hasattr(lib, name), name


@pytest.mark.parametrize('case', [
    (2, 8, 8, 32, 32, 3, 1, 1, False, True),        # implicit-GEMM 3x3 on the split engine's buffer-load path
    (1, 9, 7, 32, 16, 3, 2, 1, True, False),        # strided, odd sizes: parity-class data gradient, bias row
    (2, 8, 8, 64, 32, 1, 2, 0, False, False),       # strided 1x1
    (1, 16, 16, 3, 16, 7, 2, 3, False, False),      # 7x7 stem (Cin = 4 after padding): general kernel
    (2, 8, 8, 128, 40, 1, 1, 0, True, False),       # N not a multiple of the tile
])
def test_conv2d_kernels(case):
    G.test_conv2d(case)


@pytest.mark.parametrize('case', [(33, 64, 24, True, True, True), (64, 128, 64, True, False, False),
                                  (20, 36, 6, False, False, False)])
def test_linear_kernels(case):
    G.test_linear(case)


def test_paired_layer_kernels():
    G.test_linear_pair(33, 12, 6, True, True, False)
    G.test_linear_pair(40, 64, 64, False, False, True)
    G.test_layernorm_pair(30, 128, False, False, True)
    G.test_layernorm_pair(9, 200, True, True, False)
    G.test_patch_conv_pair(2, 8, 32, 16, 2)
    G.test_patch_conv_pair(1, 16, 64, 48, 4)                    # K = 1024: forward split-K + finish
    G.test_cross_attention_stacked_and_rows_pair()


def test_norm_softmax_attention_kernels():
    G.test_batchnorm(True, True, True, (2, 4, 4, 64))
    G.test_batchnorm(False, True, False, (2, 5, 3, 32))
    G.test_layernorm(4, 509, False, False)
    G.test_layernorm(70, 256, True, True)
    G.test_attention(2, 63, 63, 64, 4)
    G.test_attention_dropout_matches_hash_mask()
    G.test_self_attention_packed(2, 40, 64, 4)
    G.test_add_dropout_and_bcast()


def test_graph_and_resampling_kernels():
    G.test_cheby_gather_project()
    G.test_pool_upsample_layout()
    G.test_resample_hrnet(2, 4, 4, 32)


def test_tile4_pipelined_gemm_kernel():
    G.test_gemm_tile4_pipelined_kernel((1, 16, 16, 32, 128, 3, 1, 1))        # M = 256: one 256x128 tile, 4-slot ring


@pytest.mark.parametrize('B', [1, 2])
def test_mano_kernels(B):
    TMANO.test_mano_matches_oracle(B)


def test_mano_kernels_reference_golden():
    TMANO.test_mano_matches_reference_golden('left')


def test_mesh_loss_kernel():
    TL.test_fused_loss_matches_reference_golden(60)


def test_metrics_and_pose_head_kernels():
    TMET._check_against_oracle(CPU)
    TP._pose_head_kernels_vs_oracle(CPU)


@pytest.mark.parametrize('case', [
    (2, 8, 8, 64, 64, 3, 1, 1, True, True),         # 3x3: forward + stride-1 data gradient read pre-split weights
    (1, 8, 8, 64, 64, 3, 2, 1, False, False),       # strided: one pre-split tap subset per parity class
    (2, 8, 8, 64, 128, 1, 1, 0, False, True),       # 1x1
    (2, 4, 4, 96, 40, 1, 2, 0, True, False),        # N = 40: a partial 64-wide tile of plane rows
])
def test_presplit_weight_path(monkeypatch, case):
    """rih_gemm b_mode 2 (B as pre-split bf16 planes from rih_presplit_conv_weight) against F.conv2d; off by default in
    the product (ops.PRESPLIT), enabled here on the host-compiled kernels."""
    from renderih_amd import ops
    calls = []
    real = ops._presplit_weight
    monkeypatch.setattr(ops, 'PRESPLIT', True)
    monkeypatch.setattr(ops, '_presplit_weight', lambda *a, **k: (calls.append(a[2]), real(*a, **k))[1])
    G.test_conv2d(case)
    N_, H_, W_, Cin, Cout = case[:5]
    assert False in calls, calls                             # the forward operand went through it
    assert (True in calls) == (Cin > 32 and Cout % 32 == 0), calls      # ... and the data-gradient operand where eligible


def test_presplit_matrix_entry_point():
    """rih_presplit_matrix on a plain [N][K] weight + b_mode 2 GEMM == the fp32 product."""
    import math
    from renderih_amd import ops
    from renderih_amd._lib import check
    M, K, N = 70, 100, 72
    x, w = G.rnd(M, K, seed=1), G.rnd(N, K, seed=2, scale=1 / math.sqrt(K))
    Kp = 128
    planes = torch.empty(3, N, Kp // 2)
    check(ops._L().rih_presplit_matrix(w.data_ptr(), 1, K, N, K, planes.data_ptr(), Kp, 0), 'rih_presplit_matrix')
    y = torch.empty(M, N)
    ops.gemm(x, planes, y, M, N, K, K, Kp, N, a_mode=0, b_mode=2, engine=1)
    TMET.assert_close(y, (x.double() @ w.double().t()).float(), 1e-5, 1e-6, 'presplit matrix gemm')
