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


@pytest.mark.parametrize('engine', [1, 2])
@pytest.mark.parametrize('case', [(1, 8, 8, (32, 64), 40, False), (2, 4, 4, (64, 32, 96), 128, True), (1, 5, 7, (32, 32, 64, 32), 64, True)])
def test_conv1x1_cat_kernels(case, engine):
    G.test_conv1x1_cat(case, engine)


@pytest.mark.parametrize('case,engine', [((1, 8, 8, (32, 64), 40, True), 0), ((1, 4, 4, (32, 40), 32, True), 2),
                                         ((1, 4, 4, (64, 32), 64, 'ungated'), 2), ((1, 4, 4, (64, 32), 64, 'ungated'), 0)])
def test_conv1x1_cat_fallback_and_ungated_relu_kernels(case, engine):
    """Round-4 advisor findings on ops.conv1x1_cat: engine 0 (RIH_GEMM_ENGINE=0 is a documented switch; rih_gemm reads a segmented A
    on the split engines only) and channel counts off the 32-grid fall back to the concatenation instead of failing with
    RIH_EINVAL; a consumer that does not pre-gate the gradient gets the ReLU's backward gate from the function itself."""
    G.test_conv1x1_cat(case, engine)


@pytest.mark.parametrize('case', [(1, 8, 32, 32, 64, False, True), (2, 8, 32, 64, 128, True, True), (1, 16, 64, 32, 64, True, False),
                                  (1, 16, 16, 32, 64, True, True), (1, 16, 32, 32, 32, False, True), (2, 16, 16, 64, 96, True, True)])
def test_conv3x3_halo_kernels(case):
    """csrc/rih_conv3.hip on the host harness (LDS-DMA lands at the barrier that publishes it, 512 fibers per block): halo staging,
    tap windows, H2 weight planes, statistics epilogue, data gradient."""
    G.test_conv3x3_halo(case)


@pytest.mark.parametrize('case', [(1, 8, 32, 32, 32), (1, 16, 16, 64, 64), (1, 8, 32, 32, 128)])
def test_conv3x3_halo_residual_kernels(case):
    """conv3x3_halo_kernel<*, *, false, RES> on the host harness: the residual read of the epilogue (patch addressing on 8 x 32 and
    16 x 16 patches, 32- / 64- / 128-wide channel blocks) in the data gradient of ops.conv2d_skip."""
    G.test_conv3x3_halo_with_skip_gradient(case)
    G.test_conv3x3_residual_preconditions()


@pytest.mark.parametrize('case', [(1, 16, 16, 64, 256, False, True), (1, 16, 16, 128, 128, True, True), (1, 16, 8, 64, 64, False, True),
                                  (1, 32, 16, 64, 128, True, False)])
def test_panel_kernels(case):
    """csrc/rih_conv3.hip panel_kernel on the host harness (a "device" of 6 CUs: persistent workgroups walk over several row tiles,
    ragged trip counts): resident weight planes, two row stages, the epilogue through the consumed stage, residual, statistics."""
    G.test_panel_1x1(case)


@pytest.mark.parametrize('case', [(2, 16, 16, 256, 64, False, True), (1, 16, 8, 96, 128, True, True), (1, 8, 16, 64, 256, False, True),
                                  (3, 16, 8, 160, 320, True, False)])
def test_rows_kernels(case):
    """csrc/rih_conv3.hip rows_kernel on the host harness: LDS-DMA-staged weight planes on two stages, three A stages rotating,
    tiles 128 x 64 / 128 x 128 / 256 x 64, one to eight k-tiles, residual and statistics epilogues."""
    G.test_rows_1x1(case)


@pytest.mark.parametrize('case', [(2, 32, 32, True, True), (4, 16, 16, True, False)])
def test_stem_kernel(case):
    """csrc/rih_conv3.hip rows_kernel<STEM> on the host harness: the im2col loader's predication (image border, the seven padding
    taps of the last k-tile), seven k-tiles through the two-stage pipeline, statistics epilogue."""
    G.test_stem_conv(case)


def test_grouped_wgrad_128x64_kernels(monkeypatch):
    """gemm_split_multi_kernel<128, 64, 1, 0, *, 2> (round 6) on the host harness."""
    G.test_grouped_weight_gradients_on_128x64_tiles(monkeypatch)


def test_conv3_lds_image_is_conflict_free():
    """The LDS images of csrc/rih_conv3.hip: a pixel / weight row = 8 units of 16 bytes at position j ^ ((index >> 1) & 7).  A
    ds_read_b128 is served in 16-lane groups (MI355X_MICROARCH.md, LDS table); within a group every lane must hit its own 16-byte
    bank group (64 banks x 4 B = 16 groups).  Checked for every tap shift of the A window (any halo origin) and for the weight
    rows."""
    groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    for tw in (32, 16):                         # patch width; halo row pitch tw + 2; a 32-lane block = 32 / tw image rows
        pitch = tw + 2
        for row0 in range(0, 256 // tw, 32 // tw):
            for kh in range(3):
                for kw in range(3):
                    for j in range(8):
                        for g in groups:
                            banks = set()
                            for l in g:
                                row, col = row0 + (0 if tw == 32 else l >> 4) + kh, (l & (tw - 1)) + kw
                                banks.add(((row * pitch + col) * 8 + (j ^ ((col >> 1) & 7))) % 16)
                            assert len(banks) == 16, (tw, row0, kh, kw, j)
    for r0 in range(0, 128, 32):                # rows of the panel kernel: unit j at (j & ~15) | ((j & 15) ^ (row & 15)), 16 / 32 units per row
        for upr in (16, 32):
            for j in range(upr):
                for g in groups:
                    assert len({(((r0 + l) * upr + ((j & ~15) | ((j & 15) ^ ((r0 + l) & 15)))) % 16) for l in g}) == 16
    for n0 in range(0, 128, 32):                # weight rows: position j ^ ((n >> 1) & 7)
        for j in range(8):
            for g in groups:
                assert len({(((n0 + l) * 8 + (j ^ (((n0 + l) >> 1) & 7))) % 16) for l in g}) == 16


def test_parameter_bounds_are_not_trusted_outside_an_owning_scope():
    """Round-4 advisor finding: engine 2 derives its fp16 operand scale from a cached bound of the weight; outside a model forward /
    TrainStep nothing invalidates that cache when the weight is rewritten behind torch's version counter (rih_adam_multi through
    raw pointers, `w.data.mul_`) -- a conv2d after an 8-fold growth overflowed the fp16 planes (inf / NaN).  Parameters are now
    measured again at every use outside ops.owned_bounds(), and renderih_amd.optim.Adam starts a new bound epoch."""
    import torch.nn.functional as F
    from renderih_amd import ops, optim
    saved = ops.ENGINE
    ops.ENGINE = 2
    try:
        g = torch.Generator().manual_seed(3)
        x = torch.randn(1, 8, 8, 32, generator=g)
        w = torch.nn.Parameter(torch.randn(32, 32, 3, 3, generator=g) * 0.1)

        def check(what):
            y = ops.conv2d(x, w, None, stride=1, pad=1)
            want = F.conv2d(x.permute(0, 3, 1, 2), w.detach(), padding=1).permute(0, 2, 3, 1)
            assert torch.isfinite(y).all(), what
            G.assert_close(y, want, 1e-4, 1e-5, what)
        check('first call')
        w.data.mul_(64.0)                       # invisible to w._version
        check('after w.data.mul_(64)')
        # the one-launch Adam rewrites the weights through raw pointers: a bound cached INSIDE an owning scope dies with its step
        with ops.owned_bounds():
            ops.bounds_reset()
            ops.bound_weights([w])
            cached = ops.bound_of(w)
            assert ops.bound_of(w) is cached                    # trusted inside the scope
            opt = optim.Adam([w], lr=50.0)
            ops.conv2d(x, w, None, stride=1, pad=1).sum().backward()
            opt.step()
            assert ops.bound_of(w) is not cached                # new epoch: measured again
        check('after an Adam step with a huge learning rate')
    finally:
        ops.ENGINE = saved


@pytest.mark.parametrize('case', [(33, 64, 24, True, True, True), (64, 128, 64, True, False, False),
                                  (20, 36, 6, False, False, False)])
def test_linear_kernels(case):
    G.test_linear(case)


def test_paired_layer_kernels():
    G.test_linear_pair(33, 12, 6, True, True, False)
    G.test_linear_pair(20, 64, 64, False, False, True)
    G.test_layernorm_pair(30, 128, False, False, True)
    G.test_layernorm_pair(9, 200, True, True, False)
    G.test_patch_conv_pair(2, 8, 32, 16, 2)
    G.test_patch_conv_pair(1, 16, 64, 48, 4)                    # K = 1024: forward split-K + finish
    # (the descriptor-list kernels behind ops.deferred_reductions / ops.PackCache are fuzzed at the ABI level below,
    # test_batched_entry_points_fuzz_against_emulator; their host logic runs in tests/test_cpu_emulated.py)
    G.test_conv_bn_statistics_from_the_gemm_epilogue((3, 9, 7, 32, 72, 3, 2, 1, True))          # statistics epilogue + ReLU gate
    G.test_conv_bn_epilogue_statistics_survive_a_large_mean()
    G.test_adam_one_launch_matches_torch(False, 1e-2)
    G.test_grouped_weight_gradients(2)          # the grouped launch through the ops layer (ABI-level fuzz below)


def test_norm_softmax_attention_kernels():
    G.test_batchnorm(True, True, True, (2, 4, 4, 64))
    G.test_batchnorm(False, True, False, (2, 5, 3, 32))
    G.test_layernorm(4, 509, False, False)
    G.test_layernorm(70, 256, True, True)
    G.test_layernorm(131, 64, True, True)          # 16-byte kernels: 4 rows per wavefront, ragged last group
    G.test_layernorm(37, 128, False, False)
    G.test_layernorm(9, 512, True, False)
    G.test_layernorm(6, 1024, False, True)
    G.test_layernorm_pair(67, 64, True, False, True)
    G.test_layernorm_pair(40, 64, False, True, False)
    G.test_attention(2, 63, 63, 64, 4)
    G.test_attention(1, 316, 316, 64, 4)            # 16-byte softmax kernels, two column quads per lane
    G.test_attention_dropout_matches_hash_mask()
    G.test_self_attention_packed(2, 40, 64, 4)
    G.test_add_dropout_and_bcast()


@pytest.mark.parametrize('case', [(70, 64, 64, False, True, True), (33, 128, 96, True, False, True), (130, 32, 200, False, True, False),
                                  (65, 64, 128, True, False, False), (40, 30, 64, False, True, False, 0.3, False)])
def test_linear_dropout_epilogue_kernels(case):
    """The DROP variants of the split engine's kernels (64x64 and, for 200 / 128 columns, wider tiles as the planner picks them)
    against GEMM + rih_add_dropout: bit for bit, paired (batched) and single, ReLU or residual; K = 30 falls back."""
    G.check_linear_dropout_epilogue(*case)


@pytest.mark.parametrize('case', [(2, 40, 40, 256, 4, 0.0), (1, 70, 33, 128, 4, 0.1), (1, 33, 130, 64, 4, 0.05), (2, 1, 5, 32, 2, 0.0)])
def test_flash_attention_kernels(case):
    G.test_flash_attention_equals_three_kernel_path(*case)


def test_graph_and_resampling_kernels():
    G.test_cheby_gather_project()
    G.test_pool_upsample_layout()
    G.test_resample_hrnet(2, 4, 4, 32)


def test_fused_attention_forward(monkeypatch):
    """csrc/rih_attn.hip behind ops.FUSED_ATTN -- forward (one launch: QK^T, row softmax on the MFMA accumulator
    layout, dropout, PV) and the query side of the backward (one launch: dO V^T, softmax backward with the regenerated
    mask, dS K): outputs, gradients and the dropout masks against torch -- head dims 16 / 32 / 64, ragged key counts,
    several 128-row query blocks."""
    from renderih_amd import ops
    calls = []
    monkeypatch.setattr(ops, 'FUSED_ATTN', True)
    real = ops._L().rih_attention_fwd_fused
    monkeypatch.setattr(ops._L(), 'rih_attention_fwd_fused', lambda *a: (calls.append(a[9]), real(*a))[1], raising=False)
    bcalls = []
    breal = ops._L().rih_attention_bwd_dq_fused
    monkeypatch.setattr(ops._L(), 'rih_attention_bwd_dq_fused', lambda *a: (bcalls.append(a[9]), breal(*a))[1],
                        raising=False)
    kcalls = []
    kreal = ops._L().rih_attention_bwd_dkv_fused
    monkeypatch.setattr(ops._L(), 'rih_attention_bwd_dkv_fused', lambda *a: (kcalls.append(a[8]), kreal(*a))[1],
                        raising=False)
    G.test_attention(2, 63, 63, 64, 4)              # d = 16
    G.test_attention(1, 150, 190, 128, 4)           # d = 32, two query blocks, Sk not a multiple of 32
    G.test_attention(1, 127, 127, 256, 4)           # d = 64
    G.test_attention_dropout_matches_hash_mask()
    G.test_self_attention_packed(2, 40, 64, 4)      # q / k / v read in place from the packed projection
    G.test_cross_attention_packed(1, 63, 128, 4)
    assert {16, 32, 64} <= set(calls), calls
    assert {16, 32, 64} <= set(bcalls), bcalls      # the query-side backward kernel ran too (gradients checked above)
    assert {16, 32, 64} <= set(kcalls), kcalls      # ... and the key-side one


def test_fused_attention_edge_shapes(monkeypatch):
    """Tails of the fused attention kernels: one query row, one key, a key count of exactly 320 (the limit: ten MFMA tiles),
    129 query rows (a second 128-row block with a single valid row), head dim 64 with few tokens."""
    from renderih_amd import ops
    monkeypatch.setattr(ops, 'FUSED_ATTN', True)
    G.test_attention(1, 1, 1, 64, 4)
    G.test_attention(1, 3, 320, 64, 4)
    G.test_attention(2, 129, 33, 64, 4)
    G.test_attention(1, 5, 7, 256, 4)
    calls = []                                        # beyond the limit the three-launch path must take over
    real = ops._L().rih_attention_fwd_fused
    monkeypatch.setattr(ops._L(), 'rih_attention_fwd_fused', lambda *a: (calls.append(a), real(*a))[1], raising=False)
    G.test_attention(1, 4, 321, 64, 4)
    assert not calls


@pytest.mark.parametrize('B', [1, 2])
def test_mano_kernels(B):
    TMANO.test_mano_matches_oracle(B)


def test_mano_backward_chunk_groups(monkeypatch):
    """The tile-major blend kernel with more than one chunk group and a chunk loop: 40 hands = 3 chunks on '26 CUs' = 2 groups."""
    monkeypatch.setenv('HIPCPU_CUS', '26')
    TMANO.test_mano_backward_batch_independence(40)


def test_mano_kernels_reference_golden():
    TMANO.test_mano_matches_reference_golden('left')


def test_mano_fused_forward_variants_and_cache():
    TMANO.test_mano_inference_without_workspace_and_two_kernel_variant(0)
    TMANO.test_mano_inference_without_workspace_and_two_kernel_variant(1)
    TMANO.test_mano_inference_without_workspace_and_two_kernel_variant(2)      # hand-chunk-major form of the fused kernel
    TMANO.test_mano_inference_without_workspace_and_two_kernel_variant(3)
    TMANO.test_mano_reads_mutated_shapedirs()
    TMANO.test_mano_matches_oracle(17)


def test_mesh_loss_kernel():
    TL.test_fused_loss_matches_reference_golden(60)


def test_metrics_and_pose_head_kernels():
    TMET._check_against_oracle(CPU)
    TP._pose_head_kernels_vs_oracle(CPU)


def test_gemm_descriptor_fuzz_against_emulator():
    """Differential test over the rih_gemm descriptor space: the real kernels (host build: general and split-engine fast
    path, all tiles, both engines, im2col / transposed gathers, strides, padding, leading-dimension padding, split-K,
    bias / residual / ReLU / alpha, the all-ones row, two-slice batches) against the independent numpy restatement of the
    ABI in tests/abi_emulator.py, including that nothing outside the [M][N] window of C is written."""
    import ctypes as C
    import numpy as np
    from abi_emulator import EmulatedLib
    from host_kernels import load
    from renderih_amd._lib import GemmDesc
    host, emu = load(), EmulatedLib()
    # HIPCPU_FUZZ_SEED / HIPCPU_FUZZ_ITERS: other draws; HIPCPU_FUZZ_EXACT=1: operands allocated at their exact extent (no slack
    # floats behind them), so that under AddressSanitizer (tests/hipcpu/README.md) a read past an operand's end is reported
    rs = np.random.RandomState(int(os.environ.get('HIPCPU_FUZZ_SEED', '5')))
    slack = 0 if os.environ.get('HIPCPU_FUZZ_EXACT', '0') == '1' else 8
    cdiv = lambda a, b: -(-a // b)
    checked = 0
    for it in range(int(os.environ.get('HIPCPU_FUZZ_ITERS', '60'))):
        engine, a_mode, b_mode = int(rs.randint(0, 3)), int(rs.randint(0, 2)), int(rs.randint(0, 2))
        conv = rs.rand() < 0.5
        nb1 = 1
        if conv:
            Nimg, H, W = int(rs.randint(1, 3)), int(rs.randint(3, 9)), int(rs.randint(3, 9))
            Cin, KH, KW = int(rs.choice([4, 8, 32, 36, 64])), int(rs.choice([1, 2, 3])), int(rs.choice([1, 2, 3]))
            stride, pad = int(rs.choice([1, 1, 2])), int(rs.randint(0, 2))
            Ho, Wo = (H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1
            if Ho < 1 or Wo < 1:
                continue
            lda = Cin + int(rs.choice([0, 0, 4]))
            pix, taps = Nimg * Ho * Wo, KH * KW
            M, K = (pix, taps * Cin) if a_mode == 0 else (taps * Cin, pix)
            a_size = Nimg * H * W * lda
            geom = (H, W, Cin, Ho, Wo, KH, KW, stride, 1, pad, pad)
        else:
            M, K = int(rs.randint(1, 150)), int(rs.choice([4, 8, 20, 32, 36, 64, 100]))
            lda = (K if a_mode == 0 else M) + int(rs.choice([0, 0, 4, 1]))
            a_size = (M if a_mode == 0 else K) * lda
            geom = (1, 1, (K if a_mode == 0 else M), 1, 1, 1, 1, 1, 1, 0, 0)
            nb1 = int(rs.choice([1, 1, 2]))
        N = int(rs.choice([3, 8, 24, 40, 64, 72, 130]))
        ldb = (N if b_mode == 0 else K) + int(rs.choice([0, 0, 4]))
        b_size = (K if b_mode == 0 else N) * ldb
        splitk, kchunk = 1, 0
        if rs.rand() < 0.3 and K >= 64:
            kchunk = cdiv(cdiv(K, 2), 32) * 32
            splitk = cdiv(K, kchunk)
        ldc = N + int(rs.choice([0, 0, 4]))
        tile = int(rs.choice([0, 1, 2, 3])) if (engine == 0 or N <= 32) else int(rs.choice([0, 1, 2]))
        if engine == 2 and tile == 3:
            tile = 2
        plain_ep = splitk == 1
        use_bias, use_R, relu = plain_ep and rs.rand() < 0.5, plain_ep and rs.rand() < 0.3, plain_ep and rs.rand() < 0.3
        Mp, ones_row = M, 0
        if a_mode == 1 and rs.rand() < 0.4:
            Mp, ones_row = M + 4, M
        alpha = float(rs.choice([1.0, 0.5]))
        A = rs.randn(nb1 * a_size + slack).astype(np.float32)
        B = rs.randn(nb1 * b_size + slack).astype(np.float32)
        sB1 = int(rs.choice([0, b_size])) if nb1 > 1 else 0
        bias, R = rs.randn(nb1 * N).astype(np.float32), rs.randn(nb1 * Mp * (N + 4)).astype(np.float32)
        amax = None
        if engine == 2:
            # engine 2 (two-term fp16 split): operand magnitudes far outside the fp16 range, bounds as rih_absmax would write
            # them -- or a loose upper bound, or (values inside +-2^15 only) no bound at all
            mode = int(rs.randint(0, 3))
            if mode < 2:
                A *= np.float32(rs.choice([1e-6, 1.0, 3e4, 1e7]))
                B *= np.float32(rs.choice([1e-9, 1e-3, 1.0, 2e5]))
                loose = np.float32(1.0 if mode == 0 else 37.0)
                amax = (np.zeros(2048, np.float32), np.zeros(2048, np.float32))      # bound blocks: 64 partial maxima, stride 32
                amax[0][32 * int(rs.randint(0, 64))] = np.abs(A).max() * loose
                amax[1][32 * int(rs.randint(0, 64))] = np.abs(B).max() * loose
                j = 32 * int(rs.randint(0, 64))                                       # a second, smaller partial maximum
                amax[0][j] = max(amax[0][j], amax[0].max() * np.float32(rs.rand()))
            bias *= np.float32(np.abs(A).max() * np.abs(B).max())
            R *= np.float32(np.abs(A).max() * np.abs(B).max())
        outs = []
        for lib in (host, emu):
            Cc = np.full(nb1 * splitk * Mp * ldc + 8, 7.0, np.float32)
            d = GemmDesc()
            d.A, d.B, d.C = A.ctypes.data, B.ctypes.data, Cc.ctypes.data
            d.bias, d.R = (bias.ctypes.data if use_bias else 0), (R.ctypes.data if use_R else 0)
            d.M, d.N, d.K, d.lda, d.ldb, d.ldc, d.ldr = Mp, N, K, lda, ldb, ldc, N + 4
            d.a_mode, d.b_mode, d.nb1, d.nb2 = a_mode, b_mode, nb1, 1
            d.sA1, d.sB1, d.sC1 = a_size, sB1, splitk * Mp * ldc
            d.sBias1, d.sR1 = (N if rs.rand() < 2 else 0), Mp * (N + 4)
            d.splitk, d.kchunk, d.sCsplit, d.alpha, d.relu = splitk, kchunk, Mp * ldc, alpha, 1 if relu else 0
            (d.H, d.W, d.Cin, d.Ho, d.Wo, d.KH, d.KW, d.strideA, d.upS, d.padH, d.padW) = geom
            d.tile, d.engine, d.ones_row = tile, engine, ones_row
            if amax is not None:
                d.amax_a, d.amax_b = amax[0].ctypes.data, amax[1].ctypes.data
            if engine == 2:
                assert host.rih_gemm_engine(C.byref(d)) == emu.rih_gemm_engine(C.byref(d)), it
            outs.append((lib.rih_gemm(C.byref(d), None), Cc))
        (rc0, c0), (rc1, c1) = outs
        assert rc0 == 0 and rc1 == 0, (it, rc0, rc1)
        rows = ones_row + 1 if ones_row else Mp
        what = 'case %d: e%d a%d b%d conv=%s M%d N%d K%d lda%d ldb%d ldc%d nb%d sk%d tile%d geom=%s' % (
            it, engine, a_mode, b_mode, conv, Mp, N, K, lda, ldb, ldc, nb1, splitk, tile, geom)
        v0 = c0[:nb1 * splitk * Mp * ldc].reshape(nb1 * splitk, Mp, ldc)
        v1 = c1[:nb1 * splitk * Mp * ldc].reshape(nb1 * splitk, Mp, ldc)
        sc = max(float(np.abs(v1[:, :rows, :N]).max()), 1e-6)
        assert np.allclose(v0[:, :rows, :N], v1[:, :rows, :N], rtol=2e-5, atol=2e-5 * sc), what
        assert (v0[:, :, N:] == 7.0).all() and (c0[nb1 * splitk * Mp * ldc:] == 7.0).all(), 'stray write, ' + what
        checked += 1
    assert checked >= 50


def test_batched_entry_points_fuzz_against_emulator():
    """Differential test of the descriptor-list entry points (descriptors by value in the kernel argument, several launches when
    the list exceeds a pack): rih_splitk_reduce_multi (pack 60), rih_ln_param_final_multi (100), rih_pack_conv_weight_multi (56)
    and rih_adam_multi -- list lengths on both sides of the pack sizes, ragged shapes, bias rows, accumulate, channel padding --
    real kernels against the numpy restatement; and the statistics epilogue of rih_gemm (rih_gemm_desc.stats) incl. a ragged
    last row block against per-block numpy means / M2."""
    import ctypes as C
    import numpy as np
    from abi_emulator import EmulatedLib
    from host_kernels import load
    from renderih_amd._lib import GemmDesc, LnFinalDesc, PackDesc, ReduceDesc
    host, emu = load(), EmulatedLib()
    rs = np.random.RandomState(11)
    f32 = np.float32

    # ---- split-K reductions
    for n in (60, 125):
        keep, descs = [], []
        for _ in range(n):
            S, taps, cin, cv = int(rs.randint(1, 6)), int(rs.choice([1, 4, 9])), int(rs.choice([4, 8, 12])), 0
            cv = cin if rs.rand() < 0.7 else cin - 1
            M, N = taps * cin, int(rs.choice([3, 8, 33, 40]))
            has_db, acc = rs.rand() < 0.4, int(rs.rand() < 0.3)
            Mp = M + 4 if has_db else M
            P = rs.randn(S, Mp, N).astype(f32)
            init = rs.randn(N * cv * taps).astype(f32)
            keep.append((P, init, has_db))
            descs.append((S, Mp, M, N, cin, taps, cv, acc))
        outs = []
        for lib in (host, emu):
            arr = (ReduceDesc * n)()
            dst = [k[1].copy() for k in keep]
            db = [np.full(d[3], 7.0, f32) for d in descs]
            for a, (P, _, has_db), d, o, b in zip(arr, keep, descs, dst, db):
                a.P, a.dst, a.db = P.ctypes.data, o.ctypes.data, (b.ctypes.data if has_db else 0)
                a.S, a.Mp, a.M, a.N, a.Cin, a.taps, a.CinValid, a.accumulate = d
            assert lib.rih_splitk_reduce_multi(arr, n, None) == 0
            outs.append((dst, db))
        for i in range(n):
            assert np.allclose(outs[0][0][i], outs[1][0][i], rtol=1e-5, atol=1e-5), ('reduce', n, i, descs[i])
            assert np.allclose(outs[0][1][i], outs[1][1][i], rtol=1e-5, atol=1e-5), ('bias gradient', n, i, descs[i])

    # ---- LayerNorm parameter-gradient finals
    for n in (1, 101):
        shapes = [(int(rs.choice([5, 64, 200])), int(rs.randint(1, 40))) for _ in range(n)]      # (D, nblk)
        ws = [rs.randn(nb, 2, D).astype(f32) for D, nb in shapes]
        outs = []
        for lib in (host, emu):
            arr = (LnFinalDesc * n)()
            dg = [np.zeros(D, f32) for D, _ in shapes]
            db = [np.zeros(D, f32) for D, _ in shapes]
            for a, w, g, b, (D, nb) in zip(arr, ws, dg, db, shapes):
                a.ws, a.dg, a.db, a.D, a.nblk = w.ctypes.data, g.ctypes.data, b.ctypes.data, D, nb
            assert lib.rih_ln_param_final_multi(arr, n, None) == 0
            outs.append((dg, db))
        for i in range(n):
            assert np.allclose(outs[0][0][i], outs[1][0][i], rtol=1e-5, atol=1e-5) and \
                np.allclose(outs[0][1][i], outs[1][1][i], rtol=1e-5, atol=1e-5), ('ln final', n, i, shapes[i])

    # ---- weight packs
    for n in (56, 57):
        items = []
        for _ in range(n):
            Cout, Cin, KH = int(rs.choice([3, 8, 20])), int(rs.choice([3, 4, 9])), int(rs.choice([1, 2, 3, 4]))
            KW = KH
            cpad = (Cin + 3) // 4 * 4
            mode = int(rs.randint(0, 2))
            if mode == 0:
                f = (Cout, Cin, KH, KW, cpad, 0, 0, 0, 1, KH, KW)
                size = KH * KW * cpad * Cout
            else:
                step = int(rs.choice([1, 2])) if KH > 1 else 1
                kh0, kw0 = int(rs.randint(0, step)), int(rs.randint(0, step))
                Th, Tw = len(range(kh0, KH, step)), len(range(kw0, KW, step))
                if Th == 0 or Tw == 0:
                    kh0 = kw0 = 0
                    Th, Tw = len(range(0, KH, step)), len(range(0, KW, step))
                f = (Cout, Cin, KH, KW, cpad, 1, kh0, kw0, step, Th, Tw)
                size = Th * Tw * Cout * cpad
            items.append((rs.randn(Cout, Cin, KH, KW).astype(f32), f, size))
        outs = []
        for lib in (host, emu):
            arr = (PackDesc * n)()
            dst = [np.full(sz + 4, 7.0, f32) for _, _, sz in items]
            for a, (w, f, _), o in zip(arr, items, dst):
                a.w, a.dst = w.ctypes.data, o.ctypes.data
                (a.Cout, a.Cin, a.KH, a.KW, a.CinPad, a.mode, a.kh0, a.kw0, a.step, a.Th, a.Tw) = f
            assert lib.rih_pack_conv_weight_multi(arr, n, None) == 0
            outs.append(dst)
        for i in range(n):
            assert np.array_equal(outs[0][i], outs[1][i]), ('pack', n, i, items[i][1])
            assert (outs[0][i][items[i][2]:] == 7.0).all()

    # ---- statistics epilogue: per-row-block (mean, M2) of the stored tile, ragged last block
    for (M, N, K, tile, relu) in ((150, 72, 64, 2, 1), (300, 128, 96, 1, 0), (128, 40, 32, 0, 0)):
        A, B = rs.randn(M, K).astype(f32) + 3.0, rs.randn(N, K).astype(f32)
        d = GemmDesc()
        Cc = np.zeros((M, N), f32)
        d.A, d.B, d.C = A.ctypes.data, B.ctypes.data, Cc.ctypes.data
        d.M, d.N, d.K, d.lda, d.ldb, d.ldc, d.ldr = M, N, K, K, K, N, N
        d.a_mode, d.b_mode, d.nb1, d.nb2, d.splitk, d.alpha, d.relu = 0, 1, 1, 1, 1, 1.0, relu
        (d.H, d.W, d.Cin, d.Ho, d.Wo, d.KH, d.KW, d.strideA, d.upS, d.padH, d.padW) = (1, 1, K, 1, 1, 1, 1, 1, 1, 0, 0)
        d.tile, d.engine = tile, 1
        rp = host.rih_gemm_stats_rows(C.byref(d))
        assert rp == emu.rih_gemm_stats_rows(C.byref(d)) == (64 if tile < 2 else 32)
        T = -(-M // rp)
        st = np.full((T, 2, N), 7.0, f32)
        d.stats = st.ctypes.data
        assert host.rih_gemm(C.byref(d), None) == 0
        ref = A.astype(np.float64) @ B.astype(np.float64).T
        if relu:
            ref = np.maximum(ref, 0)
        for t in range(T):
            blk = ref[t * rp:(t + 1) * rp]
            assert np.allclose(st[t, 0], blk.mean(0), rtol=1e-5, atol=1e-5), ('block mean', M, N, t)
            assert np.allclose(st[t, 1], ((blk - blk.mean(0)) ** 2).sum(0), rtol=2e-4, atol=1e-4), ('block M2', M, N, t)


def test_grouped_gemm_fuzz_against_emulator():
    """Differential test of the grouped launch (rih_gemm_multi_pack / _launch: problems of one kernel variant in ONE launch, table
    in "device" memory, block -> problem map, padding blocks): random weight-gradient problems -- plain and conv-gather A
    operands, bias rows (ones_row), split-K with ragged last slices, paired batches (nb1 = 2) -- real kernel against the numpy
    restatement of rih_gemm per problem; both tile classes."""
    import ctypes as C
    import numpy as np
    from abi_emulator import EmulatedLib
    from host_kernels import load
    from renderih_amd._lib import GemmDesc
    host, emu = load(), EmulatedLib()
    rs = np.random.RandomState(23)
    f32 = np.float32
    for tile, plain, n in ((2, True, 9), (2, False, 5), (0, True, 3), (0, False, 2)):
        probs = []
        for _ in range(n):
            nb = int(rs.choice([1, 2]))
            N = int(rs.choice([36, 64, 72] if tile == 2 else [72, 136]))
            if plain:
                Kpix = int(rs.choice([32, 96, 132, 260]))
                Cin = int(rs.choice([8, 40, 68] if tile == 2 else [132]))
                geom = (1, 1, Cin, 1, 1, 1, 1, 1, 1, 0, 0)
                taps, xrows = 1, Kpix
            else:
                H = W = int(rs.choice([4, 8]))
                imgs, Cin, taps = int(rs.choice([1, 2])), int(rs.choice([8, 16] if tile == 2 else [16])), 9
                geom = (H, W, Cin, H, W, 3, 3, 1, 1, 1, 1)
                Kpix, xrows = imgs * H * W, imgs * H * W
            M = taps * Cin
            bias = bool(rs.rand() < 0.5)
            Mp = M + 4 if bias else M
            sk = int(rs.choice([1, 2, 3]))
            kchunk = -(-(-(-Kpix // sk)) // 32) * 32
            sk = -(-Kpix // kchunk)
            x = rs.randn(nb, xrows, Cin).astype(f32)
            dy = rs.randn(nb, Kpix, N).astype(f32)
            probs.append((nb, M, Mp, N, Kpix, Cin, geom, bias, sk, kchunk, x, dy))
        outs = []
        for lib in (host, emu):
            arr = (GemmDesc * n)()
            parts = []
            for d, (nb, M, Mp, N, Kpix, Cin, geom, bias, sk, kchunk, x, dy) in zip(arr, probs):
                part = np.full((nb, sk, Mp, N), 7.0, f32)
                parts.append(part)
                d.A, d.B, d.C = x.ctypes.data, dy.ctypes.data, part.ctypes.data
                d.M, d.N, d.K = Mp, N, Kpix
                d.lda, d.ldb, d.ldc = Cin, N, N
                d.a_mode, d.b_mode, d.nb1, d.nb2 = 1, 0, nb, 1
                d.sA1, d.sB1, d.sC1 = x[0].size, dy[0].size, sk * Mp * N
                d.splitk, d.kchunk, d.sCsplit = sk, (kchunk if sk > 1 else 0), Mp * N
                d.alpha, d.tile, d.engine = 1.0, tile, 1
                (d.H, d.W, d.Cin, d.Ho, d.Wo, d.KH, d.KW, d.strideA, d.upS, d.padH, d.padW) = geom
                d.ones_row = M if bias else 0
                assert lib.rih_gemm_multi_variant(C.byref(d)) == tile * 8 + 4 + (1 if plain else 0), (tile, plain)
            nbytes = int(lib.rih_gemm_multi_table_bytes(arr, n))
            assert nbytes > 0
            table = np.zeros(nbytes + 16, np.uint8)
            tp = (table.ctypes.data + 15) & ~15
            total = C.c_int32(0)
            v = lib.rih_gemm_multi_pack(arr, n, tp, C.byref(total))
            assert v == tile * 8 + 4 + (1 if plain else 0) and total.value % 8 == 0 and total.value >= 8
            assert lib.rih_gemm_multi_launch(tp, v, total.value, None) == 0
            outs.append(parts)
        for i, (nb, M, Mp, N, Kpix, Cin, geom, bias, sk, kchunk, x, dy) in enumerate(probs):
            a, b = outs[0][i], outs[1][i]
            rows = M + 1 if bias else M          # rows past the all-ones row are junk by contract
            scale = np.abs(b[:, :, :rows]).max() + 1e-6
            assert np.abs(a[:, :, :rows] - b[:, :, :rows]).max() <= 2e-6 * scale * max(1.0, Kpix ** 0.5), \
                ('grouped gemm', tile, plain, i, (nb, M, N, Kpix, sk))
    # a descriptor of another variant, and a mixed list, are refused
    d = GemmDesc()
    assert emu.rih_gemm_multi_variant(d) == -1
    arr = (GemmDesc * 2)()
    for lib in (host,):
        C.memmove(C.byref(arr[0]), C.byref(d), C.sizeof(d))
        assert lib.rih_gemm_multi_table_bytes(arr, 2) == 0


@pytest.mark.skipif(not os.environ.get('HIPCPU_MORE'), reason='set HIPCPU_MORE=1: ~1 min of extra scheduling runs')
@pytest.mark.parametrize('sched', ['reverse', 'random'])
def test_kernels_are_schedule_independent(sched):
    """Missing-barrier detector: the fibers of a block are visited in reverse / random order per scheduling round
    (HIPCPU_SCHED, read once per process -> a child process); a kernel with a data race on its LDS tiles would change
    its result.  Subset: implicit-GEMM fast path, the four-slot-ring kernel, wavefront-reduction kernels, MANO, loss."""
    import subprocess
    env = dict(os.environ, HIPCPU_SCHED=sched)
    sel = ('conv2d_kernels or tile4 or mesh_loss or metrics_and_pose or mano_kernels or mano_fused or graph_and_resampling or '
           'paired_layer or norm_softmax or batched_entry')
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.abspath(__file__), '-q', '-x', '-k', sel, '-p', 'no:cacheprovider'],
                       env=env, capture_output=True, text=True, cwd=os.path.dirname(HERE))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]

