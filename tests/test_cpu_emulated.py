"""CPU validation of the host-side logic (renderih_amd.ops + module tree) against PyTorch/the oracle, with the C ABI
emulated on host memory (tests/abi_emulator.py).  The kernels themselves are proven by the -m gpu tests; this pins
descriptor geometry, packing, strides, split-K plumbing, backward wiring and the module forward graph."""
import numpy as np
import pytest
import torch

import test_gpu_ops as G
from abi_emulator import emulated_abi
from renderih_amd import assets, testing
from renderih_amd.testing import assert_close


@pytest.fixture(autouse=True)
def _emulate(monkeypatch):
    monkeypatch.setattr(G, 'dev', lambda: torch.device('cpu'))
    with emulated_abi():
        yield


@pytest.mark.parametrize('case', G.CONV_CASES[:10])
def test_conv2d_host_logic(case):
    G.test_conv2d(case)


@pytest.mark.parametrize('case', G.LIN_CASES[:5] + G.LIN_CASES[6:7])
def test_linear_host_logic(case):
    G.test_linear(case)


def test_other_ops_host_logic():
    G.test_batchnorm(True, True, True, (2, 16, 16, 64))
    G.test_batchnorm(False, True, False, (3, 7, 9, 256))
    G.test_layernorm(4, 509, False, False)
    G.test_layernorm(300, 256, True, True)
    G.test_attention(2, 127, 127, 256, 4)
    G.test_attention(2, 126, 252, 128, 4)
    G.test_attention_dropout_matches_hash_mask()
    G.test_flash_attention_equals_three_kernel_path(2, 63, 252, 128, 4, 0.1)
    G.test_self_attention_packed(2, 190, 128, 4)
    G.test_cross_attention_packed(2, 63, 256, 4)
    G.test_add_dropout_and_bcast()
    G.test_cheby_gather_project()


def test_paired_layers_host_logic():
    """Strides between separately allocated left/right parameters, stacked QKV operand, batched split-K reduce, grouped
    LayerNorm, shared-input patch conv: the descriptor plumbing of the hands-stacked decoder path."""
    G.test_linear_pair(126, 256, 128, False, True, False)
    G.test_linear_pair(380, 64, 192, False, False, True)
    G.test_linear_pair(33, 12, 6, True, True, False)
    G.test_layernorm_pair(190, 128, False, False, True)
    G.test_layernorm_pair(126, 256, True, True, False)
    G.test_patch_conv_pair(2, 16, 256, 128, 2)
    G.test_patch_conv_pair(2, 16, 32, 48, 4)
    G.test_patch_conv_pair(3, 8, 32, 48, 1)
    G.test_patch_conv_pair(2, 16, 64, 48, 4)
    G.test_deferred_reductions_equal_immediate()
    G.test_grouped_weight_gradients(1)
    G.test_grouped_weight_gradients(2)
    G.test_pack_cache_one_launch_equals_per_call_packs()
    G.test_conv_bn_statistics_from_the_gemm_epilogue((2, 16, 16, 64, 128, 3, 1, 1, False))
    G.test_conv_bn_statistics_from_the_gemm_epilogue((3, 9, 7, 32, 72, 3, 2, 1, True))
    G.test_conv_bn_statistics_from_the_gemm_epilogue((1, 10, 10, 64, 36, 1, 1, 0, False))
    G.test_conv_bn_epilogue_statistics_survive_a_large_mean()
    G.test_conv_bn_statistics_from_the_gemm_epilogue((2, 8, 8, 256, 256, 3, 1, 1, False))
    G.test_adam_one_launch_matches_torch(False, 1e-2)
    G.test_adam_one_launch_matches_torch(True, 1e-2)
    G.test_cross_attention_stacked_and_rows_pair()


@pytest.mark.parametrize('same_graph', [True, False])
def test_paired_decoder_equals_sequential(monkeypatch, same_graph):
    """The hands-stacked decoder (one launch per left/right layer pair, shared heads over the stacked batch) computes
    what the per-hand sequence computes: outputs and every parameter / input gradient, dropout off.  same_graph=False
    exercises the fallback for user-supplied Laplacians that differ between the hands."""
    from renderih_amd import ops
    from renderih_amd.attn import GCN_ResBlock
    from renderih_amd.model import build_model
    dec = build_model(0.0).decoder
    dec.load_state_dict(testing.deterministic_state(dec.state_dict(), seed=5))
    dec.train()
    if not same_graph:
        monkeypatch.setattr(GCN_ResBlock, '_same_graph', lambda self, other: False)
    g = torch.Generator().manual_seed(11)
    x0 = torch.randn(1, dec.gf_dim, generator=g)
    f0 = [torch.randn(1, s, s, 256, generator=g) * 0.5 for s in (8, 16, 32)] + [torch.zeros(1, 64, 64, 256)]

    def run(pair):
        monkeypatch.setattr(ops, 'PAIR_HANDS', pair)
        dec.zero_grad()
        x = x0.clone().requires_grad_(True)
        fm = [t.clone().requires_grad_(True) for t in f0]
        result, params, hands, other = dec(x, fm)
        outs = {}
        for side in ('left', 'right'):
            outs['v3d_' + side] = result['verts3d'][side]
            outs['v2d_' + side] = result['verts2d'][side]
            outs['c3d_' + side] = hands[0]['verts3d'][side]
            # one tensor: the scale alone can be a near-cancelled sum whose round-off is not comparable
            outs['cam_' + side] = torch.cat([params['scale'][side][:, None], params['trans2d'][side]], 1)
            outs['m3d_' + side] = other['verts3d_MANO_list'][side][0]
            outs['m2d_' + side] = other['verts2d_MANO_list'][side][0]
        gen = torch.Generator().manual_seed(3)
        loss = sum((v * torch.randn(v.shape, generator=gen)).sum() for v in outs.values())
        loss.backward()
        grads = {k: p.grad.clone() for k, p in dec.named_parameters() if p.grad is not None}
        grads['x'] = x.grad.clone()
        for i, t in enumerate(fm[:3]):
            grads['fmap%d' % i] = t.grad.clone()
        return {k: v.detach().clone() for k, v in outs.items()}, grads

    o1, g1 = run(True)
    o0, g0 = run(False)
    for k in o0:
        assert_close(o1[k], o0[k], 1e-4, 1e-5, 'paired vs sequential ' + k)
    assert set(g1) == set(g0)
    for k in g0:
        if not testing.is_null_gradient(k):
            assert_close(g1[k], g0[k], 2e-3, 2e-4, 'paired vs sequential grad ' + k)


def test_paired_decoder_runs_with_dropout(monkeypatch):
    from renderih_amd import ops
    from renderih_amd.model import build_model
    dec = build_model(0.05).decoder
    dec.train()
    monkeypatch.setattr(ops, 'PAIR_HANDS', True)
    torch.manual_seed(0)
    x = torch.randn(1, dec.gf_dim)
    fm = [torch.randn(1, s, s, 256) * 0.5 for s in (8, 16, 32)] + [torch.zeros(1, 64, 64, 256)]
    result = dec(x, fm)[0]
    s = sum(v.sum() for v in result['verts3d'].values())
    s.backward()
    assert torch.isfinite(s) and all(torch.isfinite(p.grad).all() for p in dec.parameters() if p.grad is not None)


def test_conv3x3_halo_host_logic():
    """ops.conv3x3_halo / _h2_weight on the emulated ABI: descriptor fields, H2 operand for forward and data gradient, statistics
    blocks in patch order, the fall-back for shapes the kernel does not take."""
    G.test_conv3x3_halo((2, 16, 32, 64, 128, False, True))
    G.test_conv3x3_halo((1, 8, 64, 32, 64, True, True))
    G.test_conv3x3_halo((2, 16, 16, 64, 96, True, True))           # 16 x 16 patches, 32-channel blocks (32-row statistics blocks)
    G.test_conv2d((2, 8, 8, 32, 32, 3, 1, 1, False, True))         # W = 8: not a whole 32-pixel patch -> rih_gemm
    G.test_conv3x3_halo_with_skip_gradient((2, 8, 32, 32, 32))     # the skip gradient as the kernel's residual (ABI 19)
    G.test_conv3x3_halo_with_skip_gradient((1, 16, 16, 64, 64))
    G.test_conv3x3_residual_preconditions()


def test_panel_host_logic():
    """ops.panel_gemm on the emulated ABI: forward and data-gradient H2 operands of a 1x1 weight, the skip path's gradient as
    residual, statistics blocks of 32 / 64 rows."""
    G.test_panel_1x1((1, 16, 16, 64, 256, False, True))
    G.test_panel_1x1((1, 16, 16, 128, 128, True, True))
    G.test_panel_1x1((1, 16, 8, 64, 64, False, True))


def test_rows_host_logic(monkeypatch):
    """ops.rows_gemm on the emulated ABI: dispatch from Conv2dFn (forward and data gradient with the skip residual), H2 operands,
    statistics blocks of 32 / 64 rows, the library-side precondition check with its rih_gemm fallback."""
    G.test_rows_1x1((2, 16, 16, 256, 64, False, True))
    G.test_rows_1x1((1, 16, 8, 96, 128, True, True))
    G.test_rows_kernel_is_taken_and_falls_back(monkeypatch)
    G.test_stem_conv((2, 32, 32, True, True))


def test_conv1x1_cat_host_logic():
    """ops.conv1x1_cat on the emulated ABI: segment descriptors, per-part data gradients, column-slice weight gradients."""
    for eng in (1, 2):
        G.test_conv1x1_cat((2, 8, 8, (64, 32, 96), 128, True), eng)
        G.test_conv1x1_cat((1, 5, 7, (32, 64), 40, False), eng)
    # engine 0 (no segmented operand in rih_gemm: the emulator answers RIH_EINVAL like the library) and ragged channel counts fall
    # back to the concatenation; a consumer without gradient pre-gating gets the ReLU gate from the function
    G.test_conv1x1_cat((1, 5, 7, (32, 64), 40, True), 0)
    G.test_conv1x1_cat((1, 4, 4, (32, 40), 32, True), 2)
    G.test_conv1x1_cat((1, 4, 4, (64, 32), 64, 'ungated'), 1)


def test_fused_attention_host_logic(monkeypatch):
    """ops.FUSED_ATTN (off by default): pointer / pitch plumbing of rih_attention_fwd_fused / _bwd_dq_fused."""
    from renderih_amd import ops
    monkeypatch.setattr(ops, 'FUSED_ATTN', True)
    G.test_attention(2, 63, 63, 64, 4)
    G.test_attention_dropout_matches_hash_mask()
    G.test_self_attention_packed(2, 40, 64, 4)
    G.test_cross_attention_packed(2, 63, 128, 4)
    G.test_cross_attention_stacked_and_rows_pair()


def test_pool_layout_host_logic(monkeypatch):
    G.test_pool_upsample_layout()
    G.test_resample_hrnet(4, 5, 7, 32)
    G.test_resample_hrnet(2, 8, 8, 64)


def test_hrnet_host_logic_matches_oracle():
    """HRNet-W32 variant, B=1, eval mode (running-statistics BN: well conditioned), forward only: the module graph
    (transitions, fuse sums, heads, mid head) against the oracle at 1e-4."""
    from oracle import net_oracle
    from renderih_amd.model import build_model
    m = build_model(0.0, 'hrnet32')
    sd = testing.deterministic_state(m.state_dict(), seed=3)
    m.load_state_dict(sd)
    m.eval()
    img = testing.seeded_image(1, 4)
    with torch.no_grad():
        got = testing.flatten_outputs(m(img))
        graph = net_oracle.graph_from_dicts(assets.load_graph_dict('left'), assets.load_graph_dict('right'))
        want = testing.flatten_outputs(net_oracle.handnet_forward(sd, graph, img, training=False))
    for k in want:
        if not k.startswith('params.'):
            assert testing.rel_err(got[k], want[k]) < 1e-4, (k, testing.rel_err(got[k], want[k]))


def test_model_host_logic_matches_oracle():
    """Whole network, B=2, train mode, dropout 0: forward outputs and all parameter gradients.  Train-mode BN over
    <=128 samples amplifies fp32 round-off to ~1e-4, so the bar is anchored on the fp64 oracle (testing.py)."""
    from oracle import net_oracle
    from renderih_amd.model import build_model
    m = build_model(0.0)
    sd = testing.deterministic_state(m.state_dict(), seed=3)
    m.load_state_dict(sd)
    m.train()
    img = testing.seeded_image(2, 4)
    got = testing.flatten_outputs(m(img))
    graph = net_oracle.graph_from_dicts(assets.load_graph_dict('left'), assets.load_graph_dict('right'))
    w32, g32 = net_oracle.run(sd, graph, img, True, torch.float32, True)
    w64, g64 = net_oracle.run(sd, graph, img, True, torch.float64, True)
    for k in w64:
        # k = 6: at B = 2 the distance of an fp32 implementation from fp64 is itself a noisy sample -- with the BatchNorm
        # statistics taken from the GEMM epilogue (more exact than the shifted sums) this draw lands at 4.4 x the CPU fp32
        # run's own distance on the per-image scale head, with the standalone statistics pass at 3.0 x
        testing.assert_fp32_equivalent(got[k], w32[k], w64[k], k=6.0, what=k)
    net_oracle.scalar_loss(_unflatten(got)).backward()
    from test_gpu_model import _grad_report
    params = [(k, p.grad) for k, p in m.named_parameters() if p.grad is not None]
    assert {k for k, _ in params} == set(g64.keys())
    _grad_report(params, g32, g64)
    assert int(m.encoder.resnet.bn1.num_batches_tracked) == 1


def test_dead_mid_convolution_skip_host_logic():
    """encoder.resnet_mid.forward(drop_last=True) (RIH_SKIP_DEAD_MID, see tests/test_gpu_model.py for the whole model on the
    GPU) on the emulated ABI, module alone on small maps: the three live feature maps, the global feature, their gradients
    and every buffer -- the dropped branch's BatchNorm running statistics included -- equal the full computation."""
    from renderih_amd.encoder import resnet_mid
    g = torch.Generator().manual_seed(5)
    B = 2
    img_f = [torch.randn(B, s, s, c, generator=g) for s, c in zip((1, 2, 4, 8), (2048, 1024, 512, 256))]
    hms_f = [torch.randn(B, s, s, 128, generator=g) for s in (1, 2, 4, 8)]
    dp_f = [torch.randn(B, s, s, 128, generator=g) for s in (1, 2, 4, 8)]
    ref = resnet_mid('resnet50', in_fmapDim=[128] * 4, out_fmapDim=[256] * 4)
    sd = testing.deterministic_state(ref.state_dict(), seed=2)
    res = {}
    for drop in (False, True):
        m = resnet_mid('resnet50', in_fmapDim=[128] * 4, out_fmapDim=[256] * 4)
        m.load_state_dict(sd)
        m.train()
        ins = [[t.clone().requires_grad_() for t in lst] for lst in (img_f, hms_f, dp_f)]
        gf, fm = m(*ins, drop_last=True) if drop else m(*ins)
        assert len(fm) == 4 and (fm[3] is None) == drop
        (gf.sum() + sum((f * f).sum() for f in fm[:3])).backward()
        out = [gf.detach().clone()] + [f.detach().clone() for f in fm[:3]]
        grads = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
        dins = [t.grad.clone() for lst in ins for t in lst if t.grad is not None]
        state = {k: v.clone() for k, v in m.state_dict().items()}
        m.eval()
        with torch.no_grad():
            gf2, fm2 = m(img_f, hms_f, dp_f, drop_last=True) if drop else m(img_f, hms_f, dp_f)
        assert (fm2[3] is None) == drop
        res[drop] = (out, grads, dins, state, [gf2] + list(fm2[:3]))
    full, lean = res[False], res[True]
    assert int(full[3]['convs.3.2.num_batches_tracked']) == 1 and float(full[3]['convs.3.2.running_mean'].abs().max()) > 0
    assert set(full[1]) == set(lean[1]) and not any(k.startswith('convs.3.') for k in full[1])
    for a, b in zip(full[0] + full[2] + full[4], lean[0] + lean[2] + lean[4]):
        assert torch.equal(a, b)
    for part in (1, 3):
        for k in full[part]:
            assert torch.equal(full[part][k], lean[part][k]), k


def test_gemm_dropout_epilogue_model_host_logic(monkeypatch):
    """ops.GEMM_DROPOUT (RIH_GEMM_DROPOUT=1) through the whole network, training mode, dropout 0.1, emulated ABI: the dropout
    sites behind the decoder's Linears ride in the GEMM epilogue and draw the same mask streams in the same order, so every
    output and parameter gradient equals the two-launch form bit for bit."""
    from oracle import net_oracle
    from renderih_amd import ops
    from renderih_amd.model import build_model
    img = testing.seeded_image(1, 4)
    res = {}
    for fuse in (False, True):
        monkeypatch.setattr(ops, 'GEMM_DROPOUT', fuse)
        m = build_model(0.1)
        m.load_state_dict(testing.deterministic_state(m.state_dict(), seed=3))
        m.train()
        torch.manual_seed(99)               # DropCtx draws its base seed from the host generator
        out = m(img)
        net_oracle.scalar_loss(out).backward()
        res[fuse] = (testing.flatten_outputs(out), {k: p.grad for k, p in m.named_parameters() if p.grad is not None})
    for a, b in zip(res[False], res[True]):
        assert set(a) == set(b)
        for k in a:
            assert torch.equal(a[k], b[k]), k


def test_family_b_host_logic_matches_oracle():
    """Second model family (renderih_amd/lijun.py = common/myhand/lijun_model_graph.HandNET_GCN), B=2, train mode,
    dropout 0: forward outputs and all parameter gradients against the oracle, fp64-anchored like the first family."""
    from oracle import net_oracle
    from renderih_amd.lijun import build_graph_model
    m = build_graph_model(0.0)
    sd = testing.deterministic_state(m.state_dict(), seed=11)       # a draw without near-zero ReLU inputs that flip
    m.load_state_dict(sd)                                            # between fp32 evaluation orders (_grad_report)
    m.train()
    img = testing.seeded_image(2, 12)
    out = m(img)
    assert out[3]['verts3d_MANO_list'] == {'left': [], 'right': []} and 'hms' not in out[3]
    got = testing.flatten_outputs(out)
    graph = net_oracle.graph_from_dicts(assets.load_graph_dict('left'), assets.load_graph_dict('right'))
    w32, g32 = net_oracle.run(sd, graph, img, True, torch.float32, True)
    w64, g64 = net_oracle.run(sd, graph, img, True, torch.float64, True)
    assert set(got) == set(w64)
    for k in w64:
        testing.assert_fp32_equivalent(got[k], w32[k], w64[k], what=k)
    net_oracle.scalar_loss(out).backward()
    from test_gpu_model import _grad_report
    params = [(k, p.grad) for k, p in m.named_parameters() if p.grad is not None]
    assert {k for k, _ in params} == set(g64.keys())
    _grad_report(params, g32, g64)
    assert int(m.encoder.resnet.bn1.num_batches_tracked) == 1


def _unflatten(f):
    sides = ('left', 'right')
    return ({'verts3d': {s: f['result.verts3d.' + s] for s in sides}, 'verts2d': {s: f['result.verts2d.' + s] for s in sides}},
            {'scale': {s: f['params.scale.' + s] for s in sides}, 'trans2d': {s: f['params.trans2d.' + s] for s in sides}},
            [{'verts3d': {s: f['hand0.verts3d.' + s] for s in sides}, 'verts2d': {s: f['hand0.verts2d.' + s] for s in sides}}],
            {'hms': f['other.hms'], 'mask': f['other.mask'], 'dense': f['other.dense']})


@pytest.mark.parametrize('epoch', [0, 60])
def test_fused_loss_host_logic_matches_reference_golden(epoch):
    """Host side of the fused mesh loss (topology tables, weights, autograd plumbing) with the kernel emulated, against
    the values and gradients of the reference's own core/Loss.py (tests/golden/loss.npz)."""
    import os
    from test_oracle_golden import _loss_inputs, GOLDEN
    from renderih_amd.loss import FusedMeshLoss, calc_loss_GCN_fused
    z = np.load(os.path.join(GOLDEN, 'loss.npz'))
    t, conv, gl = _loss_inputs(z)
    preds = ['v3d_left', 'v3d_right', 'v2d_left', 'v2d_right', 'c3d_left', 'c3d_right', 'c2d_left', 'c2d_right']
    for k in preds:
        t[k].requires_grad_(True)
    fused = FusedMeshLoss(gl['left'], gl['right'], conv['left'], conv['right'])
    result = {'verts3d': {s: t['v3d_' + s] for s in ('left', 'right')}, 'verts2d': {s: t['v2d_' + s] for s in ('left', 'right')}}
    hd = [{'verts3d': {s: t['c3d_' + s] for s in ('left', 'right')}, 'verts2d': {s: t['c2d_' + s] for s in ('left', 'right')}}]
    import renderih_amd.ops as ops_mod
    orig = ops_mod._stream
    ops_mod._stream = lambda: 0
    try:
        total, mano = calc_loss_GCN_fused(fused, epoch, result, None, hd, None, t['v2d_gt_left'], t['v2d_gt_right'],
                                          t['v3d_gt_left'], t['v3d_gt_right'], t['root_rel'])
        total.backward()
    finally:
        ops_mod._stream = orig
    key = 'e%d/' % epoch
    assert abs(total.item() - float(z[key + 'total'])) <= 1e-5 * abs(float(z[key + 'total']))
    for k in ('vert2d_loss', 'vert3d_loss', 'joint_loss', 'norm_loss', 'edge_loss'):
        assert abs(mano[k].item() - float(z[key + k])) <= 1e-5 * abs(float(z[key + k])) + 1e-12, k
    for k in preds:
        assert_close(t[k].grad, torch.from_numpy(z[key + 'grad_' + k]), 1e-4, 1e-6, 'grad ' + k)
