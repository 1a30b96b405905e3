"""Pin the oracle (oracle/*.py, a CPU restatement) to vectors produced by the REAL reference
(tests/golden/make_golden.py).  CPU only."""
import os
import numpy as np
import pytest
import torch

from oracle import net_oracle, mano_oracle
from renderih_amd import assets, testing

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')


def _check(store, name, t, rtol=1e-4, atol_frac=1e-5):
    if name in store:
        testing.assert_close(t, torch.from_numpy(store[name]), rtol, atol_frac, name)
    else:
        st, sa = testing.signature(t, nsamp=len(store[name + '#samp']))
        want_st, want_sa = store[name + '#stats'], store[name + '#samp']
        assert st[4] == want_st[4], name
        testing.assert_close(torch.from_numpy(sa), torch.from_numpy(want_sa), rtol, atol_frac * 10, name + '#samp')
        np.testing.assert_allclose(st[1:4], want_st[1:4], rtol=1e-4, err_msg=name + '#stats')


def _oracle_state(encoder='resnet50'):
    """Reference-keyed state dict without importing the reference: keys/shapes from our module tree."""
    from renderih_amd.model import build_model
    m = build_model(dropout=0.0, encoder_type=encoder)
    return testing.deterministic_state(m.state_dict(), seed=0)


@pytest.mark.parametrize('encoder', ['resnet50', 'hrnet32'])
@pytest.mark.parametrize('mode', ['eval', 'train'])
def test_network_oracle_matches_reference(mode, encoder):
    z = np.load(os.path.join(GOLDEN, 'net_%s%s.npz' % ('hrnet_' if encoder == 'hrnet32' else '', mode)))
    sd = _oracle_state(encoder)
    graph = net_oracle.graph_from_dicts(assets.load_graph_dict('left'), assets.load_graph_dict('right'))
    img = testing.seeded_image(2, seed=0)
    taps = {}
    if mode == 'train':
        for k, v in sd.items():
            if v.is_floating_point() and 'running' not in k and 'dense_coor' not in k:
                v.requires_grad_(True)
    with torch.set_grad_enabled(mode == 'train'):
        out = net_oracle.handnet_forward(sd, graph, img, training=(mode == 'train'), taps=taps)
    for k, v in testing.flatten_outputs(out).items():
        _check(z, 'out/' + k, v)
    for k, v in taps.items():
        if ('tap/' + k) in z or ('tap/' + k + '#samp') in z:
            _check(z, 'tap/' + k, v)
    if mode == 'train':
        loss = net_oracle.scalar_loss(out)
        assert abs(loss.item() - float(z['loss'])) <= 1e-5 * abs(float(z['loss']))
        loss.backward()
        names = list(z['grad_names'])
        got = {k for k, v in sd.items() if v.grad is not None}
        assert set(names) == got, (set(names) ^ got)
        for k in names:
            if testing.is_null_gradient(k):
                continue    # exactly zero in exact arithmetic: pure round-off noise (testing.is_null_gradient)
            st, sa = testing.signature(sd[k].grad, nsamp=32)
            testing.assert_close(torch.from_numpy(sa), torch.from_numpy(z['grad/' + k + '#samp']),
                                 1e-3, 1e-4, 'grad/' + k)


@pytest.mark.parametrize('mode', ['eval', 'train'])
def test_family_b_oracle_matches_reference(mode):
    """Second model family (common/myhand/lijun_model_graph.HandNET_GCN): the oracle's restatement
    against outputs, taps, loss and parameter gradients of the real reference modules (make_golden.py lijun)."""
    from renderih_amd.lijun import build_graph_model
    z = np.load(os.path.join(GOLDEN, 'net_lijun_%s.npz' % mode))
    sd = testing.deterministic_state(build_graph_model(0.0).state_dict(), seed=4)
    assert net_oracle.is_family_b(sd)
    graph = net_oracle.graph_from_dicts(assets.load_graph_dict('left'), assets.load_graph_dict('right'))
    img = testing.seeded_image(2, seed=5)
    taps = {}
    if mode == 'train':
        for k, v in sd.items():
            if v.is_floating_point() and 'running' not in k and 'dense_coor' not in k:
                v.requires_grad_(True)
    with torch.set_grad_enabled(mode == 'train'):
        out = net_oracle.handnet_forward(sd, graph, img, training=(mode == 'train'), taps=taps)
    assert out[3]['verts3d_MANO_list'] == {'left': [], 'right': []}
    flat = testing.flatten_outputs(out)
    assert {('out/' + k) for k in flat} == {k.split('#')[0] for k in z.files if k.startswith('out/')}
    for k, v in flat.items():
        _check(z, 'out/' + k, v)
    for k, v in taps.items():
        if ('tap/' + k) in z or ('tap/' + k + '#samp') in z:
            _check(z, 'tap/' + k, v)
    if mode == 'train':
        loss = net_oracle.scalar_loss(out)
        assert abs(loss.item() - float(z['loss'])) <= 1e-5 * abs(float(z['loss']))
        loss.backward()
        names = [str(k) for k in z['grad_names']]
        got = {k for k, v in sd.items() if v.grad is not None}
        assert set(names) == got, (set(names) ^ got)
        # Train-mode BatchNorm over B=2 makes this fixture ill-conditioned: the reference's own fp32 gradients are up to
        # 6e-2 (median 1e-2) away from an fp64 evaluation, so two correct fp32 evaluation orders differ by ~1e-3.  The
        # bar is therefore anchored on fp64: the reference's sample must be as close to the oracle's fp64 gradient as
        # the oracle's fp32 gradient is (x2 + a small floor), all relative to the largest sampled entry.
        _, g64 = net_oracle.run({k: v.detach() for k, v in sd.items()}, graph, img, True, torch.float64, True)
        worst = 0.0
        for k in names:
            if testing.is_null_gradient(k):
                continue
            gold = torch.from_numpy(z['grad/' + k + '#samp']).double()
            s32 = torch.from_numpy(testing.signature(sd[k].grad, nsamp=32)[1]).double()
            s64 = torch.from_numpy(testing.signature(g64[k], nsamp=32)[1]).double()
            scale = float(s64.abs().max().clamp_min(1e-30))
            e_gold, e_32 = float((gold - s64).abs().max()) / scale, float((s32 - s64).abs().max()) / scale
            assert e_gold <= 2.0 * e_32 + 1e-4, (k, e_gold, e_32)
            worst = max(worst, float((gold - s32).abs().max()) / scale)
        assert worst < 2e-2, worst      # and the two fp32 evaluations stay close to each other in absolute terms


def _mano_cases(z, side):
    names = sorted({k.split('/')[2] for k in z.files if k.startswith('mano/%s/' % side)})
    return names


@pytest.mark.parametrize('side', ['right', 'left'])
def test_mano_oracle_matches_reference(side):
    z = np.load(os.path.join(GOLDEN, 'mano.npz'))
    c = mano_oracle.constants_from_dict(assets.synthetic_mano_dict(side, seed=0))
    for name in _mano_cases(z, side):
        key = 'mano/%s/%s/' % (side, name)
        g = lambda n: torch.from_numpy(z[key + n]).requires_grad_(True) if (key + n) in z.files else None
        root, pose, shape, trans, scale = g('root'), g('pose'), g('shape'), g('trans'), g('scale')
        center = int(z[key + 'meta_center'])
        v, j = mano_oracle.mano_forward(c, root, pose, shape, trans, scale,
                                        center_idx=None if center < 0 else center,
                                        use_pca=int(z[key + 'meta_ncomp']) > 0,
                                        new_skel=bool(z[key + 'meta_new_skel']))
        testing.assert_close(v, torch.from_numpy(z[key + 'v']), 1e-5, 1e-6, key + 'v')
        testing.assert_close(j, torch.from_numpy(z[key + 'j']), 1e-5, 1e-6, key + 'j')
        ((v * torch.from_numpy(z[key + 'wv'])).sum() + (j * torch.from_numpy(z[key + 'wj'])).sum()).backward()
        for nm, t in (('root', root), ('pose', pose), ('shape', shape), ('trans', trans), ('scale', scale)):
            if t is not None:
                testing.assert_close(t.grad, torch.from_numpy(z[key + 'grad_' + nm]), 1e-4, 1e-5, key + 'grad_' + nm)


def _loss_inputs(z, device='cpu'):
    from renderih_amd.decoder import GCN_vert_convert
    from renderih_amd.loss import GraphLoss
    t = {k[3:]: torch.from_numpy(z[k]).to(device) for k in z.files if k.startswith('in/')}
    conv, gl = {}, {}
    for s in ('left', 'right'):
        gd = assets.load_graph_dict(s)
        md = assets.synthetic_mano_dict(s)
        conv[s] = GCN_vert_convert(vertex_num=778, graph_perm_reverse=gd['graph_perm_reverse'], graph_perm=gd['graph_perm'])
        J = torch.from_numpy(np.asarray(md['J_regressor'].todense(), np.float32))
        gl[s] = GraphLoss(J, np.asarray(md['f']), level=4, device=device)
    return t, conv, gl


@pytest.mark.parametrize('epoch', [0, 60])
def test_loss_mirror_matches_reference(epoch):
    """renderih_amd/loss.py (torch mirror of core/Loss.py, and the checker of the fused HIP loss) against values and
    gradients the reference's own GraphLoss / calc_loss_GCN produced (tests/golden/make_golden.py loss)."""
    from renderih_amd.loss import calc_loss_GCN
    z = np.load(os.path.join(GOLDEN, 'loss.npz'))
    t, conv, gl = _loss_inputs(z)
    preds = ['v3d_left', 'v3d_right', 'v2d_left', 'v2d_right', 'c3d_left', 'c3d_right', 'c2d_left', 'c2d_right']
    for k in preds:
        t[k].requires_grad_(True)
    result = {'verts3d': {s: t['v3d_' + s] for s in ('left', 'right')}, 'verts2d': {s: t['v2d_' + s] for s in ('left', 'right')}}
    hd = [{'verts3d': {s: t['c3d_' + s] for s in ('left', 'right')}, 'verts2d': {s: t['c2d_' + s] for s in ('left', 'right')}}]
    total, mano = calc_loss_GCN(None, epoch, gl['left'], gl['right'], conv['left'], conv['right'], result, None, hd, None,
                                t['v2d_gt_left'], t['v2d_gt_right'], t['v3d_gt_left'], t['v3d_gt_right'], t['root_rel'], 256)
    total.backward()
    key = 'e%d/' % epoch
    assert abs(total.item() - float(z[key + 'total'])) <= 1e-5 * abs(float(z[key + 'total']))
    for k in ('vert2d_loss', 'vert3d_loss', 'joint_loss', 'norm_loss', 'edge_loss'):
        assert abs(mano[k].item() - float(z[key + k])) <= 1e-5 * abs(float(z[key + k])) + 1e-12, k
    for k in preds:
        testing.assert_close(t[k].grad, torch.from_numpy(z[key + 'grad_' + k]), 1e-4, 1e-6, 'grad ' + k)
