"""GPU parity of the whole pose network (HIP path) against (a) the reference-generated golden vectors and
(b) the CPU oracle on the same seeded inputs and weights -- forward (eval, train) and every parameter gradient."""
import os
import numpy as np
import pytest
import torch

from renderih_amd import assets, testing
from renderih_amd.testing import assert_close

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')


def _build(dropout=0.0, seed=0):
    from renderih_amd.model import build_model
    from renderih_amd import _lib
    _lib.load()
    m = build_model(dropout)
    sd = testing.deterministic_state(m.state_dict(), seed=seed)
    m.load_state_dict(sd)
    return m.to('cuda:0'), sd


def _check_golden(store, name, t, rtol=1e-4, atol_frac=1e-5):
    if name in store.files:
        assert_close(t, torch.from_numpy(store[name]), rtol, atol_frac, name)
    else:
        want = store[name + '#samp']
        st, sa = testing.signature(t, nsamp=len(want))
        assert st[4] == store[name + '#stats'][4], name
        assert_close(torch.from_numpy(sa), torch.from_numpy(want), rtol, atol_frac * 10, name + '#samp')


def test_native_library_is_loaded_and_has_no_fallback():
    from renderih_amd import _lib
    lib = _lib.load()
    assert os.path.exists(_lib.lib_path())
    maps = open('/proc/self/maps').read()
    assert 'librenderih_amd.so' in maps
    assert lib.rih_arch() == b'gfx950'


@pytest.mark.parametrize('mode', ['eval', 'train'])
def test_model_matches_reference_golden(mode):
    z = np.load(os.path.join(GOLDEN, 'net_%s.npz' % mode))
    m, _ = _build(0.0)
    m.train(mode == 'train')
    img = testing.seeded_image(2, 0).cuda()
    with torch.set_grad_enabled(mode == 'train'):
        out = m(img)
    errs = {}
    for k, v in testing.flatten_outputs(out).items():
        _check_golden(z, 'out/' + k, v)
    if mode == 'train':
        from oracle.net_oracle import scalar_loss
        loss = scalar_loss(out)
        assert abs(loss.item() - float(z['loss'])) <= 1e-4 * abs(float(z['loss']))
        loss.backward()
        names = [str(n) for n in z['grad_names']]
        params = dict(m.named_parameters())
        got = {k for k, p in params.items() if p.grad is not None}
        assert set(names) == got, sorted(set(names) ^ got)[:10]
        bad = []
        for k in names:
            if k.endswith('w_ks.bias'):
                continue        # zero in exact arithmetic (softmax shift invariance): round-off noise only
            want = z['grad/' + k + '#samp']
            st, sa = testing.signature(params[k].grad, nsamp=32)
            try:
                assert_close(torch.from_numpy(sa), torch.from_numpy(want), 2e-3, 2e-4, 'grad/' + k)
            except AssertionError as e:
                bad.append(str(e))
        assert not bad, '%d grads off:\n%s' % (len(bad), '\n'.join(bad[:20]))
        sd = m.state_dict()
        for k in z.files:
            if k.startswith('bnstat/'):
                assert_close(sd[k[7:]].float(), torch.from_numpy(np.asarray(z[k])).float(), 1e-4, 1e-5, k)


def test_model_matches_oracle_other_seed_and_batch():
    """Same comparison against the CPU oracle with different weights, batch 3, train mode (dropout 0)."""
    from oracle import net_oracle
    m, sd = _build(0.0, seed=5)
    m.train()
    img = testing.seeded_image(3, 11)
    graph = net_oracle.graph_from_dicts(assets.load_graph_dict('left'), assets.load_graph_dict('right'))
    sdo = {k: v.clone() for k, v in sd.items()}
    for k, v in sdo.items():
        if v.is_floating_point() and 'running' not in k and 'dense_coor' not in k:
            v.requires_grad_(True)
    want = net_oracle.handnet_forward(sdo, graph, img, training=True)
    got = m(img.cuda())
    fw, fg = testing.flatten_outputs(want), testing.flatten_outputs(got)
    for k in fw:
        assert_close(fg[k], fw[k], 1e-4, 1e-5, k)
    net_oracle.scalar_loss(want).backward()
    net_oracle.scalar_loss(got).backward()
    bad = []
    for k, p in m.named_parameters():
        if sdo[k].grad is None:
            assert p.grad is None, k
            continue
        if k.endswith('w_ks.bias'):
            continue
        try:
            assert_close(p.grad, sdo[k].grad, 2e-3, 2e-4, 'grad ' + k)
        except AssertionError as e:
            bad.append(str(e))
    assert not bad, '%d grads off:\n%s' % (len(bad), '\n'.join(bad[:20]))


def test_dropout_training_is_statistically_sane_and_deterministic_per_seed():
    m, _ = _build(0.05)
    m.train()
    img = testing.seeded_image(2, 0).cuda()
    torch.manual_seed(7)
    a = m(img)[0]['verts3d']['left']
    torch.manual_seed(7)
    b = m(img)[0]['verts3d']['left']
    torch.manual_seed(8)
    c = m(img)[0]['verts3d']['left']
    assert torch.equal(a, b)
    assert not torch.equal(a, c)
    m0, _ = _build(0.0)
    m0.train()
    ref = m0(img)[0]['verts3d']['left']
    rel = float((a - ref).abs().max() / ref.abs().max())
    assert 0 < rel < 0.5, rel
    testing.flatten_outputs(m(img))['result.verts3d.left'].abs().sum().backward()
    assert all(torch.isfinite(p.grad).all() for p in m.parameters() if p.grad is not None)


def test_full_size_batch_properties():
    """B=64 (BASELINE config 2): finite outputs, batch independence in eval mode (image i alone == image i in batch)."""
    m, _ = _build(0.0)
    m.eval()
    img = testing.seeded_image(64, 3).cuda()
    with torch.no_grad():
        big = testing.flatten_outputs(m(img))
        one = testing.flatten_outputs(m(img[5:6]))
    for k, v in big.items():
        assert torch.isfinite(v).all(), k
        assert_close(v[5:6], one[k], 1e-4, 1e-5, 'batch-independence ' + k)
