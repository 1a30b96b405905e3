"""MANO parameter head / `load_new_model` (second model family with the MANO layer inside the forward):
  * rih_pose_math.h (the code the kernels run) compiled for the host: values and vector-Jacobian products against torch
    autograd through the oracle's restatement of the reference functions;
  * the oracle against vectors of the REAL reference model (tests/golden/net_newlijun_*.npz, make_golden.py newmodel);
  * module tree / backward wiring under the emulated ABI; the HIP kernels and the whole model on the GPU."""
import ctypes
import json
import os
import subprocess
import numpy as np
import pytest
import torch

from oracle import mano_oracle, net_oracle, pose_oracle as po
from renderih_amd import assets, testing
from renderih_amd.testing import assert_close

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def mano_consts():
    """MANO constants as decoder_lijun_mano.__init__ leaves them (shapedirs sign fix, :167-169)."""
    c = {s: mano_oracle.constants_from_dict(assets.synthetic_mano_dict(s)) for s in ('left', 'right')}
    if float((c['left']['shapedirs'][:, 0, :] - c['right']['shapedirs'][:, 0, :]).abs().sum()) < 1:
        c['left']['shapedirs'][:, 0, :] *= -1
    return c


def test_pose_math_header_matches_autograd(tmp_path):
    src = tmp_path / 'w.cpp'
    src.write_text('''#include "rih_pose_math.h"
extern "C" void rot6d(int N, const float* x, float* R, float* aa) { for (int n = 0; n < N; ++n) rih_rot6d_to_rotmat_aa<float>(x + 6 * n, R + 9 * n, aa + 3 * n); }
extern "C" void rot6d_vjp(int N, const float* x, const float* dR, const float* daa, float* dx) { for (int n = 0; n < N; ++n) rih_rot6d_vjp(x + 6 * n, dR + 9 * n, daa + 3 * n, dx + 6 * n); }
extern "C" void rodr(int N, const float* a, float* R) { for (int n = 0; n < N; ++n) rih_rodrigues<float>(a + 3 * n, R + 9 * n); }
extern "C" void rodr_vjp(int N, const float* a, const float* dR, float* da) { for (int n = 0; n < N; ++n) rih_rodrigues_vjp(a + 3 * n, dR + 9 * n, da + 3 * n); }
extern "C" void hsw(int N, const float* x, float* y, float* g) { for (int n = 0; n < N; ++n) { y[n] = rih_hardswish(x[n]); g[n] = rih_hardswish_grad(x[n]); } }
''')
    lib = tmp_path / 'libw.so'
    subprocess.check_call(['g++', '-O2', '-shared', '-fPIC', '-I', os.path.join(ROOT, 'renderih_amd', 'csrc'), '-o', str(lib),
                           str(src)])
    L = ctypes.CDLL(str(lib))
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    g = torch.Generator().manual_seed(0)
    N = 4000
    x = torch.randn(N, 6, generator=g)
    xr = x.clone().requires_grad_(True)
    R = po.rot6d_to_rotmat(xr)
    aa = po.rotation_matrix_to_angle_axis(R)
    wR, wa = torch.randn(N, 3, 3, generator=g), torch.randn(N, 3, generator=g)
    ((R * wR).sum() + (aa * wa).sum()).backward()
    m = R.detach().transpose(1, 2)
    d2 = m[:, 2, 2] < 1e-6
    for mask in (d2 & (m[:, 0, 0] > m[:, 1, 1]), d2 & ~(m[:, 0, 0] > m[:, 1, 1]), ~d2 & (m[:, 0, 0] < -m[:, 1, 1]),
                 ~d2 & ~(m[:, 0, 0] < -m[:, 1, 1])):
        assert int(mask.sum()) > 50                      # every quaternion branch of comm.py:311-314 is exercised
    Rn, an, dx = np.zeros((N, 9), np.float32), np.zeros((N, 3), np.float32), np.zeros((N, 6), np.float32)
    L.rot6d(N, P(x.numpy().copy()), P(Rn), P(an))
    L.rot6d_vjp(N, P(x.numpy().copy()), P(wR.numpy().reshape(N, 9).copy()), P(wa.numpy().copy()), P(dx))
    assert np.abs(Rn.reshape(N, 3, 3) - R.detach().numpy()).max() < 1e-5
    assert np.abs(an - aa.detach().numpy()).max() < 1e-5
    gr = xr.grad.numpy()
    assert (np.abs(dx - gr) / (np.abs(gr).max(1, keepdims=True) + 1e-3)).max() < 1e-4
    a = torch.randn(N, 3, generator=g) * 1.5
    a[:5] = 0                                             # zero axis: |axis| has sub-gradient 0, no NaN
    ar = a.clone().requires_grad_(True)
    w = torch.randn(N, 3, 3, generator=g)
    Rr = po.rodrigues_batch(ar)
    (Rr * w).sum().backward()
    R2, da = np.zeros((N, 9), np.float32), np.zeros((N, 3), np.float32)
    L.rodr(N, P(a.numpy().copy()), P(R2))
    L.rodr_vjp(N, P(a.numpy().copy()), P(w.numpy().reshape(N, 9).copy()), P(da))
    assert np.abs(R2.reshape(N, 3, 3) - Rr.detach().numpy()).max() < 1e-5
    assert np.abs(da - ar.grad.numpy()).max() < 1e-4
    xs = torch.linspace(-5, 5, 1001, requires_grad=True)
    ys = torch.nn.functional.hardswish(xs)
    ys.sum().backward()
    yy, gg = np.zeros(1001, np.float32), np.zeros(1001, np.float32)
    L.hsw(1001, P(xs.detach().numpy().copy()), P(yy), P(gg))
    assert np.abs(yy - ys.detach().numpy()).max() < 1e-6 and np.abs(gg - xs.grad.numpy()).max() < 1e-6


def _oracle_state():
    from renderih_amd.lijun import build_new_model
    m = build_new_model(0.0)
    return m, testing.deterministic_state(m.state_dict(), seed=11)


_RUNS = {}


def _oracle_runs(training):
    """fp32 and fp64 oracle evaluations (outputs + gradients) of the seeded configuration, shared by the tests below."""
    if training not in _RUNS:
        _, sd = _oracle_state()
        graph = net_oracle.graph_from_dicts(assets.load_graph_dict('left'), assets.load_graph_dict('right'))
        img = testing.seeded_image(2, seed=12)
        _RUNS[training] = (net_oracle.run(sd, graph, img, training, torch.float32, True, mano=mano_consts()),
                           net_oracle.run(sd, graph, img, training, torch.float64, True, mano=mano_consts()))
    return _RUNS[training]


def test_new_model_schema_equals_reference():
    m, _ = _oracle_state()
    ref = json.load(open(os.path.join(GOLDEN, 'state_keys_newlijun.json')))
    sd = m.state_dict()
    assert list(sd.keys()) == list(ref.keys())
    assert {k: list(v.shape) for k, v in sd.items()} == ref


@pytest.mark.parametrize('mode', ['eval', 'train'])
def test_new_model_oracle_matches_reference(mode):
    z = np.load(os.path.join(GOLDEN, 'net_newlijun_%s.npz' % mode))
    training = mode == 'train'
    if training:
        (w32, g32), (_, g64) = _oracle_runs(True)
    else:
        _, sd = _oracle_state()
        graph = net_oracle.graph_from_dicts(assets.load_graph_dict('left'), assets.load_graph_dict('right'))
        w32, _ = net_oracle.run(sd, graph, testing.seeded_image(2, seed=12), False, torch.float32, False, mano=mano_consts())
    assert {('out/' + k) for k in w32} == {k.split('#')[0] for k in z.files if k.startswith('out/')}
    from test_oracle_golden import _check
    tol = (2e-3, 5e-4) if training else (1e-4, 1e-5)          # train-mode BN at B=2: see test_oracle_golden.py
    for k, v in w32.items():
        _check(z, 'out/' + k, v, *tol)
    if training:
        names = [str(k) for k in z['grad_names']]
        assert set(names) == set(g32), sorted(set(names) ^ set(g32))[:8]
        for k in names:
            if testing.is_null_gradient(k):
                continue
            gold = torch.from_numpy(z['grad/' + k + '#samp']).double()
            s32 = torch.from_numpy(testing.signature(g32[k], nsamp=32)[1]).double()
            s64 = torch.from_numpy(testing.signature(g64[k], nsamp=32)[1]).double()
            scale = float(s64.abs().max().clamp_min(1e-30))
            e_gold, e_32 = float((gold - s64).abs().max()) / scale, float((s32 - s64).abs().max()) / scale
            assert e_gold <= 2.0 * e_32 + 1e-4, (k, e_gold, e_32)


def _model_vs_oracle(device, training):
    from test_gpu_model import _grad_report
    m, sd = _oracle_state()
    m.load_state_dict(sd)
    m = m.to(device)
    m.train(training)
    img = testing.seeded_image(2, 12)
    (w32, g32), (w64, g64) = _oracle_runs(training)
    out = m(img.to(device))
    got = testing.flatten_outputs(out)
    assert set(got) == set(w64)
    for k in w64:
        testing.assert_fp32_equivalent(got[k], w32[k], w64[k], k=4.0, floor=2e-5, what=k)
    net_oracle.scalar_loss(out).backward()
    params = [(k, p.grad) for k, p in m.named_parameters() if p.grad is not None]
    assert {k for k, _ in params} == set(g64.keys())
    _grad_report(params, g32, g64)
    return m, got


def test_new_model_host_logic_matches_oracle():
    from abi_emulator import emulated_abi
    with emulated_abi():
        _model_vs_oracle(torch.device('cpu'), True)


def _pose_head_kernels_vs_oracle(d):
    from renderih_amd import pose_head
    g = torch.Generator().manual_seed(2)
    x = torch.randn(3000, 6, generator=g)
    xr = x.clone().requires_grad_(True)
    R = po.rot6d_to_rotmat(xr)
    aa = po.rotation_matrix_to_angle_axis(R)
    wR, wa = torch.randn(3000, 3, 3, generator=g), torch.randn(3000, 3, generator=g)
    ((R * wR).sum() + (aa * wa).sum()).backward()
    xg = x.detach().clone().to(d).requires_grad_(True)
    Rg, ag = pose_head.rot6d_to_rotmat_aa(xg)
    assert_close(Rg, R, 1e-4, 1e-5, 'rot6d R')
    assert_close(ag, aa, 1e-4, 1e-5, 'rot6d aa')
    ((Rg * wR.to(d)).sum() + (ag * wa.to(d)).sum()).backward()
    assert_close(xg.grad, xr.grad, 1e-3, 1e-4, 'rot6d dx')
    a = torch.randn(500, 3, generator=g) * 1.5
    a[:3] = 0
    ar = a.clone().requires_grad_(True)
    w = torch.randn(500, 3, 3, generator=g)
    (po.rodrigues_batch(ar) * w).sum().backward()
    agp = a.detach().clone().to(d).requires_grad_(True)
    Rr = pose_head.rodrigues(agp)
    assert_close(Rr, po.rodrigues_batch(a), 1e-4, 1e-5, 'rodrigues')
    (Rr * w.to(d)).sum().backward()
    assert_close(agp.grad, ar.grad, 1e-3, 1e-4, 'rodrigues da')
    t = torch.randn(7, 1024, generator=g) * 3
    for name, fn, ref in (('hardswish', pose_head.hardswish, torch.nn.functional.hardswish),
                          ('tanh3', lambda u: pose_head.tanh_scale(u, 3.0), lambda u: torch.tanh(u) * 3)):
        tr, tg = t.detach().clone().requires_grad_(True), t.detach().clone().to(d).requires_grad_(True)
        yr, yg = ref(tr), fn(tg)
        assert_close(yg, yr, 1e-4, 1e-5, name)
        gy = torch.randn(7, 1024, generator=g)
        yr.backward(gy)
        yg.backward(gy.to(d))
        assert_close(tg.grad, tr.grad, 1e-4, 1e-5, name + ' grad')
    v, j = torch.randn(5, 778, 3, generator=g) * 0.1, torch.randn(5, 21, 3, generator=g) * 0.1
    vr, jr = v.clone().requires_grad_(True), j.clone().requires_grad_(True)
    s = 0.095 / torch.linalg.norm(jr[:, 9] - jr[:, 0], dim=-1)
    outr = (vr - jr[:, 0:1]) * s.view(-1, 1, 1)
    gv, gs = torch.randn(5, 778, 3, generator=g), torch.randn(5, generator=g)
    ((outr * gv).sum() + (s * gs).sum()).backward()
    vg, jg = v.detach().clone().to(d).requires_grad_(True), j.detach().clone().to(d).requires_grad_(True)
    outg, sg = pose_head.center_scale(vg, jg)
    assert_close(outg, outr, 1e-4, 1e-5, 'center_scale')
    assert_close(sg, s, 1e-4, 1e-5, 'center_scale s')
    ((outg * gv.to(d)).sum() + (sg * gs.to(d)).sum()).backward()
    assert_close(vg.grad, vr.grad, 1e-3, 1e-4, 'center_scale dv')
    assert_close(jg.grad, jr.grad, 1e-3, 1e-4, 'center_scale dj')


@pytest.mark.gpu
def test_pose_head_kernels_match_oracle():
    """rih_pose.hip on the GPU against torch autograd through the oracle functions."""
    _pose_head_kernels_vs_oracle(torch.device('cuda:0'))


@pytest.mark.gpu
def test_new_model_matches_fp64_oracle_on_gpu():
    """Train mode: forward outputs and every parameter gradient, anchored on the oracle's fp64 run."""
    _model_vs_oracle(torch.device('cuda:0'), True)


@pytest.mark.gpu
def test_new_model_eval_matches_reference_golden_on_gpu():
    """Eval mode (running-statistics BN): strict 1e-4 parity with the vectors of the real reference model."""
    from test_gpu_model import _check_golden
    z = np.load(os.path.join(GOLDEN, 'net_newlijun_eval.npz'))
    m, sd = _oracle_state()
    m.load_state_dict(sd)
    m = m.to('cuda:0').eval()
    with torch.no_grad():
        out = m(testing.seeded_image(2, 12).cuda())
    flat = testing.flatten_outputs(out)
    assert {('out/' + k) for k in flat} == {k.split('#')[0] for k in z.files if k.startswith('out/')}
    for k, v in flat.items():
        _check_golden(z, 'out/' + k, v)
