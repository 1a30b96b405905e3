"""Evaluation metrics: oracle pinned by reference-generated vectors; the closed-form Procrustes core
of the HIP kernel checked on the host against numpy's SVD; host logic under the emulated ABI; the kernel on the GPU."""
import ctypes
import os
import subprocess
import numpy as np
import pytest
import torch

from oracle import metrics_oracle
from renderih_amd.testing import assert_close

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
GOLDEN = os.path.join(ROOT, 'tests', 'golden', 'metrics.npz')


def _golden(side):
    z = np.load(GOLDEN)
    t = lambda k: torch.from_numpy(z[k])
    return z, t('in/v_pred_' + side), t('in/v_gt_' + side), t('in/Jreg_' + side), t('in/J16_' + side)


@pytest.mark.parametrize('side', ['left', 'right'])
def test_metrics_oracle_matches_reference(side):
    """oracle/metrics_oracle.py against Jr / eval_hand2 / batch_compute_similarity_transform_torch of the reference."""
    from renderih_amd.metrics import joint_regressor_21
    z, vp, vg, Jreg, J16 = _golden(side)
    assert torch.equal(joint_regressor_21(J16), Jreg)
    m = metrics_oracle.hand_metrics(vp, vg, Jreg)
    assert_close(m['j_pred'], torch.from_numpy(z['out/j_pred_' + side]), 1e-5, 1e-6, 'j_pred')
    for k in ('j_err', 'v_err', 'j_err_ori', 'v_err_ori', 'pa_mpjpe', 'pa_mpvpe'):
        assert_close(m[k], torch.from_numpy(z['out/%s_%s' % (k, side)]), 1e-4, 1e-5, k)


def test_procrustes_core_matches_svd(tmp_path):
    """renderih_amd/csrc/rih_procrustes.h (Horn quaternion + Jacobi, the code the kernel runs) compiled for the host
    against the SVD formulation of the reference, including improper (reflected) configurations and tiny point sets."""
    src = tmp_path / 'w.cpp'
    src.write_text('''#include "rih_procrustes.h"
extern "C" void sim(int N, const double* x1, const double* x2, double* out) {
    double s1[3] = {0, 0, 0}, s2[3] = {0, 0, 0}, s12[3][3] = {{0}}, q1 = 0;
    for (int n = 0; n < N; ++n)
        for (int a = 0; a < 3; ++a) {
            s1[a] += x1[3 * n + a]; s2[a] += x2[3 * n + a]; q1 += x1[3 * n + a] * x1[3 * n + a];
            for (int b = 0; b < 3; ++b) s12[a][b] += x1[3 * n + a] * x2[3 * n + b];
        }
    double R[3][3], sc, t[3];
    rih_similarity_from_moments(N, s1, s2, s12, q1, R, &sc, t);
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) out[3 * a + b] = R[a][b];
    out[9] = sc; out[10] = t[0]; out[11] = t[1]; out[12] = t[2];
}
''')
    lib = tmp_path / 'libw.so'
    subprocess.check_call(['g++', '-O2', '-shared', '-fPIC', '-I', os.path.join(ROOT, 'renderih_amd', 'csrc'), '-o', str(lib),
                           str(src)])
    fn = ctypes.CDLL(str(lib)).sim
    rs = np.random.RandomState(0)
    for trial in range(120):
        N = [21, 778, 5, 4][trial % 4]
        x1 = np.ascontiguousarray(rs.randn(N, 3) * 0.1)
        A = np.linalg.qr(rs.randn(3, 3))[0]
        if trial % 3 == 0:
            A[:, 0] *= -1                       # improper map: the SVD route needs its reflection fix here
        x2 = np.ascontiguousarray((x1 @ A.T) * (0.5 + rs.rand()) + rs.randn(3) * 0.2 +
                                  rs.randn(N, 3) * (0.001 if trial % 5 else 0.05))
        out = np.zeros(13)
        fn(N, x1.ctypes.data_as(ctypes.c_void_p), x2.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p))
        hat = out[9] * (x1 @ out[:9].reshape(3, 3).T) + out[10:13]
        want = metrics_oracle.similarity_transform(torch.from_numpy(x1)[None], torch.from_numpy(x2)[None])[0].numpy()
        assert np.abs(hat - want).max() <= 1e-9 * max(1.0, np.abs(want).max()), trial


def _check_against_oracle(device):
    from renderih_amd.metrics import hand_metrics, eval_hand2
    for side in ('left', 'right'):
        z, vp, vg, Jreg, _ = _golden(side)
        want = metrics_oracle.hand_metrics(vp.double(), vg.double(), Jreg.double())
        got = hand_metrics(vp.to(device), vg.to(device), Jreg.to(device))
        for k in ('j_pred', 'j_err_ori', 'v_err_ori', 'j_err', 'v_err', 'pa_mpjpe', 'pa_mpvpe'):
            assert_close(got[k], want[k], 1e-4, 1e-5, '%s %s' % (k, side))
            if ('out/%s_%s' % (k, side)) in z.files:
                assert_close(got[k], torch.from_numpy(z['out/%s_%s' % (k, side)]), 1e-4, 1e-5, 'golden %s %s' % (k, side))
        # given joints + another root / bone convention (eval_hand of intag_eval.py:145-162: root 9, bone 9-0)
        jp, jg = want['j_pred'].float(), torch.matmul(Jreg, vg)
        want2 = metrics_oracle.hand_metrics(vp.double(), vg.double(), None, jp.double(), jg.double(), root_idx=9, bone=(9, 0))
        got2 = hand_metrics(vp.to(device), vg.to(device), None, jp.to(device), jg.to(device), root_idx=9, bone=(9, 0))
        for k in ('j_err', 'v_err', 'pa_mpjpe', 'pa_mpvpe'):
            assert_close(got2[k], want2[k], 1e-4, 1e-5, 'root9 %s %s' % (k, side))
    # the reference's list-appending entry point
    zl = [_golden(s) for s in ('left', 'right')]
    lists = [{'left': [], 'right': []} for _ in range(4)]
    d = lambda t: t.to(device)
    jl = {s: (torch.from_numpy(z['out/j_gt_' + s]), torch.from_numpy(z['out/j_pred_' + s])) for s, (z, *_r) in zip(('left', 'right'), zl)}
    eval_hand2(d(zl[0][2]), d(zl[1][2]), d(jl['left'][0]), d(jl['right'][0]), d(zl[0][1]), d(zl[1][1]), d(jl['left'][1]),
               d(jl['right'][1]), *lists)
    for s, (z, *_r) in zip(('left', 'right'), zl):
        assert_close(torch.from_numpy(lists[0][s][0]), torch.from_numpy(z['out/j_err_' + s]), 1e-4, 1e-5, 'eval_hand2 j ' + s)
        assert_close(torch.from_numpy(lists[1][s][0]), torch.from_numpy(z['out/v_err_' + s]), 1e-4, 1e-5, 'eval_hand2 v ' + s)
        assert lists[2][s][0] is lists[0][s][0] and lists[3][s][0] is lists[1][s][0]


def test_metrics_host_logic_emulated():
    from abi_emulator import emulated_abi
    with emulated_abi():
        _check_against_oracle(torch.device('cpu'))


@pytest.mark.gpu
def test_metrics_kernel_matches_oracle_and_reference_golden():
    assert torch.cuda.is_available()
    _check_against_oracle(torch.device('cuda:0'))
    # B = 64 images, both point-set sizes, against the oracle in double precision
    from renderih_amd.metrics import hand_metrics
    g = torch.Generator().manual_seed(3)
    _, _, _, Jreg, _ = _golden('right')
    vg = 0.1 * torch.randn(64, 778, 3, generator=g)
    A = torch.linalg.qr(torch.randn(64, 3, 3, generator=g))[0]
    vp = (vg @ A.transpose(1, 2)) * 1.3 + 0.02 * torch.randn(64, 778, 3, generator=g) + 0.1
    want = metrics_oracle.hand_metrics(vp.double(), vg.double(), Jreg.double())
    got = hand_metrics(vp.cuda(), vg.cuda(), Jreg.cuda())
    for k in ('j_err', 'v_err', 'pa_mpjpe', 'pa_mpvpe'):
        assert_close(got[k], want[k], 1e-4, 1e-5, 'B64 ' + k)


def test_evaluate_loop_matches_reference_lines():
    """renderih_amd.evaluate.evaluate against the arithmetic of apps/eval_interhand.py:298-420 restated in torch (root joint 0,
    bone (1, 0), torch.svd Procrustes), on a stand-in network with fixed predictions; host logic through the ABI emulator."""
    import sys
    import os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from abi_emulator import emulated_abi
    from renderih_amd import assets
    from renderih_amd.evaluate import evaluate
    from renderih_amd.metrics import joint_regressor_21
    from oracle import metrics_oracle
    g = torch.Generator().manual_seed(4)
    jr = {s: joint_regressor_21(torch.from_numpy(np.asarray(assets.synthetic_mano_dict(s)['J_regressor'].todense()
                                                               if hasattr(assets.synthetic_mano_dict(s)['J_regressor'], 'todense')
                                                               else assets.synthetic_mano_dict(s)['J_regressor'])).float())
          for s in ('left', 'right')}
    batches, preds = [], []
    for b in range(2):
        vg = {s: torch.randn(3, 778, 3, generator=g) * 0.05 + (0.1 if s == 'left' else -0.1) for s in ('left', 'right')}
        vp = {s: vg[s] * 1.1 + 0.01 * torch.randn(3, 778, 3, generator=g) for s in vg}
        batches.append((torch.zeros(3, 3, 8, 8), None, vg['left'], None, vg['right']))
        preds.append(vp)
    it = iter(preds)
    network = lambda img: ({'verts3d': next(it)}, None, None, None)
    with emulated_abi():
        summary, per = evaluate(network, [(b[0], torch.zeros(1), b[2], torch.zeros(1), b[4]) for b in batches], jr['left'], jr['right'])
    # reference arithmetic
    want = {k: {'left': [], 'right': []} for k in ('ori', 'scaled', 'pa')}
    ptr, gtr = [], []
    for bt, vp in zip(batches, preds):
        roots = {}
        for s, vg in (('left', bt[2]), ('right', bt[4])):
            J = lambda v: torch.einsum('jv,bvc->bjc', jr[s], v)
            jg, jp = J(vg), J(vp[s])
            roots[s] = (jp[:, 0], jg[:, 0])
            lg = torch.linalg.norm(jg[:, 1] - jg[:, 0], dim=-1)
            lp = torch.linalg.norm(jp[:, 1] - jp[:, 0], dim=-1)
            jg0, jp0 = jg - jg[:, 0:1], jp - jp[:, 0:1]
            want['ori'][s].append(torch.linalg.norm(jp0 - jg0, dim=-1).numpy())
            want['scaled'][s].append(torch.linalg.norm(jp0 * (lg / lp)[:, None, None] - jg0, dim=-1).numpy())
            hat = metrics_oracle.similarity_transform(jp0, jg0)
            want['pa'][s].append(torch.sqrt(((hat - jg0) ** 2).sum(-1)).mean(-1).numpy())
        ptr.append((roots['left'][0] - roots['right'][0]).numpy())
        gtr.append((roots['left'][1] - roots['right'][1]).numpy())
    for s in ('left', 'right'):
        for key, name in (('ori', 'j_err_ori'), ('scaled', 'j_err'), ('pa', 'pa_mpjpe')):
            w = np.concatenate(want[key][s], 0)
            assert np.abs(per[name][s] - w).max() < 2e-6 + 1e-4 * np.abs(w).max(), (s, key)
    assert abs(summary['ori joint mpjpe']['all'] - 500 * (np.concatenate(want['ori']['left']).mean() +
                                                           np.concatenate(want['ori']['right']).mean())) < 1e-3
    mrrpe = np.sqrt(((np.concatenate(ptr) - np.concatenate(gtr)) ** 2).sum(1)).mean()
    assert abs(summary['mrrpe'] - mrrpe) < 1e-6


def _cdev_case():
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'cdev.npz'))
    return {k: torch.from_numpy(g[k]) for k in g.files}


def test_cdev_oracle_matches_reference_fixture():
    """oracle compute_cdev against the reference's own utils/eval_metrics.compute_cdev (tests/golden/make_cdev_golden.py);
    one sample has hands 0.5 m apart: NaN in both."""
    from oracle import metrics_oracle
    c = _cdev_case()
    got = metrics_oracle.compute_cdev(c['pred_left'].clone(), c['pred_right'].clone(), c['gt_left'], c['gt_right'])
    assert torch.isnan(c['cdev'][2]) and torch.equal(torch.isnan(got), torch.isnan(c['cdev']))
    ok = ~torch.isnan(got)
    assert torch.allclose(got[ok], c['cdev'][ok], rtol=1e-6, atol=0)


def cdev_kernel_vs_fixture(dev):
    from renderih_amd.metrics import compute_cdev
    c = _cdev_case()
    got = compute_cdev(*(c[k].to(dev) for k in ('pred_left', 'pred_right', 'gt_left', 'gt_right'))).cpu()
    assert torch.equal(torch.isnan(got), torch.isnan(c['cdev']))
    ok = ~torch.isnan(got)
    assert torch.allclose(got[ok], c['cdev'][ok], rtol=2e-6, atol=0)


def test_cdev_kernel_on_cpu():
    import sys
    import os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from hipcpu.host_kernels import host_kernels_abi
    with host_kernels_abi():
        cdev_kernel_vs_fixture(torch.device('cpu'))
