"""Parity AT THE SHAPES bench.py TIMES (BASELINE configs[1]: B = 64, 256x256, ResNet50 variant).

(a) Every distinct convolution problem of the network at B = 64 (SURVEY 8d "per-kernel problem list": trunk, aux decoders, mid
    convs, patch convs) through ops.conv2d -- forward, data gradient and weight gradient with the tiles / split-K values the
    planner picks at those sizes (M = 262144-row tiles, the 256x128 pipelined kernel, split-K 14..128 weight gradients,
    parity-class data gradients of the strided convs) -- against an fp64 restatement (im2col + matmul in double).
    Tolerance: outputs 1e-4 relative + 1e-5 of the tensor maximum (the fp32 bar of BASELINE.json); gradients 1e-3 / 1e-4,
    as in test_gpu_ops (the weight gradient reduces over up to 262144 pixels in fp32).
(b) Training-mode forward of the whole network at B = 16 (>= 1024 samples per channel in every BatchNorm) against the CPU
    oracle, anchored on the oracle's fp64 run: measured 2e-4 .. 9e-4 between the two fp32 implementations, so the plain 1e-4
    bar does not hold in training mode even at this batch size (see the test's docstring).
(c) One full training step at B = 64: every gradient finite, loss equal between two launches of the same step
    (determinism of the split-K reductions at bench size)."""
import math
import pytest
import torch
import torch.nn.functional as F

from renderih_amd import assets, testing
from renderih_amd.testing import assert_close

pytestmark = pytest.mark.gpu
B = 64

# (H, Cin, Cout, k, stride, pad, bias)  -- input spatial size H x H
CONV_PROBLEMS = [
    (256, 3, 64, 7, 2, 3, False),                                                            # stem
    (64, 64, 64, 1, 1, 0, False), (64, 64, 64, 3, 1, 1, False), (64, 64, 256, 1, 1, 0, False), (64, 256, 64, 1, 1, 0, False),
    (64, 256, 128, 1, 1, 0, False), (64, 128, 128, 3, 2, 1, False), (32, 128, 128, 3, 1, 1, False),
    (32, 128, 512, 1, 1, 0, False), (32, 512, 128, 1, 1, 0, False), (64, 256, 512, 1, 2, 0, False),
    (32, 512, 256, 1, 1, 0, False), (32, 256, 256, 3, 2, 1, False), (16, 256, 256, 3, 1, 1, False),
    (16, 256, 1024, 1, 1, 0, False), (16, 1024, 256, 1, 1, 0, False), (32, 512, 1024, 1, 2, 0, False),
    (16, 1024, 512, 1, 1, 0, False), (16, 512, 512, 3, 2, 1, False), (8, 512, 512, 3, 1, 1, False),
    (8, 512, 2048, 1, 1, 0, False), (8, 2048, 512, 1, 1, 0, False), (16, 1024, 2048, 1, 2, 0, False),
    (8, 2048, 128, 1, 1, 0, False), (16, 128, 128, 3, 1, 1, False), (32, 128, 128, 3, 1, 1, False),        # aux decoders
    (64, 128, 128, 3, 1, 1, False), (64, 128, 42, 1, 1, 0, True), (64, 128, 8, 1, 1, 0, True),
    (8, 256, 256, 1, 1, 0, False), (16, 1280, 256, 1, 1, 0, False), (32, 768, 256, 1, 1, 0, False),       # mid convs
    (64, 512, 256, 1, 1, 0, False),
    (16, 256, 128, 2, 2, 0, True), (32, 256, 64, 4, 4, 0, True),                                          # patch convs
]


def _ref_conv_fp64(x_nchw, w, b, stride, pad, gy):
    """y, dx, dw, db of conv2d in double via im2col + matmul (rocBLAS dgemm) on the GPU."""
    xd, wd, gd = x_nchw.double(), w.double(), gy.double()
    N, Cin, H, W = xd.shape
    Cout, _, k, _ = wd.shape
    cols = F.unfold(xd, k, padding=pad, stride=stride)                      # [N, Cin*k*k, L]
    wm = wd.reshape(Cout, -1)
    y = torch.matmul(wm, cols)                                              # [N, Cout, L]
    Ho = (H + 2 * pad - k) // stride + 1
    if b is not None:
        y = y + b.double()[None, :, None]
    g2 = gd.reshape(N, Cout, -1)
    dw = torch.einsum('ncl,nkl->ck', g2, cols).reshape(wd.shape)
    dcols = torch.matmul(wm.t(), g2)
    dx = F.fold(dcols, (H, W), k, padding=pad, stride=stride)
    db = g2.sum((0, 2)) if b is not None else None
    return y.reshape(N, Cout, Ho, Ho), dx, dw, db


# (e) HRNet-W32 (BASELINE configs[3], B = 32 per GPU): every distinct convolution problem of its training forward
#     (tools/list_conv_problems.py --model hrnet32): 3x3 stems, the 32 / 64 / 128 / 256-channel branches at 64 / 32 / 16 / 8
#     pixels, strided fuse-layer chains, 1x1 fuse / transition / head convolutions (480-channel heads with 7 / 42 outputs)
HRNET_B = 32
HRNET_PROBLEMS = [
    (256, 3, 64, 3, 2, 1, False), (128, 64, 64, 3, 2, 1, False), (64, 32, 32, 3, 1, 1, False), (64, 32, 32, 3, 2, 1, False),
    (64, 32, 64, 3, 2, 1, False), (64, 32, 128, 1, 1, 0, False), (64, 64, 64, 1, 1, 0, False), (64, 64, 64, 3, 1, 1, False),
    (64, 64, 256, 1, 1, 0, False), (64, 128, 256, 3, 2, 1, True), (64, 256, 32, 1, 1, 0, False), (64, 256, 32, 3, 1, 1, False),
    (64, 256, 64, 1, 1, 0, False), (64, 256, 64, 3, 2, 1, False), (64, 256, 128, 1, 1, 0, False), (64, 256, 256, 1, 1, 0, False),
    (64, 480, 7, 1, 1, 0, True), (64, 480, 42, 1, 1, 0, True), (64, 480, 480, 1, 1, 0, True), (32, 32, 32, 3, 2, 1, False),
    (32, 32, 128, 3, 2, 1, False), (32, 64, 32, 1, 1, 0, False), (32, 64, 64, 3, 1, 1, False), (32, 64, 64, 3, 2, 1, False),
    (32, 64, 128, 3, 2, 1, False), (32, 64, 256, 1, 1, 0, False), (32, 256, 64, 1, 1, 0, False), (32, 256, 256, 1, 1, 0, False),
    (32, 256, 512, 3, 2, 1, True), (16, 32, 256, 3, 2, 1, False), (16, 64, 256, 3, 2, 1, False), (16, 128, 32, 1, 1, 0, False),
    (16, 128, 64, 1, 1, 0, False), (16, 128, 128, 3, 1, 1, False), (16, 128, 256, 3, 2, 1, False), (16, 128, 512, 1, 1, 0, False),
    (16, 256, 128, 1, 1, 0, False), (16, 256, 256, 1, 1, 0, False), (16, 256, 512, 1, 1, 0, False), (16, 512, 1024, 3, 2, 1, True),
    (8, 256, 32, 1, 1, 0, False), (8, 256, 64, 1, 1, 0, False), (8, 256, 128, 1, 1, 0, False), (8, 256, 256, 1, 1, 0, False),
    (8, 256, 256, 3, 1, 1, False), (8, 256, 1024, 1, 1, 0, False), (8, 1024, 2048, 1, 1, 0, True),
]
# (f) the second model family (lijun_model_graph, B = 64): the convolution problems it has beyond the list above, and the
#     Linear problems of its decoder (rows = 64 images x tokens; tools/list_conv_problems.py --model family_b)
FAMILY_B_CONVS = [(64, 256, 256, 1, 1, 0, False), (8, 2048, 256, 1, 1, 0, False)]
LINEAR_PROBLEMS = [(64, 2048, 509, True), (384, 252, 778, False), (8064, 256, 768, True), (16128, 128, 384, True),
                   (32256, 64, 192, True), (32256, 64, 3, True), (8192, 252, 1, True)]


@pytest.mark.parametrize('prob', CONV_PROBLEMS + FAMILY_B_CONVS + [(HRNET_B,) + q for q in HRNET_PROBLEMS])
def test_conv_problem_at_bench_batch(prob):
    from renderih_amd import ops
    B = 64
    if len(prob) == 8:          # (batch, ...): an HRNet-W32 problem at its own bench batch
        B, prob = prob[0], prob[1:]
    H, Cin, Cout, k, s, p, bias = prob
    d = torch.device('cuda:0')
    g = torch.Generator(device=d).manual_seed(1000 + H + 7 * Cin + 13 * Cout + k)
    x = torch.randn(B, Cin, H, H, device=d, generator=g)
    w = torch.randn(Cout, Cin, k, k, device=d, generator=g) / math.sqrt(Cin * k * k)
    b = torch.randn(Cout, device=d, generator=g) if bias else None
    Ho = (H + 2 * p - k) // s + 1
    gy = torch.randn(B, Cout, Ho, Ho, device=d, generator=g)
    yr, dxr, dwr, dbr = _ref_conv_fp64(x, w, b, s, p, gy)

    cpad = 4 if Cin == 3 else Cin
    xg = torch.zeros(B, H, H, cpad, device=d)
    xg[..., :Cin] = x.permute(0, 2, 3, 1)
    xg.requires_grad_(Cin != 3)
    wg = w.clone().requires_grad_(True)
    bg = b.clone().requires_grad_(True) if bias else None
    yg = ops.conv2d(xg, wg, bg, stride=s, pad=p)
    assert_close(yg.permute(0, 3, 1, 2), yr, what='y %s' % (prob,))
    yg.backward(gy.permute(0, 2, 3, 1).contiguous())
    # (round 6: gradients at north_star's 1e-4 + 1e-5 max -- ten times what the engines measure against fp64, 2-8e-7 of the maximum;
    # until round 5 these bars were 1e-3 + 1e-4 max and would have let a hundredfold regression pass)
    assert_close(wg.grad, dwr, 1e-4, 1e-5, 'dw %s' % (prob,))
    if Cin != 3:
        assert_close(xg.grad.permute(0, 3, 1, 2), dxr, 1e-4, 1e-5, 'dx %s' % (prob,))
    if bias:
        assert_close(bg.grad, dbr, 1e-4, 1e-5, 'db %s' % (prob,))


@pytest.mark.parametrize('prob', LINEAR_PROBLEMS)
def test_linear_problem_at_bench_batch(prob):
    """nn.Linear problems of the decoders at B = 64 row counts through ops.linear: output, input / weight / bias gradients
    against fp64 matmuls."""
    from renderih_amd import ops
    rows, K, N, bias = prob
    d = torch.device('cuda:0')
    g = torch.Generator(device=d).manual_seed(2000 + rows + 7 * K + 13 * N)
    x = torch.randn(rows, K, device=d, generator=g)
    w = torch.randn(N, K, device=d, generator=g) / math.sqrt(K)
    b = torch.randn(N, device=d, generator=g) if bias else None
    gy = torch.randn(rows, N, device=d, generator=g)
    yr = x.double() @ w.double().t() + (b.double() if bias else 0.0)
    dxr, dwr, dbr = gy.double() @ w.double(), gy.double().t() @ x.double(), gy.double().sum(0)
    xg, wg = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    bg = b.clone().requires_grad_(True) if bias else None
    yg = ops.linear(xg, wg, bg)
    assert_close(yg, yr, what='y %s' % (prob,))
    yg.backward(gy)
    assert_close(xg.grad, dxr, 1e-4, 1e-5, 'dx %s' % (prob,))
    assert_close(wg.grad, dwr, 1e-4, 1e-5, 'dw %s' % (prob,))
    if bias:
        assert_close(bg.grad, dbr, 1e-4, 1e-5, 'db %s' % (prob,))


def _build(dropout=0.0, seed=0):
    from renderih_amd.model import build_model
    m = build_model(dropout)
    sd = testing.deterministic_state(m.state_dict(), seed=seed)
    m.load_state_dict(sd)
    return m.to('cuda:0'), sd


def test_train_mode_forward_at_b16_vs_oracle():
    """Training-mode forward at B = 16 (batch statistics over >= 1024 samples per channel) against the CPU oracle on the same
    weights and images.  Measured on MI355X (profiles/r02/pytest_shapes_m5.log): the HIP path is 2e-4 .. 9e-4 from the fp32
    CPU oracle at this size -- NOT within 1e-4 -- so the comparison is anchored on the oracle's fp64 run to tell which side
    the distance comes from: the HIP result must be within 1e-3 of fp64 outright, and no further from fp64 than twice the
    fp32 CPU oracle's own distance (+2e-5).  The randomly initialised 50-layer network amplifies fp32 round-off through its
    batch statistics even at this batch size; eval mode (running statistics) meets 1e-4 (test_gpu_model.py)."""
    from oracle import net_oracle
    m, sd = _build(0.0, seed=3)
    m.train()
    img = testing.seeded_image(16, 21)
    graph = net_oracle.graph_from_dicts(assets.load_graph_dict('left'), assets.load_graph_dict('right'))
    w32, _ = net_oracle.run(sd, graph, img, True, torch.float32, False)
    w64, _ = net_oracle.run(sd, graph, img, True, torch.float64, False)
    with torch.no_grad():
        got = testing.flatten_outputs(m(img.cuda()))
    report = {}
    for k in w64:
        if k.startswith('params.'):
            continue        # per-image scalars can sit near zero; they are functions of the checked meshes
        e_got, e_ref = testing.assert_fp32_equivalent(got[k], w32[k], w64[k], k=2.0, floor=2e-5, what=k)
        assert e_got < 1e-3, (k, e_got)
        report[k] = (e_got, e_ref, testing.rel_err(got[k], w32[k]))
    print('B=16 train-mode forward: worst HIP-vs-fp64 %.2e, CPU-fp32-vs-fp64 %.2e, HIP-vs-CPU-fp32 %.2e'
          % (max(v[0] for v in report.values()), max(v[1] for v in report.values()), max(v[2] for v in report.values())))


def test_training_step_at_bench_batch_is_finite_and_deterministic():
    """B = 64, the exact problem sizes of the timed step: forward + scalar loss + backward twice from the same state; all
    gradients finite and bit-identical between the two runs (fixed-order split-K reductions, no atomics)."""
    from oracle import net_oracle
    m, _ = _build(0.0, seed=1)
    m.train()
    img = testing.seeded_image(64, 5).cuda()
    grads = []
    for _ in range(2):
        m.zero_grad(set_to_none=True)
        net_oracle.scalar_loss(m(img)).backward()
        grads.append({k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None})
    assert len(grads[0]) > 800          # 843 live parameter tensors (the rest never receive a gradient)
    for k, g in grads[0].items():
        assert bool(torch.isfinite(g).all()), k
        assert torch.equal(g, grads[1][k]), 'gradient of %s differs between two identical steps' % k


def test_parameter_gradients_at_bench_batch_vs_fp64_anchor():
    """(d) B = 64, the timed problem sizes: forward + scalar loss + backward of the HIP path against the fp64 run of the CPU
    oracle, on a sample of parameters that touches every kernel family at its bench-size row counts -- stem, one convolution /
    BatchNorm per trunk layer (statistics and backward over 16384 .. 1048576 samples per channel), aux decoders, mid convs,
    and per decoder level a Chebyshev block, attention projections, LayerNorm, embeddings, plus the output heads.  The fp64
    gradients come from tests/golden/b64_grads.npz (tests/golden/make_b64_grads.py: ~10 minutes of CPU, so committed as a
    fixture together with the fp32 oracle's own distance from them; tensors of more than 16384 elements are compared on a seeded
    random sample of 16384).  What an fp32 implementation can reach here is set by the network, not by the kernel: the fp32
    ORACLE's gradients are up to 3.6e-2 (relative l2) from its own fp64 run on the trunk weights, cosine 0.9994 -- so the bar is
    relative to that: per tensor, relative l2 distance from fp64 within RATIO x the fp32 oracle's (floor 1e-4) and 1 - cosine
    within RATIO^2 x the fp32 oracle's (1 - cos ~ e^2 / 2; floor 1e-6); the median ratio over the sample within 2."""
    RATIO = 4.0
    import os
    import numpy as np
    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'b64_grads.npz'))
    seed_state, seed_img, bsz = (int(v) for v in fx['meta/seeds'])
    assert bsz == B
    from oracle import net_oracle
    m, _ = _build(0.0, seed=seed_state)
    m.train()
    img = testing.seeded_image(B, seed_img).cuda()
    net_oracle.scalar_loss(m(img)).backward()
    grads = {k: p.grad for k, p in m.named_parameters() if p.grad is not None}
    names = [k[4:] for k in fx.files if k.startswith('g64/')]
    assert len(names) >= 50
    rows, bad = [], []
    for k in names:
        assert k in grads, 'no gradient for %s' % k
        g = grads[k].double().flatten().cpu()
        assert bool(torch.isfinite(g).all()) and tuple(grads[k].shape) == tuple(int(v) for v in fx['shape/' + k]), k
        r = torch.from_numpy(fx['g64/' + k]).double().flatten()
        if g.numel() > r.numel():
            import zlib
            idx = np.sort(np.random.RandomState(zlib.crc32(k.encode()) & 0x7FFFFFFF).choice(g.numel(), r.numel(), replace=False))
            g = g[torch.from_numpy(idx)]
        if testing.is_null_gradient(k):
            continue
        cos = float(torch.dot(g, r) / (g.norm() * r.norm()).clamp_min(1e-300))
        e = float((g - r).norm() / r.norm().clamp_min(1e-300))
        e32, c32 = float(fx['e32/' + k]), float(fx['c32/' + k])
        rows.append((e / max(e32, 1e-12), k, e, e32, cos, c32))
        if not (e <= RATIO * e32 + 1e-4 and 1.0 - cos <= RATIO * RATIO * (1.0 - c32) + 1e-6):
            bad.append(k)
    rows.sort(reverse=True)
    print('B=64 parameter gradients vs the fp64 anchor (%d tensors); rel. l2 distance: HIP | fp32 oracle | ratio; cosines' % len(rows))
    for ratio, k, e, e32, cos, c32 in rows[:12]:
        print('  %-66s %.2e %.2e %5.2f   %.7f %.7f' % (k, e, e32, ratio, cos, c32))
    med = sorted(r[0] for r in rows)[len(rows) // 2]
    print('  median ratio %.2f' % med)
    assert not bad, 'gradients further from fp64 than %g x the fp32 oracle: %s' % (RATIO, bad)
    assert med <= 2.0, 'median distance ratio %.2f' % med


def test_b64_matches_the_reference_fixture_directly():
    """tests/golden/net_train_b64_ref.npz (make_b64_ref.py, round 6 -- round-5 verdict item 3 iv): the REAL reference modules
    (`models.model.HandNET_GCN`, train mode, fp32) at the BENCHMARK batch B = 64 on the seeds of b64_grads.npz.  Until round 6 the
    largest batch the reference itself pinned was B = 16 and the B = 64 anchor was the oracle's.  Here the HIP path meets reference
    output directly, no oracle in between:
      * every output of the 4-tuple (signature samples): there is no fp64 anchor for the outputs at this size, so the bar is direct,
        |HIP - reference| <= 2e-3 of the tensor's maximum (two fp32 evaluation orders of this 50-layer batch-statistics network
        land 1e-4 .. 8e-4 apart), and the loss within 1e-3 relative;
      * the sampled parameter gradients: |HIP - fp64 anchor| (relative l2 on the 16384-element sample) within 4 x the REFERENCE's
        own distance e32ref from that anchor (floor 1e-4), cosine likewise, median ratio <= 2 -- test_parameter_gradients_at_
        bench_batch_vs_fp64_anchor's band with the reference's fp32 run in the role of the oracle's;
      * the direct distance |HIP - reference| / |reference| on the same samples <= 5 x e32ref + 1e-4 (triangle inequality);
      * BatchNorm running buffers after the step at 1e-3."""
    import os
    import zlib
    import numpy as np
    from oracle import net_oracle
    here = os.path.dirname(os.path.abspath(__file__))
    z = np.load(os.path.join(here, 'golden', 'net_train_b64_ref.npz'))
    fx = np.load(os.path.join(here, 'golden', 'b64_grads.npz'))
    seed_state, seed_img, bsz = (int(v) for v in z['meta/seeds'])
    assert bsz == B and [int(v) for v in fx['meta/seeds']] == [seed_state, seed_img, bsz]
    m, _ = _build(0.0, seed=seed_state)
    m.train()
    out = m(testing.seeded_image(B, seed_img).cuda())
    worst = (0.0, '')
    for k, v in testing.flatten_outputs(out).items():
        st, sa = testing.signature(v)
        assert int(st[4]) == int(z['out/' + k + '#stats'][4]), k
        ref = z['out/' + k + '#samp'].astype(np.float64)
        scale = max(float(z['out/' + k + '#stats'][3]), 1e-30)
        assert np.isfinite(sa).all(), k
        e = float(np.abs(sa.astype(np.float64) - ref).max()) / scale
        worst = max(worst, (e, k))
        assert e <= 2e-3, '%s: %.3g of the tensor maximum from the reference' % (k, e)
    loss = net_oracle.scalar_loss(out)
    assert abs(loss.item() - float(z['loss'])) <= 1e-3 * abs(float(z['loss'])), (loss.item(), float(z['loss']))
    loss.backward()
    grads = {k: p.grad for k, p in m.named_parameters() if p.grad is not None}
    names = [k[4:] for k in z.files if k.startswith('g32/')]
    assert len(names) >= 50
    rows, bad = [], []
    for k in names:
        g = grads[k].double().flatten().cpu()
        r64 = torch.from_numpy(fx['g64/' + k]).double().flatten()
        r32 = torch.from_numpy(z['g32/' + k]).double().flatten()
        if g.numel() > r64.numel():
            idx = np.sort(np.random.RandomState(zlib.crc32(k.encode()) & 0x7FFFFFFF).choice(g.numel(), r64.numel(), replace=False))
            g = g[torch.from_numpy(idx)]
        if testing.is_null_gradient(k):
            continue
        e = float((g - r64).norm() / r64.norm().clamp_min(1e-300))
        cos = float(torch.dot(g, r64) / (g.norm() * r64.norm()).clamp_min(1e-300))
        edir = float((g - r32).norm() / r32.norm().clamp_min(1e-300))
        eref, cref = float(z['e32ref/' + k]), float(z['c32ref/' + k])
        rows.append((e / max(eref, 1e-12), k, e, eref, edir))
        if not (e <= 4.0 * eref + 1e-4 and 1.0 - cos <= 16.0 * (1.0 - cref) + 1e-6 and edir <= 5.0 * eref + 1e-4):
            bad.append('%s: vs fp64 %.3g (reference %.3g), vs reference %.3g' % (k, e, eref, edir))
    rows.sort(reverse=True)
    print('B = 64 against the REFERENCE fixture: outputs worst %.3g of max (%s); %d gradient tensors; rel. l2: HIP vs fp64 | reference vs '
          'fp64 | ratio | HIP vs reference' % (worst[0], worst[1], len(rows)))
    for ratio, k, e, eref, edir in rows[:10]:
        print('  %-66s %.2e %.2e %5.2f  %.2e' % (k, e, eref, ratio, edir))
    med = sorted(r[0] for r in rows)[len(rows) // 2]
    print('  median ratio %.2f' % med)
    assert not bad, 'gradients further from fp64 / from the reference than the band allows:\n' + '\n'.join(bad)
    assert med <= 2.0, 'median distance ratio %.2f' % med
    sd = m.state_dict()
    for k in z.files:
        if k.startswith('bnstat/'):
            testing.assert_close(sd[k[7:]].float(), torch.from_numpy(np.asarray(z[k])).float(), 1e-3, 1e-4, k)
