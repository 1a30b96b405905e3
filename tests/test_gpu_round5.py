"""Round-5 parity additions (verdict items 4 and 10), all through the C ABI on the GPU:

* a CAPTURED training step replayed on data of other magnitudes than the capture batch (engine 2's operand scales come from
  device-resident bounds that must follow the data inside the hipGraph);
* HIP train-mode gradients against the REFERENCE's gradient goldens directly (B = 2 fixture, and the B = 16 fixture of
  tests/golden/make_b16_train.py), not only through the oracle;
* BASELINE configs[0]: the evaluation plumbing of apps/eval_interhand.py:255-263, 298-420 through the drop-in import paths
  (`models.model.load_model`, `models.manolayer.ManoLayer`) on 16 synthetic crops at batch 2, against the CPU oracle.
"""
import os
import sys
import numpy as np
import pytest
import torch

from renderih_amd import assets, testing

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(HERE, 'golden')


def _tools():
    import test_gpu_model as TM
    return TM._build, TM._grad_report


def _graph():
    from oracle import net_oracle
    return net_oracle.graph_from_dicts(assets.load_graph_dict('left'), assets.load_graph_dict('right'))


# ------------------------------------------------------------------------------------------------ captured step, other magnitudes
@pytest.mark.parametrize('bn_mode,bsz', [('train', 2), ('frozen', 2), ('train', 16)])
def test_captured_step_follows_the_data_magnitude(bn_mode, bsz, monkeypatch):
    """TrainStep captured on batch A (engine 2: every GEMM's power-of-two operand scales come from bound blocks written by
    kernels INSIDE the graph) and replayed on A, 1000 A, A / 1000 (A / 100 with batch statistics) and A with one pixel at 1e4: outputs and every parameter
    gradient of each replay against an eager engine-0 (native f32 MFMA) run of the same model on the same data, inside the
    suite's fp64-anchored bands: testing.assert_fp32_equivalent (k = 4 + 2e-5) for every output; for the gradients
    _grad_report's 6x band with the allowance of the HRNet B = 2 test (at most 3 % of the tensors outside, none beyond
    max(5 %, 20 x the exact-fp32 engine's own distance from fp64)).  Why not the 0.5 % of the B = 3 ResNet test: at B = 2 a single
    ReLU decision that differs between two fp32 summation orders moves a whole backward path at once -- on MI355X the excursions
    came in clusters of five tensors (profiles/r05/ab/c1_pytest_r5.log and c5_pytest_gpu.log: `layers.0.img_ex_left.encoder.
    {position_embeddings, proj, self_attn.w_qs, ff.fc1.*}` together, `layers.2.attn.{layer_norm2, ffL.*}` together), 0-2 clusters
    per variant, and WHICH clusters changed when the 3x3 convolutions moved to the halo kernel (another summation order).  What
    this test is after -- a stale bound -- shows as inf / NaN or as errors of order one, which the gross bound catches.  Nothing
    may be inf / NaN.  A stale or capture-time bound would overflow the fp16 planes at 1000 A (inf) or
    flush A / 1000 to zero.  bn_mode 'frozen' = eval-mode BatchNorm (running statistics) with autograd on: the magnitude then
    travels through the whole trunk instead of being normalised away by the stem's batch statistics.
    Round 6 (verdict item 3 i): the B = 2 probe is ill conditioned (128 statistics samples at the 8 x 8 level), so (a) the same test
    runs at B = 16 -- 1024 samples, where the reference pinned a fixture (net_train_b16.npz) -- with the ResNet test's own 0.5 %
    allowance, and (b) at B = 2 the 3 % allowance of round 5 is no longer taken on faith: the CPU fp32 ORACLE's gradients on the same
    data (another fp32 summation order of the same network, pinned to the reference) go through the same band, and the HIP path
    may have at most 1 % of its tensors outside OR no more than that independent fp32 implementation has (+ 2)."""
    from oracle import net_oracle
    from renderih_amd import ops
    from renderih_amd.train import TrainStep
    _build, _grad_report = _tools()
    if ops.ENGINE != 2:
        pytest.skip('engine 2 is not the configured engine')
    training = bn_mode == 'train'
    A = testing.seeded_image(bsz, 51)
    spike = A.clone()
    # the outlier: 1e4 with batch statistics (the stem's BatchNorm normalises it away), 1e3 with frozen statistics -- there a pixel of
    # 1e4 makes the probe itself ill conditioned: the fp64 oracle's gradients move by up to 3.3e-3 of their maximum (ten tensors by more
    # than 2e-4) when the IMAGE is perturbed by 2^-22 relative, against 1.2e-4 / none at 1e3 and 1.8e-4 / none without an outlier
    # (tools/r6_spike_conditioning.py, profiles/r06/spike_conditioning.txt).  1e3 is still 185 x the image's maximum: a bound left over
    # from the capture batch would scale it to 4e6 and overflow the fp16 planes.
    spike_mag = 1e4 if training else 1e3
    spike[0, 1, 100, 37] = spike_mag
    # (train mode: A / 100, not A / 1000 -- at A / 1000 the stem's batch variance (2e-6) drops below BatchNorm's eps (1e-5), the
    # backward through that BatchNorm amplifies round-off ~300-fold and even the exact-fp32 engine's gradients are 3 % from fp64:
    # a degenerate operating point that says nothing about operand scales.  With frozen statistics the full 1e-3 is used.)
    small = 1e-2 if training else 1e-3
    variants = [('A', A), ('1e3 A', 1e3 * A), ('%g A' % small, small * A), ('A + one %g pixel' % spike_mag, spike)]
    if bsz > 2:
        variants = variants[:3]         # (the fp64 oracle run per variant is the cost of this test: ~1 min each at B = 16)
    m2, sd = _build(0.0, seed=13)
    m2.train(training)
    m2.decoder.unsample_layer.weight.requires_grad_(False)
    m0, _ = _build(0.0, seed=13)
    m0.train(training)
    m0.decoder.unsample_layer.weight.requires_grad_(False)
    graph = _graph()
    holder = {}

    def loss_fn(out, lab):
        holder['out'] = out
        return net_oracle.scalar_loss(out)
    opt = torch.optim.SGD([p for p in m2.parameters() if p.requires_grad], lr=0.0)       # lr 0: the state stays put
    try:
        step = TrainStep(m2, opt, loss_fn, (A.cuda().clone(), {}), process_group=False, stages='auto')
        assert step.use_graph or not torch.cuda.is_available()      # (CPU dry runs of this file on the ABI emulator launch eagerly)
        for name, X in variants:
            loss = step(X.cuda(), {})
            torch.cuda.synchronize()
            assert bool(torch.isfinite(loss)), name
            got_out = {k: v.detach().clone() for k, v in testing.flatten_outputs(holder['out']).items()}
            got = [(k, p.grad.detach().clone()) for k, p in m2.named_parameters() if p.grad is not None]
            for k, v in got_out.items():
                assert bool(torch.isfinite(v).all()), (name, k)
            for k, g in got:
                assert bool(torch.isfinite(g).all()), (name, k)
            # the fp32 reference of the band: engine 0 (exact-fp32 MFMA products), launched eagerly, same weights and data
            monkeypatch.setattr(ops, 'ENGINE', 0)
            try:
                for mod in m0.modules():
                    if isinstance(mod, torch.nn.BatchNorm2d) and training:
                        mod.reset_running_stats()
                m0.zero_grad(set_to_none=True)
                out0 = m0(X.cuda())
                net_oracle.scalar_loss(out0).backward()
            finally:
                monkeypatch.setattr(ops, 'ENGINE', 2)
            w0 = {k: v.detach().cpu() for k, v in testing.flatten_outputs(out0).items()}
            g0 = {k: p.grad.detach().cpu() for k, p in m0.named_parameters() if p.grad is not None}
            w64, g64 = net_oracle.run(sd, graph, X, training, torch.float64, True)
            g64 = {k: v for k, v in g64.items() if k in g0}         # (the frozen up-sampling matrix has a gradient in the oracle)
            assert {k for k, _ in got} == set(g0), name
            worst = (0.0, 0.0)
            for k in w64:
                worst = max(worst, testing.assert_fp32_equivalent(got_out[k], w0[k], w64[k], k=4.0, floor=2e-5,
                                                                  what='%s (%s): %s' % (name, bn_mode, k)))
            if bsz > 2:
                nloose, n = _grad_report(got, g0, g64, max_loose=0.005)
                nctl = -1
            else:
                nloose, n = _grad_report(got, g0, g64, max_loose=1.0)       # (the gross bound: nothing beyond max(5 %, 20 x))
                # control: the CPU fp32 oracle through the same band
                _, g32o = net_oracle.run(sd, graph, X, training, torch.float32, True)
                nctl, _ = _grad_report([(k, g32o[k]) for k, _ in got], g0, g64, max_loose=None)
                assert nloose <= max(0.01 * n, nctl + 2), (
                    '%s: %d of %d HIP gradient tensors outside the band, the independent fp32 oracle has %d' % (name, nloose, n, nctl))
            print('replay on %-18s (%s, B = %d): outputs worst %.3g vs fp64 (engine 0: %.3g); gradients %d/%d outside the band '
                  '(fp32 oracle control: %d)' % (name, bn_mode, bsz, worst[0], worst[1], nloose, n, nctl))
    finally:
        ops.DROPOUT_SEED_TENSOR = None


# ------------------------------------------------------------------------------------------------ reference gradients, directly
def _reference_gradient_check(z, named_grads, g64_samp, k=6.0, floor=2e-4, max_loose=0.005, nsamp=32, what=''):
    """HIP gradients against the gradient SAMPLES the real reference produced (`grad/<name>#samp` of a fixture) -- the band
    of test_gpu_model._grad_report with the reference's own fp32 values in the role of the fp32 reference:
    |HIP - fp64| <= k |reference - fp64| + floor (relative to the reference tensor's max |g|, `#stats`[3]) for all but a
    fraction max_loose of the tensors, nothing beyond max(5 %, 20 x); the direct distance |HIP - reference| is printed and
    bounded by the triangle inequality's (k + 1) |reference - fp64| + floor."""
    n, loose, gross, worst, worst_direct = 0, [], [], (0.0, ''), (0.0, '')
    for name, g in named_grads:
        if testing.is_null_gradient(name):
            continue
        ref = z['grad/' + name + '#samp'].astype(np.float64)
        scale = max(float(z['grad/' + name + '#stats'][3]), 1e-30)
        assert int(z['grad/' + name + '#stats'][4]) == g.numel(), name
        samp = testing.signature(g, nsamp=nsamp)[1].astype(np.float64)
        s64 = g64_samp(name).astype(np.float64)
        e_ref = float(np.abs(ref - s64).max()) / scale
        e_got = float(np.abs(samp - s64).max()) / scale
        e_dir = float(np.abs(samp - ref).max()) / scale
        n += 1
        worst = max(worst, (e_got / max(e_ref, floor), name))
        worst_direct = max(worst_direct, (e_dir, name))
        if not (e_got <= k * e_ref + floor) or not (e_dir <= (k + 1) * e_ref + floor):
            loose.append('%s: vs fp64 %.3g, vs reference %.3g (reference vs fp64 %.3g)' % (name, e_got, e_dir, e_ref))
        if not (e_got <= max(0.05, 20 * e_ref)) or not np.isfinite(samp).all():
            gross.append('%s: %.3g' % (name, e_got))
    print('%s: %d gradient tensors against the reference\'s samples, %d outside the %gx band; worst ratio %.1fx (%s); largest direct '
          'distance from the reference %.3g of max|g| (%s)' % (what, n, len(loose), k, worst[0], worst[1], worst_direct[0], worst_direct[1]))
    for line in loose[:40]:
        print('   outside:', line)
    assert not gross, 'gradients grossly off:\n' + '\n'.join(gross[:20])
    assert len(loose) <= max_loose * n, '%d/%d gradient tensors outside the band:\n%s' % (len(loose), n, '\n'.join(loose[:20]))


def test_train_gradients_against_the_reference_golden_directly():
    """tests/golden/net_train.npz holds signatures of every parameter gradient the REAL reference computed (B = 2, train mode,
    seeded weights and image).  Rounds 1-4 compared them with the oracle only (tests/test_oracle_golden.py) and the HIP path
    with the oracle; here the HIP gradients meet the reference's values themselves.  The fp64 anchor is the oracle's fp64 run of
    the same inputs."""
    from oracle import net_oracle
    _build, _ = _tools()
    z = np.load(os.path.join(GOLDEN, 'net_train.npz'))
    m, sd = _build(0.0)
    m.train()
    img = testing.seeded_image(2, 0)
    out = m(img.cuda())
    loss = net_oracle.scalar_loss(out)
    assert abs(loss.item() - float(z['loss'])) <= 1e-3 * abs(float(z['loss']))
    loss.backward()
    names = [str(n) for n in z['grad_names']]
    params = dict(m.named_parameters())
    assert set(names) == {k for k, p in params.items() if p.grad is not None}
    _, g64 = net_oracle.run(sd, _graph(), img, True, torch.float64, True)
    _reference_gradient_check(z, [(k, params[k].grad) for k in names],
                              lambda k: testing.signature(g64[k], nsamp=32)[1], what='B = 2 reference golden')


def test_b16_train_matches_the_reference_fixture():
    """tests/golden/net_train_b16.npz (make_b16_train.py): the REAL reference modules in train mode at B = 16 -- outputs, loss,
    every parameter gradient, BatchNorm running buffers -- with the oracle's fp64 samples of the same run beside them.  HIP
    outputs: within k = 4 of the reference's own distance from fp64 (+ 2e-5), i.e. testing.assert_fp32_equivalent's band on the
    stored samples; gradients: _reference_gradient_check.  No host-side oracle run: the GPU box only reads the fixture."""
    _build, _ = _tools()
    from oracle.net_oracle import scalar_loss
    z = np.load(os.path.join(GOLDEN, 'net_train_b16.npz'))
    B = int(z['meta_B'])
    m, _ = _build(0.0, seed=int(z['meta_seed_state']))
    m.train()
    out = m(testing.seeded_image(B, int(z['meta_seed_img'])).cuda())
    worst = (0.0, '')
    for k, v in testing.flatten_outputs(out).items():
        st, sa = testing.signature(v)
        assert int(st[4]) == int(z['out/' + k + '#stats'][4]), k
        ref, s64 = z['out/' + k + '#samp'].astype(np.float64), z['out64/' + k + '#samp'].astype(np.float64)
        scale = max(float(np.abs(s64).max()), 1e-30)
        e_ref, e_got = float(np.abs(ref - s64).max()) / scale, float(np.abs(sa.astype(np.float64) - s64).max()) / scale
        assert np.isfinite(sa).all(), k
        assert e_got <= 4.0 * e_ref + 2e-5, '%s: err vs fp64 %.3g exceeds 4 x the reference\'s %.3g + 2e-5' % (k, e_got, e_ref)
        worst = max(worst, (e_got / max(e_ref, 2e-5), k))
    loss = scalar_loss(out)
    assert abs(loss.item() - float(z['loss'])) <= 1e-3 * abs(float(z['loss'])), (loss.item(), float(z['loss']))
    loss.backward()
    names = [str(n) for n in z['grad_names']]
    params = dict(m.named_parameters())
    assert set(names) == {k for k, p in params.items() if p.grad is not None}
    print('B = 16 reference fixture: outputs worst ratio %.2fx (%s)' % worst)
    _reference_gradient_check(z, [(k, params[k].grad) for k in names], lambda k: z['grad64/' + k + '#samp'],
                              what='B = 16 reference fixture')
    sd = m.state_dict()
    for k in z.files:
        if k.startswith('bnstat/'):
            testing.assert_close(sd[k[7:]].float(), torch.from_numpy(np.asarray(z[k])).float(), 1e-3, 1e-4, k)


# ------------------------------------------------------------------------------------------------ configs[0]: evaluation plumbing
class _Jr:
    """apps/eval_interhand.py:147-170 (`Jr`): the 16-joint MANO regressor + five one-hot finger tips, re-ordered to 21 joints."""

    def __init__(self, J_regressor, device='cuda'):
        J = J_regressor.clone().detach()
        tip = torch.zeros_like(J[:5])
        for i, v in enumerate((745, 317, 444, 556, 673)):
            tip[i, v] = 1.0
        J = torch.cat([J, tip], dim=0)
        order = [0, 13, 14, 15, 16, 1, 2, 3, 17, 4, 5, 6, 18, 10, 11, 12, 19, 7, 8, 9, 20]
        self.J_regressor = J[order].contiguous().to(device)

    def __call__(self, v):
        return torch.matmul(self.J_regressor, v)


def _fix_shape(mano_layer):
    """dataset/interhand.py:22-25: callers flip the left hand's first shape direction in place after construction."""
    if torch.sum(torch.abs(mano_layer['left'].shapedirs[:, 0, :] - mano_layer['right'].shapedirs[:, 0, :])) < 1:
        mano_layer['left'].shapedirs[:, 0, :] *= -1


def _script_metrics(result, v_l, v_r, J):
    """apps/eval_interhand.py:306-363, 395-404 for one batch: (ori joint error [B,21], ori vertex error [B,778], Procrustes joint
    error [B]) per hand, in plain torch on whatever device the tensors live on."""
    from oracle import metrics_oracle
    res = {}
    for side, vg in (('left', v_l), ('right', v_r)):
        jg = J[side](vg)
        root_g = jg[:, 0:1]
        jg0, vg0 = jg - root_g, vg - root_g
        vp = result['verts3d'][side]
        jp = J[side](vp)
        root_p = jp[:, 0:1]
        jp0, vp0 = jp - root_p, vp - root_p
        pa = metrics_oracle.similarity_transform(jp0.double().cpu(), jg0.double().cpu())
        res[side] = (torch.linalg.norm(jp0 - jg0, ord=2, dim=-1), torch.linalg.norm(vp0 - vg0, ord=2, dim=-1),
                     torch.sqrt(((pa - jg0.double().cpu()) ** 2).sum(-1)).mean(-1))
    return res


def test_config0_evaluation_plumbing_through_the_drop_in_paths(tmp_path):
    """BASELINE configs[0] ("batch = 2, ResNet50 encoder + MANO layer, forward via apps/eval_interhand.py on 16 crops",
    plumbing): the objects are constructed and called exactly as the script does -- `load_model(cfg)` (its line 239
    alternative; family (b) has its own tests), `load_state_dict(strict=False)`, `.eval()`, `.cuda()`, `ManoLayer(path,
    center_idx=None)` from a MANO pickle, `fix_shape`, `Jr(mano_layer.J_regressor)`, then per batch `network(imgTensors)` and the
    script's error arithmetic -- through the DROP-IN import paths `models.model` / `models.manolayer`, on 16 synthetic 256x256
    crops at batch 2 whose ground-truth meshes come from the MANO layer itself (as the dataset builder makes them,
    dataset/interhand.py:102-109).  Checked: (1) the 16 predicted meshes against the CPU oracle's forward of the same weights at
    1e-4; (2) the script's MPJPE / MPVPE / PA-MPJPE numbers computed from the HIP predictions against the same numbers computed
    from the oracle's predictions; (3) `renderih_amd.evaluate.evaluate` (the GPU-resident metrics loop) against the script's
    arithmetic.  The product has no CPU path by design, so "CPU-only" here means the oracle side of the comparison."""
    root = os.path.dirname(HERE)
    if root not in sys.path:
        sys.path.insert(0, root)
    from models.model import load_model                 # drop-in import paths (reference: models/model.py:40-60)
    from models.manolayer import ManoLayer, rodrigues_batch      # (reference: models/manolayer.py:100-322, 32-48)
    from oracle import net_oracle
    from renderih_amd.evaluate import evaluate
    torch.manual_seed(0)
    network = load_model(None)                          # cfg = None: utils/defaults.yaml values (renderih_amd.config)
    state = testing.deterministic_state(network.state_dict(), seed=23)
    torch.save({'network': {'module.' + k: v for k, v in state.items()}}, tmp_path / 'ckpt.pth')      # a DDP-saved checkpoint
    st = torch.load(tmp_path / 'ckpt.pth', map_location='cpu')
    if 'network' in st:
        st = st['network']
    try:                                                # eval_interhand.py:245-251
        missing = network.load_state_dict(st, strict=False)
        if missing.unexpected_keys:
            raise RuntimeError('prefixed keys')
    except Exception:
        network.load_state_dict({k[7:]: v for k, v in st.items()})
    network.eval()
    network.cuda()
    mano_path = {s: assets.write_synthetic_mano_pkl(str(tmp_path / ('MANO_%s.pkl' % s.upper())), s, seed=0) for s in ('left', 'right')}
    mano_layer = {s: ManoLayer(mano_path[s], center_idx=None) for s in ('left', 'right')}
    _fix_shape(mano_layer)
    J = {s: _Jr(mano_layer[s].J_regressor) for s in ('left', 'right')}
    J_cpu = {s: _Jr(mano_layer[s].J_regressor, device='cpu') for s in ('left', 'right')}

    # 16 synthetic crops + ground-truth meshes from the MANO layer (GPU kernel, the layer's own forward)
    g = torch.Generator().manual_seed(77)
    N, bs = 16, 2
    imgs = torch.randn(N, 3, 256, 256, generator=g)
    gt = {}
    for s in ('left', 'right'):
        layer = mano_layer[s].cuda()
        rootR = rodrigues_batch(torch.randn(N, 3, generator=g) * 0.5).cuda()
        v, _ = layer(rootR, (torch.randn(N, 45, generator=g) * 0.6).cuda(), torch.randn(N, 10, generator=g).cuda(),
                     trans=(torch.randn(N, 3, generator=g) * 0.05 + (0.08 if s == 'left' else -0.08)).cuda())
        gt[s] = v.detach().cpu()
    batches = [(imgs[i:i + bs], torch.zeros(bs, 21, 3), gt['left'][i:i + bs], torch.zeros(bs, 21, 3), gt['right'][i:i + bs])
               for i in range(0, N, bs)]

    graph = _graph()
    acc = {w: {s: [[], [], []] for s in ('left', 'right')} for w in ('hip', 'oracle')}
    with torch.no_grad():
        for data in batches:
            imgTensors = data[0].cuda()
            verts_left_gt, verts_right_gt = data[2].cuda(), data[4].cuda()
            result, paramsDict, handDictList, otherInfo = network(imgTensors)            # eval_interhand.py:311
            want = net_oracle.handnet_forward({k: v.clone() for k, v in state.items()}, graph, data[0], training=False)[0]
            for s in ('left', 'right'):
                testing.assert_close(result['verts3d'][s], want['verts3d'][s], 1e-4, 1e-5, 'verts3d ' + s)
            for who, res, vl, vr, JJ in (('hip', result, verts_left_gt, verts_right_gt, J),
                                         ('oracle', want, data[2], data[4], J_cpu)):
                mtr = _script_metrics(res, vl, vr, JJ)
                for s in ('left', 'right'):
                    for i in range(3):
                        acc[who][s][i].append(mtr[s][i].double().cpu().numpy())
    lines = {}
    for who in acc:
        for i, name in enumerate(('ori joint mpjpe', 'ori vert mean error', 'pa joint mean error')):
            d = {s: float(np.concatenate(acc[who][s][i], 0).mean() * 1000) for s in ('left', 'right')}
            lines[(who, name)] = (d['left'] + d['right']) / 2
    for name in ('ori joint mpjpe', 'ori vert mean error', 'pa joint mean error'):
        a, b = lines[('hip', name)], lines[('oracle', name)]
        print('configs[0] %-22s HIP %.4f mm   CPU oracle %.4f mm' % (name, a, b))
        assert abs(a - b) <= 1e-4 * abs(b) + 1e-4, (name, a, b)              # (north_star: MPJPE within 0.1 mm -- here within 1e-4 mm + 1e-4 rel)
    # the GPU-resident metrics loop on the same batches
    summary, _ = evaluate(network, batches, J['left'].J_regressor, J['right'].J_regressor, device='cuda')
    for name in ('ori joint mpjpe', 'ori vert mean error', 'pa joint mean error'):
        assert abs(summary[name]['all'] - lines[('hip', name)]) <= 1e-4 * abs(lines[('hip', name)]) + 1e-4, (name, summary[name])


# ------------------------------------------------------------------------------------------------ side streams under a gradient exchange
def test_hrnet_side_streams_survive_the_gradient_exchange():
    """Round-5 verdict item 3: TrainStep no longer switches the model's side streams off in a process that exchanges gradients
    (rounds 3-4 clamped them to 0 there, which cost HRNet-W32 its three side streams under data parallelism).  HRNet-W32 step
    captured with the RCCL bucket exchange on (world size 1: same code path) and three side streams against the same step with
    none: same live-gradient set, gradients equal to the bar of test_hrnet_side_streams_change_nothing (2e-5 of the tensor
    maximum: issue order = single-stream order), replays bit-identical, exposed-communication timer alive."""
    import subprocess
    root = os.path.dirname(HERE)
    code = r'''
import os, sys, torch
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'tests'))
import torch.distributed as dist
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29541')
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
from oracle.net_oracle import scalar_loss
from renderih_amd import testing, streams, ops
from renderih_amd.model import build_model
from renderih_amd.train import TrainStep
def build():
    m = build_model(0.0, 'hrnet32'); m.load_state_dict(testing.deterministic_state(m.state_dict(), seed=5)); m = m.cuda().train()
    m.decoder.unsample_layer.weight.requires_grad_(False); return m
img = testing.seeded_image(2, 17).cuda()
grads = {}
for side in (0, 3):
    streams.SIDE = side
    m = build()
    opt = torch.optim.SGD([p for p in m.parameters() if p.requires_grad], lr=0.0)
    step = TrainStep(m, opt, lambda out, lab: scalar_loss(out), (img.clone(), {}), force_exchange=True)
    assert step.exchange and step.overlap and step.use_graph and step.side_limit is None
    first = None
    for rep in range(3):
        loss = step(img, {})
        torch.cuda.synchronize()
        assert bool(torch.isfinite(loss))
        got = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
        if first is None:
            first = got
        else:
            for k in first:
                assert torch.equal(got[k], first[k]), 'side=%%d replay %%d differs (%%s)' %% (side, rep, k)
    assert step.comm_ms_exposed() is not None
    grads[side] = first
    ops.DROPOUT_SEED_TENSOR = None
    del step, m, opt
assert set(grads[0]) == set(grads[3])
worst = (0.0, '')
for k in grads[0]:
    err = float((grads[0][k] - grads[3][k]).abs().max()); mx = float(grads[0][k].abs().max())
    worst = max(worst, (err / max(mx, 1e-30), k))
    assert err <= 2e-5 * mx, (k, err, mx)
print('worst relative difference side 0 vs side 3: %%.3g (%%s)' %% worst)
dist.destroy_process_group()
print('HRNET-SIDE-EXCHANGE-OK')
''' % (root, root)
    p = subprocess.run([sys.executable, '-X', 'faulthandler', '-c', code], cwd=root, capture_output=True, text=True, timeout=1200)
    print(p.stdout[-1500:])
    assert p.returncode == 0 and 'HRNET-SIDE-EXCHANGE-OK' in p.stdout, (p.stdout + p.stderr)[-3000:]
