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


def _build(dropout=0.0, seed=0, encoder='resnet50'):
    from renderih_amd.model import build_model
    from renderih_amd import _lib
    _lib.load()
    m = build_model(dropout, encoder)
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


def _grad_report(named_grads, g32, g64, k=6.0, floor=2e-4, max_loose=0.03):
    """Per-tensor gradient check anchored on the fp64 oracle.  ReLU / max-pool decisions are discontinuous: an
    activation within round-off of zero can flip between two fp32 implementations and move a small decoder gradient
    by O(1/rows) (seen on CPU too: with train-mode BN at B=2..3 the fp32 oracle's trunk gradients are several % from
    its fp64 run).  So: all but a fraction `max_loose` of the tensors must be within k x (the fp32 reference's own error vs fp64) +
    floor, and every tensor within max(5%, 20 x that reference error) of max|ref|.  `max_loose` per caller follows what the GPU
    prints (profiles/r04/ab/c12_pytest_grad_reports.log, engine 2): ResNet50 model 0 of 823 tensors outside the 6x band (worst
    3.0x) -> 0.5 %; second family 2 of 843 (worst 17x, one layer4 convolution at B = 2) -> 1 %; HRNet-W32 31 / 14 of 1621 -> 3 %."""
    n, loose, gross, worst = 0, [], [], (0.0, '', 0.0, 0.0)
    for name, g in named_grads:
        if testing.is_null_gradient(name):
            continue            # exactly zero in exact arithmetic: round-off noise only
        ref64, ref32 = g64[name], g32[name]
        scale = float(ref64.abs().max().clamp_min(1e-30))
        e_got = float((g.detach().double().cpu() - ref64).abs().max()) / scale
        e_ref = float((ref32.double() - ref64).abs().max()) / scale
        n += 1
        worst = max(worst, (e_got / max(e_ref, floor), name, e_got, e_ref))
        if not (e_got <= k * e_ref + floor):
            loose.append('%s: %.3g (fp32 ref %.3g)' % (name, e_got, e_ref))
        if not (e_got <= max(0.05, 20 * e_ref)):
            gross.append('%s: %.3g' % (name, e_got))
    print('_grad_report: %d tensors, %d outside the %gx band; worst %s: %.3g vs fp32 ref %.3g (%.1fx)'
          % (n, len(loose), k, worst[1], worst[2], worst[3], worst[0]))
    for line in loose[:48]:                 # (every tensor outside the band is named in the log, not only the worst)
        print('   outside the %gx band: %s' % (k, line))
    if max_loose is None:           # a count only (the caller compares it with another implementation's count)
        return len(loose), n
    assert not gross, 'gradients grossly off:\n' + '\n'.join(gross[:20])
    assert len(loose) <= max_loose * n, '%d/%d gradient tensors outside the fp64-anchored band:\n%s' % (
        len(loose), n, '\n'.join(loose[:20]))
    return len(loose), n


def test_model_eval_matches_reference_golden():
    """Eval mode (BatchNorm on running statistics) is well conditioned: strict 1e-4 parity with the golden vectors
    the real reference produced."""
    z = np.load(os.path.join(GOLDEN, 'net_eval.npz'))
    m, _ = _build(0.0)
    m.eval()
    with torch.no_grad():
        out = m(testing.seeded_image(2, 0).cuda())
    for k, v in testing.flatten_outputs(out).items():
        _check_golden(z, 'out/' + k, v)


def test_model_train_matches_reference_golden():
    """Train mode, B=2: BatchNorm statistics over as few as 128 samples make the net ill conditioned (the CPU fp32
    oracle itself is ~2e-4 from fp64 and two fp32 implementations with different summation orders land up to
    ~3e-4 apart), so against the fp32 golden the bar is 2e-3 relative / 5e-4 of the tensor maximum here; the tight,
    fp64-anchored bar is in test_model_train_matches_fp64_oracle."""
    z = np.load(os.path.join(GOLDEN, 'net_train.npz'))
    m, _ = _build(0.0)
    m.train()
    out = m(testing.seeded_image(2, 0).cuda())
    for k, v in testing.flatten_outputs(out).items():
        _check_golden(z, 'out/' + k, v, 2e-3, 5e-4)
    from oracle.net_oracle import scalar_loss
    loss = scalar_loss(out)
    assert abs(loss.item() - float(z['loss'])) <= 1e-3 * abs(float(z['loss']))
    loss.backward()
    names = [str(n) for n in z['grad_names']]
    params = dict(m.named_parameters())
    got = {k for k, p in params.items() if p.grad is not None}
    assert set(names) == got, sorted(set(names) ^ got)[:10]     # the same 53 tensors stay grad-less
    sd = m.state_dict()
    for k in z.files:
        if k.startswith('bnstat/'):
            assert_close(sd[k[7:]].float(), torch.from_numpy(np.asarray(z[k])).float(), 1e-3, 1e-4, k)


@pytest.mark.parametrize('training', [False, True])
def test_model_matches_fp64_oracle(training):
    """Forward outputs and every parameter gradient vs the CPU oracle, anchored on its fp64 run (different weights
    and batch than the golden).  training=False: BN on running statistics but autograd on (all backward kernels,
    frozen-statistics BN backward); training=True: batch statistics."""
    from oracle import net_oracle
    m, sd = _build(0.0, seed=5)
    m.train(training)
    img = testing.seeded_image(3, 11)
    graph = net_oracle.graph_from_dicts(assets.load_graph_dict('left'), assets.load_graph_dict('right'))
    w32, g32 = net_oracle.run(sd, graph, img, training, torch.float32, True)
    w64, g64 = net_oracle.run(sd, graph, img, training, torch.float64, True)
    out = m(img.cuda())
    got = testing.flatten_outputs(out)
    report = {}
    for k in w64:
        report[k] = testing.assert_fp32_equivalent(got[k], w32[k], w64[k], k=4.0, floor=2e-5, what=k)
    net_oracle.scalar_loss(out).backward()
    params = [(k, p.grad) for k, p in m.named_parameters() if p.grad is not None]
    assert {k for k, _ in params} == set(g64.keys())
    nloose, n = _grad_report(params, g32, g64, max_loose=0.005)
    print('fp64-anchored: outputs worst %.3g (fp32 ref %.3g); grads %d/%d outside band'
          % (max(v[0] for v in report.values()), max(v[1] for v in report.values()), nloose, n))


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


# ------------------------------------------------------------------------------------------------ HRNet-W32 variant
def test_hrnet_eval_matches_reference_golden():
    """ENCODER_TYPE hrnet32 (BASELINE config 4), eval mode: strict 1e-4 against vectors the real reference produced."""
    z = np.load(os.path.join(GOLDEN, 'net_hrnet_eval.npz'))
    m, _ = _build(0.0, encoder='hrnet32')
    m.eval()
    with torch.no_grad():
        out = m(testing.seeded_image(2, 0).cuda())
    for k, v in testing.flatten_outputs(out).items():
        _check_golden(z, 'out/' + k, v)


@pytest.mark.parametrize('training', [False, True])
def test_hrnet_matches_fp64_oracle(training):
    """Forward outputs and every parameter gradient of the HRNet-W32 variant, anchored on the fp64 oracle.

    The gradient report's loose tensors at this B = 2 (31 of 1900 in train mode, inside the 3 % allowance) are NOT rounding of the
    engine-2 arithmetic: with RIH_GEMM_ENGINE=0 (native fp32 MFMA, no split at all) the SAME tensors are loose by the same factors
    (profiles/r05/ab/c2_pytest_hr_e0.log beside c1_pytest_grads.log).  They come in clusters -- all parameters upstream of one
    ReLU / stride decision that two fp32 evaluation orders take differently when BatchNorm statistics rest on 128 samples -- and
    the fp32 oracle itself lands on the other side of such decisions against its own fp64 run."""
    from oracle import net_oracle
    m, sd = _build(0.0, seed=7, encoder='hrnet32')
    m.train(training)
    img = testing.seeded_image(2, 13)
    graph = net_oracle.graph_from_dicts(assets.load_graph_dict('left'), assets.load_graph_dict('right'))
    w32, g32 = net_oracle.run(sd, graph, img, training, torch.float32, True)
    w64, g64 = net_oracle.run(sd, graph, img, training, torch.float64, True)
    out = m(img.cuda())
    got = testing.flatten_outputs(out)
    for k in w64:
        testing.assert_fp32_equivalent(got[k], w32[k], w64[k], k=4.0, floor=2e-5, what=k)
    net_oracle.scalar_loss(out).backward()
    params = [(k, p.grad) for k, p in m.named_parameters() if p.grad is not None]
    assert {k for k, _ in params} == set(g64.keys())
    _grad_report(params, g32, g64)
    if training:
        assert int(m.encoder.hrnet.bn1.num_batches_tracked) == 1


def test_captured_all_to_all_fork_join_with_three_side_streams():
    """Two fork_joins, the second reading every output of the first, forward + backward captured in one hipGraph on three side
    streams (tools/capture_fork_min.py a2a3): without the autograd hop of streams.fork_join the backward makes side streams wait
    on each other and hipStreamEndCapture overflows the stack (ROCm 7.0.2); with it the capture ends and replays."""
    import subprocess
    import sys
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools', 'capture_fork_min.py')
    env = dict(os.environ)
    env.pop('RIH_FORK_HOP', None)
    r = subprocess.run([sys.executable, '-X', 'faulthandler', tool, 'a2a3'], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    assert r.stdout.strip().splitlines()[-1].startswith('ok ')


def test_hrnet_side_streams_change_nothing(monkeypatch):
    """renderih_amd.streams.fork_join: the HRNet exchange units with their branches and fuse rows forked onto side streams give
    the same outputs and parameter gradients, bit for bit, as the single-stream order -- eagerly (twice, so that recycled
    allocator blocks are exercised) and replayed from TrainStep's hipGraph (parallel graph branches)."""
    from oracle.net_oracle import scalar_loss
    from renderih_amd import ops, streams
    from renderih_amd.train import TrainStep
    img = testing.seeded_image(2, 17).cuda()

    def run(side):
        monkeypatch.setattr(streams, 'SIDE', side)
        m, _ = _build(0.0, seed=5, encoder='hrnet32')
        m.train()
        outs = None
        for _ in range(2):
            m.zero_grad(set_to_none=True)
            for mod in m.modules():
                if isinstance(mod, torch.nn.BatchNorm2d):
                    mod.reset_running_stats()
            out = m(img)
            scalar_loss(out).backward()
            torch.cuda.synchronize()
            now = ({k: v.detach().clone() for k, v in testing.flatten_outputs(out).items()},
                   {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None})
            if outs is not None:
                for a, b in zip(outs, now):
                    for k in a:
                        assert torch.equal(a[k], b[k]), 'side=%d: second eager pass differs (%s)' % (side, k)
            outs = now
        return m, outs

    _, (o0, g0) = run(0)
    m3, (o3, g3) = run(streams.SIDE if streams.SIDE > 0 else 1)
    assert set(g0) == set(g3)
    for k in o0:
        assert torch.equal(o0[k], o3[k]), k
    for k in g0:
        # (issue order = single-stream order, so autograd sums the gradients of shared inputs in the same order; the bar below
        # is what a different summation order could cost, far under anything a race would leave)
        err = float((g0[k] - g3[k]).abs().max())
        assert err <= 2e-5 * float(g0[k].abs().max()), (k, err)
    # the same model under TrainStep's graph (weight gradients grouped: round-off differs from the eager order, replays agree)
    m3.zero_grad(set_to_none=True)
    opt = torch.optim.SGD([p for p in m3.parameters() if p.requires_grad], lr=0.0)
    try:
        step = TrainStep(m3, opt, lambda out, lab: scalar_loss(out), (img.clone(), {}), process_group=False, stages='auto')
        assert step.use_graph
        first = None
        for rep in range(3):
            step(img, {})
            torch.cuda.synchronize()
            got = {k: p.grad.clone() for k, p in m3.named_parameters() if p.grad is not None}
            assert set(got) == set(g0)
            if first is None:
                first = got
                for k in g0:
                    wk = k[:-len('bias')] + 'weight'
                    floor = 1e-5 * float(g0[wk].abs().max()) if (k.endswith('.bias') and wk in g0) else 0.0
                    err = float((got[k] - g0[k]).abs().max())
                    assert err <= 1e-4 * float(g0[k].abs().max()) + floor, (k, err)
            else:
                for k in g0:
                    assert torch.equal(got[k], first[k]), 'replay %d differs (%s)' % (rep, k)
    finally:
        ops.DROPOUT_SEED_TENSOR = None


# ------------------------------------------------------------------------------------------------ hipGraph capture
def _sgd_losses(m, img, steps, capture, dropout_seed=None, reducer=None):
    """`steps` plain-SGD steps on the scalar loss; forward+backward either launched eagerly or replayed from a hipGraph.
    `reducer`: a GradAllReducer run between backward and the optimizer step, as bench.py does for N > 1."""
    from oracle.net_oracle import scalar_loss
    from renderih_amd import ops
    params = [p for p in m.parameters() if p.requires_grad]
    opt = torch.optim.SGD(params, lr=1e-15)      # the seeded weights give gradients ~1e9: keep the walk tiny but non-zero
    ops.DROPOUT_SEED_TENSOR = dropout_seed
    losses = []
    try:
        def fwd_bwd():
            loss = scalar_loss(m(img))
            loss.backward()
            return loss
        if not capture:
            for _ in range(steps):
                opt.zero_grad(set_to_none=True)
                if dropout_seed is not None:
                    dropout_seed.add_(7)
                losses.append(float(fwd_bwd()))
                opt.step()
            return losses
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            opt.zero_grad(set_to_none=True)
            fwd_bwd()                                   # warm-up (no optimizer step: keep the weights)
        torch.cuda.current_stream().wait_stream(side)
        for mod in m.modules():                         # undo the warm-up's BatchNorm bookkeeping
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.reset_running_stats()
        opt.zero_grad(set_to_none=True)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            if dropout_seed is not None:
                dropout_seed.add_(7)
            static_loss = fwd_bwd()
        if reducer is not None:
            reducer.use_static_grads()
        for _ in range(steps):
            g.replay()
            losses.append(float(static_loss))
            if reducer is not None:
                reducer.reduce()
            opt.step()
        return losses
    finally:
        ops.DROPOUT_SEED_TENSOR = None


def test_hipgraph_training_step_matches_eager():
    """Forward + loss + backward captured in a hipGraph and replayed: same losses as launching every kernel from
    Python (dropout off: bit-identical kernels, so the trajectories agree to round-off of the loss reduction)."""
    img = testing.seeded_image(2, 21).cuda()
    m1, _ = _build(0.0, seed=9)
    m1.train()
    eager = _sgd_losses(m1, img, 3, capture=False)
    m2, _ = _build(0.0, seed=9)
    m2.train()
    graph = _sgd_losses(m2, img, 3, capture=True)
    for a, b in zip(eager, graph):
        assert abs(a - b) <= 1e-6 * abs(a), (eager, graph)
    assert eager[0] != eager[1]                     # the optimizer really moved the weights
    for (k, p), (_, q) in zip(m1.named_parameters(), m2.named_parameters()):
        assert torch.equal(p, q), k


def test_hipgraph_replay_with_gradient_reducer_tracks_eager():
    """bench.py's N > 1 step: graph replay -> GradAllReducer.reduce() (RCCL; world size 1 here, so the average is the
    local gradient) -> optimizer.  `.grad` is rebound to the reducer's bucket after the first step while the graph keeps
    writing its own buffers: the losses and the weights must still follow the eager trajectory step by step."""
    import torch.distributed as dist
    from renderih_amd.dp import GradAllReducer
    img = testing.seeded_image(2, 21).cuda()
    m1, _ = _build(0.0, seed=9)
    m1.train()
    eager = _sgd_losses(m1, img, 3, capture=False)
    m2, _ = _build(0.0, seed=9)
    m2.train()
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29533')
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    try:
        graph = _sgd_losses(m2, img, 3, capture=True, reducer=GradAllReducer(m2))
    finally:
        dist.destroy_process_group()
    for a, b in zip(eager, graph):
        assert abs(a - b) <= 1e-6 * abs(a), (eager, graph)
    assert eager[0] != eager[1] != eager[2]
    for (k, p), (_, q) in zip(m1.named_parameters(), m2.named_parameters()):
        assert torch.equal(p, q), k


def test_hipgraph_replay_draws_fresh_dropout_masks():
    """With the device-resident seed word advanced inside the graph, every replay uses new dropout masks, and the same
    word value reproduces the eager result."""
    img = testing.seeded_image(2, 22).cuda()
    m, _ = _build(0.3, seed=9)
    m.train()
    seed = torch.zeros(1, dtype=torch.int64, device='cuda')
    g_losses = _sgd_losses(m, img, 3, capture=True, dropout_seed=seed)
    assert len({round(v, 3) for v in g_losses}) == 3, g_losses          # three different masks
    m2, _ = _build(0.3, seed=9)
    m2.train()
    seed2 = torch.zeros(1, dtype=torch.int64, device='cuda')            # capture records the advance, replays run it
    e_losses = _sgd_losses(m2, img, 3, capture=False, dropout_seed=seed2)
    for a, b in zip(e_losses, g_losses):
        assert abs(a - b) <= 1e-5 * abs(a), (e_losses, g_losses)


def test_paired_decoder_equals_sequential_on_gpu(monkeypatch):
    """ops.PAIR_HANDS (one launch per left/right layer pair) against the per-hand launch sequence, same weights and
    image: eval-mode outputs at the fp32 bar.  (The paired path is the default, so its gradients are what
    test_model_matches_fp64_oracle / the train golden check; the sequential path keeps this comparison.)"""
    from renderih_amd import ops
    m, _ = _build(0.0, seed=2)
    img = testing.seeded_image(2, 7).to('cuda:0')
    m.eval()
    outs = {}
    for pair in (True, False):
        monkeypatch.setattr(ops, 'PAIR_HANDS', pair)
        with torch.no_grad():
            outs[pair] = {k: v.cpu() for k, v in testing.flatten_outputs(m(img)).items()}
    for k in outs[False]:
        assert_close(outs[True][k], outs[False][k], 1e-4, 1e-5, 'paired vs sequential ' + k)


def test_decoder_side_stream_changes_nothing(monkeypatch):
    """attn.DECODER_FORK: the decoder's image-grid encoders in a side stream beside the graph convolutions.  Same issue order, so
    the dropout seeds and autograd's summation order are the single-stream ones: training-mode outputs and every parameter
    gradient equal bit for bit, with dropout on."""
    from oracle.net_oracle import scalar_loss
    from renderih_amd import attn, ops, streams
    monkeypatch.setattr(streams, 'SIDE', max(streams.SIDE, 1))
    img = testing.seeded_image(2, 9).to('cuda:0')
    res = {}
    for fork in (False, True):
        monkeypatch.setattr(attn, 'DECODER_FORK', fork)
        m, _ = _build(0.05, seed=3)
        m.train()
        ops.DROPOUT_SEED_TENSOR = None
        torch.manual_seed(1234)
        for _ in range(2):                      # twice: the second pass runs on recycled allocator blocks
            m.zero_grad(set_to_none=True)
            out = m(img)
            scalar_loss(out).backward()
        torch.cuda.synchronize()
        res[fork] = ({k: v.detach().clone() for k, v in testing.flatten_outputs(out).items()},
                     {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None})
    for k in res[False][0]:
        assert torch.equal(res[False][0][k], res[True][0][k]), k
    assert set(res[False][1]) == set(res[True][1])
    for k in res[False][1]:
        assert torch.equal(res[False][1][k], res[True][1][k]), k


def _dead_mid_check(device, monkeypatch):
    """model.SKIP_DEAD_MID (the default since round 4): the finest mid convolution, whose output `decoder.forward` drops
    (models/decoder.py:130), reduced to what it still owes -- its BatchNorm's running statistics in training mode, nothing in
    eval mode.  Outputs, parameter gradients and EVERY buffer of the state_dict equal the full computation bit for bit."""
    from oracle.net_oracle import scalar_loss
    from renderih_amd import model as model_mod
    from renderih_amd.model import build_model
    img = testing.seeded_image(2, 21).to(device)
    res = {}
    for skip in (False, True):
        monkeypatch.setattr(model_mod, 'SKIP_DEAD_MID', skip)
        m = build_model(0.0)
        m.load_state_dict(testing.deterministic_state(m.state_dict(), seed=8))
        m = m.to(device)
        m.train()
        out = m(img)
        scalar_loss(out).backward()
        train_out = {k: v.detach().clone() for k, v in testing.flatten_outputs(out).items()}
        grads = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
        state = {k: v.detach().clone() for k, v in m.state_dict().items()}
        m.eval()
        with torch.no_grad():
            eval_out = {k: v.clone() for k, v in testing.flatten_outputs(m(img)).items()}
        res[skip] = (train_out, grads, state, eval_out)
    full, lean = res[False], res[True]
    assert int(full[2]['mid_model.convs.3.2.num_batches_tracked']) == 1
    assert not torch.equal(full[2]['mid_model.convs.3.2.running_mean'], torch.zeros_like(full[2]['mid_model.convs.3.2.running_mean']))
    for a, b, what in zip(full, lean, ('train output', 'gradient', 'state_dict entry', 'eval output')):
        assert set(a) == set(b), what
        for k in a:
            assert torch.equal(a[k], b[k]), '%s %s differs with the dead mid convolution skipped' % (what, k)


def test_dead_mid_convolution_skip_changes_nothing(monkeypatch):
    _dead_mid_check('cuda:0', monkeypatch)


# ------------------------------------------------------------------------------ second model family
def _build_b(dropout=0.0, seed=4):
    from renderih_amd import _lib
    from renderih_amd.lijun import build_graph_model
    _lib.load()
    m = build_graph_model(dropout)
    sd = testing.deterministic_state(m.state_dict(), seed=seed)
    m.load_state_dict(sd)
    return m.to('cuda:0'), sd


def test_family_b_eval_matches_reference_golden():
    """common/myhand/lijun_model_graph.HandNET_GCN, eval mode: strict 1e-4 parity with vectors the real reference
    modules produced (tests/golden/make_golden.py lijun)."""
    z = np.load(os.path.join(GOLDEN, 'net_lijun_eval.npz'))
    m, _ = _build_b(0.0)
    m.eval()
    with torch.no_grad():
        out = m(testing.seeded_image(2, 5).cuda())
    assert out[3]['verts3d_MANO_list'] == {'left': [], 'right': []} and 'hms' not in out[3]
    flat = testing.flatten_outputs(out)
    assert {('out/' + k) for k in flat} == {k.split('#')[0] for k in z.files if k.startswith('out/')}
    for k, v in flat.items():
        _check_golden(z, 'out/' + k, v)


def test_family_b_train_matches_reference_golden():
    """Train mode, B=2 (ill conditioned, see test_model_train_matches_reference_golden): forward, loss, the set of
    parameters that receive gradients, BatchNorm running statistics."""
    z = np.load(os.path.join(GOLDEN, 'net_lijun_train.npz'))
    m, _ = _build_b(0.0)
    m.train()
    out = m(testing.seeded_image(2, 5).cuda())
    for k, v in testing.flatten_outputs(out).items():
        _check_golden(z, 'out/' + k, v, 2e-3, 1e-3)
    from oracle.net_oracle import scalar_loss
    loss = scalar_loss(out)
    assert abs(loss.item() - float(z['loss'])) <= 1e-3 * abs(float(z['loss']))
    loss.backward()
    names = [str(n) for n in z['grad_names']]
    got = {k for k, p in m.named_parameters() if p.grad is not None}
    assert set(names) == got, sorted(set(names) ^ got)[:10]
    sd = m.state_dict()
    for k in z.files:
        if k.startswith('bnstat/'):
            assert_close(sd[k[7:]].float(), torch.from_numpy(np.asarray(z[k])).float(), 1e-3, 1e-4, k)


@pytest.mark.parametrize('training', [False, True])
def test_family_b_matches_fp64_oracle(training):
    """Forward outputs and every parameter gradient of the second family vs the CPU oracle, anchored on its fp64 run."""
    from oracle import net_oracle
    m, sd = _build_b(0.0, seed=11)
    m.train(training)
    img = testing.seeded_image(2, 12)
    graph = net_oracle.graph_from_dicts(assets.load_graph_dict('left'), assets.load_graph_dict('right'))
    w32, g32 = net_oracle.run(sd, graph, img, training, torch.float32, True)
    w64, g64 = net_oracle.run(sd, graph, img, training, torch.float64, True)
    out = m(img.cuda())
    got = testing.flatten_outputs(out)
    assert set(got) == set(w64)
    for k in w64:
        testing.assert_fp32_equivalent(got[k], w32[k], w64[k], k=4.0, floor=2e-5, what=k)
    net_oracle.scalar_loss(out).backward()
    params = [(k, p.grad) for k, p in m.named_parameters() if p.grad is not None]
    assert {k for k, _ in params} == set(g64.keys())
    _grad_report(params, g32, g64, max_loose=0.01)


# ------------------------------------------------------------------------------------------------ TrainStep helper
@pytest.mark.parametrize('group,stages', [(0, True), (1, True), (2, True), (0, 'auto'), (2, 'auto')])
def test_train_step_staged_graph_replay_matches_plain_backward(group, stages, monkeypatch):
    """renderih_amd.train.TrainStep (three backward stages, three hipGraphs sharing a pool, fused optimizer outside -- or, with
    stages='auto' and no gradient exchange, ONE stage and one graph: bench.py's single-rank configuration): the
    gradients it leaves in `.grad` equal those of a plain eager `loss.backward()` on the same state, for the first and for a
    replayed step, and the set of grad-less parameters is unchanged.  group 0 (weight-gradient GEMMs launched one by
    one, same split-K plan): bit for bit.  group 1 / 2 (ops.GROUP_WGRAD: the stage's weight gradients in grouped launches with
    fewer split-K slices, the launch tables re-copied from pinned memory by the graphs): equal to split-K round-off, and every
    replay bit-identical to the first."""
    from oracle.net_oracle import scalar_loss
    from renderih_amd import ops
    from renderih_amd.train import TrainStep
    monkeypatch.setattr(ops, 'GROUP_WGRAD', group)
    img = testing.seeded_image(2, 31).cuda()
    m1, _ = _build(0.0, seed=11)
    m1.train()
    m1.decoder.unsample_layer.weight.requires_grad_(False)
    scalar_loss(m1(img)).backward()
    want = {k: p.grad.clone() for k, p in m1.named_parameters() if p.grad is not None}

    m2, _ = _build(0.0, seed=11)
    m2.train()
    m2.decoder.unsample_layer.weight.requires_grad_(False)
    opt = torch.optim.SGD([p for p in m2.parameters() if p.requires_grad], lr=0.0)      # lr 0: the state stays put
    try:
        step = TrainStep(m2, opt, lambda out, lab: scalar_loss(out), (img.clone(), {}), process_group=False, stages=stages)
        assert step.use_graph and step.nstage == (3 if stages is True else 1) and step.defer_reduce and step.packs is not None
        first = None
        for rep in range(3):
            loss = step(img, {})
            assert bool(torch.isfinite(loss))
            got = {k: p.grad for k, p in m2.named_parameters() if p.grad is not None}
            assert set(got) == set(want)
            for k in want:
                if group == 0:
                    assert torch.equal(got[k], want[k]), 'replay %d: gradient of %s differs from the plain backward' % (rep, k)
                else:
                    # (a bias gradient is a column sum of the dy that also forms the weight gradient; where it is mathematically
                    # zero -- the key bias of a softmax attention -- it is pure round-off of that sum, so its error is measured
                    # against the weight gradient's magnitude as well)
                    wk = k[:-len('bias')] + 'weight'
                    floor = 1e-5 * float(want[wk].abs().max()) if (k.endswith('.bias') and wk in want) else 0.0
                    err = float((got[k] - want[k]).abs().max())
                    assert err <= 1e-4 * float(want[k].abs().max()) + floor, \
                        'replay %d: gradient of %s: max err %.3g (max |want| %.3g)' % (rep, k, err, float(want[k].abs().max()))
            if first is None:
                first = {k: v.clone() for k, v in got.items()}
            else:
                for k in want:
                    assert torch.equal(got[k], first[k]), 'replay %d is not bit-identical to the first (%s)' % (rep, k)
            if rep == 1:
                m2.zero_grad(set_to_none=True)      # the reference loop's optimizer.zero_grad(): `.grad` is re-bound next step
    finally:
        ops.DROPOUT_SEED_TENSOR = None
    # the module is unchanged for ordinary use: the trunk hook is inert outside the helper
    m2.zero_grad(set_to_none=True)
    scalar_loss(m2(img)).backward()
    assert torch.equal(m2.encoder.resnet.conv1.weight.grad, want['encoder.resnet.conv1.weight'])


def test_train_step_over_rccl_single_rank_reports_exposed_comm():
    """The N > 1 code path of TrainStep with RCCL at world size 1 (bucket copies, side-stream all-reduce, `.grad` views into
    the buckets, exposed-communication timing); the world-2 schedule itself is pinned on CPU by tests/test_train_step.py."""
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
    code = r'''
import os, sys, torch
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'tests'))
import torch.distributed as dist
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
from oracle.net_oracle import scalar_loss
from renderih_amd import testing
from renderih_amd.model import build_model
from renderih_amd.train import TrainStep
def build():
    m = build_model(0.0); m.load_state_dict(testing.deterministic_state(m.state_dict(), seed=11)); m = m.cuda().train()
    m.decoder.unsample_layer.weight.requires_grad_(False); return m
img = testing.seeded_image(2, 31).cuda()
m1 = build(); scalar_loss(m1(img)).backward()
want = {k: p.grad.clone() for k, p in m1.named_parameters() if p.grad is not None}
m2 = build()
opt = torch.optim.SGD([p for p in m2.parameters() if p.requires_grad], lr=0.0)
step = TrainStep(m2, opt, lambda out, lab: scalar_loss(out), (img.clone(), {}), force_exchange=True)
assert step.exchange and step.overlap
for rep in range(3):
    step(img, {})
got = {k: p.grad for k, p in m2.named_parameters() if p.grad is not None}
assert set(got) == set(want)
for k in want:
    assert torch.equal(got[k], want[k]), k
print('buckets', step.bucket_bytes(), 'exposed', step.comm_ms_exposed())
dist.destroy_process_group()
print('TRAINSTEP-RCCL-OK')
''' % (root, root)
    # (bitwise comparison with the plain backward: the grouped weight-gradient launches, which re-plan split-K, stay off here)
    p = subprocess.run([sys.executable, '-c', code], cwd=root, capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, RIH_WGRAD_GROUP='0'))
    assert p.returncode == 0 and 'TRAINSTEP-RCCL-OK' in p.stdout, (p.stdout + p.stderr)[-3000:]
