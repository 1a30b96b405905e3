#!/usr/bin/env python
"""Round 6, round-5 verdict item 6 ("BatchNorm apply inside a consumer: measure the one pass removal never A/B-ed"): an UPPER BOUND by
deletion.  Folding bn1 / bn2's scale + shift + ReLU into the consumer convolution's A conversion would remove their `rih_bn_apply`
launches from the forward pass at the price of extra VALU work in three kernels (halo / rows / panel forward, the weight-gradient loader)
and of raw convolution outputs saved for the backward.  Before building that: what does the step gain if those launches simply VANISH
(results are garbage -- this is a timing experiment, the switch lives in this tool only)?  The same captured training step as bench.py,
measured with (a) nothing removed, (b) every `rih_bn_apply` call with ReLU and without a residual operand removed (bn1 / bn2 of every
Bottleneck, the Conv-ReLU-BN layers do not match), interleaved.
    python tools/r6_bn_apply_upper_bound.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from renderih_amd import _lib, ops  # noqa: E402


class Proxy:
    """The loaded library with rih_bn_apply replaced by a filter."""
    def __init__(self, lib):
        self._lib, self.skip, self.skipped, self.kept = lib, False, 0, 0
        self.measure = None         # a list: (rows, C, relu, has_residual, start event, end event) per rih_bn_apply call

    def __getattr__(self, name):
        return getattr(self._lib, name)

    def rih_bn_apply(self, x, mean, invstd, gamma, beta, residual, y, rows, C, relu, mask, amax, stream):
        if self.skip and relu and not residual:
            self.skipped += 1
            return 0
        self.kept += 1
        if self.measure is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = self._lib.rih_bn_apply(x, mean, invstd, gamma, beta, residual, y, rows, C, relu, mask, amax, stream)
            e1.record()
            self.measure.append((rows, C, bool(relu), bool(residual), e0, e1))
            return rc
        return self._lib.rih_bn_apply(x, mean, invstd, gamma, beta, residual, y, rows, C, relu, mask, amax, stream)


def build_step(proxy, skip, graph=True):
    """bench.py's step: model, Adam, fused mesh loss, TrainStep (one stage: no gradient exchange), B = 64."""
    import bench
    from renderih_amd import assets, optim as rih_optim
    from renderih_amd.loss import GraphLoss, FusedMeshLoss, calc_loss_GCN_fused
    from renderih_amd.manolayer import ManoLayer
    from renderih_amd.model import build_model
    from renderih_amd.train import TrainStep
    proxy.skip, proxy.skipped, proxy.kept = skip, 0, 0
    torch.manual_seed(0)
    dev = torch.device('cuda:0')
    m = build_model(0.05).to(dev).train()
    m.decoder.unsample_layer.weight.requires_grad_(False)
    img, lab = bench.synth_batch(64, dev, 0)
    opt = rih_optim.Adam([p for p in m.parameters() if p.requires_grad], lr=3e-4, weight_decay=1e-2)
    mano = {s: ManoLayer(assets.synthetic_mano_dict(s)) for s in ('left', 'right')}
    gl = {s: GraphLoss(mano[s].J_regressor, mano[s].get_faces(), level=4, device=dev) for s in ('left', 'right')}
    cv = m.decoder.converter
    fused = FusedMeshLoss(gl['left'], gl['right'], cv['left'], cv['right'])

    def loss_fn(out, labels):
        return calc_loss_GCN_fused(fused, None, *out, labels['v2d_l'], labels['v2d_r'], labels['v3d_l'], labels['v3d_r'],
                                   labels['root_rel'])[0]
    step = TrainStep(m, opt, loss_fn, (img, lab), process_group=False, stages='auto', use_graph=graph)
    per_step = (proxy.skipped + proxy.kept) // 3        # two warm-up steps + the capture
    return step, img, lab, proxy.skipped // 3, per_step


def timed(step, img, lab, n=20):
    for _ in range(5):
        step(img, lab)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(n):
        step(img, lab)
    torch.cuda.synchronize()
    return (time.time() - t0) / n * 1e3


def main():
    lib = _lib.load()
    proxy = Proxy(lib)
    _lib._lib = proxy
    res = {}
    for rep in range(2):
        for skip in (False, True):
            step, img, lab, nskip, ncall = build_step(proxy, skip)
            ms = timed(step, img, lab)
            res.setdefault(skip, []).append(ms)
            print('rih_bn_apply launches removed per step: %3d of %3d -> %.3f ms per step' % (nskip, ncall, ms), flush=True)
            del step
            torch.cuda.empty_cache()
    a, b = min(res[False]), min(res[True])
    print('deletion experiment (the data behind the deleted launches is garbage -- NaN / unnormalised operands change the power the '
          'matrix pipes draw and with it the clock: NOT a clean bound): %.3f -> %.3f ms, %.2f %%' % (a, b, 100.0 * (a - b) / a))
    # the clean number: the durations of exactly those launches inside a valid eager step (HIP events around each call)
    step, img, lab, _, _ = build_step(proxy, False, graph=False)
    step(img, lab)
    step(img, lab)
    proxy.measure = []
    step(img, lab)
    torch.cuda.synchronize()
    recs, proxy.measure = proxy.measure, None
    pairs = []
    for _ in range(64):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        e1.record()
        pairs.append((e0, e1))
    torch.cuda.synchronize()
    empty = sorted(x.elapsed_time(y) for x, y in pairs)[32]
    tot = {True: [0, 0.0, 0.0], False: [0, 0.0, 0.0]}
    for rows, C, relu, res_, e0, e1 in recs:
        k = relu and not res_
        t = max(e0.elapsed_time(e1) - empty, 0.0)
        tot[k][0] += 1
        tot[k][1] += t
        tot[k][2] += rows * C * (8.25 if not res_ else 12.25)
        if k:
            print('   removable: rows %7d C %4d  %6.1f us  %5.0f GB/s' % (rows, C, 1e3 * t, rows * C * 8.25 / max(t, 1e-9) / 1e6))
    print('rih_bn_apply inside a valid eager step (event pair cost %.1f us taken off): removable (ReLU, no residual) %d launches %.3f ms '
          '(%.2f GB); the others %d launches %.3f ms' % (1e3 * empty, tot[True][0], tot[True][1], tot[True][2] / 1e9, tot[False][0], tot[False][1]))
    print('=> upper bound of the fold on the forward pass: %.3f ms of a %.2f ms step = %.2f %%' % (tot[True][1], a, 100.0 * tot[True][1] / a))


if __name__ == '__main__':
    main()
