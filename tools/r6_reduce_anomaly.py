#!/usr/bin/env python
"""Round 6: the torch reduction that returned a wrong value inside a captured graph (round 5, DESIGN 4: `other['hms'].pow(2).sum()`
= 0.0046 instead of 395.7 on some replays while every output and gradient of the replay was bit-identical to the eager run).
Reduced step by step -- every case captures a graph, replays it eight times on identical data and prints the scalar per replay:

  torch-only      x.pow(2).sum() of a [2, 42, 64, 64] tensor produced by a torch kernel inside the same graph
  torch-two       two such reductions in one graph (second scalar allocated after the first is freed: pool reuse)
  model-tail      the package's eval forward captured, the reduction of `hms` as the LAST node
  model-head      the same reduction issued right behind the kernel that produces `hms`, the rest of the forward after it
  model-clone     the reduction of a CLONE of hms (another allocation), last node
  model-2pass     sum over dim 1 first, then the rest (no multi-block global reduction: torch's semaphore path is not taken)
  model-keep      as model-tail, but every intermediate of the reduction (the squares) is kept alive until after the capture

Usage: python tools/r6_reduce_anomaly.py [case ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

REPS = 8


def run_case(name, build):
    torch.cuda.synchronize()
    cap = torch.cuda.Stream()
    cap.wait_stream(torch.cuda.current_stream())
    keep = []
    with torch.cuda.stream(cap):
        for _ in range(2):
            out = build(keep)
        torch.cuda.synchronize()
        del keep[:]
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=cap, capture_error_mode='thread_local'):
            out = build(keep)
    torch.cuda.current_stream().wait_stream(cap)
    vals = []
    for _ in range(REPS):
        g.replay()
        torch.cuda.synchronize()
        vals.append([float(t) for t in out['scalars']])
    with torch.no_grad():
        eager = [float(f()) for f in out['eager']]
    ok = all(v == vals[0] for v in vals) and all(abs(a - b) <= 1e-5 * abs(b) for a, b in zip(vals[0], eager))
    print('%-12s %s  replays: %s | eager from the replay\'s tensors: %s' % (
        name, 'ok   ' if ok else 'WRONG', ' '.join('/'.join('%.6g' % x for x in v) for v in vals), '/'.join('%.6g' % x for x in eager)),
        flush=True)
    return ok


def main():
    torch.cuda.set_device(0)
    from renderih_amd import testing
    from renderih_amd.model import build_model
    cases = sys.argv[1:] or ['torch-only', 'torch-two', 'model-tail', 'model-head', 'model-clone', 'model-2pass', 'model-keep']
    x0 = torch.randn(2, 42, 64, 64, device='cuda')
    m = build_model(0.0)
    m.load_state_dict(testing.deterministic_state(m.state_dict(), seed=5))
    m = m.cuda().eval()
    img = testing.seeded_image(2, 17).cuda()

    def torch_only(keep):
        x = x0 * 1.5 + 0.25
        s = 1e-3 * x.pow(2).sum()
        return {'scalars': [s], 'eager': [lambda: 1e-3 * x.pow(2).sum()]}

    def torch_two(keep):
        x = x0 * 1.5 + 0.25
        a = 1e-3 * x.pow(2).sum()
        y = x0 * 0.5 - 1.0
        b = 1e-3 * y.pow(2).sum()
        return {'scalars': [a, b], 'eager': [lambda: 1e-3 * x.pow(2).sum(), lambda: 1e-3 * y.pow(2).sum()]}

    def model_case(kind):
        def build(keep):
            with torch.no_grad():
                if kind == 'head':
                    # the encoder alone first (it produces hms), the reduction, then the whole forward again (same values)
                    hms = m.encoder(img)[0]
                    s = 1e-3 * hms.pow(2).sum()
                    out = m(img)
                else:
                    out = m(img)
                    hms = out[3]['hms']
                    if kind == 'clone':
                        hms = hms.clone()
                    if kind == '2pass':
                        s = 1e-3 * hms.pow(2).sum(1).sum()
                    elif kind == 'keep':
                        sq = hms.pow(2)
                        keep.append(sq)
                        s = 1e-3 * sq.sum()
                    else:
                        s = 1e-3 * hms.pow(2).sum()
            keep.append(out)
            return {'scalars': [s], 'eager': [lambda: 1e-3 * hms.pow(2).sum()]}
        return build
    table = {'torch-only': torch_only, 'torch-two': torch_two, 'model-tail': model_case('tail'), 'model-head': model_case('head'),
             'model-clone': model_case('clone'), 'model-2pass': model_case('2pass'), 'model-keep': model_case('keep')}
    print('device', torch.cuda.get_device_name(0), 'torch', torch.__version__, 'hip', torch.version.hip)
    bad = [c for c in cases if not run_case(c, table[c])]
    print('cases with a wrong or unstable scalar:', bad or 'none')


if __name__ == '__main__':
    main()
