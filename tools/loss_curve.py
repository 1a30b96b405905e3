#!/usr/bin/env python
"""Training-loss trajectory of the two rih_gemm engines on the same data/weights (dropout off, so the only difference
is the MFMA engine): `python tools/loss_curve.py` runs itself once per engine and prints both curves side by side.
Two optimisers: plain SGD with a small step (trajectories must agree closely: the engines differ by fp32 rounding only)
and the reference's Adam(3e-4), whose first steps are sign-like (g / |g|) and amplify rounding-level differences of
near-zero gradients chaotically -- there the curves only have to stay statistically alike."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(steps, batch, optim):
    import torch
    from bench import synth_batch
    from renderih_amd import assets
    from renderih_amd.model import build_model
    from renderih_amd.loss import GraphLoss, calc_loss_GCN
    from renderih_amd.manolayer import ManoLayer
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    model = build_model(dropout=0.0).to(dev).train()
    model.decoder.unsample_layer.weight.requires_grad_(False)
    params = [p for p in model.parameters() if p.requires_grad]
    opt = (torch.optim.Adam(params, lr=3e-4, weight_decay=1e-2) if optim == 'adam' else torch.optim.SGD(params, lr=2e-12))
    mano = {s: ManoLayer(assets.synthetic_mano_dict(s)) for s in ('left', 'right')}
    gl = {s: GraphLoss(mano[s].J_regressor, mano[s].get_faces(), level=4, device=dev) for s in ('left', 'right')}
    conv = model.decoder.converter
    img, lab = synth_batch(batch, dev, seed=0)
    out = []
    for _ in range(steps):
        o = model(img)
        loss, _ = calc_loss_GCN(None, 0, gl['left'], gl['right'], conv['left'], conv['right'], *o,
                                lab['v2d_l'], lab['v2d_r'], lab['v3d_l'], lab['v3d_r'], lab['root_rel'], 256)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        out.append(float(loss.item()))
    print(' '.join('%.6f' % v for v in out))


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'child':
        run(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4])
        sys.exit(0)
    steps, batch = 10, 16
    for optim in ('sgd', 'adam'):
        curves = {}
        for e in ('0', '1'):
            env = dict(os.environ, RIH_GEMM_ENGINE=e)
            r = subprocess.run([sys.executable, __file__, 'child', str(steps), str(batch), optim], env=env,
                               capture_output=True, text=True)
            line = [l for l in r.stdout.strip().split('\n') if l and l[0].isdigit()]
            if not line:
                print('engine', e, 'failed:', r.stderr[-2000:])
                sys.exit(1)
            curves[e] = [float(v) for v in line[-1].split()]
        print('%s: step  engine0(f32 MFMA)  engine1(bf16x3 split)  rel.diff' % optim)
        for i, (a, b) in enumerate(zip(curves['0'], curves['1'])):
            print('%3d  %16.6f  %16.6f  %.2e' % (i, a, b, abs(a - b) / max(abs(a), 1e-12)))
