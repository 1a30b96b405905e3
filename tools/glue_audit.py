#!/usr/bin/env python
"""Which torch-native device kernels does one training step launch, and from where?  Runs forward + fused loss + backward of the
configs[1] model on the numpy ABI emulator (CPU, B = 2) under a TorchDispatchMode and lists every aten op that is a kernel launch
on the GPU (add / copy_ / fill_ / zeros / cat / mul ...), with element counts and the innermost call site inside renderih_amd.
The trace of the real step (tools/step_from_trace.py --torch) says how much they cost; this says who issues them.
    python tools/glue_audit.py [--family b] [--top 40]"""
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

VIEW = ('view', 'reshape', 'as_strided', 'transpose', 'permute', 'slice', 'select', 'expand', 'unsqueeze', 'squeeze', 't.default',
        'detach', 'alias', 'unbind', 'split', 'narrow', 'empty', 'size', 'stride', '_unsafe_view', 'unflatten', 'flatten', 'chunk',
        'result_type', 'is_', 'set_', 'resize_', '_local_scalar', 'item', 'lift_fresh', 'contiguous', '_to_copy', 'clone')


class Audit(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.rows = collections.Counter()
        self.phase = 'fwd'

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        if any(v in name for v in VIEW) and not ('clone' in name or '_to_copy' in name or 'contiguous' in name):
            return out
        n = 0
        for a in list(args) + [out]:
            if isinstance(a, torch.Tensor):
                n = max(n, a.numel())
        if n < 2:
            return out
        site = '?'
        st = traceback.extract_stack()[:-1]
        if any('abi_emulator' in fr.filename or os.sep + 'oracle' + os.sep in fr.filename for fr in st):
            return out            # the emulator's own arithmetic: a HIP kernel on the GPU
        for fr in reversed(st):
            if 'renderih_amd' in fr.filename or fr.filename.endswith('bench.py'):
                site = '%s:%d %s' % (os.path.basename(fr.filename), fr.lineno, fr.name)
                break
        if site == '?':
            site = '(autograd engine) ' + 'x'.join(str(d) for d in out.shape) if isinstance(out, torch.Tensor) else site
        self.rows[(self.phase, name.replace('aten.', ''), site)] += 1
        return out


def main():
    from abi_emulator import emulated_abi
    from renderih_amd import testing, ops
    from renderih_amd.model import build_model
    top = int(sys.argv[sys.argv.index('--top') + 1]) if '--top' in sys.argv else 60
    from renderih_amd import assets
    from renderih_amd.loss import GraphLoss, FusedMeshLoss, calc_loss_GCN_fused
    from renderih_amd.manolayer import ManoLayer
    import bench
    dev = torch.device('cpu')
    with emulated_abi():
        m = build_model(0.05)
        m.load_state_dict(testing.deterministic_state(m.state_dict(), seed=3))
        m.train()
        m.decoder.unsample_layer.weight.requires_grad_(False)
        mano = {s: ManoLayer(assets.synthetic_mano_dict(s)) for s in ('left', 'right')}
        gl = {s: GraphLoss(mano[s].J_regressor, mano[s].get_faces(), level=4, device=dev) for s in ('left', 'right')}
        conv = m.decoder.converter
        fused = FusedMeshLoss(gl['left'], gl['right'], conv['left'], conv['right'])
        img, lab = bench.synth_batch(2, dev, seed=0)
        au = Audit()
        with au:
            with ops.owned_bounds():
                out = m(img)
                au.phase = 'loss'
                loss = calc_loss_GCN_fused(fused, None, *out, lab['v2d_l'], lab['v2d_r'], lab['v3d_l'], lab['v3d_r'], lab['root_rel'])[0]
                au.phase = 'bwd'
                loss.backward()
    tot = collections.Counter()
    for (ph, name, site), c in au.rows.items():
        tot[ph] += c
    print('aten kernel-launching ops per step: ' + ', '.join('%s %d' % kv for kv in sorted(tot.items())))
    for (ph, name, site), c in au.rows.most_common(top):
        print('%4d  %-4s %-28s %s' % (c, ph, name, site))


if __name__ == '__main__':
    main()
