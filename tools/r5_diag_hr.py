#!/usr/bin/env python
"""Diagnostic (round 5): are the replays of a captured HRNet-W32 step bit-identical?  tests/test_gpu_round5.py::
test_hrnet_side_streams_survive_the_gradient_exchange saw replay 1 differ from replay 0 with three side streams and the RCCL
exchange on.  Runs one configuration per process invocation: --exchange 0/1, --side N, --group G (ops.GROUP_WGRAD), --reps R;
prints, per replay pair, how many gradient tensors differ, the largest relative difference and the first names."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--exchange', type=int, default=1)
    ap.add_argument('--side', type=int, default=3)
    ap.add_argument('--group', type=int, default=2)
    ap.add_argument('--reps', type=int, default=6)
    ap.add_argument('--encoder', default='hrnet32')
    a = ap.parse_args()
    torch.cuda.set_device(0)
    if a.exchange:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29543')
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    from oracle.net_oracle import scalar_loss
    from renderih_amd import testing, streams, ops
    from renderih_amd.model import build_model
    from renderih_amd.train import TrainStep
    streams.SIDE = a.side
    ops.GROUP_WGRAD = a.group
    m = build_model(0.0, a.encoder)
    m.load_state_dict(testing.deterministic_state(m.state_dict(), seed=5))
    m = m.cuda().train()
    m.decoder.unsample_layer.weight.requires_grad_(False)
    img = testing.seeded_image(2, 17).cuda()
    opt = torch.optim.SGD([p for p in m.parameters() if p.requires_grad], lr=0.0)
    holder = {}

    def terms_of(out):
        """The addends of oracle.net_oracle.scalar_loss, one tensor each (same order)."""
        result, params, hd, other = out
        t = []
        for side in ('left', 'right'):
            t += [('v3d.' + side, result['verts3d'][side].abs().sum()), ('v2d.' + side, 1e-2 * result['verts2d'][side].abs().sum()),
                  ('c3d.' + side, hd[0]['verts3d'][side].pow(2).sum()), ('c2d.' + side, 1e-4 * hd[0]['verts2d'][side].pow(2).sum()),
                  ('scale.' + side, params['scale'][side].sum()), ('trans.' + side, params['trans2d'][side].pow(2).sum())]
        t += [('hms', 1e-3 * other['hms'].pow(2).sum()), ('mask', 1e-3 * other['mask'].abs().sum()),
              ('dense', 1e-3 * other['dense'].pow(2).sum())]
        return t

    def loss_fn(out, lab):
        holder['out'] = out
        holder['terms'] = terms_of(out)             # (extra nodes of the captured graph: they do not feed the loss)
        return scalar_loss(out)
    step = TrainStep(m, opt, loss_fn, (img.clone(), {}), force_exchange=bool(a.exchange),
                     process_group=None if a.exchange else False)
    print('config: exchange %d side %d group %d; use_graph %s stages %d side_limit %s' %
          (a.exchange, a.side, a.group, step.use_graph, step.nstage, step.side_limit), flush=True)
    first, losses, first_out = None, [], None
    for rep in range(a.reps):
        loss = step(img, {})
        torch.cuda.synchronize()
        losses.append(float(loss))
        got = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
        outs = {k: v.detach().clone() for k, v in testing.flatten_outputs(holder['out']).items()}
        with torch.no_grad():
            eager = float(scalar_loss(holder['out']))
            et = terms_of(holder['out'])
        torch.cuda.synchronize()
        tdiff = ['%s graph %.9g eager %.9g' % (n, float(a), float(b)) for (n, a), (_, b) in zip(holder['terms'], et)
                 if float(a) != float(b)]
        print('replay %d: loss in graph %.9g, recomputed eagerly from the replay\'s outputs %.9g%s' %
              (rep, float(loss), eager, ('; terms that differ: ' + '; '.join(tdiff)) if tdiff else ''), flush=True)
        if first is None:
            first, first_out = got, outs
            continue
        obad = [(float((outs[k] - first_out[k]).abs().max()) / max(float(first_out[k].abs().max()), 1e-30), k)
                for k in outs if not torch.equal(outs[k], first_out[k])]
        if obad:
            print('replay %d vs 0: OUTPUTS differ: %s' % (rep, ', '.join('%s %.3g' % (k, d) for d, k in sorted(obad, reverse=True)[:8])),
                  flush=True)
        bad = []
        for k in first:
            if not torch.equal(got[k], first[k]):
                d = float((got[k] - first[k]).abs().max()) / max(float(first[k].abs().max()), 1e-30)
                bad.append((d, k))
        print('replay %d vs 0: %d of %d tensors differ%s' % (rep, len(bad), len(first),
              ('; largest relative %.3g; first in parameter order: %s; largest: %s' %
               (max(bad)[0], ', '.join(k for _, k in bad[:4]), max(bad)[1])) if bad else ''), flush=True)
    print('losses', ' '.join('%.9g' % v for v in losses))
    if a.exchange:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
