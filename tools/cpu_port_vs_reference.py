#!/usr/bin/env python
"""How fast is the CPU oracle (the `port` that bench.py's cpu_baseline times on the GPU box, where /root/reference does not
exist) compared with the REAL reference modules?  Build container only: imports models.model.HandNET_GCN from /root/reference under
the stubs of tests/golden/ref_stubs.py and times forward + scalar-loss backward, train mode, B = 16 (3 warm-up +
10 timed iterations; bounded here to what fits a few minutes), then the same for oracle/net_oracle.py on the same weights, image
and thread count.  Writes profiles/r05/cpu_port_vs_reference.json; bench.py copies `reference_ratio` (port images/s divided by
reference images/s) into its cpu_baseline object.
    python tools/cpu_port_vs_reference.py [--batch 16] [--iters 5] [--threads 8]"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import make_golden as MG  # noqa: E402  (installs the stubs, imports the reference; /root/reference first on sys.path)

testing, net_oracle, assets = MG.testing, MG.net_oracle, MG.assets


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--iters', type=int, default=5)
    ap.add_argument('--threads', type=int, default=min(16, os.cpu_count() or 1))
    ap.add_argument('--out', default=os.path.join(ROOT, 'profiles', 'r05', 'cpu_port_vs_reference.json'))
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    torch.manual_seed(0)
    model = MG.build_reference_model(dropout=0.0)
    sd = testing.deterministic_state(model.state_dict(), seed=1)
    model.load_state_dict(sd)
    model.train()
    img = testing.seeded_image(a.batch, 2)

    def ref_iter():
        t0 = time.time()
        for p in model.parameters():
            p.grad = None
        net_oracle.scalar_loss(model(img)).backward()
        return time.time() - t0
    graph = net_oracle.graph_from_dicts(assets.load_graph_dict('left'), assets.load_graph_dict('right'))
    osd = {k: v.detach().clone() for k, v in sd.items()}
    for k, v in osd.items():
        if v.is_floating_point() and 'running' not in k and 'dense_coor' not in k:
            v.requires_grad_(True)

    def port_iter():
        t0 = time.time()
        out = net_oracle.handnet_forward(osd, graph, img, training=True)
        net_oracle.scalar_loss(out).backward()
        for v in osd.values():
            v.grad = None
        return time.time() - t0
    res = {}
    for name, fn in (('reference', ref_iter), ('port', port_iter), ('reference_again', ref_iter), ('port_again', port_iter)):
        fn()                                     # warm-up
        ts = [fn() for _ in range(a.iters)]
        res[name] = ts
        print('%-16s %s s per iteration' % (name, ' '.join('%.2f' % t for t in ts)), flush=True)
    med = lambda xs: sorted(xs)[len(xs) // 2]
    t_ref = med(res['reference'] + res['reference_again'])
    t_port = med(res['port'] + res['port_again'])
    out = {'batch': a.batch, 'threads': a.threads, 'host_cores': os.cpu_count(), 'iters_each': 2 * a.iters,
           'reference_images_per_sec': round(a.batch / t_ref, 3), 'port_images_per_sec': round(a.batch / t_port, 3),
           'reference_ratio': round(t_ref / t_port, 4),
           'what': 'median seconds per forward + backward (train mode, dropout 0) of the REAL reference modules '
                   '(models.model.HandNET_GCN from /root/reference under tests/golden/ref_stubs.py) and of oracle/net_oracle.py '
                   '(bench.py cpu_baseline kind "port") on the same weights, image and torch thread count, build container; '
                   'reference_ratio = port images/s / reference images/s',
           'seconds': res}
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, 'w') as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != 'seconds'}))


if __name__ == '__main__':
    main()
