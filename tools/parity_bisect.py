#!/usr/bin/env python
"""Where does the HIP path's distance from fp64 in TRAINING mode come from?  (at B = 16 the HIP forward
is 7.7e-4 from the oracle's fp64 run, the oracle's own fp32 run 4.5e-4.)  The same seeded network and images as
tests/test_gpu_bench_shapes.py::test_train_mode_forward_at_b16_vs_oracle, the HIP forward under one arithmetic switch at a
time, every error relative to the fp64 run (max over the mesh / map outputs, and per output):
    default | statistics pass instead of the GEMM epilogue | native-f32 MFMA engine | three-kernel attention | torch BatchNorm
    statistics in fp64 fed to the same apply kernels (upper bound of what better statistics could buy)
    python tools/parity_bisect.py [--batch 16] [--json out.json]"""
import argparse
import json
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import net_oracle  # noqa: E402
from renderih_amd import assets, ops, testing  # noqa: E402
from renderih_amd.model import build_model  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--json', default='')
    a = ap.parse_args()
    m = build_model(0.0)
    sd = testing.deterministic_state(m.state_dict(), seed=3)
    m.load_state_dict(sd)
    m = m.to('cuda:0').train()
    img = testing.seeded_image(a.batch, 21)
    graph = net_oracle.graph_from_dicts(assets.load_graph_dict('left'), assets.load_graph_dict('right'))
    w32, _ = net_oracle.run(sd, graph, img, True, torch.float32, False)
    w64, _ = net_oracle.run(sd, graph, img, True, torch.float64, False)
    keys = [k for k in w64 if not k.startswith('params.')]

    def err(got):
        return {k: testing.rel_err(got[k], w64[k]) for k in keys}

    rows = {'cpu fp32 oracle': err(w32)}
    settings = [('hip default', {}), ('hip, BatchNorm statistics by the separate pass', {'GEMM_STATS': False}),
                ('hip, native f32 MFMA engine', {'ENGINE': 0}), ('hip, three-kernel attention', {'FLASH_ATTN': False}),
                ('hip, f32 engine + statistics pass + three-kernel attention', {'GEMM_STATS': False, 'ENGINE': 0, 'FLASH_ATTN': False})]
    for name, kw in settings:
        saved = {k: getattr(ops, k) for k in kw}
        try:
            for k, v in kw.items():
                setattr(ops, k, v)
            m.load_state_dict(sd)           # (fresh running statistics: every run sees the same module state)
            with torch.no_grad():
                got = testing.flatten_outputs(m(img.cuda()))
            rows[name] = err(got)
        finally:
            for k, v in saved.items():
                setattr(ops, k, v)
    print('%-62s %10s   worst output' % ('B = %d, training-mode forward: rel. error vs the fp64 oracle' % a.batch, 'max'))
    for name, e in rows.items():
        wk = max(e, key=e.get)
        print('%-62s %10.2e   %s' % (name, e[wk], wk))
    print()
    print('per output: ' + ' | '.join('%s' % n[:18] for n in rows))
    for k in keys:
        print('%-28s ' % k + ' '.join('%9.2e' % rows[n][k] for n in rows))
    if a.json:
        json.dump(rows, open(a.json, 'w'), indent=1)


if __name__ == '__main__':
    main()
