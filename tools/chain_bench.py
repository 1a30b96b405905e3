#!/usr/bin/env python
"""Time the attention blocks of the mesh decoder at the benchmark's shapes (B = 64, both hands), forward and backward, with the
row-chain kernel (csrc/rih_chain.hip) and with the standalone launch sequence it replaces.
    python tools/chain_bench.py [--reps 20]"""
import argparse
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from renderih_amd import attn, ops  # noqa: E402

SHAPES = [(64, 256), (127, 256), (63, 256), (64, 128), (190, 128), (126, 128), (64, 64), (316, 64), (252, 64)]


def time_block(S, D, chain, reps, B=64, p=0.05):
    """Median time of one hipGraph replay of the block's forward, and of forward + backward (the eager launch sequence is
    bound by the host at these sizes)."""
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    mods = torch.nn.ModuleList([attn.SelfAttn(D, n_heads=4, dropout=p) for _ in range(2)]).to(dev)
    x = torch.randn(2, B, S, D, device=dev, requires_grad=True)
    gy = torch.randn(2, B, S, D, device=dev)
    ops.CHAIN = chain
    if ops.DROPOUT_SEED_TENSOR is None:
        ops.DROPOUT_SEED_TENSOR = torch.zeros(1, dtype=torch.int64, device=dev)
    params = [x] + list(mods.parameters())

    def fwd():
        return attn.SelfAttn.forward_pair(mods[0], mods[1], x, attn.DropCtx(p, True))

    def fwdbwd():
        return torch.autograd.grad([fwd()], params, [gy], allow_unused=True)

    out = []
    for fn in (fwd, fwdbwd):
        s_ = torch.cuda.Stream()
        s_.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s_):
            for _ in range(2):
                keep = fn()
        torch.cuda.current_stream().wait_stream(s_)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            keep = fn()
        ts = []
        for it in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            g.replay()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        out.append(ts[len(ts) // 2])
        del g, keep
    return out[0], out[1] - out[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--shapes', default='', help='e.g. 316x64,64x256')
    a = ap.parse_args()
    tot = {False: [0.0, 0.0], True: [0.0, 0.0]}
    shapes = [tuple(int(v) for v in t.split('x')) for t in a.shapes.split(',')] if a.shapes else SHAPES
    for S, D in shapes:
        r = {c: time_block(S, D, c, a.reps) for c in (False, True)}
        for c in r:
            tot[c][0] += r[c][0]
            tot[c][1] += r[c][1]
        print('S=%3d D=%3d rows=%6d | standalone fwd %7.1f bwd %7.1f us | chain fwd %7.1f bwd %7.1f us'
              % (S, D, 2 * 64 * S, r[False][0], r[False][1], r[True][0], r[True][1]), flush=True)
    print('sum                     | standalone fwd %7.1f bwd %7.1f us | chain fwd %7.1f bwd %7.1f us'
          % (tot[False][0], tot[False][1], tot[True][0], tot[True][1]))


if __name__ == '__main__':
    main()
