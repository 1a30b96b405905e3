#!/usr/bin/env python
"""TEST INFRASTRUCTURE: a training-mode fixture at B = 16 produced by the REAL reference modules (round-5 verdict item 4 iii).

The B = 2 fixtures of make_golden.py are as ill conditioned as this network gets (BatchNorm statistics over 128 samples at the
8x8 level); the only larger anchor so far, b64_grads.npz, is the ORACLE's fp64 run, not reference output.  This script runs the
reference's own `models.model.HandNET_GCN` (imported from /root/reference under the stubs of ref_stubs.py) in train mode,
dropout 0, at B = 16 on seeded weights / images, forward + `net_oracle.scalar_loss` backward, and stores

    out/<name>#stats, #samp        signature (renderih_amd.testing.signature) of every output of the 4-tuple      -- reference, fp32
    loss                           the scalar                                                                     -- reference, fp32
    grad/<name>#stats, #samp       signature (32 samples) of every parameter gradient                             -- reference, fp32
    bnstat/<name>                  a few BatchNorm running buffers after the step                                 -- reference, fp32
    out64/<name>#samp, grad64/<name>#samp, loss64      the same samples from the CPU oracle's fp64 run of the same inputs
                                                       (oracle pinned to the reference by tests/test_oracle_golden.py)

so that a GPU test can compare the HIP path with the REFERENCE's values directly and knows, per tensor, how far the reference's
own fp32 arithmetic is from fp64 -- without running anything on the GPU box's host.  Build container only (needs /root/reference):
    python tests/golden/make_b16_train.py           (~2 minutes on 8 cores)"""
import os
import sys
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402  (installs the stubs, imports the reference modules; nothing heavy at import)

testing, net_oracle, assets = MG.testing, MG.net_oracle, MG.assets
B, SEED_STATE, SEED_IMG, NSAMP_GRAD = 16, 3, 41, 32
BNKEYS = ('encoder.resnet.bn1.running_mean', 'encoder.resnet.bn1.running_var', 'encoder.resnet.layer4.2.bn3.running_mean',
          'encoder.resnet.layer4.2.bn3.running_var', 'mid_model.convs.1.2.running_var',
          'encoder.hms_decoder.models.2.3.running_mean')


def main():
    torch.manual_seed(0)
    model = MG.build_reference_model(dropout=0.0)
    sd = testing.deterministic_state(model.state_dict(), seed=SEED_STATE)
    model.load_state_dict(sd)
    model.train()
    img = testing.seeded_image(B, seed=SEED_IMG)
    store = {'meta_B': np.int64(B), 'meta_seed_state': np.int64(SEED_STATE), 'meta_seed_img': np.int64(SEED_IMG)}
    out = model(img)
    flat = testing.flatten_outputs(out)
    for k, v in flat.items():
        st, sa = testing.signature(v)
        store['out/' + k + '#stats'] = st
        store['out/' + k + '#samp'] = sa
    loss = net_oracle.scalar_loss(out)
    loss.backward()
    store['loss'] = np.float64(loss.item())
    names = []
    for k, p in model.named_parameters():
        if p.grad is None:
            continue
        names.append(k)
        st, sa = testing.signature(p.grad, nsamp=NSAMP_GRAD)
        store['grad/' + k + '#stats'] = st
        store['grad/' + k + '#samp'] = sa
    store['grad_names'] = np.array(names)
    nsd = model.state_dict()
    for k in BNKEYS:
        store['bnstat/' + k] = nsd[k].numpy()
    del model, out, loss

    # the oracle's fp64 run on the same state / image: the anchor that tells how far the reference's fp32 values are from exact
    graph = net_oracle.graph_from_dicts(assets.load_graph_dict('left'), assets.load_graph_dict('right'))
    w64, g64 = net_oracle.run(sd, graph, img, True, torch.float64, True)
    for k, v in w64.items():
        store['out64/' + k + '#samp'] = testing.signature(v)[1]
    assert set(g64) == set(names), sorted(set(g64) ^ set(names))[:8]
    for k in names:
        store['grad64/' + k + '#samp'] = testing.signature(g64[k], nsamp=NSAMP_GRAD)[1]
    # sanity of the fixture itself: the reference's fp32 run must sit within fp32 reach of the fp64 one
    worst = 0.0
    for k in flat:
        a, b = store['out/' + k + '#samp'].astype(np.float64), store['out64/' + k + '#samp'].astype(np.float64)
        worst = max(worst, float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)))
    print('reference fp32 vs oracle fp64, outputs: worst relative distance %.3g' % worst)
    assert worst < 5e-3, worst
    path = os.path.join(HERE, 'net_train_b16.npz')
    np.savez_compressed(path, **store)
    print('wrote', path, os.path.getsize(path), 'bytes,', len(store), 'arrays')


if __name__ == '__main__':
    main()
