#!/usr/bin/env python
"""Round 6: how ill conditioned is the outlier variant of tests/test_gpu_round5.py::test_captured_step_follows_the_data_magnitude
(B = 2, frozen BatchNorm statistics, one pixel of 1e4 in a unit-variance image)?  CPU only: the oracle in fp64 against (a) the oracle in
fp32 (another summation order of the same network) and (b) the oracle in fp64 on an image perturbed by 2^-22 relative -- the
representation error of an engine-2 operand.  Output kept in profiles/r06/spike_conditioning.txt."""
import sys, torch, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import net_oracle
from renderih_amd import assets, testing
from renderih_amd.model import build_model
torch.set_num_threads(8)
m = build_model(0.0); sd = testing.deterministic_state(m.state_dict(), seed=13)
graph = net_oracle.graph_from_dicts(assets.load_graph_dict('left'), assets.load_graph_dict('right'))
A = testing.seeded_image(2, 51); spike = A.clone(); spike[0,1,100,37] = 1e4
s3 = A.clone(); s3[0,1,100,37] = 1e3
for name, X in (('A', A), ('spike 1e3', s3), ('spike 1e4', spike)):
    _, g64 = net_oracle.run(sd, graph, X, False, torch.float64, True)
    _, g32 = net_oracle.run(sd, graph, X, False, torch.float32, True)
    # perturb the stem input representation like engine 2 would: relative error 2^-22 on the image -> how far do gradients move (fp64 run)?
    Xp = X.double() * (1 + (torch.rand_like(X.double()) - 0.5) * 2 ** -21)
    _, g64p = net_oracle.run(sd, graph, Xp, False, torch.float64, True)
    e32 = []; ep = []
    for k in g64:
        if testing.is_null_gradient(k): continue
        sc = float(g64[k].abs().max().clamp_min(1e-30))
        e32.append((float((g32[k].double()-g64[k]).abs().max())/sc, k)); ep.append((float((g64p[k]-g64[k]).abs().max())/sc, k))
    e32.sort(reverse=True); ep.sort(reverse=True)
    print(name, 'fp32 oracle vs fp64: worst', e32[:3], ' #>2e-4:', sum(1 for e,_ in e32 if e>2e-4), 'of', len(e32))
    print(name, 'fp64 on an image perturbed by 2^-22 relative: worst', ep[:3], ' #>2e-4:', sum(1 for e,_ in ep if e>2e-4))
